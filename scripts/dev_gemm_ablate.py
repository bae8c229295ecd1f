"""Cut experiments on the main loop of gemm_tn_f16 (PROQA_GEMM_DBG=n selects a timing-only instantiation; results are
wrong by design): what the fragment reads, the LDS-DMA stream and the epilogue each cost on top of barriers + MFMAs."""
import os, subprocess, sys
sys.path.insert(0, ".")
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import time, torch
    from proqa_amd import _lib
    lib = _lib.load(); dev = torch.device("cuda:0")
    for (N, K, name) in [(3072, 768, "ffn1"), (768, 3072, "ffn2")]:
        M = 65536
        x = torch.randn((M, K), device=dev).half(); w = (torch.randn((N, K), device=dev) * 0.02).half(); b = torch.randn(N, device=dev).half()
        y = torch.empty((M, N), dtype=torch.float16, device=dev)
        f = lambda: _lib.check(lib.proqa_gemm_tn_f16(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), M, N, K, 0, _lib.current_stream_ptr()))
        for _ in range(100): f()   # (the clocks of a fresh process take tens of ms to ramp)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): f()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        print(f"  {name}: {dt*1e6:7.1f} us  {2.0*M*N*K/dt/1e12:6.0f} TF", end="")
    print()
else:
    names = {4: "(warm-up line)", 0: "full kernel", 4: "no epilogue", 5: "no epilogue, no fragment reads", 6: "no epilogue, no DMA", 7: "no epilogue, no reads, no DMA (barriers + MFMAs)", 12: "no epilogue, every DMA from the same 64 KiB"}
    for d, nm in names.items():
        env = dict(os.environ)
        if d: env["PROQA_GEMM_DBG"] = str(d)
        print(f"DBG={d} {nm}:", end="", flush=True)
        subprocess.run([sys.executable, __file__, "child"], env=env)
