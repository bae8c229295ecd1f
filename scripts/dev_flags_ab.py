"""A/B of a PROQA_FILTER_FLAGS bit on one box (one process per setting, alternating): 2032 queries, k = 80, stream time per
search over 18M / 4.5M / 2.25M rows + digest of the result.  usage: dev_flags_ab.py <flags A> <flags B>"""
import os, subprocess, sys
sys.path.insert(0, ".")
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import time, hashlib, torch
    from proqa_amd.index import IndexFlatIP
    dev = torch.device("cuda:0")
    n, nq, k = 18_000_000, 2032, 80
    g = torch.Generator(device=dev).manual_seed(0)
    xb = torch.empty((n, 128), dtype=torch.float16, device=dev)
    for r0 in range(0, n, 2_000_000):
        xb[r0:r0 + 2_000_000] = torch.randn((2_000_000, 128), generator=g, device=dev).to(torch.float16)
    xq = torch.randn((nq, 128), generator=g, device=dev).to(torch.float16)
    out = []
    for rows in (18_000_000, 2_250_000):
        ix = IndexFlatIP(128); ix.adopt_device(xb[:rows])
        for _ in range(8): D, I = ix.search_device(xq, k)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(30): ix.search_device(xq, k)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 30 * 1e3
        h = hashlib.sha256(I.cpu().numpy().tobytes() + D.cpu().numpy().tobytes()).hexdigest()[:8]
        out.append(f"{rows // 1000}k {dt:.3f} ms {h}")
        ix.close()
    print(" | ".join(out), flush=True)
else:
    for rep in range(int(os.environ.get("AB_REPS", "3"))):
        for flags in sys.argv[1:]:
            print(f"flags={flags}: ", end="", flush=True)
            subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, PROQA_FILTER_FLAGS=flags))
