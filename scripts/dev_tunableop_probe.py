"""Does ANY library kernel beat the default choice on the encoder's dense shapes?  torch's TunableOp benchmarks every
hipBLASLt / rocBLAS solution for a shape and keeps the fastest: default vs tuned time of torch.mm / addmm (dev probe)."""
import os, sys, time
import torch
dev = torch.device("cuda:0")
M = 65536
def timeit(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
shapes = [(2304, 768, "qkv"), (768, 768, "attn-out"), (3072, 768, "ffn1"), (768, 3072, "ffn2")]
ops = {}
for (N, K, name) in shapes:
    x = torch.randn((M, K), device=dev).half(); w = (torch.randn((N, K), device=dev) * 0.02).half(); b = torch.randn(N, device=dev).half()
    ops[name] = (x, w, b)
base = {n: (timeit(lambda: torch.mm(ops[n][0], ops[n][1].t())), timeit(lambda: torch.addmm(ops[n][2], ops[n][0], ops[n][1].t()))) for n in ops}
torch.cuda.tunable.enable(True)
torch.cuda.tunable.set_max_tuning_duration(2000)
torch.cuda.tunable.set_max_tuning_iterations(50)
for n in ops:
    torch.mm(ops[n][0], ops[n][1].t()); torch.addmm(ops[n][2], ops[n][0], ops[n][1].t())   # tunes
torch.cuda.synchronize()
for n in ops:
    t_mm = timeit(lambda: torch.mm(ops[n][0], ops[n][1].t())); t_add = timeit(lambda: torch.addmm(ops[n][2], ops[n][0], ops[n][1].t()))
    print(f"{n:8s} mm default {base[n][0]*1e6:7.1f} us  tuned {t_mm*1e6:7.1f} us | addmm default {base[n][1]*1e6:7.1f} us  tuned {t_add*1e6:7.1f} us")
try:
    for r in torch.cuda.tunable.get_results(): print(r)
except Exception as e:
    print("results:", e)
