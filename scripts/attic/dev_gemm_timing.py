"""Developer probe: torch fp16 GEMM variants at the four bert-base shapes (M = 65536 tokens)."""
import sys
import time

import torch
import torch.nn.functional as F

dev = torch.device("cuda:0")
M = 65536
shapes = {"qkv": (768, 2304), "ao": (768, 768), "ff1": (768, 3072), "ff2": (3072, 768)}
g = torch.Generator(device=dev).manual_seed(0)


def bench(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t) / n


for name, (K, N) in shapes.items():
    x = torch.randn((M, K), generator=g, device=dev, dtype=torch.float16)
    w = (0.02 * torch.randn((N, K), generator=g, device=dev)).to(torch.float16)   # [out, in]
    wt = w.t().contiguous()                                                      # [in, out]
    b = torch.zeros(N, device=dev, dtype=torch.float16)
    out = torch.empty((M, N), device=dev, dtype=torch.float16)
    flops = 2.0 * M * K * N
    res = {
        "mm(x, Wt) out=": bench(lambda: torch.mm(x, wt, out=out)),
        "mm(x, W.t()) out=": bench(lambda: torch.mm(x, w.t(), out=out)),
        "F.linear(x, W, b)": bench(lambda: F.linear(x, w, b)),
        "addmm(b, x, Wt) out=": bench(lambda: torch.addmm(b, x, wt, out=out)),
    }
    print(name, {k: f"{v*1e6:.0f}us {flops/v/1e12:.0f}TF" for k, v in res.items()})
