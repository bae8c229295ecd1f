"""Developer timing of the k-means assign/update steps at group_paras.py's default shape."""
import sys
import time

import torch

sys.path.insert(0, ".")
from proqa_amd.group_paras import KMeans  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000
dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(0)
x = torch.empty((n, 128), dtype=torch.float16, device=dev)
for r0 in range(0, n, 2_000_000):
    m = min(2_000_000, n - r0)
    x[r0:r0 + m] = torch.randn((m, 128), generator=g, device=dev).to(torch.float16)
flops = 2.0 * n * k * 272
times = {}
for niter in (3, 13):
    km = KMeans(128, k, niter=niter, max_points_per_centroid=n // k + 1, verbose=False)
    torch.cuda.synchronize()
    t = time.time()
    km.train(x)
    torch.cuda.synchronize()
    times[niter] = time.time() - t
steady = (times[13] - times[3]) / 10
print(f"n={n} k={k}: {times[3]/3*1e3:.1f} ms per iteration over a 3-iteration run (one-time costs included), "
      f"{steady*1e3:.1f} ms per further Lloyd iteration (assign+update) = {flops/steady/1e15:.2f} PFLOP/s of assign work; "
      f"assign MFMA work {flops/1e12:.1f} TFLOP/iter")
