"""Thread scaling of the two CPU baselines on the GPU box's host (choose a fair thread count)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from threadpoolctl import threadpool_limits, threadpool_info
from oracle import search_oracle, bert_torch_cpu
print([(i["internal_api"], i["num_threads"]) for i in threadpool_info()], "affinity", len(os.sched_getaffinity(0)))
try:
    print("cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e:
    print("no cgroup cpu.max", e)
rng = np.random.default_rng(0)
xb = rng.standard_normal((1_000_000, 128), dtype=np.float32).astype(np.float16)
xq = rng.standard_normal((2032, 128), dtype=np.float32).astype(np.float16)
fl = 2 * 2032 * 1e6 * 128 / 1e9
for n in (8, 16, 32, 64):
    with threadpool_limits(limits=n):
        t0 = time.perf_counter(); search_oracle.topk_ip(xq, xb, 80); dt = time.perf_counter() - t0
    print(f"search numpy blas threads={n}: {dt:.2f} s  ({fl/dt:.0f} GFLOP/s)")
for w in (8, 16, 32):
    t0 = time.perf_counter(); search_oracle.topk_ip_threaded(xq, xb, 80, workers=w); dt = time.perf_counter() - t0
    print(f"search threaded workers={w}: {dt:.2f} s  ({fl/dt:.0f} GFLOP/s)")
