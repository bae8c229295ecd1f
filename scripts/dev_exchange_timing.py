"""Cost of the sharded search's exchange + merge with one rank (RCCL): whole step minus the local search, for the
in-place path, the staged torch path and the library's own communicator.  Run under torch.distributed.run --nproc-per-node 1."""
import os, sys, time
import torch
import torch.distributed as dist
sys.path.insert(0, ".")
from proqa_amd.index import ShardedIndexFlatIP, merge_topk_device
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group(backend="nccl", device_id=dev)
n, nq, k = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2_250_000, 2032, 80
g = torch.Generator(device=dev).manual_seed(0)
xb = torch.randn((n, 128), generator=g, device=dev).half(); xq = torch.randn((nq, 128), generator=g, device=dev).half()
def make(transport, staged=False):
    s = ShardedIndexFlatIP(n, preallocate=False, transport=transport); s.adopt_local(xb)
    if staged: s._merge = lambda D, I: merge_topk_device(D, I)      # any other callable: the staged path
    return s
variants = {"in place": make("torch"), "staged": make("torch", True), "cabi": make("cabi")}
def timeit(fn, reps=200):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3
for rnd in range(2):
    base = timeit(lambda: variants["in place"].local_index.search_device(xq, k))
    print(f"local search {base:.4f} ms")
    for name, s in variants.items():
        t = timeit(lambda: s.search(xq, k, force_collective=True))
        print(f"  {name:9s}: {t:.4f} ms  (+{(t - base) * 1e3:.1f} us)")
dist.destroy_process_group()
