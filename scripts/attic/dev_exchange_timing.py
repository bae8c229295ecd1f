"""Cost of the sharded search's exchange + merge with one rank (RCCL): whole step minus the local search, for the
in-place torch.distributed path, the staged torch path and the library's own communicator.  The variants are timed in
interleaved blocks (the difference of two ~1.2 ms figures drifts by more than it measures otherwise); medians are
reported.  Run under torch.distributed.run --nproc-per-node 1."""
import statistics, sys, time
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
fns = {"local": lambda: variants["in place"].local_index.search_device(xq, k)}
for name, s in variants.items():
    fns[name] = (lambda s=s: s.search(xq, k, force_collective=True))
def block(fn, reps=40):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3
for fn in fns.values():
    for _ in range(10): fn()
samples = {name: [] for name in fns}
for rnd in range(12):
    for name, fn in fns.items(): samples[name].append(block(fn))
base = statistics.median(samples["local"])
print(f"local search {base:.4f} ms (median of 12 interleaved blocks of 40)")
for name in variants:
    t = statistics.median(samples[name]); d = statistics.median([a - b for a, b in zip(samples[name], samples["local"])])
    print(f"  {name:9s}: {t:.4f} ms  (+{d * 1e3:.1f} us, median of the paired differences)")
dist.destroy_process_group()
