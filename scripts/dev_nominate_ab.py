"""Developer A/B of the int8 nomination scan against the fp16 scan: same index, same queries, ids and scores compared bit for
bit, then timed (not part of the test suite).  usage: dev_nominate_ab.py [rows] [queries] [k] [dist]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from proqa_amd.index import IndexFlatIP  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 18_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 2032
k = int(sys.argv[3]) if len(sys.argv) > 3 else 80
dist = sys.argv[4] if len(sys.argv) > 4 else "normal"
dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(0)
xb = torch.empty((n, 128), dtype=torch.float16, device=dev)
for r0 in range(0, n, 2_000_000):
    m = min(2_000_000, n - r0)
    if dist == "int":
        xb[r0:r0 + m] = torch.randint(-4, 5, (m, 128), generator=g, device=dev).to(torch.float16)
    else:
        xb[r0:r0 + m] = torch.randn((m, 128), generator=g, device=dev).to(torch.float16)
if dist == "int":
    xq = torch.randint(-4, 5, (nq, 128), generator=g, device=dev).to(torch.float16)
else:
    xq = torch.randn((nq, 128), generator=g, device=dev).to(torch.float16)
if dist == "shifted":      # a large common component: what the centring is for
    xb += 20.0
ix = IndexFlatIP(128)
ix.adopt_device(xb)
ix.set_profiling(True)
res = {}
for mode in (0, 2):
    ix.configure_nomination(mode)
    t = time.time()
    D, I = ix.search_device(xq, k)
    torch.cuda.synchronize()
    first = time.time() - t
    for _ in range(2):
        ix.search_device(xq, k)
    torch.cuda.synchronize()
    reps = 10
    best = 1e9
    t = time.time()
    for _ in range(reps):
        D, I = ix.search_device(xq, k)
        best = min(best, ix.last_stats()["filter_ms"])
    torch.cuda.synchronize()
    dt = (time.time() - t) / reps
    st = ix.last_stats()
    res[mode] = (D.clone(), I.clone())
    flops = 2.0 * nq * n * 128
    print(f"mode={mode} first call {first*1e3:.1f} ms; wall={dt*1e3:.3f} ms q/s={nq/dt:.0f} rounds={st['rounds']} fallback={st['fallback_rounds']} "
          f"cand/q={st['candidates']/nq:.0f} nominated/q={st['nominated']/nq:.0f} nomination={st['nomination']} "
          f"filter_ms={st['filter_ms']:.3f} (best {best:.3f}) total_ms={st['total_ms']:.3f} algorithmic PF/s={flops/st['filter_ms']/1e12:.3f}")
same_i = bool((res[0][1] == res[2][1]).all())
same_d = bool((res[0][0].view(torch.int32) == res[2][0].view(torch.int32)).all())
print(f"ids identical: {same_i}; scores bit-identical: {same_d}")
if not (same_i and same_d):
    bad = (res[0][1] != res[2][1]).any(dim=1).nonzero().flatten()
    print("queries that differ:", bad[:10].tolist(), "of", bad.numel())
    q = int(bad[0]) if bad.numel() else 0
    print("fp16 :", res[0][1][q, :12].tolist(), res[0][0][q, :6].tolist())
    print("int8 :", res[2][1][q, :12].tolist(), res[2][0][q, :6].tolist())
    sys.exit(1)
