"""Which rows does the int8 scan lose?  Emulates the quantisation in torch and checks the lost rows against the bound."""
import sys

import torch

sys.path.insert(0, ".")
from proqa_amd.index import IndexFlatIP  # noqa: E402

n, nq, k = 1_000_000, 2032, 80
dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(0)
xb = torch.randn((n, 128), generator=g, device=dev).to(torch.float16)
xq = torch.randn((nq, 128), generator=g, device=dev).to(torch.float16)
ix = IndexFlatIP(128)
ix.adopt_device(xb)
ix.configure_nomination(0)
D0, I0 = ix.search_device(xq, k)
ix.configure_nomination(2)
D1, I1 = ix.search_device(xq, k)
print("stats", ix.last_stats())
x = xb.float()
mu = x.mean(dim=0)
c = (x - mu).abs().max(dim=0).values
xi = torch.clamp(torch.round((x - mu) * (127.0 / c)), -127, 127)
w = xq.float() * (c / 127.0)
s_q = w.abs().max(dim=1).values / 127.0
qi = torch.clamp(torch.round(w / s_q[:, None]), -127, 127)
lost_total = 0
for q in range(0, 8):
    a = set(I0[q].tolist())
    b = set(I1[q].tolist())
    lost = sorted(a - b)
    lost_total += len(lost)
    rows = torch.tensor(lost, device=dev, dtype=torch.long)
    acc = (xi[rows] @ qi[q]).tolist()
    exact = (x[rows] @ xq[q].float()).tolist()
    approx = [(float(mu @ xq[q].float()) + float(s_q[q]) * v) for v in acc]
    kept = sorted(a & b)
    print(f"q={q}: lost {len(lost)} of {k}; tau(final)={float(D0[q, -1]):.3f}")
    for r, e, ap in list(zip(lost, exact, approx))[:6]:
        print(f"   lost row {r} (mod 128 = {r % 128}, mod 32 = {r % 32}, chunk-rel?): exact {e:.3f}, int8 approx {ap:.3f}")
    print("   kept rows mod 32:", sorted(set(r % 32 for r in kept))[:32])
    print("   lost rows mod 32:", sorted(r % 32 for r in lost))
print("lost in 8 queries:", lost_total)
# emulated nominations of the first int8 round for queries 0..7 (slab printed by PROQA_DEBUG_CAND; thresholds = 80th best of the bootstrap rows)
import os
r0 = int(os.environ.get("R0", "8192"))
r1 = int(os.environ.get("R1", "27264"))
S_boot = x[:r0] @ xq[:8].float().T
tau = S_boot.topk(k, dim=0).values[-1]
mu_q = xq[:8].float() @ mu
acc = xi[r0:r1] @ qi[:8].T
Rn = ((x - mu) * (127.0 / c) - xi).norm(dim=1).max()
Xn = xi.norm(dim=1).max()
u = w[:8] / s_q[:8, None]
M = u.norm(dim=1) * Rn + (u - qi[:8]).norm(dim=1) * Xn
T = torch.floor((tau - mu_q) / s_q[:8] - M - 1.0)
print("emulated nominations of the first round, q0..7:", (acc > T[None, :]).sum(dim=0).tolist(), " exact candidates:", ((x[r0:r1] @ xq[:8].float().T) > tau[None, :]).sum(dim=0).tolist())
