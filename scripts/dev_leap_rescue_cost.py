"""What a leap that falls short costs (dev; MI355X): 2032 queries over ROWS random rows, S of them with their best k rows planted
among the first rows (their lists are final after the bootstrap: every leaping round leaves exactly those queries short).
Timed: ordinary rounds; leaping rounds + the short queries searched again by themselves (the pause is cleared before every
search so that every search leaps); the same with PROQA_LEAP_RESCUE_MAX=0 in the environment = the flagged slabs re-scanned
for all queries.  usage: python scripts/dev_leap_rescue_cost.py [rows] [S ...]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from proqa_amd.index import IndexFlatIP  # noqa: E402

rows = int(float(sys.argv[1])) if len(sys.argv) > 1 else 18_000_000
shorts = [int(a) for a in sys.argv[2:]] or [0, 1, 8, 64, 256]
nq, k = 2032, 80
dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(0)
xb = torch.empty((rows, 128), dtype=torch.float16, device=dev)
for r0 in range(0, rows, 2_000_000):
    m = min(2_000_000, rows - r0)
    xb[r0:r0 + m] = torch.randn((m, 128), generator=g, device=dev).to(torch.float16)
xq = torch.randn((nq, 128), generator=g, device=dev).to(torch.float16)
head = xb[:4096].clone()
for s in shorts:
    xb[:4096] = head
    qs = torch.linspace(0, nq - 1, max(s, 1)).long()[:s]
    # k rows per short query, interleaved over the first rows (s * k <= 4096 * 5: several queries may share none)
    for i, q in enumerate(qs.tolist()):
        r = torch.arange(k, device=dev) * max(s, 1) + i
        r = r[r < 4096]
        xb[r] = (1.2 * xq[q].float() + 0.02 * torch.randn((r.numel(), 128), generator=g, device=dev)).to(torch.float16)
    ix = IndexFlatIP(128)
    ix.adopt_device(xb)
    ix.prepare()
    out = {}
    for mode in ("off", "auto"):
        ix.configure_leap(mode)
        for _ in range(3):
            ix.configure_leap(mode)
            D, I = ix.search_device(xq, k)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(20):
            ix.configure_leap(mode)   # (clears the pause the last search earned)
            D, I = ix.search_device(xq, k)
        torch.cuda.synchronize()
        out[mode] = ((time.perf_counter() - t) / 20 * 1e3, ix.last_stats(), I.clone())
    same = bool((out["off"][2] == out["auto"][2]).all())
    st = out["auto"][1]
    print(f"rows={rows} short queries={s} rescue_max={os.environ.get('PROQA_LEAP_RESCUE_MAX', '256')}: ordinary {out['off'][0]:.3f} ms, leaping "
          f"{out['auto'][0]:.3f} ms ({out['auto'][0] / out['off'][0] - 1:+.1%}) rank {st['leap_rank']} rounds {st['rounds']} "
          f"re-run {st['fallback_rounds']} ids equal {same}")
    ix.close()
