"""Leaping rounds (mips_index.cpp plan_leap): wall time of the k = 80 search under alternating schedules IN ONE PROCESS -- the
developer switches are read at every search, so the configurations are interleaved and share the box's clock state (dev; MI355X).
usage: python scripts/dev_leap_sweep.py ROWS NQ [reps] ["R:rank" ...]     (R:rank = PROQA_LEAP_ROUNDS / PROQA_LEAP_RANK, 0 = planner's)"""
import hashlib
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from proqa_amd.index import IndexFlatIP  # noqa: E402

rows, nq = int(float(sys.argv[1])), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 7
cfgs = sys.argv[4:] or ["off", "0:0", "2:0", "3:0", "4:0", "5:0", "6:0", "8:0"]
k = int(os.environ.get("AB_K", 80))
dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(0)
xb = torch.empty((rows, 128), dtype=torch.float16, device=dev)
for r0 in range(0, rows, 2_000_000):
    m = min(2_000_000, rows - r0)
    xb[r0:r0 + m] = torch.randn((m, 128), generator=g, device=dev).to(torch.float16)
xq = torch.randn((nq, 128), generator=g, device=dev).to(torch.float16)
ix = IndexFlatIP(128)
ix.adopt_device(xb)
ix.prepare()


def apply(cfg):
    for v in ("PROQA_LEAP", "PROQA_LEAP_ROUNDS", "PROQA_LEAP_RANK"):
        os.environ.pop(v, None)
    if cfg == "off":
        os.environ["PROQA_LEAP"] = "0"
    else:
        r, j = cfg.split(":")
        if int(r):
            os.environ["PROQA_LEAP_ROUNDS"] = r
        if int(j):
            os.environ["PROQA_LEAP_RANK"] = j


walls = {c: [] for c in cfgs}
info = {}
inner = 20 if nq > 256 else 60
for rep in range(reps + 1):
    for c in cfgs:
        apply(c)
        for _ in range(3):
            D, I = ix.search_device(xq, k)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(inner):
            D, I = ix.search_device(xq, k)
        torch.cuda.synchronize()
        if rep:
            walls[c].append((time.perf_counter() - t) / inner * 1e3)
        st = ix.last_stats()
        dig = hashlib.sha256(I.cpu().numpy().tobytes() + D.cpu().numpy().tobytes()).hexdigest()[:10]
        info[c] = (st["rounds"], st["nominated"] / nq, st["fallback_rounds"], st["nomination_state"], dig)
base = np.median(walls[cfgs[0]])
for c in cfgs:
    w = np.array(walls[c])
    print(f"rows={rows} nq={nq} k={k} {c:>6}: median {np.median(w):.4f} ms (min {w.min():.4f}, {np.median(w) / base - 1:+.1%} vs {cfgs[0]}) "
          f"rounds {info[c][0]} nominated/query {info[c][1]:.0f} fallback {info[c][2]} state {info[c][3]} digest {info[c][4]}")
