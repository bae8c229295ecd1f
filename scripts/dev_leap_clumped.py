"""Leaping rounds on a corpus whose relevant rows arrive in CLUMPS (dev; MI355X): `topics` of `clump` contiguous rows around a
common direction (passages of one article next to each other, articles in no particular order), queries near topic
directions.  The same rows in file order and shuffled; ordinary rounds, leaping rounds (automatic: with its pauses), and
leaping rounds with the pause cleared before every search (what every leap costs).
usage: python scripts/dev_leap_clumped.py [rows] [clump] [nq]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from proqa_amd.index import IndexFlatIP  # noqa: E402

rows = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2_250_000
clump = int(sys.argv[2]) if len(sys.argv) > 2 else 50
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 2032
k = 80
dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(0)
topics = rows // clump
cent = torch.randn((topics, 128), generator=g, device=dev)
xb = torch.empty((topics * clump, 128), dtype=torch.float16, device=dev)
for t0 in range(0, topics, 20000):
    t1 = min(topics, t0 + 20000)
    c = cent[t0:t1].repeat_interleave(clump, dim=0)
    xb[t0 * clump:t1 * clump] = (0.7 * c + 0.7 * torch.randn(c.shape, generator=g, device=dev)).to(torch.float16)
qt = torch.randint(0, topics, (nq,), generator=g, device=dev)
xq = (0.7 * cent[qt] + 0.7 * torch.randn((nq, 128), generator=g, device=dev)).to(torch.float16)
perm = torch.randperm(xb.shape[0], generator=g, device=dev)
for name, data in (("file order (clumps of %d)" % clump, xb), ("shuffled", xb[perm].contiguous())):
    ix = IndexFlatIP(128)
    ix.adopt_device(data)
    ix.prepare()
    res = {}
    for mode in ("off", "auto", "auto, pause cleared"):
        ix.configure_leap("off" if mode == "off" else "auto")
        for _ in range(3):
            ix.search_device(xq, k)
        short = leaps = 0
        torch.cuda.synchronize()
        t = time.perf_counter()
        n_it = 40
        for _ in range(n_it):
            if mode.endswith("cleared"):
                ix.configure_leap("auto")
            D, I = ix.search_device(xq, k)
            st = ix.last_stats()
            leaps += st["leap_rank"] > 0
            short += st["fallback_rounds"] > 0
        torch.cuda.synchronize()
        res[mode] = ((time.perf_counter() - t) / n_it * 1e3, leaps, short, I.clone() if data is xb else perm[I.clamp(min=0)])
    same = all(bool((torch.sort(res[m][3], dim=1).values == torch.sort(res["off"][3], dim=1).values).all()) for m in res)
    print(f"rows={xb.shape[0]} nq={nq} {name}: " + "; ".join(f"{m}: {v[0]:.3f} ms, leapt {v[1]}/40, fell short {v[2]}/40" for m, v in res.items())
          + f"; same ids: {same}")
    ix.close()
