"""Wall and filter time of the headline search (2032 queries, k = 80, default schedule) over the first N rows of the bench
corpus, for A/B by environment across processes (dev; MI355X).  usage: [ENV=...] [AB_NQ=2032] [AB_K=80] python scripts/dev_headline_ab.py [rows ...]"""
import hashlib
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from proqa_amd.index import IndexFlatIP  # noqa: E402

rows_list = [int(float(a)) for a in sys.argv[1:]] or [18_000_000, 2_250_000]
nq, k = int(os.environ.get("AB_NQ", 2032)), int(os.environ.get("AB_K", 80))
dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(0)
n = max(rows_list)
xb = torch.empty((n, 128), dtype=torch.float16, device=dev)
for r0 in range(0, n, 2_000_000):
    m = min(2_000_000, n - r0)
    xb[r0:r0 + m] = torch.randn((m, 128), generator=g, device=dev).to(torch.float16)
xq = torch.randn((nq, 128), generator=g, device=dev).to(torch.float16)
tag = " ".join(f"{k_}={v}" for k_, v in sorted(os.environ.items()) if k_.startswith("PROQA_")) or "default"
for rows in rows_list:
    ix = IndexFlatIP(128)
    ix.adopt_device(xb[:rows])
    for _ in range(4):
        D, I = ix.search_device(xq, k)
    ix.set_profiling(True)
    filt = []
    for _ in range(5):
        ix.search_device(xq, k)
        filt.append(ix.last_stats()["filter_ms"])
    ix.set_profiling(False)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(30):
        D, I = ix.search_device(xq, k)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t) / 30 * 1e3
    st = ix.last_stats()
    dig = hashlib.sha256(I.cpu().numpy().tobytes() + D.cpu().numpy().tobytes()).hexdigest()[:12]
    print(f"[{tag}] rows={rows}: wall {wall:.3f} ms ({nq / wall:.1f} k q/s), filter {np.median(filt):.3f} ms, chain {wall - np.median(filt):.3f} ms, "
          f"rounds {st['rounds']} nominated/query {st['nominated'] / nq:.0f} fallback {st['fallback_rounds']} digest {dig}")
    ix.close()
