"""HBM-bound batches of the int8 nomination scan: filter time, wall time and result digest per batch size (dev; MI355X).
A/B by environment, one process per setting:  PROQA_I8_DEEP_RING=0|1 python scripts/dev_small_batch_ab.py [rows] [k]"""
import hashlib
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from proqa_amd.index import IndexFlatIP  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 18_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 80
dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(0)
xb = torch.empty((n, 128), dtype=torch.float16, device=dev)
for r0 in range(0, n, 2_000_000):
    m = min(2_000_000, n - r0)
    xb[r0:r0 + m] = torch.randn((m, 128), generator=g, device=dev).to(torch.float16)
ix = IndexFlatIP(128)
ix.adopt_device(xb)
tag = f"row_split={os.environ.get('PROQA_I8_ROW_SPLIT', 'default')} deep_ring={os.environ.get('PROQA_I8_DEEP_RING', 'default')}"
for nq in (1, 32, 64, 128, 256):
    xq = torch.randn((nq, 128), generator=g, device=dev).to(torch.float16)
    for _ in range(3):
        D, I = ix.search_device(xq, k)
    ix.set_profiling(True)
    filt = []
    for _ in range(5):
        ix.search_device(xq, k)
        filt.append(ix.last_stats()["filter_ms"])
    ix.set_profiling(False)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(20):
        D, I = ix.search_device(xq, k)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t) / 20
    st = ix.last_stats()
    dig = hashlib.sha256(I.cpu().numpy().tobytes() + D.cpu().numpy().tobytes()).hexdigest()[:12]
    f = float(np.median(filt))
    print(f"{tag} rows={n} nq={nq:4d} k={k}: filter {f:.4f} ms = {n * 128 / f / 1e6:7.1f} GB/s of int8 rows ({n * 128 / f / 1e6 / 8000:.3f} of 8 TB/s), "
          f"wall {wall * 1e3:.4f} ms, nomination={st['nomination']} rounds={st['rounds']} fallback={st['fallback_rounds']} digest {dig}")
