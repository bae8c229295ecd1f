"""Repro of the fuzz case: adversarially ordered corpus (scores rise with the row number) under the int8 nomination scan."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from oracle import search_oracle  # noqa: E402
from proqa_amd.index import IndexFlatIP  # noqa: E402

dev = torch.device("cuda", 0)
rng = np.random.default_rng(3)
for n, nq, k in ((66667, 257, 2), (200000, 300, 80)):
    xb = rng.integers(-1, 2, (n, 128)).astype(np.float16)
    xb[:, 0] = np.minimum(np.arange(n) // 7, 2000)
    xq = rng.integers(0, 2, (nq, 128)).astype(np.float16)
    xq[:, 0] = 1
    Do, Io = search_oracle.topk_ip(xq, xb, k)
    for mode in (0, 2):
        ix = IndexFlatIP(128)
        ix.configure_nomination(mode)
        ix.add(xb)
        D, I = ix.search_device(torch.from_numpy(xq).to(dev), k)
        D, I = D.cpu().numpy(), I.cpu().numpy()
        st = ix.last_stats()
        ok = (I == Io).all() and (D == Do).all()
        print(f"n={n} nq={nq} k={k} mode={mode}: ok={ok} stats={st}")
        if not ok:
            bad = np.argwhere(I != Io)
            print("   first mismatches", bad[:4].tolist(), I[bad[0][0]][:6], Io[bad[0][0]][:6], D[bad[0][0]][:6], Do[bad[0][0]][:6])
