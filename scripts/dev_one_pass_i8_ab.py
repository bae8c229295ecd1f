"""One question (or a few), k in the thousands, over ROWS rows: the one-pass search with its launch over the shard on the int8
copy (default) or on the fp16 rows (PROQA_ONE_PASS_I8=0, read once: run twice).  dev; MI355X.
usage: [PROQA_ONE_PASS_I8=0] python scripts/dev_one_pass_i8_ab.py [rows] [nq,k ...]"""
import hashlib
import os
import sys
import time

import torch

sys.path.insert(0, ".")
from proqa_amd.index import IndexFlatIP  # noqa: E402

rows = int(float(sys.argv[1])) if len(sys.argv) > 1 else 18_000_000
cases = [tuple(int(v) for v in a.split(",")) for a in sys.argv[2:]] or [(1, 5000), (1, 1000), (8, 5000), (32, 5000), (32, 1000)]
dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(0)
xb = torch.empty((rows, 128), dtype=torch.float16, device=dev)
for r0 in range(0, rows, 2_000_000):
    m = min(2_000_000, rows - r0)
    xb[r0:r0 + m] = torch.randn((m, 128), generator=g, device=dev).to(torch.float16)
ix = IndexFlatIP(128)
ix.adopt_device(xb)
ix.prepare()
tag = "fp16 launch" if os.environ.get("PROQA_ONE_PASS_I8") == "0" else "int8 launch"
for nq, k in cases:
    xq = torch.randn((nq, 128), generator=g, device=dev).to(torch.float16)
    for _ in range(3):
        D, I = ix.search_device(xq, k)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(30):
        D, I = ix.search_device(xq, k)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / 30 * 1e3
    st = ix.last_stats()
    dig = hashlib.sha256(I.cpu().numpy().tobytes() + D.cpu().numpy().tobytes()).hexdigest()[:10]
    print(f"[{tag}] rows={rows} nq={nq} k={k}: {ms:.3f} ms per search, rounds {st['rounds']} fallback {st['fallback_rounds']} "
          f"nominated/query {st['nominated'] / nq:.0f} candidates/query {st['candidates'] / nq:.0f} digest {dig}")
