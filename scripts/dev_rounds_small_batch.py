"""Per-round filter times of a small-batch search over 18M rows (PROQA_DEBUG_ROUNDS=1 prints them): what part of the HBM-bound
int8 scan's time is streaming and what part is the fixed cost of its eight dependent launches (dev; MI355X)."""
import os
import sys

import torch

os.environ["PROQA_DEBUG_ROUNDS"] = "1"
sys.path.insert(0, ".")
from proqa_amd.index import IndexFlatIP  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 18_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(0)
xb = torch.empty((n, 128), dtype=torch.float16, device=dev)
for r0 in range(0, n, 2_000_000):
    m = min(2_000_000, n - r0)
    xb[r0:r0 + m] = torch.randn((m, 128), generator=g, device=dev).to(torch.float16)
xq = torch.randn((nq, 128), generator=g, device=dev).to(torch.float16)
ix = IndexFlatIP(128)
ix.adopt_device(xb)
for mode in ("auto", "off"):
    ix.configure_nomination(mode)
    for _ in range(3):
        ix.search_device(xq, 80)
    ix.set_profiling(True)
    print(f"--- nomination {mode}, {nq} queries x {n} rows (bytes per row: {128 if mode == 'auto' else 256})", file=sys.stderr, flush=True)
    ix.search_device(xq, 80)
    ix.set_profiling(False)
