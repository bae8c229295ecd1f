"""Round-schedule sweep (dev): bootstrap rows x growth cap, equal-growth rounds on / off (PROQA_EQUAL_GROWTH is read when the
library loads: run once per setting), stream time of a 2032-query search at several shard sizes."""
import os, sys
import torch
sys.path.insert(0, ".")
from proqa_amd.index import IndexFlatIP
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(0)
N = 18_000_000
xb = torch.empty((N, 128), dtype=torch.float16, device=dev)
for r0 in range(0, N, 2_000_000):
    xb[r0:r0 + 2_000_000] = torch.randn((2_000_000, 128), generator=g, device=dev).to(torch.float16)
xq = torch.randn((2032, 128), generator=g, device=dev).to(torch.float16)
sizes = [int(float(v)) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [2_250_000, 4_500_000, 9_000_000, 18_000_000]
growths = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [4, 5]
nqs = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [2032]
xq_all = xq
for nq in nqs:
  xq = xq_all[:nq]
  for n in sizes:
    for boot in (4096, 8192):
      for growth in growths:
            ix = IndexFlatIP(128); ix.adopt_device(xb[:n]); ix.configure(256, growth); ix.configure_bootstrap(boot)
            for _ in range(3): ix.search_device(xq, 80)
            ts = []
            for _ in range(12):
                ix.search_device(xq, 80); ts.append(ix.last_stats()["total_ms"])
            st = ix.last_stats()
            print(f"equal={os.environ.get('PROQA_EQUAL_GROWTH','1')} queries {nq} rows {n} boot {boot} cap {growth}: stream {sorted(ts)[len(ts)//2]:.3f} ms rounds {st['rounds']} fallback {st['fallback_rounds']} cand/q {st['candidates']/nq:.0f} nominated/q {st['nominated']/nq:.0f}")
            ix.close()
