"""Idle time between kernels in a rocprofv3 kernel-trace CSV (dev helper): python dev_trace_gaps.py trace.csv [skip_fraction]
Reports, for the last part of the trace, busy time vs span and the largest gaps with the kernels either side."""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[int(len(rows) * skip):]
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
print(f"{len(rows)} launches, span {span/1e6:.3f} ms, kernel time {busy/1e6:.3f} ms, idle {(span-busy)/1e6:.3f} ms ({100*(span-busy)/span:.1f} %)")
gaps = defaultdict(lambda: [0, 0])
for a, b in zip(rows[:-1], rows[1:]):
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    key = (a["Kernel_Name"][:48], b["Kernel_Name"][:48])
    gaps[key][0] += g
    gaps[key][1] += 1
for key, (tot, cnt) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"{tot/1e3:10.1f} us in {cnt:5d} gaps (avg {tot/cnt/1e3:7.2f} us)  {key[0]}  ->  {key[1]}")
