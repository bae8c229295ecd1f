#!/bin/bash
# per-kernel durations of the call-by-call and the pipelined phase of scripts/dev_search_pipelined.py (18M rows only)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/trace_pipelined
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o s -- python3 scripts/dev_search_pipelined.py 18000000 > $OUT/log.txt 2>&1
f=$(find $OUT/t -name '*kernel_trace.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "proqa" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# searches are delimited by the prep_queries launch
starts = [i for i, r in enumerate(rows) if "prep_queries" in r["Kernel_Name"]]
searches = [rows[a:b] for a, b in zip(starts, starts[1:] + [len(rows)])]
def summarize(group, label):
    per = collections.defaultdict(list); spans = []; busy = []
    for s in group:
        t0 = int(s[0]["Start_Timestamp"]); t1 = int(s[-1]["End_Timestamp"])
        spans.append((t1 - t0) / 1e3); busy.append(sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in s) / 1e3)
        for j, r in enumerate(s):
            per[(j, r["Kernel_Name"][35:58])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print(f"{label}: {len(group)} searches, span {sum(spans)/len(spans):.1f} us, sum of kernel durations {sum(busy)/len(busy):.1f} us")
    return {k: sum(v) / len(v) for k, v in per.items()}
n = len(searches)
a = summarize(searches[10:40], "call by call")
b = summarize(searches[50:85], "pipelined  ")
for key in sorted(a):
    if key in b: print(f"  launch {key[0]:2d} {key[1]:24s} {a[key]:9.1f} us  vs {b[key]:9.1f} us")
PY
tail -2 $OUT/log.txt | cut -c1-250
rm -rf $OUT/t
