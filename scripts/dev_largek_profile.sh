#!/bin/bash
# Kernel statistics of the top-10000 search of 6980 queries over 8.8M rows (gpurun box, repo root).
# usage: scripts/dev_largek_profile.sh [0|1]   (PROQA_ONE_PASS_COMPACT)
export TMPDIR=/tmp
export PROQA_ONE_PASS_COMPACT=${1:-1}
OUT=gpurun_out/largek_prof_$PROQA_ONE_PASS_COMPACT
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o lk -- python3 scripts/dev_largek_timing.py 6980 > $OUT/run.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
with open(sys.argv[1] + "/summary.txt", "w") as out:
    for r in rows[:8]:
        line = f'{r["Name"][:100]:100s} calls {r["Calls"]:>5s}  total {float(r["TotalDurationNs"]) / 1e6:8.2f} ms  avg {float(r["AverageNs"]) / 1e3:9.1f} us'
        print(line); out.write(line + "\n")
PY
tail -1 $OUT/run.log
rm -rf $OUT/trace
