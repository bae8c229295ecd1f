#!/bin/bash
# Per-kernel averages of the encode leg (gpurun box, repo root).  usage: scripts/dev_encode_kernels.sh [env assignments...]
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
OUT=gpurun_out/enc_kernels_$(echo "$*" | tr -c 'A-Za-z0-9=\n' '_')
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o enc -- python3 bench.py --skip-cpu --rows 200000 --queries 64 --steps 1 --warmup 1 --encode-steps 5 --corpus-passages 0 --cli-passages 0 --skip-extras --skip-float32 --skip-varlen > $OUT/run.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
with open(sys.argv[1] + "/summary.txt", "w") as out:
    for r in rows[:9]:
        line = f'{r["Name"][:90]:90s} calls {r["Calls"]:>5s}  total {float(r["TotalDurationNs"]) / 1e6:8.2f} ms  avg {float(r["AverageNs"]) / 1e3:9.1f} us'
        print(line); out.write(line + "\n")
PY
rm -rf $OUT/trace
