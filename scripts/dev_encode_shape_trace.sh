#!/bin/bash
# per-kernel averages of one encode shape (gpurun box, repo root).  usage: scripts/dev_encode_shape_trace.sh [batch] [seq_len]
export TMPDIR=/tmp PYTHONPATH=$PWD
OUT=gpurun_out/enc_shape_${1:-64}x${2:-512}
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o enc -- python3 scripts/dev_encode_shape.py ${1:-64} ${2:-512} 5 > $OUT/run.log 2>&1
tail -1 $OUT/run.log | grep passages
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open(sys.argv[1] + "/summary.txt", "w") as out:
    for r in rows[:10]:
        line = f'{r["Name"][:80]:80s} calls {r["Calls"]:>5s}  total {float(r["TotalDurationNs"]) / 1e6:8.2f} ms ({100 * float(r["TotalDurationNs"]) / tot:4.1f} %)  avg {float(r["AverageNs"]) / 1e3:9.1f} us'
        print(line); out.write(line + "\n")
PY
rm -rf $OUT/trace
