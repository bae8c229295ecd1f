#!/bin/bash
# Where an encode step's time goes on the GPU: kernel time against idle time between kernels (gpurun box, repo root).
# usage: scripts/dev_encode_gaps.sh [env assignments...]
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
OUT=gpurun_out/enc_gaps_$(echo "$*" | tr -c 'A-Za-z0-9=\n' '_')
mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o enc -- python3 bench.py --skip-cpu --rows 200000 --queries 64 --steps 1 --warmup 1 --encode-steps 6 --corpus-passages 0 --cli-passages 0 --skip-extras --skip-float32 --skip-varlen > $OUT/run.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
starts = [i for i, e in enumerate(ev) if "embed_layernorm" in e[2]]
out = open(sys.argv[1] + "/summary.txt", "w")
def say(s):
    print(s); out.write(s + "\n")
for a, b in zip(starts[2:-1], starts[3:]):          # whole steps after the warm-up
    step = ev[a:b]
    wall = ev[b][0] - step[0][0]
    busy = sum(e[1] - e[0] for e in step)
    gaps = [step[i + 1][0] - step[i][1] for i in range(len(step) - 1)] + [ev[b][0] - step[-1][1]]
    say(f"step: {len(step)} kernels, wall {wall / 1e3:8.1f} us, in kernels {busy / 1e3:8.1f} us, idle {sum(gaps) / 1e3:7.1f} us "
        f"(median gap {sorted(gaps)[len(gaps) // 2] / 1e3:.2f} us, max {max(gaps) / 1e3:.1f} us)")
a, b = starts[-2], starts[-1]
step = ev[a:b]
per = collections.defaultdict(lambda: [0, 0, 0])
for i, e in enumerate(step):
    gap_after = (step[i + 1][0] if i + 1 < len(step) else ev[b][0]) - e[1]
    k = per[e[2][:70]]
    k[0] += 1; k[1] += e[1] - e[0]; k[2] += gap_after
for name, (n, t, g) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    say(f"  {name:70s} x{n:3d}  {t / 1e3:8.1f} us  idle behind it {g / 1e3:7.1f} us")
PY
tail -1 $OUT/run.log | cut -c1-300
rm -rf $OUT/trace
