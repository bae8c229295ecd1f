#!/bin/bash
# per-launch timeline of one search (dev): bash scripts/dev_trace_search.sh <rows> <queries>
set -u
export TMPDIR=/tmp
OUT=gpurun_out/trace_search
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o s -- python3 scripts/dev_search_timing.py $1 $2 256,0 > $OUT/log.txt 2>&1
f=$(find $OUT/t -name '*kernel_trace.csv' | head -1)
python3 scripts/trace_tail.py "$f" ${3:-17}
rm -rf $OUT/t
