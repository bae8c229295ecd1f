#!/bin/bash
# kernel-time breakdown of the encode leg only (tiny search so that bench.py's encode leg dominates)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/enc_trace
rm -rf $OUT; mkdir -p $OUT
export PROQA_ENCODE_STREAMS=${PROQA_ENCODE_STREAMS:-1}
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 bench.py --skip-cpu --rows 200000 --queries 64 --steps 1 --warmup 1 --encode-steps 8 > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log | cut -c1-300
f=$(find $OUT/trace -name '*kernel_stats.csv' | head -1)
head -12 "$f" | cut -c1-160
cp "$f" $OUT/kernel_stats.csv
rm -rf $OUT/trace
