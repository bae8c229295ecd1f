#!/bin/bash
# kernel-time breakdown of the k-means Lloyd iterations (dev)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/kmeans_trace
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o k -- python3 scripts/dev_kmeans_timing.py ${1:-10e6} ${2:-10000} > $OUT/log.txt 2>&1
tail -2 $OUT/log.txt
f=$(find $OUT/t -name '*kernel_stats.csv' | head -1)
head -12 "$f" | cut -c1-170
rm -rf $OUT/t
