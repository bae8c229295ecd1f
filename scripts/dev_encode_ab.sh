#!/bin/bash
# A/B of an encoder environment switch on one box: the encode leg of bench.py, alternating.  usage: scripts/dev_encode_ab.sh VAR A B [rounds]
VAR=$1; A=$2; B=$3; N=${4:-2}
for i in $(seq $N); do for v in $A $B; do
  export $VAR=$v
  python3 bench.py --skip-cpu --rows 200000 --queries 64 --steps 1 --warmup 1 --encode-steps 20 --corpus-passages 0 --cli-passages 0 --skip-extras --skip-float32 2>/dev/null | python3 -c "
import json,sys
e=json.loads(sys.stdin.readline())['encode']
print('$VAR=$v', round(e['value']), 'passages/s', round(e['ms_per_step'],3), 'ms; varlen', round(e['varlen']['value']))"
done; done
