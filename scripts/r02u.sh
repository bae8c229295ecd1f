cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02u; mkdir -p $O
for b in 64 128 256 512 1024; do
python bench.py --rows 400000 --queries 64 --steps 2 --warmup 1 --skip-float32 --skip-cpu --corpus-passages 0 --encode-batch $b --encode-steps 16 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])['encode']
print('batch', $b, 'passages/s', round(d['value']), 'ms/step', round(d['ms_per_step'],3), 'varlen', round(d['varlen']['value']))
" >> $O/batch_sweep.txt
done
