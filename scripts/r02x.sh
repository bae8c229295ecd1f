cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02x; mkdir -p $O; rm -f $O/*.txt
timeout 1200 python -m pytest tests/test_encoder_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1
for i in 1 2; do
python bench.py --rows 400000 --queries 64 --steps 2 --warmup 1 --skip-float32 --skip-cpu --corpus-passages 0 --encode-steps 16 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])['encode']
print('passages/s', round(d['value']), 'ms/step', round(d['ms_per_step'],3), 'varlen', round(d['varlen']['value']), d['varlen']['mean_len'])
" >> $O/encode.txt
done
