#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/full_gpu_tests.txt 2>&1
echo "rc=$?"
grep -v -i "rccl\|amdgpu.ids\|HIP version\|ROCm version\|Hostname\|Librccl\|^$" gpurun_out/full_gpu_tests.txt | tail -2
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v -i "rccl\|amdgpu.ids" | tail -1
T0=$(date +%s); python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$? seconds=$(( $(date +%s) - T0 ))"
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_default.json'))
print(d['value'], d['roofline']['frac'], d['large_k']['trec_top10000']['ms_per_search'], [round(p['ms_per_search'],3) for p in d['shard_sweep']['points']], d['encode']['value'])
PY
