# Final round-end collection on the GPU box: profiles (rocprofv3 passes), default bench line, single-rank RCCL runs,
# the self-launched two-rank run (gloo: both ranks share the one GPU of the box).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=${1:-r05}
bash scripts/collect_profiles.sh ${R}_prof
O=gpurun_out/${R}_final; mkdir -p $O
python bench.py > $O/bench_default.json 2> $O/bench_default.err
cp gpurun_out/bench_full.json $O/bench_default_full.json 2>/dev/null
for t in torch cabi; do
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --force-collective --transport $t --steps 10 --skip-encode --skip-float32 --skip-cpu --skip-extras > $O/bench_rccl_1rank_$t.json 2> $O/bench_rccl_1rank_$t.err
done
PROQA_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 10 --skip-encode --skip-float32 --skip-cpu --skip-extras > $O/bench_2ranks_one_gpu_gloo.json 2> $O/bench_2ranks_one_gpu_gloo.err
python -m pytest tests -q -m gpu > $O/gpu_tests.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke()" >> $O/gpu_tests.txt 2>&1
tail -3 $O/gpu_tests.txt
