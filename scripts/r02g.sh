cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02g; mkdir -p $O
timeout 900 python -m pytest tests/test_search_gpu.py tests/test_distributed_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1
for n in 2.25e6 18e6; do
python scripts/dev_wall_timing.py $n 2032,1024,256,32,1 >> $O/wall.txt 2>&1
done
bash scripts/dev_trace_search.sh 2.25e6 2032 14 > $O/trace_2m_2032.txt 2>&1
bash scripts/dev_trace_search.sh 18e6 32 14 > $O/trace_18m_32.txt 2>&1
