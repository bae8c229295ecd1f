cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02c; mkdir -p $O
timeout 900 python -m pytest tests/test_search_gpu.py tests/test_distributed_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1
bash scripts/dev_trace_search.sh 2.25e6 2032 18 > $O/trace_2m_2032.txt 2>&1
bash scripts/dev_trace_search.sh 18e6 32 14 > $O/trace_18m_32.txt 2>&1
python scripts/dev_search_timing.py 2.25e6 2032 256,4 1024,8 1920,8 1920,6 512,8 > $O/timing_2m.txt 2>&1
python scripts/dev_search_timing.py 18e6 2032 256,4 1024,8 1920,8 1920,6 512,8 > $O/timing_18m.txt 2>&1
python scripts/dev_search_timing.py 18e6 32 256,8 1024,8 1920,8 1920,16 > $O/timing_18m_32.txt 2>&1
