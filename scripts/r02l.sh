cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02l; mkdir -p $O
PROQA_FILTER_QW=4 timeout 900 python -m pytest tests/test_search_gpu.py -x -q -m gpu > $O/pytest_qw4.txt 2>&1
for i in 1 2; do
python scripts/dev_search_timing.py 18e6 2032 256,0 2>&1 | grep variant >> $O/timing.txt
PROQA_FILTER_QW=4 python scripts/dev_search_timing.py 18e6 2032 256,0 2>&1 | grep variant | sed 's/^/QW4 /' >> $O/timing.txt
PROQA_DEBUG_NOHIT=1 python scripts/dev_search_timing.py 18e6 2032 256,0 2>&1 | grep variant | sed 's/^/NOHIT /' >> $O/timing.txt
PROQA_DEBUG_NOHIT=1 PROQA_FILTER_QW=4 python scripts/dev_search_timing.py 18e6 2032 256,0 2>&1 | grep variant | sed 's/^/NOHIT QW4 /' >> $O/timing.txt
done
