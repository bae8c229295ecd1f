cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02m; mkdir -p $O
for i in 1 2; do
for f in 0 1 2; do
PROQA_FILTER_FLAGS=$f python scripts/dev_search_timing.py 18e6 2032 256,0 2>&1 | grep variant | sed "s/^/flags=$f /" >> $O/timing.txt
done
PROQA_DEBUG_NOHIT=1 python scripts/dev_search_timing.py 18e6 2032 256,0 2>&1 | grep variant | sed 's/^/NOHIT /' >> $O/timing.txt
PROQA_DEBUG_NOHIT=1 PROQA_FILTER_QW=4 python scripts/dev_search_timing.py 18e6 2032 256,0 2>&1 | grep variant | sed 's/^/NOHIT QW4 /' >> $O/timing.txt
done
