cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02n; mkdir -p $O
for q in 1 2 4; do
PROQA_FILTER_QW=$q python scripts/dev_search_timing.py 18e6 2032 256,0 2>&1 | grep variant | sed "s/^/QW=$q /" >> $O/timing.txt
done
