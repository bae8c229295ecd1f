cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02a; mkdir -p $O
bash scripts/dev_trace_search.sh 18e6 2032 18 > $O/trace_18m_2032.txt 2>&1
bash scripts/dev_trace_search.sh 2.25e6 2032 18 > $O/trace_2m_2032.txt 2>&1
bash scripts/dev_trace_search.sh 18e6 32 14 > $O/trace_18m_32.txt 2>&1
bash scripts/dev_trace_search.sh 18e6 1 14 > $O/trace_18m_1.txt 2>&1
python scripts/dev_single_query.py > $O/single_query.txt 2>&1
