cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02d; mkdir -p $O
for n in 256 6400 32000; do for q in 2032 256; do
bash scripts/dev_trace_search.sh $n $q 12 > $O/trace_${n}_${q}.txt 2>&1
PROQA_DEBUG_NOHIT=1 bash scripts/dev_trace_search.sh $n $q 12 > $O/trace_${n}_${q}_nohit.txt 2>&1
done; done
