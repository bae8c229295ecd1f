cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02q; mkdir -p $O
timeout 900 python bench.py --rows 2000000 --steps 3 --skip-float32 --corpus-passages 1000000 > $O/bench.json 2> $O/bench.err
