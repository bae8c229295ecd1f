cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02k; mkdir -p $O
python scripts/dev_wall_timing.py 18e6 256,32,1 > $O/wall_nt.txt 2>&1
python scripts/dev_search_timing.py 18e6 32 256,0 >> $O/wall_nt.txt 2>&1
timeout 600 python -m pytest tests/test_search_gpu.py -x -q -m gpu -k "integer or fuzz or random" > $O/pytest.txt 2>&1
