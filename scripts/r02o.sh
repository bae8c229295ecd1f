cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02o; mkdir -p $O
timeout 1200 python -m pytest tests/test_search_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1
