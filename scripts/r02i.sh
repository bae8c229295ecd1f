cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02i; mkdir -p $O
timeout 900 python -m pytest tests/test_rccl_gpu.py tests/test_search_gpu.py tests/test_distributed_gpu.py tests/test_abi_and_npy.py -x -q -m gpu > $O/pytest.txt 2>&1
timeout 900 python bench.py --skip-encode --skip-float32 --steps 10 > $O/bench_search.json 2> $O/bench_search.err
