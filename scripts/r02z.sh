cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02z; mkdir -p $O; rm -f $O/*.txt
timeout 1500 python -m pytest tests/test_search_gpu.py -x -q -m gpu -k "large_k or merge or exact or page or fuzz or random" > $O/pytest.txt 2>&1
python scripts/dev_largek_timing.py 6980 > $O/largek_6980.txt 2>&1
python scripts/dev_single_query.py > $O/single.txt 2>&1
