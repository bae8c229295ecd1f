cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02ab; mkdir -p $O; rm -f $O/*.txt
timeout 1500 python -m pytest tests/test_search_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1
for i in 1 2; do
python scripts/dev_wall_timing.py 18e6 2032,256,32,1 >> $O/wall.txt 2>&1
python scripts/dev_search_timing.py 18e6 2032 256,0 2>&1 | grep variant >> $O/wall.txt
python scripts/dev_search_timing.py 18e6 32 256,0 2>&1 | grep variant >> $O/wall.txt
done
python scripts/dev_wall_timing.py 2.25e6 2032 >> $O/wall.txt 2>&1
