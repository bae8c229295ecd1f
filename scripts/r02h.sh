cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02h; mkdir -p $O
for cfg in "BOOT=4096 GROWTH=4" "BOOT=4096 GROWTH=6" "BOOT=4096 GROWTH=8" "BOOT=8192 GROWTH=4" "BOOT=8192 GROWTH=6" "BOOT=2048 GROWTH=4"; do
echo "== $cfg" >> $O/wall.txt
env $cfg python scripts/dev_wall_timing.py 2.25e6 2032 2>&1 | grep rows >> $O/wall.txt
env $cfg python scripts/dev_wall_timing.py 18e6 2032 2>&1 | grep rows >> $O/wall.txt
done
for cfg in "BOOT=4096 GROWTH=8" "BOOT=4096 GROWTH=16 PROQA_CAND_BUDGET=1280" "BOOT=8192 GROWTH=16 PROQA_CAND_BUDGET=1280" "BOOT=8192 GROWTH=8" "BOOT=4096 GROWTH=12 PROQA_CAND_BUDGET=1000"; do
echo "== $cfg" >> $O/wall_small.txt
env $cfg python scripts/dev_wall_timing.py 18e6 256,32,1 2>&1 | grep rows >> $O/wall_small.txt
done
