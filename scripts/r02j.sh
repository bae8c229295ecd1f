cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02j; mkdir -p $O
for w in 2 4 8 16 32; do echo "wgs/cu $w" >> $O/peaks.txt; PROQA_STREAM_WGS=$w python scripts/dev_peaks.py >> $O/peaks.txt 2>&1; done
