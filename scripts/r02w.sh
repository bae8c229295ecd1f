cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02w; mkdir -p $O
python scripts/dev_gemm_vs_m.py 36864 37120 37376 37632 37888 38912 40960 41216 41472 41728 41984 42240 42496 43008 44032 45056 47104 49152 51200 53248 > $O/vs_m.txt 2>&1
