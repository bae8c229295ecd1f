cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02r; mkdir -p $O
timeout 600 python scripts/dev_gemm_own.py > $O/gemm.txt 2>&1
