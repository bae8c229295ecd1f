cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02t; mkdir -p $O; rm -f $O/*.txt
timeout 600 python scripts/dev_gemm_own.py 2>&1 | grep -v "amdgpu" | cut -c1-200 >> $O/gemm.txt
