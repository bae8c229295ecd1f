cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02t; mkdir -p $O; rm -f $O/*.txt
for v in 0 1 0 1; do
echo "ring4=$v" >> $O/gemm.txt
if [ $v = 1 ]; then export PROQA_GEMM_RING4=1; else unset PROQA_GEMM_RING4; fi
timeout 600 python scripts/dev_gemm_own.py 2>&1 | grep -v "amdgpu" | cut -c1-200 >> $O/gemm.txt
done
