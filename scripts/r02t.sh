cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02t; mkdir -p $O; rm -f $O/*.txt
for i in 1 2 3; do timeout 600 python scripts/dev_gemm_own.py 2>&1 | grep -v "amdgpu" | cut -c1-200 >> $O/gemm.txt; done
timeout 900 python -m pytest tests/test_encoder_gpu.py -x -q -m gpu -k "dense or ffn1 or bert_base" > $O/pytest.txt 2>&1
