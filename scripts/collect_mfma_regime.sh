#!/bin/bash
# MFMA-regime experiment: PMC rows of the shipped filter (8 waves x 64 queries), the 4-wave x 128-query
# variant (PROQA_FILTER_QW=4) and a bare MFMA loop, each as separate SQ and GRBM passes.  Run on the GPU box.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${1:-mfma_regime}
mkdir -p $OUT
S="scripts/dev_search_timing.py 18e6 2032 256,0"
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $OUT/qw2_sq -o p -- python3 $S > $OUT/qw2_sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/qw2_grbm -o p -- python3 $S > $OUT/qw2_grbm.log 2>&1
export PROQA_FILTER_QW=4
rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $OUT/qw4_sq -o p -- python3 $S > $OUT/qw4_sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/qw4_grbm -o p -- python3 $S > $OUT/qw4_grbm.log 2>&1
unset PROQA_FILTER_QW
for z in 0 1; do
rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $OUT/bare${z}_sq -o p -- python3 scripts/dev_mfma_ref.py $z > $OUT/bare${z}_sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/bare${z}_grbm -o p -- python3 scripts/dev_mfma_ref.py $z > $OUT/bare${z}_grbm.log 2>&1
done
python3 scripts/summarize_mfma_regime.py $OUT > $OUT/mfma_regime.json
rm -rf $OUT/qw2_sq $OUT/qw2_grbm $OUT/qw4_sq $OUT/qw4_grbm $OUT/bare0_sq $OUT/bare0_grbm $OUT/bare1_sq $OUT/bare1_grbm
