#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: rocprofv3 passes of bench.py.
# Kernel trace/stats and every PMC group are separate passes (MI355X_MICROARCH.md, rocprofv3 PMC slots).
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${1:-profile}
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 bench.py --skip-cpu --steps 10 > $OUT/bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o bench -- python3 bench.py --skip-cpu --skip-encode --steps 2 --warmup 1 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o bench -- python3 bench.py --skip-cpu --skip-encode --steps 2 --warmup 1 > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq -o bench -- python3 bench.py --skip-cpu --skip-encode --steps 2 --warmup 1 > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_grbm -o bench -- python3 bench.py --skip-cpu --skip-encode --steps 2 --warmup 1 > $OUT/pmc_grbm.log 2>&1
python3 scripts/summarize_profiles.py $OUT
