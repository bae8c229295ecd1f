#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: rocprofv3 passes of bench.py.
# Kernel trace/stats and every PMC group are separate passes (MI355X_MICROARCH.md, rocprofv3 PMC slots).
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${1:-profile}
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 bench.py --skip-cpu --steps 10 --corpus-passages 0 --cli-passages 0 --skip-extras --skip-float32 > $OUT/bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o bench -- python3 bench.py --skip-cpu --skip-encode --steps 2 --warmup 1 --corpus-passages 0 --cli-passages 0 --skip-extras --skip-float32 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o bench -- python3 bench.py --skip-cpu --skip-encode --steps 2 --warmup 1 --corpus-passages 0 --cli-passages 0 --skip-extras --skip-float32 > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq -o bench -- python3 bench.py --skip-cpu --skip-encode --steps 2 --warmup 1 --corpus-passages 0 --cli-passages 0 --skip-extras --skip-float32 > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_grbm -o bench -- python3 bench.py --skip-cpu --skip-encode --steps 2 --warmup 1 --corpus-passages 0 --cli-passages 0 --skip-extras --skip-float32 > $OUT/pmc_grbm.log 2>&1
# encoder: MFMA-pipe occupancy and clock per kernel (token-size search so that the encode leg dominates)
ENC="--skip-cpu --rows 200000 --queries 64 --steps 1 --warmup 1 --encode-steps 3 --corpus-passages 0 --cli-passages 0 --skip-extras --skip-float32"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/enc_pmc_sq -o bench -- python3 bench.py $ENC > $OUT/enc_pmc_sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/enc_pmc_grbm -o bench -- python3 bench.py $ENC > $OUT/enc_pmc_grbm.log 2>&1
# encoder HBM traffic: full-size steps only, FETCH_SIZE and WRITE_SIZE in passes of their own
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/enc_pmc_fetch -o bench -- python3 bench.py $ENC --skip-varlen > $OUT/enc_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/enc_pmc_write -o bench -- python3 bench.py $ENC --skip-varlen > $OUT/enc_pmc_write.log 2>&1
python3 scripts/summarize_profiles.py $OUT
# keep gpurun_out small: the raw per-dispatch CSVs are tens of MB
cp $OUT/trace/bench_kernel_stats.csv $OUT/kernel_stats_full.csv 2>/dev/null
grep '^{' $OUT/bench_trace.log | tail -1 > $OUT/bench_under_rocprof.json
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq $OUT/pmc_grbm $OUT/enc_pmc_sq $OUT/enc_pmc_grbm $OUT/enc_pmc_fetch $OUT/enc_pmc_write
