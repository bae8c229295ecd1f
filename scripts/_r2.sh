export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/eg -o t -- python3 scripts/dev_encode_one.py 512 128 12 > gpurun_out/eg.log 2>&1
tail -1 gpurun_out/eg.log
python3 scripts/dev_trace_gaps.py $(find gpurun_out/eg -name '*kernel_trace.csv' | head -1) 0.5
rm -rf gpurun_out/eg
python3 scripts/dev_encode_one.py 512 128 12
