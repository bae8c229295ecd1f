#!/bin/bash
# Developer build with s_memtime stamps in the filter kernels (-DPROQA_FILTER_STAMPS), on the GPU box's copy of the tree only:
# per-unit MFMA section / test section / barrier / excursion ticks of mips_filter_i8.  usage: bash scripts/dev_nominate_stamps.sh
set -e
export PYTHONPATH=$PWD
cd proqa_amd/csrc
F="-x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DPROQA_FILTER_STAMPS"
hipcc $F -c mips_kernels.hip -o mips_kernels.o &
hipcc $F -c mips_index.cpp -o mips_index.o &
wait
g++ -shared -o libproqa_hip.so common.o npy_io.o wordpiece.o mips_index.o mips_kernels.o sharded_search.o encoder_kernels.o gemm_kernels.o \
    attention_kernel.o lt_gemm.o encoder.o kmeans_kernels.o microbench.o -Wl,--no-as-needed -lpthread -lm -ldl
cd ../..
for f in ${FLAGS:-0}; do
  echo "== PROQA_FILTER_FLAGS=$f"
  PROQA_FILTER_FLAGS=$f python scripts/dev_nominate_ab.py 18e6 2032 80 normal 2>&1 | grep "filter stamps\|mode=" | tail -4
done
