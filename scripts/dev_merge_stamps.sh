#!/bin/bash
# Developer build with s_memtime stamps in topk_merge (-DPROQA_MERGE_STAMPS), on the GPU box's copy of the tree only: phase
# medians of every 16th merge launch.  usage: bash scripts/dev_merge_stamps.sh [rows]
set -e
export PYTHONPATH=$PWD
cd proqa_amd/csrc
F="-x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DPROQA_MERGE_STAMPS"
hipcc $F -c mips_kernels.hip -o mips_kernels.o &
hipcc $F -c mips_index.cpp -o mips_index.o &
wait
g++ -shared -o libproqa_hip.so common.o npy_io.o wordpiece.o mips_index.o mips_kernels.o sharded_search.o encoder_kernels.o gemm_kernels.o \
    attention_kernel.o lt_gemm.o encoder.o kmeans_kernels.o microbench.o -Wl,--no-as-needed -lpthread -lm -ldl
cd ../..
PROQA_MERGE_STAMPS_DUMP=1 python scripts/dev_nominate_ab.py ${1:-18e6} 2032 80 normal 2>&1 | grep "merge phase\|re-scoring\|skew\|mode=" | tail -40
