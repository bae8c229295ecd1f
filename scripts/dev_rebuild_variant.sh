#!/bin/bash
# developer A/B of a compile-time variant on the GPU box: rebuild ONE source of libproqa_hip.so with extra flags and relink in place.
#   scripts/dev_rebuild_variant.sh mips_kernels.hip -DPROQA_TAILMASK_HOISTED      (no flags: the shipped form)
cd $(dirname $0)/../proqa_amd/csrc
SRC=$1; shift
hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c $SRC -o ${SRC%.*}.o 2>/dev/null || exit 1
g++ -shared -o libproqa_hip.so common.o npy_io.o wordpiece.o mips_index.o mips_kernels.o sharded_search.o encoder_kernels.o gemm_kernels.o attention_kernel.o lt_gemm.o encoder.o kmeans_kernels.o microbench.o -Wl,--no-as-needed -lpthread -lm -ldl
