#!/bin/bash
# Rebuild only the four-wave kernel's translation unit and relink (iteration on gemm256q.h without the 3-minute full build).
set -e
cd "$(dirname "$0")/../mps_bitsandbytes_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function "$@" -c gemm256q.hip -o gemm256q.o 2>&1 | grep -E "error|Spill|Scratch" || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmbnb_hip.so api.o quant_kernels.o matmul4_kernels.o gemm256q.o int8_kernels.o nn_kernels.o
touch -r gemm256q.o api.o quant_kernels.o matmul4_kernels.o int8_kernels.o nn_kernels.o 2>/dev/null || true
