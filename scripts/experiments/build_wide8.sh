#!/bin/bash
# Builds the 8-wide-node experiment (-DTRR_WIDE8: bvh.h, kernels.hip, trace_step.inl) as a whole library of its own,
# scratch/var/lib_wide8.so; run with TRR_LIB=scratch/var/lib_wide8.so TRR_NODES=quantised (the level kernels then read the 8-wide nodes).
set -e
cd "$(dirname "$0")/../../raytracing_amd/csrc"
mkdir -p ../../scratch/var
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -Wall -Wno-unused-function -pthread -DTRR_WIDE8 $1"
for f in kernels api build_device; do /opt/rocm/bin/hipcc $FLAGS -c $f.hip -o ../../scratch/var/w8_$f.o & done
/opt/rocm/bin/hipcc $FLAGS -x hip -c bvh.cpp -o ../../scratch/var/w8_bvh.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../../scratch/var/lib_wide8.so ../../scratch/var/w8_kernels.o ../../scratch/var/w8_api.o ../../scratch/var/w8_bvh.o ../../scratch/var/w8_build_device.o -pthread
echo built scratch/var/lib_wide8.so
