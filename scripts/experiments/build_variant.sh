#!/bin/bash
# Builds a variant of the library with extra compiler flags for kernels.hip (experiments): scratch/var/lib_<name>.so,
# used through TRR_LIB=... by scripts/experiments/quick.py.  Usage: build_variant.sh <name> "<extra flags>"
set -e
NAME=$1; EXTRA=$2
cd "$(dirname "$0")/../../raytracing_amd/csrc"
mkdir -p ../../scratch/var
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -Wall -Wno-unused-function -pthread"
/opt/rocm/bin/hipcc $FLAGS $EXTRA -c kernels.hip -o ../../scratch/var/kernels_$NAME.o ${SAVE_TEMPS:+-save-temps=obj}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../../scratch/var/lib_$NAME.so ../../scratch/var/kernels_$NAME.o api.o bvh.o build_device.o -pthread
echo built scratch/var/lib_$NAME.so
