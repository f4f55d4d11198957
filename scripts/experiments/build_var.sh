#!/bin/bash
# usage: scripts/experiments/build_var.sh name "-DFLAG ..."   -> scratch/var/lib_<name>.so
set -e
cd "$(dirname "$0")/../../raytracing_amd/csrc"
make -s
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -Wno-unused-function -pthread"
mkdir -p ../../scratch/var
/opt/rocm/bin/hipcc $F $2 -c kernels.hip -o ../../scratch/var/k_$1.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../../scratch/var/lib_$1.so ../../scratch/var/k_$1.o api.o bvh.o build_device.o -pthread
rm ../../scratch/var/k_$1.o
echo built $1
