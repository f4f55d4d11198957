#!/bin/bash
# Reproducer for DESIGN.md "Toolchain note": the round-1 trace kernel that carried the per-object and per-raycast winners'
# barycentrics in registers (the tree of commit 1bef76b^) renders wrong colours when hipcc's SLP vectoriser runs (-O2 and
# above), and the oracle's image bit for bit with -fno-slp-vectorize or at -O1.  Hit ids are right either way: only
# the carried (u, v, w) are wrong.  ROCm 7.2.0 hipcc, gfx950.
#
#   bash scripts/experiments/slp_repro/run.sh build     # here (no GPU needed): worktree + four builds of that kernel
#   gpurun -- 'bash scripts/experiments/slp_repro/run.sh run'    # on the GPU box: each build against the CPU oracle
#
# Measured (profiles/r02o_slp_repro.txt): -O1 0 pixels differ; -O2 and -O3 1647 of 76800; -O3 -fno-slp-vectorize 0;
# -O3 -fno-unroll-loops 1647.  An SLP-vectorised fp32 division alone is correctly rounded (slp_div.hip, 0 of 2^20 pairs
# differ), and today's kernels render the same image with the vectoriser on (they no longer carry those values) -- they
# are built with -fno-slp-vectorize for speed (it costs ~40 VGPRs of shuffles) and every parity test is bit-exact.
set -e
ROOT=$(cd "$(dirname "$0")/../../.." && pwd)
W=$ROOT/scratch/slp_repro_tree
BASE="--offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-function -pthread"
if [ "$1" = build ]; then
    rm -rf "$W" && git -C "$ROOT" worktree prune && git -C "$ROOT" worktree add -f "$W" 1bef76b^ -q
    # that tree is one line short of compiling (a field its device_types.h had already dropped)
    sed -i '/h.pad\[0\] = h.pad\[1\] = h.pad\[2\] = 0;/d' "$W/raytracing_amd/csrc/kernels.hip"
    make -C "$W/oracle" > /dev/null
    cp "$ROOT/scripts/experiments/slp_repro/repro.py" "$W/repro.py"
    for v in "O1:-O1" "O2:-O2" "O3:-O3" "O3_no_slp:-O3 -fno-slp-vectorize" "O3_no_unroll:-O3 -fno-unroll-loops"; do
        n=${v%%:*}; f=${v#*:}
        rm -f "$W/raytracing_amd/csrc/kernels.o"
        make -C "$W/raytracing_amd/csrc" FLAGS="$BASE $f" > /dev/null 2>&1
        cp "$W/raytracing_amd/libtrayracer_hip.so" "$W/raytracing_amd/lib_$n.so"
        echo "built lib_$n.so ($f)"
    done
    F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-value"
    /opt/rocm/bin/hipcc $F "$ROOT/scripts/experiments/slp_repro/slp_div.hip" -o "$W/slp_div"
else
    cd "$W"
    for n in O1 O2 O3 O3_no_slp O3_no_unroll; do
        TRR_LIB=$W/raytracing_amd/lib_$n.so timeout -k 10 120 python3 repro.py || exit 1
    done
    timeout -k 5 60 ./slp_div
fi
