#!/bin/bash
# One SQ pass (wave cycles, waits, VALU) + one TCC/GRBM pass for bench.py.  Usage: bash scripts/profile_sq.sh <tag> [bench args]
set -u
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-pipelined $*"
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d "$OUT/sq" -- python3 "$REPO/bench.py" $ARGS > "$OUT/sq.log" 2>&1 || exit 1
timeout -s KILL 240 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAIT_INST_LDS --output-format csv -d "$OUT/sq2" -- python3 "$REPO/bench.py" $ARGS > "$OUT/sq2.log" 2>&1 || exit 1
echo done
