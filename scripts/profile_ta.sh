#!/bin/bash
# Texture-address / vector-L1 passes for bench.py: is the per-lane-divergent node fetch bound by the TA/TCP pipeline?
# Usage: bash scripts/profile_ta.sh <tag> [bench args]
set -u
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-pipelined $*"
timeout -s KILL 240 rocprofv3 --pmc GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUSY_avr TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d "$OUT/ta1" -- python3 "$REPO/bench.py" $ARGS > "$OUT/ta1.log" 2>&1 || exit 1
timeout -s KILL 240 rocprofv3 --pmc GRBM_GUI_ACTIVE TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum --output-format csv -d "$OUT/ta2" -- python3 "$REPO/bench.py" $ARGS > "$OUT/ta2.log" 2>&1 || exit 1
timeout -s KILL 240 rocprofv3 --pmc GRBM_GUI_ACTIVE TD_TD_BUSY_sum TD_TC_STALL_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum --output-format csv -d "$OUT/ta3" -- python3 "$REPO/bench.py" $ARGS > "$OUT/ta3.log" 2>&1 || exit 1
echo done
