#!/bin/bash
# Where do the trace kernels' waves wait?  Memory-instruction latency (INST_LEVEL / INSTS = cycles in flight), TA FIFO
# back-pressure, instruction fetch.  Usage: bash scripts/profile_issue.sh <tag> [bench args]
set -u
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-pipelined $*"
timeout -s KILL 240 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_IFETCH SQ_IFETCH_LEVEL --output-format csv -d "$OUT/i1" -- python3 "$REPO/bench.py" $ARGS > "$OUT/i1.log" 2>&1 || exit 1
timeout -s KILL 240 rocprofv3 --pmc GRBM_GUI_ACTIVE SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_BUSY_CYCLES SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAVE_CYCLES --output-format csv -d "$OUT/i2" -- python3 "$REPO/bench.py" $ARGS > "$OUT/i2.log" 2>&1 || exit 1
timeout -s KILL 240 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU --output-format csv -d "$OUT/i3" -- python3 "$REPO/bench.py" $ARGS > "$OUT/i3.log" 2>&1 || exit 1
echo done
