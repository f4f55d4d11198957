#!/bin/bash
# rocprofv3 counters for ONE of the five BASELINE.json configurations (lone frames of the whole image on one GPU), one
# --pmc pass per run (the box refuses over-subscribed passes).  The profiled program is scripts/experiments/shard_trace.py
# (1 shard, K frames with a host sync between them).
# Usage (on the GPU box, from the repo root):  bash scripts/profile_cfg.sh <tag> <config 1..5> [frames]
# Output: gpurun_out/pmc_<tag>/<pass>/..., summarised by scripts/pmc_cfg_summary.py
set -u
TAG=$1; CFG=$2; K=${3:-3}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run_pass() {
  name=$1; shift
  timeout -s KILL 300 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$REPO/scripts/experiments/shard_trace.py" 1 "$K" "$CFG" > "$OUT/$name.log" 2>&1
  rc=$?
  echo "pass $name rc=$rc"
  return $rc
}
run_pass sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU || exit 1
run_pass lane GRBM_GUI_ACTIVE SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_SMEM || exit 1
run_pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE || exit 1
run_pass fetch FETCH_SIZE || exit 1
run_pass write WRITE_SIZE || exit 1
run_pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum || exit 1
# the same program under the kernel trace: kernel durations and the frame's span on the device clock
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/scripts/experiments/shard_trace.py" 1 "$K" "$CFG" > "$OUT/trace.log" 2>&1 || exit 1
echo done
