#!/bin/bash
# Collects rocprofv3 counters for bench.py, one --pmc pass per run (the box refuses/hangs on over-subscribed passes).
# Usage (on the GPU box, from the repo root):  bash scripts/profile_pmc.sh <tag> [bench args...]
# Output: gpurun_out/pmc_<tag>/<pass>/..., summarised by scripts/pmc_summary.py
set -u
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-pipelined $*"
run_pass() {
  name=$1; shift
  timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$REPO/bench.py" $ARGS > "$OUT/$name.log" 2>&1
  rc=$?
  echo "pass $name rc=$rc"
  return $rc
}
run_pass sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU || exit 1
run_pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT || exit 1
run_pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE || exit 1
run_pass fetch FETCH_SIZE || exit 1
run_pass write WRITE_SIZE || exit 1
run_pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum || exit 1
echo done
