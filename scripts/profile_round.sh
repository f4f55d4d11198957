#!/bin/bash
# The round's profile set for the headline workload, on the GPU box from the repo root: bench line, bench under
# rocprofv3 --kernel-trace --stats, PMC passes (profile_pmc.sh), issue / lane-utilisation passes (profile_issue.sh).
# Usage: bash scripts/profile_round.sh <tag>      -> gpurun_out/<tag>_*
set -u
TAG=$1
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
timeout -k 10 400 python -u bench.py > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err" || exit 1
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_trace" -- python3 "$REPO/bench.py" --steps 3 --no-cpu-baseline --no-pipelined > "$OUT/${TAG}_bench_under_rocprof.json" 2> "$OUT/${TAG}_trace.err") || exit 1
bash scripts/profile_pmc.sh "$TAG" || exit 1
bash scripts/profile_issue.sh "${TAG}_issue" || exit 1
echo profile_round done
