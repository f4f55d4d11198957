#!/bin/bash
# Runs scripts/experiments/quick.py <configs> once per environment setting given as further arguments ("A=1 B=2" each).
# Usage: bash scripts/experiments/sweep_env.sh <out file> <configs> "<env 1>" "<env 2>" ...
OUT=$1; CFGS=$2; shift 2
for e in "$@"; do
  echo "== $e" >> "$OUT"
  env $e timeout -k 10 300 python -u scripts/experiments/quick.py "$CFGS" 2>&1 | grep -v amdgpu.ids >> "$OUT" || exit 1
done
cat "$OUT"
