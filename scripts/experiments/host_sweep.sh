#!/bin/bash
# host-builder variants: step counts and frame time (scripts/experiments/tree_quality.py); one run per argument
for v in "$@"; do
  echo -n "[$v] "
  env $v timeout -k 10 300 python -u scripts/experiments/tree_quality.py ${CFG:-3} host 2>&1 | grep '^cfg' || exit 1
done
