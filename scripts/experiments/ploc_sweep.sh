#!/bin/bash
# device-build tree variants: step counts and frame time of the tree each setting produces (scripts/experiments/tree_quality.py)
# usage (GPU box): CFG=3 bash scripts/experiments/ploc_sweep.sh "VAR=val VAR=val" ...   one run per argument ("host" = the host builder)
for v in "$@"; do
  if [ "$v" = host ]; then
    timeout -k 10 200 python -u scripts/experiments/tree_quality.py ${CFG:-3} host 2>&1 | grep '^cfg' || exit 1
  else
    echo -n "[$v] "
    env $v timeout -k 10 200 python -u scripts/experiments/tree_quality.py ${CFG:-3} device 2>&1 | grep '^cfg' || exit 1
  fi
done
