#!/usr/bin/env python3
"""Summarises the counter CSVs written by scripts/profile_pmc.sh: per kernel name, sum of each counter over all
dispatches and dispatch count.  Usage: python scripts/pmc_summary.py gpurun_out/pmc_<tag> > profiles/<name>.txt"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace("void trr::", "").replace("trr::", "")
    return name.split("(")[0][:40]


def main(root):
    tot = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(lambda: defaultdict(int))
    for f in sorted(glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            c = r["Counter_Name"]
            tot[k][c] += float(r["Counter_Value"])
            calls[k][c] += 1
    counters = sorted({c for k in tot for c in tot[k]})
    for k in sorted(tot, key=lambda k: -tot[k].get("SQ_WAVE_CYCLES", 0)):
        print(f"== {k}")
        for c in counters:
            if c in tot[k]:
                print(f"   {c:32s} {tot[k][c]:18.0f}   dispatches {calls[k][c]}")
        t = tot[k]
        if t.get("SQ_WAVE_CYCLES"):
            wc = t["SQ_WAVE_CYCLES"]
            print(f"   -> wait_any/wave_cycles {t.get('SQ_WAIT_ANY', 0) / wc:.3f}  wait_inst_any {t.get('SQ_WAIT_INST_ANY', 0) / wc:.3f}  "
                  f"active_inst_any {t.get('SQ_ACTIVE_INST_ANY', 0) / wc:.3f}  active_valu {t.get('SQ_ACTIVE_INST_VALU', 0) / wc:.3f}")
        if t.get("TCC_HIT_sum") is not None and (t.get("TCC_HIT_sum", 0) + t.get("TCC_MISS_sum", 0)) > 0:
            print(f"   -> L2 hit rate {t['TCC_HIT_sum'] / (t['TCC_HIT_sum'] + t['TCC_MISS_sum']):.3f}")
        if t.get("TCP_TOTAL_CACHE_ACCESSES_sum"):
            print(f"   -> L1 miss fraction (TCC read req / TCP accesses) {t.get('TCP_TCC_READ_REQ_sum', 0) / t['TCP_TOTAL_CACHE_ACCESSES_sum']:.3f}")


if __name__ == "__main__":
    main(sys.argv[1])
