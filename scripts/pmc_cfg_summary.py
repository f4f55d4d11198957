#!/usr/bin/env python3
"""Summary of scripts/profile_cfg.sh for one configuration: per kernel and per FRAME -- HBM bytes (FETCH_SIZE doubled per the
gfx950 note of /opt/skills/guides/MI355X_MICROARCH.md, WRITE_SIZE exact, both reported by rocprofv3 in KiB), L2 hit rate,
vector-L1 miss fraction, VALU lane utilisation SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU), wait / issue shares, and
the frame's span on the device clock from the kernel trace of the same program, from which hbm_frac and valu_issue_frac.
Usage: python scripts/pmc_cfg_summary.py gpurun_out/pmc_<tag> <frames> > profiles/r03_cfg<N>_pmc_summary.txt"""
import csv
import glob
import os
import sys
from collections import defaultdict

HBM_PEAK = 8.0e12          # B/s, MI355X_MICROARCH.md
N_SIMD, CLOCK_HZ, VALU_CYCLES = 1024, 2.4e9, 2


def short(name):
    return name.replace("void trr::", "").replace("trr::", "").split("(")[0]


def main(root, frames):
    tot = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(lambda: defaultdict(int))
    owner = {}   # a counter collected by two passes is taken from the first pass only
    for f in sorted(glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True)):
        pass_name = os.path.relpath(f, root).split(os.sep)[0]
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if not k.startswith("k_"):
                continue
            if owner.setdefault(r["Counter_Name"], pass_name) != pass_name:
                continue
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[k][r["Counter_Name"]] += 1
    # frame span: kernels of our library in the kernel trace, split into `frames` equal runs of dispatches
    span_ms = None
    tr = glob.glob(os.path.join(root, "trace", "**", "*kernel_trace.csv"), recursive=True)
    if tr:
        rows = [r for r in csv.DictReader(open(tr[0])) if "trr::" in r["Kernel_Name"]]
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        n = len(rows) // frames
        spans = []
        for i in range(frames):
            part = rows[i * n:(i + 1) * n]
            spans.append((max(int(r["End_Timestamp"]) for r in part) - int(part[0]["Start_Timestamp"])) / 1e6)
        span_ms = min(spans[1:]) if frames > 1 else spans[0]
        print(f"frame span on the device clock (first kernel start to last kernel end), per frame: {[round(s, 3) for s in spans]} ms"
              f" -> {span_ms:.3f} ms used below (best frame after the first)")
    hbm = valu = 0.0
    thr = act = 0.0
    hit = miss = 0.0
    print(f"{'kernel':38s} {'launch/fr':>9s} {'VALU G/fr':>9s} {'lane util':>9s} {'wait':>6s} {'valu act':>8s} {'HBM rd MB/fr (x2)':>17s} {'HBM wr MB/fr':>12s} {'L2 hit':>7s} {'L1 miss':>7s}")
    for k in sorted(tot, key=lambda k: -tot[k].get("SQ_WAVE_CYCLES", 0)):
        t = tot[k]
        rd = 2 * t.get("FETCH_SIZE", 0.0) * 1024 / frames
        wr = t.get("WRITE_SIZE", 0.0) * 1024 / frames
        hbm += rd + wr
        valu += t.get("SQ_INSTS_VALU", 0.0) / frames
        thr += t.get("SQ_THREAD_CYCLES_VALU", 0.0)
        act_one = t.get("SQ_ACTIVE_INST_VALU", 0.0)
        act += act_one
        wc = t.get("SQ_WAVE_CYCLES", 0.0)
        lane = t.get("SQ_THREAD_CYCLES_VALU", 0.0) / (64 * act_one) if act_one else float("nan")
        h, m = t.get("TCC_HIT_sum", 0.0), t.get("TCC_MISS_sum", 0.0)
        hit += h
        miss += m
        l1 = t.get("TCP_TCC_READ_REQ_sum", 0.0) / t["TCP_TOTAL_CACHE_ACCESSES_sum"] if t.get("TCP_TOTAL_CACHE_ACCESSES_sum") else float("nan")
        print(f"{k[:38]:38s} {max(calls[k].values()) / frames:9.1f} {t.get('SQ_INSTS_VALU', 0.0) / frames / 1e9:9.3f} {lane:9.3f} "
              f"{t.get('SQ_WAIT_ANY', 0.0) / wc if wc else float('nan'):6.3f} {act_one / wc if wc else float('nan'):8.3f} {rd / 1e6:17.1f} {wr / 1e6:12.1f} "
              f"{h / (h + m) if h + m else float('nan'):7.3f} {l1:7.3f}")
    print(f"per frame: HBM traffic {hbm / 1e9:.3f} GB, VALU wave-instructions {valu / 1e9:.3f} G, L2 hit rate {hit / (hit + miss) if hit + miss else float('nan'):.3f}, "
          f"lane utilisation (all kernels) {thr / (64 * act) if act else float('nan'):.3f}")
    if span_ms:
        s = span_ms * 1e-3
        print(f"against the span: HBM {hbm / s / 1e12:.3f} TB/s = hbm_frac {hbm / s / HBM_PEAK:.4f} of 8 TB/s;  "
              f"valu_issue_frac {valu * VALU_CYCLES / (N_SIMD * CLOCK_HZ * s):.4f}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 3)
