#!/usr/bin/env python3
"""Per-FRAME totals from the counter passes of scripts/profile_pmc.sh, for bench.py's roofline object:
VALU instructions (SQ_INSTS_VALU) and HBM bytes (FETCH_SIZE / WRITE_SIZE) summed over every kernel of a frame, plus a
per-kernel breakdown.  The counter build's kernels (template argument COUNT = true, one extra frame of bench.py) are left
out; `frames` is the number of other frames the profiled command rendered (warm-up + timed).

rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB.  /opt/skills/guides/MI355X_MICROARCH.md: on gfx950 FETCH_SIZE tallies
128-B requests at 64 B for wide (16 B/lane) reads, so the read figure is doubled (an upper estimate for our gathered
16-B reads); WRITE_SIZE is exact.
Usage: python scripts/pmc_frame.py gpurun_out/pmc_<tag> <frames> <commit> > profiles/pmc_frame.json"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    return name.replace("void trr::", "").replace("trr::", "").split("(")[0]


def is_counter_build(k):
    """k_trace<COUNT>, k_trace_level<RAYGEN, COUNT, QN>: the instantiations of the counter frame."""
    if "<" not in k:
        return False
    name, args = k.split("<", 1)
    a = [x.strip() for x in args.rstrip(">").split(",")]
    if name == "k_trace":
        return a[0] == "true"
    if name == "k_trace_level":
        return len(a) > 1 and a[1] == "true"
    return False


def main(root, frames, commit):
    tot = defaultdict(lambda: defaultdict(float))
    n = defaultdict(lambda: defaultdict(int))
    for f in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            c = r["Counter_Name"]
            if c not in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU"):
                continue
            k = short(r["Kernel_Name"])
            if not k.startswith("k_") or is_counter_build(k):
                continue
            tot[k][c] += float(r["Counter_Value"])
            n[k][c] += 1
    kernels = {}
    valu = hbm = 0.0
    for k in sorted(tot):
        t = tot[k]
        b = (2 * t.get("FETCH_SIZE", 0.0) + t.get("WRITE_SIZE", 0.0)) * 1024
        # the trace kernels of the counter frame are other instantiations (left out above); set-up, resolve and assemble are
        # the same kernels in that frame, so they ran one frame more
        fr = frames if k.startswith("k_trace") else frames + 1
        kernels[k] = {"launches_per_frame": max(n[k].values()) / fr, "valu_insts_per_frame": t.get("SQ_INSTS_VALU", 0.0) / fr,
                      "salu_insts_per_frame": t.get("SQ_INSTS_SALU", 0.0) / fr, "hbm_bytes_per_frame": b / fr,
                      "fetch_bytes_per_frame_x2": 2 * t.get("FETCH_SIZE", 0.0) * 1024 / fr,
                      "write_bytes_per_frame": t.get("WRITE_SIZE", 0.0) * 1024 / fr}
        if t.get("SQ_WAVE_CYCLES"):
            kernels[k]["wait_any_over_wave_cycles"] = round(t.get("SQ_WAIT_ANY", 0.0) / t["SQ_WAVE_CYCLES"], 4)
            kernels[k]["active_valu_over_wave_cycles"] = round(t.get("SQ_ACTIVE_INST_VALU", 0.0) / t["SQ_WAVE_CYCLES"], 4)
        valu += t.get("SQ_INSTS_VALU", 0.0) / fr
        hbm += b / fr
    print(json.dumps({"source": os.path.basename(root.rstrip("/")), "commit": commit, "frames": frames,
                      "command": "bash scripts/profile_pmc.sh <tag>  (bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pipelined, one --pmc pass per run)",
                      "valu_insts_per_frame": valu, "hbm_bytes_per_frame": hbm, "kernels": kernels}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), sys.argv[3] if len(sys.argv) > 3 else "unknown")
