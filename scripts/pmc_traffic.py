#!/usr/bin/env python3
"""HBM traffic of k_trace per launch from the FETCH_SIZE / WRITE_SIZE passes of scripts/profile_pmc.sh.

rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB.  /opt/skills/guides/MI355X_MICROARCH.md: on gfx950 FETCH_SIZE tallies
128-B requests at 64 B for wide (16 B/lane) reads, so the read figure is doubled; WRITE_SIZE is exact.  Our reads are
16 B/lane but gathered, not streaming -- the guide calls other patterns uncalibrated, so the doubled figure is an
upper estimate.  Usage: python scripts/pmc_traffic.py gpurun_out/pmc_<tag> > profiles/hbm_traffic.json"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main(root):
    tot = defaultdict(lambda: defaultdict(float))
    n = defaultdict(lambda: defaultdict(int))
    for f in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                k = "k_trace" if "k_trace" in r["Kernel_Name"] else r["Kernel_Name"].split("(")[0].replace("void trr::", "").replace("trr::", "")
                tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
                n[k][r["Counter_Name"]] += 1
    out = {}
    for k in tot:
        fetch_kib, write_kib = tot[k].get("FETCH_SIZE", 0.0), tot[k].get("WRITE_SIZE", 0.0)
        launches = max(n[k].get("FETCH_SIZE", 0), n[k].get("WRITE_SIZE", 0), 1)
        out[k] = {"launches": launches, "fetch_bytes_per_launch_raw": fetch_kib * 1024 / launches,
                  "fetch_bytes_per_launch_x2": 2 * fetch_kib * 1024 / launches, "write_bytes_per_launch": write_kib * 1024 / launches,
                  "hbm_bytes_per_launch": (2 * fetch_kib + write_kib) * 1024 / launches}
    print(json.dumps({"source": os.path.basename(root.rstrip("/")), "kernels": out}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
