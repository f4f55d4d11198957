"""Per-kernel sums of the counters of rocprofv3 --pmc passes.  Usage: pmc_table.py <dir with */vm/*counter_collection.csv> ..."""
import collections
import csv
import glob
import sys

for d in sys.argv[1:]:
    for f in sorted(glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        disp = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void trr::", "").replace("trr::", "")
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k].add(r["Dispatch_Id"])
        print("==", f)
        for k, v in agg.items():
            if not k.startswith("k_"):
                continue
            print(f"{k[:40]:40s} dispatches {len(disp[k]):3d}  " + "  ".join(f"{c}={x:.4g}" for c, x in sorted(v.items())))
