"""Prints the kernels of the LAST frame of a rocprofv3 --kernel-trace run in start order: start and end relative to the frame's
first kernel, duration, gap to the previous kernel's end.  Usage: python scripts/experiments/chain.py <trace dir> <frames>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
frames = int(sys.argv[2])
rows = [r for r in csv.DictReader(open(f)) if "trr::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows) // frames
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
last_end = 0
for r in rows:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    name = r["Kernel_Name"].split("(")[0].replace("void trr::", "").replace("trr::", "")
    print(f"{s / 1e3:9.1f} us .. {e / 1e3:9.1f}  dur {(e - s) / 1e3:8.1f}  gap {(s - last_end) / 1e3:7.1f}  grid {r.get('Grid_Size', '?'):>8s}  {name[:60]}")
    last_end = max(last_end, e)
print(f"frame span {last_end / 1e3:.1f} us, {len(rows)} kernels")
