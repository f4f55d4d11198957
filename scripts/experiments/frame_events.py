"""Prints the kernels of the LAST frame of a `rocprofv3 --kernel-trace` run of shard_trace.py, one per line: start and end
(us from the frame's first kernel), duration, queue, name.  Usage: frame_events.py <rocprof output dir> <frames in the run>"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
K = int(sys.argv[2])
rows = [r for r in csv.DictReader(open(f)) if "trr::" in r["Kernel_Name"] and "quantise" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
per = len(rows) // K
rows = rows[-per:]
t0 = int(rows[0]["Start_Timestamp"])
queues = {}
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q = queues.setdefault(r.get("Queue_Id", "?"), len(queues))
    name = r["Kernel_Name"].split("(")[0].replace("void trr::", "").replace("trr::", "")
    print("%8.1f %8.1f  %7.1f us  q%-2d grid %-8s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, r.get("Grid_Size", "?"), name))
print("frame: %.1f us, %d kernels, %d queues" % ((max(int(r["End_Timestamp"]) for r in rows) - t0) / 1e3, len(rows), len(queues)))
