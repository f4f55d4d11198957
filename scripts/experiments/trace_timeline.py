import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last frame: find the last k_trace<true dispatch and take from there
idx = max(i for i, r in enumerate(rows) if "k_trace<true" in r["Kernel_Name"])
rows = rows[idx:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void trr::", "").replace("trr::", "")
    print("%9.1f us  dur %8.1f us  gap %6.1f  %s grid %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, name, r.get("Grid_Size", "?")))
    prev_end = e
print("total %.1f us" % ((prev_end - t0) / 1e3))
