import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
K = float(sys.argv[2])   # frames in the SECOND HALF of the dispatches (e.g. 1.5 for a 3-frame run)
rows = [r for r in csv.DictReader(open(f)) if "trr::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# second repetition = last half of the dispatches
rows = rows[len(rows) // 2:]
t0 = int(rows[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in rows)
span = (t1 - t0) / 1e6
by = collections.defaultdict(lambda: [0, 0.0])
ev = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void trr::", "").replace("trr::", "")
    by[name][0] += 1; by[name][1] += (e - s) / 1e6
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
busy = 0; depth = 0; last = None; conc = collections.defaultdict(float)
for t, d in ev:
    if last is not None: conc[depth] += (t - last) / 1e6
    depth += d; last = t
print("span %.2f ms over %d frames = %.2f ms/frame" % (span, K, span / K))
for k, (c, ms) in sorted(by.items(), key=lambda kv: -kv[1][1]):
    print("  %-28s calls/frame %5.1f  ms/frame %6.2f  avg us %7.1f" % (k, c / K, ms / K, ms / c * 1e3))
print("  sum of durations per frame %.2f ms" % (sum(v[1] for v in by.values()) / K))
for d in sorted(conc): print("  time with %d kernels in flight: %.2f ms/frame" % (d, conc[d] / K))
