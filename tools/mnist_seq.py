import csv, glob, os, sys, collections
f = max(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adam_multi" in r["Kernel_Name"]]
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
# steps = pairs of segments; choose the pair with the smallest span among the later ones
best = None
for k in range(2, len(idx) - 2, 1):
    seg = rows[idx[k - 2] + 1: idx[k] + 1]
    span = int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])
    if best is None or span < best[0]: best = (span, seg)
span, seg = best
print("kernels", len(seg), "span us", span / 1e3, "sum dur", round(sum(map(dur, seg)), 1))
t0 = int(seg[0]["Start_Timestamp"])
for r in seg:
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} +{dur(r):6.1f}  grid {r.get('Grid_Size_X','?'):>7s}x{r.get('Grid_Size_Y','?')}x{r.get('Grid_Size_Z','?')}  {r['Kernel_Name'][:110]}")
