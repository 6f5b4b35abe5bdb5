#!/usr/bin/env python3
"""Summarise one replayed train step from a rocprofv3 --kernel-trace CSV: kernel count, span, per-kernel totals."""
import collections, csv, glob, os, sys
d = sys.argv[1]
f = max(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adam_multi" in r["Kernel_Name"]]
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
# the step with the shortest span: a graph replay from the timed region (bench.py also runs eager steps, whose launches are host-paced)
cands = [rows[a + 1: b + 1] for a, b in zip(idx[:-1], idx[1:])]
n_mode = collections.Counter(len(c) for c in cands).most_common(1)[0][0]
step = min((c for c in cands if len(c) == n_mode), key=lambda c: int(c[-1]["End_Timestamp"]) - int(c[0]["Start_Timestamp"]))
print(f, "\nkernels in step", len(step), "span us", (int(step[-1]["End_Timestamp"]) - int(step[0]["Start_Timestamp"])) / 1e3, "sum of durations", round(sum(map(dur, step)), 1))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in step:
    k = r["Kernel_Name"][:72]; agg[k][0] += 1; agg[k][1] += dur(r)
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[: int(sys.argv[2]) if len(sys.argv) > 2 else 20]:
    print(f"{t:8.1f} us {n:3d}x  {k}")
if "--seq" in sys.argv:
    t0 = int(step[0]["Start_Timestamp"])
    for r in step:
        gap = (int(r["Start_Timestamp"]) - t0) / 1e3
        print(f"{gap:9.1f} +{dur(r):7.1f} us  grid {r.get('Grid_Size_X','?'):>8s}x{r.get('Grid_Size_Y','?')}x{r.get('Grid_Size_Z','?')} wg {r.get('Workgroup_Size_X','?'):>4s} lds {r.get('LDS_Block_Size','?'):>6s} vgpr {r.get('VGPR_Count','?'):>3s}/{r.get('Accum_VGPR_Count','?'):>3s}  {r['Kernel_Name'][:90]}")
