#!/usr/bin/env python3
"""Average kernel durations from a rocprofv3 --kernel-trace CSV directory: python tools/ktrace_avg.py DIR [name-substring ...].
kbench.py's HIP-event timing is host-bound below ~40 us per launch; this reads the device-side durations instead."""
import collections, csv, glob, os, sys

f = max(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    agg[r["Kernel_Name"][:90] + "  grid " + "x".join(r.get(k, "?") for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    if len(sys.argv) > 2 and not any(s in k for s in sys.argv[2:]):
        continue
    v = v[len(v) // 4:]                                   # drop the warm-up quarter
    print(f"{sum(v) / len(v):8.1f} us  x{len(v):4d}  {k[:100]}")
