#!/usr/bin/env python3
"""Per-kernel HBM-side traffic of a workload from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs):
    tools/pmc_generic.py <dir with pmc_fetch/ and pmc_write/> <once-per-step kernel substring> [per-step count of that kernel]
Units / corrections as tools/pmc_traffic.py (MI355X_MICROARCH.md "HBM"): KiB counters; FETCH_SIZE doubled on gfx950 (wide coalesced reads);
Infinity-Cache hits are counted (the counters sit on the L2's fabric side).  Prints MB per step per kernel name, sorted by total."""
import collections, csv, glob, os, sys
root, marker = sys.argv[1], sys.argv[2]
per_step = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0


def load(d, scale):
    f = max(glob.glob(os.path.join(root, d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    tot, calls, seen = collections.defaultdict(float), collections.Counter(), set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:96]
        tot[k] += float(r["Counter_Value"]) * 1024.0 * scale
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"]); calls[k] += 1
    return tot, calls


fe, calls = load("pmc_fetch", 2.0)
wr, _ = load("pmc_write", 1.0)
steps = sum(c for k, c in calls.items() if marker in k) / per_step
print(f"steps seen: {steps:.0f} (marker '{marker}')")
rows = sorted(set(fe) | set(wr), key=lambda k: -(fe.get(k, 0) + wr.get(k, 0)))
tf = tw = 0.0
for k in rows:
    f, w = fe.get(k, 0) / steps / 1e6, wr.get(k, 0) / steps / 1e6
    tf += f; tw += w
    if f + w >= 0.5:
        print(f"{f:10.1f} MB read {w:10.1f} MB written  {calls[k] / steps:5.1f}x/step  {k}")
print(f"{tf:10.1f} MB read {tw:10.1f} MB written  per step, all kernels")
