#!/usr/bin/env python3
"""SQ counters of one train step (rocprofv3 --pmc SQ_* pass of bench.py --no-graph): per kernel kind, MFMA-busy share of the SIMD cycles and LDS
conflict share.  tools/pmc_sq.py <dir>"""
import collections, csv, glob, os, sys
f = max(glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
per = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    e = per.setdefault(int(r["Dispatch_Id"]), [r["Kernel_Name"], {}])
    e[1][r["Counter_Name"]] = e[1].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
rows = [per[k] for k in sorted(per)]
idx = [i for i, (n, _) in enumerate(rows) if "adam_multi" in n]
step = rows[idx[-2] + 1: idx[-1] + 1]
agg = collections.OrderedDict()
for n, c in step:
    k = n[:64]
    a = agg.setdefault(k, collections.Counter()); a.update(c); a["_n"] += 1
# SQ_BUSY_CYCLES is summed over the 32 shader engines (8 XCDs x 4), SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs: with the SQs busy for the whole
# kernel, MFMA-pipe utilisation = MFMA_BUSY / (1024 * cycles) = MFMA_BUSY / (32 * SQ_BUSY).  (Cross-check: a kernel of 1.05 M 32x32x16 bf16 MFMAs counted
# 33.5 M MFMA-busy cycles = 32 per MFMA.)
print("kernel (first 64 chars of the name)                                 launches  SQ busy Mcyc  MFMA pipe busy  LDS conflict / LDS active")
tb = tm = 0.0
for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["SQ_BUSY_CYCLES"]):
    busy, mf = a["SQ_BUSY_CYCLES"], a["SQ_VALU_MFMA_BUSY_CYCLES"]
    tb += busy; tm += mf
    print(f"{k:66s} {int(a['_n']):6d} {busy / 1e6:12.2f} {100 * mf / max(32 * busy, 1):13.1f} % {a['SQ_LDS_BANK_CONFLICT'] / max(a['SQ_LDS_IDX_ACTIVE'], 1):18.3f}")
print(f"whole step: MFMA pipe busy {100 * tm / max(32 * tb, 1):.1f} % of the SIMD cycles")
