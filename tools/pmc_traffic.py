#!/usr/bin/env python3
"""Turn the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as MI355X_MICROARCH.md prescribes) of
`bench.py --no-graph` into profiles/r01_pmc_traffic.json: HBM bytes per launch of the step's heavy kernels, keyed by the
labels bench.py uses.  Units / corrections (MI355X_MICROARCH.md "HBM"): both counters are in KiB; on gfx950 FETCH_SIZE
reports exactly half of the bytes of wide (16 B/lane) coalesced reads, which is what these kernels issue, so the read
side is doubled; WRITE_SIZE is exact for 16-B streaming stores."""
import collections, csv, glob, json, os, sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"


def load(d):
    f = max(glob.glob(os.path.join(root, d, "*", "*counter_collection.csv")), key=os.path.getmtime)
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[(r["Kernel_Name"], r["Grid_Size"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


fe, wr = load("pmc_fetch"), load("pmc_write")
# (substring of the kernel name, grid size in threads) -> bench.py label, for the B=4, 128^3 bf16 workload
WANT = {
    ("4, 1, 2, 1, 0, 1>", "1048576"): "conv_up nd3 B4 S32x32x32x64 -> L32",
    ("conv_wgrad_kernelIDF16bLi3E", "131072"): "conv_wgrad nd3 B4 S32x32x32x64 L32",
    ("conv_data_kernelIDF16bLi3ELb0ELi2ELi2ELi2ELi1ELi1E", "262144"): "conv_down nd3 B4 L64x64x64x32 -> S64",
    ("adam_multi_kernel", None): "adam_multi (15.35 M params)",
}
out = {}
for (name, grid), f_kib in fe.items():
    for (pat, g), label in WANT.items():
        if pat in name and (g is None or g == grid):
            w_kib = wr.get((name, grid), 0.0)
            if label in out and out[label]["fetch_size_kib_raw"] > f_kib:
                continue                                     # several grids match: keep the heaviest dispatch
            out[label] = {"fetch_size_kib_raw": f_kib, "write_size_kib": w_kib,
                          "hbm_read_bytes_corrected": 2 * f_kib * 1024, "hbm_write_bytes": w_kib * 1024,
                          "hbm_bytes_per_launch": 2 * f_kib * 1024 + w_kib * 1024}
json.dump(out, open(os.path.join("profiles", "r01_pmc_traffic.json"), "w"), indent=1)
for k, v in out.items():
    print(f"{k:48s} read {v['hbm_read_bytes_corrected'] / 1e6:8.1f} MB  write {v['hbm_write_bytes'] / 1e6:8.1f} MB")
