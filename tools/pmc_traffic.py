#!/usr/bin/env python3
"""HBM traffic of one train step from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: separate runs, as MI355X_MICROARCH.md prescribes)
of `bench.py --no-graph`:  tools/pmc_traffic.py <dir with pmc_fetch/ and pmc_write/> [out.json]  ->  profiles/r03_pmc_traffic.json

Units / corrections (MI355X_MICROARCH.md "HBM"): both counters are in KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes of wide
(16 B/lane) coalesced reads, which is what these kernels issue, so the read side is doubled; WRITE_SIZE is exact for 16-B streaming stores.
Infinity-Cache hits are counted by both (the counters sit on the L2's fabric side): "HBM bytes" below means bytes that left the L2.

One step = the dispatches between the last two adam_multi_kernel dispatches.  Every dispatch is mapped to the bench.py timer label / kernel
family by its kind (kernel-name substring) and its occurrence number inside the step, which is fixed by the model (128^3, B = 4, bf16)."""
import collections, csv, glob, json, os, sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
out_path = sys.argv[2] if len(sys.argv) > 2 else os.path.join("profiles", "r03_pmc_traffic.json")

FAM_DOWN = "conv_data_kernel<DOWN> / down_c1 (conv forward, convT backward-data)"
FAM_UP = "conv_data_kernel<UP> / up_c1 (convT forward, conv backward-data)"
FAM_WG = "conv_wgrad_kernel + wgrad_reduce_kernel (weight gradients: main + slab reduction)"
B, ESZ = 4, 2
# the 12 conv_data_kernel dispatches of a step, in order: (label, family, algorithmic bytes = input + output (+ mask) activations + packed weights)
def act(s, c): return B * s ** 3 * c * ESZ
def bits(s, c): return B * s ** 3 * c // 8                  # a ReLU mask as bits (round 3): read by the backward-data launch, written by the forward launch
def wts(ci, co): return ci * co * 64 * ESZ
DATA_ORDER = [
    ("conv_down nd3 B4 L64x64x64x32 -> S64", FAM_DOWN, act(64, 32) + act(32, 64) + bits(32, 64) + wts(32, 64)),
    ("conv_down nd3 B4 L32x32x32x64 -> S128", FAM_DOWN, act(32, 64) + act(16, 128) + bits(16, 128) + wts(64, 128)),
    ("conv_down nd3 B4 L16x16x16x128 -> S256", FAM_DOWN, act(16, 128) + act(8, 256) + bits(8, 256) + wts(128, 256)),
    ("conv_up nd3 B4 S4x4x4x256 -> L128", FAM_UP, act(4, 256) + act(8, 128) + bits(8, 128) + wts(256, 128)),
    ("conv_up nd3 B4 S8x8x8x128 -> L64", FAM_UP, act(8, 128) + act(16, 64) + bits(16, 64) + wts(128, 64)),
    ("conv_up nd3 B4 S16x16x16x64 -> L32", FAM_UP, act(16, 64) + act(32, 32) + bits(32, 32) + wts(64, 32)),
    ("conv_down nd3 B4 L32x32x32x32 -> S64", FAM_DOWN, act(32, 32) + act(16, 64) + bits(16, 64) + wts(64, 32)),       # dec3 backward-data (+ the ReLU mask of its output, as bits)
    ("conv_down nd3 B4 L16x16x16x64 -> S128", FAM_DOWN, act(16, 64) + act(8, 128) + bits(8, 128) + wts(128, 64)),
    ("conv_down nd3 B4 L8x8x8x128 -> S256", FAM_DOWN, act(8, 128) + act(4, 256) + wts(256, 128)),                      # dec1 backward-data: dec_input's output is no ReLU output
    ("conv_up nd3 B4 S8x8x8x256 -> L128", FAM_UP, act(8, 256) + act(16, 128) + bits(16, 128) + wts(128, 256)),          # enc4 backward-data
    ("conv_up nd3 B4 S16x16x16x128 -> L64", FAM_UP, act(16, 128) + act(32, 64) + bits(32, 64) + wts(64, 128)),
    ("conv_up nd3 B4 S32x32x32x64 -> L32", FAM_UP, act(32, 64) + act(64, 32) + bits(64, 32) + wts(32, 64)),
]
C1_DOWN = [("conv_down nd3 B4 L128x128x128x1 -> S32", FAM_DOWN, B * 128 ** 3 * 4 + act(64, 32) + bits(64, 32)),            # fp32 image in
           ("conv_down nd3 B4 L64x64x64x1 -> S32", FAM_DOWN, B * 64 ** 3 * ESZ + act(32, 32) + bits(32, 32))]          # dec4 backward-data
C1_UP = [("conv_up nd3 B4 S32x32x32x32 -> L1", FAM_UP, act(32, 32) + B * 64 ** 3 * ESZ)]
WG_LAYERS = [(32, 64, 32), (16, 128, 64), (8, 256, 128), (4, 256, 128), (8, 128, 64), (16, 64, 32)]         # (S extent, Cs, Cl)
WG_ALG = sum(act(s, cs) + act(2 * s, cl) + cs * cl * 64 * 4 for s, cs, cl in WG_LAYERS)
C1_WG = [("conv_wgrad nd3 B4 S32x32x32x32 L1", FAM_WG, act(32, 32) + B * 64 ** 3 * ESZ + 32 * 64 * 4),
         ("conv_wgrad nd3 B4 S64x64x64x32 L1", FAM_WG, act(64, 32) + B * 128 ** 3 * 4 + 32 * 64 * 4)]


def load(d):
    f = max(glob.glob(os.path.join(root, d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    per = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        k = int(r["Dispatch_Id"])
        e = per.setdefault(k, [r["Kernel_Name"], 0.0])
        e[1] += float(r["Counter_Value"])
    rows = [per[k] for k in sorted(per)]
    idx = [i for i, (n, _) in enumerate(rows) if "adam_multi" in n]
    return rows[idx[-2] + 1: idx[-1] + 1]


fe, wr = load("pmc_fetch"), load("pmc_write")
assert [n for n, _ in fe] == [n for n, _ in wr], "the two passes must see the same dispatch sequence"
fams, labels = collections.defaultdict(lambda: dict(hbm_bytes_per_step=0.0, algorithmic_bytes_per_step=0.0, dispatches=0)), {}
cnt = collections.Counter()
last_fam = None
rows_out = []
for (name, f_kib), (_, w_kib) in zip(fe, wr):
    rd, wt = 2 * f_kib * 1024, w_kib * 1024
    label, fam, alg = None, None, 0.0
    if "conv_data_kernel" in name:
        label, fam, alg = DATA_ORDER[cnt["data"]]; cnt["data"] += 1
    elif "down_c1" in name:
        label, fam, alg = C1_DOWN[cnt["c1d"]]; cnt["c1d"] += 1
    elif "up_c1" in name:
        label, fam, alg = C1_UP[cnt["c1u"]]; cnt["c1u"] += 1
    elif "conv_splitk_finish" in name:
        fam = last_fam                                       # the slab sum of the data kernel in front of it: same label, no extra algorithmic bytes
        label = last_label
    elif "wgrad_c1_kernel" in name:
        label, fam, alg = C1_WG[cnt["c1w"]]; cnt["c1w"] += 1
    elif "wgrad_c1_finish" in name:
        fam, label = FAM_WG, last_label
    elif "conv_wgrad_kernel" in name:
        label, fam, alg = "conv_wgrad_multi (six layers)", FAM_WG, WG_ALG
    elif "wgrad_reduce" in name:
        label, fam = "conv_wgrad_multi (six layers)", FAM_WG
    elif "adam_multi" in name:
        label, fam, alg = "adam_multi (15.35 M params)", "adam_multi_kernel", 15346957 * 28
    else:
        fam = name.split("(")[0].split("<")[0].split("::")[-1].strip()[:60]
    last_fam, last_label = fam, label
    F = fams[fam]
    F["hbm_bytes_per_step"] += rd + wt; F["algorithmic_bytes_per_step"] += alg; F["dispatches"] += 1
    if label:
        e = labels.setdefault(label, dict(hbm_read_bytes_corrected=0.0, hbm_write_bytes=0.0, hbm_bytes_per_launch=0.0, algorithmic_bytes_per_launch=0.0))
        e["hbm_read_bytes_corrected"] += rd; e["hbm_write_bytes"] += wt; e["hbm_bytes_per_launch"] += rd + wt; e["algorithmic_bytes_per_launch"] += alg
    rows_out.append((name[:70], rd, wt))
out = {"_about": "tools/pmc_traffic.py: one train step (128^3, B=4, bf16) of bench.py --no-graph; FETCH_SIZE doubled (gfx950), WRITE_SIZE as is; bytes that left the L2 "
                 "(Infinity-Cache hits included)", "_step_total_bytes": sum(r + w for _, r, w in rows_out)}
out.update(fams); out.update(labels)
json.dump(out, open(out_path, "w"), indent=1)
print(f"step total: {out['_step_total_bytes'] / 1e6:.1f} MB over {len(rows_out)} dispatches")
for k, v in sorted(fams.items(), key=lambda kv: -kv[1]["hbm_bytes_per_step"])[:14]:
    print(f"{k[:86]:86s} {v['hbm_bytes_per_step'] / 1e6:8.1f} MB  (algorithmic {v['algorithmic_bytes_per_step'] / 1e6:7.1f} MB, {v['dispatches']} dispatches)")
