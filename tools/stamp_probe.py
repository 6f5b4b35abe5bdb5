#!/usr/bin/env python3
"""Phase timeline of conv_data_kernel workgroups (needs a library built with `make -C causal_vae_amd/csrc EXTRA=-DCVAE_STAMP`).

    python tools/stamp_probe.py enc2.bwd_data

Launches one kbench case, then prints per-phase shader-clock deltas (median over workgroups), the workgroup lifetime in
wall-clock ns, and how many workgroups shared a CU at the same time."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from causal_vae_amd import ops, _lib  # noqa: E402
from kbench import CASES  # noqa: E402

name = sys.argv[1]
case = [c for c in CASES if c[0] == name][0]
_, kind, B, sp, Cs, Cl = case
dt = torch.bfloat16
lp = tuple(2 * s for s in sp)
S = (torch.randn(B, *sp, Cs, device="cuda") * 0.5).to(dt)
Lt = (torch.randn(B, *lp, Cl, device="cuda") * 0.5).to(dt)
w = torch.randn(Cs, Cl, 4, 4, 4, device="cuda") * 0.05
FP8 = "--fp8" in sys.argv            # the fp8 forward form of the same product (dual output + amax, as in the training step)
if FP8 and kind in ("down", "up"):
    src = Lt if kind == "down" else S
    xq = ops.quantize_fp8(src.abs(), float(src.abs().max()) / 448.0)
    wq = ops.pack_weight_fp8(w, 3, kind == "up", float(w.abs().max()) / 448.0)
    amax = torch.zeros(ops.AMAX_SLOTS, dtype=torch.int32, device="cuda")
    plain = "--plain" in sys.argv      # no second output, no amax
    fn = lambda: ops.conv_fp8(kind == "up", xq, wq, None, Cs if kind == "down" else Cl, 3, "relu", acc_scale=1e-3, out8_scale=None if plain else 1.0, amax=None if plain else amax)
elif kind == "down":
    wp = ops.pack_weight(w, 3, False, dt); fn = lambda: ops._conv_down(Lt, wp, None, S, Cs, 3, None)
elif kind == "up":
    wp = ops.pack_weight(w, 3, True, dt); fn = lambda: ops._conv_up(S, wp, None, Lt, Cl, 3, None)
else:
    fn = lambda: ops._conv_wgrad(S, Lt, 3, w.shape)
for _ in range(3):
    fn()
torch.cuda.synchronize()
SL, NW = 32, 8192
buf = np.zeros(NW * SL, dtype=np.uint64)
lib = _lib.lib
lib.cvae_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert lib.cvae_debug_stamps(buf.ctypes.data, buf.size) == 0
st = buf.reshape(NW, SL).astype(np.int64)
st = st[st[:, 0] != 0]
print("workgroups stamped", len(st))
t0 = st[:, 30].min()
life = st[:, 31] - st[:, 30]
print("kernel span (wall clock, 10 ns ticks):", (st[:, 31].max() - t0) * 10, "ns; WG lifetime ns: median", np.median(life) * 10, "p10", np.percentile(life, 10) * 10, "p90", np.percentile(life, 90) * 10)
cyc = st[:, 27] - st[:, 0]
print("WG lifetime cycles: median", np.median(cyc), " => clock GHz ~", np.median(cyc / (life * 10.0)))
labels = {0: "start", 1: "index math done", 26: "mainloop done", 27: "stores drained"}
for c in range(8):
    labels[2 + 3 * c] = f"chunk{c}: loads issued + barrier"; labels[3 + 3 * c] = f"chunk{c}: LDS filled + barrier"; labels[4 + 3 * c] = f"chunk{c}: taps done"
if os.environ.get("STAMP_UPFULL"):
    labels = {0: "start", 1: "index math done", 2: "halo loads issued", 3: "halo in LDS + barrier", 27: "stores drained"}
    for c in range(8):
        labels[4 + 2 * c] = f"parity{c}: taps done"; labels[5 + 2 * c] = f"parity{c}: epilogue issued"
order = [i for i in [0, 1] + list(range(2, 26)) + [26, 27] if st[:, i].any()]
prev = order[0]
for i in order[1:]:
    d = st[:, i] - st[:, prev]
    print(f"  {labels[i]:36s} median {np.median(d):9.0f}  p10 {np.percentile(d, 10):9.0f}  p90 {np.percentile(d, 90):9.0f} cycles")
    prev = i
# concurrency: for each WG, how many WGs on the same (xcc, se, cu) overlap its midpoint
hw = st[:, 29]
xcc = (hw >> 32) & 0xF; cu = (hw >> 8) & 0xF; se = (hw >> 13) & 0x7; sh = (hw >> 12) & 1
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
mid = (st[:, 30] + st[:, 31]) // 2
conc = []
for k in np.unique(key):
    idx = np.where(key == k)[0]
    for i in idx:
        conc.append(int(((st[idx, 30] <= mid[i]) & (st[idx, 31] >= mid[i])).sum()))
print("distinct CUs seen", len(np.unique(key)), " concurrent WGs per CU: median", np.median(conc), "max", max(conc))
