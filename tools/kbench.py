#!/usr/bin/env python3
"""Per-kernel micro-benchmark of the conv family at the BASELINE shapes (B = 4, 128^3 model), through the C ABI.

    python tools/kbench.py [filter-substring ...] [--dtype bf16|f32] [--iters N]

Prints, per case, the mean launch time (HIP events over N back-to-back launches), algorithmic TFLOP/s and the
algorithmic HBM GB/s (inputs + outputs once).  Used to iterate on one kernel at a time; bench.py remains the metric."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from causal_vae_amd import ops  # noqa: E402

# name, kind, B, (sd, sh, sw), Cs, Cl   (L spatial = 2x S spatial)
CASES = [
    ("enc1.fwd", "down", 4, (64, 64, 64), 32, 1), ("enc2.fwd", "down", 4, (32, 32, 32), 64, 32),
    ("enc3.fwd", "down", 4, (16, 16, 16), 128, 64), ("enc4.fwd", "down", 4, (8, 8, 8), 256, 128),
    ("enc2.bwd_data", "up", 4, (32, 32, 32), 64, 32), ("enc3.bwd_data", "up", 4, (16, 16, 16), 128, 64),
    ("enc4.bwd_data", "up", 4, (8, 8, 8), 256, 128),
    ("enc1.wgrad", "wgrad", 4, (64, 64, 64), 32, 1), ("enc2.wgrad", "wgrad", 4, (32, 32, 32), 64, 32),
    ("enc3.wgrad", "wgrad", 4, (16, 16, 16), 128, 64), ("enc4.wgrad", "wgrad", 4, (8, 8, 8), 256, 128),
    ("dec1.fwd", "up", 4, (4, 4, 4), 256, 128), ("dec2.fwd", "up", 4, (8, 8, 8), 128, 64),
    ("dec3.fwd", "up", 4, (16, 16, 16), 64, 32), ("dec4.fwd", "up", 4, (32, 32, 32), 32, 1),
    ("dec1.bwd_data", "down", 4, (4, 4, 4), 256, 128), ("dec2.bwd_data", "down", 4, (8, 8, 8), 128, 64),
    ("dec3.bwd_data", "down", 4, (16, 16, 16), 64, 32), ("dec4.bwd_data", "down", 4, (32, 32, 32), 32, 1),
    ("dec1.wgrad", "wgrad", 4, (4, 4, 4), 256, 128), ("dec2.wgrad", "wgrad", 4, (8, 8, 8), 128, 64),
    ("dec3.wgrad", "wgrad", 4, (16, 16, 16), 64, 32), ("dec4.wgrad", "wgrad", 4, (32, 32, 32), 32, 1),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("filters", nargs="*")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--no-splitk", action="store_true", help="data kernels: no split-K scratch (unsplit launches)")
    ap.add_argument("--mask", action="store_true", help="data kernels: fuse a ReLU mask of the output's shape (the backward-data form)")
    ap.add_argument("--fp8", action="store_true", help="forward data kernels with C_in >= 32 on fp8 operands (cvae_conv_fp8: bf16 + fp8 output, amax record), as the fp8 training forward runs them")
    ap.add_argument("--fp8-plain", action="store_true", help="with --fp8: bf16 output only, no amax")
    ap.add_argument("--fp8-side", default="both", choices=["both", "out8", "amax"], help="with --fp8: which side outputs ride along")
    args = ap.parse_args()
    if args.no_splitk:
        ops.SPLIT_K = False
    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    esz = 2 if dt == torch.bfloat16 else 4
    dev = "cuda"
    tot = 0.0
    for name, kind, B, sp, Cs, Cl in CASES:
        if args.filters and not any(f in name for f in args.filters):
            continue
        lp = tuple(2 * s for s in sp)
        S = (torch.randn(B, *sp, Cs, device=dev) * 0.5).to(dt)
        Lt = (torch.randn(B, *lp, Cl, device=dev) * 0.5).to(dt)
        w = torch.randn(Cs, Cl, 4, 4, 4, device=dev) * 0.05
        bias_s, bias_l = torch.randn(Cs, device=dev), torch.randn(Cl, device=dev)
        if args.fp8 and kind in ("down", "up") and name.endswith(".fwd") and min(Cs, Cl) >= 32:
            src = Lt if kind == "down" else S
            xq = ops.quantize_fp8(src.abs(), float(src.abs().max()) / 448.0)
            wq = ops.pack_weight_fp8(w, 3, kind == "up", float(w.abs().max()) / 448.0)
            amax = torch.zeros(ops.AMAX_SLOTS, dtype=torch.int32, device=dev)
            dsc = torch.tensor([1e-3, 1.0], device=dev)
            bb = bias_s if kind == "down" else bias_l
            co = Cs if kind == "down" else Cl
            if args.fp8_plain:
                fn = lambda: ops.conv_fp8(kind == "up", xq, wq, bb, co, 3, "relu", dscale=dsc)
            else:
                fn = lambda: ops.conv_fp8(kind == "up", xq, wq, bb, co, 3, "relu", dscale=dsc, want_out8=args.fp8_side != "amax", amax=amax if args.fp8_side != "out8" else None)
        elif kind == "down":
            wp = ops.pack_weight(w, 3, False, dt)
            fn = (lambda: ops._conv_down(Lt, wp, None, S, Cs, 3, None)) if args.mask else (lambda: ops._conv_down(Lt, wp, bias_s, None, Cs, 3, "relu"))
        elif kind == "up":
            wp = ops.pack_weight(w, 3, True, dt)
            fn = (lambda: ops._conv_up(S, wp, None, Lt, Cl, 3, None)) if args.mask else (lambda: ops._conv_up(S, wp, bias_l, None, Cl, 3, "relu"))
        else:
            fn = lambda: ops._conv_wgrad(S, Lt, 3, w.shape, want_sbias=True)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.iters
        npos = B * sp[0] * sp[1] * sp[2]
        flops = 2.0 * npos * Cs * Cl * 64
        byts = (S.numel() + Lt.numel()) * esz
        tot += ms
        print(f"{name:16s} {kind:6s} {ms * 1e3:9.1f} us  {flops / ms / 1e9:8.1f} TFLOP/s  {byts / ms / 1e6:8.1f} GB/s   (S {tuple(S.shape)} L {tuple(Lt.shape)})")
    print(f"total {tot * 1e3:.1f} us")


if __name__ == "__main__":
    main()
