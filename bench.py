#!/usr/bin/env python3
"""bench.py — training samples/sec of the 3D vessel CausalVAE step on MI355X (BASELINE.json's metric).

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

One "step" = zero_grad -> forward -> ELBO -> backward -> (RCCL SUM all-reduce of the flat gradient bucket) -> Adam, on one
batch of synthetic 128^3 volumes (B = 4 per GPU, bf16 conv arithmetic, fp32 heads/losses/master weights).  Inputs are
resident in HBM before the timed region.  Rank 0 prints ONE JSON line with the whole-job throughput plus
  roofline     : the dominant conv kernel's algorithmic FLOP/s (HIP events around each of its launches inside the timed
                 region) against the dense bf16 MFMA peak of /opt/skills/guides/MI355X_MICROARCH.md (2.5 PFLOP/s);
  cpu_baseline : the CPU oracle's train step (oracle/, pinned to the reference by golden vectors) timed on this node's
                 host cores on a bounded sample of the same workload — a reported baseline, not the target.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_FLOPS = 2.5e15     # dense bf16 MFMA, MI355X_MICROARCH.md "Chip-level parameters"
PEAK_F32_FLOPS = 157.3e12    # fp32 MFMA / vector peak
METRIC = "training samples/sec on 128^3 vessel volumes (3D CausalVAE train step)"


def conv_flops(name):
    """Algorithmic FLOPs (2*MACs, no padding / zero-insertion counted: SURVEY.md §8(d)) of one conv launch from its timer label."""
    import re
    if name.startswith("conv_wgrad_multi"):                 # one grouped launch: the sum over its layers
        head, layers = name.rsplit(" ", 1)
        nd_, B_ = re.search(r"nd(\d)", head).group(1), re.search(r" B(\d+)", head).group(1)
        tot = 0.0
        for lay in layers.split(";"):
            mm = re.match(r"S(\d+)x(\d+)x(\d+)x(\d+)L(\d+)", lay)
            tot += conv_flops(f"conv_wgrad nd{nd_} B{B_} S{mm.group(1)}x{mm.group(2)}x{mm.group(3)}x{mm.group(4)} L{mm.group(5)}")
        return tot
    nd = int(re.search(r"nd(\d)", name).group(1))
    B = int(re.search(r" B(\d+)", name).group(1))
    taps = 64 if nd == 3 else 16
    if name.startswith("conv_down"):
        ld, lh, lw, Cl = map(int, re.search(r"L(\d+)x(\d+)x(\d+)x(\d+)", name).groups())
        Cs = int(re.search(r"-> S(\d+)", name).group(1))
        pos = (ld // 2 if nd == 3 else 1) * (lh // 2) * (lw // 2)
    else:
        sd, sh, sw, Cs = map(int, re.search(r"S(\d+)x(\d+)x(\d+)x(\d+)", name).groups())
        Cl = int(re.search(r"L(\d+)$", name).group(1))
        pos = sd * sh * sw
    return 2.0 * B * pos * Cs * Cl * taps


def make_batch(B, size, seed, device):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 1, size, size, size, generator=g)            # z-scored volumes (causal_cascade/dataset.py:132-134)
    m = torch.rand(B, 12, generator=g)                              # min-max normalised morphology (dataset.py:148)
    t = torch.randint(0, 19, (B,), generator=g)
    eps = torch.randn(B, 64, generator=g)
    return tuple(v.to(device) for v in (x, m, t, eps))


def cpu_baseline(B, size, budget_s, x, m, t, eps, lr=1e-3, threads=16):
    """Oracle train steps on the host CPU: bounded sample (>= 2 timed steps, stops after ~budget_s seconds)."""
    import oracle
    torch.set_num_threads(threads)
    sd = oracle.init_state_dict("bio3d", seed=42)
    state, losses, times = None, [], []
    t_all = time.time()
    step = 0
    while True:
        t0 = time.time()
        st = oracle.cascade_train_step(sd, x, m, t, eps, adam_state=state, nd=3, lr=lr)
        state = st["adam_state"]
        dt = time.time() - t0
        if step >= 1:
            times.append(dt)                                         # step 0 = warm-up (allocator, mkldnn primitive cache)
        losses.append(float(st["loss"]))
        step += 1
        if (len(times) >= 2 and time.time() - t_all > budget_s) or len(times) >= 20:
            break
    times.sort()
    med = times[len(times) // 2]
    return dict(value=B / med, unit="samples/s", cores=torch.get_num_threads(), kind="port",
                sample=f"{len(times)} timed oracle train steps (fp32, B={B}, {size}^3) after 1 warm-up; median {med * 1e3:.0f} ms/step",
                losses=[v if v == v and abs(v) != float("inf") else None for v in losses[:4]]), losses[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", type=int, default=128, help="volume edge (128 = BASELINE config)")
    ap.add_argument("--batch", type=int, default=4, help="samples per GPU")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the cpu_baseline leg (0 disables)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the cpu_baseline leg (0 = the box's CPU share: min(affinity, 16 per GPU))")
    ap.add_argument("--lr", type=float, default=1e-4, help="Adam learning rate (the reference uses 1e-3, causal_cascade/main.py:50, at which the 3D lift diverges on step 3 in the oracle too: DESIGN.md)")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--overlap-adam", action="store_true", help="run the non-encoder part of the Adam update on a side stream under the encoder backward")
    ap.add_argument("--fork", action="store_true", help="run weight-gradient kernels on a side stream (overlap with data-gradient kernels)")
    ap.add_argument("--fork-max-positions", type=int, default=0, help="with --fork: only layers with at most this many S positions per batch fork (0 = all)")
    ap.add_argument("--defer-join", action="store_true", help="with --fork: join the side stream once before the optimizer instead of after every layer")
    ap.add_argument("--no-graph", action="store_true", help="issue the step eagerly instead of replaying the captured HIP graph")
    ap.add_argument("--no-defer-wgrad", action="store_true", help="compute every conv weight gradient in its own launch (A/B of the grouped end-of-backward launch)")
    ap.add_argument("--no-overlap-exchange", action="store_true", help="N > 1: one all-reduce after the whole backward instead of the split backward")
    ap.add_argument("--force-overlap-exchange", action="store_true", help="take the split-backward capture also at N = 1 (no exchange happens)")
    ap.add_argument("--roofline-steps", type=int, default=5, help="eager steps with per-launch HIP events after the timed region")
    args = ap.parse_args()

    from causal_vae_amd import FusedAdam, _lib
    from causal_vae_amd.causal_cascade import CausalBioVAE3D, loss_function, train_step
    from causal_vae_amd.parallel import GradAllReducer, broadcast_parameters, init_distributed
    import torch.distributed as dist

    if args.no_defer_wgrad:
        from causal_vae_amd import ops as _ops1
        _ops1.DEFER_WGRAD = False
    if args.fork:
        from causal_vae_amd import ops as _ops0
        _ops0.FORK_BACKWARD = True
        _ops0.FORK_MAX_POSITIONS = args.fork_max_positions
        _ops0.DEFER_JOIN = args.defer_join
    rank, world, local_rank = init_distributed()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU path")
    if os.environ.get("CVAE_DIST_BACKEND") == "gloo":            # rehearsal: all ranks on the one visible card
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32

    torch.manual_seed(42)                                            # causal_cascade/main.py:28
    model = CausalBioVAE3D().to(dev).train().set_compute_dtype(dtype)
    broadcast_parameters(model)
    opt = FusedAdam(model.parameters(), lr=args.lr, device_step=True)   # main.py:50
    if world == 1 and args.overlap_adam:
        opt.overlap_backward(model.early_gradient_parameters())        # measured: -4 % (the streaming update slows the co-running conv kernels more than it hides)
    reducer = GradAllReducer(model.parameters()) if world > 1 else None
    x, m, t, eps = make_batch(args.batch, args.size, 1234 + rank, dev)

    # ELBO of the first batch at the initial weights (compared with the oracle's below, rank 0 / N = 1)
    with torch.no_grad():
        out = model(x, m, t, eps=eps)
        elbo0 = float(loss_function(out[0], x, out[1], m, out[2], out[3])[0])
    del out

    def eager_step():
        return train_step(model, opt, x, m, t, grad_hook=reducer)

    use_graph = not args.no_graph
    n_pre, want_split = 0, False
    if use_graph:
        from causal_vae_amd.graph import GraphedTrainStep
        # N > 1: the backward is split at the encoder output so the decoder + bottleneck gradients are exchanged under the encoder's backward
        want_split = (world > 1 and not args.no_overlap_exchange) or args.force_overlap_exchange
        try:
            gstep = GraphedTrainStep(model, opt, (x, m, t), None, reducer=reducer, warmup=3, overlap_exchange=want_split)   # None: model.forward_elbo, as train_step
        except Exception as e:                                       # noqa: BLE001 - the plain two-graph exchange is always available
            if not want_split:
                raise
            print(f"[bench] split-backward capture failed ({e!r}); falling back to one exchange after the backward", file=sys.stderr)
            want_split = False
            gstep = GraphedTrainStep(model, opt, (x, m, t), None, reducer=reducer, warmup=3)
        n_pre = 3                                                    # the capture warm-up runs 3 real steps
        step = lambda: gstep()
    else:
        step = eager_step

    traj = []
    for _ in range(max(args.warmup - n_pre, 1)):
        traj.append(step()[0].clone())
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()[0]
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    traj.append(loss.clone())
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    # ---- roofline leg: the same step issued eagerly with HIP events around every conv launch (events cannot sit inside a
    # replayed graph); same process, same buffers, directly after the timed region ----
    timer = None
    if not args.no_kernel_timer and args.roofline_steps > 0:
        from causal_vae_amd import ops as _ops
        _ops.FORK_BACKWARD = False                                   # one stream: every launch is timed alone
        timer = _lib.KernelTimer()
        eager_step()
        torch.cuda.synchronize()
        _lib.TIMER = timer
        for _ in range(args.roofline_steps):
            eager_step()
        torch.cuda.synchronize()
        _lib.TIMER = None
    fin = lambda v: float(v) if torch.isfinite(torch.as_tensor(float(v))) else None
    final_loss = fin(loss)
    traj = [fin(v) for v in traj]

    if rank == 0:
        total = world * args.batch * args.steps
        res = {
            "metric": METRIC, "value": total / elapsed, "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"3D vessel CausalVAE train step, {args.size}^3 {args.dtype} volumes, batch {args.batch}/GPU, "
                                   f"Adam lr {args.lr:g}, ELBO = MSE-sum + 2000*MSE-sum(m) + KLD", "global_batch": world * args.batch,
                       "per_gpu_batch": args.batch, "volume": [args.size] * 3, "parallelism": f"dp{world}",
                       "params": sum(p.numel() for p in model.parameters())},
            "exchange": ("split backward: decoder + bottleneck bucket all-reduced under the encoder backward" if (use_graph and want_split) else
                         ("one all-reduce after the backward" if world > 1 else "none (1 rank)")),
            "final_loss": final_loss, "loss_trajectory": traj if len(traj) <= 6 else traj[:4] + traj[-2:], "lr": args.lr, "hip_graph": use_graph, "side_stream_fork": args.fork,
        }
        if timer is not None:
            summ = timer.summary()
            per_step = {k: (n / args.roofline_steps, ms) for k, (n, ms) in summ.items()}
            # dominant kernel = the heaviest single-kernel conv launch (a conv_wgrad call is two kernels, main + slab reduction, so its
            # event time has no single row in the rocprof summary to agree with; it is listed under "kernels" like everything else)
            # Among launches within 10 % of the heaviest, an `up` launch is preferred: enc2's backward-data is the only user of its
            # template instance, so it has its own row in the rocprof summary, while the `down` rows average three layers.
            single = [k for k in summ if k.startswith(("conv_down", "conv_up"))] or list(summ)
            cost = lambda k: summ[k][0] * summ[k][1]
            top = max(cost(k) for k in single)
            dom = max((k for k in single if cost(k) >= 0.9 * top), key=lambda k: (k.startswith("conv_up"), cost(k)))
            n, ms = summ[dom]
            fl = conv_flops(dom)
            peak = PEAK_BF16_FLOPS if args.dtype == "bf16" else PEAK_F32_FLOPS
            ach = fl / (ms * 1e-3)
            traffic = None
            tfile = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
            if os.path.exists(tfile):                                # HBM bytes per launch from the committed rocprofv3 --pmc passes
                traffic = json.load(open(tfile)).get(dom, {}).get("hbm_bytes_per_launch")
            res["roofline"] = {"bound": "mfma", "kernel": dom, "achieved": ach / 1e12, "peak": peak / 1e12, "unit": "TFLOP/s",
                               "frac": ach / peak, "traffic": traffic, "avg_ms": ms, "launches_per_step": n / args.roofline_steps,
                               "timing": f"HIP events around each launch over {args.roofline_steps} eager steps run right after the timed region",
                               "algorithmic_gflop_per_launch": fl / 1e9}
            conv_ms = sum(n_ * ms_ for n_, ms_ in summ.values()) / args.roofline_steps
            res["conv_ms_per_step"] = conv_ms
            res["kernels"] = {k: {"per_step": v[0], "avg_ms": round(v[1], 4), "tflops": round(conv_flops(k) / (v[1] * 1e-3) / 1e12, 1)}
                              for k, v in sorted(per_step.items(), key=lambda kv: -kv[1][0] * kv[1][1])}
        if world == 1 and args.cpu_seconds > 0:
            aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            threads = args.cpu_threads if args.cpu_threads > 0 else min(aff, 16)
            cb, elbo_ref = cpu_baseline(args.batch, args.size, args.cpu_seconds, x.cpu(), m.cpu(), t.cpu(), eps.cpu(), args.lr, threads)
            res["cpu_baseline"] = cb
            res["elbo_rel_err"] = abs(elbo0 - elbo_ref) / abs(elbo_ref)
            res["gpu_over_cpu"] = res["value"] / cb["value"]
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
