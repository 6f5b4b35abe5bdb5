#!/usr/bin/env python3
"""bench.py — training samples/sec of the 3D vessel CausalVAE step on MI355X (BASELINE.json's metric).

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

One "step" = zero_grad -> forward -> ELBO -> backward -> (RCCL SUM all-reduce of the flat gradient bucket) -> Adam, on one
batch of synthetic 128^3 volumes (B = 4 per GPU, bf16 conv arithmetic, fp32 heads/losses/master weights).  Inputs are
resident in HBM before the timed region.  The timed region is EXACTLY K steps between barrier + synchronize brackets; it is
repeated until >= 0.2 s have been timed and the MEDIAN repetition is reported (`timed_reps`, min / max alongside): one
repetition of 20 steps is 19 ms, the same order as box-to-box noise.  Rank 0 prints ONE JSON line with the whole-job throughput plus
  roofline     : the kernel family with the largest total time per step (HIP events around each of its launches, on the stream
                 they are launched on, over eager steps run right after the timed region): algorithmic FLOP/s against the dense
                 bf16 MFMA peak of /opt/skills/guides/MI355X_MICROARCH.md (2.5 PFLOP/s); `traffic` = HBM bytes per launch from
                 the committed rocprofv3 --pmc passes of THIS round's kernels (file and commit named in the line); `step` = the
                 whole step's algorithmic FLOPs and bytes against both peaks; `top` = the five heaviest launches;
  cpu_baseline : the CPU oracle's train step (oracle/, pinned to the reference by golden vectors) timed on this node's
                 host cores on a bounded sample of the same workload — a reported baseline, not the target;
  elbo_rel_err / elbo_rel_err_after_k : ELBO of the first batch at the initial weights, and after K = 5 Adam steps with injected
                 noise, against the oracle's.

Other workloads of BASELINE.json (`--workload`): vol64-f32 (configs[2]), mnist (configs[1]), decode (configs[4]: 240 stacked
counterfactual decodes per rank), vol128-vessel (the vessel recipe's loss on binary ~10 % volumes).
"""
import argparse
import json
import os
import re
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_FLOPS = 2.5e15     # dense bf16 MFMA, MI355X_MICROARCH.md "Chip-level parameters"
PEAK_FP8_FLOPS = 5.0e15      # dense fp8 on the block-scaled MFMA (same table)
PEAK_F32_FLOPS = 157.3e12    # fp32 MFMA / vector peak
PEAK_HBM = 8.0e12            # HBM3E spec (6.3 TB/s achievable)
METRIC = "training samples/sec on 128^3 vessel volumes (3D CausalVAE train step)"
TRAFFIC_FILE = os.path.join("profiles", "r03_pmc_traffic.json")


def conv_flops(name):
    """Algorithmic FLOPs (2*MACs, no padding / zero-insertion counted: SURVEY.md §8(d)) of one conv launch from its timer label."""
    if name.startswith("conv_wgrad_multi"):                 # one grouped launch: the sum over its layers
        head, layers = name.rsplit(" ", 1)
        nd_, B_ = re.search(r"nd(\d)", head).group(1), re.search(r" B(\d+)", head).group(1)
        tot = 0.0
        for lay in layers.split(";"):
            mm = re.match(r"S(\d+)x(\d+)x(\d+)x(\d+)L(\d+)", lay)
            tot += conv_flops(f"conv_wgrad nd{nd_} B{B_} S{mm.group(1)}x{mm.group(2)}x{mm.group(3)}x{mm.group(4)} L{mm.group(5)}")
        return tot
    if name.startswith("linear_"):
        M, K, N = (int(v) for v in re.search(r"M(\d+) K(\d+) N(\d+)", name).groups())
        return 2.0 * M * K * N
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


def family(label):
    """Kernel family of a timer label = one row of the rocprofv3 summary (the template instance serving several layers)."""
    for pre, fam in (("conv_wgrad", "conv_wgrad_kernel + wgrad_reduce_kernel (weight gradients: main + slab reduction)"),
                     ("conv_down", "conv_data_kernel<DOWN> / down_c1 (conv forward, convT backward-data)"),
                     ("conv_up", "conv_data_kernel<UP> / up_c1 (convT forward, conv backward-data)"),
                     ("linear_", "linear layers (gemm / skinny kernels)")):
        if label.startswith(pre):
            return fam
    return label


def step_algorithmic(B, size, esz, n_params, conv_params):
    """Algorithmic FLOPs and HBM bytes of one train step: conv FLOPs = fwd + bwd-data + bwd-weight (no bwd-data for the
    first layer); activation bytes (7V + 6A) * esz per sample when the decoder output is resized ((3V + 6A) * esz at the native 64^3),
    Adam 28 B/param, weights: fp32 dense layers read fwd, read + written bwd; packed conv weights written once, read fwd and bwd."""
    chans = [(1, 32), (32, 64), (64, 128), (128, 256)]
    fl, A, s = 0.0, 0, size
    for i, (ci, co) in enumerate(chans):                     # encoder: output extent s / 2
        s //= 2
        f = 2.0 * B * s ** 3 * ci * co * 64
        fl += f * (3 if i else 2)
        A += s ** 3 * co
    d = 4
    for i, (ci, co) in enumerate([(256, 128), (128, 64), (64, 32), (32, 1)]):    # decoder: input extent d
        fl += 3 * 2.0 * B * d ** 3 * ci * co * 64
        d *= 2
        A += d ** 3 * co
    V = size ** 3
    act = B * ((7 * V + 6 * A) if size != 64 else (3 * V + 6 * A)) * esz
    dense = n_params - conv_params
    byt = act + 28 * n_params + 3 * 4 * dense + conv_params * (4 + 3 * esz)
    return fl, byt


def make_batch(B, size, seed, device, binary=False):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 1, size, size, size, generator=g)            # z-scored volumes (causal_cascade/dataset.py:132-134)
    if binary:                                                      # vessel recipe: min-max + mean-threshold binarised, sparse (vessel .../dataset.py:236-237)
        import torch.nn.functional as F
        sm = F.avg_pool3d(x, 5, 1, 2)
        x = (sm > torch.quantile(sm.flatten()[:: max(1, sm.numel() // 1000000)], 0.90)).float()
    m = torch.rand(B, 12, generator=g)                              # min-max normalised morphology (dataset.py:148)
    t = torch.randint(0, 19, (B,), generator=g)
    eps = torch.randn(B, 64, generator=g)
    return tuple(v.to(device) for v in (x, m, t, eps))


def cpu_info():
    """Host description for the cpu_baseline leg: logical CPUs, the affinity mask, the PHYSICAL cores inside it (distinct
    thread-sibling sets: SURVEY.md §8(d) asks for "all physical cores of the node"), the cgroup CPU quota if one is set, the model."""
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    cpus = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    cores = set()
    for c in cpus:
        try:
            cores.add(open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list").read().strip())
        except OSError:
            cores.add(str(c))
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    return dict(os_cpu_count=os.cpu_count(), affinity=len(cpus), physical_cores=len(cores), cgroup_cpu_quota=quota, cpu_model=model)


def cpu_baseline(B, size, budget_s, x, m, t, eps, lr, threads, min_steps):
    """Oracle train steps on the host CPU: bounded sample (>= min_steps steps, stops after ~budget_s seconds)."""
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
        if (len(losses) >= min_steps and len(times) >= 2 and time.time() - t_all > budget_s) or len(times) >= 20:
            break
    times.sort()
    med = times[len(times) // 2]
    return dict(value=B / med, unit="samples/s", cores=torch.get_num_threads(), kind="port",
                sample=f"{len(times)} timed oracle train steps (fp32, B={B}, {size}^3) after 1 warm-up; median {med * 1e3:.0f} ms/step",
                losses=[v if v == v and abs(v) != float("inf") else None for v in losses[:8]]), losses


def timed_region(step, steps, warmup, world, dev, min_total_s):
    """W warm-up steps, then repetitions of EXACTLY `steps` steps, each bracketed by barrier + synchronize on both sides and reduced with MAX
    over ranks; repeated until min_total_s have been timed.  Returns (per-repetition seconds, last step's outputs)."""
    import torch.distributed as dist
    out = None
    for _ in range(warmup):
        out = step()
    reps = []
    while True:
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        el = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())                                    # the same number on every rank: they all stop together
        reps.append(el)
        if sum(reps) >= min_total_s or len(reps) >= 64:
            return reps, out


def timing_fields(reps, steps, samples_per_step):
    srt = sorted(reps)
    med = srt[len(srt) // 2]
    return {"value": samples_per_step * steps / med, "ms_per_step": med / steps * 1e3, "timed_reps": len(reps),
            "ms_per_step_min": srt[0] / steps * 1e3, "ms_per_step_max": srt[-1] / steps * 1e3}


def traffic_table():
    p = os.path.join(ROOT, TRAFFIC_FILE)
    if not os.path.exists(p):
        return {}, None
    try:
        import subprocess
        commit = subprocess.run(["git", "log", "-1", "--format=%h", "--", TRAFFIC_FILE], cwd=ROOT, capture_output=True, text=True).stdout.strip() or None
    except Exception:                                               # noqa: BLE001 - no git on the box
        commit = None
    return json.load(open(p)), commit


def roofline_from_timer(timer, n_steps, dtype, step_ms, step_flops, step_bytes, linear_dtype=None, use_traffic=False):
    """Families by total time per step; the dominant one is the roofline kernel.  linear_dtype: arithmetic of the linear family when it
    differs from the convs' (fp32 MFMA linears next to bf16 convs price against the fp32 peak)."""
    summ = timer.summary() if (timer is not None and n_steps > 0) else {}
    if not summ:                                             # --roofline-steps 0: A/B runs that only want the timing
        return None, [], {}
    peak = PEAK_BF16_FLOPS if dtype == "bf16" else PEAK_F32_FLOPS
    fams = {}
    for k, (n, ms) in summ.items():
        f = fams.setdefault(family(k), dict(ms=0.0, flops=0.0, launches=0, fp8_flops=0.0))
        f["ms"] += n * ms / n_steps
        f["flops"] += n * conv_flops(k) / n_steps
        f["launches"] += n / n_steps
        if " fp8 " in k:                                     # ops.conv_fp8's launches: the block-scaled fp8 MFMA
            f["fp8_flops"] += n * conv_flops(k) / n_steps
    dom = max(fams, key=lambda f: fams[f]["ms"])
    d = fams[dom]
    if dom.startswith("linear") and linear_dtype is not None:
        peak = PEAK_BF16_FLOPS if linear_dtype == "bf16" else PEAK_F32_FLOPS
    if d["fp8_flops"] > 0:
        # a family with fp8 launches is priced at the peak its FLOPs would need: fp8 FLOPs at 5 PFLOP/s, the others at the dtype's peak (harmonic blend)
        peak = d["flops"] / (d["fp8_flops"] / PEAK_FP8_FLOPS + (d["flops"] - d["fp8_flops"]) / peak)
    # the committed counter passes describe ONE workload (128^3, B = 4, bf16 train step): other workloads report traffic = null
    traffic, commit = traffic_table() if use_traffic else ({}, None)
    commit = traffic.get("_commit", commit)
    tr = traffic.get(dom, {})
    ach = d["flops"] / (d["ms"] * 1e-3)
    roof = {"bound": "mfma", "kernel": dom, "achieved": ach / 1e12, "peak": peak / 1e12, "unit": "TFLOP/s", "frac": ach / peak,
            "traffic": tr.get("hbm_bytes_per_step"), "traffic_algorithmic": tr.get("algorithmic_bytes_per_step"),
            "traffic_source": (f"{TRAFFIC_FILE} @ {commit}" if tr else None),
            "avg_ms": d["ms"] / d["launches"], "ms_per_step": d["ms"], "launches_per_step": d["launches"],
            "algorithmic_gflop_per_step": d["flops"] / 1e9, "fp8_gflop_per_step": d["fp8_flops"] / 1e9,
            "timing": f"HIP events around each launch of the family over {n_steps} eager steps run right after the timed region",
            "step": {"algorithmic_gflop": step_flops / 1e9, "algorithmic_mb": step_bytes / 1e6, "ms": step_ms,
                     "mfma_frac": step_flops / (step_ms * 1e-3) / peak, "hbm_frac": step_bytes / (step_ms * 1e-3) / PEAK_HBM}}
    per = sorted(summ.items(), key=lambda kv: -kv[1][0] * kv[1][1])
    top = []
    for k, (n, ms) in per[:5]:
        e = {"launch": k, "per_step": n / n_steps, "avg_ms": round(ms, 4), "tflops": round(conv_flops(k) / (ms * 1e-3) / 1e12, 1)}
        t_ = traffic.get(k)
        if t_:
            e["hbm_mb_counter"] = round(t_.get("hbm_bytes_per_launch", 0) / 1e6, 1)
            e["hbm_mb_algorithmic"] = round(t_.get("algorithmic_bytes_per_launch", 0) / 1e6, 1)
        top.append(e)
    roof["top"] = top
    kernels = {k: {"per_step": n / n_steps, "avg_ms": round(ms, 4), "tflops": round(conv_flops(k) / (ms * 1e-3) / 1e12, 1)} for k, (n, ms) in per}
    fam_out = {f: {"ms_per_step": round(v["ms"], 4), "launches_per_step": v["launches"], "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1)}
               for f, v in sorted(fams.items(), key=lambda kv: -kv[1]["ms"])}
    return roof, kernels, fam_out


# ------------------------------------------------------------------------------------------------------------------ volume workloads
def run_volume(args, rank, world, dev):
    from causal_vae_amd import FusedAdam, _lib
    from causal_vae_amd import ops as _ops
    from causal_vae_amd.causal_cascade import CausalBioVAE3D, loss_function, train_step
    from causal_vae_amd.parallel import GradAllReducer, broadcast_parameters
    vessel = args.workload == "vol128-vessel"
    fp8 = args.dtype == "fp8"                                        # configs[4]: the bf16 step with its C_in >= 32 forward convs on fp8 operands
    dtype = torch.bfloat16 if args.dtype in ("bf16", "fp8") else torch.float32
    torch.manual_seed(42)                                            # causal_cascade/main.py:28
    model = CausalBioVAE3D().to(dev).train().set_compute_dtype(dtype)
    if fp8:
        model.set_fp8_forward(True)
    broadcast_parameters(model)
    opt = FusedAdam(model.parameters(), lr=args.lr, device_step=True)   # main.py:50
    if world == 1 and args.overlap_adam:
        opt.overlap_backward(model.early_gradient_parameters())        # measured: -4 % (the streaming update slows the co-running conv kernels more than it hides)
    comm = None
    if world > 1 and args.exchange == "abi":                        # the exchange through include/cvae_dp.h (reduce-scatter + all-gather on RCCL directly)
        from causal_vae_amd.parallel import RcclComm
        comm = RcclComm()
    reducer = GradAllReducer(model.parameters(), comm=comm) if world > 1 else None
    x, m, t, eps = make_batch(args.batch, args.size, 1234 + rank, dev, binary=vessel)

    if vessel:
        from causal_vae_amd.optim import clip_grad_norm_
        from causal_vae_amd.vessel import loss_function as vessel_loss, total_loss

        zero_lv = torch.zeros(args.batch, 12, device=dev)    # unit variance (a constant of the workload, not a per-step fill)

        def vessel_step_fn():
            """The vessel recipe's step on volumes (vessel_analysis/01_train/train.py:18-60, 70-86): pos-weighted MSE-sum + 0.3 * background L1 +
            0.5 * KLD + Gaussian NLL of m (unit variance: the lift has no m_logvar head), clip_grad_norm_(5.0), Adam."""
            opt.zero_grad(set_to_none=True)
            recon_x, m_hat, mu, logvar = model(x, m, t)
            with _ops.zero_pool(8, x):                       # as vessel/train.py:train_step: the loss scalars' zero fills pooled
                recon, kld, morph, sparsity = vessel_loss(recon_x, x, m_hat, m, mu, logvar, m_hat, zero_lv)
                loss = total_loss(recon, kld, morph, sparsity, beta=0.5)
            _ops.backward_from(loss)
            if reducer is not None:
                reducer()
            _, coef = clip_grad_norm_(model.parameters(), 5.0, scale_grads=False)     # the clip coefficient rides into Adam (vessel/train.py:train_step)
            opt.step(grad_scale=coef)
            return loss.detach(), recon.detach(), morph.detach()
        eager_step = vessel_step_fn
    else:
        def eager_step():
            return train_step(model, opt, x, m, t, grad_hook=reducer)

    # ELBO of the first batch at the initial weights (compared with the oracle's below, rank 0 / N = 1)
    with torch.no_grad():
        out = model(x, m, t, eps=eps)
        elbo0 = float(loss_function(out[0], x, out[1], m, out[2], out[3])[0])
    del out

    use_graph = not args.no_graph
    n_pre, want_split, capture_fallback = 0, False, None
    if use_graph and vessel:
        from causal_vae_amd.graph import GraphedCallable
        if world > 1:
            raise SystemExit("vol128-vessel is a single-GPU secondary workload")
        gc = GraphedCallable(eager_step, warmup=3)
        n_pre = 3
        step = lambda: gc()
    elif use_graph:
        from causal_vae_amd.graph import GraphedTrainStep
        # N > 1: the backward is split at the encoder output so the decoder + bottleneck gradients are exchanged under the encoder's backward
        want_split = (world > 1 and not args.no_overlap_exchange) or args.force_overlap_exchange
        try:
            gstep = GraphedTrainStep(model, opt, (x, m, t), None, reducer=reducer, warmup=3, overlap_exchange=want_split)   # None: model.forward_elbo, as train_step
            n_pre = 3                                                # the capture warm-up runs 3 real steps
        except Exception as e:                                       # noqa: BLE001 - the plain two-graph exchange is always available
            if not want_split:
                raise
            capture_fallback = repr(e)
            _ops.reset_pending_wgrads()                              # a backward that raised never ran its end-of-backward flush: drop its queue
            print(f"[bench] split-backward capture failed ({e!r}); falling back to one exchange after the backward", file=sys.stderr)
            want_split = False
            gstep = GraphedTrainStep(model, opt, (x, m, t), None, reducer=reducer, warmup=3)
            n_pre = 6                                                # both attempts trained 3 warm-up steps each
        step = lambda: gstep()
    else:
        step = eager_step

    reps, out = timed_region(step, args.steps, max(args.warmup - n_pre, 1), world, dev, args.min_timed_s)
    final_loss = out[0].clone()
    # ---- roofline leg: the same step issued eagerly with HIP events around every conv launch (events cannot sit inside a
    # replayed graph); same process, same buffers, directly after the timed region ----
    timer = None
    if not args.no_kernel_timer and args.roofline_steps > 0:
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
    if rank != 0:
        return None
    n_params = sum(p.numel() for p in model.parameters())
    conv_params = sum(p.numel() for n_, p in model.named_parameters() if ("enc_conv" in n_ or "dec_conv" in n_) and p.dim() > 1)
    res = {"metric": METRIC if args.workload == "vol128" else f"training samples/sec, workload {args.workload}", "unit": "samples/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic"}
    res.update(timing_fields(reps, args.steps, world * args.batch))
    loss_name = ("vessel recipe: pos-weighted MSE-sum + 0.3*background-L1 + 0.5*KLD + Gaussian-NLL(m), clip 5.0" if vessel
                 else "ELBO = MSE-sum + 2000*MSE-sum(m) + KLD")
    res["config"] = {"workload": f"3D vessel CausalVAE train step, {args.size}^3 {'bf16 volumes, fp8 (e4m3) forward convs (C_in >= 32), bf16 backward' if fp8 else args.dtype + ' volumes'} ({'binary ~10 % density' if vessel else 'z-scored N(0,1)'}), "
                                 f"batch {args.batch}/GPU, Adam lr {args.lr:g}, {loss_name}",
                     "global_batch": world * args.batch, "per_gpu_batch": args.batch, "volume": [args.size] * 3, "parallelism": f"dp{world}", "params": n_params}
    res["exchange_backend"] = "cvae_dp C ABI (RCCL reduce-scatter + all-gather)" if comm is not None else ("torch.distributed all_reduce (RCCL)" if world > 1 else None)
    res["exchange"] = ("split backward: decoder + bottleneck bucket all-reduced under the encoder backward" if (use_graph and want_split) else
                       ("one all-reduce after the backward" if world > 1 else "none (1 rank)"))
    if capture_fallback:
        res["capture_fallback"] = capture_fallback
    res.update({"final_loss": fin(final_loss), "lr": args.lr, "hip_graph": use_graph, "deferred_wgrad": bool(_ops.DEFER_WGRAD)})
    step_ms = res["ms_per_step"]
    fl, byt = step_algorithmic(args.batch, args.size, 4 if args.dtype == "f32" else 2, n_params, conv_params)
    if timer is not None:
        roof, kernels, fams = roofline_from_timer(timer, args.roofline_steps, "bf16" if fp8 else args.dtype, step_ms, fl, byt,
                                                  use_traffic=(args.size == 128 and args.batch == 4 and args.dtype == "bf16"))
        if fp8 and roof is not None:
            # the forward products of the six fp8 layers run at the fp8 peak, everything else at the bf16 peak: the step's MFMA fraction is the time
            # the two parts would take at their own peaks over the measured step
            s_, f8fl = args.size, 0.0
            for i, (ci, co) in enumerate([(1, 32), (32, 64), (64, 128), (128, 256)]):
                s_ //= 2
                f8fl += 2.0 * args.batch * s_ ** 3 * ci * co * 64 if i else 0.0
            d_ = 4
            for ci, co in [(256, 128), (128, 64), (64, 32)]:
                f8fl += 2.0 * args.batch * d_ ** 3 * ci * co * 64
                d_ *= 2
            roof["step"]["fp8_gflop"] = f8fl / 1e9
            roof["step"]["mfma_frac"] = (f8fl / PEAK_FP8_FLOPS + (fl - f8fl) / PEAK_BF16_FLOPS) / (step_ms * 1e-3)
            roof["step"]["mfma_frac_rule"] = "fp8 forward FLOPs at the 5 PFLOP/s fp8 peak + the rest at the 2.5 PFLOP/s bf16 peak, over the measured step"
            roof["note_fp8"] = ("`peak` of a family with fp8 launches (labels '... fp8 ...') is the harmonic blend: its fp8 FLOPs at 5 PFLOP/s, the rest at 2.5")
        res["roofline"] = roof
        res["conv_ms_per_step"] = sum(v["ms_per_step"] for v in fams.values())
        res["families"] = fams
        res["kernels"] = kernels
    if world == 1 and args.cpu_seconds > 0 and not vessel:
        info = cpu_info()
        # torch.set_num_threads(all physical cores of the node) plus an 8-thread figure.  "All physical cores" are the distinct
        # thread-sibling sets inside this process's affinity mask — but a GPU box also carries a cgroup CPU quota (16 CPUs per GPU on this pool), and
        # 128 threads on 16 CPUs' worth of quota are throttled (measured: 1.8 s/step against 0.49 s with 16 threads).  The main leg therefore uses
        # min(physical cores, quota) threads — the cores this process can actually run on — and yields the K + 1 losses of the ELBO trajectory
        # check below; `cpu_baseline_8t` and, when the quota is smaller than the mask, `cpu_baseline_all_physical` (every physical core of the
        # mask, throttled by the quota) are reported next to it.
        phys = max(1, info["physical_cores"])
        usable = phys if not info["cgroup_cpu_quota"] else max(1, min(phys, int(info["cgroup_cpu_quota"] + 0.999)))
        threads = args.cpu_threads if args.cpu_threads > 0 else usable
        K = 5
        xc, mc, tc, ec = x.cpu(), m.cpu(), t.cpu(), eps.cpu()
        cb, ref_losses = cpu_baseline(args.batch, args.size, args.cpu_seconds * 0.5, xc, mc, tc, ec, args.lr, threads, K + 1)
        cb.update(info)
        cb["threads_rule"] = "min(physical cores in the affinity mask, cgroup CPU quota); torch.set_num_threads"
        extra = [("cpu_baseline_8t", 8, 3)] if threads != 8 else []
        if phys != threads and args.cpu_threads <= 0:
            extra.append(("cpu_baseline_all_physical", phys, 2))
        for name, nthr, min_steps in extra:
            sub, _ = cpu_baseline(args.batch, args.size, args.cpu_seconds * 0.25, xc, mc, tc, ec, args.lr, nthr, min_steps)
            sub.pop("losses", None)
            cb[name] = sub
        res["cpu_baseline"] = cb
        res["elbo_rel_err"] = abs(elbo0 - ref_losses[0]) / abs(ref_losses[0])
        res["gpu_over_cpu"] = res["value"] / cb["value"]
        # the same K + 1 steps on the GPU from the same initial weights with the same injected noise: ELBO after K Adam updates
        _ops.EpsSource._instances = 0
        torch.manual_seed(42)
        m2 = CausalBioVAE3D().to(dev).train().set_compute_dtype(dtype)
        if fp8:
            m2.set_fp8_forward(True)
        o2 = FusedAdam(m2.parameters(), lr=args.lr)
        gl = [float(train_step(m2, o2, x, m, t, eps=eps)[0]) for _ in range(K + 1)]
        res["elbo_rel_err_after_k"] = {"k": K, "value": abs(gl[K] - ref_losses[K]) / abs(ref_losses[K]), "gpu": gl[K], "oracle": ref_losses[K],
                                       "trajectory_rel_err": [abs(a - b) / abs(b) for a, b in zip(gl, ref_losses)]}
    return res


# ------------------------------------------------------------------------------------------------------------------ MNIST (configs[1])
def run_mnist(args, rank, world, dev):
    from causal_vae_amd import FusedAdam, _lib
    from causal_vae_amd.graph import GraphedCallable
    from causal_vae_amd.mnist_baseline import CausalMorphVAE12, LatentDiscriminator, train_step as mnist_step
    if world > 1:
        raise SystemExit("the MNIST workload is single-GPU (BASELINE.json configs[1])")
    B = args.batch
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    g = torch.Generator().manual_seed(1234)
    x, m = torch.rand(B, 1, 28, 28, generator=g).to(dev), torch.rand(B, 12, generator=g).to(dev)
    t = torch.nn.functional.one_hot(torch.randint(0, 10, (B,), generator=g), 10).float().to(dev)
    torch.manual_seed(42)
    vae, disc = CausalMorphVAE12().to(dev).train().set_compute_dtype(dtype), LatentDiscriminator().to(dev).train()
    lin_bf16 = args.dtype == "bf16" and not args.fp32_linears
    if lin_bf16:                                            # the config is named bf16: the large linears (1024 x 3158 x 512, ...) on bf16 MFMA operands too
        from causal_vae_amd.layers import set_linear_math
        set_linear_math(vae, torch.bfloat16); set_linear_math(disc, torch.bfloat16)
    ov, od = FusedAdam(vae.parameters(), lr=1e-3, device_step=True), FusedAdam(disc.parameters(), lr=1e-3, device_step=True)
    eager = lambda: mnist_step(vae, disc, ov, od, x, m, t)
    n_pre = 0
    if not args.no_graph:
        gs = GraphedCallable(eager, warmup=3)
        n_pre = 3
        step = lambda: (gs()["loss"],)
    else:
        step = lambda: (eager()["loss"],)
    reps, out = timed_region(step, args.steps, max(args.warmup - n_pre, 1), world, dev, args.min_timed_s)
    timer = _lib.KernelTimer()
    eager(); torch.cuda.synchronize()
    _lib.TIMER = timer
    for _ in range(args.roofline_steps):
        eager()
    torch.cuda.synchronize()
    _lib.TIMER = None
    res = {"metric": "training samples/sec, MNIST CausalMorphVAE12 adversarial step (BASELINE.json configs[1])", "unit": "samples/s", "n_gpus": 1, "steps": args.steps,
           "warmup": args.warmup, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic"}
    res.update(timing_fields(reps, args.steps, B))
    res["config"] = {"workload": f"MNIST CausalMorphVAE12 adversarial step (D step + VAE step, mnist_test/01_baseline_causal_vae/train.py:34-93), batch {B}, "
                                 f"{args.dtype} convs" + (" and large linears (bf16 MFMA operands, fp32 accumulate; small heads fp32)" if lin_bf16 else ", fp32 linears"),
                     "global_batch": B, "params": sum(p.numel() for p in vae.parameters()) + sum(p.numel() for p in disc.parameters())}
    res["final_loss"] = float(out[0])
    res["hip_graph"] = not args.no_graph
    # algorithmic FLOPs of the step: 3 VAE forwards' worth is NOT what runs — one no-grad forward (D step), one forward + backward (VAE step)
    n_params = res["config"]["params"]
    fwd = 2.0 * B * (14 * 14 * 32 * 16 + 7 * 7 * 64 * 32 * 16 + 3158 * 512 + 512 * 20 + 10 * 128 + 128 * 12 + 22 * 3136 + 7 * 7 * 64 * 32 * 16 + 14 * 14 * 32 * 16)
    step_fl = fwd * (1 + 3)
    step_by = 28 * n_params + 3 * 4 * n_params + B * 4 * (2 * 784 * 3 + 6 * (6272 + 3136 + 3136 + 6272))
    lin_dt = "bf16" if lin_bf16 else "f32"
    roof, kernels, fams = roofline_from_timer(timer, args.roofline_steps, args.dtype, res["ms_per_step"], step_fl, step_by, linear_dtype=lin_dt)
    if roof is not None:
        roof["linear_arithmetic"] = lin_dt
    res["roofline"], res["families"], res["kernels"] = roof, fams, kernels
    return res


# ------------------------------------------------------------------------------------------------------------------ decode (configs[4])
def run_decode(args, rank, world, dev):
    from causal_vae_amd import _lib
    from causal_vae_amd.causal_cascade import CausalBioVAE3D
    from causal_vae_amd.counterfactual import sweep_inputs
    fp8 = args.dtype == "fp8"
    dtype = torch.float32 if args.dtype == "f32" else torch.bfloat16
    torch.manual_seed(42)
    model = CausalBioVAE3D().to(dev).eval().set_compute_dtype(dtype)
    g = torch.Generator().manual_seed(1234 + rank)
    z, m = torch.randn(args.batch, 64, generator=g).to(dev), torch.rand(args.batch, 12, generator=g).to(dev)
    z_rep, m_cf = sweep_inputs(z, m, list(range(12)), [0.0, 0.25, 0.5, 0.75, 1.0])      # 60 decodes per sample
    rows = z_rep.shape[0]
    size = None if args.decode_native else (args.size,) * 3
    plan, fp8_err = None, None
    if fp8:                                                   # static per-tensor scales from one bf16 pass over the first sample's 60 rows
        plan = model.calibrate_fp8_decoder(z_rep[:60], m_cf[:60])
        with torch.no_grad():
            a, b = model.decode(z_rep[-60:], m_cf[-60:], None, fp8_plan=plan), model.decode(z_rep[-60:], m_cf[-60:], None)
            fp8_err = float(((a - b).norm() / b.norm()).item())
        del a, b

    def step():
        with torch.no_grad():
            return (model.decode(z_rep, m_cf, size, fp8_plan=plan),)
    reps, out = timed_region(step, args.steps, max(args.warmup, 1), world, dev, args.min_timed_s)
    shape = tuple(out[0].shape)
    del out
    timer = _lib.KernelTimer()
    step(); torch.cuda.synchronize()
    _lib.TIMER = timer
    for _ in range(args.roofline_steps):
        step()
    torch.cuda.synchronize()
    _lib.TIMER = None
    if rank != 0:
        return None
    res = {"metric": "counterfactual decodes/sec (batched M-sweep decode, BASELINE.json configs[4])", "unit": "decodes/s", "n_gpus": world, "steps": args.steps,
           "warmup": args.warmup, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic"}
    res.update(timing_fields(reps, args.steps, world * rows))
    res["config"] = {"workload": f"batched counterfactual decode: {args.batch} samples x 12 features x 5 values = {rows} stacked rows per rank -> dec_input -> dec_conv -> "
                                 f"{'native 64^3' if size is None else 'trilinear resize to %d^3' % args.size}, {args.dtype} convs "
                                 "(replaces the per-value loop of vessel_analysis/04_generate_counterfactual/generate_counterfactual.py:77-99)",
                     "rows_per_rank": rows, "output_shape": list(shape)}
    fl = 2.0 * rows * 64 * (4 ** 3 * 256 * 128 + 8 ** 3 * 128 * 64 + 16 ** 3 * 64 * 32 + 32 ** 3 * 32 * 1) + 2.0 * rows * 76 * 16384
    A = rows * (8 ** 3 * 128 + 16 ** 3 * 64 + 32 ** 3 * 32 + 64 ** 3)
    esz = {"bf16": 2, "f32": 4, "fp8": 1}[args.dtype]
    A_io = 2 * A * esz if not fp8 else rows * (16384 * 2 + 16384 + 2 * (8 ** 3 * 128 + 16 ** 3 * 64) + 32 ** 3 * 32 * (1 + 1) + 64 ** 3 * 2)   # fp8 codes between the layers, the 1-channel layer's input included
    byt = A_io + rows * 16384 * esz + (rows * args.size ** 3 * 4 if size is not None else rows * 64 ** 3 * 4)
    if fp8:
        res["fp8"] = {"format": "OCP e4m3, static per-tensor scales (calibrated on the first sample's 60 rows), fp32 accumulate; the dec_input linear stays bf16; the 1-channel output layer reads the fp8 "
                                "codes of the layer before it and multiplies them with bf16 weights", "rel_l2_vs_bf16_decode_on_last_60_rows": fp8_err}
    roof, kernels, fams = roofline_from_timer(timer, args.roofline_steps, "bf16" if fp8 else args.dtype, res["ms_per_step"], fl, byt)
    if fp8 and roof is not None:
        f8fl = 2.0 * rows * 64 * (4 ** 3 * 256 * 128 + 8 ** 3 * 128 * 64 + 16 ** 3 * 64 * 32)
        roof["step"]["fp8_gflop"] = f8fl / 1e9
        roof["step"]["mfma_frac"] = (f8fl / PEAK_FP8_FLOPS + (fl - f8fl) / PEAK_BF16_FLOPS) / (res["ms_per_step"] * 1e-3)
        roof["step"]["mfma_frac_rule"] = "fp8 layer FLOPs at the 5 PFLOP/s fp8 peak + the rest at the 2.5 PFLOP/s bf16 peak, over the measured sweep"
    res["roofline"], res["families"], res["kernels"] = roof, fams, kernels
    return res


SECONDARY = (("mnist", "mnist", None, {}), ("vol64-f32", "vol64-f32", None, {}), ("vol128-fp8", "vol128", "fp8", {}), ("decode-bf16", "decode", "bf16", {}), ("decode-fp8", "decode", "fp8", {}))


def secondary_lines(args, dev):
    """The other BASELINE.json configurations, measured in the SAME default run (so the driver's record carries them): configs[1] MNIST bf16
    batch 1024, configs[2] 64^3 fp32 batch 16, configs[4] the 240-row counterfactual decode in bf16 and fp8 (and, with the fp8 train step,
    its train leg).  Same timed-region protocol as the headline (barrier + synchronize brackets, median repetition); compact fields only —
    the full line of each is `python bench.py --workload ...`."""
    import copy
    import gc
    out = {}
    for name, workload, dtype, extra in SECONDARY:
        a = copy.copy(args)
        dsz, dB, ddt = WORKLOAD_DEFAULTS[workload]
        a.workload, a.size, a.batch, a.dtype = workload, dsz, dB, dtype or ddt
        a.cpu_seconds, a.roofline_steps, a.min_timed_s, a.steps, a.warmup = 0.0, 2, 0.2, 20, 5
        for k, v in extra.items():
            setattr(a, k, v)
        gc.collect()
        torch.cuda.empty_cache()
        try:
            fn = run_mnist if workload == "mnist" else (run_decode if workload == "decode" else run_volume)
            r = fn(a, 0, 1, dev)
            roof = r.get("roofline") or {}
            out[name] = {"metric": r["metric"], "value": r["value"], "unit": r["unit"], "ms_per_step": r["ms_per_step"], "ms_per_step_min": r["ms_per_step_min"],
                         "dtype": r["dtype"], "workload": r["config"]["workload"], "final_loss": r.get("final_loss"),
                         "roofline_step": roof.get("step"), "roofline_kernel": {k: roof.get(k) for k in ("kernel", "achieved", "peak", "unit", "frac", "avg_ms")} if roof else None}
            if "fp8" in r:
                out[name]["fp8"] = r["fp8"]
        except Exception as e:                                      # noqa: BLE001 - a secondary line must never take the headline down
            out[name] = {"error": repr(e)}
    return out


WORKLOAD_DEFAULTS = {"vol128": (128, 4, "bf16"), "vol128-vessel": (128, 4, "bf16"), "vol64-f32": (64, 16, "f32"), "mnist": (28, 1024, "bf16"), "decode": (128, 4, "bf16")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="vol128", choices=["vol128", "vol64-f32", "mnist", "decode", "vol128-vessel"])
    ap.add_argument("--size", type=int, default=None, help="volume edge (default: the workload's)")
    ap.add_argument("--batch", type=int, default=None, help="samples per GPU (default: the workload's)")
    ap.add_argument("--dtype", default=None, choices=["bf16", "f32", "fp8"], help="fp8 (e4m3 conv operands): the decode workload, or vol128 = the bf16 train step with its C_in >= 32 forward convs on fp8")
    ap.add_argument("--min-timed-s", type=float, default=0.2, help="repeat the K-step timed region until this much has been timed; the median repetition is reported")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="budget of the cpu_baseline leg: half for the main figure, a quarter each for the 8-thread and all-physical-cores ones (0 disables)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the main cpu_baseline leg (0 = min(physical cores in the affinity mask, cgroup CPU quota))")
    ap.add_argument("--lr", type=float, default=1e-4, help="Adam learning rate (the reference uses 1e-3, causal_cascade/main.py:50, at which the 3D lift diverges on step 3 in the oracle too: DESIGN.md)")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--overlap-adam", action="store_true", help="run the non-encoder part of the Adam update on a side stream under the encoder backward")
    ap.add_argument("--fork", action="store_true", help="run weight-gradient kernels on a side stream (overlap with data-gradient kernels)")
    ap.add_argument("--fork-max-positions", type=int, default=0, help="with --fork: only layers with at most this many S positions per batch fork (0 = all)")
    ap.add_argument("--defer-join", action="store_true", help="with --fork: join the side stream once before the optimizer instead of after every layer")
    ap.add_argument("--no-graph", action="store_true", help="issue the step eagerly instead of replaying the captured HIP graph")
    ap.add_argument("--fp32-linears", action="store_true", help="mnist workload: keep every Linear on the exact-fp32 MFMA (default with bf16: large ones on bf16 operands)")
    ap.add_argument("--no-splitk", action="store_true", help="conv data kernels without their split-K scratch (A/B)")
    ap.add_argument("--no-defer-wgrad", action="store_true", help="compute every conv weight gradient in its own launch (A/B of the grouped end-of-backward launch)")
    ap.add_argument("--no-overlap-exchange", action="store_true", help="N > 1: one all-reduce after the whole backward instead of the split backward")
    ap.add_argument("--force-overlap-exchange", action="store_true", help="take the split-backward capture also at N = 1 (no exchange happens)")
    ap.add_argument("--roofline-steps", type=int, default=5, help="eager steps with per-launch HIP events after the timed region")
    ap.add_argument("--exchange", default="torch", choices=["torch", "abi"], help="N > 1: gradient exchange through torch.distributed's all_reduce on RCCL (default) or through "
                    "the C ABI of include/cvae_dp.h (libcvae_dp.so: reduce-scatter + all-gather on RCCL directly)")
    ap.add_argument("--no-secondary", action="store_true", help="default vol128 run at N = 1: skip the compact lines of the other BASELINE.json configurations")
    ap.add_argument("--decode-native", action="store_true", help="decode workload: stop at the decoder's native 64^3 (no resize)")
    args = ap.parse_args()
    dsz, dB, ddt = WORKLOAD_DEFAULTS[args.workload]
    args.size = args.size or dsz
    args.batch = args.batch or dB
    args.dtype = args.dtype or ddt
    if args.dtype == "fp8" and args.workload not in ("decode", "vol128"):
        raise SystemExit("--dtype fp8: the decode sweep (--workload decode) or the 128^3 train step with fp8 forward convs (--workload vol128)")

    from causal_vae_amd import ops as _ops
    from causal_vae_amd.parallel import init_distributed
    import torch.distributed as dist
    if args.no_defer_wgrad:
        _ops.DEFER_WGRAD = False
    if args.no_splitk:
        _ops.SPLIT_K = False
    if args.fork:
        _ops.FORK_BACKWARD, _ops.FORK_MAX_POSITIONS, _ops.DEFER_JOIN = True, args.fork_max_positions, args.defer_join
    rank, world, local_rank = init_distributed()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU path")
    if os.environ.get("CVAE_DIST_BACKEND") == "gloo":            # rehearsal: all ranks on the one visible card
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if args.workload == "mnist":
        res = run_mnist(args, rank, world, dev)
    elif args.workload == "decode":
        res = run_decode(args, rank, world, dev)
    else:
        res = run_volume(args, rank, world, dev)
    if args.workload == "vol128" and world == 1 and not args.no_secondary and args.cpu_seconds > 0 and res is not None:      # the full-report run (A/B runs pass --cpu-seconds 0)
        res["secondary"] = secondary_lines(args, dev)
    if rank == 0:
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
