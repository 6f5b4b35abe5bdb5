#!/usr/bin/env python3
"""Multi-process checks of the data-parallel path on the GPU box (launched by tests/test_parallel_gpu.py, never collected by pytest).

    python -m torch.distributed.run --nproc-per-node 2 ... tests/dist_worker.py <mode>        (CVAE_DIST_BACKEND=gloo: both ranks on one card)
    RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=... python tests/dist_worker.py nccl1    (one-rank RCCL group)

Every mode ends with rank 0 printing "DIST_WORKER_OK <mode>"; any assertion failure exits non-zero.
The checker is the CPU oracle (oracle/): N ranks with summed gradients must equal the oracle's sum of per-micro-batch gradients
(per-rank BatchNorm, the default) or its single global-batch step (sync_bn / sync_pos_weight).
"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

NOISE_KEY = "mechanism_net.0.bias"


def rel(a, b):
    return abs(float(a) - float(b)) / max(abs(float(b)), 1e-30)


def grad_close(got, ref, what, l2=2e-3, linf=2e-2):
    got, ref = got.detach().cpu().double(), ref.double()
    nrm = float(ref.norm())
    assert float((got - ref).norm()) <= l2 * nrm + 1e-12, (what, "rel L2", float((got - ref).norm()) / max(nrm, 1e-30))
    assert float((got - ref).abs().max()) <= linf * float(ref.abs().max()) + 1e-12, (what, "max err")


def adam_close(p, ref, what, lr=1e-3):
    d = (p.detach().cpu().float() - ref.float()).abs()
    assert float(d.max()) <= 2.0 * lr + 1e-6, (what, float(d.max()))
    frac = float((d > 0.21 * lr).float().mean())
    assert frac < max(2e-3, 2.5 / d.numel()), (what, "fraction of weights off by more than 0.21*lr", frac)


def global_batch(world, per_rank, size, seed=77):
    g = torch.Generator().manual_seed(seed)
    n = world * per_rank
    # distinct treatments inside every micro-batch: two equal one-hots at B = 2 give train-mode BatchNorm1d a variance of exactly 0 (rstd = 316,
    # ReLU at exactly 0), which turns fp32 rounding into O(1) gradient differences between any two implementations
    t = (torch.arange(n) * 5 + 3) % 19
    return (torch.randn(n, 1, size, size, size, generator=g), torch.rand(n, 12, generator=g), t, torch.randn(n, 64, generator=g))


def mode_dp_step(rank, world, dev):
    """Summed gradients over ranks == the oracle's sum of per-micro-batch gradients (per-rank BatchNorm), then one Adam step; and the three
    ways of issuing the step (eager + hook, two-graph exchange, split-backward exchange) are bit-identical on Philox noise."""
    import oracle
    from causal_vae_amd import FusedAdam, ops
    from causal_vae_amd.causal_cascade import CausalBioVAE3D, train_step
    from causal_vae_amd.graph import GraphedTrainStep
    from causal_vae_amd.parallel import GradAllReducer, broadcast_parameters
    B, S = 2, 64                                                 # 64^3: the fused bottleneck (and with it the split backward) needs a >= 4^3 encoder output
    x, m, t, eps = global_batch(world, B, S)
    sl = slice(rank * B, (rank + 1) * B)
    xd, md, td, ed = (v[sl].to(dev) for v in (x, m, t, eps))
    # ---- (a) injected eps vs the oracle ----
    torch.manual_seed(42)
    model = CausalBioVAE3D().to(dev).train()
    broadcast_parameters(model)
    opt = FusedAdam(model.parameters(), lr=1e-3)
    red = GradAllReducer(model.parameters())
    loss, _, _ = train_step(model, opt, xd, md, td, eps=ed, grad_hook=red)
    tot = loss.clone()
    dist.all_reduce(tot)
    sd = oracle.init_state_dict("bio3d", seed=42)
    parts = [oracle.cascade_train_step({k: v.clone() for k, v in sd.items()}, x[i * B:(i + 1) * B], m[i * B:(i + 1) * B], t[i * B:(i + 1) * B],
                                       eps[i * B:(i + 1) * B], nd=3, apply_update=False) for i in range(world)]
    assert rel(tot, sum(float(p["loss"]) for p in parts)) < 1e-4
    gsum = {k: sum(p["grads"][k] for p in parts) for k in parts[0]["grads"]}
    for k, p in model.named_parameters():
        if k != NOISE_KEY:
            grad_close(p.grad, gsum[k], k)
    st = oracle.adam_init(sd)
    oracle.adam_update(sd, gsum, st, lr=1e-3)
    for k, p in model.named_parameters():
        if k != NOISE_KEY:
            adam_close(p, sd[k], k)
    # ---- (b) the three issue modes on Philox noise: 6 steps each, bit-identical losses and weights; ranks draw different noise ----
    runs = []
    for mode in ("eager", "graph", "split"):
        ops.EpsSource._instances = 0
        torch.manual_seed(42)
        mdl = CausalBioVAE3D().to(dev).train()
        broadcast_parameters(mdl)
        o = FusedAdam(mdl.parameters(), lr=1e-4, device_step=True)
        r = GradAllReducer(mdl.parameters())
        if mode == "eager":
            losses = [float(train_step(mdl, o, xd, md, td, grad_hook=r)[0]) for _ in range(6)][3:]
        else:
            gs = GraphedTrainStep(mdl, o, (xd, md, td), None, reducer=r, warmup=3, overlap_exchange=(mode == "split"))
            losses = [float(gs()[0]) for _ in range(3)]
        torch.cuda.synchronize()
        runs.append((losses, [p.detach().clone() for p in mdl.parameters()]))
    for losses, params in runs[1:]:
        assert losses == runs[0][0], (runs[0][0], losses)
        for (k, _), p, q in zip(mdl.named_parameters(), runs[0][1], params):
            assert torch.equal(p, q), k
    mine = torch.tensor(runs[0][0], device=dev)
    both = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(both, mine)
    assert not torch.equal(both[0], both[1])                     # different samples and different Philox subsequences per rank
    for p in runs[0][1]:                                         # but the replicas stay identical: same summed gradients everywhere
        q = p.clone()
        dist.broadcast(q, src=0)
        assert torch.equal(p, q)


def mode_sync_bn(rank, world, dev):
    """convert_sync_batchnorm: the N-rank step equals the oracle's single-process step on the GLOBAL batch (BatchNorm1d statistics over all
    N * B samples): loss, every gradient, the Adam step, the running statistics."""
    import oracle
    from causal_vae_amd import FusedAdam
    from causal_vae_amd.causal_cascade import CausalBioVAE3D, train_step
    from causal_vae_amd.parallel import GradAllReducer, broadcast_parameters, convert_sync_batchnorm
    for S, fused in ((32, False), (64, True)):                   # 32^3: layer-by-layer path (ops.SyncBatchNorm1dTrain); 64^3: inside the fused bottleneck (cvae_bottleneck_*_sync)
        B = 2
        x, m, t, eps = global_batch(world, B, S, seed=78)
        sl = slice(rank * B, (rank + 1) * B)
        torch.manual_seed(42)
        model = convert_sync_batchnorm(CausalBioVAE3D().to(dev).train())
        assert model.fuse_bottleneck is True
        broadcast_parameters(model)
        opt = FusedAdam(model.parameters(), lr=1e-3)
        loss, _, _ = train_step(model, opt, x[sl].to(dev), m[sl].to(dev), t[sl].to(dev), eps=eps[sl].to(dev), grad_hook=GradAllReducer(model.parameters()))
        assert (model._enc_out is not None) == fused, (S, fused)
        tot = loss.clone()
        dist.all_reduce(tot)
        sd = oracle.init_state_dict("bio3d", seed=42)
        ref = oracle.cascade_train_step(sd, x, m, t, eps, nd=3, lr=1e-3)
        assert rel(tot, ref["loss"]) < 1e-4, (float(tot), float(ref["loss"]))
        for k, p in model.named_parameters():
            if k != NOISE_KEY:
                grad_close(p.grad, ref["grads"][k], k)
                adam_close(p, sd[k], k)
        for k in ("mechanism_net.1.running_mean", "mechanism_net.1.running_var"):
            torch.testing.assert_close(model.state_dict()[k].cpu(), sd[k], rtol=1e-5, atol=1e-6)
        # the replicas hold the same statistics bit for bit (every rank combines the gathered statistics in rank order)
        for k in ("mechanism_net.1.running_mean", "mechanism_net.1.running_var"):
            mine = model.state_dict()[k].clone()
            other = mine.clone()
            dist.broadcast(other, src=0)
            assert torch.equal(mine, other), k


def mode_pos_weight(rank, world, dev):
    """vessel loss with sync_pos_weight: recon / sparsity summed over ranks and the gradient of this rank's slice equal the oracle's loss on
    the global batch (pos_weight is a batch-global scalar, vessel_analysis/01_train/train.py:30-36); without it they differ."""
    import oracle
    from causal_vae_amd import ops
    g = torch.Generator().manual_seed(5)
    n = 2 * world
    x = (torch.rand(n, 1, 24, 40, 40, generator=g) < torch.tensor([0.03, 0.2, 0.1, 0.4][:n]).view(n, 1, 1, 1, 1)).float()     # very uneven densities per sample
    r = torch.rand(n, 1, 24, 40, 40, generator=g)
    sl = slice(2 * rank, 2 * rank + 2)
    rr = r.clone().requires_grad_(True)
    z = torch.zeros(n, 3)
    ref_recon, _, _, ref_sp = oracle.vessel_loss(rr, x, z, z, z, z, z, z)
    (ref_recon + 0.3 * ref_sp).backward()
    out = {}
    for sync in (True, False):
        rl = r[sl].to(dev).requires_grad_(True)
        recon, sp = ops.VesselRecon.apply(rl, x[sl].to(dev), sync, None)
        (recon + 0.3 * sp).backward()
        tot = torch.stack([recon.detach(), sp.detach()])
        dist.all_reduce(tot)
        out[sync] = (tot.cpu(), rl.grad.cpu())
    assert rel(out[True][0][0], ref_recon) < 1e-5 and rel(out[True][0][1], ref_sp) < 1e-5
    torch.testing.assert_close(out[True][1], rr.grad[sl], rtol=1e-5, atol=1e-6)
    # per-rank pos_weight weights every positive voxel differently (the SUM barely moves: pos_weight * positives ~ N (1 - pos_frac) by construction)
    d = (out[False][1] - rr.grad[sl]).norm() / rr.grad[sl].norm()
    assert float(d) > 0.05, float(d)


def mode_eps(rank, world, dev):
    """Ranks that share torch.manual_seed(42) still draw different reparameterisation noise (Philox subsequence = rank)."""
    from causal_vae_amd.causal_cascade import CausalBioVAE3D
    torch.manual_seed(42)
    model = CausalBioVAE3D().to(dev)
    e = model._eps.draw(torch.empty(4, 64, device=dev))
    both = [torch.zeros_like(e) for _ in range(world)]
    dist.all_gather(both, e)
    assert not torch.equal(both[0], both[1]) and abs(float(both[0].mean())) < 0.3 and abs(float(both[1].std()) - 1) < 0.2
    corr = float((both[0] * both[1]).mean())
    assert abs(corr) < 0.2, corr


def mode_nccl1(rank, world, dev):
    """One-rank RCCL group: the split-backward capture with its asynchronous bucket exchange (always_exchange: the collectives really run)
    replays the same training as the eager step with the same hook — RCCL + HIP-graph replay + private capture stream on one card."""
    from causal_vae_amd import FusedAdam, ops
    from causal_vae_amd.causal_cascade import CausalBioVAE3D, train_step
    from causal_vae_amd.graph import GraphedTrainStep
    from causal_vae_amd.parallel import GradAllReducer
    assert dist.get_backend() == "nccl" and world == 1
    g = torch.Generator().manual_seed(3)
    x, m, t = torch.randn(2, 1, 64, 64, 64, generator=g).to(dev), torch.rand(2, 12, generator=g).to(dev), torch.randint(0, 19, (2,), generator=g).to(dev)
    runs = []
    for mode in ("eager", "graph", "split"):
        ops.EpsSource._instances = 0
        torch.manual_seed(42)
        model = CausalBioVAE3D().to(dev).train().set_compute_dtype(torch.bfloat16)
        opt = FusedAdam(model.parameters(), lr=1e-4, device_step=True)
        red = GradAllReducer(model.parameters(), always_exchange=True, comm=COMM[0])
        assert red.active()
        if mode == "eager":
            losses = [float(train_step(model, opt, x, m, t, grad_hook=red)[0]) for _ in range(6)][3:]
        else:
            gs = GraphedTrainStep(model, opt, (x, m, t), None, reducer=red, warmup=3, overlap_exchange=(mode == "split"))
            assert (gs.g1b is not None) == (mode == "split") and gs.g2 is not None
            losses = [float(gs()[0]) for _ in range(3)]
        torch.cuda.synchronize()
        runs.append((losses, [p.detach().clone() for p in model.parameters()]))
    for losses, params in runs[1:]:
        assert losses == runs[0][0], (runs[0][0], losses)
        for p, q in zip(runs[0][1], params):
            assert torch.equal(p, q)
    assert len(set(runs[0][0])) == 3


COMM = [None]


def mode_dp_abi1(rank, world, dev):
    """The same three-way check with the exchange behind the C ABI of include/cvae_dp.h (libcvae_dp.so: cvae_dp_init on a fresh unique id, the bucket through
    cvae_dp_allreduce_sum, the asynchronous exchange on the communicator's own stream) instead of torch.distributed — a one-rank communicator, so the sums
    are the identity; what runs is the bootstrap, the handle's life cycle and the stream ordering around graph replays."""
    from causal_vae_amd.parallel import RcclComm
    COMM[0] = RcclComm()
    assert COMM[0].world == 1 and COMM[0].rank == 0
    flat = torch.arange(1000, dtype=torch.float32, device=dev)
    COMM[0].all_reduce_sum(flat)
    COMM[0].async_on_side_stream(flat).wait()
    assert torch.equal(flat.cpu(), torch.arange(1000, dtype=torch.float32))
    # the two halves called directly DO go through RCCL on one rank (ncclReduceScatter / ncclAllGather in place): symbols, datatypes, the stream argument
    for dt in (torch.float32, torch.bfloat16):
        buf = torch.arange(4096, device=dev).to(dt)
        COMM[0].reduce_scatter_sum(buf)
        COMM[0].all_gather(buf)
        torch.cuda.synchronize()
        assert torch.equal(buf.cpu(), torch.arange(4096).to(dt))
    try:
        mode_nccl1(rank, world, dev)
    finally:
        COMM[0].close()
        COMM[0] = None


MODES = dict(dp_step=mode_dp_step, sync_bn=mode_sync_bn, pos_weight=mode_pos_weight, eps=mode_eps, nccl1=mode_nccl1, dp_abi1=mode_dp_abi1)


def main():
    mode = sys.argv[1]
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    backend = "nccl" if mode in ("nccl1", "dp_abi1") else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(0)
    dist.init_process_group(backend=backend, rank=rank, world_size=world)      # before any other GPU work of this process
    dev = torch.device("cuda", 0)                                               # every rank on the one visible card
    torch.cuda.set_device(dev)
    MODES[mode](rank, world, dev)
    dist.barrier()
    if rank == 0:
        print("DIST_WORKER_OK", mode, flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
