"""world_size-2 gloo test of the data-parallel layer (runs on CPU): the SUM all-reduce through one flat bucket gives every
rank the gradient of the global batch, parameters stay identical across ranks, and broadcast_parameters aligns them."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from causal_vae_amd.parallel import GradAllReducer, broadcast_parameters, init_distributed, all_reduce_scalars
    r, w, _ = init_distributed("gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)                                   # ranks start different on purpose
    lin = torch.nn.Sequential(torch.nn.Linear(12, 7), torch.nn.ReLU(), torch.nn.Linear(7, 3))
    broadcast_parameters(lin)
    g = torch.Generator().manual_seed(5)
    xs = torch.randn(world * 4, 12, generator=g)
    ys = torch.randn(world * 4, 3, generator=g)
    x, y = xs[rank * 4:(rank + 1) * 4], ys[rank * 4:(rank + 1) * 4]
    loss = ((lin(x) - y) ** 2).sum()                                # sum-reduced like the reference's ELBO
    loss.backward()
    GradAllReducer(lin.parameters())()
    (tot,) = all_reduce_scalars(loss)
    ret[rank] = dict(params=[p.detach().clone() for p in lin.parameters()], grads=[p.grad.clone() for p in lin.parameters()],
                     total=float(tot), xs=xs, ys=ys)
    dist.destroy_process_group()


def test_sum_allreduce_equals_global_batch_gradient():
    world, port = 2, _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    r0, r1 = ret[0], ret[1]
    for a, b in zip(r0["params"], r1["params"]):
        assert torch.equal(a, b)                                    # broadcast aligned the replicas
    for a, b in zip(r0["grads"], r1["grads"]):
        assert torch.equal(a, b)                                    # every rank holds the same reduced gradient
    lin = torch.nn.Sequential(torch.nn.Linear(12, 7), torch.nn.ReLU(), torch.nn.Linear(7, 3))
    with torch.no_grad():
        for p, q in zip(lin.parameters(), r0["params"]):
            p.copy_(q)
    loss = ((lin(r0["xs"]) - r0["ys"]) ** 2).sum()                  # single process, global batch
    loss.backward()
    for p, gsum in zip(lin.parameters(), r0["grads"]):
        torch.testing.assert_close(gsum, p.grad, rtol=1e-5, atol=1e-6)
    assert abs(r0["total"] - float(loss)) < 1e-3 * abs(float(loss))


def _worker_split(rank, world, port, ret):
    """Two buckets exchanged asynchronously, gradients bound as views of the buckets (the GraphedTrainStep(overlap_exchange=True) recipe)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from causal_vae_amd.parallel import GradAllReducer, broadcast_parameters, init_distributed
    init_distributed("gloo")
    torch.manual_seed(7)
    lin = torch.nn.Sequential(torch.nn.Linear(12, 7), torch.nn.ReLU(), torch.nn.Linear(7, 3))
    broadcast_parameters(lin)
    g = torch.Generator().manual_seed(5)
    xs, ys = torch.randn(world * 4, 12, generator=g), torch.randn(world * 4, 3, generator=g)
    loss = ((lin(xs[rank * 4:(rank + 1) * 4]) - ys[rank * 4:(rank + 1) * 4]) ** 2).sum()
    late, early = list(lin[0].parameters()), list(lin[2].parameters())          # backward reaches lin[2] first
    ge = torch.autograd.grad(loss, early + [lin[0].weight], retain_graph=True)   # stand-in for "split at an interior tensor"
    for p, gr in zip(early, ge):
        p.grad = gr
    ra, rb = GradAllReducer(early), GradAllReducer(late)
    ra.pack()
    wa = ra.all_reduce_async()
    gl = torch.autograd.grad(loss, late)
    for p, gr in zip(late, gl):
        p.grad = gr
    rb.pack()
    wb = rb.all_reduce_async()
    wa.wait(); wb.wait()
    ra.bind_views(); rb.bind_views()
    assert all(p.grad.data_ptr() >= r._flat.data_ptr() for r in (ra, rb) for p in r.params)
    ret[rank] = [p.grad.clone() for p in lin.parameters()]
    dist.destroy_process_group()


def test_two_async_buckets_bound_as_views_equal_the_single_exchange():
    world, port = 2, _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_worker_split, args=(world, port, ret), nprocs=world, join=True)
    torch.manual_seed(7)
    lin = torch.nn.Sequential(torch.nn.Linear(12, 7), torch.nn.ReLU(), torch.nn.Linear(7, 3))
    g = torch.Generator().manual_seed(5)
    xs, ys = torch.randn(world * 4, 12, generator=g), torch.randn(world * 4, 3, generator=g)
    ((lin(xs) - ys) ** 2).sum().backward()
    for a, b, p in zip(ret[0], ret[1], lin.parameters()):
        assert torch.equal(a, b)
        torch.testing.assert_close(a, p.grad, rtol=1e-5, atol=1e-6)


def test_oracle_two_microbatches_sum_equals_dp_definition():
    """The DP parity definition (parallel.py docstring) stated on the oracle.  (1) The losses are SUM-reduced, so with BatchNorm1d out of
    the picture (eval mode) the sum of per-micro-batch gradients IS the global-batch gradient, for every parameter (fp32 summation order
    only).  (2) In train mode the two differ only through mechanism_net's batch statistics: parameters upstream of m_hat's consumer agree
    closely, and the per-rank definition changes the reconstruction term by well under a percent."""
    from oracle import functional as ofn
    from oracle.steps import _leaves, trainable_keys
    sd = oracle.init_state_dict("bio2d", seed=42)
    g = torch.Generator().manual_seed(3)
    x, m, t, eps = torch.randn(4, 1, 64, 64, generator=g), torch.rand(4, 12, generator=g), torch.randint(0, 19, (4,), generator=g), torch.randn(4, 64, generator=g)

    def eval_grads(sl):
        leaves = _leaves(sd)
        out = ofn.bio_vae_forward(leaves, x[sl], m[sl], t[sl], eps[sl], nd=2, training=False)
        loss = ofn.cascade_loss(out["recon_x"], x[sl], out["m_hat"], m[sl], out["mu"], out["logvar"], 2000.0)[0]
        keys = trainable_keys(sd)
        return dict(zip(keys, torch.autograd.grad(loss, [leaves[k] for k in keys], allow_unused=True))), float(loss)
    (ga, la), (gb, lb), (gf, lf) = eval_grads(slice(0, 2)), eval_grads(slice(2, 4)), eval_grads(slice(0, 4))
    assert abs(la + lb - lf) <= 1e-5 * abs(lf)
    for k, gfull in gf.items():
        if gfull is None:                                           # BatchNorm affine parameters see no gradient path difference; unused ones stay None
            assert ga[k] is None and gb[k] is None
            continue
        torch.testing.assert_close(ga[k] + gb[k], gfull, rtol=2e-4, atol=2e-5 * float(gfull.abs().max()), msg=lambda s_, k=k: f"{k}: {s_}")
    parts = [oracle.cascade_train_step({k: v.clone() for k, v in sd.items()}, x[i:i + 2], m[i:i + 2], t[i:i + 2], eps[i:i + 2], apply_update=False) for i in (0, 2)]
    full = oracle.cascade_train_step({k: v.clone() for k, v in sd.items()}, x, m, t, eps, apply_update=False)
    for k in ("enc_conv.0.weight", "fc_mu.weight", "dec_conv.6.weight"):          # train mode: per-rank BN statistics vs global ones
        gsum = parts[0]["grads"][k] + parts[1]["grads"][k]
        cos = float(torch.dot(gsum.flatten(), full["grads"][k].flatten()) / (gsum.norm() * full["grads"][k].norm()))
        assert cos > 0.9, (k, cos)
    assert abs(float(parts[0]["recon"] + parts[1]["recon"]) - float(full["recon"])) / float(full["recon"]) < 0.01


def _worker_misc(rank, world, port, ret):
    """sync_buffers (BatchNorm running statistics before a checkpoint) and the per-rank Philox subsequence."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from causal_vae_amd import ops
    from causal_vae_amd.parallel import init_distributed, sync_buffers
    sub_env = ops.EpsSource().subsequence()                         # before the group exists: the torchrun RANK
    init_distributed("gloo")
    bn = torch.nn.BatchNorm1d(5)
    with torch.no_grad():
        bn.running_mean.fill_(float(rank + 1)); bn.running_var.fill_(float(10 * (rank + 1))); bn.num_batches_tracked.fill_(3 + rank)
    sync_buffers(bn, mode="average")
    avg = (bn.running_mean.clone(), bn.running_var.clone(), int(bn.num_batches_tracked))
    with torch.no_grad():
        bn.running_mean.fill_(float(rank + 1))
    sync_buffers(bn, mode="broadcast")
    src = ops.EpsSource()
    ret[rank] = dict(avg=avg, bcast=bn.running_mean.clone(), sub=src.subsequence(), sub_env=sub_env, inst=src.instance)
    dist.destroy_process_group()


def test_sync_buffers_and_per_rank_noise_streams():
    world, port = 2, _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_worker_misc, args=(world, port, ret), nprocs=world, join=True)
    for r in (0, 1):
        assert torch.equal(ret[r]["avg"][0], torch.full((5,), 1.5)) and torch.equal(ret[r]["avg"][1], torch.full((5,), 15.0)) and ret[r]["avg"][2] == 4
        assert torch.equal(ret[r]["bcast"], torch.full((5,), 1.0))
    # every rank calls torch.manual_seed(42) (bench.py, reference main.py:28): the reparameterisation noise must still differ per rank
    assert ret[0]["sub"] != ret[1]["sub"] and ret[0]["sub_env"] != ret[1]["sub_env"]
    assert ret[0]["sub"] >> 32 == 0 and ret[1]["sub"] >> 32 == 1 and ret[0]["inst"] == ret[1]["inst"] == 1     # second EpsSource of each process
