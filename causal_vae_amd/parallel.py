"""Data parallelism for the train step: one process per GPU, ONE sum all-reduce of the gradients per step over RCCL/xGMI.

The reference is single-process (SURVEY.md §2 "Parallelism"); this layer is the addition named by the north star.  The
losses are SUM-reduced over the batch (causal_cascade/train.py:7,10,13), so the gradient of the global batch is the SUM
(not the mean) of the per-rank gradients — `GradAllReducer` therefore reduces with op=SUM and never divides.

Per-rank statistics: BatchNorm1d in `mechanism_net` normalises with the statistics of the rank-local batch (the parity
definition used by tests/test_parallel_gloo.py: an N-rank step equals a single-process step over the same N micro-batches
with BN applied per micro-batch and gradients summed).

The bucket is one flat fp32 buffer (61.4 MB for the 3D model): a single large message keeps the xGMI links at their
bandwidth-bound rate instead of paying per-tensor latency 24 times.
"""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Initialise torch.distributed from the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).
    backend defaults to 'nccl' (= RCCL on ROCm) when a GPU is present, else 'gloo'.  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            # CVAE_DIST_BACKEND=gloo lets several ranks share ONE card (rehearsal of the N > 1 path on a 1-GPU box)
            backend = os.environ.get("CVAE_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


class GradAllReducer:
    """Sums `.grad` of `params` across ranks through one flat bucket.  Call it between backward and optimizer.step
    (the `grad_hook` of causal_cascade.train.train_step)."""

    def __init__(self, params, group=None, always_exchange=False):
        """always_exchange: issue the collective even in a one-rank group (a one-rank RCCL all-reduce is the identity; used to run the
        capture / async-exchange machinery against RCCL on a single card)."""
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.always_exchange = bool(always_exchange)
        self._flat = None

    def world_size(self):
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    def active(self):
        """True when a step has to exchange gradients: more than one rank, or always_exchange inside an initialised group."""
        return self.world_size() > 1 or (self.always_exchange and dist.is_initialized())

    def _grads(self):
        return [p.grad for p in self.params if p.grad is not None]

    def _copy(self, to_flat):
        """One multi-tensor launch (cvae_multi_copy) on GPU tensors; plain torch copies on CPU tensors (gloo tests)."""
        grads = self._grads()
        n = sum(g.numel() for g in grads)
        if self._flat is None or self._flat.numel() != n or (grads and self._flat.device != grads[0].device):
            self._flat = torch.empty(n, dtype=torch.float32, device=grads[0].device)
        offs, off = [], 0
        for g in grads:
            offs.append(off)
            off += g.numel()
        if grads and grads[0].is_cuda and all(g.is_contiguous() and g.dtype == torch.float32 for g in grads):
            import ctypes as C
            from ._lib import lib, check, stream
            k = len(grads)
            gp = [g.data_ptr() for g in grads]
            fp = [self._flat.data_ptr() + 4 * o for o in offs]
            src, dst = (gp, fp) if to_flat else (fp, gp)
            check(lib.cvae_multi_copy((C.c_void_p * k)(*src), (C.c_void_p * k)(*dst), (C.c_int64 * k)(*[g.numel() for g in grads]), k, stream()),
                  "multi_copy")
            return
        for g, o in zip(grads, offs):
            v = self._flat[o:o + g.numel()]
            if to_flat:
                v.copy_(g.reshape(-1))
            else:
                g.copy_(v.view_as(g))

    def pack(self):
        """Copy every gradient into the flat fp32 bucket (capturable: fixed addresses once the bucket exists)."""
        if self.params and self.params[0].is_cuda:
            from . import ops
            ops.join_side_streams()                          # weight gradients still running on a side stream (ops.DEFER_JOIN)
        self._copy(True)

    def all_reduce(self):
        dist.all_reduce(self._flat, op=dist.ReduceOp.SUM, group=self.group)

    def all_reduce_async(self):
        """Start the exchange of the packed bucket and return its Work handle (None without a process group): on RCCL it runs on the
        communicator's stream, ordered after what the current stream holds so far, and `.wait()` orders the current stream after it."""
        if not self.active() or self._flat is None:
            return None
        return dist.all_reduce(self._flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def unpack(self):
        self._copy(False)

    def bind_views(self):
        """Instead of copying the reduced bucket back: make every `.grad` a view of its slice of the bucket (no launch; the optimizer
        then reads the bucket directly).  The views stay valid until the next zero_grad(set_to_none=True)."""
        off = 0
        for p in self.params:
            if p.grad is None:
                continue
            n = p.grad.numel()
            p.grad = self._flat[off:off + n].view_as(p.grad)
            off += n

    def __call__(self):
        if not self.active() or not self._grads():
            return
        self.pack()
        self.all_reduce()
        self.unpack()


def broadcast_parameters(module, src=0, group=None):
    """Make every rank start from rank `src`'s parameters and buffers."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)


def all_reduce_scalars(*scalars, group=None):
    """Sum 0-dim loss tensors over ranks for logging (one tiny message)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return scalars
    buf = torch.stack([s.detach().float() for s in scalars])
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return tuple(buf.unbind(0))


def convert_sync_batchnorm(module, group=None):
    """Make every train-mode BatchNorm1d of `module` normalise with the statistics of the GLOBAL batch (ops.SyncBatchNorm1dTrain: 2 x F-float
    all-reduces forward, one backward) instead of the rank-local one — the exact-equivalence option of SURVEY.md §8(e): with it (and
    summed gradients) an N-rank step of B samples each equals the reference's single-process step on N * B samples.  The fused bottleneck
    (ops.BioBottleneck) computes mechanism_net's BatchNorm inside its own launches, so models that have it fall back to the layer-by-layer
    path (`fuse_bottleneck = False`: ~0.25 ms per step at 128^3).  Returns the module."""
    from .layers import BatchNorm1d
    found = False
    for m in module.modules():
        if isinstance(m, BatchNorm1d):
            m.sync, m.sync_group, found = True, group, True
    if found and hasattr(module, "fuse_bottleneck"):
        module.fuse_bottleneck = False
    return module


def sync_buffers(module, group=None, mode="average"):
    """BatchNorm running statistics are updated from rank-local batches and drift apart; call this before saving a checkpoint (or once
    per epoch) so that every rank — and the file rank 0 writes — holds the same buffers.  mode 'average': mean over ranks of the floating
    buffers (integer ones, num_batches_tracked, take the maximum); 'broadcast': the values of the group's first rank."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    world = dist.get_world_size(group)
    for b in module.buffers():
        if mode == "broadcast":
            dist.broadcast(b.data, src=(dist.get_global_rank(group, 0) if group is not None else 0), group=group)
        elif b.is_floating_point():
            dist.all_reduce(b.data, op=dist.ReduceOp.SUM, group=group)
            b.data.div_(world)
        else:
            dist.all_reduce(b.data, op=dist.ReduceOp.MAX, group=group)
