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


class RcclComm:
    """This rank's RCCL communicator behind the C ABI of include/cvae_dp.h (libcvae_dp.so): cvae_dp_init / cvae_dp_allreduce_sum, the
    exchange SURVEY.md §8(b) names, with no torch.distributed call on the data path.  The 128-byte unique id is created by rank 0 and
    handed round once, through the process group the launcher set up (any backend) — bootstrap only.

    all_reduce_sum(flat) reduces a flat fp32 / bf16 bucket in place as reduce-scatter + all-gather (both use every xGMI link of the fully
    connected node), enqueued on the current HIP stream; `async_on_side_stream` runs it on the communicator's own stream instead, ordered
    after what the current stream holds, and returns a handle whose wait() orders the current stream after the exchange."""
    _lib = None

    @classmethod
    def lib(cls):
        if cls._lib is None:
            import ctypes as C
            here = os.path.dirname(os.path.abspath(__file__))
            path = os.path.join(here, "libcvae_dp.so")
            if not os.path.exists(path):
                raise ImportError(f"{path} is missing: build it with `make -C causal_vae_amd/csrc` (the RCCL exchange has no fallback)")
            try:                                             # one RCCL per process: let the loader see the one torch already brought
                C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), mode=C.RTLD_GLOBAL)
            except OSError:
                pass
            L = C.CDLL(path)
            L.cvae_dp_strerror.restype = C.c_char_p
            L.cvae_dp_last_rccl_error.restype = C.c_char_p
            L.cvae_dp_unique_id.argtypes = [C.c_void_p]
            L.cvae_dp_init.argtypes = [C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
            for name in ("cvae_dp_allreduce_sum", "cvae_dp_reduce_scatter_sum", "cvae_dp_all_gather"):
                getattr(L, name).argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
            L.cvae_dp_destroy.argtypes = [C.c_void_p]
            L.cvae_dp_world.argtypes = [C.c_void_p]
            L.cvae_dp_rank.argtypes = [C.c_void_p]
            cls._lib = L
        return cls._lib

    @staticmethod
    def _check(rc, what):
        if rc != 0:
            L = RcclComm.lib()
            raise RuntimeError(f"{what}: {L.cvae_dp_strerror(rc).decode()} ({L.cvae_dp_last_rccl_error().decode()})")

    def __init__(self, rank=None, world=None, group=None, device=None):
        import ctypes as C
        L = self.lib()
        have_pg = dist.is_initialized()
        self.rank = (dist.get_rank(group) if have_pg else 0) if rank is None else int(rank)
        self.world = (dist.get_world_size(group) if have_pg else 1) if world is None else int(world)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        idbuf = (C.c_char * 128)()
        if self.rank == 0:
            self._check(L.cvae_dp_unique_id(idbuf), "cvae_dp_unique_id")
        if self.world > 1:
            if not have_pg:
                raise RuntimeError("RcclComm: world > 1 needs an initialised torch.distributed group to hand the unique id round")
            on_gpu = dist.get_backend(group) == "nccl"
            t = torch.tensor(list(bytes(idbuf)), dtype=torch.uint8, device=self.device if on_gpu else "cpu")
            dist.broadcast(t, src=(dist.get_global_rank(group, 0) if group is not None else 0), group=group)
            idbuf = (C.c_char * 128)(*bytes(t.cpu().tolist()))
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            self._check(L.cvae_dp_init(self.rank, self.world, idbuf, C.byref(self._h)), "cvae_dp_init")
        self._side = None

    def all_reduce_sum(self, flat):
        if flat.dtype not in (torch.float32, torch.bfloat16) or not flat.is_contiguous() or not flat.is_cuda:
            raise RuntimeError("RcclComm.all_reduce_sum: a contiguous fp32 / bf16 GPU tensor expected")
        code = 0 if flat.dtype == torch.float32 else 1
        self._check(self.lib().cvae_dp_allreduce_sum(self._h, flat.data_ptr(), flat.numel(), code, torch.cuda.current_stream(flat.device).cuda_stream), "cvae_dp_allreduce_sum")

    def reduce_scatter_sum(self, flat):
        """First half of the exchange: afterwards slice `rank` of numel / world elements holds the sum (numel % world == 0)."""
        code = 0 if flat.dtype == torch.float32 else 1
        self._check(self.lib().cvae_dp_reduce_scatter_sum(self._h, flat.data_ptr(), flat.numel(), code, torch.cuda.current_stream(flat.device).cuda_stream), "cvae_dp_reduce_scatter_sum")

    def all_gather(self, flat):
        """Second half: every rank's slice is copied to all ranks."""
        code = 0 if flat.dtype == torch.float32 else 1
        self._check(self.lib().cvae_dp_all_gather(self._h, flat.data_ptr(), flat.numel(), code, torch.cuda.current_stream(flat.device).cuda_stream), "cvae_dp_all_gather")

    def async_on_side_stream(self, flat):
        if self._side is None:
            self._side = torch.cuda.Stream(device=flat.device)
        self._side.wait_stream(torch.cuda.current_stream(flat.device))
        with torch.cuda.stream(self._side):
            self.all_reduce_sum(flat)
        side = self._side

        class _Work:
            def wait(self_inner):
                torch.cuda.current_stream(flat.device).wait_stream(side)
        return _Work()

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self.lib().cvae_dp_destroy(self._h)
            self._h = None


class GradAllReducer:
    """Sums `.grad` of `params` across ranks through one flat bucket.  Call it between backward and optimizer.step
    (the `grad_hook` of causal_cascade.train.train_step)."""

    def __init__(self, params, group=None, always_exchange=False, comm=None):
        """always_exchange: issue the collective even in a one-rank group (a one-rank RCCL all-reduce is the identity; used to run the
        capture / async-exchange machinery against RCCL on a single card).
        comm: an RcclComm — the exchange then goes through the C ABI of include/cvae_dp.h (reduce-scatter + all-gather on RCCL directly)
        instead of torch.distributed.all_reduce; same sums."""
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.always_exchange = bool(always_exchange)
        self.comm = comm
        self._flat = None

    def world_size(self):
        if self.comm is not None:
            return self.comm.world
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    def active(self):
        """True when a step has to exchange gradients: more than one rank, or always_exchange inside an initialised group."""
        return self.world_size() > 1 or (self.always_exchange and (dist.is_initialized() or self.comm is not None))

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
        if self.comm is not None:
            self.comm.all_reduce_sum(self._flat)
            return
        dist.all_reduce(self._flat, op=dist.ReduceOp.SUM, group=self.group)

    def all_reduce_async(self):
        """Start the exchange of the packed bucket and return its Work handle (None without a process group): on RCCL it runs on the
        communicator's stream, ordered after what the current stream holds so far, and `.wait()` orders the current stream after it."""
        if not self.active() or self._flat is None:
            return None
        if self.comm is not None:
            return self.comm.async_on_side_stream(self._flat)
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
    (ops.BioBottleneck) keeps its launches: the per-rank statistics of mechanism_net.0's output (a function of t alone) are gathered at the top
    of the step (2 x 64 floats), its backward all-reduces 2 x 64 floats and finishes mechanism_net.0's gradients in one extra small launch.
    Returns the module."""
    from .layers import BatchNorm1d
    for m in module.modules():
        if isinstance(m, BatchNorm1d):
            m.sync, m.sync_group = True, group
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
