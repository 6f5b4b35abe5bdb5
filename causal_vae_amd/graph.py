"""Whole-step HIP-graph capture: zero_grad -> forward -> ELBO -> backward -> Adam recorded once, replayed per step.

The model's step is ~150 small-to-medium kernel launches; eager Python issues them at ~4-5 us each, which is a third of
the step at 128^3 / B=4.  Every C-ABI entry point only enqueues on the current stream (include/cvae_hip.h), the Adam step
count and the Philox call count live on the device, and nothing in the step syncs with the host, so the whole step is
capturable with torch.cuda.graph (HIP graphs underneath) — no tracing compiler involved.

With a GradAllReducer (N > 1 ranks) the step is captured as two graphs around the eager RCCL all-reduce:
graph A = zero_grad..backward + pack of the flat gradient bucket, graph B = unpack + Adam.
overlap_exchange=True (models that expose their encoder output, `model._enc_out`) splits the backward there instead: graph A1 = forward +
the backward of decoder and bottleneck + pack of their bucket (~80 % of the gradient bytes), whose all-reduce then travels over xGMI
while graph A2 = the encoder's backward + pack of its bucket runs; graph B waits for both exchanges.
"""
import torch

from . import ops

from ._lib import CvaeError
from .optim import FusedAdam


def _capture(graph, **kw):
    """torch.cuda.graph(...) in "thread_local" capture-error mode: other threads of the process keep working during the capture.  With a
    process group alive, RCCL's watchdog thread polls its events (hipEventQuery) at any time; under the default "global" mode such a call
    from another thread invalidates the capture (hipErrorStreamCaptureUnsupported — seen intermittently with a one-rank RCCL group)."""
    return torch.cuda.graph(graph, capture_error_mode="thread_local", **kw)


def _check_sync_bn_capturable(model):
    """A SyncBatchNorm model (parallel.convert_sync_batchnorm) issues two small collectives INSIDE forward / backward, i.e. inside the captured region.
    RCCL collectives can be captured into a HIP graph (torch's ProcessGroupNCCL supports it); gloo's cannot (they go through the host)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return
    for mod in model.modules():
        if getattr(mod, "sync", False):
            group = getattr(mod, "sync_group", None)
            if dist.get_world_size(group) > 1 and dist.get_backend(group) != "nccl":
                raise CvaeError("GraphedTrainStep: SyncBatchNorm collectives can only be captured on an RCCL ('nccl') process group; run the eager train_step "
                                f"on '{dist.get_backend(group)}'")


class GraphedTrainStep:
    def __init__(self, model, optimizer, batch, loss_fn=None, reducer=None, warmup=3, overlap_exchange=False):
        """batch: (x, m, t) example tensors on the GPU (their storage becomes the static input buffers).
        loss_fn(model_outputs, x, m) -> (loss, *others): 0-dim tensors; `loss` is back-propagated.  None: the model's own
        forward_elbo(x, m, t) (what causal_cascade.train.train_step runs)."""
        if not isinstance(optimizer, FusedAdam) or not optimizer.device_step:
            raise CvaeError("GraphedTrainStep needs FusedAdam(..., device_step=True): the step count must live on the device")
        self.model, self.opt, self.reducer = model, optimizer, reducer
        _check_sync_bn_capturable(model)
        self.x, self.m, self.t = (b.clone() for b in batch)
        self.loss_fn = loss_fn
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):                          # allocator warm-up, lazy kernel attributes, Adam state
                self._fwd_bwd()
                self._reduce_eager()
                self.opt.step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.g1 = torch.cuda.CUDAGraph()
        self.g2 = self.g1b = None
        if overlap_exchange:
            self._capture_split()
        elif reducer is None or not reducer.active():
            with _capture(self.g1):
                self.out = self._fwd_bwd()
                self.opt.step()
        else:
            with _capture(self.g1):
                self.out = self._fwd_bwd()
                reducer.pack()
            self.g2 = torch.cuda.CUDAGraph()
            with _capture(self.g2, pool=self.g1.pool()):
                reducer.unpack()
                self.opt.step()

    def _fwd_bwd(self):
        self.opt.zero_grad(set_to_none=True)
        if self.loss_fn is None:
            bump = self.opt.claim_step_counter(self.x.device) if not getattr(self.opt, "_early", None) else None
            res = self.model.forward_elbo(self.x, self.m, self.t, bump=bump)
        else:
            res = self.loss_fn(self.model(self.x, self.m, self.t), self.x, self.m)
        ops.backward_from(res[0])
        return tuple(r.detach() for r in res)

    def _capture_split(self):
        from .parallel import GradAllReducer
        if self.loss_fn is not None or not hasattr(self.model, "early_gradient_parameters"):
            raise CvaeError("overlap_exchange needs the model's own forward_elbo and early_gradient_parameters()")
        params_a = [p for p in self.model.early_gradient_parameters() if p.requires_grad]
        ids_a = {id(p) for p in params_a}
        params_b = [p for p in self.model.parameters() if p.requires_grad and id(p) not in ids_a]
        group = self.reducer.group if self.reducer is not None else None
        always = self.reducer.always_exchange if self.reducer is not None else False
        comm = self.reducer.comm if self.reducer is not None else None
        self.red_a, self.red_b = GradAllReducer(params_a, group, always, comm), GradAllReducer(params_b, group, always, comm)
        self.red_a.pack(); self.red_b.pack()                 # the warm-up left every .grad in place: the flat buckets are allocated here,
        torch.cuda.synchronize()                             # in the ordinary pool, because RCCL touches them outside the graphs
        if hasattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch"):
            torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)      # the warm-up ran on another side stream: harmless here
        cap = torch.cuda.Stream()                            # ONE capture stream: the autograd nodes built in A1 run again in A2
        with _capture(self.g1, stream=cap):
            self.opt.zero_grad(set_to_none=True)
            res = self.model.forward_elbo(self.x, self.m, self.t, bump=self.opt.claim_step_counter(self.x.device))
            h = self.model._enc_out
            if h is None or not h.requires_grad:
                raise CvaeError("overlap_exchange: the model did not take the fused path that exposes its encoder output")
            ga = torch.autograd.grad(res[0], params_a + [h], grad_outputs=ops.cached_one(res[0]), retain_graph=True, allow_unused=True)
            for p, g in zip(params_a, ga[:-1]):
                p.grad = g
            gh = ga[-1]
            self.out = tuple(r.detach() for r in res)
            self.red_a.pack()
        self.g1b = torch.cuda.CUDAGraph()
        with _capture(self.g1b, pool=self.g1.pool(), stream=cap):
            gb = torch.autograd.grad(h, params_b, grad_outputs=gh, allow_unused=True)
            for p, g in zip(params_b, gb):
                p.grad = g
            self.red_b.pack()
        self.model._enc_out = None
        del h, gh, ga, gb, res
        self.g2 = torch.cuda.CUDAGraph()
        with _capture(self.g2, pool=self.g1.pool(), stream=cap):
            self.red_a.bind_views()                          # Adam reads the reduced buckets in place: no copy back
            self.red_b.bind_views()
            self.opt.step()

    def _reduce_eager(self):
        if self.reducer is not None:
            self.reducer()

    def __call__(self, x=None, m=None, t=None):
        """Replay one step; new inputs (same shapes) are copied into the static buffers first.  Returns the static loss tensors."""
        if x is not None:
            self.x.copy_(x, non_blocking=True)
        if m is not None:
            self.m.copy_(m, non_blocking=True)
        if t is not None:
            self.t.copy_(t, non_blocking=True)
        self.g1.replay()
        if self.g1b is not None:                             # split backward: bucket A is on the wire while the encoder's backward runs
            wa = self.red_a.all_reduce_async()
            self.g1b.replay()
            wb = self.red_b.all_reduce_async()
            for w in (wa, wb):
                if w is not None:
                    w.wait()
            self.g2.replay()
        elif self.g2 is not None:
            self.reducer.all_reduce()
            self.g2.replay()
        return self.out


class GraphedCallable:
    """Capture ANY enqueue-only step into one HIP graph: `fn()` is run `warmup` times eagerly (allocator, lazy kernel attributes, optimizer
    state), captured once, and replayed by `__call__()`.  `fn` must close over static tensors (copy new batches into them), use
    FusedAdam(device_step=True) optimizers and never sync with the host.  Used for the MNIST adversarial step (two optimizers), which at
    batch 1024 is ~110 launches of a few microseconds each: pure launch latency when issued from Python."""

    def __init__(self, fn, warmup=3):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with _capture(self.graph):
            self.out = fn()

    def __call__(self):
        self.graph.replay()
        return self.out
