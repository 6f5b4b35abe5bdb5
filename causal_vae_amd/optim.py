"""Adam and gradient clipping on the HIP kernels (cvae_adam_step / cvae_sqnorm / cvae_clip_coef).

`FusedAdam` is a torch.optim.Optimizer, so `train_one_epoch(model, loader, optimizer, device)` takes it wherever the
reference passes `optim.Adam(model.parameters(), lr=1e-3)` (causal_cascade/main.py:50); stock torch optimizers keep
working too because gradients arrive through autograd in `param.grad`.
"""
import torch

from . import _lib as L
from ._lib import lib, check, ptr, stream


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam semantics (no weight decay, no amsgrad).  One multi-tensor launch per parameter group
    (cvae_adam_multi: the pointer table rides in the kernel arguments), 28 B/param of HBM traffic.

    device_step=True keeps the step counter on the device and derives the bias corrections in-kernel, which makes
    `step()` safe to capture into a HIP graph (the captured launch replays with an advancing counter)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, device_step=False):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self.device_step = device_step
        self._step_dev = None
        self._early, self._early_hooks, self._side = [], [], None
        self._early_left, self._early_ran, self._counted = 0, False, False

    # ---- Adam of the early-finished gradients under the rest of the backward --------------------------------------------------
    def overlap_backward(self, early_params):
        """Run the update of `early_params` on a side stream as soon as the LAST of their gradients has been accumulated, i.e. while
        the backward of everything upstream of them is still running (Adam is HBM-bound, the remaining convolutions are not).
        step() then joins the side stream and updates the other parameters.  Single parameter group, no grad_scale; call with
        an empty list to switch it off.  Gradient exchange between ranks must not be pending on these parameters."""
        for h in self._early_hooks:
            h.remove()
        self._early_hooks, self._early = [], [p for p in early_params if p.requires_grad]
        if not self._early:
            return
        if len(self.param_groups) != 1:
            raise L.CvaeError("FusedAdam.overlap_backward: one parameter group expected")
        self._early_ids = {id(p) for p in self._early}
        self._early_left, self._early_ran = len(self._early), False
        for p in self._early:
            self._early_hooks.append(p.register_post_accumulate_grad_hook(self._early_hook))

    def _early_hook(self, p):
        self._early_left -= 1
        if self._early_left == 0:
            from . import ops
            ops.flush_pending_wgrads()                       # queued conv weight gradients must exist before they are read
            main = torch.cuda.current_stream()
            if self._side is None:
                self._side = torch.cuda.Stream()
            with torch.no_grad():
                self._count_step(self._early[0].device)          # on the main stream, before the fork
                self._side.wait_stream(main)
                with torch.cuda.stream(self._side):
                    self._update(self.param_groups[0], self._early, None)
            self._early_ran = True

    def zero_grad(self, set_to_none=True):
        self._early_left, self._early_ran = len(self._early), False
        return super().zero_grad(set_to_none=set_to_none)

    def claim_step_counter(self, device):
        """The device step counter, for a caller that promises to advance it by exactly one before this step's step() (the flagship model's ELBO launch
        does: one single-block launch fewer).  step() then does not count again.  None when the counter lives on the host."""
        if not self.device_step:
            return None
        if self._step_dev is None:
            self._step_dev = torch.zeros((), dtype=torch.int32, device=device)
        self._counted = True
        return self._step_dev

    def _count_step(self, device):
        if self.device_step and not self._counted:
            if self._step_dev is None:
                self._step_dev = torch.zeros((), dtype=torch.int32, device=device)
            check(lib.cvae_counter_add(ptr(self._step_dev), 1, stream()), "counter_add")
            self._counted = True

    def _update(self, group, ps, grad_scale):
        import ctypes as C
        b1, b2 = group["betas"]
        L.require_gpu(*ps)
        t = None
        gs = []
        for p in ps:
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise L.CvaeError("FusedAdam: contiguous float32 parameters expected")
            st = self.state[p]
            if not st:
                st["step"] = 0
                st["exp_avg"] = torch.zeros_like(p)
                st["exp_avg_sq"] = torch.zeros_like(p)
            st["step"] += 1
            if t is None:
                t = st["step"]
            elif t != st["step"]:
                raise L.CvaeError("FusedAdam: parameters of one group must share a step count")
            gs.append(p.grad if p.grad.is_contiguous() else p.grad.contiguous())
        n = len(ps)
        arr = lambda vals: (C.c_void_p * n)(*vals)
        sizes = (C.c_int64 * n)(*[p.numel() for p in ps])
        step_ptr = ptr(self._step_dev) if self.device_step else None
        check(lib.cvae_adam_multi(arr([p.data_ptr() for p in ps]), arr([g.data_ptr() for g in gs]),
                                  arr([self.state[p]["exp_avg"].data_ptr() for p in ps]),
                                  arr([self.state[p]["exp_avg_sq"].data_ptr() for p in ps]), sizes, n, group["lr"], b1, b2,
                                  group["eps"], 1.0 - b1 ** t, 1.0 - b2 ** t, step_ptr, ptr(grad_scale), stream()), "adam_multi")

    # ---- checkpointing: the step count lives on the device when device_step=True (it advances under HIP-graph replay while the host
    # copies in self.state do not), so it is written back before saving and restored after loading; without this a resumed run would
    # restart Adam's bias correction on warm moments ----
    def _sync_step_from_device(self):
        if self.device_step and self._step_dev is not None:
            t = int(self._step_dev.item())
            for st in self.state.values():
                if "step" in st:
                    st["step"] = t

    def state_dict(self):
        self._sync_step_from_device()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        """Restores IN PLACE where the moment buffers and the device step counter already exist: a HIP graph captured with this optimizer
        (graph.GraphedTrainStep) replays against those very tensors, and fresh ones would be silently ignored by the replayed update."""
        old = {p: (st.get("exp_avg"), st.get("exp_avg_sq")) for p, st in self.state.items()}
        super().load_state_dict(state_dict)
        for p, st in self.state.items():
            for key, prev in zip(("exp_avg", "exp_avg_sq"), old.get(p, (None, None))):
                new = st.get(key)
                if prev is not None and new is not None and new is not prev and prev.shape == new.shape and prev.device == new.device:
                    prev.copy_(new)
                    st[key] = prev
        steps = {int(st["step"]) for st in self.state.values() if "step" in st}
        if len(steps) > 1:
            raise L.CvaeError("FusedAdam.load_state_dict: parameters carry different step counts")
        if self.device_step and steps:
            dev = next(iter(self.state)).device
            t = steps.pop()
            if self._step_dev is not None and self._step_dev.device == dev:
                self._step_dev.fill_(t)
            else:
                self._step_dev = torch.full((), t, dtype=torch.int32, device=dev)
        self._counted = False

    @torch.no_grad()
    def step(self, closure=None, grad_scale=None):
        """grad_scale: optional 0-dim device tensor multiplied into every gradient (clip_grad_norm_ coefficient)."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        from . import ops
        ops.join_side_streams()                              # weight gradients still running on a side stream (ops.DEFER_JOIN)
        early_done = self._early_ran
        if early_done and grad_scale is not None:
            raise L.CvaeError("FusedAdam: grad_scale cannot be combined with overlap_backward (part of the update has already run)")
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.grad is not None and not (early_done and id(p) in self._early_ids)]
            if ps:
                self._count_step(ps[0].device)
                self._update(group, ps, grad_scale)
        if early_done:
            torch.cuda.current_stream().wait_stream(self._side)
        self._counted, self._early_ran, self._early_left = False, False, len(self._early)
        return loss


@torch.no_grad()
def clip_grad_norm_(parameters, max_norm, scale_grads=True):
    """torch.nn.utils.clip_grad_norm_ (vessel_analysis/01_train/train.py:85) without a host sync: returns
    (total_norm ** 2, coef) as 0-dim device tensors, coef = min(1, max_norm/(norm+1e-6)).  scale_grads=True (the torch semantics): the
    gradients are scaled in place by coef.  scale_grads=False: they are left as they are — hand coef to FusedAdam.step(grad_scale=coef),
    which multiplies it into every gradient as it reads it (same update, one launch per tensor fewer).  The norm is one multi-tensor launch."""
    import ctypes as C
    from . import ops
    ops.join_side_streams()
    params = [p for p in parameters if p.grad is not None]
    if not params:
        return None, None
    dev = params[0].device
    sq = torch.zeros((), dtype=torch.float32, device=dev)
    coef = torch.empty((), dtype=torch.float32, device=dev)
    nws = int(lib.cvae_reduce_workspace_bytes())
    ws = torch.empty(nws // 4, dtype=torch.float32, device=dev)     # per-workgroup partial sums, added in a fixed order (no float atomics)
    gs = []
    for p in params:
        if not p.grad.is_contiguous():
            p.grad = p.grad.contiguous()
        if p.grad.dtype != torch.float32:
            raise L.CvaeError("clip_grad_norm_: float32 gradients expected")
        gs.append(p.grad)
    n = len(gs)
    check(lib.cvae_sqnorm_multi((C.c_void_p * n)(*[g.data_ptr() for g in gs]), (C.c_int64 * n)(*[g.numel() for g in gs]), n, ptr(sq), ptr(ws), nws, stream()), "sqnorm_multi")
    check(lib.cvae_clip_coef(ptr(sq), ptr(coef), float(max_norm), stream()), "clip_coef")
    if scale_grads:
        for g in gs:
            check(lib.cvae_scale(ptr(g), g.numel(), ptr(coef), stream()), "scale")
    return sq, coef
