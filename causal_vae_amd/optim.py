"""Adam and gradient clipping on the HIP kernels (cvae_adam_step / cvae_sqnorm / cvae_clip_coef).

`FusedAdam` is a torch.optim.Optimizer, so `train_one_epoch(model, loader, optimizer, device)` takes it wherever the
reference passes `optim.Adam(model.parameters(), lr=1e-3)` (causal_cascade/main.py:50); stock torch optimizers keep
working too because gradients arrive through autograd in `param.grad`.
"""
import torch

from . import _lib as L
from ._lib import lib, check, ptr, stream


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam semantics (no weight decay, no amsgrad): one kernel launch per parameter tensor, 28 B/param."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    @torch.no_grad()
    def step(self, closure=None, grad_scale=None):
        """grad_scale: optional 0-dim device tensor multiplied into every gradient (clip_grad_norm_ coefficient)."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                L.require_gpu(p)
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise L.CvaeError("FusedAdam: contiguous float32 parameters expected")
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                st["step"] += 1
                t = st["step"]
                g = p.grad.contiguous()
                check(lib.cvae_adam_step(ptr(p), ptr(g), ptr(st["exp_avg"]), ptr(st["exp_avg_sq"]), p.numel(), group["lr"], b1, b2,
                                         group["eps"], 1.0 - b1 ** t, 1.0 - b2 ** t, ptr(grad_scale), stream()), "adam_step")
        return loss


@torch.no_grad()
def clip_grad_norm_(parameters, max_norm):
    """torch.nn.utils.clip_grad_norm_ (vessel_analysis/01_train/train.py:85) without a host sync: returns
    (total_norm ** 2, coef) as 0-dim device tensors; gradients are scaled in place by coef = min(1, max_norm/(norm+1e-6))."""
    params = [p for p in parameters if p.grad is not None]
    if not params:
        return None, None
    dev = params[0].device
    sq = torch.zeros((), dtype=torch.float32, device=dev)
    coef = torch.empty((), dtype=torch.float32, device=dev)
    for p in params:
        g = p.grad
        if not g.is_contiguous():
            g = p.grad = g.contiguous()
        check(lib.cvae_sqnorm(ptr(g), ptr(sq), g.numel(), stream()), "sqnorm")
    check(lib.cvae_clip_coef(ptr(sq), ptr(coef), float(max_norm), stream()), "clip_coef")
    for p in params:
        check(lib.cvae_scale(ptr(p.grad), p.grad.numel(), ptr(coef), stream()), "scale")
    return sq, coef
