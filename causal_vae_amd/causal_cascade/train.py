"""Train-loop surface of causal_cascade/train.py on the HIP kernels.

loss_function(recon_x, x, m_hat, m, mu, logvar, gamma=2000.0) -> (loss, recon_loss, m_loss)      (reference :5-17)
train_one_epoch(model, loader, optimizer, device) -> float                                         (reference :19-39)
The per-step `.item()` host syncs of the reference (:36-37) are replaced by on-device accumulation with one sync per
epoch; the returned number (sum of batch losses / len(dataset)) is the same.
"""
import torch

from .. import ops


def loss_function(recon_x, x, m_hat, m, mu, logvar, gamma=2000.0):
    # recon = MSE-sum(recon_x, x); m_loss = MSE-sum(m_hat, m); kld = -0.5*sum(1 + logvar - mu^2 - e^logvar); loss = recon + gamma*m_loss + kld
    loss, recon_loss, m_loss, _kld = ops.Elbo.apply(recon_x, x, m_hat, m, mu, logvar, gamma)
    return loss, recon_loss, m_loss


def train_step(model, optimizer, x, m, t, eps=None, gamma=2000.0, grad_hook=None):
    """zero_grad -> forward -> ELBO -> backward -> [grad_hook, e.g. the data-parallel all-reduce] -> optimizer.step.
    Returns (loss, recon_loss, m_loss) as 0-dim device tensors (no host sync)."""
    optimizer.zero_grad(set_to_none=True)
    if hasattr(model, "forward_elbo"):      # same numbers; skips materialising recon_x when the resize is an exact 2x (models.py)
        # FusedAdam's device step counter rides on the ELBO launch (no launch of its own); claimed only once gradients are sure to follow
        bump = optimizer.claim_step_counter(x.device) if (hasattr(optimizer, "claim_step_counter") and not getattr(optimizer, "_early", None)) else None
        loss, l_recon, l_m = model.forward_elbo(x, m, t, eps=eps, gamma=gamma, bump=bump)
    else:
        recon_x, m_hat, mu, logvar = model(x, m, t) if eps is None else model(x, m, t, eps=eps)
        loss, l_recon, l_m = loss_function(recon_x, x, m_hat, m, mu, logvar, gamma)
    ops.backward_from(loss)
    if grad_hook is not None:
        grad_hook()
    optimizer.step()
    return loss.detach(), l_recon.detach(), l_m.detach()


def train_one_epoch(model, loader, optimizer, device, grad_hook=None, progress=False):
    model.train()
    total = None
    it = loader
    if progress:
        from tqdm import tqdm
        it = tqdm(loader, desc="Training")
    for x, m, t in it:
        x, m, t = x.to(device, non_blocking=True), m.to(device, non_blocking=True), t.to(device, non_blocking=True)
        loss, _, _ = train_step(model, optimizer, x, m, t, grad_hook=grad_hook)
        total = loss if total is None else total + loss
    return float(total.item()) / len(loader.dataset)
