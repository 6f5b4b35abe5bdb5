"""Functional CPU restatement of the reference train steps.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Autograd here is stock PyTorch
CPU autograd over the functional forward of oracle/functional.py; Adam is
restated from its published update rule (torch.optim.Adam defaults used by the
reference: lr, betas=(0.9, 0.999), eps=1e-8, no weight decay, no amsgrad —
causal_cascade/main.py:50, mnist_test/01_baseline_causal_vae/train.py:21-22).
"""
import torch
import torch.nn.functional as F

from . import functional as fn

_BUFFER_SUFFIXES = ("running_mean", "running_var", "num_batches_tracked")


def trainable_keys(sd):
    return [k for k in sd if not k.endswith(_BUFFER_SUFFIXES)]


def _leaves(sd):
    """Detached copy of sd whose trainable entries require grad."""
    out = {}
    for k, v in sd.items():
        out[k] = v.detach().clone()
        if not k.endswith(_BUFFER_SUFFIXES):
            out[k].requires_grad_(True)
    return out


def adam_init(sd):
    return {"step": 0,
            "m": {k: torch.zeros_like(sd[k]) for k in trainable_keys(sd)},
            "v": {k: torch.zeros_like(sd[k]) for k in trainable_keys(sd)}}


def adam_update(sd, grads, state, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8):
    """In-place Adam step, the non-foreach/non-fused formulation of torch.optim.Adam:
    m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
    p -= (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)."""
    state["step"] += 1
    t = state["step"]
    bc1, bc2 = 1 - beta1 ** t, 1 - beta2 ** t
    with torch.no_grad():
        for k, g in grads.items():
            if g is None:
                continue
            m, v = state["m"][k], state["v"][k]
            m.mul_(beta1).add_(g, alpha=1 - beta1)
            v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
            denom = (v.sqrt() / (bc2 ** 0.5)).add_(eps)
            sd[k].addcdiv_(m, denom, value=-lr / bc1)


def clip_grad_norm(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_ (vessel_analysis/01_train/train.py:85): global L2
    norm over all grads; scale by max_norm / (norm + 1e-6) clamped to 1."""
    gs = [g for g in grads.values() if g is not None]
    total = torch.sqrt(sum((g.detach() ** 2).sum() for g in gs))
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    with torch.no_grad():
        for g in gs:
            g.mul_(coef)
    return total


def cascade_train_step(sd, x, m, t, eps, adam_state=None, *, lr=1e-3, gamma=2000.0, nd=None,
                       apply_update=True, conv_dtype=None):
    """One iteration of train_one_epoch (causal_cascade/train.py:26-37): zero_grad ->
    forward -> loss -> backward -> Adam.  ``sd`` is updated in place (parameters and BN
    running stats).  Returns dict(loss, recon, m_loss, kld, grads, outputs)."""
    leaves = _leaves(sd)
    out = fn.bio_vae_forward(leaves, x, m, t, eps, nd=nd, training=True, conv_dtype=conv_dtype)
    loss, recon, m_loss, kld = fn.cascade_loss(out["recon_x"], x, out["m_hat"], m,
                                               out["mu"], out["logvar"], gamma)
    keys = trainable_keys(sd)
    gl = torch.autograd.grad(loss, [leaves[k] for k in keys], allow_unused=True)
    grads = dict(zip(keys, gl))
    with torch.no_grad():
        for k in sd:                                   # BN running stats updated by the forward
            if k.endswith(_BUFFER_SUFFIXES):
                sd[k].copy_(leaves[k])
    if apply_update:
        if adam_state is None:
            adam_state = adam_init(sd)
        adam_update(sd, grads, adam_state, lr=lr)
    return dict(loss=loss.detach(), recon=recon.detach(), m_loss=m_loss.detach(), kld=kld.detach(),
                grads=grads, outputs={k: v.detach() for k, v in out.items()}, adam_state=adam_state)


def cvae_train_step(sd, x, t, eps, adam_state=None, *, lr=1e-3, apply_update=True):
    """One iteration of the ConditionalVAE loop (mnist_test/03_measurement_approach/cvae_train.py:31-47)."""
    leaves = _leaves(sd)
    out = fn.cvae_forward(leaves, x, t, eps)
    loss, recon, kld = fn.cvae_loss(out["recon_x"], x, out["mu"], out["logvar"])
    keys = trainable_keys(sd)
    grads = dict(zip(keys, torch.autograd.grad(loss, [leaves[k] for k in keys])))
    if apply_update:
        if adam_state is None:
            adam_state = adam_init(sd)
        adam_update(sd, grads, adam_state, lr=lr)
    return dict(loss=loss.detach(), recon=recon.detach(), kld=kld.detach(), grads=grads,
                outputs={k: v.detach() for k, v in out.items()}, adam_state=adam_state)


def mnist_adversarial_step(sd_vae, sd_d, x, m, t, eps_d, eps_vae, eps_adv, adam_vae=None, adam_d=None,
                           *, lr=1e-3, beta=1.0, lambda_adv=10.0, apply_update=True, gaussian=False):
    """One iteration of train_model's inner loop (mnist_test/01_baseline_causal_vae/train.py:34-93).
    ``gaussian=True`` is the 06_model_experiment variant (mnist_test/06_model_experiment/train.py:40-97): 6-tuple forward with
    the decoder on the real m, and the morph term is the Gaussian NLL (:79) instead of 100 * MSE-sum.

    The reference draws six eps per step; only three reach any result and
    are injected here: ``eps_d`` (:51, D's input), ``eps_vae`` (:67 forward, recon path) and
    ``eps_adv`` (:78, adversarial path).  D's gradients from the VAE step are discarded by the
    reference (opt_d.zero_grad at the next iteration, :41), so they are not returned.
    """
    t_dim = t.shape[1]
    t_idx = torch.argmax(t, dim=1)                                        # :36
    # ---- D step (:41-59) ----
    with torch.no_grad():
        fwd = fn.morph_vae6_forward if gaussian else fn.morph_vae_forward
        o = fwd(sd_vae, x, m, t, torch.zeros_like(eps_d))                 # :49 (eps unused for mu/logvar)
        z_d = o["mu"] + eps_d * torch.exp(0.5 * o["logvar"])              # :50-52
    ld = _leaves(sd_d)
    loss_d = F.cross_entropy(fn.discriminator_forward(ld, z_d), t_idx)    # :55-56
    kd = trainable_keys(sd_d)
    grads_d = dict(zip(kd, torch.autograd.grad(loss_d, [ld[k] for k in kd])))
    if apply_update:
        adam_d = adam_init(sd_d) if adam_d is None else adam_d
        adam_update(sd_d, grads_d, adam_d, lr=lr)                         # :59
    # ---- VAE step (:65-89) — uses the *updated* discriminator ----
    lv = _leaves(sd_vae)
    o = fwd(lv, x, m, t, eps_vae)                                         # :67
    z_sample = fn.reparameterize(o["mu"], o["logvar"], eps_adv)           # :78
    d_fake = fn.discriminator_forward(sd_d, z_sample)                     # :79
    loss, recon, kld, morph, adv = fn.mnist_vae_losses(o["recon_x"], x, o["m_hat"], m, o["mu"], o["logvar"],
                                                       d_fake, beta=beta, lambda_adv=lambda_adv, t_dim=t_dim)
    if gaussian:
        morph = fn.gaussian_nll(m, o["m_mu"], o["m_logvar"])              # 06/train.py:79
        loss = recon + kld + morph + adv
    kv = trainable_keys(sd_vae)
    grads_v = dict(zip(kv, torch.autograd.grad(loss, [lv[k] for k in kv])))
    if apply_update:
        adam_vae = adam_init(sd_vae) if adam_vae is None else adam_vae
        adam_update(sd_vae, grads_v, adam_vae, lr=lr)                     # :89
    return dict(loss_d=loss_d.detach(), loss=loss.detach(), recon=recon.detach(), kld=kld.detach(),
                morph=morph.detach(), adv=adv.detach(), grads_vae=grads_v, grads_d=grads_d,
                outputs={k: v.detach() for k, v in o.items()}, adam_vae=adam_vae, adam_d=adam_d)
