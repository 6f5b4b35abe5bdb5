"""The adversarial train step of mnist_test/01_baseline_causal_vae/train.py:34-93 on the HIP kernels.

train_step(vae, discriminator, opt_vae, opt_d, x, m, t, eps=None) runs one iteration of the loop body: the D step
(cross-entropy of D(z) against argmax t, :41-59) then the VAE step (BCE-sum + BETA*KLD + 100*MSE-sum(m) +
LAMBDA_ADV*100*KL(uniform || softmax D(z')), :65-89).  Of the reference's six eps draws per step only three reach a
result; the two redundant no-grad VAE forwards (:47-48) are not re-executed.  `eps` = (eps_d, eps_vae,
eps_adv) injects them for parity runs.
train_model(loader, epochs) mirrors train_model() (:11-103) for a caller-supplied loader of (x, m, t_onehot) batches.
"""
import torch

from .. import ops
from ..optim import FusedAdam
from .config import CONFIG
from .models import CausalMorphVAE12, LatentDiscriminator


def train_step(vae, discriminator, opt_vae, opt_d, x, m, t, eps=None, beta=None, lambda_adv=None):
    beta = CONFIG["BETA"] if beta is None else beta
    lambda_adv = CONFIG["LAMBDA_ADV"] if lambda_adv is None else lambda_adv
    eps_d, eps_vae, eps_adv = eps if eps is not None else (None, None, None)
    t_indices = torch.argmax(t, dim=1)
    with ops.zero_pool(8, x):                                # the step's five scalar loss sums share one zeroed block
        return _train_step(vae, discriminator, opt_vae, opt_d, x, m, t, t_indices, eps_d, eps_vae, eps_adv, beta, lambda_adv)


def _train_step(vae, discriminator, opt_vae, opt_d, x, m, t, t_indices, eps_d, eps_vae, eps_adv, beta, lambda_adv):
    # ---- 1. discriminator ----
    # The reference runs the whole VAE under no_grad here (:47) and keeps mu, logvar; the VAE step then runs the same encoder on the same x, m, t at the
    # same weights (opt_d.step() in between touches the discriminator only, :57-59) — the two encoder passes compute the same numbers.  ONE encoder pass,
    # with its autograd graph, serves both: the D step reads its values (detached), the VAE step continues from it (round 3: 9 launches fewer).
    opt_d.zero_grad(set_to_none=True)
    opt_vae.zero_grad(set_to_none=True)
    h = vae.encode(x, m, t)
    with torch.no_grad():
        z, _, _ = ops.LatentHead.apply(h.detach(), eps_d if eps_d is not None else vae._eps.draw(h.new_empty(h.shape[0], vae.z_dim)), None, False)
    loss_d = ops.SoftmaxCE.apply(discriminator(z), t_indices)
    ops.backward_from(loss_d)
    opt_d.step()
    # ---- 2. VAE ----
    # the adversarial term back-propagates THROUGH the (updated) discriminator; its own parameter gradients from this pass are dead in the reference too
    # (opt_d.zero_grad() drops them before the next D step reads anything), so they are not computed
    d_params = [p for p in discriminator.parameters() if p.requires_grad]
    for p in d_params:
        p.requires_grad_(False)
    try:
        recon_x, m_hat, mu, logvar, kld, z_sample = vae.forward_train(x, m, t, eps_vae, eps_adv, h=h)
        loss_recon = ops.bce_sum(recon_x.view(-1, 784), x.view(-1, 784))
        # loss = recon + BETA * kld + 100 * morph + LAMBDA_ADV * 100 * adv: one launch for the sum and the four logged terms, one for their gradients
        loss, parts = ops.weighted_sum([loss_recon, kld, ops.sse(m_hat, m), ops.UniformKL.apply(discriminator(z_sample))],
                                       [1.0, beta, 100.0, lambda_adv * 100.0], return_terms=True)
        ops.backward_from(loss)
    finally:
        for p in d_params:
            p.requires_grad_(True)
    opt_vae.step()
    return dict(loss=loss.detach(), loss_d=loss_d.detach(), recon=parts[0], kld=parts[1], morph=parts[2], adv=parts[3])


def train_model(train_loader, epochs=None, device=None, verbose=True):
    device = CONFIG["DEVICE"] if device is None else device
    epochs = CONFIG["EPOCHS"] if epochs is None else epochs
    vae, discriminator = CausalMorphVAE12().to(device), LatentDiscriminator().to(device)
    opt_vae, opt_d = FusedAdam(vae.parameters(), lr=CONFIG["LR"]), FusedAdam(discriminator.parameters(), lr=CONFIG["LR"])
    n = len(train_loader.dataset)
    for epoch in range(epochs):
        vae.train(); discriminator.train()
        tot = None
        for x, m, t in train_loader:
            r = train_step(vae, discriminator, opt_vae, opt_d, x.to(device), m.to(device), t.to(device))
            tot = r["loss"] if tot is None else tot + r["loss"]
        if verbose:
            print(f"Epoch {epoch + 1:02d} | Avg Loss: {float(tot.item()) / n:.1f}")
    return vae
