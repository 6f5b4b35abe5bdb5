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
    # ---- 1. discriminator ----
    opt_d.zero_grad(set_to_none=True)
    with torch.no_grad():
        _, _, mu, logvar = vae(x, m, t, eps=eps_d)
        z = vae.reparameterize(mu, logvar, eps_d)
    loss_d = ops.SoftmaxCE.apply(discriminator(z), t_indices)
    loss_d.backward()
    opt_d.step()
    # ---- 2. VAE ----
    opt_vae.zero_grad(set_to_none=True)
    recon_x, m_hat, mu, logvar = vae(x, m, t, eps=eps_vae)
    loss_recon = ops.bce_sum(recon_x.view(-1, 784), x.view(-1, 784))
    loss_kld = ops.KLD.apply(mu, logvar) * beta
    loss_morph = ops.sse(m_hat, m) * 100
    z_sample = vae.reparameterize(mu, logvar, eps_adv)
    loss_adv = ops.UniformKL.apply(discriminator(z_sample)) * lambda_adv * 100
    loss = loss_recon + loss_kld + loss_morph + loss_adv
    loss.backward()
    opt_vae.step()
    return dict(loss=loss.detach(), loss_d=loss_d.detach(), recon=loss_recon.detach(), kld=loss_kld.detach(),
                morph=loss_morph.detach(), adv=loss_adv.detach())


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
