"""VAE-side losses and step of mnist_test/06_model_experiment/train.py:65-97 on the HIP kernels: BCE-sum + BETA*KLD +
Gaussian NLL of m under N(m_mu, exp(m_logvar)) (:79) + the adversarial uniform-KL term; the D step is the baseline's."""
import torch

from .. import ops
from ..mnist_baseline.config import CONFIG


def vae_losses(recon_x, x, mu, logvar, m, m_mu, m_logvar, d_logits_fake, beta=None, lambda_adv=None):
    beta = CONFIG["BETA"] if beta is None else beta
    lambda_adv = CONFIG["LAMBDA_ADV"] if lambda_adv is None else lambda_adv
    loss_recon = ops.bce_sum(recon_x.view(-1, 784), x.view(-1, 784))
    loss_kld = ops.KLD.apply(mu, logvar) * beta
    loss_morph = ops.GaussNLL.apply(m, m_mu, m_logvar)
    loss_adv = ops.UniformKL.apply(d_logits_fake) * lambda_adv * 100
    return loss_recon + loss_kld + loss_morph + loss_adv, loss_recon, loss_kld, loss_morph, loss_adv


def train_step(vae, discriminator, opt_vae, opt_d, x, m, t, eps=None):
    eps_d, eps_vae, eps_adv = eps if eps is not None else (None, None, None)
    t_indices = torch.argmax(t, dim=1)
    opt_d.zero_grad(set_to_none=True)
    with torch.no_grad():
        out = vae(x, m, t, eps=eps_d)
        z = vae.reparameterize(out[2], out[3], eps_d)
    loss_d = ops.SoftmaxCE.apply(discriminator(z), t_indices)
    loss_d.backward()
    opt_d.step()
    opt_vae.zero_grad(set_to_none=True)
    recon_x, _, mu, logvar, m_mu, m_logvar = vae(x, m, t, eps=eps_vae)
    d_fake = discriminator(vae.reparameterize(mu, logvar, eps_adv))
    loss, l_recon, l_kld, l_morph, l_adv = vae_losses(recon_x, x, mu, logvar, m, m_mu, m_logvar, d_fake)
    loss.backward()
    opt_vae.step()
    return dict(loss=loss.detach(), loss_d=loss_d.detach(), recon=l_recon.detach(), kld=l_kld.detach(), morph=l_morph.detach(), adv=l_adv.detach())
