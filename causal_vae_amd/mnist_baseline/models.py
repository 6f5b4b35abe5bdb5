"""MNIST baseline models on the HIP kernels: drop-ins for mnist_test/01_baseline_causal_vae/models.py.

CausalMorphVAE12 (:6-72): attributes enc_conv / enc_fc / morph_predictor / dec_fc / dec_conv / reparameterize, forward(x, m, t)
-> (recon_x, m_hat, mu, logvar) with t a float one-hot; the consumers' `dec_fc(..).view(-1, 64, 7, 7)` -> `dec_conv` pattern
(visualize.py:87-89) works unchanged.  LatentDiscriminator (:93-111): `net`.
"""
import torch
import torch.nn as nn

from .. import layers as hl
from .. import ops
from .config import CONFIG


class CausalMorphVAE12(nn.Module):
    def __init__(self):
        super().__init__()
        self.m_dim, self.t_dim, self.z_dim = CONFIG["M_DIM"], CONFIG["T_DIM"], CONFIG["Z_DIM"]
        self.enc_conv = hl.ConvStack(hl.Conv2d(1, 32, 4, 2, 1), nn.ReLU(), hl.Conv2d(32, 64, 4, 2, 1), nn.ReLU(), nn.Flatten())
        self.enc_flat_dim = 64 * 7 * 7
        self.enc_fc = hl.MLP(hl.Linear(self.enc_flat_dim + self.m_dim + self.t_dim, 512), nn.ReLU(), hl.Linear(512, self.z_dim * 2))
        self.morph_predictor = hl.MLP(hl.Linear(self.t_dim, 128), nn.ReLU(), hl.Linear(128, self.m_dim))
        self.dec_fc = hl.MLP(hl.Linear(self.m_dim + self.z_dim, self.enc_flat_dim), nn.ReLU())
        self.dec_conv = hl.DeconvStack(hl.ConvTranspose2d(64, 32, 4, 2, 1), nn.ReLU(), hl.ConvTranspose2d(32, 1, 4, 2, 1), nn.Sigmoid())
        self._eps = ops.EpsSource()

    def set_compute_dtype(self, dtype):
        hl.set_compute_dtype(self, dtype)
        return self

    def reparameterize(self, mu, logvar, eps=None):
        if eps is None:
            eps = self._eps.draw(mu)
        return ops.Reparameterize.apply(mu, logvar, eps)

    def decode(self, m_hat, z):
        """Decoder half: cat[m_hat, z] -> dec_fc -> dec_conv (check_mnist_counterfactual.py:72-74, any number of rows at once)."""
        return self.dec_conv(self.dec_fc(ops.cat([m_hat, z])).view(-1, 64, 7, 7))

    def encode(self, x, m, t):
        """Encoder half: enc_conv -> cat[x_feat, m, t] -> enc_fc; returns the [B, 2 z] head (mu | logvar) (models.py:55-60 of the reference)."""
        return self.enc_fc(self.enc_conv.forward_cat(x, [m, t]))

    def forward_train(self, x, m, t, eps_vae=None, eps_adv=None, h=None):
        """forward() plus what the adversarial step takes from the same latent head (train.py:65-81 of the reference): the KLD sum and the second
        sample z' = reparameterize(mu, logvar) that feeds the discriminator — mu | logvar are consumed where the encoder leaves them (ops.LatentHead),
        both noise draws are one Philox launch.  Returns (recon_x, m_hat, mu, logvar, kld, z_adv)."""
        if h is None:                                        # h: the encoder head of THIS x, m, t at THESE weights, computed by the caller (train_step shares it with the D step)
            h = self.encode(x, m, t)
        B, Z = h.shape[0], self.z_dim
        if eps_vae is None or eps_adv is None:
            e = self._eps.draw(h.new_empty(2 * B, Z))
            eps_vae, eps_adv = (e[:B] if eps_vae is None else eps_vae), (e[B:] if eps_adv is None else eps_adv)
        z, z_adv, kld = ops.LatentHead.apply(h, eps_vae, eps_adv, True)
        m_hat = self.morph_predictor(t)
        recon_x = self.dec_conv(self.dec_fc(ops.cat([m_hat, z])).view(-1, 64, 7, 7))
        mu, logvar = h.chunk(2, dim=1)
        return recon_x, m_hat, mu, logvar, kld, z_adv

    def forward(self, x, m, t, eps=None):
        mu, logvar = self.encode(x, m, t).chunk(2, dim=1)
        z = self.reparameterize(mu, logvar, eps)
        m_hat = self.morph_predictor(t)
        h = self.dec_fc(ops.cat([m_hat, z])).view(-1, 64, 7, 7)
        recon_x = self.dec_conv(h)
        return recon_x, m_hat, mu, logvar


class LatentDiscriminator(nn.Module):
    def __init__(self):
        super().__init__()
        self.z_dim, self.t_dim = CONFIG["Z_DIM"], CONFIG["T_DIM"]
        self.net = hl.MLP(hl.Linear(self.z_dim, 64), nn.LeakyReLU(0.2), hl.Linear(64, 64), nn.LeakyReLU(0.2), hl.Linear(64, self.t_dim))

    def forward(self, z):
        return self.net(z)
