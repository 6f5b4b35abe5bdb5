"""Gaussian-head MNIST variant on the HIP kernels: drop-in for mnist_test/06_model_experiment/models.py:6-85.

Differences from the baseline: P(M|T) is a Gaussian predicted by `morph_predictor_shared` +
`morph_predictor_mu` / `morph_predictor_logvar`; the decoder is fed the REAL m (:79); forward returns the 6-tuple
(recon_x, m_hat, mu, logvar, m_mu, m_logvar); `morph_predictor(t)` stays available as a helper returning the mean (:52-55).
"""
import torch
import torch.nn as nn

from .. import layers as hl
from .. import ops
from ..mnist_baseline.config import CONFIG


class CausalMorphVAE12(nn.Module):
    def __init__(self):
        super().__init__()
        self.m_dim, self.t_dim, self.z_dim = CONFIG["M_DIM"], CONFIG["T_DIM"], CONFIG["Z_DIM"]
        self.enc_conv = hl.ConvStack(hl.Conv2d(1, 32, 4, 2, 1), nn.ReLU(), hl.Conv2d(32, 64, 4, 2, 1), nn.ReLU(), nn.Flatten())
        self.enc_flat_dim = 64 * 7 * 7
        self.enc_fc = hl.MLP(hl.Linear(self.enc_flat_dim + self.m_dim + self.t_dim, 512), nn.ReLU(), hl.Linear(512, self.z_dim * 2))
        self.morph_predictor_shared = hl.MLP(hl.Linear(self.t_dim, 128), nn.ReLU())
        self.morph_predictor_mu = hl.Linear(128, self.m_dim)
        self.morph_predictor_logvar = hl.Linear(128, self.m_dim)
        self.dec_fc = hl.MLP(hl.Linear(self.m_dim + self.z_dim, self.enc_flat_dim), nn.ReLU())
        self.dec_conv = hl.DeconvStack(hl.ConvTranspose2d(64, 32, 4, 2, 1), nn.ReLU(), hl.ConvTranspose2d(32, 1, 4, 2, 1), nn.Sigmoid())
        self._eps = ops.EpsSource()

    def set_compute_dtype(self, dtype):
        hl.set_compute_dtype(self, dtype)
        return self

    def morph_predictor(self, t):
        return self.morph_predictor_mu(self.morph_predictor_shared(t))

    def reparameterize(self, mu, logvar, eps=None):
        if eps is None:
            eps = self._eps.draw(mu)
        return ops.Reparameterize.apply(mu, logvar, eps)

    def forward(self, x, m, t, eps=None):
        mu, logvar = self.enc_fc(self.enc_conv.forward_cat(x, [m, t])).chunk(2, dim=1)
        z = self.reparameterize(mu, logvar, eps)
        h = self.morph_predictor_shared(t)
        m_mu, m_logvar = self.morph_predictor_mu(h), self.morph_predictor_logvar(h)
        recon_x = self.dec_conv(self.dec_fc(ops.cat([m, z])).view(-1, 64, 7, 7))       # decoder sees the real m
        return recon_x, m_mu, mu, logvar, m_mu, m_logvar
