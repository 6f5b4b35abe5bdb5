"""ConditionalVAE on the HIP kernels: drop-in for mnist_test/03_measurement_approach/cvae_models.py:7-85 (the T -> X baseline that
ignores M): attributes enc_conv / enc_fc_mu / enc_fc_logvar / dec_fc / dec_conv, encode(x, t) / decode(z, t) / reparameterize /
forward(x, t) -> (recon_x, mu, logvar) with t a float one-hot; same state_dict keys and shapes.  The third encoder conv takes the
7x7 map to 3x3 (odd input extent: floor division, the last row / column is only ever read as padding's neighbour)."""
import torch
import torch.nn as nn

from .. import layers as hl
from .. import ops
from .config import CONFIG


class ConditionalVAE(nn.Module):
    def __init__(self):
        super().__init__()
        self.z_dim, self.t_dim = CONFIG["Z_DIM"], CONFIG["T_DIM"]
        self.enc_conv = hl.ConvStack(hl.Conv2d(1, 32, 4, 2, 1), nn.ReLU(), hl.Conv2d(32, 64, 4, 2, 1), nn.ReLU(),
                                     hl.Conv2d(64, 64, 4, 2, 1), nn.ReLU())
        self.enc_fc_mu = hl.Linear(576 + self.t_dim, self.z_dim)
        self.enc_fc_logvar = hl.Linear(576 + self.t_dim, self.z_dim)
        self.dec_fc = hl.Linear(self.z_dim + self.t_dim, 64 * 7 * 7)
        self.dec_conv = hl.DeconvStack(hl.ConvTranspose2d(64, 32, 4, 2, 1), nn.ReLU(), hl.ConvTranspose2d(32, 1, 4, 2, 1), nn.Sigmoid())
        self._eps = ops.EpsSource()

    def set_compute_dtype(self, dtype):
        hl.set_compute_dtype(self, dtype)
        return self

    def encode(self, x, t):
        h = self.enc_conv(x)                                  # [B, 576] in the reference's NCHW flatten order
        h_t = ops.cat([h.view(h.size(0), -1), t])
        return self.enc_fc_mu(h_t), self.enc_fc_logvar(h_t)

    def decode(self, z, t):
        h = self.dec_fc(ops.cat([z, t])).view(-1, 64, 7, 7)
        return self.dec_conv(h)

    def reparameterize(self, mu, logvar, eps=None):
        if eps is None:
            eps = self._eps.draw(mu)
        return ops.Reparameterize.apply(mu, logvar, eps)

    def forward(self, x, t, eps=None):
        mu, logvar = self.encode(x, t)
        z = self.reparameterize(mu, logvar, eps)
        return self.decode(z, t), mu, logvar
