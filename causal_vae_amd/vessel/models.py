"""CausalVesselVAE on MI355X — drop-in for vessel_analysis/00_core/models.py:9-166 (the 2D 768 x 1280 vessel model).

Same constructor order (so `torch.manual_seed(s); CausalVesselVAE()` draws the reference's weights), same attribute tree and
state_dict keys (enc_conv.{0,1,3,4,..}, enc_fc, morph_predictor_shared/_mu/_logvar, dec_fc, dec_conv.{1,2,5,6,..}), same
forward(x, m, t) -> (recon_x, m_hat, mu, logvar, m_mu, m_logvar).  Encoder: 7 x [Conv2d(k4,s2,p1) + BatchNorm2d + LeakyReLU(0.2)]
on the conv_down kernels; decoder: 7 x [Upsample(x2, nearest) + Conv2d(k3,s1,p1) (+ BatchNorm2d + ReLU | Sigmoid)], each
Upsample + Conv2d pair as ONE transposed k4/s2/p1 product on the conv_up kernels (ops.Conv3ToK4).  The reference builds a first,
dead `dec_conv` (:71-105) that is overwritten at :108-134 — its RNG draws are reproduced so the live weights match.
"""
import torch
import torch.nn as nn

from .. import layers as hl
from .. import ops
from .config import CONFIG


class CausalVesselVAE(nn.Module):
    def __init__(self):
        super().__init__()
        self.m_dim, self.t_dim, self.z_dim = CONFIG["M_DIM"], CONFIG["T_DIM"], CONFIG["Z_DIM"]
        lrelu = lambda: nn.LeakyReLU(0.2)
        enc, cin = [], 1
        for cout in (32, 64, 128, 256, 512, 512, 512):
            enc += [hl.Conv2d(cin, cout, 4, 2, 1), hl.BatchNorm2d(cout), lrelu()]
            cin = cout
        self.enc_conv = hl.BNConvStack(*enc, nn.Flatten())
        self.enc_flat_dim = 512 * 6 * 10
        self.enc_fc = hl.MLP(hl.Linear(self.enc_flat_dim + self.m_dim + self.t_dim, 1024), hl.BatchNorm1d(1024), lrelu(),
                             hl.Linear(1024, self.z_dim * 2))
        self.morph_predictor_shared = hl.MLP(hl.Linear(self.t_dim, 64), lrelu(), hl.Linear(64, 64), lrelu())
        self.morph_predictor_mu = hl.Linear(64, self.m_dim)
        self.morph_predictor_logvar = hl.Linear(64, self.m_dim)
        self.dec_fc = hl.MLP(hl.Linear(self.m_dim + self.z_dim, 1024), hl.BatchNorm1d(1024), lrelu(), hl.Linear(1024, self.enc_flat_dim), nn.ReLU())
        # the reference's first, overwritten decoder (:70-105): nine ConvTranspose2d(k4,s2,p1) (+ BatchNorm2d, which draw nothing) —
        # only their RNG draws matter, so the live decoder below starts from the same generator state
        for cin, cout in ((512, 512),) * 4 + ((512, 256), (256, 128), (128, 64), (64, 32), (32, 1)):
            nn.ConvTranspose2d(cin, cout, 4, 2, 1)
        dec, cin = [], 512
        for cout in (512, 512, 256, 128, 64, 32):
            dec += [nn.Upsample(scale_factor=2, mode="nearest"), hl.UpConv2dK3(cin, cout, 3, 1, 1), hl.BatchNorm2d(cout), nn.ReLU()]
            cin = cout
        dec += [nn.Upsample(scale_factor=2, mode="nearest"), hl.UpConv2dK3(32, 1, 3, 1, 1), nn.Sigmoid()]
        self.dec_conv = hl.UpConvStack(*dec)
        self._eps = ops.EpsSource()

    def set_compute_dtype(self, dtype):
        hl.set_compute_dtype(self, dtype)
        return self

    def reparameterize(self, mu, logvar, eps=None):
        if eps is None:
            eps = self._eps.draw(mu)
        return ops.Reparameterize.apply(mu, logvar, eps)

    def forward(self, x, m, t, eps=None):
        if x.dim() != 4 or x.shape[1] != 1 or tuple(x.shape[2:]) != (CONFIG["IMG_HEIGHT"], CONFIG["IMG_WIDTH"]):
            raise RuntimeError(f"CausalVesselVAE expects [B, 1, {CONFIG['IMG_HEIGHT']}, {CONFIG['IMG_WIDTH']}] (the 512 x 6 x 10 latent map is hard-wired, "
                               f"models.py:45,162), got {tuple(x.shape)}")
        x_feat = self.enc_conv(x)
        mu, logvar = self.enc_fc(ops.cat([x_feat, m, t])).chunk(2, dim=1)
        logvar = ops.Clamp.apply(logvar, -10.0, 10.0)                      # :148
        mu = ops.Clamp.apply(mu, -100.0, 100.0)                            # :149
        z = self.reparameterize(mu, logvar, eps)
        h = self.morph_predictor_shared(t)
        m_mu = self.morph_predictor_mu(h)
        m_logvar = ops.Clamp.apply(self.morph_predictor_logvar(h), -10.0, 10.0)     # :156
        h_dec = self.dec_fc(ops.cat([m, z])).view(-1, 512, 6, 10)          # the decoder sees the REAL m (:161)
        recon_x = self.dec_conv(h_dec)
        return recon_x, m_mu, mu, logvar, m_mu, m_logvar
