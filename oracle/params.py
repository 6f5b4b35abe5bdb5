"""Parameter initialisation that reproduces the reference constructors' RNG draws.

TEST INFRASTRUCTURE (see oracle/__init__.py).

The reference builds its layers with stock ``torch.nn`` constructors, whose
default initialisers draw from the global torch generator in construction
order.  ``init_state_dict(kind, seed)`` instantiates *bare layers* in exactly
that order and files their tensors under the reference's ``state_dict`` key
names, so ``torch.manual_seed(s); RefClass().state_dict()`` and
``init_state_dict(kind, s)`` are bit-identical (pinned by the golden test).

Construction orders followed:
  bio2d / bio3d : causal_cascade/models.py:12-55  (enc_conv, enc_fc, fc_mu,
                  fc_logvar, mechanism_net, dec_input, dec_conv)
  morph12       : mnist_test/01_baseline_causal_vae/models.py:19-48
  morph12g      : mnist_test/06_model_experiment/models.py:19-50 (Gaussian head)
  disc          : mnist_test/01_baseline_causal_vae/models.py:102-108
  vessel2d      : vessel_analysis/00_core/models.py:32-134 (incl. the dead first dec_conv :70-105, whose draws advance the RNG)
"""
from collections import OrderedDict

import torch
import torch.nn as nn

MODEL_KINDS = ("bio2d", "bio3d", "morph12", "morph12g", "disc", "vessel2d", "cvae")


def _file(sd, prefix, layer):
    for k, v in layer.state_dict().items():
        sd[f"{prefix}.{k}"] = v.detach().clone()


def _bio(nd, img_channels, m_dim, t_dim, latent_dim):
    conv = nn.Conv2d if nd == 2 else nn.Conv3d
    convT = nn.ConvTranspose2d if nd == 2 else nn.ConvTranspose3d
    flat = 256 * 4 ** nd
    sd = OrderedDict()
    chans = [img_channels, 32, 64, 128, 256]
    for i in range(4):                                  # models.py:13-16
        _file(sd, f"enc_conv.{2 * i}", conv(chans[i], chans[i + 1], 4, 2, 1))
    _file(sd, "enc_fc.0", nn.Linear(flat + m_dim + t_dim, 512))   # :25
    _file(sd, "enc_fc.2", nn.Linear(512, 256))                     # :27
    _file(sd, "fc_mu", nn.Linear(256, latent_dim))                 # :30
    _file(sd, "fc_logvar", nn.Linear(256, latent_dim))             # :31
    _file(sd, "mechanism_net.0", nn.Linear(t_dim, 64))             # :35
    _file(sd, "mechanism_net.1", nn.BatchNorm1d(64))               # :36
    _file(sd, "mechanism_net.3", nn.Linear(64, 64))                # :38
    _file(sd, "mechanism_net.5", nn.Linear(64, m_dim))             # :40
    _file(sd, "dec_input", nn.Linear(latent_dim + m_dim, flat))    # :44
    dch = [256, 128, 64, 32, img_channels]
    for i in range(4):                                  # :51-54
        _file(sd, f"dec_conv.{2 * i}", convT(dch[i], dch[i + 1], 4, 2, 1))
    return sd


def _morph12(m_dim, t_dim, z_dim, gaussian_head):
    sd = OrderedDict()
    _file(sd, "enc_conv.0", nn.Conv2d(1, 32, 4, 2, 1))
    _file(sd, "enc_conv.2", nn.Conv2d(32, 64, 4, 2, 1))
    flat = 64 * 7 * 7
    _file(sd, "enc_fc.0", nn.Linear(flat + m_dim + t_dim, 512))
    _file(sd, "enc_fc.2", nn.Linear(512, 2 * z_dim))
    if gaussian_head:                                   # 06/models.py:34-39
        _file(sd, "morph_predictor_shared.0", nn.Linear(t_dim, 128))
        _file(sd, "morph_predictor_mu", nn.Linear(128, m_dim))
        _file(sd, "morph_predictor_logvar", nn.Linear(128, m_dim))
    else:                                               # 01/models.py:33-37
        _file(sd, "morph_predictor.0", nn.Linear(t_dim, 128))
        _file(sd, "morph_predictor.2", nn.Linear(128, m_dim))
    _file(sd, "dec_fc.0", nn.Linear(m_dim + z_dim, flat))
    _file(sd, "dec_conv.0", nn.ConvTranspose2d(64, 32, 4, 2, 1))
    _file(sd, "dec_conv.2", nn.ConvTranspose2d(32, 1, 4, 2, 1))
    return sd


def _vessel2d(m_dim, t_dim, z_dim):
    sd = OrderedDict()
    cin = 1
    for i, cout in enumerate((32, 64, 128, 256, 512, 512, 512)):          # models.py:32-40
        _file(sd, f"enc_conv.{3 * i}", nn.Conv2d(cin, cout, 4, 2, 1))
        _file(sd, f"enc_conv.{3 * i + 1}", nn.BatchNorm2d(cout))
        cin = cout
    flat = 512 * 6 * 10
    _file(sd, "enc_fc.0", nn.Linear(flat + m_dim + t_dim, 1024))          # :46-50
    _file(sd, "enc_fc.1", nn.BatchNorm1d(1024))
    _file(sd, "enc_fc.3", nn.Linear(1024, 2 * z_dim))
    _file(sd, "morph_predictor_shared.0", nn.Linear(t_dim, 64))           # :54-61
    _file(sd, "morph_predictor_shared.2", nn.Linear(64, 64))
    _file(sd, "morph_predictor_mu", nn.Linear(64, m_dim))
    _file(sd, "morph_predictor_logvar", nn.Linear(64, m_dim))
    _file(sd, "dec_fc.0", nn.Linear(m_dim + z_dim, 1024))                 # :64-69
    _file(sd, "dec_fc.1", nn.BatchNorm1d(1024))
    _file(sd, "dec_fc.3", nn.Linear(1024, flat))
    for ci, co in ((512, 512),) * 4 + ((512, 256), (256, 128), (128, 64), (64, 32), (32, 1)):   # the dead decoder :70-105 (RNG only)
        nn.ConvTranspose2d(ci, co, 4, 2, 1)
    cin = 512
    for i, cout in enumerate((512, 512, 256, 128, 64, 32)):               # :108-131
        _file(sd, f"dec_conv.{4 * i + 1}", nn.Conv2d(cin, cout, 3, 1, 1))
        _file(sd, f"dec_conv.{4 * i + 2}", nn.BatchNorm2d(cout))
        cin = cout
    _file(sd, "dec_conv.25", nn.Conv2d(32, 1, 3, 1, 1))                   # :133
    return sd


def _cvae(t_dim, z_dim):
    """ConditionalVAE (mnist_test/03_measurement_approach/cvae_models.py:13-49), construction order = RNG order."""
    sd = OrderedDict()
    _file(sd, "enc_conv.0", nn.Conv2d(1, 32, 4, 2, 1))
    _file(sd, "enc_conv.2", nn.Conv2d(32, 64, 4, 2, 1))
    _file(sd, "enc_conv.4", nn.Conv2d(64, 64, 4, 2, 1))
    _file(sd, "enc_fc_mu", nn.Linear(576 + t_dim, z_dim))
    _file(sd, "enc_fc_logvar", nn.Linear(576 + t_dim, z_dim))
    _file(sd, "dec_fc", nn.Linear(z_dim + t_dim, 64 * 7 * 7))
    _file(sd, "dec_conv.0", nn.ConvTranspose2d(64, 32, 4, 2, 1))
    _file(sd, "dec_conv.2", nn.ConvTranspose2d(32, 1, 4, 2, 1))
    return sd


def _disc(z_dim, t_dim):
    sd = OrderedDict()
    _file(sd, "net.0", nn.Linear(z_dim, 64))
    _file(sd, "net.2", nn.Linear(64, 64))
    _file(sd, "net.4", nn.Linear(64, t_dim))
    return sd


def init_state_dict(kind, seed=None, *, img_channels=1, m_dim=12, t_dim=None,
                    latent_dim=64, z_dim=10):
    """Return the freshly-initialised ``state_dict`` of reference model ``kind``.

    ``seed`` (if given) is applied with ``torch.manual_seed`` first, mirroring
    ``set_seed(42)`` before model construction (causal_cascade/main.py:28,44).
    """
    if seed is not None:
        torch.manual_seed(seed)
    if kind in ("bio2d", "bio3d"):
        return _bio(2 if kind == "bio2d" else 3, img_channels, m_dim,
                    19 if t_dim is None else t_dim, latent_dim)
    if kind in ("morph12", "morph12g"):
        return _morph12(m_dim, 10 if t_dim is None else t_dim, z_dim, kind == "morph12g")
    if kind == "disc":
        return _disc(z_dim, 10 if t_dim is None else t_dim)
    if kind == "cvae":
        return _cvae(10 if t_dim is None else t_dim, z_dim)
    if kind == "vessel2d":
        return _vessel2d(m_dim, 19 if t_dim is None else t_dim, 128)
    raise ValueError(f"unknown model kind {kind!r}; expected one of {MODEL_KINDS}")
