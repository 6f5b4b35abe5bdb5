"""causal_vae_amd — the CausalVAE training step on MI355X (gfx950): PyTorch-ROCm host code over hand-written HIP kernels
reached through the C ABI in include/cvae_hip.h.  Importing the package loads libcvae_hip.so and fails loudly if it has
not been built; nothing in here falls back to eager PyTorch or to the CPU oracle.

Sub-packages mirror the reference's directories:
  causal_cascade   CausalBioVAE (2D), CausalBioVAE3D (the volume lift), loss_function, train_one_epoch
  mnist_baseline   CONFIG, CausalMorphVAE12, LatentDiscriminator, train_step / train_model
  vessel           the vessel recipe's loss_function (pos-weighted MSE + sparsity + KLD + Gaussian NLL) and step
"""
from . import _lib                      # noqa: F401  (raises ImportError when the library is missing)
from .layers import set_compute_dtype   # noqa: F401
from .optim import FusedAdam, clip_grad_norm_   # noqa: F401

__version__ = "0.1.0"
