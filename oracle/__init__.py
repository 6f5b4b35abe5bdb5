"""CPU oracle for the CausalVAE training step — TEST INFRASTRUCTURE, NOT PRODUCT.

Plain-PyTorch (CPU, fp32) restatement of the arithmetic on the reference's hot
path, written in a *functional* form over a
``state_dict`` so that it shares no code and no module classes with the product
package ``causal_vae_amd``.

Who may import this package: ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` — as the checker / the reported CPU
baseline, never as the thing measured or shipped.  ``causal_vae_amd`` never
imports it and has no CPU fallback.

Parity pin: every function here is checked against golden vectors captured by
importing the reference's own classes in the build container
(``tools/make_golden.py`` -> ``tests/golden/*.npz``, test
``tests/test_oracle_golden.py``).  The 3D model has no reference counterpart
(the reference is 2D only); it is the dimensional lift of
``causal_cascade/models.py`` defined in SURVEY.md §8(a), pinned through the 2D
goldens plus the slice-degeneracy test (3D with a single depth tap == 2D).
"""
from .params import init_state_dict, MODEL_KINDS  # noqa: F401
from .functional import (  # noqa: F401
    bio_vae_forward, morph_vae_forward, morph_vae6_forward, discriminator_forward,
    reparameterize, cascade_loss, vessel_loss, mnist_vae_losses, gaussian_nll, bio_decode, morph_decode, vessel_vae_forward,
)
from .steps import (  # noqa: F401
    adam_init, adam_update, cascade_train_step, cvae_train_step, mnist_adversarial_step, clip_grad_norm,
)
