from .models import CausalMorphVAE12          # noqa: F401
from .train import vae_losses, train_step     # noqa: F401
