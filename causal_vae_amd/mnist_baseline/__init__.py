from .config import CONFIG, FEATURE_NAMES                                   # noqa: F401
from .models import CausalMorphVAE12, LatentDiscriminator                  # noqa: F401
from .train import train_step, train_model                                  # noqa: F401
