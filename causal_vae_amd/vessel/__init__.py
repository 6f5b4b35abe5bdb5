from .train import loss_function, total_loss, train_step, train_one_epoch, validate   # noqa: F401
from .models import CausalVesselVAE               # noqa: F401
from .config import CONFIG                         # noqa: F401
