from .models import CausalBioVAE, CausalBioVAE3D          # noqa: F401
from .train import loss_function, train_one_epoch, train_step   # noqa: F401
