from .config import CONFIG                                                   # noqa: F401
from .models import ConditionalVAE                                          # noqa: F401
from .train import loss_function, train_step, train_cvae                    # noqa: F401
