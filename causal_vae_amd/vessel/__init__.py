from .train import loss_function, total_loss   # noqa: F401
