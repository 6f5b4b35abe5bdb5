"""The ConditionalVAE loop of mnist_test/03_measurement_approach/cvae_train.py:11-62 on the HIP kernels.

train_step(model, optimizer, x, t, eps=None): zero_grad -> forward -> BCE-sum + KLD (:37-45) -> backward -> step; returns the three
loss terms as 0-dim device tensors.  train_cvae(loader, epochs) mirrors train_cvae() for a caller-supplied loader of (x, m, t_onehot)
batches (m is ignored, :27-28)."""
from .. import ops
from ..optim import FusedAdam
from .config import CONFIG
from .models import ConditionalVAE


def loss_function(recon_x, x, mu, logvar):
    loss_recon = ops.bce_sum(recon_x.view(-1, 784), x.view(-1, 784))
    loss_kld = ops.KLD.apply(mu, logvar) * 1.0
    return loss_recon + loss_kld, loss_recon, loss_kld


def train_step(model, optimizer, x, t, eps=None):
    optimizer.zero_grad(set_to_none=True)
    recon_x, mu, logvar = model(x, t) if eps is None else model(x, t, eps=eps)
    loss, loss_recon, loss_kld = loss_function(recon_x, x, mu, logvar)
    ops.backward_from(loss)
    optimizer.step()
    return dict(loss=loss.detach(), recon=loss_recon.detach(), kld=loss_kld.detach())


def train_cvae(train_loader, epochs=None, device=None, verbose=True):
    device = CONFIG["DEVICE"] if device is None else device
    epochs = CONFIG["EPOCHS"] if epochs is None else epochs
    model = ConditionalVAE().to(device)
    optimizer = FusedAdam(model.parameters(), lr=CONFIG["LR"])
    n = len(train_loader.dataset)
    for epoch in range(epochs):
        model.train()
        tot = None
        for x, _, t in train_loader:
            r = train_step(model, optimizer, x.to(device), t.to(device))
            tot = r["loss"] if tot is None else tot + r["loss"]
        if verbose:
            print(f"Epoch {epoch + 1:02d} | Avg Loss: {float(tot.item()) / n:.1f}")
    return model
