"""The vessel recipe's loss (vessel_analysis/01_train/train.py:18-60) on the HIP kernels, for volumes or images.

loss_function(recon_x, x, m_hat, m, mu, logvar, m_mu, m_logvar) -> (recon_loss, kld_loss, morph_loss, sparsity_loss)
total_loss(...) composes them as train_one_epoch does (:82): recon + BETA*kld + morph(*LAMBDA_MORPH in the k-fold
variant, train_kfold.py:71) + 0.3*sparsity.
"""
from .. import ops


def loss_function(recon_x, x, m_hat, m, mu, logvar, m_mu, m_logvar, sync_pos_weight=False, group=None):
    """sync_pos_weight (data parallelism only): pos_weight from the GLOBAL batch, as the reference's single process computes it (:30-36):
    (sum x, N) are all-reduced over `group`; False = each rank weights with its own micro-batch's pos_weight (the per-rank definition)."""
    recon_loss, sparsity_loss = ops.VesselRecon.apply(recon_x, x, sync_pos_weight, group)
    kld_loss = ops.KLD.apply(mu, logvar)
    morph_loss = ops.GaussNLL.apply(m, m_mu, m_logvar)
    return recon_loss, kld_loss, morph_loss, sparsity_loss


def total_loss(recon, kld, morph, sparsity, beta=0.5, lambda_morph=1.0):
    """recon + beta * kld + lambda_morph * morph + 0.3 * sparsity.  Device scalars go through one launch (ops.weighted_sum) instead of six scalar
    kernels forward and as many backward; anything else (CPU tensors, Python numbers) takes the plain expression."""
    terms = (recon, kld, morph, sparsity)
    import torch
    if all(isinstance(v, torch.Tensor) and v.is_cuda and v.numel() == 1 for v in terms):
        return ops.weighted_sum(terms, (1.0, beta, lambda_morph, 0.3))
    return recon + beta * kld + lambda_morph * morph + 0.3 * sparsity


def train_step(vae, opt_vae, x, m, t, eps=None, beta=0.5, lambda_morph=1.0, max_norm=5.0):
    """One iteration of train_one_epoch's body (vessel_analysis/01_train/train.py:70-86): zero_grad -> 6-tuple forward -> vessel loss ->
    backward -> clip_grad_norm_(5.0) -> step.  With a FusedAdam the norm and the clip coefficient are computed and applied on the device
    (optim.clip_grad_norm_: one multi-tensor norm launch, the coefficient multiplied into the gradients by the Adam launch itself — the update
    torch's in-place clip + Adam gives, without ~2 launches per parameter tensor and without a host sync); any other optimizer gets torch's in-place clip.
    Returns (loss, recon, kld, morph) as 0-dim device tensors."""
    from ..optim import FusedAdam, clip_grad_norm_
    opt_vae.zero_grad(set_to_none=True)
    out = vae(x, m, t) if eps is None else vae(x, m, t, eps=eps)
    from .. import ops
    with ops.zero_pool(8, x):                                # the loss terms' zeroed scalar outputs: one fill launch instead of one each
        recon, kld, morph, sparsity = loss_function(out[0], x, out[1], m, out[2], out[3], out[4], out[5])
        loss = total_loss(recon, kld, morph, sparsity, beta=beta, lambda_morph=lambda_morph)
    loss.backward()
    params = [p for p in vae.parameters() if p.grad is not None]
    if isinstance(opt_vae, FusedAdam):
        _, coef = clip_grad_norm_(params, max_norm, scale_grads=False)
        opt_vae.step(grad_scale=coef)
    else:
        import torch
        torch.nn.utils.clip_grad_norm_(params, max_norm=max_norm)
        opt_vae.step()
    return loss.detach(), recon.detach(), kld.detach(), morph.detach()


def train_one_epoch(epoch, vae, train_loader, opt_vae, device="cuda", beta=0.5):
    """train_one_epoch(epoch, vae, train_loader, opt_vae) of the reference (:62-98): returns sum of batch losses / len(dataset); the
    per-batch `.item()` syncs are replaced by on-device accumulation."""
    vae.train()
    total = None
    for x, m, t in train_loader:
        x, m, t = x.to(device, non_blocking=True), m.to(device, non_blocking=True), t.to(device, non_blocking=True)
        loss = train_step(vae, opt_vae, x, m, t, beta=beta)[0]
        total = loss if total is None else total + loss
    return float(total.item()) / len(train_loader.dataset)


def validate(vae, val_loader, device="cuda", beta=0.5, verbose=False):
    """validate(vae, val_loader) of the reference (vessel_analysis/01_train/train.py:100-133): eval mode (BatchNorm on its running
    statistics), no_grad, the 6-tuple forward and the same loss composition as training (recon + BETA * kld + morph + 0.3 * sparsity);
    returns sum of batch losses / len(dataset).  The reference's four `.item()` syncs per batch become one per call (sums stay on the
    device); verbose=True prints its "[Val Breakdown]" line."""
    import torch
    was_training = vae.training
    vae.eval()
    tot = None
    with torch.no_grad():
        for x, m, t in val_loader:
            x, m, t = x.to(device, non_blocking=True), m.to(device, non_blocking=True), t.to(device, non_blocking=True)
            recon_x, m_hat, mu, logvar, m_mu, m_logvar = vae(x, m, t)
            recon, kld, morph, sparsity = loss_function(recon_x, x, m_hat, m, mu, logvar, m_mu, m_logvar)
            loss = total_loss(recon, kld, morph, sparsity, beta=beta)
            row = torch.stack([loss, recon, kld, morph])
            tot = row if tot is None else tot + row
    if was_training:
        vae.train()
    n = len(val_loader.dataset)
    if tot is None:
        return 0.0
    val_loss, avg_recon, avg_kld, avg_morph = (float(v) / n for v in tot.cpu())
    if verbose:
        print(f"   [Val Breakdown] Recon: {avg_recon:.1f} | KLD: {avg_kld:.1f} | Morph: {avg_morph:.1f}")
    return val_loss
