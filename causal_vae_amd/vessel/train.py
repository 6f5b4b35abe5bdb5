"""The vessel recipe's loss (vessel_analysis/01_train/train.py:18-60) on the HIP kernels, for volumes or images.

loss_function(recon_x, x, m_hat, m, mu, logvar, m_mu, m_logvar) -> (recon_loss, kld_loss, morph_loss, sparsity_loss)
total_loss(...) composes them as train_one_epoch does (:82): recon + BETA*kld + morph(*LAMBDA_MORPH in the k-fold
variant, train_kfold.py:71) + 0.3*sparsity.
"""
from .. import ops


def loss_function(recon_x, x, m_hat, m, mu, logvar, m_mu, m_logvar):
    recon_loss, sparsity_loss = ops.VesselRecon.apply(recon_x, x)
    kld_loss = ops.KLD.apply(mu, logvar)
    morph_loss = ops.GaussNLL.apply(m, m_mu, m_logvar)
    return recon_loss, kld_loss, morph_loss, sparsity_loss


def total_loss(recon, kld, morph, sparsity, beta=0.5, lambda_morph=1.0):
    return recon + beta * kld + lambda_morph * morph + 0.3 * sparsity
