"""Functional CPU restatement of the reference forward passes and losses.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Everything takes a plain
``state_dict`` (name -> tensor) plus inputs and an *injected* ``eps`` (the
reference draws it with ``torch.randn_like`` inside ``reparameterize``; a CPU
generator stream cannot be reproduced on the GPU, so parity runs inject it —
SURVEY.md §7 "RNG").
"""
import torch
import torch.nn.functional as F


def _conv(nd):
    return F.conv2d if nd == 2 else F.conv3d


def _convT(nd):
    return F.conv_transpose2d if nd == 2 else F.conv_transpose3d


def _lin(sd, name, x):
    return F.linear(x, sd[name + ".weight"], sd[name + ".bias"])


def reparameterize(mu, logvar, eps):
    """z = mu + eps * exp(logvar / 2)   (causal_cascade/models.py:65-68)."""
    return mu + eps * torch.exp(0.5 * logvar)


def _bn1d(sd, prefix, x, training, update_running=True, momentum=0.1, bn_eps=1e-5):
    """BatchNorm1d as torch.nn.BatchNorm1d applies it (causal_cascade/models.py:36).

    Training mode normalises with the *biased* batch variance and updates the
    running stats with the *unbiased* one; B == 1 raises like the reference.
    """
    w, b = sd[prefix + ".weight"], sd[prefix + ".bias"]
    if not training:
        rm, rv = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
        return (x - rm) / torch.sqrt(rv + bn_eps) * w + b
    if x.shape[0] <= 1:
        raise ValueError(f"Expected more than 1 value per channel when training, got input size {tuple(x.shape)}")
    # y = (x - mean_B) / sqrt(var_B(biased) + eps) * w + b; running stats <- (1-momentum) old + momentum new,
    # the running variance taking the unbiased estimate.  F.batch_norm is the aten kernel nn.BatchNorm1d calls;
    # its fused backward is used (a hand-composed one differs by ~1e-4 rel on the cancelling dL/dW terms).
    if update_running and (prefix + ".running_mean") in sd:
        with torch.no_grad():
            sd[prefix + ".num_batches_tracked"] += 1
        return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"], w, b, True, momentum, bn_eps)
    return F.batch_norm(x, None, None, w, b, True, momentum, bn_eps)


class _RoundBoth(torch.autograd.Function):
    """Value rounded to `dtype` on the way forward, gradient rounded to `dtype` on the way back: a tensor that is STORED in that dtype
    together with its gradient (the conv activations of the bf16 build)."""

    @staticmethod
    def forward(ctx, x, dtype):
        ctx.dtype = dtype
        return x.to(dtype).float()

    @staticmethod
    def backward(ctx, g):
        return g.to(ctx.dtype).float(), None


class _RoundFwd(torch.autograd.Function):
    """Value rounded to `dtype` forward, gradient untouched: an operand copied to that dtype whose master (and gradient) stay fp32
    (the conv weights of the bf16 build)."""

    @staticmethod
    def forward(ctx, x, dtype):
        return x.to(dtype).float()

    @staticmethod
    def backward(ctx, g):
        return g, None


def bio_vae_forward(sd, x, m, t, eps, *, nd=None, training=True, update_running=True,
                    keep_acts=False, conv_dtype=None):
    """CausalBioVAE.forward (causal_cascade/models.py:70-89) for nd=2, or its 3D lift
     for nd=3.  ``t`` is an int64 class index.  Returns a dict with
    recon_x, m_hat, mu, logvar, z (+ per-layer activations when keep_acts).

    conv_dtype=torch.bfloat16 is NOT a reference mode: it restates, in fp32 CPU arithmetic, where the MI355X build's bf16 configuration
    rounds — conv operands (activations, weights) and every stored conv activation / activation gradient are rounded to bf16, sums stay
    fp32, the dense middle and the losses stay fp32 — so a test can separate "bf16 storage" from "kernel error"."""
    nd = x.dim() - 2 if nd is None else nd
    conv, convT = _conv(nd), _convT(nd)
    q = (lambda v: _RoundBoth.apply(v, conv_dtype)) if conv_dtype is not None else (lambda v: v)
    qw = (lambda v: _RoundFwd.apply(v, conv_dtype)) if conv_dtype is not None else (lambda v: v)
    t_dim = sd["mechanism_net.0.weight"].shape[1]
    t_onehot = F.one_hot(t, num_classes=t_dim).float()                    # :71
    acts = {}
    h = qw(x)
    for i in range(4):                                                    # :12-16
        h = q(F.relu(conv(h, qw(sd[f"enc_conv.{2*i}.weight"]), sd[f"enc_conv.{2*i}.bias"], stride=2, padding=1)))
        acts[f"enc{i+1}"] = h
    pool = F.adaptive_avg_pool2d if nd == 2 else F.adaptive_avg_pool3d
    feat = pool(h, (4,) * nd).flatten(1)                                  # :18-19
    acts["x_feat"] = feat
    h = torch.cat([feat, m, t_onehot], dim=1)                             # :61
    h = F.relu(_lin(sd, "enc_fc.0", h))
    h = F.relu(_lin(sd, "enc_fc.2", h))
    mu, logvar = _lin(sd, "fc_mu", h), _lin(sd, "fc_logvar", h)           # :63
    z = reparameterize(mu, logvar, eps)                                   # :74
    g = _lin(sd, "mechanism_net.0", t_onehot)                             # :77
    g = F.relu(_bn1d(sd, "mechanism_net.1", g, training, update_running))
    g = F.relu(_lin(sd, "mechanism_net.3", g))
    m_hat = _lin(sd, "mechanism_net.5", g)
    d = _lin(sd, "dec_input", torch.cat([z, m_hat], dim=1))               # :80-81
    d = q(d.view(-1, 256, *([4] * nd)))                                   # :82
    acts["dec_in"] = d
    for i in range(4):                                                    # :51-54
        d = convT(d, qw(sd[f"dec_conv.{2*i}.weight"]), sd[f"dec_conv.{2*i}.bias"], stride=2, padding=1)
        if i < 3:
            d = F.relu(d)
        d = q(d)
        acts[f"dec{i+1}"] = d
    mode = "bilinear" if nd == 2 else "trilinear"
    recon_x = F.interpolate(d, size=x.shape[2:], mode=mode, align_corners=False)   # :87
    out = dict(recon_x=recon_x, m_hat=m_hat, mu=mu, logvar=logvar, z=z)
    if keep_acts:
        out["acts"] = acts
    return out


def bio_decode(sd, z, m_hat, size=None, nd=2):
    """Decoder half of CausalBioVAE.forward (causal_cascade/models.py:80-87): the per-row computation the reference's
    counterfactual loops repeat (vessel_analysis/04_generate_counterfactual/generate_counterfactual.py:97-99)."""
    convT = _convT(nd)
    d = _lin(sd, "dec_input", torch.cat([z, m_hat], dim=1)).view(-1, 256, *([4] * nd))
    for i in range(4):
        d = convT(d, sd[f"dec_conv.{2*i}.weight"], sd[f"dec_conv.{2*i}.bias"], stride=2, padding=1)
        if i < 3:
            d = F.relu(d)
    if size is not None and tuple(size) != tuple(d.shape[2:]):
        d = F.interpolate(d, size=tuple(size), mode="bilinear" if nd == 2 else "trilinear", align_corners=False)
    return d


def morph_decode(sd, m_hat, z):
    """dec_fc -> view(-1, 64, 7, 7) -> dec_conv (mnist_test/01_baseline_causal_vae/check_mnist_counterfactual.py:72-74)."""
    return _morph_decode(sd, m_hat, z)


def _morph_encode(sd, x, m, t):
    h = F.relu(F.conv2d(x, sd["enc_conv.0.weight"], sd["enc_conv.0.bias"], stride=2, padding=1))
    h = F.relu(F.conv2d(h, sd["enc_conv.2.weight"], sd["enc_conv.2.bias"], stride=2, padding=1))
    h = torch.cat([h.flatten(1), m, t], dim=1)
    h = _lin(sd, "enc_fc.2", F.relu(_lin(sd, "enc_fc.0", h)))
    return h.chunk(2, dim=1)


def _morph_decode(sd, cond, z):
    h = F.relu(_lin(sd, "dec_fc.0", torch.cat([cond, z], dim=1))).view(-1, 64, 7, 7)
    h = F.relu(F.conv_transpose2d(h, sd["dec_conv.0.weight"], sd["dec_conv.0.bias"], stride=2, padding=1))
    return torch.sigmoid(F.conv_transpose2d(h, sd["dec_conv.2.weight"], sd["dec_conv.2.bias"], stride=2, padding=1))


def morph_vae_forward(sd, x, m, t, eps):
    """CausalMorphVAE12.forward (mnist_test/01_baseline_causal_vae/models.py:55-72).
    ``t`` is a float one-hot.  Decoder input is cat[m_hat, z] (:67)."""
    mu, logvar = _morph_encode(sd, x, m, t)                               # :58-60
    z = reparameterize(mu, logvar, eps)                                   # :61
    m_hat = _lin(sd, "morph_predictor.2", F.relu(_lin(sd, "morph_predictor.0", t)))   # :64
    recon_x = _morph_decode(sd, m_hat, z)                                 # :67-70
    return dict(recon_x=recon_x, m_hat=m_hat, mu=mu, logvar=logvar, z=z)


def morph_vae6_forward(sd, x, m, t, eps):
    """Gaussian-head variant (mnist_test/06_model_experiment/models.py:62-85): the
    decoder is fed the *real* m (:79) and the 6-tuple adds m_mu, m_logvar."""
    mu, logvar = _morph_encode(sd, x, m, t)
    z = reparameterize(mu, logvar, eps)
    h = F.relu(_lin(sd, "morph_predictor_shared.0", t))                   # :69
    m_mu, m_logvar = _lin(sd, "morph_predictor_mu", h), _lin(sd, "morph_predictor_logvar", h)
    recon_x = _morph_decode(sd, m, z)
    return dict(recon_x=recon_x, m_hat=m_mu, mu=mu, logvar=logvar, z=z, m_mu=m_mu, m_logvar=m_logvar)


def _bn_train(sd, prefix, x, training, update_running=True, momentum=0.1, bn_eps=1e-5):
    """nn.BatchNorm1d / nn.BatchNorm2d through the aten kernel the reference's modules call (batch statistics in training mode,
    running statistics updated in place unless told otherwise)."""
    w, b = sd[prefix + ".weight"], sd[prefix + ".bias"]
    rm, rv = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    if training and update_running:
        with torch.no_grad():
            sd[prefix + ".num_batches_tracked"] += 1
        return F.batch_norm(x, rm, rv, w, b, True, momentum, bn_eps)
    if training:
        return F.batch_norm(x, None, None, w, b, True, momentum, bn_eps)
    return F.batch_norm(x, rm, rv, w, b, False, momentum, bn_eps)


def vessel_vae_forward(sd, x, m, t, eps, *, training=True, update_running=True):
    """CausalVesselVAE.forward (vessel_analysis/00_core/models.py:142-166): t is the one-hot [B, 19] float the vessel dataset
    yields (dataset.py), the decoder sees the real m (:161), logvar / mu / m_logvar are clamped (:148-149, :156)."""
    h = x
    for i in range(7):                                                    # :32-40
        h = F.conv2d(h, sd[f"enc_conv.{3*i}.weight"], sd[f"enc_conv.{3*i}.bias"], stride=2, padding=1)
        h = F.leaky_relu(_bn_train(sd, f"enc_conv.{3*i+1}", h, training, update_running), 0.2)
    h = torch.cat([h.flatten(1), m, t], dim=1)                            # :144-145
    h = F.leaky_relu(_bn_train(sd, "enc_fc.1", _lin(sd, "enc_fc.0", h), training, update_running), 0.2)
    mu, logvar = _lin(sd, "enc_fc.3", h).chunk(2, dim=1)                  # :146
    logvar = torch.clamp(logvar, min=-10, max=10)                         # :148
    mu = torch.clamp(mu, min=-100, max=100)                               # :149
    z = reparameterize(mu, logvar, eps)
    g = F.leaky_relu(_lin(sd, "morph_predictor_shared.0", t), 0.2)        # :153
    g = F.leaky_relu(_lin(sd, "morph_predictor_shared.2", g), 0.2)
    m_mu = _lin(sd, "morph_predictor_mu", g)
    m_logvar = torch.clamp(_lin(sd, "morph_predictor_logvar", g), min=-10, max=10)      # :156
    d = _lin(sd, "dec_fc.0", torch.cat([m, z], dim=1))                    # :161-162
    d = F.leaky_relu(_bn_train(sd, "dec_fc.1", d, training, update_running), 0.2)
    d = F.relu(_lin(sd, "dec_fc.3", d)).view(-1, 512, 6, 10)              # :163
    for i in range(6):                                                    # :108-131
        d = F.interpolate(d, scale_factor=2, mode="nearest")
        d = F.conv2d(d, sd[f"dec_conv.{4*i+1}.weight"], sd[f"dec_conv.{4*i+1}.bias"], stride=1, padding=1)
        d = F.relu(_bn_train(sd, f"dec_conv.{4*i+2}", d, training, update_running))
    d = F.interpolate(d, scale_factor=2, mode="nearest")
    recon_x = torch.sigmoid(F.conv2d(d, sd["dec_conv.25.weight"], sd["dec_conv.25.bias"], stride=1, padding=1))     # :133
    return dict(recon_x=recon_x, m_hat=m_mu, mu=mu, logvar=logvar, z=z, m_mu=m_mu, m_logvar=m_logvar)


def cvae_forward(sd, x, t, eps):
    """ConditionalVAE.forward (mnist_test/03_measurement_approach/cvae_models.py:51-85): q(z | x, t), p(x | z, t); t a float one-hot."""
    h = F.relu(F.conv2d(x, sd["enc_conv.0.weight"], sd["enc_conv.0.bias"], stride=2, padding=1))          # :23-30, 28 -> 14 -> 7 -> 3
    h = F.relu(F.conv2d(h, sd["enc_conv.2.weight"], sd["enc_conv.2.bias"], stride=2, padding=1))
    h = F.relu(F.conv2d(h, sd["enc_conv.4.weight"], sd["enc_conv.4.bias"], stride=2, padding=1))
    h_t = torch.cat([h.flatten(1), t], dim=1)                                                              # :54-58
    mu, logvar = _lin(sd, "enc_fc_mu", h_t), _lin(sd, "enc_fc_logvar", h_t)                               # :60-61
    z = reparameterize(mu, logvar, eps)                                                                    # :76-79
    recon_x = cvae_decode(sd, z, t)
    return dict(recon_x=recon_x, mu=mu, logvar=logvar, z=z)


def cvae_decode(sd, z, t):
    """ConditionalVAE.decode (cvae_models.py:64-74)."""
    h = _lin(sd, "dec_fc", torch.cat([z, t], dim=1)).view(-1, 64, 7, 7)
    h = F.relu(F.conv_transpose2d(h, sd["dec_conv.0.weight"], sd["dec_conv.0.bias"], stride=2, padding=1))
    return torch.sigmoid(F.conv_transpose2d(h, sd["dec_conv.2.weight"], sd["dec_conv.2.bias"], stride=2, padding=1))


def cvae_loss(recon_x, x, mu, logvar):
    """BCE-sum + KLD (cvae_train.py:37-45) -> (loss, recon, kld)."""
    recon = F.binary_cross_entropy(recon_x.reshape(-1, 784), x.reshape(-1, 784), reduction="sum")
    kld = kld_sum(mu, logvar)
    return recon + kld, recon, kld


def discriminator_forward(sd, z):
    """LatentDiscriminator.forward (mnist_test/01_baseline_causal_vae/models.py:102-111)."""
    h = F.leaky_relu(_lin(sd, "net.0", z), 0.2)
    h = F.leaky_relu(_lin(sd, "net.2", h), 0.2)
    return _lin(sd, "net.4", h)


def kld_sum(mu, logvar):
    """-0.5 * sum(1 + logvar - mu^2 - exp(logvar))   (causal_cascade/train.py:13)."""
    return -0.5 * torch.sum(1 + logvar - mu * mu - torch.exp(logvar))


def cascade_loss(recon_x, x, m_hat, m, mu, logvar, gamma=2000.0):
    """ELBO of causal_cascade/train.py:5-17 -> (loss, recon_loss, m_loss); kld is
    returned as a 4th element for inspection (the reference returns 3)."""
    recon = ((recon_x - x) ** 2).sum()                                    # :7
    m_loss = ((m_hat - m) ** 2).sum()                                     # :10
    kld = kld_sum(mu, logvar)                                             # :13
    return recon + gamma * m_loss + kld, recon, m_loss, kld               # :16


def gaussian_nll(m, m_mu, m_logvar):
    """0.5 * sum(logvar + (m - mu)^2 / exp(logvar))   (vessel train.py:56-58; 06/train.py:79)."""
    return 0.5 * torch.sum(m_logvar + (m - m_mu) ** 2 / torch.exp(m_logvar))


def vessel_loss(recon_x, x, m_hat, m, mu, logvar, m_mu, m_logvar):
    """vessel_analysis/01_train/train.py:18-60 -> (recon, kld, morph, sparsity)."""
    mse = (recon_x - x) ** 2                                              # :27
    with torch.no_grad():                                                 # :30-36
        pos_fraction = x.sum() / (x.numel() + 1e-6)
        pos_weight = torch.clamp((1.0 - pos_fraction) / (pos_fraction + 1e-6), min=1.0, max=50.0)
    recon = torch.sum(mse * (1.0 + (pos_weight - 1.0) * x))               # :38-41
    sparsity = torch.sum(torch.abs(recon_x) * (x < 0.1).float())          # :45-46
    return recon, kld_sum(mu, logvar), gaussian_nll(m, m_mu, m_logvar), sparsity


def mnist_vae_losses(recon_x, x, m_hat, m, mu, logvar, d_logits_fake, *, beta=1.0,
                     lambda_adv=10.0, t_dim=10):
    """VAE-side loss terms of mnist_test/01_baseline_causal_vae/train.py:70-87."""
    recon = F.binary_cross_entropy(recon_x.reshape(-1, 784), x.reshape(-1, 784), reduction="sum")  # :70
    kld = kld_sum(mu, logvar) * beta                                      # :71-72
    morph = ((m_hat - m) ** 2).sum() * 100                                # :73
    log_probs = F.log_softmax(d_logits_fake, dim=1)                       # :83
    uniform = torch.full_like(d_logits_fake, 1.0 / t_dim)                 # :82
    adv = F.kl_div(log_probs, uniform, reduction="batchmean") * lambda_adv * 100   # :85
    return recon + kld + morph + adv, recon, kld, morph, adv              # :87
