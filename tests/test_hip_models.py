"""GPU parity tests of the whole drop-in models and train steps against (a) the golden vectors captured from the
reference classes and (b) the CPU oracle on the same seeded inputs.  Tolerances: fp32 path rtol 1e-4 on activations and
gradients, 1e-4 relative on the ELBO (the BASELINE target); the bf16 path states its own.
"""
import copy
import io
import math

import numpy as np

import pytest
import torch

import oracle
from oracle import functional as ofn

pytestmark = pytest.mark.gpu
if torch.cuda.is_available():
    import causal_vae_amd
    from causal_vae_amd import FusedAdam
    from causal_vae_amd import ops as ops_mod
    from causal_vae_amd._lib import CvaeError
    from causal_vae_amd.causal_cascade import CausalBioVAE, CausalBioVAE3D, loss_function, train_one_epoch, train_step
    from causal_vae_amd.mnist_baseline import CausalMorphVAE12, LatentDiscriminator
    from causal_vae_amd.mnist_baseline import train_step as mnist_train_step

DEV = "cuda"
NOISE_KEY = "mechanism_net.0.bias"      # exactly-zero gradient (bias before train-mode BN): rounding noise only


def rel(a, b):
    return abs(float(a) - float(b)) / max(abs(float(b)), 1e-30)


def grad_close(got, ref, what, l2=2e-3, linf=2e-2):
    """Gradients of a ReLU network computed by two fp32 implementations differ in isolated elements (an activation within
    rounding distance of 0 flips its mask), so the check is per tensor: relative L2 error and max error against max |ref|."""
    got, ref = got.detach().cpu().double(), ref.double()
    nrm = float(ref.norm())
    assert float((got - ref).norm()) <= l2 * nrm + 1e-12, (what, "rel L2", float((got - ref).norm()) / max(nrm, 1e-30))
    assert float((got - ref).abs().max()) <= linf * float(ref.abs().max()) + 1e-12, (what, "max err", float((got - ref).abs().max()), float(ref.abs().max()))


def adam_close(p, ref, what, lr=1e-3):
    """Weights after ONE Adam step.  The first step moves each weight by lr*g/(|g|+1e-8): where a gradient is ~0 its
    rounding noise decides between -lr, 0 and +lr, so isolated elements may differ by up to 2*lr; everything else by ~1e-7."""
    d = (p.detach().cpu().float() - ref.float()).abs()
    assert float(d.max()) <= 2.0 * lr + 1e-6, (what, float(d.max()))
    frac = float((d > 0.21 * lr).float().mean())
    assert frac < max(2e-3, 2.5 / d.numel()), (what, "fraction of weights off by more than 0.21*lr", frac)


@pytest.mark.parametrize("case", ["bio2d_b4_64x96", "bio2d_b2_64x64", "bio2d_b3_128x160"])
def test_bio2d_matches_reference_golden(golden, case):
    """CausalBioVAE (fp32 MFMA path) vs tensors produced by the reference class itself: forward, ELBO, grads, Adam."""
    g = golden(case)
    torch.manual_seed(42)
    model = CausalBioVAE().to(DEV)
    for k, v in model.state_dict().items():
        g.check("sd0", k, v, rtol=0, atol=0)                     # identical initialisation for the reference's seed
    model.train()
    opt = FusedAdam(model.parameters(), lr=1e-3)
    x, m, t, eps = (g.t("in/" + k).to(DEV) for k in ("x", "m", "t", "eps"))
    opt.zero_grad()
    recon_x, m_hat, mu, logvar = model(x, m, t, eps=eps)
    for k, v in dict(recon_x=recon_x, m_hat=m_hat, mu=mu, logvar=logvar).items():
        g.check("out", k, v, rtol=1e-4, atol=2e-5)
    loss, l_recon, l_m = loss_function(recon_x, x, m_hat, m, mu, logvar)
    assert rel(loss, g.t("loss/loss")) < 1e-4 and rel(l_recon, g.t("loss/recon")) < 1e-4 and rel(l_m, g.t("loss/m_loss")) < 1e-4
    loss.backward()
    for k, p in model.named_parameters():
        if k == NOISE_KEY:
            assert p.grad.abs().max() < 0.05
            continue
        scale = float(g.z[f"grad/{k}#digest"][1]) / max(p.numel(), 1)          # mean |grad|
        g.check("grad", k, p.grad, rtol=2e-3, atol=2e-3 * scale + 1e-6)
    opt.step()
    for k, v in model.state_dict().items():
        if k == NOISE_KEY:
            assert (v.cpu() - g.t("sd1/" + k)).abs().max() <= 2.0e-3 + 1e-6
            continue
        g.check("sd1", k, v, rtol=1e-4, atol=2.1e-4 if "weight" in k or "bias" in k else 1e-5)
    model.eval()
    with torch.no_grad():                                        # causal_cascade/analyze.py:10-23 access pattern
        from causal_vae_amd import ops
        m_hat_eval = model.mechanism_net(ops.one_hot(t, 19))
    g.check("eval", "m_hat", m_hat_eval, rtol=1e-3, atol=1e-3)


def test_morph12_matches_reference_golden(golden):
    g = golden("morph12_b8")
    torch.manual_seed(42)
    vae, disc = CausalMorphVAE12().to(DEV), LatentDiscriminator().to(DEV)
    for k, v in vae.state_dict().items():
        g.check("sd0", k, v, rtol=0, atol=0)
    for k, v in disc.state_dict().items():
        g.check("sdd0", k, v, rtol=0, atol=0)
    vae.train(); disc.train()
    x, m, t = (g.t("in/" + k).to(DEV) for k in ("x", "m", "t"))
    recon_x, m_hat, mu, logvar = vae(x, m, t, eps=g.t("fwd/eps").to(DEV))
    for k, v in dict(recon_x=recon_x, m_hat=m_hat, mu=mu, logvar=logvar).items():
        g.check("fwd", k, v, rtol=1e-4, atol=1e-5)
    opt_vae, opt_d = FusedAdam(vae.parameters(), lr=1e-3), FusedAdam(disc.parameters(), lr=1e-3)
    eps = tuple(g.t("step/" + k).to(DEV) for k in ("eps_d", "eps_vae", "eps_adv"))
    r = mnist_train_step(vae, disc, opt_vae, opt_d, x, m, t, eps=eps)
    for k in ("loss_d", "loss", "recon", "kld", "morph", "adv"):
        assert rel(r[k], g.t("step/" + k)) < 1e-4, (k, float(r[k]), float(g.t("step/" + k)))
    for k, v in vae.state_dict().items():
        if g.has("sd1/" + k):
            adam_close(v, g.t("sd1/" + k), k)
    for k, v in disc.state_dict().items():
        adam_close(v, g.t("sdd1/" + k), k)


def test_conditional_vae_matches_reference_golden(golden):
    """ConditionalVAE (mnist_test/03_measurement_approach/cvae_models.py + cvae_train.py:31-47): init, forward, BCE + KLD, every
    gradient and one Adam step against the vectors captured from the reference; then the bf16 conv path against the fp32 one."""
    from causal_vae_amd.mnist_cvae import ConditionalVAE, loss_function as cvae_loss, train_step as cvae_step
    g = golden("mnist_cvae_b8")
    torch.manual_seed(42)
    vae = ConditionalVAE().to(DEV).train()
    for k, v in vae.state_dict().items():
        g.check("sd0", k, v, rtol=0, atol=0)
    x, t, eps = g.t("in/x").to(DEV), g.t("in/t").to(DEV), g.t("fwd/eps").to(DEV)
    recon_x, mu, logvar = vae(x, t, eps=eps)
    for k, v in dict(recon_x=recon_x, mu=mu, logvar=logvar).items():
        g.check("fwd", k, v, rtol=1e-4, atol=1e-5)
    loss, recon, kld = cvae_loss(recon_x, x, mu, logvar)
    for k, v in dict(loss=loss, recon=recon, kld=kld).items():
        assert rel(v, g.t("fwd/" + k)) < 1e-5, (k, float(v), float(g.t("fwd/" + k)))
    loss.backward()
    for k, p in vae.named_parameters():
        if g.has("grad/" + k):
            grad_close(p.grad, g.t("grad/" + k), k)
        else:                                                 # large tensors are pinned by their digest (sums, head, tail)
            g.check("grad", k, p.grad, rtol=2e-3, atol=2e-4)
    vae.zero_grad()
    opt = FusedAdam(vae.parameters(), lr=1e-3)
    r = cvae_step(vae, opt, x, t, eps=eps)
    assert rel(r["loss"], g.t("fwd/loss")) < 1e-5
    for k, v in vae.state_dict().items():
        if g.has("sd1/" + k):
            adam_close(v, g.t("sd1/" + k), k)
    torch.manual_seed(42)
    vb = ConditionalVAE().to(DEV).train().set_compute_dtype(torch.bfloat16)
    rb, mub, lvb = vb(x, t, eps=eps)
    lb = cvae_loss(rb, x, mub, lvb)[0]
    assert rel(lb, g.t("fwd/loss")) < 2e-3 and float((rb - g.t("fwd/recon_x").to(DEV)).abs().max()) < 2e-2


def _oracle_step(kind, x, m, t, eps, nd):
    sd = oracle.init_state_dict(kind, seed=42)
    sd0 = {k: v.clone() for k, v in sd.items()}
    st = oracle.cascade_train_step(sd, x, m, t, eps, nd=nd)
    return sd0, sd, st


@pytest.mark.parametrize("B,size", [(2, 32), (2, 64), (2, 48), (2, 40), (16, 64), (2, 128)])   # (16, 64): BASELINE.json configs[2]; 40: 5 -> 2 floors; 128: the bench volume
def test_bio3d_fp32_matches_oracle(B, size):
    """3D lift vs the CPU oracle: forward, ELBO (<= 1e-4 rel), gradients, one Adam step."""
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(B, 1, size, size, size, generator=g)
    m = torch.rand(B, 12, generator=g)
    t = torch.randint(0, 19, (B,), generator=g)
    eps = torch.randn(B, 64, generator=g)
    sd0, sd1, st = _oracle_step("bio3d", x, m, t, eps, 3)
    torch.manual_seed(42)
    model = CausalBioVAE3D().to(DEV).train()
    assert all(torch.equal(model.state_dict()[k].cpu(), sd0[k]) for k in sd0)
    opt = FusedAdam(model.parameters(), lr=1e-3)
    loss, l_recon, l_m = train_step(model, opt, x.to(DEV), m.to(DEV), t.to(DEV), eps=eps.to(DEV))
    assert rel(loss, st["loss"]) < 1e-4, (float(loss), float(st["loss"]))
    assert rel(l_recon, st["recon"]) < 1e-4 and rel(l_m, st["m_loss"]) < 1e-4
    for k, p in model.named_parameters():
        if k == NOISE_KEY:
            continue
        grad_close(p.grad, st["grads"][k], k)
        adam_close(p, sd1[k], k)
    for k in ("mechanism_net.1.running_mean", "mechanism_net.1.running_var", "mechanism_net.1.num_batches_tracked"):
        torch.testing.assert_close(model.state_dict()[k].cpu(), sd1[k], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("H,W", [(100, 100), (72, 90)])
def test_bio2d_odd_intermediate_extents_match_oracle(H, W):
    """2D model on sizes that are not multiples of 16: the encoder floors (100 -> 50 -> 25 -> 12 -> 6), the decoder emits 64 x 64 and the
    bilinear resize is a non-integer ratio; forward, ELBO, gradients and one Adam step against the CPU oracle (layer-by-layer path)."""
    g = torch.Generator().manual_seed(77)
    B = 3
    x, m = torch.randn(B, 1, H, W, generator=g), torch.rand(B, 12, generator=g)
    t, eps = torch.randint(0, 19, (B,), generator=g), torch.randn(B, 64, generator=g)
    sd0, sd1, st = _oracle_step("bio2d", x, m, t, eps, 2)
    torch.manual_seed(42)
    model = CausalBioVAE().to(DEV).train()
    assert all(torch.equal(model.state_dict()[k].cpu(), sd0[k]) for k in sd0)
    opt = FusedAdam(model.parameters(), lr=1e-3)
    loss, l_recon, l_m = train_step(model, opt, x.to(DEV), m.to(DEV), t.to(DEV), eps=eps.to(DEV))
    assert rel(loss, st["loss"]) < 1e-4 and rel(l_recon, st["recon"]) < 1e-4 and rel(l_m, st["m_loss"]) < 1e-4
    for k, p in model.named_parameters():
        if k != NOISE_KEY:
            grad_close(p.grad, st["grads"][k], k)
            adam_close(p, sd1[k], k)


# Stated bf16 bounds (conv GEMM inputs rounded to bf16 = 2^-9 relative, fp32 accumulate, fp32 heads / losses / master weights): static CAPS per layer.
# The operative bound in test_bio3d_bf16_elbo_vs_fp32_oracle is derived per batch from the oracle alone (fp32 oracle vs bf16-rounding oracle, both CPU).
# rel-L2 = ||got - ref||_2 / ||ref||_2 against the fp32 CPU oracle.
BF16_RECON_REL_L2 = 2e-2                 # the decoder output resized to the input grid
BF16_GRAD_REL_L2 = {                     # per-layer weight gradients: the error grows along the bf16 backward chain (decoder -> bottleneck -> encoder).
    # Large in relative terms because at the initial weights on N(0,1) volumes d(recon) ~ -2x is noise, and every weight gradient is a
    # heavily cancelling sum of it; the SAME numbers come out of an fp32 computation whose conv operands are rounded to bf16
    # (test_bio3d_bf16_matches_bf16_rounding_oracle bounds the kernels against that at ~1e-2): bf16 storage, not the kernels.
    "dec_conv.6.weight": 0.10, "dec_conv.4.weight": 0.10, "dec_conv.2.weight": 0.10, "dec_conv.0.weight": 0.10, "dec_input.weight": 0.10,
    "fc_mu.weight": 0.10, "fc_logvar.weight": 0.09, "enc_fc.2.weight": 0.11, "enc_fc.0.weight": 0.15,
    "enc_conv.6.weight": 0.15, "enc_conv.4.weight": 0.17, "enc_conv.2.weight": 0.20, "enc_conv.0.weight": 0.22,
    "mechanism_net.0.weight": 0.02, "mechanism_net.3.weight": 0.02, "mechanism_net.5.weight": 0.02,
}


@pytest.mark.parametrize("B,size", [(2, 64), (4, 128)])          # (4, 128): the bench workload itself (BASELINE.json north_star)
def test_bio3d_bf16_elbo_vs_fp32_oracle(B, size):
    """bf16 conv path (fp32 accumulate, fp32 heads and losses) against the fp32 CPU oracle on the same batch: the ELBO within the BASELINE
    target of 1e-4 relative, each of its three terms, the reconstruction itself (relative L2), and every weight gradient by relative L2 (not
    only its direction) within the stated bf16 bounds above."""
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(B, 1, size, size, size, generator=g)
    m, t, eps = torch.rand(B, 12, generator=g), torch.randint(0, 19, (B,), generator=g), torch.randn(B, 64, generator=g)
    _, _, st = _oracle_step("bio3d", x, m, t, eps, 3)
    torch.manual_seed(42)
    model = CausalBioVAE3D().to(DEV).train().set_compute_dtype(torch.bfloat16)
    xd, md, td, ed = x.to(DEV), m.to(DEV), t.to(DEV), eps.to(DEV)
    # the reconstruction and the latent heads (layer-by-layer bf16 path under no_grad; BatchNorm buffers restored afterwards)
    bn = model.mechanism_net[1]
    saved = (bn.running_mean.clone(), bn.running_var.clone(), bn.num_batches_tracked.clone())
    with torch.no_grad():
        recon_x, m_hat, mu, logvar = model(xd, md, td, eps=ed)
        kld = ops_mod.KLD.apply(mu, logvar)
    bn.running_mean.copy_(saved[0]); bn.running_var.copy_(saved[1]); bn.num_batches_tracked.copy_(saved[2])
    ref = st["outputs"]["recon_x"]
    rl2 = float((recon_x.cpu() - ref).norm() / ref.norm())
    print(f"bf16 recon_x rel-L2 vs fp32 oracle: {rl2:.3e}; kld rel err {rel(kld, st['kld']):.3e}")
    assert rl2 < BF16_RECON_REL_L2
    assert rel(kld, st["kld"]) < 1e-3                                        # the heads are fp32: only their bf16-conv inputs differ
    torch.testing.assert_close(m_hat.cpu(), st["outputs"]["m_hat"], rtol=1e-5, atol=1e-6)     # mechanism_net never sees a bf16 tensor
    opt = FusedAdam(model.parameters(), lr=1e-3)
    loss, l_recon, l_m = train_step(model, opt, xd, md, td, eps=ed)                            # the fused fast path, as bench.py runs it
    err = rel(loss, st["loss"])
    print(f"bf16 ELBO rel err vs fp32 oracle: {err:.3e} (loss {float(loss):.4f} vs {float(st['loss']):.4f}); recon {rel(l_recon, st['recon']):.3e}, "
          f"m_loss {rel(l_m, st['m_loss']):.3e}")
    assert err < 1e-4                                                       # BASELINE.md target
    assert rel(l_recon, st["recon"]) < 1e-4 and rel(l_m, st["m_loss"]) < 1e-5
    # What may bf16 cost?  Not what the HIP build happens to produce: the gap between the fp32 oracle and the SAME oracle with its conv operands,
    # activations and activation gradients rounded to bf16 (pure CPU fp32 arithmetic, conv_dtype=bfloat16) is what bf16 STORAGE costs on this batch.
    # The build's own distance from the fp32 oracle must stay within 1.5x that gap (+ 0.01: two bf16 computations that sum in different orders also
    # differ from each other, test_bio3d_bf16_matches_bf16_rounding_oracle) — and under the static table above, which only caps it.
    sd_r = oracle.init_state_dict("bio3d", seed=42)
    st_r = oracle.cascade_train_step(sd_r, x, m, t, eps, nd=3, apply_update=False, conv_dtype=torch.bfloat16)
    for k, p in model.named_parameters():
        if k.endswith(".weight") and p.dim() > 1:
            a, b = p.grad.cpu().double().flatten(), st["grads"][k].double().flatten()
            r_ = st_r["grads"][k].double().flatten()
            cos = float(torch.dot(a, b) / (a.norm() * b.norm()))
            l2 = float((a - b).norm() / b.norm())
            gap = float((r_ - b).norm() / b.norm())
            print(f"  {k}: rel-L2 {l2:.4f} (bf16-rounding oracle vs fp32 oracle: {gap:.4f})  cos {cos:.5f}")
            assert l2 < 1.5 * gap + 0.01, (k, l2, gap)
            assert l2 < BF16_GRAD_REL_L2[k], (k, l2)
            assert cos > (0.98 if k == "enc_conv.0.weight" else 0.99), (k, cos)


@pytest.mark.parametrize("B,size", [(2, 64), (4, 128)], ids=["2x64", "4x128"])          # (4, 128): the bench workload itself
def test_bio3d_bf16_matches_bf16_rounding_oracle(B, size):
    """The same bf16 step against the oracle run with conv_dtype=bfloat16 (fp32 CPU arithmetic that rounds exactly where the bf16 build stores
    bf16: conv operands, conv activations and their gradients).  Against THAT every weight gradient agrees 3-5x tighter than against the pure-fp32
    oracle (bound 5e-2 here, 0.09-0.22 there): the several-percent gaps of test_bio3d_bf16_elbo_vs_fp32_oracle are bf16 storage on noise-like
    gradients, not kernel error.  It is not tighter still because two bf16 implementations that sum in different orders flip different ReLU
    masks and diverge chaotically to 2-3 % in the first-layer gradients; kernel by kernel they match the rounded reference at 1e-5
    (tests/test_hip_ops.py::test_bench_shape_launches_match_rounded_operand_reference)."""
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(B, 1, size, size, size, generator=g)
    m, t, eps = torch.rand(B, 12, generator=g), torch.randint(0, 19, (B,), generator=g), torch.randn(B, 64, generator=g)
    sd = oracle.init_state_dict("bio3d", seed=42)
    st = oracle.cascade_train_step(sd, x, m, t, eps, nd=3, apply_update=False, conv_dtype=torch.bfloat16)
    torch.manual_seed(42)
    model = CausalBioVAE3D().to(DEV).train().set_compute_dtype(torch.bfloat16)
    opt = FusedAdam(model.parameters(), lr=1e-3)
    loss, l_recon, l_m = train_step(model, opt, x.to(DEV), m.to(DEV), t.to(DEV), eps=eps.to(DEV))
    print(f"bf16 HIP vs bf16-rounding oracle: ELBO rel err {rel(loss, st['loss']):.3e}")
    assert rel(loss, st["loss"]) < 2e-6 and rel(l_recon, st["recon"]) < 2e-6
    worst = {}
    for k, p in model.named_parameters():
        if k == NOISE_KEY:
            continue
        a, b = p.grad.cpu().double().flatten(), st["grads"][k].double().flatten()
        worst[k] = float((a - b).norm() / b.norm())
        print(f"  {k}: rel-L2 vs bf16-rounding oracle {worst[k]:.2e}")
    bad = {k: v for k, v in worst.items() if v >= 5e-2}
    assert not bad, bad


def test_consumer_access_patterns_and_checkpoint_interchange():
    """What the reference's analysis scripts do to a trained model (SURVEY.md §8(b) census), on synthetic inputs."""
    torch.manual_seed(0)
    vae = CausalMorphVAE12().to(DEV).eval()
    x, m = torch.rand(5, 1, 28, 28, device=DEV), torch.rand(5, 12, device=DEV)
    t = torch.eye(10, device=DEV)[torch.tensor([1, 2, 3, 4, 5])]
    with torch.no_grad():
        x_feat = vae.enc_conv(x)                                             # visualize.py:26-28
        mu, logvar = vae.enc_fc(torch.cat([x_feat, m, t], dim=1)).chunk(2, dim=1)
        z = vae.reparameterize(mu, logvar)                                   # 2-argument call, device RNG
        m_hat = vae.morph_predictor(t)                                       # visualize.py:34
        h = vae.dec_fc(torch.cat([m_hat, z], dim=1)).view(-1, 64, 7, 7)      # visualize.py:87-89
        recon = vae.dec_conv(h)
    assert x_feat.shape == (5, 3136) and recon.shape == (5, 1, 28, 28) and recon.dtype == torch.float32
    assert float(recon.min()) >= 0 and float(recon.max()) <= 1 and torch.isfinite(z).all()
    assert not hasattr(vae, "dec_adapter")                                   # analyze_vessel.py:93 discriminator
    # state_dict round trip through torch.save / load_state_dict, and interchange with a reference-keyed dict
    buf = io.BytesIO(); torch.save(vae.state_dict(), buf); buf.seek(0)
    vae2 = CausalMorphVAE12().to(DEV)
    vae2.load_state_dict(torch.load(buf, map_location=DEV))
    ref_sd = oracle.init_state_dict("morph12", seed=7)
    vae2.load_state_dict(ref_sd)                                             # strict: same keys, same shapes
    with torch.no_grad():
        out = vae2(x, m, t, eps=torch.zeros(5, 10, device=DEV))
    ref = ofn.morph_vae_forward(ref_sd, x.cpu(), m.cpu(), t.cpu(), torch.zeros(5, 10))
    torch.testing.assert_close(out[0].cpu(), ref["recon_x"], rtol=1e-4, atol=1e-5)
    # BatchNorm1d with a batch of one raises ValueError in train mode like the reference
    bio = CausalBioVAE().to(DEV).train()
    with pytest.raises(ValueError, match="more than 1 value per channel"):
        bio(torch.zeros(1, 1, 64, 64, device=DEV), torch.zeros(1, 12, device=DEV), torch.zeros(1, dtype=torch.long, device=DEV))
    with pytest.raises(CvaeError, match="CPU tensor"):
        bio(torch.zeros(2, 1, 64, 64), torch.zeros(2, 12), torch.zeros(2, dtype=torch.long))


def test_train_one_epoch_surface_and_stock_optimizer():
    """train_one_epoch(model, loader, optimizer, device) with a stock torch optimizer: grads arrive through autograd."""
    torch.manual_seed(42)
    model = CausalBioVAE().to(DEV)
    g = torch.Generator().manual_seed(5)
    data = [(torch.randn(1, 64, 64, generator=g), torch.rand(12, generator=g), torch.randint(0, 19, (), generator=g)) for _ in range(8)]
    loader = torch.utils.data.DataLoader(data, batch_size=4, shuffle=False)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)                      # the reference's optimizer (main.py:50)
    l1 = train_one_epoch(model, loader, opt, DEV)
    l2 = train_one_epoch(model, loader, opt, DEV)
    assert l1 > 0 and l2 < l1                                                # the ELBO goes down
    assert all(p.grad is not None for p in model.parameters())


def test_bio3d_128_smoke_bf16():
    """BASELINE config 4 shape on one GPU: B=4, 128^3, bf16: runs, finite, loss decreases over 3 steps."""
    torch.manual_seed(42)
    model = CausalBioVAE3D().to(DEV).train().set_compute_dtype(torch.bfloat16)
    opt = FusedAdam(model.parameters(), lr=1e-4)
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(4, 1, 128, 128, 128, generator=g).to(DEV)
    m, t = torch.rand(4, 12, generator=g).to(DEV), torch.randint(0, 19, (4,), generator=g).to(DEV)
    losses = [float(train_step(model, opt, x, m, t)[0]) for _ in range(3)]
    assert all(map(lambda v: v == v and v < 1e9, losses)) and losses[-1] < losses[0], losses


def test_bio3d_three_steps_track_the_oracle():
    """Three consecutive train steps (Adam state carried) vs the oracle: per-step ELBO within 1e-3 relative (Adam's
    sign-sensitive first steps amplify fp32 rounding noise, so later steps are held to a looser bound than step 1)."""
    g = torch.Generator().manual_seed(77)
    B, S = 2, 32
    sd = oracle.init_state_dict("bio3d", seed=42)
    torch.manual_seed(42)
    model = CausalBioVAE3D().to(DEV).train()
    opt, state = FusedAdam(model.parameters(), lr=1e-3), None
    for step in range(3):
        x, m = torch.randn(B, 1, S, S, S, generator=g), torch.rand(B, 12, generator=g)
        t, eps = torch.randint(0, 19, (B,), generator=g), torch.randn(B, 64, generator=g)
        st = oracle.cascade_train_step(sd, x, m, t, eps, adam_state=state, nd=3)
        state = st["adam_state"]
        loss, _, _ = train_step(model, opt, x.to(DEV), m.to(DEV), t.to(DEV), eps=eps.to(DEV))
        assert rel(loss, st["loss"]) < (1e-4 if step == 0 else 1e-3), (step, float(loss), float(st["loss"]))


def test_graphed_step_matches_eager_steps():
    """The captured HIP graph replays the same arithmetic as the eager step (device-side Adam step count and Philox call
    count advance under replay): 3 capture-warm-up steps + 3 replays == 6 eager steps on the same batch."""
    from causal_vae_amd.graph import GraphedTrainStep
    g = torch.Generator().manual_seed(11)
    x, m = torch.randn(2, 1, 32, 32, 32, generator=g).to(DEV), torch.rand(2, 12, generator=g).to(DEV)
    t = torch.randint(0, 19, (2,), generator=g).to(DEV)
    lf = lambda o, xx, mm: loss_function(o[0], xx, o[1], mm, o[2], o[3])
    ops_mod.EpsSource._instances = 0                      # same Philox subsequence for the arms compared below
    torch.manual_seed(42)
    m_e = CausalBioVAE3D().to(DEV).train()
    o_e = FusedAdam(m_e.parameters(), lr=1e-4, device_step=True)
    eager = [float(train_step(m_e, o_e, x, m, t)[0]) for _ in range(6)]
    ops_mod.EpsSource._instances = 0                      # same Philox subsequence for the arms compared below
    torch.manual_seed(42)
    m_g = CausalBioVAE3D().to(DEV).train()
    o_g = FusedAdam(m_g.parameters(), lr=1e-4, device_step=True)
    gs = GraphedTrainStep(m_g, o_g, (x, m, t), lf, warmup=3)
    graphed = [float(gs()[0]) for _ in range(3)]
    assert graphed == eager[3:], (eager, graphed)                   # bit for bit: no kernel on the path uses float atomics
    assert len(set(graphed)) == 3                                   # the replays really advance (weights + eps change)
    for (k, p), q in zip(m_e.named_parameters(), m_g.parameters()):
        assert torch.equal(p.detach(), q.detach()), k               # every weight identical after 6 steps, mechanism_net.0.bias included


@pytest.mark.parametrize("size,dtype,fused", [(64, torch.bfloat16, True), (32, torch.float32, True), (32, torch.float32, False), (48, torch.bfloat16, False)])
def test_two_runs_of_the_same_training_are_bit_identical(size, dtype, fused):
    """No kernel of the train step accumulates across workgroups with float atomics (slabs / partial rows added in index order instead), so
    6 eager steps from the same seed give the same losses and the same weights bit for bit — on the fused fast path and on the
    layer-by-layer path (linear split-K slabs, channel sums, loss reductions)."""
    g = torch.Generator().manual_seed(21)
    x, m = torch.randn(3, 1, size, size, size, generator=g).to(DEV), torch.rand(3, 12, generator=g).to(DEV)
    t = torch.randint(0, 19, (3,), generator=g).to(DEV)
    runs = []
    for _ in range(2):
        ops_mod.EpsSource._instances = 0                      # same Philox subsequence for the arms compared below
        torch.manual_seed(42)
        model = CausalBioVAE3D().to(DEV).train().set_compute_dtype(dtype)
        model.fuse_bottleneck = fused
        model.fuse_recon_loss = fused
        opt = FusedAdam(model.parameters(), lr=1e-4, device_step=True)
        losses = [tuple(float(v) for v in train_step(model, opt, x, m, t)) for _ in range(6)]
        runs.append((losses, {k: p.detach().clone() for k, p in model.named_parameters()}))
    assert runs[0][0] == runs[1][0], (runs[0][0], runs[1][0])
    for k in runs[0][1]:
        assert torch.equal(runs[0][1][k], runs[1][1][k]), k


@pytest.mark.parametrize("size", [64, 128])
def test_grouped_and_per_layer_weight_gradients_are_bit_identical(size):
    """ops.DEFER_WGRAD (conv weight gradients queued during the backward and run as ONE grouped launch) vs a launch per layer: a layer's slab
    count depends on that layer only, so the fp32 summation order — and every bit of every gradient, loss and weight after 4 steps — is the
    same whichever way the caller batches the launches (128^3: the bench workload's layer sizes, where the slab rule actually splits)."""
    g = torch.Generator().manual_seed(23)
    B = 2 if size == 128 else 3
    x, m = torch.randn(B, 1, size, size, size, generator=g).to(DEV), torch.rand(B, 12, generator=g).to(DEV)
    t = torch.randint(0, 19, (B,), generator=g).to(DEV)
    runs, old = [], ops_mod.DEFER_WGRAD
    try:
        for defer in (True, False):
            ops_mod.DEFER_WGRAD = defer
            ops_mod.EpsSource._instances = 0
            torch.manual_seed(42)
            model = CausalBioVAE3D().to(DEV).train().set_compute_dtype(torch.bfloat16)
            opt = FusedAdam(model.parameters(), lr=1e-4, device_step=True)
            losses = [tuple(float(v) for v in train_step(model, opt, x, m, t)) for _ in range(4)]
            runs.append((losses, {k: p.detach().clone() for k, p in model.named_parameters()}, {k: p.grad.detach().clone() for k, p in model.named_parameters()}))
    finally:
        ops_mod.DEFER_WGRAD = old
    assert runs[0][0] == runs[1][0], (runs[0][0], runs[1][0])
    for k in runs[0][1]:
        assert torch.equal(runs[0][2][k], runs[1][2][k]), ("grad", k)
        assert torch.equal(runs[0][1][k], runs[1][1][k]), k


@pytest.mark.parametrize("graphed", [False, True], ids=["eager", "graph"])
def test_training_state_checkpoint_resumes_the_same_run(graphed):
    """causal_vae_amd.checkpoint: model (reference keys only) + FusedAdam (moments and the DEVICE step count) + the Philox call count.  Three
    steps, save, three more == load into a fresh model / optimizer and run the same three: identical losses and weights, i.e. the resumed run
    neither replays the noise of steps 1-3 nor restarts Adam's bias correction (the step count advances on the device under graph replay)."""
    from causal_vae_amd import checkpoint
    from causal_vae_amd.graph import GraphedTrainStep
    g = torch.Generator().manual_seed(31)
    x, m = torch.randn(2, 1, 32, 32, 32, generator=g).to(DEV), torch.rand(2, 12, generator=g).to(DEV)
    t = torch.randint(0, 19, (2,), generator=g).to(DEV)

    def make():
        ops_mod.EpsSource._instances = 0
        torch.manual_seed(42)
        mdl = CausalBioVAE3D().to(DEV).train()
        return mdl, FusedAdam(mdl.parameters(), lr=1e-4, device_step=True)
    m_a, o_a = make()
    if graphed:
        gs = GraphedTrainStep(m_a, o_a, (x, m, t), None, warmup=3)          # the 3 capture warm-up steps ARE steps 1-3
        step_a = lambda: float(gs()[0])
    else:
        step_a = lambda: float(train_step(m_a, o_a, x, m, t)[0])
        [step_a() for _ in range(3)]
    buf = io.BytesIO()
    torch.save(checkpoint.training_state(m_a, o_a), buf)
    assert set(checkpoint.training_state(m_a, o_a)["model"]) == set(oracle.init_state_dict("bio3d", seed=42))     # reference keys, nothing extra
    cont = [step_a() for _ in range(3)]
    buf.seek(0)
    m_b, o_b = make()
    checkpoint.load_training_state(torch.load(buf, map_location=DEV), m_b, o_b)
    assert int(o_b._step_dev) == 3 and m_b._eps.state()["calls"] == 3
    resumed = [float(train_step(m_b, o_b, x, m, t)[0]) for _ in range(3)]
    assert resumed == cont, (cont, resumed)
    for (k, p), q in zip(m_a.named_parameters(), m_b.parameters()):
        assert torch.equal(p.detach(), q.detach()), k
    assert o_b.state_dict()["state"][0]["step"] == 6


def test_deferred_wgrad_survives_partial_and_failed_backwards():
    """The deferred conv weight gradients (ops.DEFER_WGRAD) must not depend on how the backward pass was asked for or how the previous one ended:
    (a) torch.autograd.grad(loss, [x]) with trainable conv weights — the engine drops their gradients, the end-of-backward launch must still write
    into memory it owns (the queue holds the storages) and the NEXT full backward must be exact; (b) a backward that raises never runs its flush
    callback — the following step must notice the stale queue and produce the same gradients as a fresh process; (c) a conv weight shared by two
    layers (the engine sums two gradients before any flush) gets both computed at once."""
    g = torch.Generator().manual_seed(77)
    x, m = torch.randn(2, 1, 32, 32, 32, generator=g).to(DEV), torch.rand(2, 12, generator=g).to(DEV)
    t, eps = torch.randint(0, 19, (2,), generator=g).to(DEV), torch.randn(2, 64, generator=g).to(DEV)

    def grads(model):
        for p in model.parameters():
            p.grad = None
        loss = model.forward_elbo(x, m, t, eps=eps)[0]
        ops_mod.backward_from(loss)
        return {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    torch.manual_seed(42)
    model = CausalBioVAE3D().to(DEV).train().set_compute_dtype(torch.bfloat16)
    ref = grads(model)
    # (a) input gradient only
    h = torch.randn(2, 8, 8, 8, 64, generator=g).to(DEV).to(torch.bfloat16).requires_grad_(True)
    w = model.enc_conv[4].weight
    y = ops_mod.ConvDown.apply(h, w, model.enc_conv[4].bias, 3, "relu", False, False)
    gx, = torch.autograd.grad(y.float().sum(), [h])
    assert bool(torch.isfinite(gx.float()).all()) and not ops_mod._WG_PENDING and not ops_mod._WG_QUEUED[0]
    again = grads(model)
    for k in ref:
        assert torch.equal(ref[k], again[k]), k
    # (b) a backward that raises after a conv layer has queued its gradient

    class Boom(torch.autograd.Function):
        @staticmethod
        def forward(ctx, v):
            return v.clone()

        @staticmethod
        def backward(ctx, gr):
            raise RuntimeError("boom")
    h2 = torch.randn(2, 8, 8, 8, 64, generator=g).to(DEV).to(torch.bfloat16).requires_grad_(True)
    y2 = ops_mod.ConvDown.apply(Boom.apply(h2), w, model.enc_conv[4].bias, 3, "relu", False, False)
    w.grad = None
    model.enc_conv[4].bias.grad = None
    with pytest.raises(RuntimeError, match="boom"):
        y2.float().sum().backward()
    assert ops_mod._WG_QUEUED[0] and ops_mod._WG_PENDING          # the engine dropped the callback: this is the stale state
    after = grads(model)
    assert not ops_mod._WG_PENDING and not ops_mod._WG_QUEUED[0]
    for k in ref:
        assert torch.equal(ref[k], after[k]), k
    # (c) one weight, two uses in one graph
    w.grad = None
    model.enc_conv[4].bias.grad = None
    ha, hb = (torch.randn(2, 8, 8, 8, 64, generator=g).to(DEV).to(torch.bfloat16) for _ in range(2))
    singles = []
    for hh in (ha, hb):
        w.grad = None
        ops_mod.ConvDown.apply(hh.clone().requires_grad_(True), w, None, 3, "relu", False, False).float().sum().backward()
        singles.append(w.grad.clone())
    w.grad = None
    ya = ops_mod.ConvDown.apply(ha.clone().requires_grad_(True), w, None, 3, "relu", False, False)
    yb = ops_mod.ConvDown.apply(hb.clone().requires_grad_(True), w, None, 3, "relu", False, False)
    (ya.float().sum() + yb.float().sum()).backward()
    assert torch.allclose(w.grad, singles[0] + singles[1], rtol=1e-5, atol=1e-4), float((w.grad - singles[0] - singles[1]).abs().max())


def test_fused_adam_load_state_dict_after_capture_reaches_the_replayed_update():
    """FusedAdam.load_state_dict restores the moments and the device step counter IN PLACE: a graph captured before the load keeps replaying
    against the same tensors, so a checkpoint loaded after capture is what the replayed update continues from."""
    from causal_vae_amd.graph import GraphedTrainStep
    g = torch.Generator().manual_seed(5)
    x, m = torch.randn(2, 1, 32, 32, 32, generator=g).to(DEV), torch.rand(2, 12, generator=g).to(DEV)
    t = torch.randint(0, 19, (2,), generator=g).to(DEV)
    ops_mod.EpsSource._instances = 0
    torch.manual_seed(42)
    model = CausalBioVAE3D().to(DEV).train()
    opt = FusedAdam(model.parameters(), lr=1e-4, device_step=True)
    gs = GraphedTrainStep(model, opt, (x, m, t), None, warmup=3)
    sd_model = {k: v.detach().clone() for k, v in model.state_dict().items()}
    sd_opt = {"state": {i: {k: (v.clone() if torch.is_tensor(v) else v) for k, v in st.items()} for i, st in opt.state_dict()["state"].items()},
              "param_groups": opt.state_dict()["param_groups"]}
    eps_state = model._eps.state()
    a = [float(gs()[0]) for _ in range(3)]
    model.load_state_dict(sd_model)                     # nn.Module.load_state_dict copies in place
    opt.load_state_dict(sd_opt)
    model._eps.load_state(eps_state)
    b = [float(gs()[0]) for _ in range(3)]
    assert a == b, (a, b)
    assert int(opt._step_dev) == 6


@pytest.mark.parametrize("B,size", [(2, 64), (4, 128)], ids=["2x64", "4x128"])
def test_fp8_forward_train_step_close_to_bf16_step(B, size):
    """BASELINE.json configs[4], train leg: the bf16 step with the forward products of enc_conv[2..6] / dec_conv[0..4] on fp8 (e4m3) operands
    (model.set_fp8_forward; scaled MFMA, delayed per-tensor scales, bf16 backward) against the bf16 step from the same weights on the same batch
    (causal_cascade/train.py:19-39).  Stated bounds: ELBO 1e-4 relative (the BASELINE target, here between the two precisions), reconstruction
    rel-L2 5e-2 (e4m3 carries 3 mantissa bits: 2^-4 per operand, averaged down by the K = 2048-long sums), per-layer gradient rel-L2 as listed —
    the gradients themselves are bf16 products of (fp8-forward) activations, so they inherit the forward's error through the ReLU masks."""
    g = torch.Generator().manual_seed(91)
    x, m = torch.randn(B, 1, size, size, size, generator=g).to(DEV), torch.rand(B, 12, generator=g).to(DEV)
    t, eps = torch.randint(0, 19, (B,), generator=g).to(DEV), torch.randn(B, 64, generator=g).to(DEV)
    res = {}
    for mode in ("bf16", "fp8"):
        torch.manual_seed(42)
        model = CausalBioVAE3D().to(DEV).train().set_compute_dtype(torch.bfloat16)
        if mode == "fp8":
            model.set_fp8_forward(True)
            model.forward_elbo(x, m, t, eps=eps)                     # step 0 calibrates the scales (bf16 arithmetic)
            assert model._fp8.calibrated
        for p in model.parameters():
            p.grad = None
        out_cl, m_hat, mu, logvar = model._forward_cl(x, m, t, eps)
        loss, recon, m_loss, _ = ops_mod.ElboUp2x.apply(out_cl, x, m_hat, m, mu, logvar, 2000.0) if ops_mod.ElboUp2x.supported(out_cl, x) else \
            ops_mod.Elbo.apply(model._resize_to(out_cl, x), x, m_hat, m, mu, logvar, 2000.0)
        ops_mod.backward_from(loss)
        res[mode] = dict(loss=float(loss), out=out_cl.detach().float().cpu(), mu=mu.detach().cpu(), grads={k: p.grad.detach().cpu() for k, p in model.named_parameters()})
    a, b = res["bf16"], res["fp8"]
    rel = lambda u, v: float((u - v).norm() / v.norm().clamp_min(1e-30))
    assert abs(b["loss"] - a["loss"]) <= 1e-4 * abs(a["loss"]), (a["loss"], b["loss"])
    assert rel(b["out"], a["out"]) < 5e-2, rel(b["out"], a["out"])
    assert rel(b["mu"], a["mu"]) < 5e-2, rel(b["mu"], a["mu"])
    worst = {}
    for k in a["grads"]:
        if k == "mechanism_net.0.bias":                              # true gradient 0 (feeds train-mode BatchNorm): rounding noise in both arms
            continue
        worst[k] = rel(b["grads"][k], a["grads"][k])
    print("fp8-forward vs bf16 gradient rel-L2:", {k: round(v, 4) for k, v in worst.items()})
    for k, v in worst.items():
        # Observed at random init (printed above), bounded with ~1.3x margin: the backward pass is the bf16 one applied to fp8-forward activations, so the
        # difference enters through every ReLU mask and activation the forward changed and grows towards the input end of the backward chain
        # (the last decoder layers 2e-3 .. 2e-2, the first decoder layer / bottleneck 0.25, the encoder 0.35 - 0.40); the same layers are the widest between
        # bf16 and the fp32 oracle (BF16_GRAD_REL_L2).  mechanism_net sees no conv activations at all.
        if k.startswith("mechanism_net"):
            bound = 1e-3
        elif k.startswith("enc_conv"):
            bound = 0.52
        elif k.startswith(("enc_fc", "fc_", "dec_input", "dec_conv.0")):
            bound = 0.36
        elif k.startswith("dec_conv.2"):
            bound = 0.2
        else:
            bound = 0.05
        assert v < bound, (k, v)
    cat = lambda gr: torch.cat([v.flatten() for k, v in gr.items() if k != "mechanism_net.0.bias"])
    ga, gb = cat(a["grads"]), cat(b["grads"])
    cos = float((ga * gb).sum() / (ga.norm() * gb.norm()))
    assert cos > 0.9, cos
    # delayed scaling keeps the step capturable and stable: three more steps change the scales by less than the headroom
    s0 = model._fp8.scales.scale.clone()
    opt = FusedAdam(model.parameters(), lr=1e-4)
    for _ in range(3):
        train_step(model, opt, x, m, t, eps=eps)
    ratio = (model._fp8.scales.scale / s0).cpu()
    assert float(ratio.max()) < 2.0 and float(ratio.min()) > 0.5, ratio


def test_fp8_forward_training_tracks_the_bf16_run_over_many_steps():
    """The fp8-forward step is for TRAINING: 15 optimizer steps (FusedAdam, lr 1e-4 — the reference's 1e-3, causal_cascade/main.py:50, diverges on unit-variance
    random volumes in every precision, the CPU oracle included) on a fixed batch from the same weights, with the same Philox noise, in both precisions.  The two loss trajectories stay within 1 % of each other at every step and fall by
    the same amount (the delayed scales follow the activations as they change); the HIP-graph replay of the fp8 step reproduces its eager trajectory bit for
    bit (scales, amax records and counters all live on the device)."""
    from causal_vae_amd.graph import GraphedTrainStep
    g = torch.Generator().manual_seed(93)
    B, size, steps = 2, 64, 15
    x, m = torch.randn(B, 1, size, size, size, generator=g).to(DEV), torch.rand(B, 12, generator=g).to(DEV)
    t = torch.randint(0, 19, (B,), generator=g).to(DEV)
    runs = {}
    for mode in ("bf16", "fp8", "fp8-graph"):
        ops_mod.EpsSource._instances = 0                      # the same Philox stream in every arm
        torch.manual_seed(42)
        model = CausalBioVAE3D().to(DEV).train().set_compute_dtype(torch.bfloat16)
        if mode != "bf16":
            model.set_fp8_forward(True)
        opt = FusedAdam(model.parameters(), lr=1e-4, device_step=True)
        if mode == "fp8-graph":
            gs = GraphedTrainStep(model, opt, (x, m, t), None, warmup=3)        # 3 eager steps (the first one calibrates), then replays
            losses = [None] * 3 + [float(gs()[0]) for _ in range(steps - 3)]
        else:
            losses = [float(train_step(model, opt, x, m, t)[0]) for _ in range(steps)]
        torch.cuda.synchronize()
        runs[mode] = losses
    a, b, c = runs["bf16"], runs["fp8"], runs["fp8-graph"]
    for i, (u, v) in enumerate(zip(a, b)):
        assert abs(u - v) <= 1e-2 * abs(u), (i, u, v)
    assert a[-1] < a[0] and b[-1] < b[0], (a[0], a[-1], b[0], b[-1])                    # both runs train
    assert abs((a[0] - a[-1]) - (b[0] - b[-1])) <= 0.05 * abs(a[0] - a[-1]), (a, b)
    assert c[3:] == b[3:], (b, c)                                                         # graph replay == eager, bit for bit


def test_elbo_in_launch_finish_equals_the_two_launch_form():
    """cvae_elbo_up2x_fwd with a ticket: the workgroup that arrives last sums the partials inside the forward launch (sc1 stores, drained, agent-scope ticket
    add; sc1 loads by the last arriver — MI355X_MICROARCH.md, valid forms) instead of a finish launch.  Same sums in the same order: all four outputs are
    bit-identical to the two-launch form, on every one of 40 back-to-back launches interleaved with a bandwidth-heavy kernel (uneven load: a stale or
    early read of a partial would show as a different bit pattern), the ticket is left at zero, and the step counter that rides along advances once per call."""
    g = torch.Generator().manual_seed(55)
    B = 4
    src = (torch.randn(B, 64, 64, 64, 1, generator=g) * 0.3).to(DEV).to(torch.bfloat16)
    x = torch.randn(B, 1, 128, 128, 128, generator=g).to(DEV)
    m_hat, m = torch.rand(B, 12, generator=g).to(DEV), torch.rand(B, 12, generator=g).to(DEV)
    mu, logvar = torch.randn(B, 64, generator=g).to(DEV), (torch.randn(B, 64, generator=g) * 0.1).to(DEV)
    junk = torch.empty(64 << 20, device=DEV)
    old = ops_mod.ElboUp2x.IN_LAUNCH_FINISH
    try:
        ops_mod.ElboUp2x.IN_LAUNCH_FINISH = False
        ref = [v.clone() for v in ops_mod.ElboUp2x.apply(src, x, m_hat, m, mu, logvar, 2000.0)]
        ops_mod.ElboUp2x.IN_LAUNCH_FINISH = True
        bump = torch.zeros((), dtype=torch.int32, device=DEV)
        for it in range(40):
            if it % 3:
                junk.mul_(1.0001)                                   # a streaming kernel right in front: the forward's workgroups start under load
            out = ops_mod.ElboUp2x.apply(src.requires_grad_(it % 2 == 0), x, m_hat, m, mu, logvar, 2000.0, bump)       # with and without the t1 side output
            for a, b in zip(out, ref):
                assert torch.equal(a, b), (it, float(a), float(b))
        assert int(bump) == 40
        assert int(ops_mod.ElboUp2x._tickets[src.device].abs().sum()) == 0
    finally:
        ops_mod.ElboUp2x.IN_LAUNCH_FINISH = old


@pytest.mark.parametrize("cls,shape,dtype,fp8", [("3d", (2, 1, 64, 64, 64), torch.bfloat16, False), ("3d", (2, 1, 32, 32, 32), torch.float32, False),
                                                 ("2d", (3, 1, 64, 96), torch.bfloat16, False), ("3d", (2, 1, 64, 64, 64), torch.bfloat16, True)])
def test_relu_mask_bits_give_the_gradients_of_tensor_masks(cls, shape, dtype, fp8):
    """ops.MASK_BITS: the forward launches leave their ReLU masks as bits and the backward-data launches read those instead of the saved activations
    (67 MB -> 4 MB for enc_conv[2]'s data gradient at 4 x 128^3).  Same masks, same arithmetic: loss and every gradient are bit-identical to the
    run with the activations as masks (causal_cascade/models.py:12-20, 50-55: every ReLU between two convolutions)."""
    Model = CausalBioVAE3D if cls == "3d" else CausalBioVAE
    g = torch.Generator().manual_seed(41)
    B = shape[0]
    x, m = torch.randn(*shape, generator=g).to(DEV), torch.rand(B, 12, generator=g).to(DEV)
    t, eps = torch.randint(0, 19, (B,), generator=g).to(DEV), torch.randn(B, 64, generator=g).to(DEV)
    runs = []
    old = ops_mod.MASK_BITS
    try:
        for use_bits in (False, True):
            ops_mod.MASK_BITS = use_bits
            torch.manual_seed(42)
            model = Model().to(DEV).train().set_compute_dtype(dtype)
            if fp8:
                model.set_fp8_forward(True)
                model.forward_elbo(x, m, t, eps=eps) if hasattr(model, "forward_elbo") else None     # calibration step
            for p in model.parameters():
                p.grad = None
            before = dict(ops_mod.BITS_STATS)
            recon, m_hat, mu, logvar = model(x, m, t, eps=eps)
            loss = ((recon - x) ** 2).sum() + 3.0 * ((m_hat - m) ** 2).sum() - 0.5 * torch.sum(1 + logvar - mu.pow(2) - logvar.exp())
            loss.backward()
            made, used = ops_mod.BITS_STATS["produced"] - before["produced"], ops_mod.BITS_STATS["consumed"] - before["consumed"]
            # every conv ReLU: enc2..enc4 and dec1..dec3 leave bits through the counted entry points (the image layer and the fp8 forward's products through their own), and the six data gradients above them read bits (bf16: the two single-channel ends included; an fp32 model keeps the tensor form at both)
            assert (made, used) == ((0, 0) if not use_bits else ((0 if fp8 else 6), 6 if dtype == torch.bfloat16 else 4)), (made, used)
            runs.append((float(loss), {k: p.grad.clone() for k, p in model.named_parameters()}))
    finally:
        ops_mod.MASK_BITS = old
    assert runs[0][0] == runs[1][0]
    for k in runs[0][1]:
        assert torch.equal(runs[0][1][k], runs[1][1][k]), k


def test_split_backward_capture_matches_eager_steps():
    """GraphedTrainStep(overlap_exchange=True): the backward captured in two graphs around the encoder output (the multi-GPU exchange
    overlap; no process group here, so no exchange happens) == the eager step: losses and weights after 3 + 3 steps, every gradient
    present and finite."""
    from causal_vae_amd.graph import GraphedTrainStep
    g = torch.Generator().manual_seed(12)
    x, m = torch.randn(2, 1, 64, 64, 64, generator=g).to(DEV), torch.rand(2, 12, generator=g).to(DEV)
    t = torch.randint(0, 19, (2,), generator=g).to(DEV)
    ops_mod.EpsSource._instances = 0                      # same Philox subsequence for the arms compared below
    torch.manual_seed(42)
    m_e = CausalBioVAE3D().to(DEV).train().set_compute_dtype(torch.bfloat16)
    o_e = FusedAdam(m_e.parameters(), lr=1e-4, device_step=True)
    eager = [float(train_step(m_e, o_e, x, m, t)[0]) for _ in range(6)]
    ops_mod.EpsSource._instances = 0                      # same Philox subsequence for the arms compared below
    torch.manual_seed(42)
    m_g = CausalBioVAE3D().to(DEV).train().set_compute_dtype(torch.bfloat16)
    o_g = FusedAdam(m_g.parameters(), lr=1e-4, device_step=True)
    gs = GraphedTrainStep(m_g, o_g, (x, m, t), None, warmup=3, overlap_exchange=True)
    graphed = [float(gs()[0]) for _ in range(3)]
    assert graphed == eager[3:], (eager, graphed)                   # same kernels: the losses are bit-identical
    assert len(set(graphed)) == 3
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in m_g.parameters())
    assert len(gs.red_a.params) + len(gs.red_b.params) == len(list(m_g.parameters())) and len(gs.red_b.params) == 8
    # the two captures flush their deferred conv weight gradients as two grouped launches (decoder, encoder) where the eager step has one: a layer's
    # slab count depends on that layer only (cvae_conv_wgrad_multi), so the summation order, and with it every bit, is the same
    for (k, p), q in zip(m_e.named_parameters(), m_g.parameters()):
        assert torch.equal(p.detach(), q.detach()), k


def test_gaussian_head_variant_matches_reference_golden(golden):
    """mnist_test/06_model_experiment CausalMorphVAE12 (6-tuple, decoder on the real m) vs tensors from the reference class."""
    from causal_vae_amd.mnist_gaussian import CausalMorphVAE12 as GaussVAE
    from causal_vae_amd import ops
    g = golden("morph12g_b8")
    torch.manual_seed(42)
    vae = GaussVAE().to(DEV).train()
    for k, v in vae.state_dict().items():
        g.check("sd0", k, v, rtol=0, atol=0)
    x, m, t = (g.t("in/" + k).to(DEV) for k in ("x", "m", "t"))
    out = vae(x, m, t, eps=g.t("fwd/eps").to(DEV))
    assert len(out) == 6
    for k, v in zip(("recon_x", "m_hat", "mu", "logvar", "m_mu", "m_logvar"), out):
        g.check("fwd", k, v, rtol=1e-4, atol=1e-5)
    nll = ops.GaussNLL.apply(m, out[4], out[5])
    assert rel(nll, g.t("fwd/nll")) < 1e-4
    with torch.no_grad():
        torch.testing.assert_close(vae.morph_predictor(t), out[4], rtol=1e-5, atol=1e-6)


def test_batched_counterfactual_decode_equals_per_value_loop():
    """One stacked decode == the reference's per-(feature, value) batch-1 decodes, checked against the oracle's decoder."""
    from causal_vae_amd.counterfactual import batched_counterfactual, sweep_inputs
    torch.manual_seed(42)
    model = CausalBioVAE3D().to(DEV).eval()
    sd = oracle.init_state_dict("bio3d", seed=42)
    g = torch.Generator().manual_seed(8)
    z, m = torch.randn(2, 64, generator=g), torch.rand(2, 12, generator=g)
    feats, vals = [0, 5, 11], [0.0, 0.25, 0.5, 0.75, 1.0]
    out = batched_counterfactual(model, z.to(DEV), m.to(DEV), feats, vals)            # native 64^3 decode
    assert out.shape == (2, 3, 5, 1, 64, 64, 64)
    for b, fi, vi in ((0, 0, 0), (1, 2, 4), (0, 1, 3)):                             # spot-check three single decodes
        m1 = m[b:b + 1].clone(); m1[0, feats[fi]] = vals[vi]
        ref = ofn.bio_decode(sd, z[b:b + 1], m1, nd=3)
        torch.testing.assert_close(out[b, fi, vi].cpu(), ref[0], rtol=1e-4, atol=1e-5)
    up = batched_counterfactual(model, z.to(DEV), m.to(DEV), [3], [0.1, 0.9], size=(96, 80, 72))
    ref = ofn.bio_decode(sd, z[1:2], torch.cat([m[1:2, :3], torch.tensor([[0.9]]), m[1:2, 4:]], 1), size=(96, 80, 72), nd=3)
    torch.testing.assert_close(up[1, 0, 1].cpu(), ref[0], rtol=1e-4, atol=1e-5)
    # MNIST: decode(m_hat, z)
    torch.manual_seed(42)
    vae = CausalMorphVAE12().to(DEV).eval()
    sdm = oracle.init_state_dict("morph12", seed=42)
    zz, mm = torch.randn(3, 10, generator=g), torch.rand(3, 12, generator=g)
    outm = batched_counterfactual(vae, zz.to(DEV), mm.to(DEV), [1, 2], [0.0, 1.0])
    z_rep, m_cf = sweep_inputs(zz, mm, [1, 2], [0.0, 1.0])
    torch.testing.assert_close(outm.reshape(-1, 1, 28, 28).cpu(), ofn.morph_decode(sdm, m_cf, z_rep), rtol=1e-4, atol=1e-5)


def test_fp8_and_bf16_sweep_decode_close_to_fp32_decode():
    """BASELINE.json configs[4] at its full size: 4 samples x 12 features x 5 values = 240 stacked rows decoded in ONE call, in bf16 and on the
    fp8 (e4m3) conv path, against the fp32 HIP decode of the same rows (itself checked against the oracle on three rows here and in
    test_batched_counterfactual_decode_equals_per_value_loop).  Measured (tools/fp8_probe.py, random-init weights): rel-L2 1.4e-3 bf16,
    3.0e-3 fp8 (3.0e-3 on the 180 rows the scales were NOT calibrated on); per-row SSE against a binary target differs by < 8e-5 relative.
    Bounds: 3x those.  The fp8 scales are static per tensor, from the first sample's 60 rows only."""
    from causal_vae_amd.counterfactual import batched_counterfactual, sweep_inputs
    torch.manual_seed(42)
    model = CausalBioVAE3D().to(DEV).eval()
    sd = oracle.init_state_dict("bio3d", seed=42)
    g = torch.Generator().manual_seed(8)
    z, m = torch.randn(4, 64, generator=g), torch.rand(4, 12, generator=g)
    feats, vals = list(range(12)), [0.0, 0.25, 0.5, 0.75, 1.0]
    x = (torch.rand(240, 1, 64, 64, 64, generator=g) < 0.1).float().to(DEV)
    zd, md = z.to(DEV), m.to(DEV)
    ref = batched_counterfactual(model, zd, md, feats, vals)
    assert ref.shape == (4, 12, 5, 1, 64, 64, 64)
    zr_c, mc_c = sweep_inputs(z, m, feats, vals)
    ref_rows = ref.reshape(240, 1, 64, 64, 64).cpu()
    for r0 in range(0, 240, 60):                                   # EVERY row of the fp32 sweep against the oracle's decode of that row
        torch.testing.assert_close(ref_rows[r0:r0 + 60], ofn.bio_decode(sd, zr_c[r0:r0 + 60], mc_c[r0:r0 + 60], nd=3), rtol=1e-4, atol=1e-5)
    del ref_rows
    model.set_compute_dtype(torch.bfloat16)
    b16 = batched_counterfactual(model, zd, md, feats, vals, precision="bf16")
    assert model.dec_conv.compute_dtype == torch.bfloat16
    z_rep, m_cf = sweep_inputs(zd, md, feats, vals)
    plan = model.calibrate_fp8_decoder(z_rep[:60], m_cf[:60])
    assert [e["sx"] is not None for e in plan] == [True, True, True, False]       # the single-channel output layer keeps bf16 weights and output ...
    assert plan[-1]["sx8"] > 0                                                      # ... and reads the fp8 codes of the layer before it (cvae_conv_up_c1_fp8in)
    f8 = batched_counterfactual(model, zd, md, feats, vals, fp8_plan=plan)
    plan_r2 = model.calibrate_fp8_decoder(z_rep[:60], m_cf[:60], c1_fp8_input=False)   # round 2's plan: bf16 between the last two layers
    assert "sx8" not in plan_r2[-1]
    f8_r2 = batched_counterfactual(model, zd, md, feats, vals, fp8_plan=plan_r2)
    assert f8.shape == ref.shape and f8.dtype == ref.dtype
    sse = lambda a: ((a.reshape(240, -1) - x.reshape(240, -1)) ** 2).sum(1)
    rl2 = lambda a, b: float((a - b).norm() / b.norm())
    for name, got, bound in (("bf16", b16, 4.5e-3), ("fp8", f8, 9e-3), ("fp8, bf16 hand-off", f8_r2, 9e-3)):
        assert rl2(got, ref) < bound, (name, rl2(got, ref))
        assert rl2(got[1:], ref[1:]) < bound, (name, "rows outside the calibration set")
        d = float(((sse(got) - sse(ref)).abs() / sse(ref)).max())
        assert d < 2.5e-4, (name, "per-row SSE", d)
    # the fp8 result is a different computation from the bf16 one, not an alias of it
    assert rl2(f8, b16) > 1e-4
    # resize stage after the fp8 convs, and the eager per-row form gives the same rows (static scales: no batch dependence)
    up = batched_counterfactual(model, zd[:1], md[:1], [3], [0.1, 0.9], size=(96, 80, 72), fp8_plan=plan)
    assert up.shape == (1, 1, 2, 1, 96, 80, 72)
    one = model.decode(z_rep[77:78], m_cf[77:78], fp8_plan=plan)
    torch.testing.assert_close(one[0], f8.reshape(240, 1, 64, 64, 64)[77], rtol=0, atol=0)


def test_counterfactual_effect_is_preserved():
    """What a counterfactual sweep is FOR is the intervention effect decode(z, m') - decode(z, m) (generate_counterfactual.py:77-99 plots exactly these
    differences).  Against the CPU oracle's effect, on weights where the effect is >= 1 % of the output norm (dec_input's m-columns scaled by 10: 4.2 %;
    at random init it is 0.5 %, below bf16 / fp8 resolution — tools/effect_probe.py): the default (fp32) sweep holds 1e-4, bf16 8e-2, fp8 2.5e-1
    (measured 3.6e-6 / 5.2e-2 / 1.7e-1).  The error of a low-precision sweep relative to an effect of relative size f is eps_decode / f — hence fp32 is
    batched_counterfactual's default, and it must hold 1e-4 at the untrained model's 0.5 % too."""
    from causal_vae_amd.counterfactual import batched_counterfactual, sweep_inputs
    g = torch.Generator().manual_seed(8)
    z, m = torch.randn(2, 64, generator=g), torch.rand(2, 12, generator=g)
    feats, vals = [0, 5, 11], [0.0, 1.0]
    z_rep, m_cf = sweep_inputs(z, m, feats, vals)
    eff = lambda o: o.reshape(2, len(feats), 2, -1)[:, :, 1] - o.reshape(2, len(feats), 2, -1)[:, :, 0]
    for gain, bounds in ((10.0, {"fp32": 1e-4, "bf16": 8e-2, "fp8": 2.5e-1}), (1.0, {"fp32": 1e-4})):
        sd = oracle.init_state_dict("bio3d", seed=42)
        sd["dec_input.weight"][:, 64:] *= gain
        ref = ofn.bio_decode(sd, z_rep, m_cf, nd=3)
        e_ref = eff(ref)
        frac = float(e_ref.norm() / ref.norm()) * 2 ** 0.5          # effect rows against one decode's rows
        assert (frac >= 0.01) == (gain == 10.0), frac
        torch.manual_seed(42)
        model = CausalBioVAE3D().to(DEV).eval()
        model.load_state_dict({k: v.to(DEV) for k, v in sd.items()})
        zd, md = z.to(DEV), m.to(DEV)
        got = {"fp32": batched_counterfactual(model, zd, md, feats, vals).cpu()}                     # the default IS the exact one
        if "bf16" in bounds:
            got["bf16"] = batched_counterfactual(model, zd, md, feats, vals, precision="bf16").cpu()
            assert model.dec_conv.compute_dtype == torch.float32                                          # restored
            model.set_compute_dtype(torch.bfloat16)
            plan = model.calibrate_fp8_decoder(z_rep.to(DEV), m_cf.to(DEV))
            got["fp8"] = batched_counterfactual(model, zd, md, feats, vals, fp8_plan=plan).cpu()
        for name, o in got.items():
            err = float((eff(o) - e_ref).norm() / e_ref.norm())
            assert err < bounds[name], (gain, name, err, "effect / output", frac)


def test_gaussian_head_adversarial_step_matches_oracle():
    """06_model_experiment train loop body (D step, then VAE step with the Gaussian-NLL morph term) vs the oracle's step."""
    from causal_vae_amd.mnist_gaussian import CausalMorphVAE12 as GaussVAE, train_step as gauss_step
    g = torch.Generator().manual_seed(3)
    B = 16
    x, m = torch.rand(B, 1, 28, 28, generator=g), torch.rand(B, 12, generator=g)
    t = torch.nn.functional.one_hot(torch.randint(0, 10, (B,), generator=g), 10).float()
    eps = tuple(torch.randn(B, 10, generator=g) for _ in range(3))
    sd_v, sd_d = oracle.init_state_dict("morph12g", seed=42), oracle.init_state_dict("disc", seed=7)
    vae, disc = GaussVAE().to(DEV).train(), LatentDiscriminator().to(DEV).train()
    vae.load_state_dict(sd_v); disc.load_state_dict(sd_d)
    ref = oracle.mnist_adversarial_step(sd_v, sd_d, x, m, t, *eps, gaussian=True)
    opt_vae, opt_d = FusedAdam(vae.parameters(), lr=1e-3), FusedAdam(disc.parameters(), lr=1e-3)
    r = gauss_step(vae, disc, opt_vae, opt_d, x.to(DEV), m.to(DEV), t.to(DEV), eps=tuple(e.to(DEV) for e in eps))
    for k in ("loss_d", "loss", "recon", "kld", "morph", "adv"):
        assert rel(r[k], ref[k]) < 1e-4, (k, float(r[k]), float(ref[k]))
    for k, v in vae.state_dict().items():
        adam_close(v, sd_v[k], k)
    for k, v in disc.state_dict().items():
        adam_close(v, sd_d[k], k)


@pytest.mark.parametrize("cls,shape,dtype", [("3d", (3, 1, 32, 32, 32), torch.float32), ("3d", (4, 1, 64, 64, 64), torch.bfloat16),
                                             ("2d", (5, 1, 64, 96), torch.float32), ("3d", (16, 1, 32, 32, 32), torch.float32)])
def test_fused_bottleneck_equals_layer_by_layer_path(cls, shape, dtype):
    """ops.BioBottleneck (4 + 4 launches) against the same model run layer by layer: outputs, every gradient, BN buffers."""
    Model = CausalBioVAE3D if cls == "3d" else CausalBioVAE
    g = torch.Generator().manual_seed(11)
    B = shape[0]
    x, m = torch.randn(*shape, generator=g).to(DEV), torch.rand(B, 12, generator=g).to(DEV)
    t = torch.randint(0, 19, (B,), generator=g).to(DEV)
    eps = torch.randn(B, 64, generator=g).to(DEV)
    runs = {}
    for fused in (False, True):
        torch.manual_seed(42)
        model = Model().to(DEV).train().set_compute_dtype(dtype)
        model.fuse_bottleneck = fused
        recon, m_hat, mu, logvar = model(x, m, t, eps=eps)
        loss = ((recon - x) ** 2).sum() + 3.0 * ((m_hat - m) ** 2).sum() - 0.5 * torch.sum(1 + logvar - mu.pow(2) - logvar.exp())
        loss.backward()
        bn = model.mechanism_net[1]
        runs[fused] = dict(out=(recon, m_hat, mu, logvar), grads={k: p.grad.clone() for k, p in model.named_parameters()},
                           bn=(bn.running_mean.clone(), bn.running_var.clone(), int(bn.num_batches_tracked)))
    a, b = runs[False], runs[True]
    tight = dtype == torch.float32
    for u, v, name in zip(a["out"], b["out"], ("recon", "m_hat", "mu", "logvar")):
        torch.testing.assert_close(v, u, rtol=1e-4 if tight else 2e-2, atol=1e-5 if tight else 2e-2, msg=lambda s: f"{name}: {s}")
    assert a["bn"][2] == b["bn"][2] == 1
    torch.testing.assert_close(b["bn"][0], a["bn"][0], rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(b["bn"][1], a["bn"][1], rtol=1e-5, atol=1e-7)
    for k in a["grads"]:
        if k == NOISE_KEY:
            continue
        if tight:
            grad_close(b["grads"][k], a["grads"][k].cpu(), k, l2=1e-3, linf=1e-2)    # atomics order + isolated ReLU-mask flips
        else:       # bf16 convs: a flipped bf16 rounding in dec_input's output / the pooled gradient moves conv gradients by ~1 bf16 ulp
            grad_close(b["grads"][k], a["grads"][k].cpu(), k, l2=2e-2, linf=5e-2)


@pytest.mark.parametrize("cls,shape,dtype", [("3d", (2, 1, 128, 128, 128), torch.float32), ("3d", (2, 1, 128, 128, 128), torch.bfloat16),
                                             ("2d", (4, 1, 128, 128), torch.float32)])     # the decoder emits 64^nd: these are the exact-2x resizes
def test_forward_elbo_equals_forward_plus_loss_function(cls, shape, dtype):
    """model.forward_elbo (exact-2x resize folded into the ELBO, recon_x never written) == loss_function(model(x, m, t)):
    the three returned numbers and every gradient."""
    Model = CausalBioVAE3D if cls == "3d" else CausalBioVAE
    g = torch.Generator().manual_seed(5)
    B = shape[0]
    x, m = torch.randn(*shape, generator=g).to(DEV), torch.rand(B, 12, generator=g).to(DEV)
    t = torch.randint(0, 19, (B,), generator=g).to(DEV)
    eps = torch.randn(B, 64, generator=g).to(DEV)
    runs = []
    for fused in (False, True):
        torch.manual_seed(42)
        model = Model().to(DEV).train().set_compute_dtype(dtype)
        if fused:
            out = model.forward_elbo(x, m, t, eps=eps)
            assert ops_mod.ElboUp2x.supported(model._forward_cl(x, m, t, eps)[0], x)
        else:
            model.fuse_recon_loss = False
            o = model(x, m, t, eps=eps)
            out = loss_function(o[0], x, o[1], m, o[2], o[3])
        model.zero_grad()
        out[0].backward()
        runs.append((out, {k: p.grad.clone() for k, p in model.named_parameters()}))
    (oa, ga), (ob, gb) = runs
    for u, v, name in zip(oa, ob, ("loss", "recon", "m_loss")):
        assert rel(v, u) < 2e-6, (name, float(u), float(v))
    for k in ga:
        if k != NOISE_KEY:
            grad_close(gb[k], ga[k].cpu(), k, l2=1e-3 if dtype == torch.float32 else 2e-2, linf=1e-2 if dtype == torch.float32 else 5e-2)


def test_elbo_values_are_bit_reproducible_and_onehot_input_equals_label_input():
    """(a) forward_elbo twice on the same inputs: loss / recon / m_loss are identical bit for bit (fixed-order partial sums, no atomics
    on the loss path).  (b) ops.BioBottleneck fed the int64 labels (the one-hot is written by its first launch) == fed F.one_hot(t)."""
    g = torch.Generator().manual_seed(9)
    x, m = torch.randn(2, 1, 128, 128, 128, generator=g).to(DEV), torch.rand(2, 12, generator=g).to(DEV)
    t = torch.randint(0, 19, (2,), generator=g).to(DEV)
    eps = torch.randn(2, 64, generator=g).to(DEV)
    torch.manual_seed(1)
    model = CausalBioVAE3D().to(DEV).train().set_compute_dtype(torch.bfloat16)
    outs = []
    for _ in range(2):
        o = model.forward_elbo(x, m, t, eps=eps)
        outs.append([v.detach().clone() for v in o])
    for u, v in zip(*outs):
        assert torch.equal(u, v), (float(u), float(v))
    # (b) the bottleneck alone, both forms of t
    h = torch.randn(2, 8, 8, 8, 256, generator=g).to(DEV).relu().to(torch.bfloat16)
    bn = model.mechanism_net[1]
    lin = [model.enc_fc[0], model.enc_fc[2], model.fc_mu, model.fc_logvar, model.mechanism_net[0]]
    params = [p for l in lin for p in (l.weight, l.bias)] + [bn.weight, bn.bias]
    params += [p for l in (model.mechanism_net[3], model.mechanism_net[5], model.dec_input) for p in (l.weight, l.bias)]
    res = []
    for tt in (t, ops_mod.one_hot(t, 19)):
        rm, rv, nbt = bn.running_mean.clone(), bn.running_var.clone(), bn.num_batches_tracked.clone()
        with torch.no_grad():
            res.append(ops_mod.BioBottleneck.apply(h, m, tt, eps, *[p.detach() for p in params], rm, rv, nbt, bn.momentum, bn.eps, (4, 4, 4)))
    for u, v in zip(*res):
        assert torch.equal(u, v)


def test_noise_drawn_by_the_bottleneck_launch_is_the_philox_stream():
    """The fused step draws the reparameterisation noise in the bottleneck's first launch: same numbers, same counter bump as EpsSource.draw
    (cvae_philox_normal_advance) — two steps with the folded draw == two steps fed philox_normal(...) of calls 0 and 1 explicitly, bit for bit."""
    g = torch.Generator().manual_seed(23)
    x, m = torch.randn(2, 1, 64, 64, 64, generator=g).to(DEV), torch.rand(2, 12, generator=g).to(DEV)
    t = torch.tensor([4, 11]).to(DEV)
    res = []
    for explicit in (False, True):
        ops_mod.EpsSource._instances = 0
        torch.manual_seed(3)
        model = CausalBioVAE3D().to(DEV).train().set_compute_dtype(torch.bfloat16)
        losses = []
        for call in range(2):
            eps = None
            if explicit:
                eps = ops_mod.philox_normal((2, 64), torch.initial_seed(), call << 24, DEV, None, model._eps.subsequence())
            losses.append(model.forward_elbo(x, m, t, eps=eps)[0].detach().clone())
        res.append(losses)
        if not explicit:
            assert int(model._eps.counter.item()) == 2 and model._eps.state()["calls"] == 2
    assert not torch.equal(res[0][0], res[0][1])                 # the second call drew fresh numbers
    for u, v in zip(*res):
        assert torch.equal(u, v), (float(u), float(v))


def test_sync_batchnorm_inside_the_fused_bottleneck():
    """parallel.convert_sync_batchnorm keeps ops.BioBottleneck (cvae_bottleneck_*_sync).  (a) One rank: the gathered-statistics path is the plain
    fused step bit for bit — loss, every gradient, the running statistics.  (b) The statistics hand-off itself: the two halves of a batch of 4 as two
    "ranks" (their local statistics stacked by hand) give m_hat and the running statistics of the whole batch in one process (fp32 rounding apart),
    identical on both.  The backward hand-off (the all-reduced sums) is checked against the oracle's global-batch step by tests/dist_worker.py sync_bn."""
    from causal_vae_amd.parallel import convert_sync_batchnorm
    g = torch.Generator().manual_seed(21)
    x, m = torch.randn(4, 1, 64, 64, 64, generator=g).to(DEV), torch.rand(4, 12, generator=g).to(DEV)
    t = torch.tensor([3, 8, 13, 18]).to(DEV)
    eps = torch.randn(4, 64, generator=g).to(DEV)
    runs = []
    for sync in (False, True):
        torch.manual_seed(5)
        model = CausalBioVAE3D().to(DEV).train().set_compute_dtype(torch.bfloat16)
        if sync:
            convert_sync_batchnorm(model)
            assert model.fuse_bottleneck and model.mechanism_net[1].sync
        loss, _, _ = model.forward_elbo(x, m, t, eps=eps)
        ops_mod.backward_from(loss)
        assert model._enc_out is not None                    # the fused path ran
        runs.append((loss.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters()}, model.mechanism_net[1].running_mean.clone(),
                     model.mechanism_net[1].running_var.clone()))
    assert torch.equal(runs[0][0], runs[1][0])
    for k in runs[0][1]:
        assert torch.equal(runs[0][1][k], runs[1][1][k]), (k, float((runs[0][1][k] - runs[1][1][k]).abs().max()), float(runs[0][1][k].abs().max()))
    assert torch.equal(runs[0][2], runs[1][2]) and torch.equal(runs[0][3], runs[1][3])
    # (b) two half-batches with hand-stacked statistics against the whole batch
    lin0, bn = model.mechanism_net[0], model.mechanism_net[1]
    whole = ops_mod.bottleneck_bn_rank_stats(lin0.weight, lin0.bias, t)
    halves = torch.cat([ops_mod.bottleneck_bn_rank_stats(lin0.weight, lin0.bias, t[i:i + 2]) for i in (0, 2)])
    h = (lin0.weight.detach()[:, t].t() + lin0.bias.detach()).double()
    torch.testing.assert_close(whole[0, 0].double(), h.sum(0), rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(whole[0, 1].double(), ((h - h.mean(0)) ** 2).sum(0), rtol=1e-5, atol=1e-7)
    hy = torch.randn(4, 8, 8, 8, 256, generator=g).to(DEV).relu().to(torch.bfloat16)
    lins = [model.enc_fc[0], model.enc_fc[2], model.fc_mu, model.fc_logvar, lin0]
    params = [p for l in lins for p in (l.weight, l.bias)] + [bn.weight, bn.bias]
    params += [p for l in (model.mechanism_net[3], model.mechanism_net[5], model.dec_input) for p in (l.weight, l.bias)]
    def run(rows, stats):
        ps = [p.detach().clone().requires_grad_(True) for p in params]
        rm, rv, nbt = torch.zeros(64, device=DEV), torch.ones(64, device=DEV), torch.zeros((), dtype=torch.long, device=DEV)
        extra = () if stats is None else ((None, stats),)
        out = ops_mod.BioBottleneck.apply(hy[rows], m[rows], t[rows], eps[rows], *ps, rm, rv, nbt, bn.momentum, bn.eps, (4, 4, 4), *extra)
        return out, ps, rm, rv
    (mu_w, lv_w, mh_w, dec_w), ps_w, rm_w, rv_w = run(slice(0, 4), None)
    outs = [run(slice(i, i + 2), halves) for i in (0, 2)]
    torch.testing.assert_close(torch.cat([o[0][2] for o in outs]), mh_w, rtol=1e-5, atol=1e-6)          # m_hat of the global batch
    for o in outs:
        torch.testing.assert_close(o[2], rm_w, rtol=1e-6, atol=1e-7)
        torch.testing.assert_close(o[3], rv_w, rtol=1e-5, atol=1e-7)
    assert torch.equal(outs[0][2], outs[1][2]) and torch.equal(outs[0][3], outs[1][3])                     # replicas agree bit for bit


def test_adam_overlapped_with_backward_gives_the_same_training():
    """FusedAdam.overlap_backward: the non-encoder update runs on a side stream under the encoder backward; same losses and
    parameters as the plain step, bit for bit."""
    from causal_vae_amd.graph import GraphedTrainStep
    g = torch.Generator().manual_seed(13)
    x, m = torch.randn(2, 1, 32, 32, 32, generator=g).to(DEV), torch.rand(2, 12, generator=g).to(DEV)
    t = torch.randint(0, 19, (2,), generator=g).to(DEV)
    runs = []
    for mode in ("plain", "overlap", "overlap-graph"):
        ops_mod.EpsSource._instances = 0                      # same Philox subsequence for the arms compared below
        torch.manual_seed(42)
        model = CausalBioVAE3D().to(DEV).train()
        opt = FusedAdam(model.parameters(), lr=1e-4, device_step=True)
        if mode != "plain":
            opt.overlap_backward(model.early_gradient_parameters())
        if mode == "overlap-graph":
            gs = GraphedTrainStep(model, opt, (x, m, t), None, warmup=3)
            losses = [float(gs()[0]) for _ in range(3)]
        else:
            losses = [float(train_step(model, opt, x, m, t)[0]) for _ in range(6)][3:]
        torch.cuda.synchronize()
        runs.append((losses, [p.detach().clone() for p in model.parameters()], [int(opt.state[p]["step"]) for p in model.parameters()]))
    for losses, params, steps in runs[1:]:
        assert losses == runs[0][0], (runs[0][0], losses)            # bit for bit
        for (k, _), p, q in zip(CausalBioVAE3D().named_parameters(), runs[0][1], params):
            assert torch.equal(p, q), k
    assert len(set(runs[1][2])) == 1 and runs[1][2][0] == 6          # every parameter stepped exactly once per iteration


def test_vessel2d_matches_reference_golden(golden):
    """CausalVesselVAE (fp32 MFMA path) at the reference's 768 x 1280 resolution vs tensors produced by the reference class itself:
    init, 6-tuple forward, the vessel loss, every gradient, the BatchNorm buffers (tools/make_golden.py:vessel2d_case)."""
    from conftest import vessel2d_inputs
    from causal_vae_amd.vessel import CausalVesselVAE, loss_function as vessel_loss, total_loss
    g = golden("vessel2d_b4")
    torch.manual_seed(42)
    model = CausalVesselVAE().to(DEV).train()
    for k, v in model.state_dict().items():
        g.check("sd0", k, v, rtol=0, atol=0)
    B, seed = (int(v) for v in g.t("in/seed"))
    x, m, t, eps = (v.to(DEV) for v in vessel2d_inputs(B, seed))
    out = model(x, m, t, eps=eps)
    assert len(out) == 6
    names = ("recon_x", "m_hat", "mu", "logvar", "m_mu", "m_logvar")
    for k, v in zip(names, out):
        g.check("fwd", k, v, rtol=1e-3, atol=2e-4)
    recon, kld, morph, sparsity = vessel_loss(out[0], x, out[1], m, out[2], out[3], out[4], out[5])
    total = total_loss(recon, kld, morph, sparsity, beta=0.5, lambda_morph=1.0)
    for k, v in dict(recon=recon, kld=kld, morph=morph, sparsity=sparsity, total=total).items():
        assert rel(v, g.t("fwd/" + k)) < 1e-4, (k, float(v), float(g.t("fwd/" + k)))
    total.backward()
    noise = {f"enc_conv.{3 * i}.bias" for i in range(7)} | {f"dec_conv.{4 * i + 1}.bias" for i in range(6)} | {"enc_fc.0.bias", "dec_fc.0.bias"}
    for k, p in model.named_parameters():
        d = g.z["grad/" + k + "#digest"]
        f = p.grad.detach().double().flatten().cpu()
        ref_l2 = math.sqrt(d[2])
        if k in noise:              # a bias in front of a train-mode BatchNorm: exactly zero gradient in real arithmetic, rounding noise in both
            assert float(f.abs().max()) < 1e-5 * 4e6, k
            continue
        got_l2 = float(f.norm())
        assert abs(got_l2 - ref_l2) <= 2e-3 * ref_l2, (k, got_l2, ref_l2)
        assert abs(float(f.sum()) - d[0]) <= 5e-3 * d[1] + 1e-6, (k, "sum", float(f.sum()), d[0])
        np_head = f[:8].numpy()
        # single entries of a BatchNorm-chain gradient are sums with heavy cancellation: summation order (ours vs aten's) moves them by ~0.3 %
        assert np.allclose(np_head, d[3:3 + len(np_head)], rtol=5e-2, atol=1e-2 * float(f.abs().max())), (k, np_head, d[3:11])
    for k in g.keys("sd1"):
        g.check("sd1", k, model.state_dict()[k], rtol=1e-3, atol=1e-5)


def test_vessel2d_eval_mode_and_validate_match_reference_golden(golden):
    """Eval-mode CausalVesselVAE (BatchNorm2d / BatchNorm1d on running statistics) against the reference class in eval mode on the same batch
    (tools/make_golden.py:vessel2d_case, second file), and validate(vae, val_loader) (vessel_analysis/01_train/train.py:100-133): eval mode,
    no_grad, same loss composition, sum / len(dataset); the model is left in the mode it came in."""
    from conftest import vessel2d_inputs
    from causal_vae_amd.vessel import CausalVesselVAE, loss_function as vessel_loss, total_loss, validate
    g, ge = golden("vessel2d_b4"), golden("vessel2d_b4_eval")
    torch.manual_seed(42)
    model = CausalVesselVAE().to(DEV)
    sd = model.state_dict()
    for k in g.keys("sd1"):                                   # the buffers the reference's training forward left behind
        sd[k].copy_(g.t("sd1/" + k).to(DEV))
    B, seed = (int(v) for v in ge.t("in/seed"))
    x, m, t, eps = (v.to(DEV) for v in vessel2d_inputs(B, seed))
    model.eval()
    with torch.no_grad():
        out = model(x, m, t, eps=eps)
        terms = vessel_loss(out[0], x, out[1], m, out[2], out[3], out[4], out[5])
        total = total_loss(*terms, beta=0.5)
    for k, v in zip(("recon_x", "m_hat", "mu", "logvar", "m_mu", "m_logvar"), out):
        ge.check("eval", k, v, rtol=1e-3, atol=2e-4)
    for k, v in zip(("recon", "kld", "morph", "sparsity", "total"), (*terms, total)):
        assert rel(v, ge.t("eval/" + k)) < 1e-4, (k, float(v), float(ge.t("eval/" + k)))
    # validate(): two batches of 2 from the same 4 samples.  The forward draws its own eps there (reference: reparameterize inside forward),
    # so only the eps-independent structure is checked against the golden: mode handling, no gradient state, finite, per-sample scale.
    data = [(x[i].cpu(), m[i].cpu(), t[i].cpu()) for i in range(B)]
    loader = torch.utils.data.DataLoader(data, batch_size=2, shuffle=False)
    model.train()
    before = {k: v.clone() for k, v in model.state_dict().items()}
    val = validate(model, loader, device=DEV)
    assert model.training                                                     # restored
    assert all(torch.equal(before[k], v) for k, v in model.state_dict().items())   # eval mode: no running-stat update, no weight change
    assert all(p.grad is None for p in model.parameters())
    ref_per_sample = float(ge.t("eval/total")) / B
    assert math.isfinite(val) and abs(val - ref_per_sample) < 0.05 * ref_per_sample, (val, ref_per_sample)   # eps only enters through z (5 % bound)


def test_vessel2d_bf16_step_tracks_fp32():
    """bf16 conv arithmetic on the 2D vessel model: loss within 1e-2 of the fp32 path on the same batch, finite gradients."""
    from conftest import vessel2d_inputs
    from causal_vae_amd.vessel import CausalVesselVAE, loss_function as vessel_loss, total_loss
    x, m, t, eps = (v.to(DEV) for v in vessel2d_inputs(2, 99))
    tot = {}
    for dtype in (torch.float32, torch.bfloat16):
        torch.manual_seed(42)
        model = CausalVesselVAE().to(DEV).train().set_compute_dtype(dtype)
        out = model(x, m, t, eps=eps)
        total = total_loss(*vessel_loss(out[0], x, out[1], m, out[2], out[3], out[4], out[5]))
        total.backward()
        assert all(torch.isfinite(p.grad).all() for p in model.parameters())
        tot[dtype] = float(total)
    assert abs(tot[torch.bfloat16] - tot[torch.float32]) <= 1e-2 * abs(tot[torch.float32]), tot


def test_vessel2d_train_step_clip_and_adam_follow_torch():
    """The vessel recipe's step (forward, vessel loss, backward, clip_grad_norm_ 5.0, Adam 1e-4) with FusedAdam == the same model stepped
    with torch.nn.utils.clip_grad_norm_ + torch.optim.Adam on its (HIP-computed) gradients: losses and updated weights over 2 steps."""
    from conftest import vessel2d_inputs
    from causal_vae_amd.vessel import CausalVesselVAE, train_step as vessel_step
    x, m, t, eps = (v.to(DEV) for v in vessel2d_inputs(4, 7))
    res = []
    for fused in (True, False):
        torch.manual_seed(42)
        model = CausalVesselVAE().to(DEV).train()
        opt = FusedAdam(model.parameters(), lr=1e-4) if fused else torch.optim.Adam(model.parameters(), lr=1e-4)
        losses = [float(vessel_step(model, opt, x, m, t, eps=eps)[0]) for _ in range(2)]
        res.append((losses, {k: p.detach().clone() for k, p in model.named_parameters()}))
    (la, pa), (lb, pb) = res
    for a, b in zip(la, lb):
        assert rel(a, b) < 5e-5, (la, lb)
    noise = {f"enc_conv.{3 * i}.bias" for i in range(7)} | {f"dec_conv.{4 * i + 1}.bias" for i in range(6)} | {"enc_fc.0.bias", "dec_fc.0.bias"}
    for k in pa:
        p, q = pa[k], pb[k]
        assert float((p - q).abs().max()) <= 2 * 1e-4 * 2 + 1e-7          # Adam: +-lr per step where a clipped gradient is ~0
        if k in noise:                                                     # zero-gradient biases in front of a BatchNorm: Adam steps on rounding noise
            continue
        # the two arms differ in the order the clipped gradient is rounded (scale-then-Adam vs Adam-with-scale), which moves isolated
        # ~0-gradient weights by a whole step; the bound is 15 % of the two steps' travel — a wrong update rule would show up as ~100 %
        assert float((p - q).abs().mean()) <= 3e-5, k


@pytest.mark.parametrize("B,dtype,tol,lin", [(128, torch.float32, 1e-4, None), (1024, torch.float32, 1e-4, None), (1024, torch.bfloat16, 2e-3, None),
                                              (1024, torch.bfloat16, 2e-3, torch.bfloat16)], ids=["b128-f32", "b1024-f32", "b1024-bf16", "b1024-bf16-linears"])
def test_mnist_batch_1024_step_matches_oracle(B, dtype, tol, lin):
    """BASELINE.json configs[0] (batch 128, fp32: the reference's own mnist_test/01 configuration, config.py BATCH_SIZE) and configs[1] (the MNIST
    CausalMorphVAE12 adversarial step at batch 1024; fp32 parity and bf16 = the configuration named there) against the CPU oracle's step on the
    same batch: every loss term."""
    g = torch.Generator().manual_seed(1024)
    x, m = torch.rand(B, 1, 28, 28, generator=g), torch.rand(B, 12, generator=g)
    t = torch.nn.functional.one_hot(torch.randint(0, 10, (B,), generator=g), 10).float()
    eps = tuple(torch.randn(B, 10, generator=g) for _ in range(3))
    sd_v, sd_d = oracle.init_state_dict("morph12", seed=42), oracle.init_state_dict("disc", seed=7)
    vae, disc = CausalMorphVAE12().to(DEV).train(), LatentDiscriminator().to(DEV).train()
    vae.load_state_dict(sd_v); disc.load_state_dict(sd_d)
    vae.set_compute_dtype(dtype)
    if lin is not None:                                      # the large linears (1024 x 3158 x 512, 1024 x 22 x 3136) on bf16 MFMA operands too
        from causal_vae_amd.layers import set_linear_math
        set_linear_math(vae, lin); set_linear_math(disc, lin)
    ref = oracle.mnist_adversarial_step(sd_v, sd_d, x, m, t, *eps, apply_update=False)
    opt_vae, opt_d = FusedAdam(vae.parameters(), lr=1e-3), FusedAdam(disc.parameters(), lr=1e-3)
    # the oracle's VAE half uses the UPDATED discriminator; apply_update=False keeps both on the initial weights, so mirror that: lr = 0 for D
    opt_d.param_groups[0]["lr"] = 0.0
    r = mnist_train_step(vae, disc, opt_vae, opt_d, x.to(DEV), m.to(DEV), t.to(DEV), eps=tuple(e.to(DEV) for e in eps))
    for k in ("loss_d", "loss", "recon", "kld", "morph", "adv"):
        assert rel(r[k], ref[k]) < tol, (k, float(r[k]), float(ref[k]))


def test_mnist_adversarial_step_replays_as_one_hip_graph():
    """The whole MNIST step (D update, then VAE update; two device-step FusedAdams, Philox eps) captured once and replayed == eager steps."""
    from causal_vae_amd.graph import GraphedCallable
    g = torch.Generator().manual_seed(77)
    B = 256
    x, m = torch.rand(B, 1, 28, 28, generator=g).to(DEV), torch.rand(B, 12, generator=g).to(DEV)
    t = torch.nn.functional.one_hot(torch.randint(0, 10, (B,), generator=g), 10).float().to(DEV)
    runs = []
    for graphed in (False, True):
        ops_mod.EpsSource._instances = 0                      # same Philox subsequence for the arms compared below
        torch.manual_seed(42)
        vae, disc = CausalMorphVAE12().to(DEV).train(), LatentDiscriminator().to(DEV).train()
        ov, od = FusedAdam(vae.parameters(), lr=1e-3, device_step=True), FusedAdam(disc.parameters(), lr=1e-3, device_step=True)
        step = lambda: mnist_train_step(vae, disc, ov, od, x, m, t)
        if graphed:
            gs = GraphedCallable(step, warmup=3)
            losses = [{k: float(v) for k, v in gs().items()} for _ in range(3)]
        else:
            losses = [{k: float(v) for k, v in step().items()} for _ in range(6)][3:]
        runs.append(losses)
    for a, b in zip(*runs):
        for k in a:
            assert a[k] == b[k], (k, runs)                          # bit for bit: deterministic reductions everywhere
    assert len({l["loss"] for l in runs[1]}) == 3
