"""Pins the CPU oracle (oracle/) to golden vectors captured from the imported reference classes
(tools/make_golden.py).  Runs without a GPU.  fp32 tolerance: rtol 1e-6 on forward values (same aten
kernels, same order), 2e-5 on gradients/updated weights (autograd accumulation order may differ)."""
import numpy as np
import pytest
import torch

import oracle
from oracle import functional as ofn

BIO_CASES = ["bio2d_b4_64x96", "bio2d_b2_64x64", "bio2d_b3_128x160"]
# A Linear bias feeding train-mode BatchNorm1d (causal_cascade/models.py:35-36) has a mathematically zero
# gradient (BN subtracts the batch mean), so its computed value is rounding noise of order 1e-3.
NOISE_KEY = "mechanism_net.0.bias"


@pytest.mark.parametrize("case", BIO_CASES)
def test_bio2d_init_matches_reference(golden, case):
    g = golden(case)
    sd = oracle.init_state_dict("bio2d", seed=42)
    assert sorted(sd) == g.keys("sd0")
    for k, v in sd.items():
        g.check("sd0", k, v, rtol=0, atol=0)


@pytest.mark.parametrize("case", BIO_CASES)
def test_bio2d_forward_loss_grads_adam(golden, case):
    g = golden(case)
    sd = oracle.init_state_dict("bio2d", seed=42)
    x, m, t, eps = g.t("in/x"), g.t("in/m"), g.t("in/t"), g.t("in/eps")
    fwd = ofn.bio_vae_forward({k: v.clone() for k, v in sd.items()}, x, m, t, eps, keep_acts=True)
    for k in ("recon_x", "m_hat", "mu", "logvar", "z"):
        g.check("out", k, fwd[k], rtol=1e-6, atol=1e-6)
    for k in g.keys("act"):
        g.check("act", k, fwd["acts"][k], rtol=1e-6, atol=1e-6)
    st = oracle.cascade_train_step(sd, x, m, t, eps)
    for k in ("loss", "recon", "m_loss"):
        g.check("loss", k, st[k], rtol=1e-6, atol=1e-4)
    g.check("loss", "kld", st["kld"], rtol=1e-4, atol=2e-2)      # fixture kld = loss - recon - 2000 m (cancellation)
    for k in g.keys("grad"):
        if k == NOISE_KEY:      # exactly 0 in real arithmetic: pure rounding noise in reference and oracle alike
            assert st["grads"][k].abs().max() < 0.05
            continue
        g.check("grad", k, st["grads"][k], rtol=2e-5, atol=2e-5)
    for k in g.keys("sd1"):                                        # after one Adam step + BN running stats
        if k == NOISE_KEY:      # Adam turns that noise into +-lr steps; BN cancels the bias, outputs unaffected
            assert (sd[k] - g.t("sd1/" + k)).abs().max() <= 2.0e-3 + 1e-6
            continue
        g.check("sd1", k, sd[k], rtol=2e-5, atol=2e-6)
    sd_eval = {k: v.clone() for k, v in sd.items()}
    ev = ofn.bio_vae_forward(sd_eval, x, m, t, eps, training=False)
    g.check("eval", "m_hat", ev["m_hat"], rtol=1e-5, atol=1e-6)


def test_bn1d_batch_of_one_raises_like_reference():
    sd = oracle.init_state_dict("bio2d", seed=0)
    with pytest.raises(ValueError):
        ofn.bio_vae_forward(sd, torch.zeros(1, 1, 64, 64), torch.zeros(1, 12), torch.zeros(1, dtype=torch.long),
                            torch.zeros(1, 64))


def test_morph12_forward_and_adversarial_step(golden):
    g = golden("morph12_b8")
    sd = oracle.init_state_dict("morph12", seed=42)
    sdd = oracle.init_state_dict("disc")            # drawn right after the VAE, same generator stream
    for k, v in sd.items():
        g.check("sd0", k, v, rtol=0, atol=0)
    for k, v in sdd.items():
        g.check("sdd0", k, v, rtol=0, atol=0)
    x, m, t = g.t("in/x"), g.t("in/m"), g.t("in/t")
    fwd = ofn.morph_vae_forward(sd, x, m, t, g.t("fwd/eps"))
    for k in ("recon_x", "m_hat", "mu", "logvar", "z"):
        g.check("fwd", k, fwd[k], rtol=1e-6, atol=1e-6)
    st = oracle.mnist_adversarial_step(sd, sdd, x, m, t, g.t("step/eps_d"), g.t("step/eps_vae"), g.t("step/eps_adv"))
    for k in ("loss_d", "loss", "recon", "kld", "morph", "adv"):
        g.check("step", k, st[k], rtol=2e-6, atol=1e-4)
    for k in g.keys("gradd"):
        g.check("gradd", k, st["grads_d"][k], rtol=2e-5, atol=1e-6)
    for k in g.keys("gradv"):
        g.check("gradv", k, st["grads_vae"][k], rtol=5e-5, atol=5e-5)
    for k in g.keys("sd1"):
        g.check("sd1", k, sd[k], rtol=2e-5, atol=2e-6)
    for k in g.keys("sdd1"):
        g.check("sdd1", k, sdd[k], rtol=2e-5, atol=2e-6)


def test_morph12_gaussian_head_forward(golden):
    g = golden("morph12g_b8")
    sd = oracle.init_state_dict("morph12g", seed=42)
    for k, v in sd.items():
        g.check("sd0", k, v, rtol=0, atol=0)
    x, m, t = g.t("in/x"), g.t("in/m"), g.t("in/t")
    fwd = ofn.morph_vae6_forward(sd, x, m, t, g.t("fwd/eps"))
    for k in ("recon_x", "m_hat", "mu", "logvar", "z", "m_mu", "m_logvar"):
        g.check("fwd", k, fwd[k], rtol=1e-6, atol=1e-6)
    g.check("fwd", "nll", ofn.gaussian_nll(m, fwd["m_mu"], fwd["m_logvar"]), rtol=1e-6, atol=1e-5)


@pytest.mark.parametrize("tag", ["d10", "d001", "d60"])
def test_vessel_loss_matches_reference(golden, tag):
    g = golden("vessel_loss")
    a = {k: g.t(f"{tag}/{k}") for k in ("x", "recon_x", "m", "m_mu", "m_logvar", "mu", "logvar")}
    for k in ("recon_x", "m_mu", "m_logvar", "mu", "logvar"):
        a[k].requires_grad_(True)
    recon, kld, morph, sparsity = ofn.vessel_loss(a["recon_x"], a["x"], a["m_mu"], a["m"], a["mu"], a["logvar"],
                                                  a["m_mu"], a["m_logvar"])
    total = recon + 0.5 * kld + morph + 0.3 * sparsity
    total.backward()
    for k, v in dict(recon=recon, kld=kld, morph=morph, sparsity=sparsity, total=total).items():
        g.check(tag, k, v, rtol=2e-6, atol=1e-4)
    for k in ("recon_x", "m_mu", "m_logvar", "mu", "logvar"):
        g.check(tag, "g_" + k, a[k].grad, rtol=1e-5, atol=1e-6)


def test_vessel2d_matches_reference(golden):
    """CausalVesselVAE (vessel_analysis/00_core/models.py:9-166): init, forward, the vessel loss and every gradient of the oracle's
    restatement against the reference class run on the same batch (B = 2, 768 x 1280; tools/make_golden.py:vessel2d_case)."""
    from conftest import vessel2d_inputs
    g = golden("vessel2d_b4")
    sd = oracle.init_state_dict("vessel2d", seed=42)
    assert sorted(sd) == g.keys("sd0")
    for k, v in sd.items():
        g.check("sd0", k, v, rtol=0, atol=0)
    B, seed = (int(v) for v in g.t("in/seed"))
    x, m, t, eps = vessel2d_inputs(B, seed)
    g.check("in", "x", x, rtol=0, atol=0)
    torch.testing.assert_close(m, g.t("in/m"), rtol=0, atol=0)
    leaves = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone()) for k, v in sd.items()}
    out = ofn.vessel_vae_forward(leaves, x, m, t, eps)
    for k in ("recon_x", "m_hat", "mu", "logvar", "m_mu", "m_logvar"):
        g.check("fwd", k, out[k], rtol=2e-5, atol=2e-6)
    recon, kld, morph, sparsity = ofn.vessel_loss(out["recon_x"], x, out["m_hat"], m, out["mu"], out["logvar"], out["m_mu"], out["m_logvar"])
    total = recon + 0.5 * kld + morph + 0.3 * sparsity
    for k, v in dict(recon=recon, kld=kld, morph=morph, sparsity=sparsity, total=total).items():
        g.check("fwd", k, v, rtol=2e-5, atol=1e-3)
    keys = [k for k in g.keys("grad")]
    noise = {f"enc_conv.{3 * i}.bias" for i in range(7)} | {f"dec_conv.{4 * i + 1}.bias" for i in range(6)} | {"enc_fc.0.bias", "dec_fc.0.bias"}
    grads = torch.autograd.grad(total, [leaves[k] for k in keys])
    for k, gr in zip(keys, grads):
        scale = float(gr.abs().max())
        if k in noise:              # biases in front of a train-mode BatchNorm: zero in exact arithmetic, rounding noise in both implementations
            continue
        g.check("grad", k, gr, rtol=5e-4, atol=5e-4 * scale)
    for k in g.keys("sd1"):
        g.check("sd1", k, leaves[k], rtol=2e-5, atol=1e-6)
    # eval mode on the same batch with the running statistics just updated — the forward validate() runs (01_train/train.py:100-133)
    ge = golden("vessel2d_b4_eval")
    with torch.no_grad():
        ev = ofn.vessel_vae_forward({k: v.detach() for k, v in leaves.items()}, x, m, t, eps, training=False)
        r2, k2, mo2, sp2 = ofn.vessel_loss(ev["recon_x"], x, ev["m_hat"], m, ev["mu"], ev["logvar"], ev["m_mu"], ev["m_logvar"])
    for k in ("recon_x", "m_hat", "mu", "logvar", "m_mu", "m_logvar"):
        ge.check("eval", k, ev[k], rtol=5e-5, atol=5e-6)
    for k, v in dict(recon=r2, kld=k2, morph=mo2, sparsity=sp2, total=r2 + 0.5 * k2 + mo2 + 0.3 * sp2).items():
        ge.check("eval", k, v, rtol=5e-5, atol=1e-3)


def test_bio3d_degenerates_to_2d_slicewise():
    """a 3D conv whose weight is zero except one depth tap reproduces the 2D
    result slice-wise — ties the 3D lift's conv/convT arithmetic to the golden-pinned 2D path."""
    import torch.nn.functional as F
    torch.manual_seed(3)
    w2 = torch.randn(8, 4, 4, 4); b = torch.randn(8)
    x3 = torch.randn(2, 4, 6, 10, 12)
    w3 = torch.zeros(8, 4, 4, 4, 4); w3[:, :, 1] = w2            # tap kd=1 reads input depth 2*od
    y3 = F.conv3d(x3, w3, b, stride=2, padding=1)
    for od in range(3):
        y2 = F.conv2d(x3[:, :, 2 * od], w2, b, stride=2, padding=1)
        torch.testing.assert_close(y3[:, :, od], y2, rtol=1e-5, atol=1e-5)
    wt2 = torch.randn(4, 8, 4, 4)
    wt3 = torch.zeros(4, 8, 4, 4, 4); wt3[:, :, 1] = wt2         # convT tap kd=1 writes depth 2*id
    y3 = F.conv_transpose3d(x3, wt3, None, stride=2, padding=1)
    for i in range(6):
        y2 = F.conv_transpose2d(x3[:, :, i], wt2, None, stride=2, padding=1)
        torch.testing.assert_close(y3[:, :, 2 * i], y2, rtol=1e-5, atol=1e-5)


def test_bio3d_oracle_shapes_and_keys():
    sd2 = oracle.init_state_dict("bio2d", seed=1)
    sd3 = oracle.init_state_dict("bio3d", seed=1)
    assert list(sd2) == list(sd3)                                 # identical module tree / key names
    assert sum(v.numel() for k, v in sd3.items() if "running" not in k and "num_batches" not in k) == 15_346_957
    assert sd3["enc_fc.0.weight"].shape == (512, 16384 + 12 + 19)
    assert sd3["dec_input.weight"].shape == (16384, 76)
    x = torch.randn(2, 1, 32, 32, 32)
    out = ofn.bio_vae_forward(sd3, x, torch.rand(2, 12), torch.tensor([3, 7]), torch.randn(2, 64))
    assert out["recon_x"].shape == x.shape and out["mu"].shape == (2, 64)


def test_conditional_vae_init_forward_loss_grads_adam(golden):
    """ConditionalVAE (mnist_test/03_measurement_approach): init order, forward, BCE + KLD, every gradient, one Adam step."""
    g = golden("mnist_cvae_b8")
    sd = oracle.init_state_dict("cvae", seed=42)
    assert sorted(sd) == g.keys("sd0")
    for k, v in sd.items():
        g.check("sd0", k, v, rtol=0, atol=0)
    x, t, eps = g.t("in/x"), g.t("in/t"), g.t("fwd/eps")
    st = oracle.cvae_train_step(sd, x, t, eps)
    for k in ("recon_x", "mu", "logvar", "z"):
        g.check("fwd", k, st["outputs"][k], rtol=1e-6, atol=1e-6)
    for k in ("loss", "recon", "kld"):
        g.check("fwd", k, st[k], rtol=1e-6, atol=1e-4)
    for k in g.keys("grad"):
        g.check("grad", k, st["grads"][k], rtol=2e-5, atol=2e-5)
    for k in g.keys("sd1"):
        g.check("sd1", k, sd[k], rtol=2e-5, atol=2e-6)
