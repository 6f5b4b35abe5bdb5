#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own classes (build container only).

Run from the repo root:  PYTHONDONTWRITEBYTECODE=1 python tools/make_golden.py

What is imported from /root/reference (read-only, nothing is written there):
  * causal_cascade/models.py  -> CausalBioVAE           (a1-a5)
  * causal_cascade/train.py   -> loss_function          (a6)
  * mnist_test/01_baseline_causal_vae/{config,models}.py -> CausalMorphVAE12, LatentDiscriminator (a8)
  * mnist_test/06_model_experiment/{config,models}.py    -> Gaussian-head CausalMorphVAE12
  * vessel_analysis/00_core/models.py: the text of ``CausalVesselVAE`` is compiled with its two unimportable imports dropped (a11)
  * vessel_analysis/01_train/train.py: only the text of ``loss_function`` is compiled (the module
    itself cannot be imported: it pulls tifffile/torchvision through ``dataset``) (a10)
The MNIST adversarial loop body (mnist_test/01_baseline_causal_vae/train.py:34-93) cannot be imported
(torchvision at module import); its goldens are produced by driving the *imported reference models*
with the same torch calls the loop makes (F.cross_entropy, F.binary_cross_entropy, F.kl_div,
optim.Adam), in the loop's order, with the loop's six eps draws per step.

The fixtures hold data only: inputs, injected eps, outputs, loss scalars, gradients (full for small
tensors, digests for large ones) and state_dict digests.  /root/reference never travels.
"""
import ast
import importlib
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
sys.dont_write_bytecode = True
torch.set_num_threads(8)
FULL_LIMIT = 20000          # tensors up to this many elements are stored whole


def digest(t):
    """[sum, abs-sum, sum of squares, first 8 values, last 8 values] in float64."""
    f = t.detach().double().flatten()
    head = torch.zeros(8, dtype=torch.float64); tail = torch.zeros(8, dtype=torch.float64)
    n = min(8, f.numel())
    head[:n] = f[:n]; tail[:n] = f[-n:]
    return torch.cat([torch.stack([f.sum(), f.abs().sum(), (f * f).sum()]), head, tail]).numpy()


def pack(prefix, named, store):
    for k, v in named.items():
        v = v.detach()
        store[f"{prefix}/{k}#digest"] = digest(v)
        store[f"{prefix}/{k}#shape"] = np.array(v.shape, dtype=np.int64)
        if v.numel() <= FULL_LIMIT:
            store[f"{prefix}/{k}"] = v.numpy().copy()


def import_from(dirname, *modules):
    """Import bare-named modules (config, models, ...) from one reference directory."""
    for name in ("config", "models", "train", "dataset", "cvae_models"):
        sys.modules.pop(name, None)
    sys.path.insert(0, dirname)
    try:
        return [importlib.import_module(mod) for mod in modules]
    finally:
        sys.path.remove(dirname)


def bio2d_case(name, B, H, W, seed_data):
    models, train = import_from(os.path.join(REF, "causal_cascade"), "models", "train")
    torch.manual_seed(42)                              # causal_cascade/main.py:28
    model = models.CausalBioVAE(img_channels=1, m_dim=12, t_dim=19, latent_dim=64)
    model.train()
    store = {}
    pack("sd0", model.state_dict(), store)
    g = torch.Generator().manual_seed(seed_data)
    x = torch.randn(B, 1, H, W, generator=g)
    m = torch.rand(B, 12, generator=g)
    t = torch.randint(0, 19, (B,), generator=g)
    acts = {}
    hooks = []
    for i in range(4):
        hooks.append(model.enc_conv[2 * i + 1].register_forward_hook(
            lambda _m, _i, o, k=f"enc{i+1}": acts.__setitem__(k, o.detach().clone())))
    for i in range(3):
        hooks.append(model.dec_conv[2 * i + 1].register_forward_hook(
            lambda _m, _i, o, k=f"dec{i+1}": acts.__setitem__(k, o.detach().clone())))
    hooks.append(model.dec_conv[6].register_forward_hook(
        lambda _m, _i, o: acts.__setitem__("dec4", o.detach().clone())))
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)          # main.py:50
    opt.zero_grad()
    torch.manual_seed(999)
    recon_x, m_hat, mu, logvar = model(x, m, t)
    torch.manual_seed(999)
    eps = torch.randn(B, 64)                                      # the eps forward() drew
    z = mu + eps * torch.exp(0.5 * logvar)
    for h in hooks:
        h.remove()
    loss, l_recon, l_m = train.loss_function(recon_x, x, m_hat, m, mu, logvar)
    kld = loss - l_recon - 2000.0 * l_m
    loss.backward()
    grads = {k: p.grad.clone() for k, p in model.named_parameters()}
    opt.step()
    # self-check of the eps recipe: decode by hand from z in eval-free manner
    with torch.no_grad():
        t1h = F.one_hot(t, 19).float()
        chk = model.dec_input(torch.cat([z, m_hat], 1)).view(-1, 256, 4, 4)
    store.update({"in/x": x.numpy(), "in/m": m.numpy(), "in/t": t.numpy(), "in/eps": eps.numpy()})
    pack("out", dict(recon_x=recon_x, m_hat=m_hat, mu=mu, logvar=logvar, z=z), store)
    pack("act", acts, store)
    pack("loss", dict(loss=loss, recon=l_recon, m_loss=l_m, kld=kld), store)
    pack("grad", grads, store)
    pack("sd1", model.state_dict(), store)                        # after one Adam step (+BN running stats)
    # second forward in eval mode with the updated weights (BN running stats path; analyze.py:10-23)
    model.eval()
    with torch.no_grad():
        m_hat_eval = model.mechanism_net(F.one_hot(t, 19).float())
    pack("eval", dict(m_hat=m_hat_eval), store)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **store)
    print(name, "loss", float(loss), "keys", len(store))


def morph_case(name, B, seed_data, gaussian_head):
    sub = "06_model_experiment" if gaussian_head else "01_baseline_causal_vae"
    config, models = import_from(os.path.join(REF, "mnist_test", sub), "config", "models")
    CONFIG = config.CONFIG
    torch.manual_seed(CONFIG["SEED"])
    vae = models.CausalMorphVAE12(); disc = models.LatentDiscriminator()
    vae.train(); disc.train()
    store = {}
    pack("sd0", vae.state_dict(), store)
    pack("sdd0", disc.state_dict(), store)
    g = torch.Generator().manual_seed(seed_data)
    x = torch.rand(B, 1, 28, 28, generator=g)
    m = torch.rand(B, 12, generator=g)
    t = F.one_hot(torch.randint(0, 10, (B,), generator=g), 10).float()
    store.update({"in/x": x.numpy(), "in/m": m.numpy(), "in/t": t.numpy()})
    # ---- plain forward with a known eps ----
    torch.manual_seed(999)
    outs = vae(x, m, t)
    torch.manual_seed(999)
    eps = torch.randn(B, 10)
    names = ["recon_x", "m_hat", "mu", "logvar", "m_mu", "m_logvar"][:len(outs)]
    fwd = dict(zip(names, outs))
    fwd["z"] = fwd["mu"] + eps * torch.exp(0.5 * fwd["logvar"])
    store["fwd/eps"] = eps.numpy()
    pack("fwd", fwd, store)
    if gaussian_head:
        nll = 0.5 * torch.sum(fwd["m_logvar"] + (m - fwd["m_mu"]) ** 2 / fwd["m_logvar"].exp())   # 06/train.py:79
        pack("fwd", dict(nll=nll), store)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **store)
        print(name, "keys", len(store))
        return
    # ---- one adversarial step, in the order of 01/train.py:34-93 ----
    opt_vae = torch.optim.Adam(vae.parameters(), lr=CONFIG["LR"])
    opt_d = torch.optim.Adam(disc.parameters(), lr=CONFIG["LR"])
    torch.manual_seed(4321)
    e = [None] * 6
    t_indices = torch.argmax(t, dim=1)
    opt_d.zero_grad()
    with torch.no_grad():
        _, _, mu, logvar = vae(x, m, t)                 # draw #1
        z = vae.reparameterize(mu, logvar).detach()     # draw #2
        _, _, mu, logvar = vae(x, m, t)                 # draw #3
        std = torch.exp(0.5 * logvar)
        eps4 = torch.randn_like(std)                    # draw #4
        z = mu + eps4 * std
    d_logits = disc(z)
    loss_d = F.cross_entropy(d_logits, t_indices)
    loss_d.backward()
    grads_d = {k: p.grad.clone() for k, p in disc.named_parameters()}
    opt_d.step()
    opt_vae.zero_grad()
    recon_x, m_hat, mu, logvar = vae(x, m, t)           # draw #5
    loss_recon = F.binary_cross_entropy(recon_x.view(-1, 784), x.view(-1, 784), reduction='sum')
    loss_kld = -0.5 * torch.sum(1 + logvar - mu.pow(2) - logvar.exp()) * CONFIG["BETA"]
    loss_morph = F.mse_loss(m_hat, m, reduction='sum') * 100
    z_sample = vae.reparameterize(mu, logvar)           # draw #6
    d_fake = disc(z_sample)
    target_uniform = torch.full_like(d_fake, 1.0 / CONFIG["T_DIM"])
    loss_adv = F.kl_div(F.log_softmax(d_fake, dim=1), target_uniform, reduction='batchmean') * CONFIG["LAMBDA_ADV"] * 100
    loss = loss_recon + loss_kld + loss_morph + loss_adv
    loss.backward()
    grads_v = {k: p.grad.clone() for k, p in vae.named_parameters()}
    opt_vae.step()
    torch.manual_seed(4321)
    for i in range(6):
        e[i] = torch.randn(B, 10)
    assert torch.equal(e[3], eps4), "eps draw-order recipe broken"
    assert torch.allclose(z_sample, mu + e[5] * torch.exp(0.5 * logvar), atol=0, rtol=0)
    store.update({"step/eps_d": e[3].numpy(), "step/eps_vae": e[4].numpy(), "step/eps_adv": e[5].numpy()})
    pack("step", dict(loss_d=loss_d, loss=loss, recon=loss_recon, kld=loss_kld, morph=loss_morph, adv=loss_adv,
                      recon_x=recon_x, mu=mu, logvar=logvar, m_hat=m_hat, d_logits=d_logits), store)
    pack("gradv", grads_v, store)
    pack("gradd", grads_d, store)
    pack("sd1", vae.state_dict(), store)
    pack("sdd1", disc.state_dict(), store)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **store)
    print(name, "loss", float(loss), "loss_d", float(loss_d), "keys", len(store))


def cvae_case(name, B, seed_data):
    """ConditionalVAE (mnist_test/03_measurement_approach/cvae_models.py:7-85) and one iteration of its loop (cvae_train.py:27-47)."""
    config, models = import_from(os.path.join(REF, "mnist_test", "03_measurement_approach"), "config", "cvae_models")
    CONFIG = config.CONFIG
    torch.manual_seed(CONFIG["SEED"])
    vae = models.ConditionalVAE()
    vae.train()
    store = {}
    pack("sd0", vae.state_dict(), store)
    g = torch.Generator().manual_seed(seed_data)
    x = torch.rand(B, 1, 28, 28, generator=g)
    t = F.one_hot(torch.randint(0, 10, (B,), generator=g), 10).float()
    store.update({"in/x": x.numpy(), "in/t": t.numpy()})
    opt = torch.optim.Adam(vae.parameters(), lr=CONFIG["LR"])
    opt.zero_grad()
    torch.manual_seed(999)
    recon_x, mu, logvar = vae(x, t)
    torch.manual_seed(999)
    eps = torch.randn(B, CONFIG["Z_DIM"])
    store["fwd/eps"] = eps.numpy()
    loss_recon = F.binary_cross_entropy(recon_x.view(-1, 784), x.view(-1, 784), reduction='sum')
    loss_kld = -0.5 * torch.sum(1 + logvar - mu.pow(2) - logvar.exp()) * 1.0
    loss = loss_recon + loss_kld
    loss.backward()
    grads = {k: p.grad.clone() for k, p in vae.named_parameters()}
    opt.step()
    pack("fwd", dict(recon_x=recon_x, mu=mu, logvar=logvar, z=mu + eps * torch.exp(0.5 * logvar), loss=loss, recon=loss_recon, kld=loss_kld), store)
    pack("grad", grads, store)
    pack("sd1", vae.state_dict(), store)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **store)
    print(name, "loss", float(loss), "keys", len(store))


def vessel_loss_case(name):
    src = open(os.path.join(REF, "vessel_analysis", "01_train", "train.py")).read()
    fn_node = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "loss_function")
    ns = {"torch": torch, "F": F}
    exec(compile(ast.Module(body=[fn_node], type_ignores=[]), "<reference vessel loss_function>", "exec"), ns)
    ref_loss = ns["loss_function"]
    store = {}
    g = torch.Generator().manual_seed(77)
    for tag, shape, density in (("d10", (2, 1, 48, 80), 0.10), ("d001", (2, 1, 16, 16, 16), 0.004), ("d60", (3, 1, 24, 40), 0.6)):
        x = (torch.rand(*shape, generator=g) < density).float()
        recon_x = torch.rand(*shape, generator=g) * 1.2 - 0.1
        B = shape[0]
        m = torch.randn(B, 12, generator=g); m_mu = torch.randn(B, 12, generator=g); m_logvar = torch.randn(B, 12, generator=g) * 0.5
        mu = torch.randn(B, 128, generator=g); logvar = torch.randn(B, 128, generator=g) * 0.3
        recon_x.requires_grad_(True); m_mu.requires_grad_(True); m_logvar.requires_grad_(True)
        mu.requires_grad_(True); logvar.requires_grad_(True)
        recon, kld, morph, sparsity = ref_loss(recon_x, x, m_mu, m, mu, logvar, m_mu, m_logvar)
        total = recon + 0.5 * kld + morph + 0.3 * sparsity          # train.py:82 (BETA = 0.5, config.py)
        total.backward()
        store.update({f"{tag}/x": x.numpy(), f"{tag}/recon_x": recon_x.detach().numpy(), f"{tag}/m": m.numpy(),
                      f"{tag}/m_mu": m_mu.detach().numpy(), f"{tag}/m_logvar": m_logvar.detach().numpy(),
                      f"{tag}/mu": mu.detach().numpy(), f"{tag}/logvar": logvar.detach().numpy()})
        pack(tag, dict(recon=recon, kld=kld, morph=morph, sparsity=sparsity, total=total,
                       g_recon_x=recon_x.grad, g_m_mu=m_mu.grad, g_m_logvar=m_logvar.grad,
                       g_mu=mu.grad, g_logvar=logvar.grad), store)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **store)
    print(name, "keys", len(store))


def vessel2d_inputs(B, seed):
    """Deterministic synthetic batch in the vessel dataset's format (dataset.py:192-248): binary sparse image, standardised m, one-hot t."""
    g = torch.Generator().manual_seed(seed)
    x = (torch.rand(B, 1, 768, 1280, generator=g) < 0.08).float()
    m = torch.randn(B, 12, generator=g)
    t = F.one_hot(torch.randint(0, 19, (B,), generator=g), 19).float()
    eps = torch.randn(B, 128, generator=g)
    return x, m, t, eps


def vessel2d_case(name, B=4, seed_data=4321):
    """CausalVesselVAE (vessel_analysis/00_core/models.py:9-166).  The module imports torchvision and a bare `config`, neither importable
    here, so the class text up to `class CausalViTVAE` is compiled with those two import lines dropped and CONFIG injected; the loss is the
    reference loss_function text as in vessel_loss_case; total as train_one_epoch composes it (01_train/train.py:82)."""
    src = open(os.path.join(REF, "vessel_analysis", "00_core", "models.py")).read()
    src = src.replace("import torchvision\n", "").replace("from config import CONFIG\n", "")
    ns = {"CONFIG": {"M_DIM": 12, "T_DIM": 19, "Z_DIM": 128}}
    exec(compile(src[:src.index("class CausalViTVAE")], "<reference vessel models.py>", "exec"), ns)
    lsrc = open(os.path.join(REF, "vessel_analysis", "01_train", "train.py")).read()
    fn_node = next(n for n in ast.parse(lsrc).body if isinstance(n, ast.FunctionDef) and n.name == "loss_function")
    lns = {"torch": torch, "F": F}
    exec(compile(ast.Module(body=[fn_node], type_ignores=[]), "<reference vessel loss_function>", "exec"), lns)
    torch.manual_seed(42)
    model = ns["CausalVesselVAE"]()
    model.train()
    store = {}
    pack("sd0", model.state_dict(), store)
    x, m, t, eps = vessel2d_inputs(B, seed_data)
    model.reparameterize = lambda mu, logvar: mu + eps * torch.exp(0.5 * logvar)     # :137-140 with the draw injected
    recon_x, m_hat, mu, logvar, m_mu, m_logvar = model(x, m, t)
    recon, kld, morph, sparsity = lns["loss_function"](recon_x, x, m_hat, m, mu, logvar, m_mu, m_logvar)
    total = recon + 0.5 * kld + morph + 0.3 * sparsity
    total.backward()
    store.update({"in/m": m.numpy(), "in/t": t.numpy(), "in/eps": eps.numpy(), "in/seed": np.array([B, seed_data], dtype=np.int64)})
    pack("in", dict(x=x), store)
    pack("fwd", dict(recon_x=recon_x, m_hat=m_hat, mu=mu, logvar=logvar, m_mu=m_mu, m_logvar=m_logvar, recon=recon, kld=kld, morph=morph,
                     sparsity=sparsity, total=total), store)
    pack("grad", {k: p.grad for k, p in model.named_parameters()}, store)
    pack("sd1", {k: v for k, v in model.state_dict().items() if "running" in k or "num_batches" in k}, store)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **store)
    print(name, "total", float(total), "keys", len(store))
    # eval mode on the same batch, as validate() runs it (vessel_analysis/01_train/train.py:100-133): BatchNorm layers on the running
    # statistics the training forward above has just updated (sd1), no_grad, the same loss composition.  Kept in a file of its own.
    model.eval()
    with torch.no_grad():
        recon_x, m_hat, mu, logvar, m_mu, m_logvar = model(x, m, t)
        recon, kld, morph, sparsity = lns["loss_function"](recon_x, x, m_hat, m, mu, logvar, m_mu, m_logvar)
        total = recon + 0.5 * kld + morph + 0.3 * sparsity
    ev = {"in/seed": np.array([B, seed_data], dtype=np.int64)}
    pack("eval", dict(recon_x=recon_x, m_hat=m_hat, mu=mu, logvar=logvar, m_mu=m_mu, m_logvar=m_logvar, recon=recon, kld=kld, morph=morph,
                      sparsity=sparsity, total=total), ev)
    np.savez_compressed(os.path.join(OUT, name + "_eval.npz"), **ev)
    print(name + "_eval", "total", float(total), "keys", len(ev))


def main():
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == "vessel2d":
        vessel2d_case("vessel2d_b4")
        return
    if len(sys.argv) > 1 and sys.argv[1] == "cvae":
        sys.modules.setdefault("torchvision", types.ModuleType("torchvision"))
        cvae_case("mnist_cvae_b8", 8, 1234)
        return
    sys.modules.setdefault("torchvision", types.ModuleType("torchvision"))   # unused import in some ref files
    bio2d_case("bio2d_b4_64x96", 4, 64, 96, 1234)       # non-identity bilinear + non-divisible adaptive pool (6 -> 4)
    bio2d_case("bio2d_b2_64x64", 2, 64, 64, 1235)       # identity resize, identity pool
    bio2d_case("bio2d_b3_128x160", 3, 128, 160, 1236)   # 8x10 -> 4x4 pool (mixed window sizes), 2x/2.5x upsample
    morph_case("morph12_b8", 8, 1234, gaussian_head=False)
    morph_case("morph12g_b8", 8, 1234, gaussian_head=True)
    cvae_case("mnist_cvae_b8", 8, 1234)
    vessel_loss_case("vessel_loss")
    vessel2d_case("vessel2d_b4")          # B = 4: at B = 2 a train-mode BatchNorm1d backward is (g1 - g2)(1 - xhat^2) ~ eps/var, pure cancellation


if __name__ == "__main__":
    main()
