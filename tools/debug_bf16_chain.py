#!/usr/bin/env python3
"""Where does the bf16 build leave the bf16-rounding oracle?  Compares, layer by layer, the activation gradients of one 64^3 step
(product hooks vs oracle hooks) and the single-channel kernels against torch on bf16-rounded operands."""
import os, sys
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import oracle
from oracle import functional as ofn
from causal_vae_amd import ops
from causal_vae_amd.causal_cascade import CausalBioVAE3D

DEV = "cuda"
rl2 = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
bf = lambda t: t.bfloat16().float()

# ---- unit: down_c1 (dec4 backward-data) and up_c1 (dec4 forward) on bf16 operands ----
g = torch.Generator().manual_seed(0)
w = torch.randn(32, 1, 4, 4, 4, generator=g) * 0.1
gy = bf(torch.randn(2, 1, 64, 64, 64, generator=g))
ref = F.conv3d(gy, bf(w), None, stride=2, padding=1)                      # = convT backward-data for weight [32][1][...]
out = ops._conv_down(gy.view(2, 64, 64, 64, 1).to(DEV).bfloat16(), w.to(DEV), None, None, 32, 3, None)
print("down_c1 vs fp32 math on bf16 operands (before output rounding):", rl2(out.float().cpu().permute(0, 4, 1, 2, 3), ref), " vs rounded ref:", rl2(out.float().cpu().permute(0, 4, 1, 2, 3), bf(ref)))
xs = bf(torch.randn(2, 32, 32, 32, 32, generator=g))
ref = F.conv_transpose3d(xs, bf(w), None, stride=2, padding=1)
out = ops._conv_up(xs.permute(0, 2, 3, 4, 1).contiguous().to(DEV).bfloat16(), w.to(DEV), None, None, 1, 3, None)
print("up_c1 vs rounded ref:", rl2(out.float().cpu().permute(0, 4, 1, 2, 3), bf(ref)))

# ---- chain: activation gradients ----
B, size = 2, 64
g = torch.Generator().manual_seed(1234)
x = torch.randn(B, 1, size, size, size, generator=g)
m, t, eps = torch.rand(B, 12, generator=g), torch.randint(0, 19, (B,), generator=g), torch.randn(B, 64, generator=g)
sd = oracle.init_state_dict("bio3d", seed=42)
leaves = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone()) for k, v in sd.items()}
o = ofn.bio_vae_forward(leaves, x, m, t, eps, nd=3, keep_acts=True, conv_dtype=torch.bfloat16)
ograd = {}
for k, a in o["acts"].items():
    if a.requires_grad:
        a.register_hook(lambda gr, k=k: ograd.__setitem__(k, gr.clone()))
loss = ofn.cascade_loss(o["recon_x"], x, o["m_hat"], m, o["mu"], o["logvar"], 2000.0)[0]
loss.backward()
torch.manual_seed(42)
model = CausalBioVAE3D().to(DEV).train().set_compute_dtype(torch.bfloat16)
pacts, pgrad = {}, {}
import causal_vae_amd.layers as hl
orig = hl._ConvBase.forward_cl
names = iter(["enc1", "enc2", "enc3", "enc4", "dec1", "dec2", "dec3", "dec4"])
def patched(self, x_cl, *a, **k):
    y = orig(self, x_cl, *a, **k)
    n = next(names)
    pacts[n] = y
    if y.requires_grad:
        y.register_hook(lambda gr, n=n: pgrad.__setitem__(n, gr.clone()))
    return y
hl._ConvBase.forward_cl = patched
out = model(x.to(DEV), m.to(DEV), t.to(DEV), eps=eps.to(DEV))
from causal_vae_amd.causal_cascade import loss_function
loss_function(out[0], x.to(DEV), out[1], m.to(DEV), out[2], out[3])[0].backward()
for n in ["dec4", "dec3", "dec2", "dec1", "enc4", "enc3", "enc2", "enc1"]:
    pa = pacts[n].float().cpu().permute(0, 4, 1, 2, 3)
    oa = o["acts"][n].detach()
    line = f"{n}: activation rel-L2 {rl2(pa, oa):.2e}"
    if n in pgrad and n in ograd:
        pg = pgrad[n].float().cpu().permute(0, 4, 1, 2, 3)
        og = ograd[n]                                         # gradient arriving at the rounded activation (fp32, before rounding / masking)
        relu = n != "dec4"
        og_m = bf(og) * ((oa > 0).float() if relu else 1.0)   # what the product stores: rounded, and masked by the producer's ReLU
        pg_m = pg * ((pa > 0).float() if relu else 1.0)
        line += f" | gradient (rounded + masked) rel-L2 {rl2(pg_m, og_m):.2e}  sum ratio {float(pg_m.double().sum() / og_m.double().sum()):.6f}"
    print(line)
