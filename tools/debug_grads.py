import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import oracle
from causal_vae_amd import FusedAdam
from causal_vae_amd.causal_cascade import CausalBioVAE3D, loss_function
B, size = 2, int(sys.argv[1]) if len(sys.argv) > 1 else 32
g = torch.Generator().manual_seed(1234)
x = torch.randn(B, 1, size, size, size, generator=g); m = torch.rand(B, 12, generator=g)
t = torch.randint(0, 19, (B,), generator=g); eps = torch.randn(B, 64, generator=g)
print("t =", t.tolist())
sd = oracle.init_state_dict("bio3d", seed=42)
st = oracle.cascade_train_step(sd, x, m, t, eps, nd=3, apply_update=False)
torch.manual_seed(42)
model = CausalBioVAE3D().cuda().train()
out = model(x.cuda(), m.cuda(), t.cuda(), eps=eps.cuda())
loss = loss_function(out[0], x.cuda(), out[1], m.cuda(), out[2], out[3])[0]
loss.backward()
for k in ("recon_x", "m_hat", "mu", "logvar"):
    a, b = dict(zip(("recon_x", "m_hat", "mu", "logvar"), out))[k].detach().cpu(), st["outputs"][k]
    print(f"{k:10s} max|ref| {float(b.abs().max()):.3e} max err {float((a-b).abs().max()):.3e}")
for k, p in model.named_parameters():
    gr = st["grads"][k]; e = (p.grad.cpu() - gr).abs()
    print(f"{k:28s} max|ref| {float(gr.abs().max()):.3e} mean|ref| {float(gr.abs().mean()):.3e} max err {float(e.max()):.3e} at {tuple(int(v) for v in (e == e.max()).nonzero()[0])}")
