import torch, sys
sys.path.insert(0, '.')
from causal_vae_amd import FusedAdam
from causal_vae_amd.causal_cascade import CausalBioVAE3D
m = CausalBioVAE3D(img_channels=1, m_dim=12, t_dim=19, latent_dim=64).cuda()
n = sum(p.numel() for p in m.parameters())
opt = FusedAdam(m.parameters(), lr=1e-3, device_step=True)
for p in m.parameters(): p.grad = torch.randn_like(p) * 1e-3
big = torch.empty(512 << 20, dtype=torch.uint8, device='cuda')
for _ in range(3): opt.step()
ts = []
for _ in range(20):
    big.zero_()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); opt.step(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
ts.sort()
print("params", n, "adam us median", ts[10], "min", ts[0], "GB/s", n * 28 / ts[10] / 1e3)
