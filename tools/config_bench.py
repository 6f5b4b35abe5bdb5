#!/usr/bin/env python3
"""Step time of the BASELINE.json parity configurations and the 2D vessel model on one MI355X (informational; bench.py stays the metric).

    python tools/config_bench.py        # prints samples/s for: MNIST bf16 B=1024, 3D 64^3 fp32 B=16, 2D vessel 768x1280 bf16 B=8"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from causal_vae_amd import FusedAdam
from causal_vae_amd.causal_cascade import CausalBioVAE3D, train_step
from causal_vae_amd.mnist_baseline import CausalMorphVAE12, LatentDiscriminator
from causal_vae_amd.mnist_baseline import train_step as mnist_step
from causal_vae_amd.vessel import CausalVesselVAE, train_step as vessel_step

DEV = "cuda"


def timeit(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


g = torch.Generator().manual_seed(0)
# configs[1]: MNIST bf16, batch 1024
B = 1024
x, m = torch.rand(B, 1, 28, 28, generator=g).to(DEV), torch.rand(B, 12, generator=g).to(DEV)
t = torch.nn.functional.one_hot(torch.randint(0, 10, (B,), generator=g), 10).float().to(DEV)
vae, disc = CausalMorphVAE12().to(DEV).train().set_compute_dtype(torch.bfloat16), LatentDiscriminator().to(DEV).train()
ov, od = FusedAdam(vae.parameters(), lr=1e-3), FusedAdam(disc.parameters(), lr=1e-3)
dt = timeit(lambda: mnist_step(vae, disc, ov, od, x, m, t))
print(f"MNIST CausalMorphVAE12 bf16 B=1024 (eager)      {dt * 1e3:8.3f} ms/step  {B / dt:10.0f} samples/s")
from causal_vae_amd.graph import GraphedCallable
vae2, disc2 = CausalMorphVAE12().to(DEV).train().set_compute_dtype(torch.bfloat16), LatentDiscriminator().to(DEV).train()
ov2, od2 = FusedAdam(vae2.parameters(), lr=1e-3, device_step=True), FusedAdam(disc2.parameters(), lr=1e-3, device_step=True)
gs = GraphedCallable(lambda: mnist_step(vae2, disc2, ov2, od2, x, m, t))
dt = timeit(gs, n=50)
print(f"MNIST CausalMorphVAE12 bf16 B=1024 (HIP graph)  {dt * 1e3:8.3f} ms/step  {B / dt:10.0f} samples/s")
# configs[2]: 3D 64^3 fp32, batch 16
B = 16
x, m = torch.randn(B, 1, 64, 64, 64, generator=g).to(DEV), torch.rand(B, 12, generator=g).to(DEV)
t = torch.randint(0, 19, (B,), generator=g).to(DEV)
model = CausalBioVAE3D().to(DEV).train()
opt = FusedAdam(model.parameters(), lr=1e-4)
dt = timeit(lambda: train_step(model, opt, x, m, t))
print(f"3D CausalBioVAE3D fp32 64^3 B=16 (eager)          {dt * 1e3:8.3f} ms/step  {B / dt:10.0f} samples/s")
del model, opt
# 2D vessel model (reference batch 8)
B = 8
x = (torch.rand(B, 1, 768, 1280, generator=g) < 0.08).float().to(DEV)
m = torch.randn(B, 12, generator=g).to(DEV)
t = torch.nn.functional.one_hot(torch.randint(0, 19, (B,), generator=g), 19).float().to(DEV)
for dtype in (torch.bfloat16, torch.float32):
    model = CausalVesselVAE().to(DEV).train().set_compute_dtype(dtype)
    opt = FusedAdam(model.parameters(), lr=1e-4)
    dt = timeit(lambda: vessel_step(model, opt, x, m, t), n=5, warm=2)
    print(f"2D CausalVesselVAE {str(dtype).split('.')[-1]:8s} 768x1280 B=8 (eager) {dt * 1e3:8.3f} ms/step  {B / dt:10.1f} samples/s")
    del model, opt
