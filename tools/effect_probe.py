"""GPU probe: how well do the fp32 / bf16 / fp8 decodes preserve the INTERVENTION EFFECT decode(z, m') - decode(z, m) of a counterfactual sweep
(vessel_analysis/04_generate_counterfactual/generate_counterfactual.py:77-99) against the CPU oracle, as a function of how large the effect is
relative to the output (dec_input's m-columns scaled by `gain`)?  Numbers behind tests/test_hip_models.py::test_counterfactual_effect_*."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import oracle
from causal_vae_amd.causal_cascade import CausalBioVAE3D
from causal_vae_amd.counterfactual import sweep_inputs

dev = torch.device("cuda")
g = torch.Generator().manual_seed(8)
z, m = torch.randn(2, 64, generator=g), torch.rand(2, 12, generator=g)
feats, vals = [0, 5, 11], [0.0, 1.0]
z_rep, m_cf = sweep_inputs(z, m, feats, vals)
for gain in (1.0, 10.0, 30.0, 100.0):
    sd = oracle.init_state_dict("bio3d", seed=42)
    sd["dec_input.weight"][:, 64:] *= gain
    ref = oracle.bio_decode(sd, z_rep, m_cf, nd=3)
    eff = lambda o: (o.view(2, len(feats), 2, -1)[:, :, 1] - o.view(2, len(feats), 2, -1)[:, :, 0])
    e_ref = eff(ref)
    torch.manual_seed(42)
    model = CausalBioVAE3D().to(dev).eval()
    model.load_state_dict({k: v.to(dev) for k, v in sd.items()})
    out = {}
    with torch.no_grad():
        out["fp32"] = model.decode(z_rep.to(dev), m_cf.to(dev)).cpu()
        model.set_compute_dtype(torch.bfloat16)
        out["bf16"] = model.decode(z_rep.to(dev), m_cf.to(dev)).cpu()
        plan = model.calibrate_fp8_decoder(z_rep.to(dev), m_cf.to(dev))
        out["fp8"] = model.decode(z_rep.to(dev), m_cf.to(dev), fp8_plan=plan).cpu()
    print(f"gain {gain}: effect / output norm {float(e_ref.norm() / ref.norm() * (2 ** 0.5)):.4f}")
    for k, o in out.items():
        print(f"   {k}: decode rel-L2 {float((o - ref).norm() / ref.norm()):.3e}   effect rel-L2 {float((eff(o) - e_ref).norm() / e_ref.norm()):.3e}")
