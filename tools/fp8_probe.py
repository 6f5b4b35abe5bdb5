"""GPU probe: error of the bf16 and fp8 (e4m3) counterfactual decodes against the fp32 HIP decode on a 240-row sweep (numbers behind the bounds in
tests/test_hip_models.py::test_fp8_and_bf16_sweep_decode_close_to_fp32_decode)."""
import torch
from causal_vae_amd.causal_cascade import CausalBioVAE3D
from causal_vae_amd.counterfactual import sweep_inputs

dev = torch.device("cuda")
torch.manual_seed(42)
model = CausalBioVAE3D().to(dev).eval()
g = torch.Generator().manual_seed(8)
z, m = torch.randn(4, 64, generator=g).to(dev), torch.rand(4, 12, generator=g).to(dev)
z_rep, m_cf = sweep_inputs(z, m, list(range(12)), [0.0, 0.25, 0.5, 0.75, 1.0])
x = (torch.rand(240, 1, 64, 64, 64, generator=g) < 0.1).float().to(dev)
with torch.no_grad():
    ref = model.decode(z_rep, m_cf)
    model.set_compute_dtype(torch.bfloat16)
    b16 = model.decode(z_rep, m_cf)
    for head in (1.0, 2.0):
        plan = model.calibrate_fp8_decoder(z_rep[:60], m_cf[:60], headroom=head)
        f8 = model.decode(z_rep, m_cf, fp8_plan=plan)
        rel = lambda a: float(((a - ref).norm() / ref.norm()).item())
        sse = lambda a: ((a - x) ** 2).flatten(1).sum(1)
        print("headroom", head, "rel-L2 bf16", rel(b16), "fp8", rel(f8), "fp8 rows 60+ (not calibrated on)", float(((f8[60:] - ref[60:]).norm() / ref[60:].norm()).item()))
        print("  max abs bf16", float((b16 - ref).abs().max()), "fp8", float((f8 - ref).abs().max()), "ref range", float(ref.min()), float(ref.max()))
        print("  per-row SSE rel diff: bf16 max", float(((sse(b16) - sse(ref)).abs() / sse(ref)).max()), "fp8 max", float(((sse(f8) - sse(ref)).abs() / sse(ref)).max()))
        print("  scales", [(e["index"], e["sx"], e.get("sw")) for e in plan])
        # the sweep's own signal: difference between the two extreme values of a feature, per row pair
        d_ref = ref.view(4, 12, 5, -1)[:, :, 4] - ref.view(4, 12, 5, -1)[:, :, 0]
        d_f8 = f8.view(4, 12, 5, -1)[:, :, 4] - f8.view(4, 12, 5, -1)[:, :, 0]
        d_b16 = b16.view(4, 12, 5, -1)[:, :, 4] - b16.view(4, 12, 5, -1)[:, :, 0]
        print("  intervention effect (v=1 minus v=0) rel-L2: bf16", float(((d_b16 - d_ref).norm() / d_ref.norm()).item()), "fp8", float(((d_f8 - d_ref).norm() / d_ref.norm()).item()),
              "effect/recon norm", float((d_ref.norm() / ref.norm()).item()))
