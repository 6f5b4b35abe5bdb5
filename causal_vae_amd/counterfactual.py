"""Batched counterfactual sweeps.

The reference intervenes on one morphology feature at a time and decodes every (feature, value) pair with its own
batch-1 forward (vessel_analysis/04_generate_counterfactual/generate_counterfactual.py:77-99;
mnist_test/01_baseline_causal_vae/check_mnist_counterfactual.py:83-101).  Abduction gives one z per sample; every intervened
m' shares it, so the whole sweep is one [B * n_features * n_values, Z + M] decode.
"""
import torch


def sweep_inputs(z, m, features, values):
    """z [B, Z], m [B, M] -> (z_rep, m_cf) with rows ordered (sample, feature, value); m_cf[., f] is set to the value."""
    B, M = m.shape
    F, V = len(features), len(values)
    m_cf = m[:, None, None, :].expand(B, F, V, M).clone()
    vals = torch.as_tensor(values, dtype=m.dtype, device=m.device)
    for i, f in enumerate(features):
        m_cf[:, i, :, f] = vals
    z_rep = z[:, None, None, :].expand(B, F, V, z.shape[1]).reshape(B * F * V, -1).contiguous()
    return z_rep, m_cf.reshape(B * F * V, M).contiguous()


@torch.no_grad()
def batched_counterfactual(model, z, m, features, values, size=None, fp8_plan=None):
    """Decode every intervention do(m_f = v) in one call.  Works with CausalBioVAE / CausalBioVAE3D (decode(z, m, size)) and
    CausalMorphVAE12 (decode(m, z)).  Returns [B, n_features, n_values, C, (D,) H, W].  fp8_plan (CausalBioVAE only, from
    model.calibrate_fp8_decoder): the fp8 (e4m3) conv path of BASELINE.json configs[4]."""
    z_rep, m_cf = sweep_inputs(z, m, features, values)
    if hasattr(model, "dec_input"):
        out = model.decode(z_rep, m_cf, size, fp8_plan=fp8_plan) if fp8_plan is not None else model.decode(z_rep, m_cf, size)
    else:
        out = model.decode(m_cf, z_rep)
    return out.view(z.shape[0], len(features), len(values), *out.shape[1:])
