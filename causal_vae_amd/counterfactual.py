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
def batched_counterfactual(model, z, m, features, values, size=None, fp8_plan=None, precision=None):
    """Decode every intervention do(m_f = v) in one call.  Works with CausalBioVAE / CausalBioVAE3D (decode(z, m, size)) and
    CausalMorphVAE12 (decode(m, z)).  Returns [B, n_features, n_values, C, (D,) H, W].

    precision — what the sweep is FOR decides it.  A sweep is read through differences decode(z, m') - decode(z, m); a low-precision decode
    carries an error of eps * |output| in every row, so the error relative to an effect of size f * |output| is eps / f
    (tools/effect_probe.py, tests/test_hip_models.py::test_counterfactual_effect_is_preserved):
      "fp32" (default)  exact-fp32 MFMA: effect error ~3e-6 at any effect size — the reference-exact sweep;
      "bf16"            eps ~ 2e-3: 5 % effect error at f = 4 %, 40 % at f = 0.5 % (an untrained model): for effects of a few per cent and up;
      "fp8"             (fp8_plan from model.calibrate_fp8_decoder; BASELINE.json configs[4]) 16 % effect error at best (3 mantissa bits on the
                        effect itself), 70 % at f = 0.5 %: visual sweeps and throughput runs, not effect measurements;
      "model"           whatever model.set_compute_dtype chose.
    Passing fp8_plan without a precision means "fp8"."""
    z_rep, m_cf = sweep_inputs(z, m, features, values)
    if precision is None:
        precision = "fp8" if fp8_plan is not None else "fp32"
    if precision not in ("fp32", "bf16", "fp8", "model"):
        raise ValueError(f"precision {precision!r}: expected 'fp32', 'bf16', 'fp8' or 'model'")
    if precision == "fp8" and fp8_plan is None:
        raise ValueError("precision 'fp8' needs fp8_plan = model.calibrate_fp8_decoder(...)")
    stacks = [s_ for s_ in (getattr(model, "enc_conv", None), getattr(model, "dec_conv", None)) if s_ is not None and hasattr(s_, "compute_dtype")]
    prev = [s_.compute_dtype for s_ in stacks]
    want = {"fp32": torch.float32, "bf16": torch.bfloat16}.get(precision)
    try:
        if want is not None and hasattr(model, "set_compute_dtype") and any(p_ != want for p_ in prev):
            model.set_compute_dtype(want)
        if hasattr(model, "dec_input"):
            out = model.decode(z_rep, m_cf, size, fp8_plan=fp8_plan) if precision == "fp8" else model.decode(z_rep, m_cf, size)
        else:
            out = model.decode(m_cf, z_rep)
    finally:
        if want is not None and hasattr(model, "set_compute_dtype") and prev and any(p_ != want for p_ in prev):
            model.set_compute_dtype(prev[0])
    return out.view(z.shape[0], len(features), len(values), *out.shape[1:])
