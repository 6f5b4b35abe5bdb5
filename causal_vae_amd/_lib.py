"""ctypes binding of libcvae_hip.so — the C ABI declared in include/cvae_hip.h.

There is NO fallback: importing this module without the built library raises, and every op checks the
device of its tensors — the product path either runs the gfx950 kernels or fails loudly.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# CVAE_HIP_LIB (development only): another build of the same sources, e.g. with different tuning constants (make BUILD=build_x LIB=... EXTRA=-D...),
# for A/B runs in one GPU session; it must export the same C ABI (checked symbol by symbol below).
LIB_PATH = os.environ.get("CVAE_HIP_LIB") or os.path.join(_HERE, "libcvae_hip.so")

F32, BF16, FP8 = 0, 1, 2
ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_LEAKY02 = 0, 1, 2, 3
_ACT = {None: ACT_NONE, "none": ACT_NONE, "relu": ACT_RELU, "sigmoid": ACT_SIGMOID, "leaky02": ACT_LEAKY02}


class CvaeError(RuntimeError):
    pass


if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "(or `make -C causal_vae_amd/csrc`).  causal_vae_amd has no CPU / eager fallback.")
lib = C.CDLL(LIB_PATH)

_p, _i64, _i, _f, _sz, _u64 = C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_size_t, C.c_uint64

# name -> argtypes (restype is int unless listed in _RESTYPE); mirrors include/cvae_hip.h one to one.
SIGNATURES = {
    "cvae_version": [],
    "cvae_strerror": [_i],
    "cvae_ncs_to_nsc": [_p, _p, _i64, _i64, _i64, _i, _i, _p],
    "cvae_nsc_to_ncs": [_p, _p, _i64, _i64, _i64, _i, _i, _p],
    "cvae_cast": [_p, _p, _i64, _i, _i, _p],
    "cvae_copy_panel": [_p, _p, _i64, _i64, _i64, _i64, _i64, _p],
    "cvae_copy_panels": [_p, _p, _p, _i, _p, _i64, _i64, _i64, _i, _p],
    "cvae_onehot_panel": [_p, _p, _i64, _i64, _i64, _i64, _p],
    "cvae_conv_packed_weight_bytes": [_i64, _i64, _i, _i],
    "cvae_conv_pack_weight": [_p, _p, _i64, _i64, _i, _i, _i, _p],
    "cvae_conv_pack_weights": [_p, _p, _p, _p, _p, _i, _i, _i, _p],
    "cvae_conv_pack_weight_pairs": [_p, _p, _p, _p, _p, _i, _i, _i, _p],
    "cvae_conv_data_workspace_bytes": [_i64] * 9 + [_i, _i],
    "cvae_conv_down": [_p, _p, _p, _p, _p] + [_i64] * 9 + [_i, _i, _i, _p, _sz, _p],
    "cvae_conv_up": [_p, _p, _p, _p, _p] + [_i64] * 9 + [_i, _i, _i, _p, _sz, _p],
    "cvae_quantize_fp8": [_p, _i, _p, _i64, _f, _p],
    "cvae_conv_pack_weight_fp8": [_p, _p, _i64, _i64, _i, _i, _f, _p],
    "cvae_conv_up_fp8": [_p, _p, _p, _p, _i, _f, _f] + [_i64] * 9 + [_i, _i, _p],
    "cvae_conv_up_c1_fp8in": [_p, _p, _p, _p, _f, _i64, _i64, _i64, _i64, _i64, _i, _i, _p],
    "cvae_quantize_fp8_dev": [_p, _i, _p, _i64, _p, _p, _p],
    "cvae_absmax": [_p, _i, _i64, _p, _p],
    "cvae_conv_pack_weights_fp8": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _p],
    "cvae_conv_fp8": [_i, _p, _p, _p, _p, _i, _p, _p, _f, _f, _p] + [_i64] * 9 + [_i, _i, _p, _sz, _i, _p, _p],
    "cvae_conv_down_bits": [_p, _p, _p, _p, _p, _p] + [_i64] * 9 + [_i, _i, _i, _p, _sz, _p],
    "cvae_conv_up_bits": [_p, _p, _p, _p, _p, _p] + [_i64] * 9 + [_i, _i, _i, _p, _sz, _p],
    "cvae_fp8_scale_update": [_p, _p, _p, _i, _f, _p, _p, _p, _i, _p, _p, _p],
    "cvae_conv_pack_weight_pairs_f8": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p],
    "cvae_conv_down_image_f8": [_p, _i, _p, _p, _p, _p, _p, _p, _p] + [_i64] * 8 + [_i, _i, _p],
    "cvae_conv_image_supported": [_p, _i64, _i, _i],
    "cvae_conv_down_image": [_p, _i, _p, _p, _p, _p] + [_i64] * 8 + [_i, _i, _i, _p],
    "cvae_conv_wgrad_image": [_p, _p, _i, _p, _p, _p, _sz] + [_i64] * 8 + [_i, _i, _p],
    "cvae_conv_wgrad_workspace_bytes": [_i64, _i64, _i],
    "cvae_conv_wgrad": [_p, _p, _p, _p, _i, _p, _sz] + [_i64] * 9 + [_i, _i, _p],
    "cvae_conv_wgrad_multi": [_i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _p],
    "cvae_channel_sum_workspace_bytes": [_i64, _i64, _i],
    "cvae_channel_sum": [_p, _p, _i64, _i64, _i, _p, _sz, _p],
    "cvae_act_fwd": [_p, _p, _i64, _i, _i, _p],
    "cvae_act_bwd": [_p, _p, _p, _i64, _i, _i, _p],
    "cvae_adaptive_avgpool_fwd": [_p, _p] + [_i64] * 9 + [_i, _p],
    "cvae_adaptive_avgpool_bwd": [_p, _p, _p] + [_i64] * 9 + [_i, _p],
    "cvae_upsample_linear_fwd": [_p, _p] + [_i64] * 8 + [_i, _p],
    "cvae_upsample_linear_bwd": [_p, _p] + [_i64] * 8 + [_i, _p],
    "cvae_linear_workspace_bytes": [_i64, _i64, _i64, _i],
    "cvae_linear_fwd": [_p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _i, _p, _sz, _p],
    "cvae_linear_bwd_data": [_p, _p, _p, _i64, _i64, _i64, _i64, _i64, _p, _i, _p, _sz, _p],
    "cvae_linear_bwd_weight": [_p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _p, _i, _p, _sz, _p],
    "cvae_conv_up_variant": [_p, _p, _p, _p, _p] + [_i64] * 9 + [_i, _i, _i, _p, _sz, _i, _i, _i64, _p],
    "cvae_conv_down_variant": [_p, _p, _p, _p, _p] + [_i64] * 9 + [_i, _i, _i, _p, _sz, _i, _p],
    "cvae_linear_fwd_bf16": [_p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _i, _p, _sz, _p],
    "cvae_linear_bwd_data_bf16": [_p, _p, _p, _i64, _i64, _i64, _i64, _i64, _p, _sz, _p],
    "cvae_small_dense_supported": [_i64, _i64, _i64],
    "cvae_small_dense_workspace_bytes": [_i64, _i64, _i64],
    "cvae_small_dense_fwd": [_p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _i, _p],
    "cvae_small_dense_bwd_data": [_p, _p, _p, _p, _i, _p, _i, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _p],
    "cvae_small_dense_bwd_weight": [_p, _p, _p, _p, _p, _i, _i64, _i64, _i64, _i64, _i64, _i64, _p, _sz, _p],
    "cvae_linear_bwd_data_inact": [_p, _p, _p, _i64, _i64, _i64, _i64, _i64, _p, _i64, _i, _i, _p, _sz, _p],
    "cvae_linear_bwd_weight_bf16": [_p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _p, _sz, _p],
    "cvae_bn1d_train_fwd": [_p] * 8 + [_i64, _i64, _f, _f, _p],
    "cvae_bn1d_train_bwd": [_p] * 8 + [_i64, _i64, _p],
    "cvae_bn1d_eval_fwd": [_p] * 6 + [_i64, _i64, _f, _p],
    "cvae_bn1d_stats": [_p, _p, _f, _p, _i64, _i64, _p],
    "cvae_bn1d_apply_stats": [_p] * 5 + [_f] + [_p] * 5 + [_i64, _i64, _f, _f, _p],
    "cvae_bn1d_bwd_sums": [_p] * 5 + [_i64, _i64, _p],
    "cvae_bn1d_bwd_apply": [_p] * 6 + [_f, _p, _i64, _i64, _p],
    "cvae_philox_normal": [_p, _i64, _u64, _u64, _u64, _p, _p],
    "cvae_philox_normal_advance": [_p, _i64, _u64, _u64, _u64, _p, _p],
    "cvae_reduce_workspace_bytes": [],
    "cvae_reparam_kld_fwd": [_p, _p, _p, _p, _p, _i64, _p, _sz, _p],
    "cvae_reparam_kld_bwd": [_p, _p, _f] + [_p] * 5 + [_i64, _p],
    "cvae_latent_head_fwd": [_p, _p, _p, _p, _p, _p, _i64, _i64, _p],
    "cvae_latent_head_bwd": [_p, _p, _p, _p, _p, _p, _p, _i64, _i64, _p],
    "cvae_sse_fwd": [_p, _p, _p, _i64, _p, _sz, _p],
    "cvae_sse_bwd": [_p, _p, _p, _f, _p, _i64, _p],
    "cvae_combine3": [_p, _f, _f, _p],
    "cvae_bce_fwd": [_p, _p, _p, _i64, _p, _sz, _p],
    "cvae_bce_bwd": [_p, _p, _p, _p, _i64, _p],
    "cvae_sum_fwd": [_p, _p, _i64, _p, _sz, _p],
    "cvae_wmse_sparsity_fwd": [_p, _p, _p, _p, _i64, _i64, _p, _sz, _p],
    "cvae_wmse_sparsity_bwd": [_p] * 6 + [_i64, _i64, _p],
    "cvae_gauss_nll_fwd": [_p, _p, _p, _p, _i64, _p, _sz, _p],
    "cvae_gauss_nll_bwd": [_p] * 6 + [_i64, _p],
    "cvae_softmax_ce_fwd": [_p, _p, _p, _i64, _i64, _p, _sz, _p],
    "cvae_softmax_ce_bwd": [_p, _p, _p, _p, _i64, _i64, _p],
    "cvae_uniform_kl_fwd": [_p, _p, _i64, _i64, _p, _sz, _p],
    "cvae_uniform_kl_bwd": [_p, _p, _p, _i64, _i64, _p],
    "cvae_adam_step": [_p, _p, _p, _p, _i64, _f, _f, _f, _f, _f, _f, _p, _p],
    "cvae_adam_multi": [_p, _p, _p, _p, _p, _i, _f, _f, _f, _f, _f, _f, _p, _p, _p],
    "cvae_multi_copy": [_p, _p, _p, _i, _p],
    "cvae_counter_add": [_p, _i, _p],
    "cvae_sqnorm": [_p, _p, _i64, _p, _sz, _p],
    "cvae_sqnorm_multi": [_p, _p, _i, _p, _p, _sz, _p],
    "cvae_weighted_sum": [_p, _p, _i, _p, _p, _i, _p],
    "cvae_scale": [_p, _i64, _p, _p],
    "cvae_clip_coef": [_p, _p, _f, _p],
    "cvae_up2x_supported": [_i64] * 7,
    "cvae_up2x_fwd": [_p, _p] + [_i64] * 7 + [_i, _p],
    "cvae_elbo_up2x_partials": [_i64] * 4,
    "cvae_elbo_up2x_fwd": [_p] * 6 + [_f, _p, _p, _p, _p, _p] + [_i64] * 9 + [_i, _p],
    "cvae_elbo_up2x_bwd": [_p] * 5 + [_f] + [_p] * 5 + [_i64] * 9 + [_i, _p],
    "cvae_conv3_to_k4": [_p, _p, _i64, _i64, _p],
    "cvae_k4_to_conv3_grad": [_p, _p, _i64, _i64, _p],
    "cvae_clamp_fwd": [_p, _p, _f, _f, _i64, _p],
    "cvae_clamp_bwd": [_p, _p, _p, _f, _f, _i64, _p],
    "cvae_bn2d_workspace_bytes": [_i64, _i64],
    "cvae_bn2d_fwd": [_p] * 8 + [_i64, _i64, _f, _f, _i, _i, _i, _p, _sz, _p],
    "cvae_bn2d_bwd": [_p] * 9 + [_i64, _i64, _i, _i, _p, _sz, _p],
    "cvae_bottleneck_sizes": [_p, _p, _p, _p, _p, _p],
    "cvae_bottleneck_fwd": [_p] * 10 + [_f, _f, _i, _p, _p, _p, _p, _p, _i, _p],
    "cvae_bottleneck_bwd": [_p] * 12 + [_i, _p, _p, _p, _p, _i, _p],
    "cvae_bottleneck_bn_local_stats": [_p] * 5 + [_i64, _i64, _i64, _p],
    "cvae_bottleneck_fwd_sync": [_p] * 10 + [_f, _f, _i, _p, _p, _p, _p, _p, _i, _p, _i, _p],
    "cvae_bottleneck_fwd_ex": [_p] * 10 + [_f, _f, _i, _p, _p, _p, _p, _p, _i, _p, _i, _p, _p],
    "cvae_bottleneck_bwd_sync": [_p] * 12 + [_i, _p, _p, _p, _p, _i, _p, _p, _p],
    "cvae_bottleneck_bn_bwd_finish": [_p] * 7 + [_i, _p],
}
_RESTYPE = {"cvae_strerror": C.c_char_p, "cvae_conv_packed_weight_bytes": _sz, "cvae_conv_wgrad_workspace_bytes": _sz,
            "cvae_conv_data_workspace_bytes": _sz, "cvae_elbo_up2x_partials": _i64, "cvae_channel_sum_workspace_bytes": _sz,
            "cvae_linear_workspace_bytes": _sz, "cvae_reduce_workspace_bytes": _sz, "cvae_bn2d_workspace_bytes": _sz,
            "cvae_small_dense_workspace_bytes": _sz}

for _name, _args in SIGNATURES.items():
    _fn = getattr(lib, _name)          # AttributeError here = header and library disagree: fail at import
    _fn.argtypes = _args
    _fn.restype = _RESTYPE.get(_name, _i)


class BottleneckDims(C.Structure):
    """cvae_bottleneck_dims"""
    _fields_ = [(n, _i64) for n in ("M", "D", "H", "W", "C", "OD", "OH", "OW", "m_dim", "t_dim", "N1", "N2", "Z", "HM")]


BOTTLENECK_PARAMS = ("W1", "b1", "W2", "b2", "Wmu", "bmu", "Wlv", "blv", "Wm0", "bm0", "gamma", "beta", "Wm3", "bm3", "Wm5", "bm5", "Wd", "bd")
BOTTLENECK_SAVED = ("h1", "h2", "mu", "logvar", "xhat", "invstd", "a1n", "a2", "m_hat", "zm")


class BottleneckPtrs18(C.Structure):
    """cvae_bottleneck_params / cvae_bottleneck_grads: 18 device pointers in BOTTLENECK_PARAMS order"""
    _fields_ = [(n, _p) for n in BOTTLENECK_PARAMS]


class BottleneckSaved(C.Structure):
    """cvae_bottleneck_saved"""
    _fields_ = [(n, _p) for n in BOTTLENECK_SAVED]


class BottleneckNoise(C.Structure):
    """cvae_bottleneck_noise"""
    _fields_ = [("seed", _u64), ("subsequence", _u64), ("call_counter", _p)]


class KernelTimer:
    """Optional HIP-event timing of individual C-ABI launches (bench.py's roofline leg).  Events are recorded on torch's
    current stream — the stream the kernels are enqueued on — and only read after a device sync, so timing adds no sync."""

    def __init__(self):
        self.events = {}

    def run(self, name, fn, *args):
        s = torch.cuda.Event(enable_timing=True)
        e = torch.cuda.Event(enable_timing=True)
        s.record()
        rc = fn(*args)
        e.record()
        self.events.setdefault(name, []).append((s, e))
        return rc

    def summary(self):
        """name -> (launches, mean milliseconds).  Call after torch.cuda.synchronize()."""
        return {k: (len(v), sum(a.elapsed_time(b) for a, b in v) / len(v)) for k, v in self.events.items()}


TIMER = None          # set to a KernelTimer to time conv launches


def timed(name, fn, *args):
    if TIMER is None:
        return fn(*args)
    return TIMER.run(name, fn, *args)


def strerror(code):
    return lib.cvae_strerror(code).decode()


def check(rc, what):
    if rc != 0:
        raise CvaeError(f"{what} failed: {strerror(rc)} (code {rc})")


def dtype_code(dt):
    if dt == torch.float32:
        return F32
    if dt == torch.bfloat16:
        return BF16
    raise CvaeError(f"unsupported compute dtype {dt}; the gfx950 kernels take float32 or bfloat16")


def act_code(act):
    return _ACT[act]


def ptr(t):
    return None if t is None else t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise CvaeError("causal_vae_amd runs on MI355X only: got a CPU tensor (there is no CPU fallback; "
                            "move the model and its inputs to 'cuda')")
