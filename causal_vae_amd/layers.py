"""Drop-in nn.Module layers whose arithmetic runs in libcvae_hip.so.

Each class subclasses the stock torch layer it replaces, so constructor arguments, default initialisation (and therefore
the RNG draws of `torch.manual_seed(42); Model()`), parameter names, shapes and `state_dict()` keys are the reference's;
only `forward` changes.  The containers (ConvStack / DeconvStack / MLP) are `nn.Sequential`s with the reference's child
indices (so `enc_conv.0.weight` etc. interchange with reference checkpoints) whose forward runs the fused channels-last
pipeline: conv+bias+ReLU in one kernel, ReLU masks folded into the neighbouring backward kernels.
"""
import torch
import torch.nn as nn

from . import ops
from ._lib import CvaeError, require_gpu


def _check_k4s2p1(mod, nd):
    ok = (tuple(mod.kernel_size) == (4,) * nd and tuple(mod.stride) == (2,) * nd and tuple(mod.padding) == (1,) * nd
          and tuple(mod.dilation) == (1,) * nd and mod.groups == 1)
    if hasattr(mod, "output_padding"):
        ok = ok and tuple(mod.output_padding) == (0,) * nd
    if getattr(mod, "padding_mode", "zeros") != "zeros" or not ok:
        raise CvaeError(f"{type(mod).__name__}: the gfx950 kernels implement kernel 4 / stride 2 / padding 1 only "
                        f"(got k={mod.kernel_size}, s={mod.stride}, p={mod.padding})")


class _ConvBase:
    """Shared by the four conv classes: `compute_dtype` and the channels-last entry point `forward_cl`."""
    compute_dtype = torch.float32
    _nd = 2
    _fn = ops.ConvDown

    def forward_cl(self, x_cl, act=None, in_is_relu_out=False, grad_premasked=False, packed=None, out_dtype=None, f8=None):
        """f8 (training forward on fp8 operands, causal_vae_amd.fp8): dict(xq, wq, dscale, want_out8, amax); the layer leaves the codes of its
        result in f8["y8"] when asked.  The backward pass is the bf16 one either way."""
        if f8 is not None:
            if self._fn is ops.ConvDown:
                return self._fn.apply(x_cl, self.weight, self.bias, self._nd, act, in_is_relu_out, grad_premasked, packed, out_dtype, f8)
            return self._fn.apply(x_cl, self.weight, self.bias, self._nd, act, in_is_relu_out, grad_premasked, packed, f8)
        if out_dtype is not None:
            return self._fn.apply(x_cl, self.weight, self.bias, self._nd, act, in_is_relu_out, grad_premasked, packed, out_dtype)
        return self._fn.apply(x_cl, self.weight, self.bias, self._nd, act, in_is_relu_out, grad_premasked, packed)

    def forward(self, x):                                   # NC(D)HW fp32 in / out, like the stock layer
        require_gpu(x)
        y = self.forward_cl(ops.ToChannelsLast.apply(x, self.compute_dtype))
        return ops.FromChannelsLast.apply(y, self._nd)


class Conv2d(_ConvBase, nn.Conv2d):
    _nd, _fn = 2, ops.ConvDown

    def __init__(self, *a, **k):
        nn.Conv2d.__init__(self, *a, **k)
        _check_k4s2p1(self, 2)


class Conv3d(_ConvBase, nn.Conv3d):
    _nd, _fn = 3, ops.ConvDown

    def __init__(self, *a, **k):
        nn.Conv3d.__init__(self, *a, **k)
        _check_k4s2p1(self, 3)


class ConvTranspose2d(_ConvBase, nn.ConvTranspose2d):
    _nd, _fn = 2, ops.ConvUp

    def __init__(self, *a, **k):
        nn.ConvTranspose2d.__init__(self, *a, **k)
        _check_k4s2p1(self, 2)


class ConvTranspose3d(_ConvBase, nn.ConvTranspose3d):
    _nd, _fn = 3, ops.ConvUp

    def __init__(self, *a, **k):
        nn.ConvTranspose3d.__init__(self, *a, **k)
        _check_k4s2p1(self, 3)


class Linear(nn.Linear):
    math = None            # torch.bfloat16 (set_linear_math): bf16 MFMA operands for large-batch products; None / float32: exact fp32

    def forward(self, x, act=None, in_act=None, grad_premasked=False):
        require_gpu(x)
        return ops.Linear.apply(x, self.weight, self.bias, act, self.math, in_act, grad_premasked)


class BatchNorm1d(nn.BatchNorm1d):
    sync = False            # True (parallel.convert_sync_batchnorm): train-mode batch statistics span the ranks of `sync_group`
    sync_group = None

    def forward(self, x):
        require_gpu(x)
        if x.dim() != 2:
            raise CvaeError("BatchNorm1d: [B, F] input expected")
        if self.training:
            if x.shape[0] <= 1:           # same error, same wording as torch (reference behaviour)
                raise ValueError(f"Expected more than 1 value per channel when training, got input size {x.size()}")
            if self.momentum is None:
                raise CvaeError("BatchNorm1d: cumulative-average momentum=None is not implemented")
            if self.track_running_stats and self.num_batches_tracked is not None:
                self.num_batches_tracked.add_(1)
            rm = self.running_mean if self.track_running_stats else None
            rv = self.running_var if self.track_running_stats else None
            if self.sync:
                return ops.SyncBatchNorm1dTrain.apply(x, self.weight, self.bias, rm, rv, float(self.momentum), float(self.eps), self.sync_group)
            return ops.BatchNorm1dTrain.apply(x, self.weight, self.bias, rm, rv, float(self.momentum), float(self.eps))
        if x.requires_grad and torch.is_grad_enabled():
            raise CvaeError("BatchNorm1d in eval mode is forward-only here (the reference consumers run it under no_grad)")
        return ops.bn1d_eval(x, self.weight, self.bias, self.running_mean, self.running_var, float(self.eps))


class BatchNorm2d(nn.BatchNorm2d):
    """nn.BatchNorm2d on channels-last tensors; `forward_cl(x_cl, act)` fuses the activation that follows it."""

    def forward_cl(self, x_cl, act=None):
        require_gpu(x_cl)
        if self.training:
            if x_cl.numel() // x_cl.shape[-1] <= 1:
                raise ValueError(f"Expected more than 1 value per channel when training, got input size {tuple(x_cl.shape)}")
            if self.momentum is None:
                raise CvaeError("BatchNorm2d: cumulative-average momentum=None is not implemented")
            if self.track_running_stats and self.num_batches_tracked is not None:
                self.num_batches_tracked.add_(1)
        use_batch = self.training or not self.track_running_stats
        rm = self.running_mean if self.track_running_stats else None
        rv = self.running_var if self.track_running_stats else None
        return ops.BatchNorm2dAct.apply(x_cl, self.weight, self.bias, rm, rv, float(self.momentum or 0.0), float(self.eps), use_batch, act)

    def forward(self, x):                                   # NCHW fp32 in / out, like the stock layer
        require_gpu(x)
        return ops.FromChannelsLast.apply(self.forward_cl(ops.ToChannelsLast.apply(x, torch.float32)), 2)


class UpConv2dK3(nn.Conv2d):
    """The nn.Conv2d(Cin, Cout, 3, 1, 1) of an `nn.Upsample(scale_factor=2, mode='nearest') -> Conv2d` pair.  `forward_up2_cl` runs the
    PAIR (nearest x2, then the conv) as one transposed k4/s2/p1 product on the existing kernels (ops.Conv3ToK4)."""
    compute_dtype = torch.float32

    def __init__(self, *a, **k):
        nn.Conv2d.__init__(self, *a, **k)
        if (tuple(self.kernel_size), tuple(self.stride), tuple(self.padding), tuple(self.dilation), self.groups) != ((3, 3), (1, 1), (1, 1), (1, 1), 1) \
                or self.padding_mode != "zeros":
            raise CvaeError("UpConv2dK3: kernel 3 / stride 1 / padding 1 only")

    def forward_up2_cl(self, x_cl, act=None):
        k4 = ops.Conv3ToK4.apply(self.weight)
        return ops.ConvUp.apply(x_cl, k4, self.bias, 2, act, False, False, None)

    def forward(self, x):
        raise CvaeError("UpConv2dK3 is the conv of an Upsample(x2, nearest) + Conv2d(k3) pair: call the enclosing decoder stack "
                        "(a bare 3x3 convolution has no gfx950 kernel here)")


def _image_cl(x, compute_dtype):
    """The network input NC(D)HW as the first conv's channels-last operand.  A single-channel image IS channels-last already; when the
    kernels can read it in its own dtype (fp32 batch, bf16 model) it is handed over as a view — no cast launch, no bf16 copy — and the
    second value is the dtype the first conv must produce; otherwise (None) the usual converting copy is made."""
    if (x.shape[1] == 1 and not x.requires_grad and x.dtype == torch.float32 and compute_dtype != torch.float32 and x.is_contiguous()
            and ops.image_direct_ok(x, compute_dtype)):
        sp = tuple(x.shape[2:])
        return x.view(x.shape[0], *((1,) + sp if len(sp) == 2 else sp), 1), compute_dtype
    return ops.ToChannelsLast.apply(x, compute_dtype), None


_ACT_NAME = {nn.ReLU: "relu", nn.Sigmoid: "sigmoid"}


def _act_of(mod):
    if isinstance(mod, nn.LeakyReLU):
        if abs(mod.negative_slope - 0.2) > 1e-12:
            raise CvaeError("only LeakyReLU(0.2) is implemented")
        return "leaky02"
    return _ACT_NAME.get(type(mod))


class ConvStack(nn.Sequential):
    """Encoder: [Conv, ReLU]* (+ AdaptiveAvgPool) + Flatten.  forward(x NC(D)HW fp32) -> [B, F] fp32."""
    compute_dtype = torch.float32

    def conv_weights(self):
        return [m.weight for m in self if isinstance(m, _ConvBase)]

    def features_cl(self, x, packed=None, f8=None):
        """Run the conv chain; returns the last activation channels-last (compute dtype) and the layers left over.
        `packed`: this stack's entries of an ops.pack_weights call made by the caller (one launch for the whole model).
        f8: a causal_vae_amd.fp8.Fp8Forward whose "enc" layers run their forward product on fp8 operands."""
        mods = list(self)
        convs = [m for m in mods if isinstance(m, _ConvBase)]
        if not convs or x.shape[1] != convs[0].in_channels:
            raise RuntimeError(f"expected input with {convs[0].in_channels if convs else '?'} channels, got {tuple(x.shape)}")
        h, first_dtype = _image_cl(x, self.compute_dtype)
        packed = iter(packed if packed is not None else ops.pack_weights([m.weight for m in convs], convs[0]._nd, self.compute_dtype))
        i, prev_act, j, h8 = 0, None, 0, None
        while i < len(mods) and isinstance(mods[i], _ConvBase):
            act = _act_of(mods[i + 1]) if i + 1 < len(mods) else None
            rec = f8.layer("enc", j) if f8 is not None else None
            # every consumer of a ReLU output inside this stack (next conv, final pool) folds that ReLU's mask
            if rec is not None:
                h, h8 = f8.run(rec, mods[i], h, h8, act, prev_act == "relu", act == "relu", next(packed))
            else:
                # a bf16 layer in front of an fp8 one leaves the codes of its result itself when it can (the image layer's kernel): no quantise pass
                side = f8.side_for("enc", j + 1, mods[i], h, first_dtype if i == 0 else None) if f8 is not None else None
                h = mods[i].forward_cl(h, act=act, in_is_relu_out=(prev_act == "relu"), grad_premasked=(act == "relu"), packed=next(packed),
                                       out_dtype=first_dtype if i == 0 else None, f8=side)
                h8 = side.get("y8") if side is not None else None
            prev_act = act
            i += 2 if act else 1
            j += 1
        return h, mods[i:], prev_act

    def _pooled(self, x):
        require_gpu(x)
        h, rest, last_act = self.features_cl(x)
        nd = x.dim() - 2
        out_size = tuple(h.shape[1:4])                       # bare Flatten == pooling with 1-voxel windows
        for m in rest:
            if isinstance(m, (nn.AdaptiveAvgPool2d, nn.AdaptiveAvgPool3d)):
                o = m.output_size if isinstance(m.output_size, tuple) else (m.output_size,) * nd
                out_size = ((1,) + tuple(o)) if nd == 2 else tuple(o)
            elif not isinstance(m, nn.Flatten):
                raise CvaeError(f"ConvStack: unsupported trailing layer {type(m).__name__}")
        return h, out_size, last_act == "relu"

    def forward(self, x):
        return ops.AdaptiveAvgPoolFlatten.apply(*self._pooled(x))

    def forward_cat(self, x, extras):
        """torch.cat([self(x), *extras], dim=1) without the intermediate: the features are written into the wide matrix directly (ops.FlattenCat)."""
        return ops.FlattenCat.apply(*self._pooled(x), *extras)


class DeconvStack(nn.Sequential):
    """Decoder: [ConvTranspose, ReLU]* + ConvTranspose (+ Sigmoid).  forward(h NC(D)HW fp32) -> NC(D)HW fp32."""
    compute_dtype = torch.float32

    def forward_cl(self, h):
        return self.forward_from_cl(ops.ToChannelsLast.apply(h, self.compute_dtype))

    def conv_weights(self):
        return [m.weight for m in self if isinstance(m, _ConvBase)]

    def forward_from_cl(self, x, packed=None, f8=None):
        """The deconv chain on an input that is already channels-last in the compute dtype.  f8: a causal_vae_amd.fp8.Fp8Forward whose "dec"
        layers run their forward product on fp8 operands."""
        mods = list(self)
        convs = [m for m in mods if isinstance(m, _ConvBase)]
        packed = iter(packed if packed is not None else ops.pack_weights([m.weight for m in convs], convs[0]._nd, self.compute_dtype))
        i, prev_act, j, x8 = 0, None, 0, None
        while i < len(mods):
            if not isinstance(mods[i], _ConvBase):
                raise CvaeError(f"DeconvStack: unexpected layer {type(mods[i]).__name__} at index {i}")
            act = _act_of(mods[i + 1]) if i + 1 < len(mods) else None
            nxt = i + (2 if act else 1)
            rec = f8.layer("dec", j) if f8 is not None else None
            if rec is not None:
                x, x8 = f8.run(rec, mods[i], x, x8, act, prev_act == "relu", act == "relu" and nxt < len(mods), next(packed))
            else:
                x, x8 = mods[i].forward_cl(x, act=act, in_is_relu_out=(prev_act == "relu"), grad_premasked=(act == "relu" and nxt < len(mods)), packed=next(packed)), None
            prev_act = act
            i = nxt
            j += 1
        return x

    def forward(self, h):
        require_gpu(h)
        return ops.FromChannelsLast.apply(self.forward_cl(h), h.dim() - 2)

    # ---- fp8 (e4m3) inference: the batched counterfactual decode ----
    def calibrate_fp8(self, x_cl, headroom=1.0, c1_fp8_input=True):
        """Per-tensor scales and fp8 weight panels for the ConvTranspose layers with more than one output channel, from ONE bf16 pass over the
        calibration rows x_cl [B, .., C] (channels-last, any float dtype): scale = headroom * amax / 448 for every fp8 layer's input and weight.
        c1_fp8_input: a single-channel 3D output layer behind an fp8 layer takes that layer's fp8 codes as its input (False: bf16 between the two, round 2's plan).
        Returns the plan forward_fp8 takes.  Host syncs here (amax readbacks): call it once, outside any timed or captured region."""
        mods = list(self)
        convs = [(i, m) for i, m in enumerate(mods) if isinstance(m, _ConvBase)]
        nd = convs[0][1]._nd
        plan, x = [], x_cl.to(torch.bfloat16)
        with torch.no_grad():
            for j, (i, conv) in enumerate(convs):
                act = _act_of(mods[i + 1]) if i + 1 < len(mods) else None
                if conv.weight.shape[1] > 1 and conv.weight.shape[0] % 32 == 0 and conv.weight.shape[1] % 32 == 0:
                    sx = max(float(x.float().abs().max()), 1e-12) * headroom / ops.FP8_MAX
                    sw = max(float(conv.weight.abs().max()), 1e-12) / ops.FP8_MAX
                    plan.append(dict(index=i, act=act, sx=sx, sw=sw, wq=ops.pack_weight_fp8(conv.weight.detach(), nd, True, sw)))
                elif (nd == 3 and tuple(conv.weight.shape[:2]) == (32, 1) and plan and plan[-1]["sx"] is not None and c1_fp8_input
                      and isinstance(conv, ConvTranspose3d)):
                    # the single-channel output layer behind an fp8 layer reads that layer's fp8 codes (cvae_conv_up_c1_fp8in): the 32-channel tensor between the last
                    # two layers — the largest of the chain — is written and read at one byte per element
                    plan.append(dict(index=i, act=act, sx=None, sx8=max(float(x.float().abs().max()), 1e-12) * headroom / ops.FP8_MAX))
                else:
                    plan.append(dict(index=i, act=act, sx=None))
                x = ops.ConvUp.apply(x, conv.weight.detach(), conv.bias.detach() if conv.bias is not None else None, nd, act, False, False, None)
        return plan

    def forward_fp8(self, x_cl, plan):
        """The deconv chain on fp8 operands where the plan has scales (bf16 for the remaining layers, i.e. the single-channel output layer):
        activations travel as fp8 codes between consecutive fp8 layers.  Inference only (no autograd)."""
        mods = list(self)
        nd = mods[plan[0]["index"]]._nd
        with torch.no_grad():
            x, x_is_q = x_cl, False
            for j, e in enumerate(plan):
                conv = mods[e["index"]]
                bias = conv.bias.detach() if conv.bias is not None else None
                if e["sx"] is None and e.get("sx8") is not None and x_is_q:
                    x, x_is_q = ops.conv_up_c1_fp8in(x, conv.weight.detach(), bias, e["sx8"], nd, e["act"]), False
                    continue
                if e["sx"] is None:
                    if x_is_q:
                        raise CvaeError("forward_fp8: an fp8 layer cannot feed a non-fp8 layer without its output scale")
                    x = ops.ConvUp.apply(x.to(torch.bfloat16), conv.weight.detach(), bias, nd, e["act"], False, False, None)
                    continue
                if not x_is_q:
                    x = ops.quantize_fp8(x, e["sx"])
                nxt = plan[j + 1] if j + 1 < len(plan) else None
                out_scale = (nxt["sx"] if nxt["sx"] is not None else nxt.get("sx8")) if nxt is not None else None
                x = ops.conv_up_fp8(x, e["wq"], bias, conv.weight.shape[1], nd, e["act"], e["sx"] * e["sw"], out_scale)
                x_is_q = out_scale is not None
        return x


MLP_CHAIN = __import__("os").environ.get("CVAE_MLP_CHAIN", "1") != "0"      # (env switch: A/B runs) activation gradients handed to the next layer's GEMM


class MLP(nn.Sequential):
    """Sequential of Linear / BatchNorm1d / activation layers; Linear + activation pairs run as one kernel."""

    def forward(self, x):
        require_gpu(x)
        mods = list(self)
        i = 0
        # batches above 16 (GEMM path): a Linear + activation whose output feeds the next Linear directly hands the activation gradient to that layer's data-gradient
        # GEMM (ops.Linear in_act / grad_premasked) — inside this Sequential the activation output has no other consumer
        chain = MLP_CHAIN and x.dim() == 2 and x.shape[0] > 16 and torch.is_grad_enabled()
        prev_act = None                                      # activation whose output is the current x (and whose Linear was told its gradient arrives premasked)
        while i < len(mods):
            m = mods[i]
            if isinstance(m, Linear):
                act = _act_of(mods[i + 1]) if i + 1 < len(mods) else None
                nxt = i + (2 if act else 1)
                hand = chain and act in ("relu", "sigmoid", "leaky02") and nxt < len(mods) and isinstance(mods[nxt], Linear)
                x = m(x, act=act, in_act=prev_act, grad_premasked=hand)
                prev_act = act if hand else None
                i += 2 if act else 1
            elif isinstance(m, BatchNorm1d):
                x = m(x)
                i += 1
            elif _act_of(m):
                x = ops.Activation.apply(x, _act_of(m))
                i += 1
            else:
                raise CvaeError(f"MLP: unsupported layer {type(m).__name__}")
        return x


class BNConvStack(nn.Sequential):
    """Encoder of CausalVesselVAE: [Conv2d(k4,s2,p1), BatchNorm2d, LeakyReLU(0.2)]* + Flatten.  forward(x NCHW fp32) -> [B, F] fp32."""
    compute_dtype = torch.float32

    def forward(self, x):
        require_gpu(x)
        mods = list(self)
        h, first_dtype = _image_cl(x, self.compute_dtype)
        i = 0
        while i < len(mods) and isinstance(mods[i], _ConvBase):
            conv, bn = mods[i], mods[i + 1]
            if not isinstance(bn, BatchNorm2d) or _act_of(mods[i + 2]) is None:
                raise CvaeError("BNConvStack: Conv2d, BatchNorm2d, activation triples expected")
            h = bn.forward_cl(conv.forward_cl(h, act=None, out_dtype=first_dtype if i == 0 else None), act=_act_of(mods[i + 2]))
            i += 3
        if i != len(mods) - 1 or not isinstance(mods[i], nn.Flatten):
            raise CvaeError("BNConvStack: trailing Flatten expected")
        return ops.FromChannelsLast.apply(h, 2).flatten(1)


class UpConvStack(nn.Sequential):
    """Decoder of CausalVesselVAE: [Upsample(x2, nearest), Conv2d(k3,s1,p1), BatchNorm2d, ReLU]* + Upsample, Conv2d(k3), Sigmoid.
    forward(h NCHW fp32) -> NCHW fp32; every Upsample + Conv2d pair runs as one transposed-conv product."""
    compute_dtype = torch.float32

    def forward_cl(self, h):
        mods = list(self)
        x = ops.ToChannelsLast.apply(h, self.compute_dtype)
        i = 0
        while i < len(mods):
            up, conv = mods[i], mods[i + 1]
            if not isinstance(up, nn.Upsample) or up.mode != "nearest" or float(up.scale_factor) != 2.0 or not isinstance(conv, UpConv2dK3):
                raise CvaeError("UpConvStack: Upsample(scale_factor=2, mode='nearest') + Conv2d(k3) pairs expected")
            if i + 2 < len(mods) and isinstance(mods[i + 2], BatchNorm2d):
                x = mods[i + 2].forward_cl(conv.forward_up2_cl(x, act=None), act=_act_of(mods[i + 3]))
                i += 4
            else:
                act = _act_of(mods[i + 2]) if i + 2 < len(mods) else None
                x = conv.forward_up2_cl(x, act=act)
                i += 3 if act else 2
        return x

    def forward(self, h):
        require_gpu(h)
        return ops.FromChannelsLast.apply(self.forward_cl(h), 2)


def set_linear_math(module, dtype):
    """torch.bfloat16: every Linear of `module` multiplies with bf16 MFMA operands (fp32 accumulate, fp32 tensors) where the batch is above the
    skinny range and the product is large enough to be matrix-bound (ops.LINEAR_BF16_MIN_WORK) — the MNIST heads at batch 1024.  torch.float32
    (default): exact-fp32 MFMA everywhere, which the 3D ELBO target (1e-4) needs.  Independent of set_compute_dtype (the convs)."""
    if dtype not in (torch.float32, torch.bfloat16, None):
        raise CvaeError(f"linear math must be float32 or bfloat16, got {dtype}")
    for m in module.modules():
        if isinstance(m, Linear):
            m.math = torch.bfloat16 if dtype == torch.bfloat16 else None
    return module


def set_compute_dtype(module, dtype):
    """bf16 or fp32 conv arithmetic (weights stay fp32 masters; linears, losses and BN always fp32)."""
    if dtype not in (torch.float32, torch.bfloat16):
        raise CvaeError(f"compute dtype must be float32 or bfloat16, got {dtype}")
    for m in module.modules():
        if isinstance(m, (_ConvBase, ConvStack, DeconvStack, BNConvStack, UpConvStack, UpConv2dK3)):
            m.compute_dtype = dtype
    return module
