"""torch.autograd.Function wrappers over the C ABI (include/cvae_hip.h).

The reference reaches its arithmetic through autograd (`loss.backward()`, causal_cascade/train.py:33), so the kernels hook
in here: every Function.forward / backward enqueues HIP kernels on torch's current stream and nothing else.  Conv
activations are channels-last [B, D, H, W, C] (D = 1 for 2D) in the compute dtype (float32 or bfloat16); linear layers,
losses, BN and sampling are fp32.
"""
import ctypes as C_

import torch

from . import _lib as L
from ._lib import lib, check, ptr, stream


_SIDE = {}
FORK_BACKWARD = False      # True: weight-gradient kernels run on a side stream next to the data-gradient kernels
FORK_MAX_POSITIONS = 0     # > 0: only layers with at most this many S positions per batch fork (the small, latency-bound layers)
DEFER_JOIN = False         # True: the side stream is not joined after each layer but once, by join_side_streams(), before the gradients are
                           # consumed (exchange / optimizer): the weight gradients then trail the data-gradient chain instead of gating it
_PENDING = {}              # device key -> tensors produced on the side stream since the last join (kept alive until then)


class _Fork:
    """`with _Fork() as f:` runs its body on a side HIP stream that first waits for the current stream; f.join() makes
    the current stream wait for the side work.  The weight gradient of a layer and the data gradient of the same layer
    only share inputs, so the two kernels (often too small to fill 256 CUs each) overlap.  Captured graphs record the
    fork/join as dependencies."""

    def __init__(self, device, positions=None):
        self.main = torch.cuda.current_stream(device)
        key = (device.index if device.index is not None else torch.cuda.current_device())
        self.key = key
        if key not in _SIDE:
            _SIDE[key] = torch.cuda.Stream(device=device)
        on = FORK_BACKWARD and (FORK_MAX_POSITIONS <= 0 or positions is None or positions <= FORK_MAX_POSITIONS)
        self.side = _SIDE[key] if on else None
        self.ctx = None

    def __enter__(self):
        if self.side is not None:
            self.side.wait_stream(self.main)
            self.ctx = torch.cuda.stream(self.side)
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
        return False

    def join(self, *tensors):
        if self.side is None:
            return
        if DEFER_JOIN:
            _PENDING.setdefault(self.key, []).extend(t for t in tensors if t is not None)
            return
        self.main.wait_stream(self.side)
        for t in tensors:
            if t is not None:
                t.record_stream(self.main)


def join_side_streams():
    """Make the current stream wait for every weight gradient still running on a side stream (DEFER_JOIN).  Called before anything reads
    `.grad`: the gradient exchange, clip_grad_norm_, optimizer.step()."""
    for key, tensors in list(_PENDING.items()):
        if not tensors:
            continue
        main = torch.cuda.current_stream(tensors[0].device)
        main.wait_stream(_SIDE[key])
        for t in tensors:
            t.record_stream(main)
        tensors.clear()


def _cl_dims(t):
    assert t.dim() == 5 and t.is_contiguous(), "channels-last [B, D, H, W, C] contiguous tensor expected"
    return t.shape


def _empty(shape, dtype, like):
    return torch.empty(shape, dtype=dtype, device=like.device)


_ZERO_POOL = [None, 0]


class zero_pool:
    """`with ops.zero_pool(n):` — the accumulating scalar reductions inside the block (loss sums: `*out += ...` kernels) take their zeroed 0-dim outputs
    from ONE torch.zeros(n) made on entry instead of one fill launch each.  The fill belongs to the step that uses it, so a captured step stays
    replayable; more than n requests fall back to their own zeros."""

    def __init__(self, n, like):
        self.n, self.like = int(n), like

    def __enter__(self):
        self.prev = list(_ZERO_POOL)
        _ZERO_POOL[0], _ZERO_POOL[1] = torch.zeros(self.n, dtype=torch.float32, device=self.like.device), 0
        return self

    def __exit__(self, *exc):
        _ZERO_POOL[0], _ZERO_POOL[1] = self.prev
        return False


def _scalar(like):
    pool, used = _ZERO_POOL
    if pool is not None and used < pool.numel() and pool.device == like.device:
        _ZERO_POOL[1] = used + 1
        return pool[used]
    return torch.zeros((), dtype=torch.float32, device=like.device)


def _scratch(nbytes, like):
    """(pointer, bytes) of `nbytes` of device scratch for one call (None, 0 when nbytes == 0).  Reductions leave their per-workgroup partial
    sums here and add them in a fixed order (no float atomics).  A fresh block per call: safe on any stream, and inside a capture it
    comes from the graph's pool."""
    if not nbytes:
        return None, None, 0
    t = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=like.device)
    return t, t.data_ptr(), nbytes


_RED_WS = None


def _red_ws(like):
    """Scratch of cvae_reduce_workspace_bytes() for the scalar loss reductions."""
    global _RED_WS
    if _RED_WS is None:
        _RED_WS = int(lib.cvae_reduce_workspace_bytes())
    return _scratch(_RED_WS, like)


_ONES = {}


def cached_one(loss):
    """A ones scalar like `loss`, kept per (device, dtype): seeds a backward without the fill kernel autograd launches for the implicit 1."""
    key = (loss.device, loss.dtype)
    one = _ONES.get(key)
    if one is None:
        one = torch.ones((), dtype=loss.dtype, device=loss.device)
        if not (loss.is_cuda and torch.cuda.is_current_stream_capturing()):      # a tensor born inside a capture lives in that graph's pool
            _ONES[key] = one
    return one


def backward_from(loss):
    """loss.backward() seeded with cached_one(loss)."""
    loss.backward(cached_one(loss))


# ------------------------------------------------------------------------------------------------ layout plumbing
class ToChannelsLast(torch.autograd.Function):
    """NC(D)HW fp32 -> channels-last [B, D, H, W, C] in `dtype` (cvae_ncs_to_nsc)."""

    @staticmethod
    def forward(ctx, x, dtype):
        L.require_gpu(x)
        x = x.contiguous()
        B, C = x.shape[0], x.shape[1]
        sp = tuple(x.shape[2:])
        S = 1
        for s in sp:
            S *= s
        d5 = (1,) + sp if len(sp) == 2 else sp
        out = _empty((B,) + d5 + (C,), dtype, x)
        check(lib.cvae_ncs_to_nsc(ptr(x), ptr(out), B, C, S, L.dtype_code(x.dtype), L.dtype_code(dtype), stream()), "ncs_to_nsc")
        ctx.meta = (x.shape, x.dtype, S)
        return out

    @staticmethod
    def backward(ctx, g):
        shape, dt, S = ctx.meta
        g = g.contiguous()
        out = _empty(shape, dt, g)
        check(lib.cvae_nsc_to_ncs(ptr(g), ptr(out), shape[0], shape[1], S, L.dtype_code(g.dtype), L.dtype_code(dt), stream()), "nsc_to_ncs")
        return out, None


class FromChannelsLast(torch.autograd.Function):
    """channels-last [B, D, H, W, C] -> NC(D)HW fp32; nd selects whether D is dropped."""

    @staticmethod
    def forward(ctx, x, nd):
        L.require_gpu(x)
        B, D, H, W, Cc = _cl_dims(x)
        sp = (H, W) if nd == 2 else (D, H, W)
        out = _empty((B, Cc) + sp, torch.float32, x)
        check(lib.cvae_nsc_to_ncs(ptr(x), ptr(out), B, Cc, D * H * W, L.dtype_code(x.dtype), L.F32, stream()), "nsc_to_ncs")
        ctx.meta = (x.shape, x.dtype)
        return out

    @staticmethod
    def backward(ctx, g):
        shape, dt = ctx.meta
        g = g.contiguous()
        out = _empty(shape, dt, g)
        B, D, H, W, Cc = shape
        check(lib.cvae_ncs_to_nsc(ptr(g), ptr(out), B, Cc, D * H * W, L.dtype_code(g.dtype), L.dtype_code(dt), stream()), "ncs_to_nsc")
        return out, None


def _copy_panels(panels, wide, col0, gather):
    """cvae_copy_panels on fp32 [B, w_i] matrices (row strides taken from the tensors) and the wide matrix they sit side by side in."""
    import ctypes as C
    k = len(panels)
    check(lib.cvae_copy_panels((C.c_void_p * k)(*[t.data_ptr() for t in panels]), (C.c_int64 * k)(*[t.shape[1] for t in panels]), (C.c_int64 * k)(*[t.stride(0) for t in panels]),
                               k, ptr(wide), wide.shape[0], wide.stride(0), col0, int(gather), stream()), "copy_panels")


def _as_panel(t):
    """A [B, w] fp32 matrix whose rows are contiguous (a column slice of a wider matrix qualifies: its row stride travels with it)."""
    if t.dtype != torch.float32 or t.dim() != 2:
        raise L.CvaeError("cat: fp32 [B, n] matrices expected")
    return t if (t.stride(1) == 1 and t.stride(0) >= t.shape[1]) or t.shape[1] == 0 else t.contiguous()


class Cat(torch.autograd.Function):
    """torch.cat(tensors, dim=1) for fp32 [B, n_i] matrices: one launch (cvae_copy_panels), one more for all the gradients."""

    @staticmethod
    def forward(ctx, *ts):
        L.require_gpu(*ts)
        if len(ts) > 8:
            raise L.CvaeError("cat: at most 8 tensors")
        ts = [_as_panel(t) for t in ts]
        B = ts[0].shape[0]
        widths = [t.shape[1] for t in ts]
        out = _empty((B, sum(widths)), torch.float32, ts[0])
        _copy_panels(ts, out, 0, False)
        ctx.widths = widths
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        B = g.shape[0]
        need = [i for i in range(len(ctx.widths)) if ctx.needs_input_grad[i]]
        outs = [None] * len(ctx.widths)
        if need:
            # the needed pieces are gathered in one launch: consecutive runs of needed columns would be simplest, but any subset works by
            # giving the skipped pieces a width-0 panel... they still shift the offsets, so gather run by run
            col, runs, cur = 0, [], None
            for i, w in enumerate(ctx.widths):
                if ctx.needs_input_grad[i]:
                    outs[i] = _empty((B, w), torch.float32, g)
                    if cur is None:
                        cur = (col, [])
                        runs.append(cur)
                    cur[1].append(outs[i])
                else:
                    cur = None
                col += w
            for col0, panels in runs:
                _copy_panels(panels, g, col0, True)
        return tuple(outs)


def cat(tensors):
    return Cat.apply(*tensors)


def one_hot(t, n_classes):
    """F.one_hot(t, n).float() (causal_cascade/models.py:71)."""
    L.require_gpu(t)
    t = t.contiguous()
    if t.dtype != torch.int64:
        raise L.CvaeError("class index tensor must be int64")
    out = _empty((t.shape[0], n_classes), torch.float32, t)
    check(lib.cvae_onehot_panel(ptr(t), ptr(out), t.shape[0], n_classes, n_classes, 0, stream()), "onehot_panel")
    return out


def cast(x, dtype):
    if x.dtype == dtype:
        return x
    x = x.contiguous()
    out = torch.empty_like(x, dtype=dtype)
    check(lib.cvae_cast(ptr(x), ptr(out), x.numel(), L.dtype_code(x.dtype), L.dtype_code(dtype), stream()), "cast")
    return out


class Cast(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dtype):
        ctx.src = x.dtype
        return cast(x, dtype)

    @staticmethod
    def backward(ctx, g):
        return cast(g.contiguous(), ctx.src), None


# ------------------------------------------------------------------------------------------------ conv family
def _taps(nd):
    return 64 if nd == 3 else 16


def pack_weight(w, nd, for_up, dtype):
    """fp32 [Cs][Cl][k..] -> MFMA operand panels (cvae_conv_pack_weight); the fp32 weight itself when Cl == 1."""
    Cs, Cl = w.shape[0], w.shape[1]
    w = w.contiguous()
    if Cl == 1:
        return w
    out = torch.empty(Cs * Cl * _taps(nd), dtype=dtype, device=w.device)
    check(lib.cvae_conv_pack_weight(ptr(w), ptr(out), Cs, Cl, nd, int(for_up), L.dtype_code(dtype), stream()), "conv_pack_weight")
    return out


def pack_weights(weights, nd, dtype, f8spec=None):
    """Pack the weights of several conv layers for BOTH directions in one launch (cvae_conv_pack_weight_pairs).
    Returns [(packed_down, packed_up)] per weight; a Cl == 1 layer gets its fp32 weight back for both (not packed).
    f8spec (fp8 training forward, causal_vae_amd.fp8): {id(weight): (f8dir, f8out, inv_scale_dev, amax)} — that weight's forward-direction panel
    (f8dir 1 = down, 2 = up) is written as fp8 codes into f8out instead of bf16 (its entry in the returned pair is None), max |w| recorded."""
    import ctypes as C
    outs, ws, pd, pu, cs, cl, fd, fo, fi, fa = [], [], [], [], [], [], [], [], [], []
    for w0 in weights:
        w = w0.contiguous()
        Cs, Cl = w.shape[0], w.shape[1]
        if Cl == 1:
            outs.append((w, w))
            continue
        n = Cs * Cl * _taps(nd)
        spec = f8spec.get(id(w0)) if f8spec else None
        if spec is not None:
            one = torch.empty(n, dtype=dtype, device=w.device)
            d, u = (None, one) if spec[0] == 1 else (one, None)
        else:
            both = torch.empty(2 * n, dtype=dtype, device=w.device)
            d, u = both[:n], both[n:]
        outs.append((d, u))
        ws.append(w.data_ptr()); pd.append(d.data_ptr() if d is not None else None); pu.append(u.data_ptr() if u is not None else None); cs.append(Cs); cl.append(Cl)
        fd.append(spec[0] if spec else 0); fo.append(spec[1].data_ptr() if spec else None); fi.append(spec[2].data_ptr() if spec else None)
        fa.append(spec[3].data_ptr() if (spec and spec[3] is not None) else None)
    k = len(ws)
    if k:
        vp = lambda v: (C.c_void_p * k)(*v)
        if any(fd):
            check(lib.cvae_conv_pack_weight_pairs_f8(vp(ws), vp(pd), vp(pu), (C.c_int64 * k)(*cs), (C.c_int64 * k)(*cl), (C.c_int * k)(*fd), vp(fo), vp(fi), vp(fa),
                                                     k, nd, L.dtype_code(dtype), stream()), "conv_pack_weight_pairs_f8")
        else:
            check(lib.cvae_conv_pack_weight_pairs(vp(ws), vp(pd), vp(pu), (C.c_int64 * k)(*cs), (C.c_int64 * k)(*cl), k, nd, L.dtype_code(dtype), stream()), "conv_pack_weight_pairs")
    return outs


SPLIT_K = True      # hand the conv data kernels their split-K scratch (tests switch it off to cover the unsplit path)


def _can_defer(weight, bias):
    """A weight gradient may be computed at the END of the backward pass only if nothing looks at it earlier: no accumulation onto an
    existing .grad (AccumulateGrad would add the still-empty tensor), no tensor hooks (pre- or post-accumulate), no double backward, and
    no second use of the same weight in this pass (the engine would add two still-empty tensors before the flush)."""
    if torch.is_grad_enabled():
        return False
    for p in (weight, bias):
        if p is None:
            continue
        if not p.is_leaf or p.grad is not None or p._backward_hooks or getattr(p, "_post_accumulate_grad_hooks", None):
            return False                                     # a derived weight (ops.Conv3ToK4) hands its gradient on at once
    if _wg_task_is_current():
        if id(weight) in _WG_SHARED:
            return False
        if any(e["wid"] == id(weight) for e in _WG_PENDING):
            _wgrad_flush()                                   # the first use's gradient must exist before the engine sums the two
            _WG_SHARED.add(id(weight))
            return False
    return True


def _conv_data_workspace(device, B, sd, sh, sw, Cs, ld, lh, lw, Cl, nd, for_up):
    """Split-K scratch for the small-grid layers (cvae_conv_data_workspace_bytes; 0 bytes for the large ones)."""
    nbytes = lib.cvae_conv_data_workspace_bytes(B, sd, sh, sw, Cs, ld, lh, lw, Cl, nd, for_up) if SPLIT_K else 0
    if not nbytes:
        return None, 0
    return torch.empty(nbytes // 4, dtype=torch.float32, device=device), nbytes


def image_direct_ok(x, dtype):
    """True when the first conv can read the 1-channel image `x` in ITS dtype while computing in `dtype` (cvae_conv_down_image)."""
    return x.dtype != dtype and x.is_contiguous() and bool(lib.cvae_conv_image_supported(ptr(x), x.shape[-2] if x.dim() == 5 and x.shape[-1] == 1 else x.shape[-1],
                                                                                        L.dtype_code(x.dtype), L.dtype_code(dtype)))


MASK_BITS = __import__("os").environ.get("CVAE_MASK_BITS", "1") != "0"     # (env switch: A/B runs) ReLU masks travel as bits (1/16 of the saved activation's bytes) wherever producer and consumer are this library's conv launches
BITS_STATS = {"produced": 0, "consumed": 0}      # launches that left / read a mask in bit form (tests check that the model really takes the path)


def _bits_for(t):
    """int32 tensor for the ReLU mask bits of channels-last tensor `t` (one dword per 32 consecutive elements), or None when the channel count does not allow it."""
    return torch.empty(t.numel() // 32, dtype=torch.int32, device=t.device) if (t.shape[-1] % 32 == 0 and t.numel() % 32 == 0) else None


def _conv_down(Lt, wp, bias, mask, Cs, nd, act, out_dtype=None, mask_bits=None, want_bits=None):
    """mask_bits: the ReLU mask as bits (replaces `mask`).  want_bits (True / False; None = plain return): return (S, bits) — the ReLU mask bits of the result
    when asked for and supported, else None."""
    B, ld, lh, lw, Cl = _cl_dims(Lt)
    sd, sh, sw = (ld // 2 if nd == 3 else 1), lh // 2, lw // 2
    if out_dtype is not None and out_dtype != Lt.dtype:      # single-channel image read in its own dtype, S in the compute dtype
        if Cl != 1:
            raise L.CvaeError("a mixed-dtype conv is available for the single-channel image layer only")
        S = _empty((B, sd, sh, sw, Cs), out_dtype, Lt)
        bits = _bits_for(S) if (want_bits and out_dtype == torch.bfloat16 and mask is None) else None
        if bits is not None:
            check(L.timed(f"conv_down nd{nd} B{B} L{ld}x{lh}x{lw}x{Cl} -> S{Cs}", lib.cvae_conv_down_image_f8, ptr(Lt), L.dtype_code(Lt.dtype), ptr(wp), ptr(bias), ptr(S), None, None,
                          None, ptr(bits), B, sd, sh, sw, Cs, ld, lh, lw, nd, L.act_code(act), stream()), "conv_down_image_f8")
        else:
            check(L.timed(f"conv_down nd{nd} B{B} L{ld}x{lh}x{lw}x{Cl} -> S{Cs}", lib.cvae_conv_down_image, ptr(Lt), L.dtype_code(Lt.dtype), ptr(wp), ptr(bias), ptr(mask), ptr(S),
                          B, sd, sh, sw, Cs, ld, lh, lw, nd, L.dtype_code(out_dtype), L.act_code(act), stream()), "conv_down_image")
        return (S, bits) if want_bits is not None else S
    S = _empty((B, sd, sh, sw, Cs), Lt.dtype, Lt)
    ws, nbytes = _conv_data_workspace(Lt.device, B, sd, sh, sw, Cs, ld, lh, lw, Cl, nd, 0)
    label = f"conv_down nd{nd} B{B} L{ld}x{lh}x{lw}x{Cl} -> S{Cs}"
    # the bit forms: MFMA layers in any dtype; the single-channel end (dec4's data gradient applies d3's mask) in its bf16 16-byte-row form only
    c1_ok = Cl != 1 or (Lt.dtype == torch.bfloat16 and lw % 8 == 0)
    bits = _bits_for(S) if (want_bits and Cl != 1 and Cs % 64 == 0 and Cl % 16 == 0) else None
    use_mb = mask_bits if (mask_bits is not None and c1_ok) else None      # the bit form of the mask wins over the tensor form when the launch can read it
    if use_mb is None and mask is not None:
        bits = None                                          # a tensor mask: the plain entry point (backward launches want no bits of their own anyway)
    if DOWN_VARIANT is not None and Cl != 1 and use_mb is None:
        check(lib.cvae_conv_down_variant(ptr(Lt), ptr(wp), ptr(bias), ptr(mask), ptr(S), B, sd, sh, sw, Cs, ld, lh, lw, Cl, nd, L.dtype_code(Lt.dtype), L.act_code(act),
                                         ptr(ws), nbytes, int(DOWN_VARIANT), stream()), "conv_down_variant")
        return (S, None) if want_bits is not None else S
    if use_mb is not None or bits is not None:
        check(L.timed(label, lib.cvae_conv_down_bits, ptr(Lt), ptr(wp), ptr(bias), ptr(use_mb), ptr(S), ptr(bits), B, sd, sh, sw, Cs, ld, lh, lw, Cl, nd, L.dtype_code(Lt.dtype),
                      L.act_code(act), ptr(ws), nbytes, stream()), "conv_down_bits")
        BITS_STATS["produced"] += bits is not None
        BITS_STATS["consumed"] += use_mb is not None
        return (S, bits) if want_bits is not None else S
    check(L.timed(label, lib.cvae_conv_down, ptr(Lt), ptr(wp), ptr(bias), ptr(mask), ptr(S),
                  B, sd, sh, sw, Cs, ld, lh, lw, Cl, nd, L.dtype_code(Lt.dtype), L.act_code(act), ptr(ws), nbytes, stream()), "conv_down")
    return (S, None) if want_bits is not None else S


_GRAD_AT_APPLY = [True]


def _backward_may_follow(ctx):
    """Inside Function.forward: was autograd recording when apply() was called, and does any input ask for a gradient?  (ctx.needs_input_grad alone ignores
    torch.no_grad(), and grad mode is always off inside forward.)  Decides whether a ReLU layer leaves its mask bits: an inference pass writes none."""
    return _GRAD_AT_APPLY[0] and any(ctx.needs_input_grad)


class _GradModeAtApply:
    """Mixin for the conv Functions: records torch.is_grad_enabled() at the call."""

    @classmethod
    def apply(cls, *args, **kwargs):
        _GRAD_AT_APPLY[0] = torch.is_grad_enabled()
        try:
            return super().apply(*args, **kwargs)
        finally:
            _GRAD_AT_APPLY[0] = True


DOWN_VARIANT = None  # test hook: xpair (0 / 1) for cvae_conv_down_variant — two samples per tile off / on for every multi-channel `down` launch; None = automatic
UP_VARIANT = None    # test hook: (upfull, xpair, c1_walk_units) for cvae_conv_up_variant / the xpair of cvae_conv_fp8; None = the library's automatic choice


def _conv_up(St, wp, bias, mask, Cl, nd, act, l_dims=None, mask_bits=None, want_bits=None):
    """l_dims: spatial extent (ld, lh, lw) of the result when it is not 2 s — the data gradient of a conv over an odd extent (l = 2 s + 1).
    mask_bits / want_bits: as in _conv_down."""
    B, sd, sh, sw, Cs = _cl_dims(St)
    ld, lh, lw = ((2 * sd if nd == 3 else 1), 2 * sh, 2 * sw) if l_dims is None else tuple(int(v) for v in l_dims)
    Lt = _empty((B, ld, lh, lw, Cl), St.dtype, St)
    ws, nbytes = _conv_data_workspace(St.device, B, sd, sh, sw, Cs, ld, lh, lw, Cl, nd, 1)
    if UP_VARIANT is not None:
        check(lib.cvae_conv_up_variant(ptr(St), ptr(wp), ptr(bias), ptr(mask), ptr(Lt), B, sd, sh, sw, Cs, ld, lh, lw, Cl, nd, L.dtype_code(St.dtype), L.act_code(act),
                                       ptr(ws), nbytes, int(UP_VARIANT[0]), int(UP_VARIANT[1]), int(UP_VARIANT[2]), stream()), "conv_up_variant")
        return (Lt, None) if want_bits is not None else Lt
    label = f"conv_up nd{nd} B{B} S{sd}x{sh}x{sw}x{Cs} -> L{Cl}"
    bits_ok = Cl != 1 and Cl % 32 == 0 and Cs % 16 == 0
    use_mb = mask_bits if (mask_bits is not None and bits_ok) else None
    bits = _bits_for(Lt) if (want_bits and bits_ok and (mask is None or use_mb is not None)) else None
    if use_mb is not None or bits is not None:
        check(L.timed(label, lib.cvae_conv_up_bits, ptr(St), ptr(wp), ptr(bias), ptr(use_mb), ptr(Lt), ptr(bits), B, sd, sh, sw, Cs, ld, lh, lw, Cl, nd, L.dtype_code(St.dtype),
                      L.act_code(act), ptr(ws), nbytes, stream()), "conv_up_bits")
        BITS_STATS["produced"] += bits is not None
        BITS_STATS["consumed"] += use_mb is not None
        return (Lt, bits) if want_bits is not None else Lt
    check(L.timed(label, lib.cvae_conv_up, ptr(St), ptr(wp), ptr(bias), ptr(mask), ptr(Lt),
                  B, sd, sh, sw, Cs, ld, lh, lw, Cl, nd, L.dtype_code(St.dtype), L.act_code(act), ptr(ws), nbytes, stream()), "conv_up")
    return (Lt, None) if want_bits is not None else Lt


DEFER_WGRAD = True     # weight gradients of the MFMA conv layers are queued during a backward pass and computed by ONE grouped launch (+ one
                       # reduce launch) at its end (cvae_conv_wgrad_multi): the small layers run in the shadow of the large ones
_WG_PENDING = []
_WG_QUEUED = [False]
_WG_TASK = [-1]            # id of the autograd graph task the queue (and its end-of-backward callback) belongs to
_WG_SHARED = set()         # weights met twice in this pass (shared by two layers): their gradients are computed at once


def _graph_task_id():
    f = getattr(torch._C, "_current_graph_task_id", None)
    return f() if f is not None else -1


def _wg_task_is_current():
    return _WG_QUEUED[0] and _WG_TASK[0] == _graph_task_id()


def reset_pending_wgrads():
    """Forget queued weight gradients without computing them.  A backward pass that raised never runs its end-of-backward callback (the
    engine drops it), so its entries — and the "callback queued" flag — would otherwise outlive it; every deferral checks for that
    itself (a queue that belongs to another graph task is stale), and callers that catch a failed step / capture may call this too."""
    _WG_PENDING.clear()
    _WG_SHARED.clear()
    _WG_QUEUED[0] = False
    _WG_TASK[0] = -1


def _wgrad_flush():
    """Run every queued weight gradient (grouped by device / nd / dtype, <= 8 layers per launch).  Installed as the autograd engine's
    end-of-backward callback, so `.grad` is complete when loss.backward() / torch.autograd.grad() returns — any optimizer works.
    Every entry owns a reference to the STORAGE of its outputs (not to the tensors: a second tensor reference would make AccumulateGrad
    clone the still-empty gradient instead of adopting it), so the launch never writes to memory the allocator has handed on — also when
    nobody kept the gradient (torch.autograd.grad(loss, [x]) with trainable weights)."""
    if not _WG_PENDING:
        return
    todo = list(_WG_PENDING)
    _WG_PENDING.clear()
    groups = {}
    for e in todo:
        groups.setdefault((e["S"].device, e["nd"], e["S"].dtype), []).append(e)
    for (dev, nd, dt), es in groups.items():
        with torch.cuda.device(dev):
            cur = torch.cuda.current_stream(dev)
            for c0 in range(0, len(es), 8):
                ch = es[c0:c0 + 8]
                k = len(ch)
                vp = lambda key: (C_.c_void_p * k)(*[((e[key] if isinstance(e[key], int) else e[key].data_ptr()) if e[key] is not None else None) for e in ch])
                dims = (C_.c_int64 * (9 * k))(*[v for e in ch for v in e["dims"]])
                label = f"conv_wgrad_multi nd{nd} B{ch[0]['dims'][0]} " + ";".join("S{1}x{2}x{3}x{4}L{8}".format(*e["dims"]) for e in ch)
                check(L.timed(label, lib.cvae_conv_wgrad_multi, k, vp("S"), vp("L"), vp("dW"), vp("db"), (C_.c_int * k)(*[e["side"] for e in ch]), vp("ws"),
                              (C_.c_size_t * k)(*[e["nbytes"] for e in ch]), dims, nd, L.dtype_code(dt), stream()), "conv_wgrad_multi")
                for e in ch:                                 # scratch allocated on a forked side stream (FORK_BACKWARD), used here on the caller's:
                    if e["alloc_stream"] != cur.cuda_stream:  # tell the caching allocator, or the side stream's pool may hand it on too early
                        e["ws"].record_stream(cur)


def _wgrad_callback():
    _WG_QUEUED[0] = False
    _WG_TASK[0] = -1
    _WG_SHARED.clear()
    _wgrad_flush()


def flush_pending_wgrads():
    """Compute queued weight gradients now (FusedAdam.overlap_backward reads gradients before the backward pass has ended)."""
    _wgrad_flush()


def _conv_wgrad(St, Lt, nd, wshape, want_sbias=False, want_lbias=False, may_defer=False, wid=None):
    """dW and, in the same pass, the bias gradient: want_sbias = per-channel sum of S (Conv layer), want_lbias = of L (ConvTranspose).
    may_defer: the caller has checked that nothing reads the returned tensors before the backward pass ends (see _can_defer)."""
    B, sd, sh, sw, Cs = _cl_dims(St)
    _, ld, lh, lw, Cl = _cl_dims(Lt)
    dW = torch.empty(wshape, dtype=torch.float32, device=St.device)
    db = torch.empty(Cl if want_lbias else Cs, dtype=torch.float32, device=St.device) if (want_sbias or want_lbias) else None
    nbytes = lib.cvae_conv_wgrad_workspace_bytes(Cs, Cl, nd)
    ws = torch.empty(max(nbytes, 4) // 4, dtype=torch.float32, device=St.device)
    if Lt.dtype != St.dtype:                                 # the image layer of a bf16 model: L is the fp32 input itself
        if Cl != 1 or want_lbias:
            raise L.CvaeError("a mixed-dtype weight gradient is available for the single-channel image layer only")
        check(L.timed(f"conv_wgrad nd{nd} B{B} S{sd}x{sh}x{sw}x{Cs} L{Cl}", lib.cvae_conv_wgrad_image, ptr(St), ptr(Lt), L.dtype_code(Lt.dtype), ptr(dW), ptr(db), ptr(ws), nbytes,
                      B, sd, sh, sw, Cs, ld, lh, lw, nd, L.dtype_code(St.dtype), stream()), "conv_wgrad_image")
        return (dW, db) if want_sbias else dW
    exact2x = lh == 2 * sh and lw == 2 * sw and (nd == 2 or ld == 2 * sd)
    if (DEFER_WGRAD and may_defer and Cl != 1 and Cs % 64 == 0 and Cl % 32 == 0 and B > 0 and (exact2x or not want_lbias)):
        try:
            tid = _graph_task_id()
            if _WG_QUEUED[0] and _WG_TASK[0] != tid:         # left behind by a backward pass that raised: its callback never ran
                reset_pending_wgrads()
            if not _WG_QUEUED[0]:
                torch.autograd.Variable._execution_engine.queue_callback(_wgrad_callback)     # only legal inside a backward pass
                _WG_QUEUED[0] = True
                _WG_TASK[0] = tid
            # the outputs are queued by ADDRESS plus a reference to their STORAGE: a second reference to the tensors dW / db would make
            # AccumulateGrad clone them (still empty) instead of adopting them as .grad
            _WG_PENDING.append(dict(S=St, L=Lt, dW=dW.data_ptr(), db=(db.data_ptr() if db is not None else None), side=1 if want_lbias else 0, ws=ws,
                                    keep=(dW.untyped_storage(), db.untyped_storage() if db is not None else None), wid=wid,
                                    alloc_stream=torch.cuda.current_stream(St.device).cuda_stream,
                                    nbytes=nbytes, nd=nd, dims=(B, sd, sh, sw, Cs, ld, lh, lw, Cl)))
            return (dW, db) if (want_sbias or want_lbias) else dW
        except RuntimeError:
            pass                                             # not inside a backward pass (a direct call): compute now
    check(L.timed(f"conv_wgrad nd{nd} B{B} S{sd}x{sh}x{sw}x{Cs} L{Cl}", lib.cvae_conv_wgrad, ptr(St), ptr(Lt), ptr(dW), ptr(db), 1 if want_lbias else 0, ptr(ws), nbytes,
                  B, sd, sh, sw, Cs, ld, lh, lw, Cl, nd, L.dtype_code(St.dtype), stream()), "conv_wgrad")
    return (dW, db) if (want_sbias or want_lbias) else dW


def _channel_sum(x):
    Cc = x.shape[-1]
    out = torch.empty(Cc, dtype=torch.float32, device=x.device)
    P = x.numel() // Cc
    _t, wp, wb = _scratch(lib.cvae_channel_sum_workspace_bytes(P, Cc, L.dtype_code(x.dtype)), x)
    check(lib.cvae_channel_sum(ptr(x), ptr(out), P, Cc, L.dtype_code(x.dtype), wp, wb, stream()), "channel_sum")
    return out


def _act_bwd(g, y, act):
    out = torch.empty_like(g)
    check(lib.cvae_act_bwd(ptr(g), ptr(y), ptr(out), g.numel(), L.act_code(act), L.dtype_code(g.dtype), stream()), "act_bwd")
    return out


class ConvDown(_GradModeAtApply, torch.autograd.Function):
    """nn.Conv{2,3}d(k=4, s=2, p=1) + bias + activation on channels-last tensors.

    in_is_relu_out: the input is itself a ReLU output, so the ReLU mask of the *producer* is fused into this op's
      backward-data epilogue (the gradient it returns is already masked).
    grad_premasked: the consumer of this op's output fuses this op's ReLU mask the same way, so the incoming
      gradient needs no separate activation-backward pass.  (ReLU masking is idempotent; the flags only save traffic.)
    """

    @staticmethod
    def forward(ctx, x, weight, bias, nd, act, in_is_relu_out, grad_premasked, packed=None, out_dtype=None, f8=None):
        """out_dtype (single-channel image layer only): compute / output dtype when it differs from x's — the fp32 input batch of a bf16 model
        is read as it is, without a cast pass.
        f8 (causal_vae_amd.fp8): the forward product on fp8 operands — dict(xq = codes of x, wq = fp8 weight panels, dscale, want_out8, amax); the
        codes of the result are left in f8["y8"].  x (bf16) is still what the backward pass reads."""
        L.require_gpu(x, weight, bias)
        Cs = weight.shape[0]
        if out_dtype is not None and out_dtype == x.dtype:
            out_dtype = None
        if f8 is not None and f8.get("side"):
            # the single-channel image layer of an fp8 forward: bf16 arithmetic as always, and the result a second time as fp8 codes (+ its amax)
            # for the fp8 conv that follows (cvae_conv_down_image_f8)
            B, ld, lh, lw, Cl = _cl_dims(x)
            if Cl != 1 or (out_dtype or x.dtype) != torch.bfloat16:
                raise L.CvaeError("the fp8 side output exists for the single-channel image layer of a bf16 model only")
            sd, sh, sw = (ld // 2 if nd == 3 else 1), lh // 2, lw // 2
            y = _empty((B, sd, sh, sw, Cs), torch.bfloat16, x)
            y8 = torch.empty((B, sd, sh, sw, Cs), dtype=torch.uint8, device=x.device)
            f8["bits"] = _bits_for(y) if (MASK_BITS and act == "relu" and _backward_may_follow(ctx)) else None
            check(L.timed(f"conv_down nd{nd} B{B} L{ld}x{lh}x{lw}x{Cl} -> S{Cs}", lib.cvae_conv_down_image_f8, ptr(x), L.dtype_code(x.dtype), ptr(weight.contiguous()), ptr(bias),
                          ptr(y), ptr(y8), ptr(f8["inv_scale"]), ptr(f8.get("amax")), ptr(f8["bits"]), B, sd, sh, sw, Cs, ld, lh, lw, nd, L.act_code(act), stream()), "conv_down_image_f8")
            f8["y8"] = y8
            bits = f8.get("bits")
        elif f8 is not None:
            want_bits = MASK_BITS and act == "relu" and _backward_may_follow(ctx)
            res = conv_fp8(False, f8["xq"], f8["wq"], bias, Cs, nd, act, dscale=f8["dscale"], want_out8=f8.get("want_out8", False), amax=f8.get("amax"), want_bits=want_bits)
            y, f8["y8"], bits = res
        else:
            wp = packed[0] if packed is not None else pack_weight(weight, nd, False, out_dtype or x.dtype)
            y, bits = _conv_down(x, wp, bias, None, Cs, nd, act, out_dtype, want_bits=MASK_BITS and act == "relu" and _backward_may_follow(ctx))     # no backward to come (inference): no mask
        if bits is not None:
            y._relu_bits = bits                              # travels with the activation to the layer whose backward applies this ReLU
        ctx.x_bits = getattr(x, "_relu_bits", None) if (MASK_BITS and in_is_relu_out) else None
        ctx.save_for_backward(x, weight, y)
        ctx.bias_ref = bias
        ctx.packed_bwd = packed[1] if packed is not None else None
        ctx.cfg = (nd, act, in_is_relu_out, grad_premasked, bias is not None)
        return y

    @staticmethod
    def backward(ctx, g):
        x, weight, y = ctx.saved_tensors
        nd, act, in_relu, premasked, has_bias = ctx.cfg
        g = g.contiguous()
        if act not in (None, "none") and not (premasked and act == "relu"):
            g = _act_bwd(g, y, act)
        dx = dw = db = None
        want_db = has_bias and ctx.needs_input_grad[2]
        fork = _Fork(g.device, g.shape[0] * g.shape[1] * g.shape[2] * g.shape[3])
        with fork:
            if ctx.needs_input_grad[1]:
                defer = _can_defer(weight, ctx.bias_ref if want_db else None)
                if want_db:
                    dw, db = _conv_wgrad(g, x, nd, weight.shape, want_sbias=True, may_defer=defer, wid=id(weight))
                else:
                    dw = _conv_wgrad(g, x, nd, weight.shape, may_defer=defer, wid=id(weight))
            elif want_db:
                db = _channel_sum(g)
        if ctx.needs_input_grad[0]:
            wp_up = ctx.packed_bwd if ctx.packed_bwd is not None else pack_weight(weight, nd, True, g.dtype)
            if x.dtype != g.dtype:
                raise L.CvaeError("the image read in its own dtype carries no gradient (cast it first if it needs one)")
            dx = _conv_up(g, wp_up, None, x if in_relu else None, weight.shape[1], nd, None, l_dims=x.shape[1:4], mask_bits=ctx.x_bits if in_relu else None)
        fork.join(dw, db)
        return dx, dw, db, None, None, None, None, None, None, None


class ConvUp(_GradModeAtApply, torch.autograd.Function):
    """nn.ConvTranspose{2,3}d(k=4, s=2, p=1) + bias + activation on channels-last tensors (flags as ConvDown)."""

    @staticmethod
    def forward(ctx, x, weight, bias, nd, act, in_is_relu_out, grad_premasked, packed=None, f8=None):
        L.require_gpu(x, weight, bias)
        Cl = weight.shape[1]
        if f8 is not None:                                   # forward product on fp8 operands (see ConvDown.forward)
            res = conv_fp8(True, f8["xq"], f8["wq"], bias, Cl, nd, act, dscale=f8["dscale"], want_out8=f8.get("want_out8", False), amax=f8.get("amax"),
                           want_bits=MASK_BITS and act == "relu" and _backward_may_follow(ctx))
            y, f8["y8"], bits = res
        else:
            wp = packed[1] if packed is not None else pack_weight(weight, nd, True, x.dtype)
            y, bits = _conv_up(x, wp, bias, None, Cl, nd, act, want_bits=MASK_BITS and act == "relu" and _backward_may_follow(ctx))
        if bits is not None:
            y._relu_bits = bits
        ctx.x_bits = getattr(x, "_relu_bits", None) if (MASK_BITS and in_is_relu_out) else None
        ctx.save_for_backward(x, weight, y)
        ctx.bias_ref = bias
        ctx.packed_bwd = packed[0] if packed is not None else None
        ctx.cfg = (nd, act, in_is_relu_out, grad_premasked, bias is not None)
        return y

    @staticmethod
    def backward(ctx, g):
        x, weight, y = ctx.saved_tensors
        nd, act, in_relu, premasked, has_bias = ctx.cfg
        g = g.contiguous()
        if act not in (None, "none") and not (premasked and act == "relu"):
            g = _act_bwd(g, y, act)
        dx = dw = db = None
        fork = _Fork(g.device, x.shape[0] * x.shape[1] * x.shape[2] * x.shape[3])
        with fork:
            want_db = has_bias and ctx.needs_input_grad[2]
            if ctx.needs_input_grad[1]:
                defer = _can_defer(weight, ctx.bias_ref if want_db else None)
                if want_db:
                    dw, db = _conv_wgrad(x, g, nd, weight.shape, want_lbias=True, may_defer=defer, wid=id(weight))
                else:
                    dw = _conv_wgrad(x, g, nd, weight.shape, may_defer=defer, wid=id(weight))
            elif want_db:
                db = _channel_sum(g)
        if ctx.needs_input_grad[0]:
            wp_dn = ctx.packed_bwd if ctx.packed_bwd is not None else pack_weight(weight, nd, False, g.dtype)
            dx = _conv_down(g, wp_dn, None, x if in_relu else None, weight.shape[0], nd, None, mask_bits=ctx.x_bits if in_relu else None)
        fork.join(dw, db)
        return dx, dw, db, None, None, None, None, None, None


# ------------------------------------------------------------------------------------------------ fp8 inference (decode only)
FP8_MAX = 448.0      # OCP e4m3


def quantize_fp8(x, scale):
    """fp8 (e4m3) codes of x / scale as a uint8 tensor of x's shape (x: fp32 or bf16, contiguous)."""
    L.require_gpu(x)
    x = x.contiguous()
    q = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    check(lib.cvae_quantize_fp8(ptr(x), L.dtype_code(x.dtype), ptr(q), x.numel(), 1.0 / float(scale), stream()), "quantize_fp8")
    return q


def pack_weight_fp8(w, nd, for_up, scale):
    """fp32 [Cs][Cl][k..] -> the MFMA operand panels of pack_weight as fp8 codes of w / scale."""
    w = w.contiguous()
    out = torch.empty(w.numel(), dtype=torch.uint8, device=w.device)
    check(lib.cvae_conv_pack_weight_fp8(ptr(w), ptr(out), w.shape[0], w.shape[1], nd, int(for_up), 1.0 / float(scale), stream()), "conv_pack_weight_fp8")
    return out


def conv_up_fp8(Sq, wq, bias, Cl, nd, act, acc_scale, out_scale=None):
    """nn.ConvTranspose{2,3}d(k4, s2, p1) + bias + activation on fp8 operands (forward only).  Sq: uint8 codes [B, sd, sh, sw, Cs]; result bf16
    (out_scale None) or fp8 codes of result / out_scale."""
    return conv_fp8(True, Sq, wq, bias, Cl, nd, act, acc_scale=acc_scale, out8_scale=out_scale, codes_only=out_scale is not None)


def conv_up_c1_fp8in(Sq, weight, bias, in_scale, nd, act):
    """The single-channel nn.ConvTranspose3d(32, 1, 4, 2, 1) end of a decoder fed by fp8 codes (forward only; cvae_conv_up_c1_fp8in): Sq uint8 [B, sd, sh, sw, 32] =
    codes of activation / in_scale, weight / bias the layer's fp32 parameters.  Returns bf16 [B, 2sd, 2sh, 2sw, 1]."""
    L.require_gpu(Sq, weight)
    if Sq.dtype != torch.uint8 or nd != 3 or Sq.dim() != 5 or Sq.shape[-1] != 32 or tuple(weight.shape[:2]) != (32, 1):
        raise L.CvaeError(f"conv_up_c1_fp8in: 3D fp8 codes [B, d, h, w, 32] and a ConvTranspose3d(32, 1) weight expected, got {tuple(Sq.shape)} {Sq.dtype} / {tuple(weight.shape)}")
    Sq, weight = Sq.contiguous(), weight.contiguous().float()
    B, sd, sh, sw, Cs = Sq.shape
    out = torch.empty(B, 2 * sd, 2 * sh, 2 * sw, 1, dtype=torch.bfloat16, device=Sq.device)
    check(L.timed(f"up_c1_fp8in nd{nd} B{B} S{sd}x{sh}x{sw}x{Cs} L1", lib.cvae_conv_up_c1_fp8in, ptr(Sq), ptr(weight), ptr(bias.contiguous().float() if bias is not None else None),
                  ptr(out), float(in_scale), B, sd, sh, sw, Cs, nd, L.act_code(act), stream()), "conv_up_c1_fp8in")
    return out


AMAX_SLOTS = 4096        # CVAE_AMAX_SLOTS


def conv_fp8(up, xq, wq, bias, Cout, nd, act, acc_scale=None, out8_scale=None, codes_only=False, dscale=None, want_out8=False, amax=None, want_bits=None):
    """One fp8 (e4m3) product on the block-scaled MFMA (cvae_conv_fp8; forward only).  up False: nn.Conv (k4, s2, p1) of xq [B, ld, lh, lw, Cin];
    up True: nn.ConvTranspose of xq [B, sd, sh, sw, Cin].  xq, wq: uint8 codes (quantize_fp8 / pack_weight_fp8 panels).
    Scales by value (acc_scale = s_x * s_w; out8_scale = the scale of the fp8 copy of the result) or on the device (dscale: float32 tensor
    {acc_scale, 1 / out8_scale}, re-read by every launch — the training step's delayed scaling).
    Returns the bf16 result; with codes_only the fp8 codes instead; with want_out8 (or out8_scale and not codes_only) the pair (bf16, codes); with
    want_bits given (True / False) always the triple (bf16, codes or None, ReLU mask bits of the result or None).
    amax: optional uint32 tensor of AMAX_SLOTS words that records max |result|."""
    L.require_gpu(xq)
    if xq.dtype != torch.uint8 or wq.dtype != torch.uint8:
        raise L.CvaeError("conv_fp8: activations and weight panels must be fp8 codes (uint8 tensors from quantize_fp8 / pack_weight_fp8)")
    xq = xq.contiguous()
    B, d, h, w_, Cin = _cl_dims(xq)
    if wq.numel() != Cin * Cout * 4 ** nd:
        raise L.CvaeError(f"conv_fp8: weight panels hold {wq.numel()} codes, expected Cin * Cout * 4^nd = {Cin * Cout * 4 ** nd}")
    if up:
        sd, sh, sw, Cs, Cl = d, h, w_, Cin, Cout
        ld, lh, lw = (2 * sd if nd == 3 else 1), 2 * sh, 2 * sw
        oshape = (B, ld, lh, lw, Cl)
    else:
        ld, lh, lw, Cl, Cs = d, h, w_, Cin, Cout
        sd, sh, sw = (ld // 2 if nd == 3 else 1), lh // 2, lw // 2
        oshape = (B, sd, sh, sw, Cs)
    if dscale is None and acc_scale is None:
        raise L.CvaeError("conv_fp8: give acc_scale (by value) or dscale (device pair)")
    pair = (want_out8 or out8_scale is not None) and not codes_only
    out = torch.empty(oshape, dtype=torch.uint8 if codes_only else torch.bfloat16, device=xq.device)
    out8 = torch.empty(oshape, dtype=torch.uint8, device=xq.device) if pair else None
    ws, nbytes = (None, 0) if codes_only else _conv_data_workspace(xq.device, B, sd, sh, sw, Cs, ld, lh, lw, Cl, nd, int(bool(up)))
    label = (f"conv_up fp8 nd{nd} B{B} S{sd}x{sh}x{sw}x{Cs} -> L{Cl}" if up else f"conv_down fp8 nd{nd} B{B} L{ld}x{lh}x{lw}x{Cl} -> S{Cs}")
    bits = _bits_for(out) if (want_bits and not codes_only and Cout % 32 == 0) else None
    check(L.timed(label, lib.cvae_conv_fp8, int(bool(up)), ptr(xq), ptr(wq), ptr(bias), ptr(out), L.FP8 if codes_only else L.BF16, ptr(out8), ptr(dscale),
                  float(acc_scale if acc_scale is not None else 1.0), (1.0 / float(out8_scale)) if out8_scale is not None else 1.0, ptr(amax),
                  B, sd, sh, sw, Cs, ld, lh, lw, Cl, nd, L.act_code(act), ptr(ws), nbytes, -1 if UP_VARIANT is None else int(UP_VARIANT[1]), ptr(bits), stream()), "conv_fp8")
    if want_bits is not None:                                # the training forward's call: always the triple (result, codes or None, mask bits or None)
        return out, out8, bits
    return (out, out8) if pair else out


def absmax(x, slots):
    """Record max |x| in `slots` (AMAX_SLOTS uint32 / int32 words of float bits; atomicMax — accumulates over calls until cleared)."""
    L.require_gpu(x)
    x = x.contiguous()
    check(lib.cvae_absmax(ptr(x), L.dtype_code(x.dtype), x.numel(), ptr(slots), stream()), "absmax")


def quantize_fp8_dev(x, inv_scale_dev, amax=None):
    """quantize_fp8 with 1 / scale read from a device float (one-element tensor); optionally records max |x|."""
    L.require_gpu(x)
    x = x.contiguous()
    q = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    check(lib.cvae_quantize_fp8_dev(ptr(x), L.dtype_code(x.dtype), ptr(q), x.numel(), ptr(inv_scale_dev), ptr(amax), stream()), "quantize_fp8_dev")
    return q


def pack_weights_fp8(weights, nd, for_up, inv_scale_devs, amaxes=None, outs=None):
    """fp8 operand panels of several conv weights in one launch (cvae_conv_pack_weights_fp8): 1 / s_w of each from device memory, max |w| recorded.
    outs: optional preallocated uint8 tensors (a captured step reuses them)."""
    k = len(weights)
    ws = [w.contiguous() for w in weights]
    if outs is None:
        outs = [torch.empty(w.numel(), dtype=torch.uint8, device=w.device) for w in ws]
    if k:
        vp = lambda ts: (C_.c_void_p * k)(*[(t.data_ptr() if t is not None else None) for t in ts])
        check(lib.cvae_conv_pack_weights_fp8(vp(ws), vp(outs), (C_.c_int64 * k)(*[w.shape[0] for w in ws]), (C_.c_int64 * k)(*[w.shape[1] for w in ws]),
                                             (C_.c_int * k)(*[int(bool(f)) for f in for_up]), vp(inv_scale_devs), vp(amaxes if amaxes is not None else [None] * k),
                                             k, nd, stream()), "conv_pack_weights_fp8")
    return outs


class Fp8Scales:
    """Device-resident delayed scaling for the fp8 forward of a training step: `n` tracked tensors (activations and weights), each with an amax
    record, a scale and its reciprocal; `layers` = [(input tensor, weight tensor, output tensor or -1)] per fp8 product, whose {s_in * s_w, 1 / s_out}
    pairs land in `dscale[l]`.  update() (one tiny launch, capturable) turns the amaxes recorded since the last call into the scales of the next
    step: scale = headroom * amax / 448 — e4m3 is a floating-point format, so headroom costs no precision until the subnormals."""

    def __init__(self, n, layers, device, headroom=2.0):
        self.n, self.layers, self.headroom = int(n), [tuple(int(v) for v in l) for l in layers], float(headroom)
        self.amax = torch.zeros(self.n, AMAX_SLOTS, dtype=torch.int32, device=device)
        self.scale = torch.ones(self.n, dtype=torch.float32, device=device)
        self.inv_scale = torch.ones(self.n, dtype=torch.float32, device=device)
        self.dscale = torch.zeros(max(len(self.layers), 1), 2, dtype=torch.float32, device=device)
        self.ticket = torch.zeros(1, dtype=torch.int32, device=device)
        k = len(self.layers)
        self._li = [(C_.c_int * max(k, 1))(*([l[j] for l in self.layers] or [0])) for j in range(3)]

    def update(self):
        check(lib.cvae_fp8_scale_update(ptr(self.amax), ptr(self.scale), ptr(self.inv_scale), self.n, self.headroom, self._li[0], self._li[1], self._li[2],
                                        len(self.layers), ptr(self.dscale), ptr(self.ticket), stream()), "fp8_scale_update")

    def state(self):
        return {"scale": self.scale.detach().cpu().clone(), "amax": self.amax.detach().cpu().clone()}

    def load_state(self, st):
        self.scale.copy_(st["scale"]); self.inv_scale.copy_(1.0 / st["scale"].to(self.scale.device)); self.amax.copy_(st["amax"])


class Activation(torch.autograd.Function):
    """Stand-alone ReLU / Sigmoid / LeakyReLU(0.2) (used where no producer kernel can fuse it, e.g. after BatchNorm1d)."""

    @staticmethod
    def forward(ctx, x, act):
        L.require_gpu(x)
        x = x.contiguous()
        y = torch.empty_like(x)
        check(lib.cvae_act_fwd(ptr(x), ptr(y), x.numel(), L.act_code(act), L.dtype_code(x.dtype), stream()), "act_fwd")
        ctx.save_for_backward(y)
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        return _act_bwd(g.contiguous(), y, ctx.act), None


# ------------------------------------------------------------------------------------------------ pool / resize
class AdaptiveAvgPoolFlatten(torch.autograd.Function):
    """nn.AdaptiveAvgPool{2,3}d(out) + nn.Flatten on a channels-last input -> fp32 [B, C*prod(out)] in NC(D)HW flatten order.
    relu_input: the input is a ReLU output whose mask is fused into the backward."""

    @staticmethod
    def forward(ctx, x, out_size, relu_input):
        L.require_gpu(x)
        B, D, H, W, Cc = _cl_dims(x)
        OD, OH, OW = out_size
        F = Cc * OD * OH * OW
        out = _empty((B, F), torch.float32, x)
        check(lib.cvae_adaptive_avgpool_fwd(ptr(x), ptr(out), B, D, H, W, Cc, OD, OH, OW, F, L.dtype_code(x.dtype), stream()), "avgpool_fwd")
        ctx.save_for_backward(x)
        ctx.cfg = (out_size, relu_input)
        return out

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        (OD, OH, OW), relu_input = ctx.cfg
        B, D, H, W, Cc = x.shape
        g = g.contiguous()
        dx = torch.empty_like(x)
        check(lib.cvae_adaptive_avgpool_bwd(ptr(g), ptr(x) if relu_input else None, ptr(dx), B, D, H, W, Cc, OD, OH, OW, g.shape[1],
                                            L.dtype_code(x.dtype), stream()), "avgpool_bwd")
        return dx, None, None


class FlattenCat(torch.autograd.Function):
    """cat([Flatten(AdaptiveAvgPool(x)), *extras], dim=1) written in place: the pooled / flattened features go straight into the wide matrix (row stride
    = its width) and the extras follow in one panel launch; the backward reads the feature columns of the incoming gradient where they lie."""

    @staticmethod
    def forward(ctx, x, out_size, relu_input, *extras):
        L.require_gpu(x, *extras)
        B, D, H, W, Cc = _cl_dims(x)
        OD, OH, OW = out_size
        F = Cc * OD * OH * OW
        extras = [_as_panel(e) for e in extras]
        if len(extras) > 8 or any(e.shape[0] != B for e in extras):
            raise L.CvaeError("FlattenCat: up to 8 extras with the batch of x")
        K = F + sum(e.shape[1] for e in extras)
        out = _empty((B, K), torch.float32, x)
        check(lib.cvae_adaptive_avgpool_fwd(ptr(x), ptr(out), B, D, H, W, Cc, OD, OH, OW, K, L.dtype_code(x.dtype), stream()), "avgpool_fwd")
        if extras:
            _copy_panels(extras, out, F, False)
        ctx.save_for_backward(x)
        ctx.cfg = (out_size, relu_input, F, [e.shape[1] for e in extras])
        return out

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        (OD, OH, OW), relu_input, F, widths = ctx.cfg
        B, D, H, W, Cc = x.shape
        g = g.contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            check(lib.cvae_adaptive_avgpool_bwd(ptr(g), ptr(x) if relu_input else None, ptr(dx), B, D, H, W, Cc, OD, OH, OW, g.shape[1],
                                                L.dtype_code(x.dtype), stream()), "avgpool_bwd")
        outs, col = [], F
        for i, w in enumerate(widths):
            if ctx.needs_input_grad[3 + i]:
                o = _empty((B, w), torch.float32, g)
                _copy_panels([o], g, col, True)
                outs.append(o)
            else:
                outs.append(None)
            col += w
        return (dx, None, None) + tuple(outs)


class UpsampleLinear(torch.autograd.Function):
    """F.interpolate(mode='bilinear'|'trilinear', align_corners=False) on a channels-last input -> fp32 channels-last."""

    @staticmethod
    def forward(ctx, x, size):
        L.require_gpu(x)
        B, d, h, w, Cc = _cl_dims(x)
        D, H, W = size
        out = _empty((B, D, H, W, Cc), torch.float32, x)
        if Cc == 1 and lib.cvae_up2x_supported(B, d, h, w, D, H, W):          # exact 2x, one channel: the block kernel of csrc/recon_loss.hip
            check(lib.cvae_up2x_fwd(ptr(x), ptr(out), B, d, h, w, D, H, W, L.dtype_code(x.dtype), stream()), "up2x_fwd")
        else:
            check(lib.cvae_upsample_linear_fwd(ptr(x), ptr(out), B, d, h, w, D, H, W, Cc, L.dtype_code(x.dtype), stream()), "upsample_fwd")
        ctx.meta = (x.shape, x.dtype, size)
        return out

    @staticmethod
    def backward(ctx, g):
        shape, dt, (D, H, W) = ctx.meta
        B, d, h, w, Cc = shape
        g = g.contiguous()
        dx = _empty(shape, dt, g)
        check(lib.cvae_upsample_linear_bwd(ptr(g), ptr(dx), B, d, h, w, D, H, W, Cc, L.dtype_code(dt), stream()), "upsample_bwd")
        return dx, None


# ------------------------------------------------------------------------------------------------ linear / BN
LINEAR_BF16_MIN_WORK = 1 << 24      # M * K * N below which a bf16-math Linear stays on the fp32 kernels (launch-bound there, and exact)


SMALL_DENSE = __import__("os").environ.get("CVAE_SMALL_DENSE", "1") != "0"    # (env switch: A/B runs) small layers at large batch through csrc/small_dense.hip


class Linear(torch.autograd.Function):
    """nn.Linear + optional fused activation.  fp32 tensors always; math = torch.float32: exact-fp32 MFMA (default);
    math = torch.bfloat16: operands rounded to bf16 inside the GEMM kernels (fp32 accumulate) for batches > 16 and M*K*N >= LINEAR_BF16_MIN_WORK."""

    @staticmethod
    def forward(ctx, x, weight, bias, act, math=None, in_act=None, grad_premasked=False):
        """in_act: x is the OUTPUT of that activation and this layer is its only consumer — the data gradient leaves this layer's backward already multiplied by
        the activation's derivative (cvae_linear_bwd_data_inact); grad_premasked: the gradient arriving at this layer's output already carries act'(y) (the next
        layer was given in_act = act).  layers.MLP sets the pair for consecutive layers at batch sizes above 16: one elementwise launch fewer per layer."""
        L.require_gpu(x, weight, bias)
        if x.dtype != torch.float32:
            raise L.CvaeError("linear layers run in float32")
        x = x.contiguous()
        M, K = x.shape
        N = weight.shape[0]
        b16 = math == torch.bfloat16 and M > 16 and M * K * N >= LINEAR_BF16_MIN_WORK
        y = _empty((M, N), torch.float32, x)
        small = SMALL_DENSE and not b16 and bool(lib.cvae_small_dense_supported(M, K, N))     # small layer, large batch: the weight lives in LDS (csrc/small_dense.hip)
        if small:
            check(L.timed(f"linear_fwd M{M} K{K} N{N} small", lib.cvae_small_dense_fwd, ptr(x), ptr(weight.contiguous()), ptr(bias), ptr(y), M, K, N, K, N, L.act_code(act), stream()),
                  "small_dense_fwd")
            ctx.save_for_backward(x, weight, y)
            ctx.cfg = (act, bias is not None, b16)
            ctx.chain = (in_act if in_act not in (None, "none") else None, bool(grad_premasked))
            ctx.small = True
            return y
        ctx.small = False
        _t, wp, wb = _scratch(lib.cvae_linear_workspace_bytes(M, K, N, 0), x)
        if b16:
            check(L.timed(f"linear_fwd M{M} K{K} N{N} bf16", lib.cvae_linear_fwd_bf16, ptr(x), ptr(weight), ptr(bias), ptr(y), M, K, N, K, N, L.act_code(act), wp, wb, stream()),
                  "linear_fwd_bf16")
        else:
            check(L.timed(f"linear_fwd M{M} K{K} N{N}", lib.cvae_linear_fwd, ptr(x), ptr(weight), ptr(bias), ptr(y), M, K, N, K, N, L.act_code(act), wp, wb, stream()), "linear_fwd")
        ctx.save_for_backward(x, weight, y)
        if (in_act not in (None, "none") or grad_premasked) and M <= 16:
            raise L.CvaeError("Linear: in_act / grad_premasked are for batches above 16 (the skinny kernels fuse the activation gradient on the consuming side)")
        ctx.cfg = (act, bias is not None, b16)
        ctx.chain = (in_act if in_act not in (None, "none") else None, bool(grad_premasked))
        return y

    @staticmethod
    def backward(ctx, g):
        x, weight, y = ctx.saved_tensors
        act, has_bias, b16 = ctx.cfg
        in_act, premasked = ctx.chain
        g = g.contiguous()
        M, K = x.shape
        N = weight.shape[0]
        if ctx.small:
            # this layer's activation gradient (unless the next layer already applied it) and the previous layer's (in_act) both ride in these launches
            has_act = act not in (None, "none") and not premasked
            ya, ac = (ptr(y), L.act_code(act)) if has_act else (None, L.ACT_NONE)
            dx = dw = db = None
            if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
                dw = torch.empty_like(weight)
                db = _empty((N,), torch.float32, g) if (has_bias and ctx.needs_input_grad[2]) else None
                _t, wp, wb = _scratch(lib.cvae_small_dense_workspace_bytes(M, K, N), g)
                check(L.timed(f"linear_bwd_weight M{M} K{K} N{N} small", lib.cvae_small_dense_bwd_weight, ptr(g), ptr(x), ptr(dw), ptr(db), ya, ac, M, K, N, N, K, N, wp, wb,
                              stream()), "small_dense_bwd_weight")
                if not ctx.needs_input_grad[1]:
                    dw = None
            if ctx.needs_input_grad[0]:
                dx = _empty((M, K), torch.float32, g)
                check(L.timed(f"linear_bwd_data M{M} K{K} N{N} small", lib.cvae_small_dense_bwd_data, ptr(g), ptr(weight.contiguous()), ptr(dx), ya, ac,
                              ptr(x) if in_act is not None else None, L.act_code(in_act) if in_act is not None else L.ACT_NONE, M, K, N, N, K, N, K, stream()),
                      "small_dense_bwd_data")
            return dx, dw, db, None, None, None, None
        fused = act not in (None, "none") and M <= 16          # skinny kernels apply act'(y) on the fly
        if act not in (None, "none") and not fused and not premasked:
            g = _act_bwd(g, y, act)
        ya, ac = (ptr(y), L.act_code(act)) if fused else (None, L.ACT_NONE)
        dx = dw = db = None
        fork = _Fork(g.device)
        with fork:
            if ctx.needs_input_grad[1]:
                dw = torch.empty_like(weight)
                if has_bias and ctx.needs_input_grad[2]:
                    db = _empty((N,), torch.float32, g)
                _t, wp, wb = _scratch(lib.cvae_linear_workspace_bytes(M, K, N, 2), g)
                if b16:
                    check(L.timed(f"linear_bwd_weight M{M} K{K} N{N} bf16", lib.cvae_linear_bwd_weight_bf16, ptr(g), ptr(x), ptr(dw), ptr(db), M, K, N, N, K, wp, wb, stream()),
                          "linear_bwd_weight_bf16")
                else:
                    check(L.timed(f"linear_bwd_weight M{M} K{K} N{N}", lib.cvae_linear_bwd_weight, ptr(g), ptr(x), ptr(dw), ptr(db), M, K, N, N, K, ya, ac, wp, wb, stream()),
                          "linear_bwd_weight")
            elif has_bias and ctx.needs_input_grad[2]:
                db = _channel_sum(_act_bwd(g, y, act) if fused else g)
        if ctx.needs_input_grad[0]:
            dx = _empty((M, K), torch.float32, g)
            _t2, wp2, wb2 = _scratch(lib.cvae_linear_workspace_bytes(M, K, N, 1), g)
            if in_act is not None:
                check(L.timed(f"linear_bwd_data M{M} K{K} N{N}" + (" bf16" if b16 else ""), lib.cvae_linear_bwd_data_inact, ptr(g), ptr(weight), ptr(dx), M, K, N, N, K, ptr(x), K,
                              L.act_code(in_act), int(b16), wp2, wb2, stream()), "linear_bwd_data_inact")
            elif b16:
                check(L.timed(f"linear_bwd_data M{M} K{K} N{N} bf16", lib.cvae_linear_bwd_data_bf16, ptr(g), ptr(weight), ptr(dx), M, K, N, N, K, wp2, wb2, stream()), "linear_bwd_data_bf16")
            else:
                check(L.timed(f"linear_bwd_data M{M} K{K} N{N}", lib.cvae_linear_bwd_data, ptr(g), ptr(weight), ptr(dx), M, K, N, N, K, ya, ac, wp2, wb2, stream()), "linear_bwd_data")
        fork.join(dw, db)
        return dx, dw, db, None, None, None, None


class BatchNorm1dTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum, eps):
        L.require_gpu(x)
        x = x.contiguous()
        B, F = x.shape
        y = torch.empty_like(x)
        mean = _empty((F,), torch.float32, x)
        rstd = _empty((F,), torch.float32, x)
        check(lib.cvae_bn1d_train_fwd(ptr(x), ptr(weight), ptr(bias), ptr(y), ptr(mean), ptr(rstd), ptr(running_mean), ptr(running_var),
                                      B, F, momentum, eps, stream()), "bn1d_train_fwd")
        ctx.save_for_backward(x, weight, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, g):
        x, weight, mean, rstd = ctx.saved_tensors
        g = g.contiguous()
        B, F = x.shape
        dx = torch.empty_like(x)
        dw = _empty((F,), torch.float32, x)
        db = _empty((F,), torch.float32, x)
        check(lib.cvae_bn1d_train_bwd(ptr(g), ptr(x), ptr(weight), ptr(mean), ptr(rstd), ptr(dx), ptr(dw), ptr(db), B, F, stream()), "bn1d_train_bwd")
        return dx, dw, db, None, None, None, None


def _group_size(group):
    import torch.distributed as dist
    return dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1


def _all_reduce_sum(t, group):
    import torch.distributed as dist
    if _group_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


class SyncBatchNorm1dTrain(torch.autograd.Function):
    """Train-mode BatchNorm1d whose batch statistics span the data-parallel ranks of `group` (SURVEY.md §8(e) "optional tiny collectives"):
    two all-reduces of F floats forward (sum, then sum of squared deviations: the two-pass form of the single-rank kernel) and one of 2 F
    floats backward.  Every rank must hold the same per-rank batch size.  The weight / bias gradients returned are this rank's PARTIAL sums
    — the gradient all-reduce of the step adds them up, like every other parameter gradient."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum, eps, group):
        L.require_gpu(x)
        x = x.contiguous()
        B, F = x.shape
        inv_n = 1.0 / float(B * _group_size(group))
        s1 = _empty((F,), torch.float32, x)
        check(lib.cvae_bn1d_stats(ptr(x), None, 0.0, ptr(s1), B, F, stream()), "bn1d_stats")
        _all_reduce_sum(s1, group)
        s2 = _empty((F,), torch.float32, x)
        check(lib.cvae_bn1d_stats(ptr(x), ptr(s1), inv_n, ptr(s2), B, F, stream()), "bn1d_stats")
        _all_reduce_sum(s2, group)
        y = torch.empty_like(x)
        mean, rstd = _empty((F,), torch.float32, x), _empty((F,), torch.float32, x)
        check(lib.cvae_bn1d_apply_stats(ptr(x), ptr(weight), ptr(bias), ptr(s1), ptr(s2), inv_n, ptr(y), ptr(mean), ptr(rstd), ptr(running_mean),
                                        ptr(running_var), B, F, momentum, eps, stream()), "bn1d_apply_stats")
        ctx.save_for_backward(x, weight, mean, rstd)
        ctx.cfg = (inv_n, group)
        return y

    @staticmethod
    def backward(ctx, g):
        x, weight, mean, rstd = ctx.saved_tensors
        inv_n, group = ctx.cfg
        g = g.contiguous()
        B, F = x.shape
        local = _empty((2 * F,), torch.float32, x)
        check(lib.cvae_bn1d_bwd_sums(ptr(g), ptr(x), ptr(mean), ptr(rstd), ptr(local), B, F, stream()), "bn1d_bwd_sums")
        db, dw = local[:F].clone(), local[F:].clone()           # this rank's share of d beta / d gamma
        _all_reduce_sum(local, group)
        dx = torch.empty_like(x)
        check(lib.cvae_bn1d_bwd_apply(ptr(g), ptr(x), ptr(weight), ptr(mean), ptr(rstd), ptr(local), inv_n, ptr(dx), B, F, stream()), "bn1d_bwd_apply")
        return dx, dw, db, None, None, None, None, None


def bn1d_eval(x, weight, bias, running_mean, running_var, eps):
    L.require_gpu(x)
    x = x.contiguous()
    y = torch.empty_like(x)
    check(lib.cvae_bn1d_eval_fwd(ptr(x), ptr(weight), ptr(bias), ptr(running_mean), ptr(running_var), ptr(y), x.shape[0], x.shape[1], eps, stream()),
          "bn1d_eval_fwd")
    return y


# ------------------------------------------------------------------------------------------------ sampling + losses
def philox_normal(shape, seed, offset, device, call_counter=None, subsequence=0):
    """N(0,1) draws.  call_counter: optional device int32 tensor added (<< 24) to the offset and incremented afterwards —
    the device-side call count that keeps a captured HIP graph drawing fresh numbers on every replay.  subsequence: which of the
    2^64 independent streams of `seed` (Philox counter words 2-3)."""
    out = torch.empty(shape, dtype=torch.float32, device=device)
    fn = lib.cvae_philox_normal if call_counter is None else lib.cvae_philox_normal_advance
    check(fn(ptr(out), out.numel(), seed & 0xFFFFFFFFFFFFFFFF, offset & 0xFFFFFFFFFFFFFFFF, subsequence & 0xFFFFFFFFFFFFFFFF, ptr(call_counter), stream()),
          "philox_normal")
    return out


def dist_rank():
    """Rank of this process in the default process group (the torchrun RANK before the group exists, 0 for a single process)."""
    import os
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank()
    return int(os.environ.get("RANK", "0"))


class EpsSource:
    """Per-model device Philox stream for `reparameterize(mu, logvar)` (the reference draws torch.randn_like there).

    Key = torch.initial_seed() (the reference seeds everything with 42, causal_cascade/main.py:28); subsequence = (rank << 32) | instance,
    so the ranks of a data-parallel job — which all call torch.manual_seed(42) — and the models of one process draw independent noise;
    the call counter lives on the device (it advances under HIP-graph replay) and is part of `state()` / `load_state()` so that a resumed
    run continues the stream instead of replaying it (causal_vae_amd.checkpoint)."""
    _instances = 0

    def __init__(self):
        self.counter = None
        self.instance = EpsSource._instances
        EpsSource._instances += 1
        self._pending = None

    def subsequence(self, rank=None):
        return ((dist_rank() if rank is None else int(rank)) << 32) | (self.instance & 0xFFFFFFFF)

    def _ensure(self, device):
        if self.counter is None or self.counter.device != torch.device(device):
            start = 0 if self._pending is None else int(self._pending)
            self.counter = torch.full((), start, dtype=torch.int32, device=device)
            self._pending = None

    def draw(self, like):
        self._ensure(like.device)
        return philox_normal(like.shape, torch.initial_seed(), 0, like.device, self.counter, self.subsequence())

    def noise_args(self, device):
        """(seed, subsequence, device call counter) of the NEXT draw, for a launch that makes the draw itself (ops.BioBottleneck's `noise`): the numbers and
        the counter bump are draw()'s."""
        self._ensure(device)
        return (torch.initial_seed() & 0xFFFFFFFFFFFFFFFF, self.subsequence() & 0xFFFFFFFFFFFFFFFF, self.counter)

    def state(self):
        """{'calls': number of draws so far (one host sync), 'instance': which of the process's streams this model draws from}."""
        return {"calls": int(self.counter.item()) if self.counter is not None else int(self._pending or 0), "instance": int(self.instance)}

    def load_state(self, st):
        calls = int(st["calls"])
        if "instance" in st:                                 # the stream id travels with the checkpoint: construction order in the resuming process
            self.instance = int(st["instance"])              # (a second model, a helper built first) must not change the noise
        if self.counter is not None:
            self.counter.fill_(calls)
        else:
            self._pending = calls


class Reparameterize(torch.autograd.Function):
    """z = mu + eps * exp(logvar / 2) (causal_cascade/models.py:65-68) with an explicit eps."""

    @staticmethod
    def forward(ctx, mu, logvar, eps):
        L.require_gpu(mu, logvar, eps)
        mu, logvar, eps = mu.contiguous(), logvar.contiguous(), eps.contiguous()
        z = torch.empty_like(mu)
        check(lib.cvae_reparam_kld_fwd(ptr(mu), ptr(logvar), ptr(eps), ptr(z), None, mu.numel(), None, 0, stream()), "reparam_fwd")
        ctx.save_for_backward(mu, logvar, eps)
        return z

    @staticmethod
    def backward(ctx, g):
        mu, logvar, eps = ctx.saved_tensors
        g = g.contiguous()
        dmu, dlv = torch.empty_like(mu), torch.empty_like(mu)
        check(lib.cvae_reparam_kld_bwd(ptr(g), None, 1.0, ptr(mu), ptr(logvar), ptr(eps), ptr(dmu), ptr(dlv), mu.numel(), stream()), "reparam_bwd")
        return dmu, dlv, None


class LatentHead(torch.autograd.Function):
    """(z1, z2, kld) from the encoder's [B, 2 Z] head h = (mu | logvar): z_k = mu + eps_k exp(logvar / 2) (eps2 / z2 optional), kld as ops.KLD — one launch,
    and one for d h in the backward (cvae_latent_head_fwd / _bwd) instead of chunk copies, three kernels, their gradient adds and a cat."""

    @staticmethod
    def forward(ctx, h, eps1, eps2, want_kld):
        L.require_gpu(h, eps1)
        if h.dtype != torch.float32 or h.dim() != 2 or h.shape[1] % 2:
            raise L.CvaeError("LatentHead: fp32 [B, 2 Z] head expected")
        h, eps1 = h.contiguous(), eps1.contiguous()
        B, Z = h.shape[0], h.shape[1] // 2
        if tuple(eps1.shape) != (B, Z) or (eps2 is not None and tuple(eps2.shape) != (B, Z)):
            raise L.CvaeError("LatentHead: eps must be [B, Z]")
        eps2 = None if eps2 is None else eps2.contiguous()
        z1 = _empty((B, Z), torch.float32, h)
        z2 = None if eps2 is None else _empty((B, Z), torch.float32, h)
        kld = torch.empty((), dtype=torch.float32, device=h.device) if want_kld else None
        check(lib.cvae_latent_head_fwd(ptr(h), ptr(eps1), ptr(eps2), ptr(z1), ptr(z2), ptr(kld), B, Z, stream()), "latent_head_fwd")
        ctx.save_for_backward(h, eps1, eps2)
        return z1, z2, kld

    @staticmethod
    def backward(ctx, g1, g2, gk):
        h, eps1, eps2 = ctx.saved_tensors
        B, Z = h.shape[0], h.shape[1] // 2
        dh = torch.empty_like(h)
        g1 = None if g1 is None else g1.contiguous()
        g2 = None if g2 is None else g2.contiguous()
        gk = None if gk is None else gk.float().contiguous()
        check(lib.cvae_latent_head_bwd(ptr(g1), ptr(g2), ptr(gk), ptr(h), ptr(eps1), ptr(eps2), ptr(dh), B, Z, stream()), "latent_head_bwd")
        return dh, None, None, None


class KLD(torch.autograd.Function):
    """-0.5 * sum(1 + logvar - mu^2 - exp(logvar))   (causal_cascade/train.py:13)."""

    @staticmethod
    def forward(ctx, mu, logvar):
        L.require_gpu(mu, logvar)
        mu, logvar = mu.contiguous(), logvar.contiguous()
        out = _scalar(mu)
        _t, wp, wb = _red_ws(mu)
        check(lib.cvae_reparam_kld_fwd(ptr(mu), ptr(logvar), None, None, ptr(out), mu.numel(), wp, wb, stream()), "kld_fwd")
        ctx.save_for_backward(mu, logvar)
        return out

    @staticmethod
    def backward(ctx, g):
        mu, logvar = ctx.saved_tensors
        g = g.contiguous()
        dmu, dlv = torch.empty_like(mu), torch.empty_like(mu)
        check(lib.cvae_reparam_kld_bwd(None, ptr(g), 1.0, ptr(mu), ptr(logvar), None, ptr(dmu), ptr(dlv), mu.numel(), stream()), "kld_bwd")
        return dmu, dlv


class _PairLoss(torch.autograd.Function):
    """sum-reduced elementwise loss of (a, b) with gradient to a only (b is data)."""

    @staticmethod
    def forward(ctx, a, b, kind):
        L.require_gpu(a, b)
        if a.dtype != torch.float32 or b.dtype != torch.float32:
            raise L.CvaeError("loss inputs must be float32")
        a, b = a.contiguous(), b.contiguous()
        if a.shape != b.shape:
            raise RuntimeError(f"The size of tensor a {tuple(a.shape)} must match the size of tensor b {tuple(b.shape)}")
        out = _scalar(a)
        fwd = lib.cvae_sse_fwd if kind == "sse" else lib.cvae_bce_fwd
        _t, wp, wb = _red_ws(a)
        check(fwd(ptr(a), ptr(b), ptr(out), a.numel(), wp, wb, stream()), kind + "_fwd")
        ctx.save_for_backward(a, b)
        ctx.kind = kind
        return out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = g.contiguous()
        da = torch.empty_like(a)
        if ctx.kind == "sse":
            check(lib.cvae_sse_bwd(ptr(a), ptr(b), ptr(g), 1.0, ptr(da), a.numel(), stream()), "sse_bwd")
        else:
            check(lib.cvae_bce_bwd(ptr(a), ptr(b), ptr(g), ptr(da), a.numel(), stream()), "bce_bwd")
        return da, None, None


class Elbo(torch.autograd.Function):
    """loss = SSE(recon_x, x) + gamma * SSE(m_hat, m) + KLD(mu, logvar)   (causal_cascade/train.py:5-17) in one node:
    one zeroed 4-float buffer, three reductions, one combine; the backward scales each branch by the upstream gradient of
    `loss` on the device (no host sync, no scalar torch kernels).  Returns (loss, recon, m_loss, kld)."""

    @staticmethod
    def forward(ctx, recon_x, x, m_hat, m, mu, logvar, gamma):
        L.require_gpu(recon_x, x, m_hat, m, mu, logvar)
        ts = [t.contiguous() for t in (recon_x, x, m_hat, m, mu, logvar)]
        if any(t.dtype != torch.float32 for t in ts):
            raise L.CvaeError("loss inputs must be float32")
        recon_x, x, m_hat, m, mu, logvar = ts
        if recon_x.shape != x.shape or m_hat.shape != m.shape:
            raise RuntimeError(f"The size of tensor a {tuple(recon_x.shape)} must match the size of tensor b {tuple(x.shape)}")
        buf = torch.zeros(4, dtype=torch.float32, device=x.device)
        base = buf.data_ptr()
        _t, wp, wb = _red_ws(x)                               # one scratch block: the three reductions run in stream order
        check(lib.cvae_sse_fwd(ptr(recon_x), ptr(x), base + 4, recon_x.numel(), wp, wb, stream()), "sse_fwd")
        check(lib.cvae_sse_fwd(ptr(m_hat), ptr(m), base + 8, m.numel(), wp, wb, stream()), "sse_fwd")
        check(lib.cvae_reparam_kld_fwd(ptr(mu), ptr(logvar), None, None, base + 12, mu.numel(), wp, wb, stream()), "kld_fwd")
        check(lib.cvae_combine3(base, float(gamma), 1.0, stream()), "combine3")
        ctx.save_for_backward(*ts)
        ctx.gamma = float(gamma)
        ctx.set_materialize_grads(False)
        return buf[0], buf[1], buf[2], buf[3]

    @staticmethod
    def backward(ctx, g_loss, g_recon, g_m, g_kld):
        recon_x, x, m_hat, m, mu, logvar = ctx.saved_tensors
        gam = ctx.gamma
        if g_recon is not None or g_m is not None or g_kld is not None:      # rare: someone back-propagates a single term too
            z = torch.zeros((), dtype=torch.float32, device=x.device)
            gl = g_loss if g_loss is not None else z
            s_r = gl + (g_recon if g_recon is not None else z)
            s_m = gam * gl + (g_m if g_m is not None else z)
            s_k = gl + (g_kld if g_kld is not None else z)
            scales = ((s_r.contiguous(), 1.0), (s_m.contiguous(), 1.0), (s_k.contiguous(), 1.0))
        else:
            if g_loss is None:
                return None, None, None, None, None, None, None
            gl = g_loss.contiguous()
            scales = ((gl, 1.0), (gl, gam), (gl, 1.0))
        d_recon = d_mhat = dmu = dlv = None
        if ctx.needs_input_grad[0]:
            d_recon = torch.empty_like(recon_x)
            check(lib.cvae_sse_bwd(ptr(recon_x), ptr(x), ptr(scales[0][0]), scales[0][1], ptr(d_recon), x.numel(), stream()), "sse_bwd")
        if ctx.needs_input_grad[2]:
            d_mhat = torch.empty_like(m_hat)
            check(lib.cvae_sse_bwd(ptr(m_hat), ptr(m), ptr(scales[1][0]), scales[1][1], ptr(d_mhat), m.numel(), stream()), "sse_bwd")
        if ctx.needs_input_grad[4] or ctx.needs_input_grad[5]:
            dmu, dlv = torch.empty_like(mu), torch.empty_like(mu)
            check(lib.cvae_reparam_kld_bwd(None, ptr(scales[2][0]), scales[2][1], ptr(mu), ptr(logvar), None, ptr(dmu), ptr(dlv), mu.numel(), stream()), "kld_bwd")
        return d_recon, None, d_mhat, None, dmu, dlv, None


class ElboUp2x(torch.autograd.Function):
    """The same ELBO with the exact-2x resize of the decoder output folded in: recon = SSE(F.interpolate(src, size(x)), x) without
    ever writing the resized volume (csrc/recon_loss.hip).  src: channels-last decoder output [B, d, h, w, 1] (conv dtype);
    x: [B, 1, D, H, W] / [B, 1, H, W] fp32 with D = 2d (3D), H = 2h, W = 2w.  Returns (loss, recon, m_loss, kld)."""

    @staticmethod
    def dims(src, x):
        B, d, h, w, Cc = _cl_dims(src)
        sp = tuple(x.shape[2:])
        D, H, W = sp if len(sp) == 3 else (1,) + sp
        return B, d, h, w, Cc, D, H, W

    @staticmethod
    def supported(src, x):
        B, d, h, w, Cc, D, H, W = ElboUp2x.dims(src, x)
        return (Cc == 1 and x.shape[0] == B and x.shape[1] == 1 and x.dtype == torch.float32 and bool(lib.cvae_up2x_supported(B, d, h, w, D, H, W)))

    _tickets = {}                  # device -> the zero-initialised arrival word of the in-launch finish (the kernel leaves it zero)
    IN_LAUNCH_FINISH = __import__("os").environ.get("CVAE_ELBO_TWO_LAUNCH") != "1"        # False: the two-launch form (tests compare the two bit for bit; the env switch is for A/B runs)

    @staticmethod
    def forward(ctx, src, x, m_hat, m, mu, logvar, gamma, bump=None):
        """bump: optional device int32 tensor incremented by the launch (the optimizer's device step counter, FusedAdam.claim_step_counter)."""
        L.require_gpu(src, x, m_hat, m, mu, logvar)
        src = src.contiguous()
        x, m_hat, m, mu, logvar = (t.contiguous().float() for t in (x, m_hat, m, mu, logvar))
        if m_hat.shape != m.shape or mu.shape != logvar.shape:
            raise RuntimeError(f"The size of tensor a {tuple(m_hat.shape)} must match the size of tensor b {tuple(m.shape)}")
        B, d, h, w, Cc, D, H, W = ElboUp2x.dims(src, x)
        buf = torch.empty(4 + lib.cvae_elbo_up2x_partials(B, d, h, w), dtype=torch.float32, device=x.device)
        # a backward will follow: the forward launch also leaves t1 = U_w^T (up(src) - x), so the step reads x (33.5 MB at 128^3, B = 4) once
        t1 = torch.empty(B * D * H * w, dtype=torch.float32, device=x.device) if any(ctx.needs_input_grad[i] for i in (0, 2, 4, 5)) else None
        ticket = None
        if ElboUp2x.IN_LAUNCH_FINISH:
            ticket = ElboUp2x._tickets.get(x.device)
            if ticket is None:
                ticket = ElboUp2x._tickets[x.device] = torch.zeros(33 * 32, dtype=torch.int32, device=x.device)      # CVAE_ELBO_TICKET_WORDS
        check(lib.cvae_elbo_up2x_fwd(ptr(src), ptr(x), ptr(m_hat), ptr(m), ptr(mu), ptr(logvar), float(gamma), ptr(buf), buf.data_ptr() + 16, ptr(t1), ptr(ticket),
                                     ptr(bump) if ticket is not None else None, B, d, h, w, D, H, W, m.numel(), mu.numel(), L.dtype_code(src.dtype), stream()), "elbo_up2x_fwd")
        if bump is not None and ticket is None:
            check(lib.cvae_counter_add(ptr(bump), 1, stream()), "counter_add")
        ctx.save_for_backward(src, x, m_hat, m, mu, logvar, t1)
        ctx.gamma = float(gamma)
        ctx.set_materialize_grads(False)
        return buf[0], buf[1], buf[2], buf[3]

    @staticmethod
    def backward(ctx, g_loss, g_recon, g_m, g_kld):
        src, x, m_hat, m, mu, logvar, t1 = ctx.saved_tensors
        if g_recon is not None or g_m is not None or g_kld is not None:
            raise L.CvaeError("ElboUp2x back-propagates the total loss only; for a single term use loss_function on the model's recon_x")
        if g_loss is None:
            return None, None, None, None, None, None, None, None
        B, d, h, w, Cc, D, H, W = ElboUp2x.dims(src, x)
        dsrc, d_mhat, dmu, dlv = torch.empty_like(src), torch.empty_like(m_hat), torch.empty_like(mu), torch.empty_like(mu)
        check(lib.cvae_elbo_up2x_bwd(ptr(t1), ptr(m_hat), ptr(m), ptr(mu), ptr(logvar), ctx.gamma, ptr(g_loss.contiguous()), ptr(dsrc),
                                     ptr(d_mhat), ptr(dmu), ptr(dlv), B, d, h, w, D, H, W, m.numel(), mu.numel(), L.dtype_code(src.dtype), stream()), "elbo_up2x_bwd")
        return dsrc, None, d_mhat, None, dmu, dlv, None, None


class WeightedSum(torch.autograd.Function):
    """sum_i w_i * t_i of 0-dim device tensors as one launch (and one for all the gradients): a loss composed of several terms.
    Returns (total, weighted terms [n]); only the total carries a gradient (the weighted terms are what a training loop logs)."""

    @staticmethod
    def forward(ctx, weights, *terms):
        import ctypes as C
        L.require_gpu(*terms)
        n = len(terms)
        if not 1 <= n <= 8 or len(weights) != n:
            raise L.CvaeError("WeightedSum: 1..8 terms, one weight each")
        ts = [t.float().contiguous() for t in terms]
        if any(t.numel() != 1 for t in ts):
            raise L.CvaeError("WeightedSum: scalar (0-dim) terms expected")
        out = torch.empty(n + 1, dtype=torch.float32, device=ts[0].device)
        check(lib.cvae_weighted_sum((C.c_void_p * n)(*[t.data_ptr() for t in ts]), (C.c_float * n)(*[float(w) for w in weights]), n, None, ptr(out), 0, stream()), "weighted_sum")
        ctx.weights = [float(w) for w in weights]
        ctx.shapes = [t.shape for t in terms]
        total, parts = out[0], out[1:]
        ctx.mark_non_differentiable(parts)
        ctx.set_materialize_grads(False)                     # no zeros launch for the gradient slot of `parts`
        return total, parts

    @staticmethod
    def backward(ctx, g, _gparts):
        if g is None:
            return (None,) * (1 + len(ctx.weights))
        import ctypes as C
        n = len(ctx.weights)
        out = torch.empty(n, dtype=torch.float32, device=g.device)
        check(lib.cvae_weighted_sum(None, (C.c_float * n)(*ctx.weights), n, ptr(g.float().contiguous()), ptr(out), 1, stream()), "weighted_sum_bwd")
        return (None,) + tuple(out[i].view(ctx.shapes[i]) for i in range(n))


def weighted_sum(terms, weights, return_terms=False):
    """sum_i weights[i] * terms[i]; with return_terms also the n weighted terms (a [n] tensor, no gradient)."""
    total, parts = WeightedSum.apply(tuple(weights), *terms)
    return (total, parts) if return_terms else total


def sse(a, b):
    """F.mse_loss(a, b, reduction='sum') (gradient flows to a)."""
    return _PairLoss.apply(a, b, "sse")


def bce_sum(p, x):
    """F.binary_cross_entropy(p, x, reduction='sum')."""
    return _PairLoss.apply(p, x, "bce")


class VesselRecon(torch.autograd.Function):
    """(recon_loss, sparsity_loss) of vessel_analysis/01_train/train.py:27-46 (pos-weighted MSE-sum, background L1).
    group / sync: under data parallelism pos_weight is a BATCH-GLOBAL scalar in the reference (:30-36); sync=True all-reduces (sum x) over
    `group` and uses the global element count, so every rank weights with the pos_weight of the whole batch (one 4-byte message)."""

    @staticmethod
    def forward(ctx, r, x, sync=False, group=None):
        L.require_gpu(r, x)
        r, x = r.contiguous(), x.contiguous()
        sx = _scalar(r)
        out2 = torch.zeros(2, dtype=torch.float32, device=r.device)
        _t, wp, wb = _red_ws(r)
        check(lib.cvae_sum_fwd(ptr(x), ptr(sx), x.numel(), wp, wb, stream()), "sum_fwd")
        n_pos = x.numel()
        if sync and _group_size(group) > 1:
            _all_reduce_sum(sx, group)
            n_pos *= _group_size(group)
        check(lib.cvae_wmse_sparsity_fwd(ptr(r), ptr(x), ptr(sx), ptr(out2), r.numel(), n_pos, wp, wb, stream()), "wmse_sparsity_fwd")
        ctx.save_for_backward(r, x, sx)
        ctx.n_pos = n_pos
        return out2[0], out2[1]

    @staticmethod
    def backward(ctx, g_recon, g_sp):
        r, x, sx = ctx.saved_tensors
        dr = torch.empty_like(r)
        g_recon = g_recon.contiguous() if g_recon is not None else None
        g_sp = g_sp.contiguous() if g_sp is not None else None
        check(lib.cvae_wmse_sparsity_bwd(ptr(r), ptr(x), ptr(sx), ptr(g_recon), ptr(g_sp), ptr(dr), r.numel(), ctx.n_pos, stream()), "wmse_sparsity_bwd")
        return dr, None, None, None


class GaussNLL(torch.autograd.Function):
    """0.5 * sum(logvar + (m - mu)^2 / exp(logvar))   (vessel_analysis/01_train/train.py:56-58)."""

    @staticmethod
    def forward(ctx, m, mu, logvar):
        L.require_gpu(m, mu, logvar)
        m, mu, logvar = m.contiguous(), mu.contiguous(), logvar.contiguous()
        out = _scalar(m)
        _t, wp, wb = _red_ws(m)
        check(lib.cvae_gauss_nll_fwd(ptr(m), ptr(mu), ptr(logvar), ptr(out), m.numel(), wp, wb, stream()), "gauss_nll_fwd")
        ctx.save_for_backward(m, mu, logvar)
        return out

    @staticmethod
    def backward(ctx, g):
        m, mu, logvar = ctx.saved_tensors
        g = g.contiguous()
        dmu, dlv = torch.empty_like(mu), torch.empty_like(mu)
        check(lib.cvae_gauss_nll_bwd(ptr(m), ptr(mu), ptr(logvar), ptr(g), ptr(dmu), ptr(dlv), m.numel(), stream()), "gauss_nll_bwd")
        return None, dmu, dlv


class SoftmaxCE(torch.autograd.Function):
    """F.cross_entropy(logits, target) with mean reduction (mnist_test/01_baseline_causal_vae/train.py:56)."""

    @staticmethod
    def forward(ctx, logits, target):
        L.require_gpu(logits, target)
        logits, target = logits.contiguous(), target.contiguous()
        out = _scalar(logits)
        _t, wp, wb = _red_ws(logits)
        check(lib.cvae_softmax_ce_fwd(ptr(logits), ptr(target), ptr(out), logits.shape[0], logits.shape[1], wp, wb, stream()), "softmax_ce_fwd")
        ctx.save_for_backward(logits, target)
        return out

    @staticmethod
    def backward(ctx, g):
        logits, target = ctx.saved_tensors
        g = g.contiguous()
        dl = torch.empty_like(logits)
        check(lib.cvae_softmax_ce_bwd(ptr(logits), ptr(target), ptr(g), ptr(dl), logits.shape[0], logits.shape[1], stream()), "softmax_ce_bwd")
        return dl, None


class UniformKL(torch.autograd.Function):
    """F.kl_div(F.log_softmax(logits, 1), full(1/C), reduction='batchmean') (mnist_test/01_baseline_causal_vae/train.py:82-85)."""

    @staticmethod
    def forward(ctx, logits):
        L.require_gpu(logits)
        logits = logits.contiguous()
        out = _scalar(logits)
        _t, wp, wb = _red_ws(logits)
        check(lib.cvae_uniform_kl_fwd(ptr(logits), ptr(out), logits.shape[0], logits.shape[1], wp, wb, stream()), "uniform_kl_fwd")
        ctx.save_for_backward(logits)
        return out

    @staticmethod
    def backward(ctx, g):
        (logits,) = ctx.saved_tensors
        g = g.contiguous()
        dl = torch.empty_like(logits)
        check(lib.cvae_uniform_kl_bwd(ptr(logits), ptr(g), ptr(dl), logits.shape[0], logits.shape[1], stream()), "uniform_kl_bwd")
        return dl


# ------------------------------------------------------------------------------------------------ fused bottleneck
class BioBottleneck(torch.autograd.Function):
    """Everything between CausalBioVAE's last encoder conv and first decoder conv as 5 + 5 launches (csrc/bottleneck.hip):
    pool + flatten + cat, enc_fc, fc_mu / fc_logvar, reparameterize, mechanism_net (train-mode BatchNorm1d), cat, dec_input.

    inputs : y_cl [B, D, H, W, C] (last encoder activation, a ReLU output), m [B, m_dim], t_onehot [B, t_dim] (or the int64 labels [B]), eps [B, Z],
             the 18 parameters in _lib.BOTTLENECK_PARAMS order, then (running_mean, running_var, num_batches_tracked, momentum,
             bn_eps, out_size).  returns (mu, logvar, m_hat, dec_cl [B, OD, OH, OW, C] in y_cl's dtype).
    Same arithmetic (fp32) as the layer-by-layer path, which stays the general fallback (eval mode, B > 16, odd pool windows).
    An optional last argument `sync` = (group, rank_stats) makes mechanism_net's BatchNorm1d a SyncBatchNorm over the ranks of `group`:
    rank_stats is bottleneck_bn_rank_stats(...) (this step's gathered per-rank statistics); the backward all-reduces 2 * HM floats and finishes
    mechanism_net.0's gradients in one extra small launch (include/cvae_hip.h, cvae_bottleneck_*_sync).  A further optional argument
    `noise` = EpsSource.noise_args(device) makes `eps` an output of the first launch (the draw of EpsSource.draw without its launch).
    """

    @staticmethod
    def supported(y_cl, out_size, training):
        B, D, H, W, C = y_cl.shape
        OD, OH, OW = out_size
        return training and 2 <= B <= 16 and C % 64 == 0 and D % OD == 0 and H % OH == 0 and W % OW == 0

    @staticmethod
    def forward(ctx, y_cl, m, t_onehot, eps, *rest):
        params, (rm, rv, nbt, momentum, bn_eps, out_size), sync = rest[:18], rest[18:24], (rest[24] if len(rest) > 24 else None)
        noise = rest[25] if len(rest) > 25 else None
        ctx.n_extra = len(rest) - 18
        L.require_gpu(y_cl, m, t_onehot, eps, *params)
        params = [p.contiguous() for p in params]
        W1, W2, Wmu, Wm0 = params[0], params[2], params[4], params[8]
        t_labels = None
        if t_onehot.dim() == 1 and not t_onehot.is_floating_point():           # class indices: the one-hot is written by the first launch
            t_labels = t_onehot.contiguous().long()
            t_onehot = torch.empty(t_labels.shape[0], Wm0.shape[1], dtype=torch.float32, device=y_cl.device)
        y_cl, m, t_onehot, eps = y_cl.contiguous(), m.contiguous().float(), t_onehot.contiguous().float(), eps.contiguous().float()
        B, D, H, W, C = y_cl.shape
        dims = L.BottleneckDims(B, D, H, W, C, *out_size, m.shape[1], t_onehot.shape[1], W1.shape[0], W2.shape[0], Wmu.shape[0], Wm0.shape[0])
        sizes = [C_.c_int64() for _ in range(5)]
        check(lib.cvae_bottleneck_sizes(C_.byref(dims), *[C_.byref(v) for v in sizes]), "bottleneck_sizes")
        K1, K4, n_fwd, n_dzm, n_dx = (v.value for v in sizes)
        if W1.shape[1] != K1 or params[16].shape != (C * out_size[0] * out_size[1] * out_size[2], K4):
            raise L.CvaeError(f"BioBottleneck: enc_fc.0 expects {W1.shape[1]} inputs, the pooled features + m + t give {K1}")
        dev, f32 = y_cl.device, torch.float32
        new = lambda *shape: torch.empty(*shape, dtype=f32, device=dev)
        xcat, partial, dzm_acc = new(B, K1), new(n_fwd), new(n_dzm)
        N1, N2, Z, HM, DM = dims.N1, dims.N2, dims.Z, dims.HM, dims.m_dim
        saved = dict(h1=new(B, N1), h2=new(B, N2), mu=new(B, Z), logvar=new(B, Z), xhat=new(B, HM), invstd=new(HM), a1n=new(B, HM), a2=new(B, HM),
                     m_hat=new(B, DM), zm=new(B, K4))
        dec_cl = torch.empty(B, *out_size, C, dtype=y_cl.dtype, device=dev)
        pstruct = L.BottleneckPtrs18(*[ptr(p) for p in params])
        sstruct = L.BottleneckSaved(*[ptr(saved[k]) for k in L.BOTTLENECK_SAVED])
        rank_stats, ranks = None, 0
        if sync is not None:
            rank_stats = sync[1].contiguous()
            ranks = rank_stats.shape[0]
            if tuple(rank_stats.shape) != (ranks, 2, HM) or rank_stats.dtype != f32:
                raise L.CvaeError(f"BioBottleneck: sync rank_stats must be float32 [ranks, 2, {HM}], got {tuple(rank_stats.shape)}")
        nstruct = None
        if noise is not None:
            if noise[2].dtype != torch.int32 or noise[2].device != dev:
                raise L.CvaeError("BioBottleneck: the noise call counter must be an int32 tensor on the activations' device")
            nstruct = C_.byref(L.BottleneckNoise(int(noise[0]), int(noise[1]), ptr(noise[2])))
        check(lib.cvae_bottleneck_fwd_ex(C_.byref(dims), C_.byref(pstruct), ptr(y_cl), ptr(m), ptr(t_onehot), ptr(t_labels), ptr(eps), ptr(rm), ptr(rv), ptr(nbt), float(momentum),
                                         float(bn_eps), 1, ptr(xcat), ptr(partial), ptr(dzm_acc), C_.byref(sstruct), ptr(dec_cl), L.dtype_code(y_cl.dtype), ptr(rank_stats), ranks,
                                         nstruct, stream()), "bottleneck_fwd")
        ctx.dims, ctx.scratch = dims, n_dx
        ctx.sync = None if sync is None else (sync[0], ranks)
        ctx.save_for_backward(y_cl, t_onehot, eps, xcat, *params, *[saved[k] for k in L.BOTTLENECK_SAVED], dzm_acc)
        ctx.mark_non_differentiable(*[t for t in (rm, rv, nbt) if t is not None])
        return saved["mu"], saved["logvar"], saved["m_hat"], dec_cl

    @staticmethod
    def backward(ctx, g_mu, g_logvar, g_mhat, g_dec):
        y_cl, t_onehot, eps, xcat = ctx.saved_tensors[:4]
        params, saved, dzm_part = ctx.saved_tensors[4:22], ctx.saved_tensors[22:-1], ctx.saved_tensors[-1]
        dims, n_dx = ctx.dims, ctx.scratch
        dev, f32 = y_cl.device, torch.float32
        if g_dec is None:
            g_dec = torch.zeros(dims.M, dims.OD, dims.OH, dims.OW, dims.C, dtype=y_cl.dtype, device=dev)
        fix = lambda g: None if g is None else g.contiguous().float()
        g_mu, g_logvar, g_mhat, g_dec = fix(g_mu), fix(g_logvar), fix(g_mhat), g_dec.contiguous()
        grads = [torch.empty_like(p) for p in params]
        g1, dx_part = (torch.empty(n, dtype=f32, device=dev) for n in (dims.M * (dims.N1 + dims.N2), n_dx))
        dy_cl = torch.empty_like(y_cl)
        pstruct = L.BottleneckPtrs18(*[ptr(p) for p in params])
        gstruct = L.BottleneckPtrs18(*[ptr(g) for g in grads])
        sstruct = L.BottleneckSaved(*[ptr(t) for t in saved])
        bn_dy = bn_sums = None
        if ctx.sync is not None:
            bn_dy, bn_sums = torch.empty(dims.M, dims.HM, dtype=f32, device=dev), torch.empty(2, dims.HM, dtype=f32, device=dev)
        check(lib.cvae_bottleneck_bwd_sync(C_.byref(dims), C_.byref(pstruct), C_.byref(gstruct), C_.byref(sstruct), ptr(g_dec), ptr(g_mu), ptr(g_logvar), ptr(g_mhat),
                                           ptr(t_onehot), ptr(eps), ptr(xcat), ptr(y_cl), 1, ptr(dzm_part), ptr(g1), ptr(dx_part), ptr(dy_cl),
                                           L.dtype_code(y_cl.dtype), ptr(bn_dy), ptr(bn_sums), stream()), "bottleneck_bwd")
        if ctx.sync is not None:
            group, ranks = ctx.sync
            _all_reduce_sum(bn_sums, group)                  # 2 * HM floats: sum(dy), sum(dy * xhat) over the global batch
            check(lib.cvae_bottleneck_bn_bwd_finish(C_.byref(dims), C_.byref(pstruct), C_.byref(gstruct), C_.byref(sstruct), ptr(t_onehot), ptr(bn_dy), ptr(bn_sums), ranks,
                                                    stream()), "bottleneck_bn_bwd_finish")
        return (dy_cl, None, None, None, *grads, *([None] * ctx.n_extra))


def bottleneck_bn_rank_stats(Wm0, bm0, t, group=None):
    """Per-rank statistics of mechanism_net.0's output for the SyncBatchNorm form of BioBottleneck: float32 [ranks, 2, HM] = every rank's (sum, squared
    deviations from its own mean) over its B samples — one small launch and one all-gather of 2 * HM floats.  The layer's input is t alone, so this runs at the
    top of the step, before the encoder.  t: int64 labels [B] or the float one-hot [B, t_dim].  Every rank must hold the same B."""
    import torch.distributed as dist
    L.require_gpu(Wm0, bm0, t)
    Wm0, bm0 = Wm0.contiguous(), bm0.contiguous()
    HM, T = Wm0.shape
    labels = t.contiguous().long() if (t.dim() == 1 and not t.is_floating_point()) else None
    onehot = None if labels is not None else t.contiguous().float()
    B = t.shape[0]
    local = torch.empty(2, HM, dtype=torch.float32, device=Wm0.device)
    check(lib.cvae_bottleneck_bn_local_stats(ptr(Wm0), ptr(bm0), ptr(onehot), ptr(labels), ptr(local), B, T, HM, stream()), "bottleneck_bn_local_stats")
    world = _group_size(group)
    if world == 1:
        return local.unsqueeze(0)
    parts = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(parts, local, group=group)
    return torch.stack(parts)


# ------------------------------------------------------------------------------------------------ CausalVesselVAE extras
class BatchNorm2dAct(torch.autograd.Function):
    """nn.BatchNorm2d (+ the activation that follows it) on a channels-last tensor [B, D, H, W, C] (csrc/vessel2d.hip)."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum, eps, training, act):
        L.require_gpu(x, weight, bias)
        x = x.contiguous()
        Cc = x.shape[-1]
        P = x.numel() // Cc
        y = torch.empty_like(x)
        if training:
            mean, rstd = _empty((Cc,), torch.float32, x), _empty((Cc,), torch.float32, x)
        else:
            mean, rstd = running_mean.float().clone(), torch.rsqrt(running_var.float() + eps)
        _t, wp, wb = _scratch(lib.cvae_bn2d_workspace_bytes(P, Cc) if training else 0, x)
        check(lib.cvae_bn2d_fwd(ptr(x), ptr(weight), ptr(bias), ptr(y), ptr(mean), ptr(rstd), ptr(running_mean) if training else None,
                                ptr(running_var) if training else None, P, Cc, float(momentum), float(eps), 1 if training else 0, L.act_code(act),
                                L.dtype_code(x.dtype), wp, wb, stream()), "bn2d_fwd")
        ctx.save_for_backward(x, y, weight, mean, rstd)
        ctx.cfg = (training, act)
        return y

    @staticmethod
    def backward(ctx, g):
        x, y, weight, mean, rstd = ctx.saved_tensors
        training, act = ctx.cfg
        if not training:
            raise L.CvaeError("BatchNorm2d in eval mode is forward-only here (the reference consumers run it under no_grad)")
        Cc = x.shape[-1]
        P = x.numel() // Cc
        g = g.contiguous()
        dx = torch.empty_like(x)
        dw, db = _empty((Cc,), torch.float32, x), _empty((Cc,), torch.float32, x)
        _t, wp, wb = _scratch(lib.cvae_bn2d_workspace_bytes(P, Cc), x)
        check(lib.cvae_bn2d_bwd(ptr(x), ptr(g), ptr(y), ptr(weight), ptr(mean), ptr(rstd), ptr(dx), ptr(dw), ptr(db), P, Cc, L.act_code(act),
                                L.dtype_code(x.dtype), wp, wb, stream()), "bn2d_bwd")
        return dx, dw, db, None, None, None, None, None, None


class Clamp(torch.autograd.Function):
    """torch.clamp(x, min, max) with its pass-through gradient mask (vessel_analysis/00_core/models.py:148-149,156)."""

    @staticmethod
    def forward(ctx, x, lo, hi):
        L.require_gpu(x)
        x = x.contiguous().float()
        y = torch.empty_like(x)
        check(lib.cvae_clamp_fwd(ptr(x), ptr(y), float(lo), float(hi), x.numel(), stream()), "clamp_fwd")
        ctx.save_for_backward(x)
        ctx.lim = (float(lo), float(hi))
        return y

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        g = g.contiguous()
        dx = torch.empty_like(x)
        check(lib.cvae_clamp_bwd(ptr(x), ptr(g), ptr(dx), ctx.lim[0], ctx.lim[1], x.numel(), stream()), "clamp_bwd")
        return dx, None, None


class Conv3ToK4(torch.autograd.Function):
    """Weight of nn.Conv2d(Cin, Cout, 3, 1, 1) applied after a nearest x2 upsample -> the equivalent transposed k4/s2/p1 weight
    [Cin][Cout][4][4] for ConvUp (K4 = A W3 A^T, csrc/vessel2d.hip); the backward maps dK4 to dW3."""

    @staticmethod
    def forward(ctx, w3):
        L.require_gpu(w3)
        w3 = w3.contiguous()
        Cout, Cin = w3.shape[0], w3.shape[1]
        if tuple(w3.shape[2:]) != (3, 3):
            raise L.CvaeError("Conv3ToK4: a [Cout, Cin, 3, 3] weight expected")
        k4 = torch.empty(Cin, Cout, 4, 4, dtype=torch.float32, device=w3.device)
        check(lib.cvae_conv3_to_k4(ptr(w3), ptr(k4), Cout, Cin, stream()), "conv3_to_k4")
        ctx.shape = (Cout, Cin)
        return k4

    @staticmethod
    def backward(ctx, g):
        Cout, Cin = ctx.shape
        g = g.contiguous()
        dw3 = torch.empty(Cout, Cin, 3, 3, dtype=torch.float32, device=g.device)
        check(lib.cvae_k4_to_conv3_grad(ptr(g), ptr(dw3), Cout, Cin, stream()), "k4_to_conv3_grad")
        return dw3
