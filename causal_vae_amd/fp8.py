"""fp8 (OCP e4m3) forward of the conv stacks inside a training step — BASELINE.json configs[4], "as configs[3] with fp8 conv inputs".

The step is the reference's (causal_cascade/train.py:19-39); what changes is the arithmetic of the forward products with C_in >= 32
(enc_conv[2..6], dec_conv[0..4] of causal_cascade/models.py:12-20, 50-55): fp8 operands with per-tensor scales, fp32 accumulation on the
block-scaled CDNA4 MFMA (csrc/conv_mfma.hip, cvae_conv_fp8), twice the bf16 rate.  The backward pass is the bf16 one: every fp8 layer
leaves its result twice, bf16 for the backward pass (ReLU masks, weight gradients) and fp8 codes for the next layer.

Scaling is *delayed*: every producer records the largest value it wrote (amax), and once per step one tiny launch turns the records into
the scales of the NEXT step (ops.Fp8Scales).  Everything lives on the device, so the step stays one capturable HIP graph.  The very first
step of a model calibrates instead: it runs the bf16 kernels and only records the amaxes.
"""
import torch

from . import ops
from ._lib import CvaeError


def _eligible(weight, for_up):
    Cs, Cl = weight.shape[0], weight.shape[1]
    return (Cs % 32 == 0 and Cl % 32 == 0 and Cl > 1) if for_up else (Cl % 32 == 0 and Cs % 64 == 0)


class Fp8Forward:
    """State of the fp8 forward of one model: which conv layers run fp8, their device scales, amax records and fp8 weight panels.

    `stacks` = [(name, [conv modules in order], for_up)], e.g. [("enc", enc convs, False), ("dec", dec convs, True)]."""

    def __init__(self, stacks, nd, device, headroom=2.0):
        self.nd, self.device = nd, device
        self.layers = {}                                     # (stack name, conv index) -> layer record
        tensors, recs = 0, []
        for name, convs, for_up in stacks:
            prev = None
            for j, conv in enumerate(convs):
                if not _eligible(conv.weight, for_up):
                    prev = None
                    continue
                rec = dict(stack=name, j=j, conv=conv, for_up=for_up, a_in=tensors, w=tensors + 1, out=-1, l=len(recs))
                tensors += 2
                if prev is not None:
                    prev["out"] = rec["a_in"]                # the previous fp8 layer writes this layer's input codes itself
                    rec["fed"] = True
                else:
                    rec["fed"] = False                       # a bf16 producer: the input is quantised in front of this layer
                self.layers[(name, j)] = rec
                recs.append(rec)
                prev = rec
        if not recs:
            raise CvaeError("fp8 forward: no conv layer of this model qualifies (C_in % 32 == 0 and C_out % 64 / 32 == 0)")
        self.recs = recs
        self.scales = ops.Fp8Scales(tensors, [(r["a_in"], r["w"], r["out"]) for r in recs], device, headroom)
        self.panels = [torch.empty(r["conv"].weight.numel(), dtype=torch.uint8, device=device) for r in recs]
        self.calibrated = False
        self.calibrating = False

    # ---- per step -------------------------------------------------------------------------------------------
    def begin_step(self):
        """Scales of this step from the amaxes of the last one (one launch); the fp8 weight panels are written by the model's one weight-pack launch
        (pack_spec).  The first step calibrates: bf16 arithmetic, amaxes recorded."""
        self.calibrating = not self.calibrated
        if self.calibrating:
            for r in self.recs:
                ops.absmax(r["conv"].weight.detach(), self.scales.amax[r["w"]])
            return
        self.scales.update()

    def pack_spec(self):
        """What ops.pack_weights needs to write this step's fp8 panels: {id(weight): (direction, panel, 1 / s_w on the device, amax record)}; None while calibrating."""
        if self.calibrating:
            return None
        sc = self.scales
        return {id(r["conv"].weight): (2 if r["for_up"] else 1, self.panels[r["l"]], sc.inv_scale[r["w"]:r["w"] + 1], sc.amax[r["w"]]) for r in self.recs}

    def end_step(self):
        if self.calibrating:
            self.scales.update()                             # records -> the scales of step 1
            self.calibrating, self.calibrated = False, True

    def layer(self, stack, j):
        return self.layers.get((stack, j))

    def side_for(self, stack, j_next, conv, h, out_dtype):
        """The side-output request for the bf16 layer `conv` in front of fp8 layer (stack, j_next), or None: only the single-channel image layer's
        kernel writes codes itself (cvae_conv_down_image_f8); any other producer is followed by a quantise launch (see run)."""
        rec = self.layers.get((stack, j_next))
        if rec is None or rec["fed"] or self.calibrating:
            return None
        if conv.weight.shape[1] != 1 or (out_dtype or h.dtype) != torch.bfloat16 or not isinstance(conv, torch.nn.modules.conv._ConvNd) or conv.transposed:
            return None
        if not ops.image_direct_ok(h, torch.bfloat16) and h.dtype != torch.bfloat16:
            return None
        sc = self.scales
        return dict(side=True, inv_scale=sc.inv_scale[rec["a_in"]:rec["a_in"] + 1], amax=sc.amax[rec["a_in"]])

    def run(self, rec, conv, h, prev8, act, in_relu, premasked, packed):
        """One fp8 layer: returns (bf16 result, fp8 codes of it or None).  h: the bf16 input (kept for the backward pass); prev8: its codes when
        the previous layer was an fp8 layer."""
        sc = self.scales
        if self.calibrating:
            ops.absmax(h.detach(), sc.amax[rec["a_in"]])
            return conv.forward_cl(h, act=act, in_is_relu_out=in_relu, grad_premasked=premasked, packed=packed), None
        xq = prev8 if prev8 is not None else ops.quantize_fp8_dev(h.detach(), sc.inv_scale[rec["a_in"]:rec["a_in"] + 1], sc.amax[rec["a_in"]])
        f8 = dict(xq=xq, wq=self.panels[rec["l"]], dscale=sc.dscale[rec["l"]], want_out8=rec["out"] >= 0, amax=(sc.amax[rec["out"]] if rec["out"] >= 0 else None))
        y = conv.forward_cl(h, act=act, in_is_relu_out=in_relu, grad_premasked=premasked, packed=packed, f8=f8)
        return y, f8.get("y8")

    def state(self):
        return dict(self.scales.state(), calibrated=self.calibrated)

    def load_state(self, st):
        self.scales.load_state(st)
        self.calibrated = bool(st["calibrated"])
