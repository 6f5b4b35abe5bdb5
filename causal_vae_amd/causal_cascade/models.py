"""CausalBioVAE on MI355X — drop-in for the reference class (causal_cascade/models.py:5-89) and its 3D lift.

Same constructor, same sub-module attribute tree (enc_conv / enc_fc / fc_mu / fc_logvar / mechanism_net / dec_input /
dec_conv), same state_dict keys and shapes, same forward(x, m, t) -> (recon_x, m_hat, mu, logvar); every sub-module stays
individually callable (the reference's analysis scripts call `model.mechanism_net(...)`, causal_cascade/analyze.py:15,23).
`CausalBioVAE3D` is the volume model of SURVEY.md §8(a): Conv3d / ConvTranspose3d / AdaptiveAvgPool3d((4,4,4)) /
trilinear, flatten_dim 16384.
"""
import torch
import torch.nn as nn

from .. import layers as hl
from .. import ops


class CausalBioVAE(nn.Module):
    _ND = 2

    def __init__(self, img_channels=1, m_dim=12, t_dim=19, latent_dim=64):
        super().__init__()
        nd = self._ND
        Conv = hl.Conv2d if nd == 2 else hl.Conv3d
        ConvT = hl.ConvTranspose2d if nd == 2 else hl.ConvTranspose3d
        Pool = nn.AdaptiveAvgPool2d if nd == 2 else nn.AdaptiveAvgPool3d
        self.m_dim, self.t_dim = m_dim, t_dim
        # layer construction order == the reference's, so a given torch seed yields the reference's weights
        self.enc_conv = hl.ConvStack(
            Conv(img_channels, 32, 4, 2, 1), nn.ReLU(),
            Conv(32, 64, 4, 2, 1), nn.ReLU(),
            Conv(64, 128, 4, 2, 1), nn.ReLU(),
            Conv(128, 256, 4, 2, 1), nn.ReLU(),
            Pool((4,) * nd), nn.Flatten())
        self.flatten_dim = 256 * 4 ** nd
        self.enc_fc = hl.MLP(hl.Linear(self.flatten_dim + m_dim + t_dim, 512), nn.ReLU(), hl.Linear(512, 256), nn.ReLU())
        self.fc_mu = hl.Linear(256, latent_dim)
        self.fc_logvar = hl.Linear(256, latent_dim)
        self.mechanism_net = hl.MLP(hl.Linear(t_dim, 64), hl.BatchNorm1d(64), nn.ReLU(), hl.Linear(64, 64), nn.ReLU(),
                                    hl.Linear(64, m_dim))
        self.dec_input = hl.Linear(latent_dim + m_dim, self.flatten_dim)
        self.dec_conv = hl.DeconvStack(
            ConvT(256, 128, 4, 2, 1), nn.ReLU(),
            ConvT(128, 64, 4, 2, 1), nn.ReLU(),
            ConvT(64, 32, 4, 2, 1), nn.ReLU(),
            ConvT(32, img_channels, 4, 2, 1))
        self._eps = ops.EpsSource()

    # ---- precision -------------------------------------------------------------------------------------------
    def set_compute_dtype(self, dtype):
        """torch.float32 (exact-fp32 MFMA; parity mode) or torch.bfloat16 (bf16 MFMA, fp32 accumulate) for the convs."""
        hl.set_compute_dtype(self, dtype)
        return self

    def set_fp8_forward(self, on=True, headroom=2.0):
        """Training forward of the conv layers with C_in >= 32 on fp8 (e4m3) operands (BASELINE.json configs[4]; causal_vae_amd.fp8): the bf16 model's
        step with the forward products of enc_conv[2..6] / dec_conv[0..4] on the block-scaled MFMA; the backward pass stays bf16.  Needs the bf16
        compute dtype and the fused training path (2 <= B <= 16); the first step after switching it on calibrates the scales (and runs bf16)."""
        if on and self.enc_conv.compute_dtype != torch.bfloat16:
            raise ops.L.CvaeError("set_fp8_forward: set_compute_dtype(torch.bfloat16) first (the backward pass and the single-channel layers run bf16)")
        self._fp8_on, self._fp8_headroom = bool(on), float(headroom)
        if not on:
            self._fp8 = None
        return self

    _fp8_on, _fp8, _fp8_headroom = False, None, 2.0

    def _fp8_plan(self, device):
        if not self._fp8_on:
            return None
        if self._fp8 is None or self._fp8.device != device:
            from ..fp8 import Fp8Forward
            convs = lambda stack: [m_ for m_ in stack if isinstance(m_, hl._ConvBase)]
            self._fp8 = Fp8Forward([("enc", convs(self.enc_conv), False), ("dec", convs(self.dec_conv), True)], self._ND, device, self._fp8_headroom)
        return self._fp8

    # ---- reference surface -----------------------------------------------------------------------------------
    def encode(self, x, m, t_onehot):
        h = self.enc_fc(self.enc_conv.forward_cat(x, [m, t_onehot]))
        return self.fc_mu(h), self.fc_logvar(h)

    def reparameterize(self, mu, logvar, eps=None):
        """z = mu + eps*exp(logvar/2).  eps defaults to a device Philox draw keyed by torch's seed (the CPU generator stream
        of the reference cannot be reproduced on a GPU — parity runs pass `eps` explicitly)."""
        if eps is None:
            eps = self._eps.draw(mu)
        return ops.Reparameterize.apply(mu, logvar, eps)

    def decode_cl(self, z_m_input):
        x_feat = self.dec_input(z_m_input).view(-1, 256, *([4] * self._ND))
        return self.dec_conv.forward_cl(x_feat)              # channels-last [B, D, H, W, C], compute dtype

    def calibrate_fp8_decoder(self, z, m_hat, headroom=1.0, c1_fp8_input=True):
        """Scales + fp8 weight panels for decode(..., fp8_plan=...) from one bf16 pass over calibration rows (DeconvStack.calibrate_fp8)."""
        with torch.no_grad():
            x_feat = self.dec_input(ops.cat([z, m_hat])).view(-1, 256, *([4] * self._ND))
            return self.dec_conv.calibrate_fp8(ops.ToChannelsLast.apply(x_feat, torch.bfloat16), headroom, c1_fp8_input)

    def decode(self, z, m_hat, size=None, fp8_plan=None):
        """Decoder half only: [z, m_hat] -> dec_input -> dec_conv -> resize to `size` (default: the native 64^nd).
        Rows are independent, so a whole counterfactual sweep (abduct z once, stack every intervened m') decodes in ONE call
        instead of the reference's per-value loop (vessel_analysis/04_generate_counterfactual/generate_counterfactual.py:77-99).
        fp8_plan (from calibrate_fp8_decoder; inference only): the ConvTranspose layers with C_out > 1 run on fp8 (e4m3) operands."""
        nd = self._ND
        if fp8_plan is not None:
            with torch.no_grad():
                x_feat = self.dec_input(ops.cat([z, m_hat])).view(-1, 256, *([4] * nd))
                out_cl = self.dec_conv.forward_fp8(ops.ToChannelsLast.apply(x_feat, torch.bfloat16), fp8_plan)
        else:
            out_cl = self.decode_cl(ops.cat([z, m_hat]))
        native = tuple(out_cl.shape[1:4])
        size = native if size is None else (tuple(size) if nd == 3 else (1,) + tuple(size))
        rec = ops.Cast.apply(out_cl, torch.float32) if size == native else ops.UpsampleLinear.apply(out_cl, size)
        if out_cl.shape[-1] != 1:
            return ops.FromChannelsLast.apply(rec, nd)
        return rec.view(rec.shape[0], 1, *(size if nd == 3 else size[1:]))

    fuse_bottleneck = True     # training forward: run pool .. dec_input as ops.BioBottleneck (4 + 4 launches) when the shapes allow

    def _fused_bottleneck(self, x, m, t, eps):
        """The same computation as encode -> reparameterize -> mechanism_net -> dec_input, layer for layer, in csrc/bottleneck.hip.
        Returns None (caller takes the layer-by-layer path) in eval mode, for B > 16 / B == 1, or pool windows that do not tile."""
        nd = self._ND
        out_size = (4,) * 3 if nd == 3 else (1, 4, 4)
        bn = self.mechanism_net[1]
        # decide before any kernel runs: training-mode BN, 2 <= B <= 16, and the /16 encoder output must tile into 4^nd windows
        sp = [s // 16 for s in x.shape[2:]]
        if (not self.training or not torch.is_grad_enabled() or not (2 <= x.shape[0] <= 16) or any(s < 4 or s % 4 for s in sp)
                or any(s % 16 for s in x.shape[2:]) or not bn.track_running_stats or bn.momentum is None):
            return None
        sync = (None,)
        if getattr(bn, "sync", False):
            # SyncBatchNorm (parallel.convert_sync_batchnorm): mechanism_net.0 sees t only, so its per-rank statistics are gathered here, ahead of the encoder
            sync = ((bn.sync_group, ops.bottleneck_bn_rank_stats(self.mechanism_net[0].weight, self.mechanism_net[0].bias, t, bn.sync_group)),)
        we, wd = self.enc_conv.conv_weights(), self.dec_conv.conv_weights()
        f8 = self._fp8_plan(x.device)
        if f8 is not None:
            f8.begin_step()
        packed = ops.pack_weights(we + wd, nd, self.enc_conv.compute_dtype, f8spec=f8.pack_spec() if f8 is not None else None)   # every conv weight of the model, one launch
        h, rest, last_act = self.enc_conv.features_cl(x, packed=packed[:len(we)], f8=f8)
        self._enc_out = h                                    # graph.GraphedTrainStep splits the backward here (exchange overlap)
        if not ops.BioBottleneck.supported(h, out_size, True) or last_act != "relu":
            raise ops.L.CvaeError("fused bottleneck: unexpected encoder output " + str(tuple(h.shape)))
        noise = None
        if eps is None:                                      # the draw of self._eps.draw(...), made by the bottleneck's first launch
            eps = torch.empty(x.shape[0], self.fc_mu.out_features, dtype=torch.float32, device=x.device)
            noise = self._eps.noise_args(x.device)
        lin = [self.enc_fc[0], self.enc_fc[2], self.fc_mu, self.fc_logvar, self.mechanism_net[0]]
        params = [p for l in lin for p in (l.weight, l.bias)] + [bn.weight, bn.bias]
        params += [p for l in (self.mechanism_net[3], self.mechanism_net[5], self.dec_input) for p in (l.weight, l.bias)]
        if t.dim() != 1 or t.dtype != torch.int64:          # BioBottleneck builds the one-hot itself from int64 labels (one_hot raises otherwise)
            t = ops.one_hot(t, self.t_dim)
        mu, logvar, m_hat, dec_cl = ops.BioBottleneck.apply(h, m, t, eps, *params, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                                                            bn.momentum, bn.eps, out_size, *sync, noise)
        out_cl = self.dec_conv.forward_from_cl(dec_cl, packed=packed[len(we):], f8=f8)
        if f8 is not None:
            f8.end_step()
        return mu, logvar, m_hat, out_cl

    def _forward_cl(self, x, m, t, eps):
        """Everything up to the decoder output, channels-last [B, d, h, w, C] in the conv dtype (before the resize to x's size)."""
        nd = self._ND
        if x.dim() != nd + 2:
            raise RuntimeError(f"{type(self).__name__} expects a {nd + 2}-D input [B, C, {'D, ' if nd == 3 else ''}H, W], got {tuple(x.shape)}")
        ops.L.require_gpu(x, m, t)                          # no CPU fallback: fail before any launch is attempted
        fused = self._fused_bottleneck(x, m, t, eps) if self.fuse_bottleneck else None
        if fused is not None:
            mu, logvar, m_hat, out_cl = fused
        else:
            t_onehot = ops.one_hot(t, self.t_dim)
            mu, logvar = self.encode(x, m, t_onehot)
            z = self.reparameterize(mu, logvar, eps)
            m_hat = self.mechanism_net(t_onehot)
            out_cl = self.decode_cl(ops.cat([z, m_hat]))
        return out_cl, m_hat, mu, logvar

    def _resize_to(self, out_cl, x):
        nd = self._ND
        size = tuple(x.shape[2:]) if nd == 3 else (1,) + tuple(x.shape[2:])
        if tuple(out_cl.shape[1:4]) == size:
            recon_cl = ops.Cast.apply(out_cl, torch.float32)   # F.interpolate to the same size is the identity
        else:
            recon_cl = ops.UpsampleLinear.apply(out_cl, size)
        B, C = x.shape[0], out_cl.shape[-1]
        if C != 1:
            return ops.FromChannelsLast.apply(recon_cl, nd)
        return recon_cl.view(B, 1, *x.shape[2:])               # C == 1: channels-last and NC(D)HW coincide

    def forward(self, x, m, t, eps=None):
        out_cl, m_hat, mu, logvar = self._forward_cl(x, m, t, eps)
        return self._resize_to(out_cl, x), m_hat, mu, logvar

    def forward_elbo(self, x, m, t, eps=None, gamma=2000.0, bump=None):
        """forward + loss_function(recon_x, x, m_hat, m, mu, logvar, gamma) -> (loss, recon_loss, m_loss) in one call.  When the resize
        is the exact 2x of the benchmark shape the reconstruction term is computed from the decoder output directly
        (ops.ElboUp2x): the resized volume is never written; otherwise this is literally forward() followed by the ELBO node."""
        out_cl, m_hat, mu, logvar = self._forward_cl(x, m, t, eps)
        # bump: a device step counter (FusedAdam.claim_step_counter) that this step's launches must advance by one; the ELBO launch takes it along
        if self.fuse_recon_loss and ops.ElboUp2x.supported(out_cl, x):
            loss, recon, m_loss, _ = ops.ElboUp2x.apply(out_cl, x, m_hat, m, mu, logvar, gamma, bump)
        else:
            loss, recon, m_loss, _ = ops.Elbo.apply(self._resize_to(out_cl, x), x, m_hat, m, mu, logvar, gamma)
            if bump is not None:
                ops.check(ops.lib.cvae_counter_add(ops.ptr(bump), 1, ops.stream()), "counter_add")
        return loss, recon, m_loss

    fuse_recon_loss = True

    _enc_out = None

    def early_gradient_parameters(self):
        """Parameters whose gradients are complete before the encoder's backward starts (everything but enc_conv): candidates for
        FusedAdam.overlap_backward."""
        enc = {id(p) for p in self.enc_conv.parameters()}
        return [p for p in self.parameters() if id(p) not in enc]


class CausalBioVAE3D(CausalBioVAE):
    """The 3D vessel-volume model: x is [B, 1, D, H, W]."""
    _ND = 3
