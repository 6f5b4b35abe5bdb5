/* cvae_hip.h — C ABI of libcvae_hip.so: the MI355X (gfx950) kernels behind the CausalVAE train step.
 *
 * Boundary contract (SURVEY.md §8(b) "lower boundary"):
 *   - extern "C", plain device pointers + sizes; no torch / C++ types.
 *   - every call only ENQUEUES work on `stream` (a hipStream_t passed as void*); it never allocates or
 *     frees device memory, never synchronises, never throws.  All buffers (inputs, outputs, workspace)
 *     are owned by the caller (the torch caching allocator on the Python side).
 *   - return 0 (CVAE_OK) or a negative CVAE_E_* code; cvae_strerror() names it.
 *
 * Tensor conventions
 *   - conv activations are CHANNELS-LAST: [B, D, H, W, C] (D == 1 for 2D), dtype = CVAE_F32 or CVAE_BF16.
 *   - every k4/s2/p1 convolution pair is described by its SMALL tensor S [B, sd, sh, sw, Cs] and its LARGE
 *     tensor L [B, ld, lh, lw, Cl] with l = 2*s - 1 + k (k = 0..3) per strided dim; nd = 2 keeps D unstrided
 *     (sd == ld == 1, one depth tap).  The weight W is fp32 [Cs][Cl][taps] — exactly nn.Conv{2,3}d.weight
 *     ([C_out][C_in][k..], S = output) and nn.ConvTranspose{2,3}d.weight ([C_in][C_out][k..], S = input).
 *       down : S = act(gather(L, W) + bias)   = Conv forward            = ConvTranspose backward-data
 *       up   : L = act(scatter(S, W) + bias)  = ConvTranspose forward   = Conv backward-data
 *       wgrad: dW[cs][cl][k] = sum S[.., cs] * L[2s-1+k, cl]            = both weight gradients
 *   - linear layers, losses, BN, reparameterisation and Adam are fp32.
 *
 * Reference interfaces replaced (file:line in /root/reference):
 *   conv_down/up/wgrad ........ nn.Conv2d / nn.ConvTranspose2d fwd+bwd   causal_cascade/models.py:12-16,50-55;
 *                                                                        mnist_test/01_baseline_causal_vae/models.py:19-23,45-48
 *   adaptive_avgpool .......... nn.AdaptiveAvgPool2d((4,4)) + Flatten    causal_cascade/models.py:18-19
 *   upsample_linear ........... F.interpolate(bilinear, align_corners=False)  causal_cascade/models.py:87
 *   linear_* .................. nn.Linear (+ReLU/LeakyReLU)              causal_cascade/models.py:24-31,35-44
 *   bn1d_* .................... nn.BatchNorm1d(64)                       causal_cascade/models.py:36
 *   reparam_kld_* ............. reparameterize + KLD term                causal_cascade/models.py:65-68, train.py:13
 *   sse / bce / wmse_sparsity / gauss_nll .. loss terms                  causal_cascade/train.py:7,10;
 *                                            mnist_test/01_baseline_causal_vae/train.py:70; vessel_analysis/01_train/train.py:27-58
 *   softmax_ce / uniform_kl ... F.cross_entropy, F.kl_div(log_softmax)   mnist_test/01_baseline_causal_vae/train.py:56,82-85
 *   adam_step / sqnorm ........ optim.Adam.step, clip_grad_norm_         causal_cascade/main.py:50, train.py:34;
 *                                                                        vessel_analysis/01_train/train.py:85-86
 */
#ifndef CVAE_HIP_H
#define CVAE_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CVAE_OK             0
#define CVAE_E_BADSHAPE    (-1)
#define CVAE_E_DTYPE       (-2)
#define CVAE_E_UNSUPPORTED (-3)
#define CVAE_E_WORKSPACE   (-4)
#define CVAE_E_LAUNCH      (-5)
#define CVAE_E_NULLPTR     (-6)

#define CVAE_F32  0
#define CVAE_BF16 1
#define CVAE_FP8  2          /* OCP e4m3 codes with a per-tensor scale kept by the caller (cvae_conv_fp8 ...) */

#define CVAE_ACT_NONE    0
#define CVAE_ACT_RELU    1
#define CVAE_ACT_SIGMOID 2
#define CVAE_ACT_LEAKY02 3   /* LeakyReLU(0.2) */

int         cvae_version(void);
const char* cvae_strerror(int code);

/* ---- layout / precision plumbing -------------------------------------------------------------- */
/* [B, C, S] (torch NC(D)HW, S = spatial voxels) -> channels-last [B, S, C], converting dtype on the way. */
int cvae_ncs_to_nsc(const void* src, void* dst, int64_t B, int64_t C, int64_t S, int src_dtype, int dst_dtype, void* stream);
int cvae_nsc_to_ncs(const void* src, void* dst, int64_t B, int64_t C, int64_t S, int src_dtype, int dst_dtype, void* stream);
int cvae_cast(const void* src, void* dst, int64_t n, int src_dtype, int dst_dtype, void* stream);
/* dst[b, col0 + j] = src[b, j] for j < cols (fp32): writes one panel of a concatenated [B, dst_stride] matrix. */
int cvae_copy_panel(const float* src, float* dst, int64_t B, int64_t cols, int64_t src_stride, int64_t dst_stride, int64_t col0, void* stream);
/* Up to 8 panels in one launch (host arrays of device pointers / widths / row strides): gather == 0: wide[b][col0 + off_i + j] = panels[i][b * strides[i] + j]
 * with off_i = widths[0] + .. + widths[i-1] (torch.cat(dim=1)); gather != 0: the reverse copy (the column ranges of `wide` out into the panels: cat's backward). */
int cvae_copy_panels(float* const* panels, const int64_t* widths, const int64_t* strides, int count, float* wide, int64_t B, int64_t wide_stride, int64_t col0,
                     int gather, void* stream);
/* dst[b, col0 + t[b]] = 1, other columns of the panel 0 (F.one_hot(t, n).float() into a concat panel). */
int cvae_onehot_panel(const int64_t* t, float* dst, int64_t B, int64_t n_classes, int64_t dst_stride, int64_t col0, void* stream);

/* ---- k4 s2 p1 convolution family ---------------------------------------------------------------- */
/* fp32 W [Cs][Cl][taps] -> MFMA operand panels in the compute dtype.
 *   for_up == 0: [tap][Cl/16][Cs][16]   (down: N = Cs, K = (tap, cl))
 *   for_up == 1: [tap][Cs/16][Cl][16]   (up  : N = Cl, K = (tap, cs))
 * Not needed (pass the fp32 W itself as `w`) when Cl == 1. */
size_t cvae_conv_packed_weight_bytes(int64_t Cs, int64_t Cl, int nd, int dtype);
int cvae_conv_pack_weight(const float* w, void* packed, int64_t Cs, int64_t Cl, int nd, int for_up, int dtype, void* stream);

/* The same packing for a LIST of weights in one launch (host arrays of `count` entries; all tensors share nd and dtype). */
int cvae_conv_pack_weights(const float* const* w, void* const* packed, const int64_t* Cs, const int64_t* Cl, const int* for_up,
                           int count, int nd, int dtype, void* stream);

/* Both packings (for_up = 0 and 1) of a LIST of weights in one launch through an LDS transpose (Cs and Cl multiples of 16). */
int cvae_conv_pack_weight_pairs(const float* const* w, void* const* packed_down, void* const* packed_up, const int64_t* Cs, const int64_t* Cl,
                                int count, int nd, int dtype, void* stream);

/* Optional scratch for cvae_conv_down (for_up = 0) / cvae_conv_up (for_up = 1): layers whose output grid is too small to fill
 * the chip (8^3, 4^3 volumes) split the input-channel loop over workgroups and sum fp32 partial tiles from this buffer.
 * Returns 0 when the launch needs none.  Passing NULL / a smaller buffer is always valid: the launch then runs unsplit. */
size_t cvae_conv_data_workspace_bytes(int64_t B, int64_t sd, int64_t sh, int64_t sw, int64_t Cs,
                                      int64_t ld, int64_t lh, int64_t lw, int64_t Cl, int nd, int for_up);
/* S = act(gather(L, w) + bias) [then * (mask > 0) if mask != NULL].  bias fp32 [Cs] or NULL; mask has S's shape/dtype. */
int cvae_conv_down(const void* L, const void* w, const float* bias, const void* mask, void* S,
                   int64_t B, int64_t sd, int64_t sh, int64_t sw, int64_t Cs,
                   int64_t ld, int64_t lh, int64_t lw, int64_t Cl, int nd, int dtype, int act,
                   void* workspace, size_t workspace_bytes, void* stream);
/* L = act(scatter(S, w) + bias) [then * (mask > 0) if mask != NULL].  bias fp32 [Cl] or NULL; mask has L's shape/dtype. */
int cvae_conv_up(const void* S, const void* w, const float* bias, const void* mask, void* L,
                 int64_t B, int64_t sd, int64_t sh, int64_t sw, int64_t Cs,
                 int64_t ld, int64_t lh, int64_t lw, int64_t Cl, int nd, int dtype, int act,
                 void* workspace, size_t workspace_bytes, void* stream);
/* ReLU masks as BITS.  The backward-data launch of a layer whose input was a ReLU output zeroes its result where that activation was not positive
 * (`mask` above = the saved activation, read in full: 67 MB for enc_conv[2]'s backward at 4 x 128^3).  In the bit form the PRODUCING forward launch leaves
 * one bit per element of its result — dword i covers elements 32 i .. 32 i + 31 in memory order (a position's 32-channel block; channel counts % 32 == 0),
 * bit set <=> element > 0 — and the backward launch reads those (1/16 of the bf16 bytes).
 *   relu_bits_out (optional): numel / 32 dwords that receive the mask of THIS launch's result;  mask_bits (optional): the mask to apply (replaces `mask`).
 * Same arithmetic as cvae_conv_down / cvae_conv_up otherwise. */
int cvae_conv_down_bits(const void* L, const void* w, const float* bias, const void* mask_bits, void* S, void* relu_bits_out,
                        int64_t B, int64_t sd, int64_t sh, int64_t sw, int64_t Cs,
                        int64_t ld, int64_t lh, int64_t lw, int64_t Cl, int nd, int dtype, int act,
                        void* workspace, size_t workspace_bytes, void* stream);
int cvae_conv_up_bits(const void* S, const void* w, const float* bias, const void* mask_bits, void* L, void* relu_bits_out,
                      int64_t B, int64_t sd, int64_t sh, int64_t sw, int64_t Cs,
                      int64_t ld, int64_t lh, int64_t lw, int64_t Cl, int nd, int dtype, int act,
                      void* workspace, size_t workspace_bytes, void* stream);
/* The first conv of the encoder (Cl == 1) with the IMAGE read in the dtype it is stored in (l_dtype) while S is written in the compute
 * dtype (`dtype`): the fp32 input volume of a bf16 model feeds the kernels directly — no cast pass, no bf16 copy of the batch
 * (causal_cascade/models.py:13: nn.Conv2d(img_channels, 32, 4, 2, 1) on the fp32 batch).  cvae_conv_down_image = cvae_conv_down with
 * Cl == 1 (w = the fp32 master weight); cvae_conv_wgrad_image = cvae_conv_wgrad with Cl == 1 and the S-side bias sum.  The mixed
 * (fp32 image, bf16 S) forward reads 16-byte rows: it needs lw % 4 == 0 and a 16-byte aligned image (cvae_conv_image_supported == 1),
 * otherwise cast the image and use the plain entry points. */
int cvae_conv_image_supported(const void* L, int64_t lw, int l_dtype, int dtype);
int cvae_conv_down_image(const void* L, int l_dtype, const float* w, const float* bias, const void* mask, void* S,
                         int64_t B, int64_t sd, int64_t sh, int64_t sw, int64_t Cs, int64_t ld, int64_t lh, int64_t lw, int nd, int dtype, int act, void* stream);
int cvae_conv_wgrad_image(const void* S, const void* L, int l_dtype, float* dW, float* dbias, void* workspace, size_t workspace_bytes,
                          int64_t B, int64_t sd, int64_t sh, int64_t sw, int64_t Cs, int64_t ld, int64_t lh, int64_t lw, int nd, int dtype, void* stream);
/* ---- fp8 (OCP e4m3) products (BASELINE.json configs[4]): the batched counterfactual decode (replaces the per-value decode loop of
 * vessel_analysis/04_generate_counterfactual/generate_counterfactual.py:77-99) and the forward convolutions of the train step
 * (causal_cascade/train.py:19-39 with fp8 conv inputs; the backward pass stays bf16) ----
 * A tensor x is held as fp8 codes q with a per-tensor scale s kept by the caller: x ~ s * q.  Products accumulate in fp32 on the block-scaled
 * CDNA4 MFMA (v_mfma_scale_f32_32x32x64_f8f6f4, unit block scales): twice the bf16 FLOPs per clock.  Channel counts: C_in % 32 == 0,
 * C_out % 64 == 0 (conv) / C_out % 32 == 0, C_out > 1 (ConvTranspose).
 *   cvae_quantize_fp8           dst[i] = fp8(src[i] * inv_scale)               (src fp32 or bf16; saturating)
 *   cvae_quantize_fp8_dev       the same with 1 / s read from DEVICE memory; amax_slots (optional): records max |src| (see below)
 *   cvae_absmax                 max |src| into amax_slots (calibration of the first step)
 *   cvae_conv_pack_weight_fp8   fp32 [Cs][Cl][taps] -> operand panels [tap][C_in / 32][C_out][32] of fp8(w * inv_scale)
 *   cvae_conv_pack_weights_fp8  the same for a list of layers in one launch, 1 / s_w read from device memory, max |w| recorded
 *   cvae_conv_fp8               up = 0: S = act(conv(L_q, w_q) * acc_scale + bias); up = 1: L = act(scatter(S_q, w_q) * acc_scale + bias), acc_scale =
 *                               s_in * s_w.  out_dtype CVAE_BF16: `out` bf16, and `out8` (optional) a second copy as codes fp8(out * out8_inv_scale) for
 *                               the next fp8 layer; out_dtype CVAE_FP8: `out` holds the codes only.  dscale (optional, device): {acc_scale,
 *                               out8_inv_scale} read at run time instead of the two by-value arguments (delayed scaling under graph replay).
 *                               amax_slots (optional): records max |out|.  workspace: cvae_conv_data_workspace_bytes of the same geometry (split-K
 *                               of the small `down` grids; NULL = unsplit).  xpair: as in cvae_conv_up_variant (-1 = automatic).  relu_bits_out
 *                               (optional): the ReLU mask of `out` as bits (cvae_conv_down_bits).
 *   cvae_conv_up_fp8            = cvae_conv_fp8(up = 1) with by-value scales and no side outputs (round-2 entry point, kept)
 *   cvae_fp8_scale_update       once per step: for each of n tracked tensors, scale[i] = headroom * amax_i / 448 (amax_i = the largest value
 *                               recorded in its CVAE_AMAX_SLOTS words since the last call; the words are cleared; nothing recorded = scale kept),
 *                               inv_scale[i] = 1 / scale[i]; then for each fp8 layer l: dscale[2l] = scale[layer_in[l]] * scale[layer_w[l]],
 *                               dscale[2l + 1] = 1 / scale[layer_out[l]] (0 when layer_out[l] < 0).  layer_* are HOST arrays; `ticket` is one device
 *                               word, zero before the first call (the kernel's workgroups count themselves in on it and the last one resets it).
 * An amax record is CVAE_AMAX_SLOTS unsigned words holding float bits (non-negative floats order like unsigned integers; atomicMax, so the
 * result does not depend on the order of arrival; one word per workgroup of the producing launch, modulo the slot count); the caller zero-fills it once. */
#define CVAE_AMAX_SLOTS 4096
int cvae_quantize_fp8(const void* src, int src_dtype, void* dst, int64_t n, float inv_scale, void* stream);
int cvae_quantize_fp8_dev(const void* src, int src_dtype, void* dst, int64_t n, const float* inv_scale_dev, void* amax_slots, void* stream);
int cvae_absmax(const void* src, int dtype, int64_t n, void* amax_slots, void* stream);
int cvae_conv_pack_weight_fp8(const float* w, void* packed, int64_t Cs, int64_t Cl, int nd, int for_up, float inv_scale, void* stream);
int cvae_conv_pack_weights_fp8(const float* const* w, void* const* packed, const int64_t* Cs, const int64_t* Cl, const int* for_up,
                               const float* const* inv_scale_dev, void* const* amax_slots, int count, int nd, void* stream);
int cvae_conv_fp8(int up, const void* in8, const void* w8, const float* bias, void* out, int out_dtype, void* out8, const float* dscale, float acc_scale,
                  float out8_inv_scale, void* amax_slots, int64_t B, int64_t sd, int64_t sh, int64_t sw, int64_t Cs, int64_t ld, int64_t lh, int64_t lw, int64_t Cl,
                  int nd, int act, void* workspace, size_t workspace_bytes, int xpair, void* relu_bits_out, void* stream);
int cvae_conv_up_fp8(const void* S, const void* w, const float* bias, void* L, int out_dtype, float acc_scale, float out_inv_scale,
                     int64_t B, int64_t sd, int64_t sh, int64_t sw, int64_t Cs, int64_t ld, int64_t lh, int64_t lw, int64_t Cl, int nd, int act,
                     void* stream);
/* The decode sweep's last layer fed by the fp8 layer before it: nn.ConvTranspose3d(32, 1, 4, 2, 1) (causal_cascade/models.py:54 lifted to 3D) of S8 [B][sd][sh][sw][32] e4m3
 * codes (activation / in_scale, e.g. the codes cvae_conv_up_fp8 leaves with out_dtype CVAE_FP8), fp32 master weight w [32][1][64] and bias as they are: L [B][2sd][2sh][2sw][1] bf16 =
 * act(in_scale * (S8 (*) w) + bias).  The 32-channel tensor between the last two layers travels at one byte per element and is never widened in memory.  3D, Cs == 32 only. */
int cvae_conv_up_c1_fp8in(const void* S8, const float* w, const float* bias, void* L, float in_scale, int64_t B, int64_t sd, int64_t sh, int64_t sw, int64_t Cs, int nd, int act,
                          void* stream);
/* cvae_conv_down_image with bf16 S and the fp8 side channel of a training forward whose next conv runs on fp8 operands: S8 (optional) = fp8(S * *inv_scale_dev),
 * amax_slots (optional) records max |S|, relu_bits_out (optional) receives the ReLU mask of S as bits (cvae_conv_down_bits; this entry point also serves a
 * bf16 step that only wants the bits: S8 = NULL).  Needs the 16-byte-row form (cvae_conv_image_supported). */
int cvae_conv_down_image_f8(const void* L, int l_dtype, const float* w, const float* bias, void* S, void* S8, const float* inv_scale_dev, void* amax_slots,
                            void* relu_bits_out, int64_t B, int64_t sd, int64_t sh, int64_t sw, int64_t Cs, int64_t ld, int64_t lh, int64_t lw, int nd, int act,
                            void* stream);
int cvae_fp8_scale_update(void* amax_slots, float* scale, float* inv_scale, int n, float headroom, const int* layer_in, const int* layer_w, const int* layer_out,
                          int n_layers, float* dscale, void* ticket, void* stream);
/* cvae_conv_pack_weight_pairs for a training step with fp8 forward products: f8dir[i] = 1 / 2 writes the `down` / `up` panel of weight i as fp8 codes of
 * w * *inv_scale_dev[i] into f8out[i] (layout of cvae_conv_pack_weight_fp8) INSTEAD of its bf16 panel (packed_down[i] / packed_up[i] may then be NULL), and
 * records max |w| in amax_slots[i] (optional); f8dir[i] = 0 (or f8dir == NULL): both bf16 / fp32 panels as cvae_conv_pack_weight_pairs. */
int cvae_conv_pack_weight_pairs_f8(const float* const* w, void* const* packed_down, void* const* packed_up, const int64_t* Cs, const int64_t* Cl,
                                   const int* f8dir, void* const* f8out, const float* const* inv_scale_dev, void* const* amax_slots,
                                   int count, int nd, int dtype, void* stream);
/* ---- Exact-2x linear resize (decoder output d x h x w, one channel -> 2d x 2h x 2w; D == d == 1 for 2D) fused with the ELBO ----
 * causal_cascade/models.py:84-87 + train.py:5-17: the resized volume is recomputed from the small tensor wherever it is needed
 * instead of being written and re-read (csrc/recon_loss.hip).  cvae_up2x_supported: 1 when the shapes qualify (w % 4 == 0). */
int cvae_up2x_supported(int64_t B, int64_t d, int64_t h, int64_t w, int64_t D, int64_t H, int64_t W);
/* dst fp32 [B][D][H][W] = F.interpolate(src, (D, H, W), 'trilinear' | 'bilinear', align_corners=False) */
int cvae_up2x_fwd(const void* src, float* dst, int64_t B, int64_t d, int64_t h, int64_t w, int64_t D, int64_t H, int64_t W, int dtype, void* stream);
/* out4 = {loss, recon, m_loss, kld}: recon = sum (up(src) - x)^2, m_loss = sum (m_hat - m)^2 (n_m elements),
 * kld = -0.5 sum (1 + logvar - mu^2 - exp(logvar)) (n_z elements), loss = recon + gamma * m_loss + kld.
 * partial: scratch of cvae_elbo_up2x_partials(B, d, h, w) floats (per-workgroup sums, added in a fixed order: no atomics). */
#define CVAE_ELBO_TICKET_WORDS (33 * 32)
int64_t cvae_elbo_up2x_partials(int64_t B, int64_t d, int64_t h, int64_t w);
int cvae_elbo_up2x_fwd(const void* src, const float* x, const float* m_hat, const float* m, const float* mu, const float* logvar, float gamma,
                       float* out4, float* partial, float* t1, void* ticket, int* bump, int64_t B, int64_t d, int64_t h, int64_t w, int64_t D, int64_t H, int64_t W,
                       int64_t n_m, int64_t n_z, int dtype, void* stream);
/* ticket (optional): CVAE_ELBO_TICKET_WORDS device words, zero before the first call — the launch then finishes the sums itself (the workgroup that arrives
 * last, found by a two-level arrival count, adds the partials in index order: same bits as the two-launch form, one dependent launch fewer; the words are
 * zero again when the launch ends).  bump (optional,
 * needs ticket): a device int incremented once by that last workgroup — the optimizer's device step counter rides along (cvae_adam_multi's step_dev). */
/* t1 (B*D*H*w floats, or NULL when no backward follows): the same launch also leaves U_w^T (up(src) - x), the part of the backward that needs x,
 * so that a training step reads x once.  cvae_elbo_up2x_bwd turns it into the gradients of `loss` scaled by the device scalar *g_loss (NULL = 1):
 * dsrc (dtype), d_mhat, dmu, dlv. */
int cvae_elbo_up2x_bwd(const float* t1, const float* m_hat, const float* m, const float* mu, const float* logvar, float gamma, const float* g_loss,
                       void* dsrc, float* d_mhat, float* dmu, float* dlv, int64_t B, int64_t d, int64_t h, int64_t w, int64_t D, int64_t H, int64_t W,
                       int64_t n_m, int64_t n_z, int dtype, void* stream);

/* ---- CausalVesselVAE extras (vessel_analysis/00_core/models.py:9-166): BatchNorm2d, clamp, Upsample(nearest x2) + Conv2d(k3, s1, p1) ----
 * nn.Upsample(scale_factor=2, 'nearest') followed by nn.Conv2d(Cin, Cout, 3, 1, 1) equals the transposed k4/s2/p1 product of
 * cvae_conv_up with K4[cin][cout] = A W3[cout][cin] A^T, A = [[0,0,1],[0,1,1],[1,1,0],[1,0,0]] (csrc/vessel2d.hip): the layer runs on
 * cvae_conv_up / cvae_conv_down / cvae_conv_wgrad with the transformed weight; its gradient maps back with A^T . A. */
int cvae_conv3_to_k4(const float* w3 /* [Cout][Cin][3][3] */, float* k4 /* [Cin][Cout][4][4] */, int64_t Cout, int64_t Cin, void* stream);
int cvae_k4_to_conv3_grad(const float* dk4, float* dw3, int64_t Cout, int64_t Cin, void* stream);
/* torch.clamp(x, lo, hi) and its backward (the gradient passes where lo <= x <= hi); fp32 */
int cvae_clamp_fwd(const float* x, float* y, float lo, float hi, int64_t n, void* stream);
int cvae_clamp_bwd(const float* x, const float* g, float* dx, float lo, float hi, int64_t n, void* stream);
/* nn.BatchNorm2d (+ fused activation `act`) on a channels-last tensor [P = B*H*W][C] (C % 8 == 0).  training: two-pass batch
 * statistics (mean, then sum of squared deviations) into mean / rstd, running_* updated with the unbiased variance (may be NULL);
 * otherwise mean / rstd are inputs (running mean, 1/sqrt(running_var + eps)).  y = act((x - mean) rstd gamma + beta). */
/* workspace: cvae_bn2d_workspace_bytes(P, C) — per-workgroup partial rows of the channel statistics, added in index order by the
 * finalize launches (no float atomics; required in training mode and by the backward). */
size_t cvae_bn2d_workspace_bytes(int64_t P, int64_t C);
int cvae_bn2d_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, float* running_mean, float* running_var,
                  int64_t P, int64_t C, float momentum, float eps, int training, int act, int dtype, void* workspace, size_t workspace_bytes, void* stream);
/* training-mode backward through the activation and the normalisation: dx, dgamma, dbeta (y = the forward's output, for act') */
int cvae_bn2d_bwd(const void* x, const void* dy, const void* y, const float* gamma, const float* mean, const float* rstd, void* dx, float* dgamma,
                  float* dbeta, int64_t P, int64_t C, int act, int dtype, void* workspace, size_t workspace_bytes, void* stream);

/* ---- The dense bottleneck of CausalBioVAE in 5 + 5 launches (batch M <= 16, fp32 arithmetic) --------------------------------
 * Replaces, between the last encoder conv and the first decoder conv (causal_cascade/models.py:57-79):
 *   AdaptiveAvgPool + Flatten, cat([x_feat, m, t]), enc_fc (Linear-ReLU-Linear-ReLU), fc_mu, fc_logvar, reparameterize,
 *   mechanism_net (Linear, BatchNorm1d, ReLU, Linear, ReLU, Linear), cat([z, m_hat]), dec_input (+ the NC(D)HW -> channels-last
 *   move into the decoder).  All pointers are device pointers; weights are the nn.Linear / BatchNorm1d tensors as they are.
 * y_cl: last encoder activation [M][D][H][W][C] (conv dtype), pooled to OD x OH x OW windows (D % OD == 0 etc.).
 * N1 / N2: widths of enc_fc.0 / enc_fc.2; Z: latent; HM: mechanism_net width; K1 = C*OD*OH*OW + m_dim + t_dim; K4 = Z + m_dim. */
typedef struct { int64_t M, D, H, W, C, OD, OH, OW, m_dim, t_dim, N1, N2, Z, HM; } cvae_bottleneck_dims;
typedef struct {                 /* parameters (fp32, torch layouts [out][in]) */
    const float *W1, *b1, *W2, *b2, *Wmu, *bmu, *Wlv, *blv, *Wm0, *bm0, *gamma, *beta, *Wm3, *bm3, *Wm5, *bm5, *Wd, *bd;
} cvae_bottleneck_params;
typedef struct {                 /* their gradients (overwritten), same order and shapes */
    float *dW1, *db1, *dW2, *db2, *dWmu, *dbmu, *dWlv, *dblv, *dWm0, *dbm0, *dgamma, *dbeta, *dWm3, *dbm3, *dWm5, *dbm5, *dWd, *dbd;
} cvae_bottleneck_grads;
typedef struct {                 /* activations kept for backward, all [M][dim] fp32: h1 [N1], h2 [N2], mu / logvar [Z],
                                    xhat / a1n / a2 [HM], invstd [HM] (one row), m_hat [m_dim], zm [K4] = cat(z, m_hat) */
    float *h1, *h2, *mu, *logvar, *xhat, *invstd, *a1n, *a2, *m_hat, *zm;
} cvae_bottleneck_saved;
/* Scratch sizes (in floats) for the given dims: xcat needs M*K1; the three partial buffers as returned. */
int cvae_bottleneck_sizes(const cvae_bottleneck_dims* dims, int64_t* K1, int64_t* K4, int64_t* fwd_partial_floats, int64_t* dzm_partial_floats,
                          int64_t* dx_partial_floats);
/* Forward.  t_onehot [M][t_dim] (F.one_hot(t).float(): an INPUT when t_labels is NULL; with t_labels [M] (int64 class indices) the call
 * writes it itself, saving the caller's one-hot launch), eps [M][Z] (the reparameterisation noise).  bn_training: batch statistics + running-stat /
 * num_batches_tracked update (running_* may be NULL), else running statistics.  Outputs: saved->mu / logvar / m_hat (the
 * model's outputs) and dec_cl [M][OD][OH][OW][C] (conv dtype) = dec_input(cat(z, m_hat)) viewed [M, C, 4..] channels-last.
 * dzm_acc: scratch of dzm_partial_floats (cvae_bottleneck_sizes) the BACKWARD fills with per-workgroup partials of d(zm) (one slot per
 * workgroup, plain stores, summed in index order: no float atomics); the forward does not touch it (kept in the signature so one
 * buffer can serve the pair). */
int cvae_bottleneck_fwd(const cvae_bottleneck_dims* dims, const cvae_bottleneck_params* params, const void* y_cl, const float* m, float* t_onehot,
                        const int64_t* t_labels, const float* eps, float* running_mean, float* running_var, long long* num_batches_tracked, float momentum, float bn_eps,
                        int bn_training, float* xcat, float* fwd_partial, float* dzm_acc, const cvae_bottleneck_saved* saved, void* dec_cl, int dtype,
                        void* stream);
/* Backward (training-mode BatchNorm only).  g_dec_cl: gradient of dec_cl; g_mu / g_logvar / g_mhat: gradients arriving at the
 * three outputs (NULL = zero).  Writes every parameter gradient and dy_cl, the gradient of y_cl (zeroed where y_cl <= 0 when
 * relu_mask).  g1 is scratch of M*(N1+N2) floats; dzm_partial: dzm_partial_floats of scratch (any contents on entry). */
int cvae_bottleneck_bwd(const cvae_bottleneck_dims* dims, const cvae_bottleneck_params* params, const cvae_bottleneck_grads* grads,
                        const cvae_bottleneck_saved* saved, const void* g_dec_cl, const float* g_mu, const float* g_logvar, const float* g_mhat,
                        const float* t_onehot, const float* eps, const float* xcat, const void* y_cl, int relu_mask, float* dzm_partial, float* g1,
                        float* dx_partial, void* dy_cl, int dtype, void* stream);

/* ---- The same pair with mechanism_net's BatchNorm1d normalising over the GLOBAL batch of `bn_ranks` data-parallel ranks (SyncBatchNorm; replaces what
 * nn.SyncBatchNorm.convert_sync_batchnorm does to causal_cascade/models.py:36 under DDP).  The library runs no collective itself; the caller moves
 * 2 * HM floats per rank twice per step:
 *   1. cvae_bottleneck_bn_local_stats: local_stats [2][HM] = this rank's (sum, squared deviations from its own mean) of mechanism_net.0's output
 *      (it depends on t only, so it can run — and the all-gather can travel — before the encoder).  Wm0 [HM][t_dim], bm0 [HM]; t_onehot [M][t_dim] or t_labels [M]
 *      (either may be NULL);
 *   2. all-gather -> bn_rank_stats [bn_ranks][2][HM]; cvae_bottleneck_fwd_sync combines them in rank order (Chan's update: identical bits on every rank),
 *      normalises with the global mean / variance and updates the running statistics with the global unbiased variance.  bn_rank_stats NULL: cvae_bottleneck_fwd;
 *   3. cvae_bottleneck_bwd_sync: as cvae_bottleneck_bwd, but mechanism_net's backward stops at the BatchNorm: bn_dy [M][HM] = the masked gradient at its output,
 *      bn_local_sums [2][HM] = this rank's (sum dy, sum dy * xhat) (also written to dbeta / dgamma: the rank's share, summed by the gradient exchange);
 *      both NULL: cvae_bottleneck_bwd;
 *   4. all-reduce (sum) of bn_local_sums -> bn_sums; cvae_bottleneck_bn_bwd_finish writes dWm0 / dbm0 from bn_dy and the global sums (one small launch). */
int cvae_bottleneck_bn_local_stats(const float* Wm0, const float* bm0, const float* t_onehot, const int64_t* t_labels, float* local_stats, int64_t M, int64_t t_dim,
                                   int64_t HM, void* stream);
int cvae_bottleneck_fwd_sync(const cvae_bottleneck_dims* dims, const cvae_bottleneck_params* params, const void* y_cl, const float* m, float* t_onehot,
                             const int64_t* t_labels, const float* eps, float* running_mean, float* running_var, long long* num_batches_tracked, float momentum,
                             float bn_eps, int bn_training, float* xcat, float* fwd_partial, float* dzm_acc, const cvae_bottleneck_saved* saved, void* dec_cl, int dtype,
                             const float* bn_rank_stats, int bn_ranks, void* stream);
/* The general forward: as cvae_bottleneck_fwd_sync, plus (noise != NULL) the reparameterisation noise drawn by the first launch: eps [M][Z] becomes an OUTPUT
 * holding exactly the numbers cvae_philox_normal_advance(eps, M * Z, seed, 0, subsequence, call_counter) would have written, and *call_counter is incremented
 * the same way (one launch fewer per step; the reference draws torch.randn_like in reparameterize, causal_cascade/models.py:65-68). */
typedef struct { uint64_t seed, subsequence; int* call_counter; } cvae_bottleneck_noise;
int cvae_bottleneck_fwd_ex(const cvae_bottleneck_dims* dims, const cvae_bottleneck_params* params, const void* y_cl, const float* m, float* t_onehot,
                           const int64_t* t_labels, float* eps, float* running_mean, float* running_var, long long* num_batches_tracked, float momentum,
                           float bn_eps, int bn_training, float* xcat, float* fwd_partial, float* dzm_acc, const cvae_bottleneck_saved* saved, void* dec_cl, int dtype,
                           const float* bn_rank_stats, int bn_ranks, const cvae_bottleneck_noise* noise, void* stream);
int cvae_bottleneck_bwd_sync(const cvae_bottleneck_dims* dims, const cvae_bottleneck_params* params, const cvae_bottleneck_grads* grads,
                             const cvae_bottleneck_saved* saved, const void* g_dec_cl, const float* g_mu, const float* g_logvar, const float* g_mhat,
                             const float* t_onehot, const float* eps, const float* xcat, const void* y_cl, int relu_mask, float* dzm_partial, float* g1,
                             float* dx_partial, void* dy_cl, int dtype, float* bn_dy, float* bn_local_sums, void* stream);
int cvae_bottleneck_bn_bwd_finish(const cvae_bottleneck_dims* dims, const cvae_bottleneck_params* params, const cvae_bottleneck_grads* grads,
                                  const cvae_bottleneck_saved* saved, const float* t_onehot, const float* bn_dy, const float* bn_sums, int bn_ranks, void* stream);

/* dW fp32 [Cs][Cl][taps] (overwritten) = sum over batch and positions.  workspace: cvae_conv_wgrad_workspace_bytes().
 * dbias (optional, overwritten) rides along in the same pass, where both tensors are read anyway:
 *   dbias_side 0: fp32 [Cs] = sum over batch and positions of S — the bias gradient of a Conv layer (S = output gradient);
 *   dbias_side 1: fp32 [Cl] = the same sum of L — the bias gradient of a ConvTranspose layer (L = output gradient). */
size_t cvae_conv_wgrad_workspace_bytes(int64_t Cs, int64_t Cl, int nd);
int cvae_conv_wgrad(const void* S, const void* L, float* dW, float* dbias, int dbias_side, void* workspace, size_t workspace_bytes,
                    int64_t B, int64_t sd, int64_t sh, int64_t sw, int64_t Cs,
                    int64_t ld, int64_t lh, int64_t lw, int64_t Cl, int nd, int dtype, void* stream);
/* The weight gradients of `count` (<= 8) layers in ONE main launch and ONE reduce launch: what a backward pass defers to its end, so
 * that the small layers (a few tiles each, launch- and latency-bound alone) run in the shadow of the large ones.  Arrays of `count`
 * entries; dims: count rows of 9 int64 {B, sd, sh, sw, Cs, ld, lh, lw, Cl}; every layer needs its OWN workspace
 * (cvae_conv_wgrad_workspace_bytes) because they run concurrently.  All layers share nd and dtype; Cl != 1, Cs % 64 == 0,
 * Cl % 32 == 0; an entry with dbias_side 1 needs L == 2 S (otherwise use cvae_conv_wgrad for that layer).  The partial sums of a layer
 * are split over fewer workgroups than in its own launch (the layers share the chip), so the results equal those of `count` calls of
 * cvae_conv_wgrad up to the fp32 summation order; they are reproducible bit for bit from run to run (no atomics). */
int cvae_conv_wgrad_multi(int count, const void* const* S, const void* const* L, float* const* dW, float* const* dbias, const int* dbias_side,
                          void* const* workspace, const size_t* workspace_bytes, const int64_t* dims, int nd, int dtype, void* stream);
/* out[c] = sum_p x[p, c] over a channels-last [P, C] tensor (bias gradients). out is overwritten.  workspace (optional,
 * cvae_channel_sum_workspace_bytes): partial rows of a multi-workgroup pass, added in index order (no float atomics); without it
 * the rows of one channel group are summed by ONE workgroup (same result bits for a given geometry, slower for large P). */
size_t cvae_channel_sum_workspace_bytes(int64_t P, int64_t C, int dtype);
int cvae_channel_sum(const void* x, float* out, int64_t P, int64_t C, int dtype, void* workspace, size_t workspace_bytes, void* stream);
/* y = act(x) elementwise (nn.ReLU / nn.Sigmoid / nn.LeakyReLU(0.2) when not fused into a producer). */
int cvae_act_fwd(const void* x, void* y, int64_t n, int act, int dtype, void* stream);
/* dx = dy * act'(y) elementwise (y = saved activation OUTPUT); dtype applies to all three. */
int cvae_act_bwd(const void* dy, const void* y, void* dx, int64_t n, int act, int dtype, void* stream);

/* ---- pooling / resize ----------------------------------------------------------------------------- */
/* x channels-last [B, D, H, W, C] -> out fp32 [B, out_stride], columns (c, od, oh, ow) flattened like
 * nn.AdaptiveAvgPool(d)+Flatten on NC(D)HW; adaptive windows [floor(i*I/O), ceil((i+1)*I/O)). */
int cvae_adaptive_avgpool_fwd(const void* x, float* out, int64_t B, int64_t D, int64_t H, int64_t W, int64_t C,
                              int64_t OD, int64_t OH, int64_t OW, int64_t out_stride, int dtype, void* stream);
/* dx[b, d, h, w, c] = sum over windows containing (d, h, w) of dout / |window|, times (x > 0) when relu_mask_x != NULL. */
int cvae_adaptive_avgpool_bwd(const float* dout, const void* relu_mask_x, void* dx, int64_t B, int64_t D, int64_t H, int64_t W, int64_t C,
                              int64_t OD, int64_t OH, int64_t OW, int64_t dout_stride, int dtype, void* stream);
/* (bi|tri)linear resize, align_corners = False, channels-last C channels; src dtype `dtype`, dst fp32. */
int cvae_upsample_linear_fwd(const void* src, float* dst, int64_t B, int64_t d, int64_t h, int64_t w,
                             int64_t D, int64_t H, int64_t W, int64_t C, int dtype, void* stream);
/* transpose of the above: dsrc (dtype `dtype`) = A^T ddst (fp32). */
int cvae_upsample_linear_bwd(const float* ddst, void* dsrc, int64_t B, int64_t d, int64_t h, int64_t w,
                             int64_t D, int64_t H, int64_t W, int64_t C, int dtype, void* stream);

/* ---- fp32 linear layers ------------------------------------------------------------------------------ */
/* Long reductions over few output tiles (the 16415-wide encoder layer at batch 4) are split across workgroups; each split leaves its
 * partial tile in a slab of `workspace` (cvae_linear_workspace_bytes(M, K, N, op); op 0 = fwd, 1 = bwd_data, 2 = bwd_weight) and a finish
 * launch adds the slabs in index order — no float atomics.  workspace may be NULL / smaller: the product then runs unsplit. */
size_t cvae_linear_workspace_bytes(int64_t M, int64_t K, int64_t N, int op);
/* y[M, N] = act(x[M, K] @ W[N, K]^T + b). */
int cvae_linear_fwd(const float* x, const float* W, const float* b, float* y, int64_t M, int64_t K, int64_t N,
                    int64_t x_stride, int64_t y_stride, int act, void* workspace, size_t workspace_bytes, void* stream);
/* dx[M, K] = g[M, N] @ W[N, K] with g = dy * act'(y_act) (y_act = the layer's saved OUTPUT, same shape/stride as dy; pass
 * NULL / CVAE_ACT_NONE for g = dy).  The fused activation gradient is available for M <= 16 (the model's batch sizes). */
int cvae_linear_bwd_data(const float* dy, const float* W, float* dx, int64_t M, int64_t K, int64_t N,
                         int64_t dy_stride, int64_t dx_stride, const float* y_act, int act, void* workspace, size_t workspace_bytes, void* stream);
/* dW[N, K] = g^T x ; db[N] = column sums of g (db may be NULL).  Both overwritten.  g as above. */
int cvae_linear_bwd_weight(const float* dy, const float* x, float* dW, float* db, int64_t M, int64_t K, int64_t N,
                           int64_t dy_stride, int64_t x_stride, const float* y_act, int act, void* workspace, size_t workspace_bytes, void* stream);

/* ---- BatchNorm1d --------------------------------------------------------------------------------------- */
/* train: batch statistics (biased var) normalise; running stats updated with momentum (unbiased var). B >= 2. */
int cvae_bn1d_train_fwd(const float* x, const float* w, const float* b, float* y, float* save_mean, float* save_rstd,
                        float* running_mean, float* running_var, int64_t B, int64_t F, float momentum, float eps, void* stream);
int cvae_bn1d_train_bwd(const float* dy, const float* x, const float* w, const float* save_mean, const float* save_rstd,
                        float* dx, float* dw, float* db, int64_t B, int64_t F, void* stream);
int cvae_bn1d_eval_fwd(const float* x, const float* w, const float* b, const float* running_mean, const float* running_var,
                       float* y, int64_t B, int64_t F, float eps, void* stream);
/* The same BatchNorm1d cut where its batch sums appear, for statistics that span data-parallel ranks (SyncBN; the host all-reduces
 * the F-float sum vectors between the calls: a per-rank batch of 4 then normalises like the reference's single-process batch of 32,
 * causal_cascade/models.py:36 at the global batch):
 *   cvae_bn1d_stats        out[f] = sum_b x[b][f] (mean_sum NULL), else sum_b (x[b][f] - mean_sum[f] * inv_n)^2
 *   cvae_bn1d_apply_stats  y, save_mean, save_rstd, running stats from the reduced sums; inv_n = 1 / global batch
 *   cvae_bn1d_bwd_sums     sums[0:F] = sum_b dy (this rank's d beta), sums[F:2F] = sum_b dy * xhat (this rank's d gamma)
 *   cvae_bn1d_bwd_apply    dx from the REDUCED sums */
int cvae_bn1d_stats(const float* x, const float* mean_sum, float inv_n, float* out, int64_t B, int64_t F, void* stream);
int cvae_bn1d_apply_stats(const float* x, const float* w, const float* b, const float* mean_sum, const float* sqdev_sum, float inv_n, float* y,
                          float* save_mean, float* save_rstd, float* running_mean, float* running_var, int64_t B, int64_t F, float momentum,
                          float eps, void* stream);
int cvae_bn1d_bwd_sums(const float* dy, const float* x, const float* save_mean, const float* save_rstd, float* sums, int64_t B, int64_t F, void* stream);
int cvae_bn1d_bwd_apply(const float* dy, const float* x, const float* w, const float* save_mean, const float* save_rstd, const float* sums, float inv_n,
                        float* dx, int64_t B, int64_t F, void* stream);

/* ---- sampling + losses (all reductions accumulate into a caller-zeroed fp32 scalar) -------------------------
 * Every reduction takes (workspace, workspace_bytes): cvae_reduce_workspace_bytes() of scratch for the per-workgroup partial sums, which a
 * one-block finish launch adds in index order — no float atomics, the sums are bit-reproducible.  NULL / too small: the reduction runs
 * as a single workgroup (same contract, slower for large n). */
size_t cvae_reduce_workspace_bytes(void);
/* eps ~ N(0,1): Philox4x32-10 + Box-Muller, counter-based (seed = key, offset = counter words 0-1, subsequence = counter words 2-3)
 * -> reproducible per launch; different subsequences are independent streams of one seed (the Python side passes
 * (rank << 32) | model instance, so data-parallel ranks that share torch.manual_seed(42) still draw different noise).  call_counter
 * (optional device int): offset += *call_counter << 24, so a captured HIP graph draws fresh numbers on every replay. */
int cvae_philox_normal(float* out, int64_t n, uint64_t seed, uint64_t offset, uint64_t subsequence, const int* call_counter, void* stream);
/* The same draw followed by *call_counter += 1 (one launch for n <= 16384): what EpsSource issues once per forward pass. */
int cvae_philox_normal_advance(float* out, int64_t n, uint64_t seed, uint64_t offset, uint64_t subsequence, int* call_counter, void* stream);
/* z = mu + eps*exp(logvar/2) (if z != NULL);  *kld += -0.5*sum(1 + logvar - mu^2 - exp(logvar)) (if kld != NULL). */
int cvae_reparam_kld_fwd(const float* mu, const float* logvar, const float* eps, float* z, float* kld, int64_t n, void* workspace, size_t workspace_bytes, void* stream);
/* dmu = dz + gk*mu ; dlogvar = dz*eps*0.5*exp(logvar/2) + gk*0.5*(exp(logvar) - 1), gk = *gkld * gk_scale;
 * dz / gkld (device scalar) may be NULL. */
int cvae_reparam_kld_bwd(const float* dz, const float* gkld, float gk_scale, const float* mu, const float* logvar, const float* eps,
                         float* dmu, float* dlogvar, int64_t n, void* stream);
/* The same head on the encoder's [B][2 Z] output rows (mu | logvar) without splitting them: z1 = mu + eps1 exp(logvar / 2), an optional second sample z2
 * from eps2, and *kld = the KLD sum (assigned; any of z1 / z2 / kld may be NULL).  One launch; B * Z <= 2^24.  The backward writes d h [B][2 Z] from
 * d z1, d z2 (either may be NULL) and the KLD's incoming gradient *gkld (NULL: 0). */
int cvae_latent_head_fwd(const float* h, const float* eps1, const float* eps2, float* z1, float* z2, float* kld, int64_t B, int64_t Z, void* stream);
int cvae_latent_head_bwd(const float* dz1, const float* dz2, const float* gkld, const float* h, const float* eps1, const float* eps2, float* dh,
                         int64_t B, int64_t Z, void* stream);
/* *out += sum (a - b)^2 */
int cvae_sse_fwd(const float* a, const float* b, float* out, int64_t n, void* workspace, size_t workspace_bytes, void* stream);
/* da = 2*(a - b) * (*gout) * scale   (db = -da is formed by the caller when needed) */
int cvae_sse_bwd(const float* a, const float* b, const float* gout, float scale, float* da, int64_t n, void* stream);
/* out4[0] = out4[1] + wb*out4[2] + wc*out4[3]: total loss from its three terms, on the device */
int cvae_combine3(float* out4, float wb, float wc, void* stream);
/* F.binary_cross_entropy(p, x, reduction='sum') with torch's log clamp at -100 */
int cvae_bce_fwd(const float* p, const float* x, float* out, int64_t n, void* workspace, size_t workspace_bytes, void* stream);
int cvae_bce_bwd(const float* p, const float* x, const float* gout, float* dp, int64_t n, void* stream);
/* vessel recon terms (vessel_analysis/01_train/train.py:27-46): stats[0] = sum(x) must be produced first by
 * cvae_sum_fwd; then out[0] += sum (r-x)^2 (1 + (pw-1) x), out[1] += sum |r| [x < 0.1], pw = clamp((1-pf)/(pf+1e-6), 1, 50),
 * pf = *sum_x / (n_pos + 1e-6).  n_pos = the element count sum_x was taken over: n for a single process; under data parallelism the
 * caller all-reduces sum_x and passes the GLOBAL count, which reproduces the reference's batch-global pos_weight (:30-36). */
int cvae_sum_fwd(const float* x, float* out, int64_t n, void* workspace, size_t workspace_bytes, void* stream);
int cvae_wmse_sparsity_fwd(const float* r, const float* x, const float* sum_x, float* out2, int64_t n, int64_t n_pos, void* workspace, size_t workspace_bytes, void* stream);
int cvae_wmse_sparsity_bwd(const float* r, const float* x, const float* sum_x, const float* g_recon, const float* g_sparsity,
                           float* dr, int64_t n, int64_t n_pos, void* stream);
/* *out += 0.5 * sum(logvar + (m - mu)^2 / exp(logvar)) */
int cvae_gauss_nll_fwd(const float* m, const float* mu, const float* logvar, float* out, int64_t n, void* workspace, size_t workspace_bytes, void* stream);
int cvae_gauss_nll_bwd(const float* m, const float* mu, const float* logvar, const float* gout, float* dmu, float* dlogvar, int64_t n, void* stream);
/* *out += mean_b CE(logits[b, :], target[b])  (F.cross_entropy, reduction='mean'); dlogits = (softmax - onehot)/B * gout */
int cvae_softmax_ce_fwd(const float* logits, const int64_t* target, float* out, int64_t B, int64_t C, void* workspace, size_t workspace_bytes, void* stream);
int cvae_softmax_ce_bwd(const float* logits, const int64_t* target, const float* gout, float* dlogits, int64_t B, int64_t C, void* stream);
/* *out += F.kl_div(log_softmax(logits), full(1/C), reduction='batchmean'); dlogits = (softmax - 1/C)/B * gout */
int cvae_uniform_kl_fwd(const float* logits, float* out, int64_t B, int64_t C, void* workspace, size_t workspace_bytes, void* stream);
int cvae_uniform_kl_bwd(const float* logits, const float* gout, float* dlogits, int64_t B, int64_t C, void* stream);

/* The same three products with bf16 MFMA operands (fp32 tensors in memory, rounded to bf16 on the way into LDS, fp32 accumulate and output):
 * 16x the matrix rate of the exact-fp32 form, relative error ~2^-9 per operand.  M > 16 only (smaller batches are HBM-bound skinny kernels:
 * CVAE_E_UNSUPPORTED); no fused activation gradient; workspace as for the fp32 entry points (cvae_linear_workspace_bytes). */
int cvae_linear_fwd_bf16(const float* x, const float* W, const float* b, float* y, int64_t M, int64_t K, int64_t N, int64_t x_stride, int64_t y_stride,
                         int act, void* workspace, size_t workspace_bytes, void* stream);
int cvae_linear_bwd_data_bf16(const float* dy, const float* W, float* dx, int64_t M, int64_t K, int64_t N, int64_t dy_stride, int64_t dx_stride,
                              void* workspace, size_t workspace_bytes, void* stream);
/* ---- nn.Linear (+ activation) for SMALL layers at LARGE batch (csrc/small_dense.hip): M > 16 rows, K, N <= 512 and N * K <= 12288 weights — the MLP heads of
 * mnist_test/01_baseline_causal_vae/models.py:24-37, 93-111 at batch 1024.  The weight lives in LDS; one launch forward, one for the data gradient, two for the
 * weight + bias gradient (per-workgroup partials in `workspace`, summed in index order: no atomics).  fp32 throughout.
 *   fwd         y = act(x W^T + b)
 *   bwd_data    dx = ((dy * act'(y_act)) W) * in_act'(x_in)     y_act / x_in may be NULL (ACT_NONE): both activation gradients ride in this launch
 *   bwd_weight  dW = (dy * act'(y_act))^T x,  db = column sums of (dy * act'(y_act))  (db may be NULL) */
int cvae_small_dense_supported(int64_t M, int64_t K, int64_t N);
size_t cvae_small_dense_workspace_bytes(int64_t M, int64_t K, int64_t N);
int cvae_small_dense_fwd(const float* x, const float* W, const float* b, float* y, int64_t M, int64_t K, int64_t N, int64_t x_stride, int64_t y_stride, int act, void* stream);
int cvae_small_dense_bwd_data(const float* dy, const float* W, float* dx, const float* y_act, int act, const float* x_in, int in_act, int64_t M, int64_t K, int64_t N,
                              int64_t dy_stride, int64_t dx_stride, int64_t y_stride, int64_t x_stride, void* stream);
int cvae_small_dense_bwd_weight(const float* dy, const float* x, float* dW, float* db, const float* y_act, int act, int64_t M, int64_t K, int64_t N, int64_t dy_stride,
                                int64_t x_stride, int64_t y_stride, void* workspace, size_t workspace_bytes, void* stream);

/* dx = (dy . W) * act'(x_in): cvae_linear_bwd_data(_bf16) with the derivative of the activation that produced this layer's INPUT x_in [M][K] (taken from its output,
 * i.e. from x_in: ReLU / LeakyReLU / Sigmoid) applied in the GEMM's epilogue (or its split-K slab sum) — in an MLP the previous layer then needs no activation-gradient
 * pass of its own (causal_cascade/models.py:24-31's Linear-ReLU-Linear chains; one launch fewer per layer at batch sizes above 16).  bf16_math as cvae_linear_*_bf16. */
int cvae_linear_bwd_data_inact(const float* dy, const float* W, float* dx, int64_t M, int64_t K, int64_t N, int64_t dy_stride, int64_t dx_stride,
                               const float* x_in, int64_t x_stride, int in_act, int bf16_math, void* workspace, size_t workspace_bytes, void* stream);
int cvae_linear_bwd_weight_bf16(const float* dy, const float* x, float* dW, float* db, int64_t M, int64_t K, int64_t N, int64_t dy_stride,
                                int64_t x_stride, void* workspace, size_t workspace_bytes, void* stream);

/* cvae_conv_down with the two-samples-per-tile form (layers at most half a tile wide: 2D grids up to 8 wide — the 7 x 7 maps of the MNIST model —, used
 * automatically from 512 workgroups up) forced off (0) / on (1) for this call, -1 = automatic.  Same products in the same order: bit-identical results. */
int cvae_conv_down_variant(const void* L, const void* w, const float* bias, const void* mask, void* S,
                           int64_t B, int64_t sd, int64_t sh, int64_t sw, int64_t Cs,
                           int64_t ld, int64_t lh, int64_t lw, int64_t Cl, int nd, int dtype, int act,
                           void* workspace, size_t workspace_bytes, int xpair, void* stream);
/* cvae_conv_up with the kernel form chosen by the CALLER for this one call instead of by launch size (the library keeps no process-wide tuning state):
 *   upfull        -1 automatic; 0 / 1: never / whenever it fits — the whole-K `up` kernel (bf16, 64 / 128 / 256 input channels, 32 output channels);
 *   xpair         -1 automatic; 0 / 1: one sample per tile / two side by side — 3D layers at most 4 source voxels wide (the decoder's 4^3 input), 2D layers
 *                 at most 8 wide (automatic from 2048 / 512 workgroups up);
 *   c1_walk_units  0 automatic; > 0: the single-channel output layer (bf16, 3D) walks z columns when the launch has at least 2 x this many tiles.
 * Every form computes the same products in the same order (bit-identical results; the tests run each narrow case through both). */
int cvae_conv_up_variant(const void* S, const void* w, const float* bias, const void* mask, void* L,
                         int64_t B, int64_t sd, int64_t sh, int64_t sw, int64_t Cs,
                         int64_t ld, int64_t lh, int64_t lw, int64_t Cl, int nd, int dtype, int act,
                         void* workspace, size_t workspace_bytes, int upfull, int xpair, int64_t c1_walk_units, void* stream);

/* ---- optimiser ---------------------------------------------------------------------------------------------- */
/* torch.optim.Adam (no weight decay / amsgrad) on flat fp32 buffers; bias corrections bc1 = 1-b1^t, bc2 = 1-b2^t
 * computed by the caller.  grad_scale: optional device scalar multiplied into g first (gradient clipping). */
int cvae_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                   float bc1, float bc2, const float* grad_scale, void* stream);
/* The same update for a LIST of tensors in one launch (host arrays of `count` device pointers / sizes).  step_dev: optional
 * device int holding the step number t; when non-NULL the kernel derives bc1/bc2 from it (HIP-graph replay safe). */
int cvae_adam_multi(float* const* p, const float* const* g, float* const* m, float* const* v, const int64_t* n, int count,
                    float lr, float beta1, float beta2, float eps, float bc1, float bc2, const int* step_dev,
                    const float* grad_scale, void* stream);
/* dst[i][0..n[i]) = src[i][0..n[i]) for a LIST of fp32 tensors in one launch (gradient bucket pack / unpack). */
int cvae_multi_copy(const float* const* src, float* const* dst, const int64_t* n, int count, void* stream);
/* *counter += delta (device int; step counters that must advance inside a captured graph) */
int cvae_counter_add(int* counter, int delta, void* stream);
/* *out += sum g^2 */
int cvae_sqnorm(const float* g, float* out, int64_t n, void* workspace, size_t workspace_bytes, void* stream);
/* Weighted sum of up to 8 scalar (0-dim, device) loss terms: backward == 0: out[0] = sum_i weights[i] * *terms[i] and out[1 + i] = weights[i] * *terms[i]
 * (host arrays of device pointers / host floats; `out` holds count + 1 floats); backward != 0: out[i] = weights[i] * *g (g NULL = 1) for i < count —
 * the gradient of every term. */
int cvae_weighted_sum(const float* const* terms, const float* weights, int count, const float* g, float* out, int backward, void* stream);
/* The same sum over a LIST of tensors (host arrays of `count` device pointers / sizes) in one launch + one finish; workspace: cvae_reduce_workspace_bytes()
 * (CVAE_E_WORKSPACE below count + 1 floats). */
int cvae_sqnorm_multi(const float* const* g, const int64_t* n, int count, float* out, void* workspace, size_t workspace_bytes, void* stream);
/* g *= *scale  (in place; scale is a device scalar) */
int cvae_scale(float* g, int64_t n, const float* scale, void* stream);
/* *scale = min(1, max_norm / (sqrt(*sqnorm) + 1e-6))   (clip_grad_norm_ coefficient, on device) */
int cvae_clip_coef(const float* sqnorm, float* scale, float max_norm, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CVAE_HIP_H */
