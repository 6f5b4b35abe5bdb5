// common.h — shared device helpers for libcvae_hip (gfx950 / CDNA4 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/cvae_hip.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define CVAE_WAVE 64

#define CVAE_CHECK_LAUNCH()                                   \
    do {                                                      \
        hipError_t e__ = hipGetLastError();                   \
        if (e__ != hipSuccess) return CVAE_E_LAUNCH;          \
    } while (0)

static inline int cvae_grid_1d(int64_t n, int block, int max_blocks = 256 * 8) {
    int64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > max_blocks) g = max_blocks;
    return (int)g;
}

// OCP fp8 e4m3 ("e4m3fn": max 448, no infinities) — the gfx950 format (MI300X's fnuz encoding is a different one).  One byte of storage;
// values carry a per-tensor scale that lives outside the tensor (the inference-only decode path, csrc/conv_mfma.hip).
struct fp8 { unsigned char b; };
#define CVAE_FP8_MAX 448.f

__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(bf16 x) { return (float)x; }
__device__ __forceinline__ float to_f32(fp8 x) { return __builtin_amdgcn_cvt_f32_fp8((int)x.b, 0); }
template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float x) { return (bf16)x; }   // v_cvt_pk_bf16_f32: RNE, NaN-safe
// two floats -> one dword of two bf16 (lo = a): ONE v_cvt_pk_bf16_f32, where two scalar casts cost two conversions and a v_perm to join them
typedef __bf16 cvae_bf16x2 __attribute__((ext_vector_type(2)));
typedef float cvae_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2_bf16(float a, float b) {
    const cvae_f32x2 f = {a, b};
    const cvae_bf16x2 r = __builtin_convertvector(f, cvae_bf16x2);
    return __builtin_bit_cast(uint32_t, r);
}
template <> __device__ __forceinline__ fp8 from_f32<fp8>(float x) {                       // saturating (the hardware conversion would give NaN past 448)
    x = fminf(fmaxf(x, -CVAE_FP8_MAX), CVAE_FP8_MAX);
    return fp8{(unsigned char)(__builtin_amdgcn_cvt_pk_fp8_f32(x, x, 0, false) & 0xff)};
}

// 8 floats -> 8 fp8 (e4m3) codes of v * mul, saturating at +-448 (v_cvt_pk_fp8_f32 alone would produce NaN past the range)
__device__ __forceinline__ uint2 pack8_fp8(const float (&v)[8], float mul) {
    float c[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) c[q] = __builtin_amdgcn_fmed3f(v[q] * mul, -CVAE_FP8_MAX, CVAE_FP8_MAX);
    int lo = __builtin_amdgcn_cvt_pk_fp8_f32(c[0], c[1], 0, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(c[2], c[3], lo, true);
    int hi = __builtin_amdgcn_cvt_pk_fp8_f32(c[4], c[5], 0, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(c[6], c[7], hi, true);
    return make_uint2((unsigned)lo, (unsigned)hi);
}
// The MFMA epilogues leave lane (r, h) with channels 8h..8h+7 (`a`) and 16+8h..23+8h (`b`) of its position's 32-channel block, as 8-byte fp8 pieces.
// One v_permlane32_swap per dword hands lane h = 0 channels 0..15 and lane h = 1 channels 16..31: ONE 16-byte store per lane at channel 16 h instead
// of two 8-byte ones (the swap exchanges the first operand of lanes 32-63 with the second operand of lanes 0-31).
__device__ __forceinline__ uint4 fp8_pair_to_16(uint2 a, uint2 b) {
    const auto x = __builtin_amdgcn_permlane32_swap(a.x, b.x, false, false);
    const auto y = __builtin_amdgcn_permlane32_swap(a.y, b.y, false, false);
    return make_uint4(x[0], y[0], x[1], y[1]);
}
// fp8 side channel of a producing kernel (all members may be null): `dscale` = device floats {acc_scale, 1 / s_out8} that replace by-value scales
// (a captured training step re-reads them on every replay: delayed scaling), `out8` = a second copy of the result as fp8 codes of result / s_out8
// (what the next fp8 layer reads, while the bf16 result stays for the backward pass), `amax` = CVAE_AMAX_SLOTS words that receive max |result| as
// float bits (non-negative floats order like unsigned integers).
struct F8Side {
    const float* dscale;
    fp8* out8;
    unsigned* amax;
    // ReLU masks as bits (round 3): one bit per element of the result, in element order — dword i covers elements 32 i .. 32 i + 31 (channel counts are
    // multiples of 32: a dword is one position's 32-channel block), bit set <=> element > 0.  `bits_out`: the producing launch leaves the mask of ITS result
    // (1/16 of the bf16 bytes) so that the backward launch that applies this ReLU reads `mask_bits` instead of the whole saved activation.
    const unsigned* mask_bits;
    unsigned* bits_out;
};
// this lane's two mask bytes (channels 8h..8h+7 -> `b0`, 16+8h..23+8h -> `b1` of its position's 32-channel block) -> the block's dword, in the lanes with
// h = 0 (one v_permlane32_swap; every lane must take part)
__device__ __forceinline__ unsigned mask_bytes_to_dword(unsigned b0, unsigned b1) {
    const unsigned a = b0 | (b1 << 16);
    const auto x = __builtin_amdgcn_permlane32_swap(a, a, false, false);       // lanes 0-31: (own, the partner lane's)
    return (x[0] & 0xffu) | ((x[1] & 0xffu) << 8) | (((x[0] >> 16) & 0xffu) << 16) | ((x[1] >> 16) << 24);
}
__device__ __forceinline__ unsigned mask_byte_of(const float (&v)[8]) {
    unsigned b = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) b |= (v[q] > 0.f ? 1u : 0u) << q;
    return b;
}
// max |.| of the workgroup (`red`: one float per wave, in LDS), then ONE atomicMax per WORKGROUP on the slot of its linear index: with
// CVAE_AMAX_SLOTS = 4096 the workgroups of a launch rarely share a word.  (One atomic per wave on 64 slots cost the 64 -> 32 channel layer 14 of
// its 23 us: same-address atomics serialise at the memory side at ~100 ns each.)  Every thread of the workgroup must call this.
__device__ __forceinline__ void amax_publish_wg(unsigned* slots, float amx, unsigned wg, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amx = fmaxf(amx, __shfl_xor(amx, o, 64));
    const int nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = amx;
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = red[0];
        for (int i = 1; i < nw; ++i) m = fmaxf(m, red[i]);
        if (m > 0.f) atomicMax(slots + (wg & (CVAE_AMAX_SLOTS - 1)), __float_as_uint(m));
    }
}

__device__ __forceinline__ float apply_act(float v, int act) {
    switch (act) {
        case CVAE_ACT_RELU: return v > 0.f ? v : 0.f;
        case CVAE_ACT_SIGMOID: return 1.f / (1.f + __expf(-v));
        case CVAE_ACT_LEAKY02: return v > 0.f ? v : 0.2f * v;
        default: return v;
    }
}
// The same with the activation fixed at compile time where a launch can afford a template instance per code: EPI 0 = none, 1 = ReLU
// (one v_max), 2 = the runtime code.  A runtime switch inside an unrolled epilogue compiles to a scalar branch PER ELEMENT (~100 taken
// branches per wave in the single-channel kernels), which costs more than the arithmetic it selects.
// ReLU as ONE instruction: fmaxf(v, 0) on a value the compiler cannot prove canonical (an MFMA result read back from an AGPR, a lane swap) costs
// a second v_max that quietens a possible signalling NaN first (and it folds v_med3(v, 0, inf) back into that pair), hence the asm.  NaN -> 0.
__device__ __forceinline__ float relu_f32(float v) {
    float r;
    asm("v_max_f32_e32 %0, 0, %1" : "=v"(r) : "v"(v));
    return r;
}
template <int EPI> __device__ __forceinline__ float apply_act_t(float v, int act) {
    if (EPI == 0) return v;
    if (EPI == 1) return relu_f32(v);
    return apply_act(v, act);
}
#define CVAE_EPI_OF(act) ((act) == CVAE_ACT_NONE ? 0 : ((act) == CVAE_ACT_RELU ? 1 : 2))
// derivative expressed through the activation OUTPUT y
__device__ __forceinline__ float act_grad_from_out(float y, int act) {
    switch (act) {
        case CVAE_ACT_RELU: return y > 0.f ? 1.f : 0.f;
        case CVAE_ACT_SIGMOID: return y * (1.f - y);
        case CVAE_ACT_LEAKY02: return y > 0.f ? 1.f : 0.2f;
        default: return 1.f;
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Block-wide sum; result valid in thread 0. `red` must hold blockDim.x/64 floats.
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    float r = 0.f;
    if (threadIdx.x == 0)
        for (int i = 0; i < nw; ++i) r += red[i];
    return r;
}

// 8 fp8 (e4m3) codes -> 8 bf16 (exact: every e4m3 value is a bf16 value)
__device__ __forceinline__ uint4 fp8x8_to_bf16x8(uint2 c) {
    const auto a = __builtin_amdgcn_cvt_pk_f32_fp8((int)c.x, false), b = __builtin_amdgcn_cvt_pk_f32_fp8((int)c.x, true);
    const auto d = __builtin_amdgcn_cvt_pk_f32_fp8((int)c.y, false), e = __builtin_amdgcn_cvt_pk_f32_fp8((int)c.y, true);
    return make_uint4(pack2_bf16(a[0], a[1]), pack2_bf16(b[0], b[1]), pack2_bf16(d[0], d[1]), pack2_bf16(e[0], e[1]));
}

// ---- Philox4x32-10 -> four N(0,1) draws (Box-Muller), shared by cvae_philox_normal* (losses.hip) and the bottleneck's first launch (bottleneck.hip):
// counter words 0-1 = position in the stream (4 normals each), words 2-3 = subsequence, key = seed.
__device__ __forceinline__ void philox_round(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}
__device__ __forceinline__ void philox_normal4(uint64_t ctr, uint64_t subseq, uint64_t seed, float v[4]) {
    uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = (uint32_t)subseq, c3 = (uint32_t)(subseq >> 32);
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) { philox_round(c0, c1, c2, c3, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
    const float u0 = ((float)(c0 >> 8) + 0.5f) * (1.f / 16777216.f), u1 = ((float)(c1 >> 8) + 0.5f) * (1.f / 16777216.f);
    const float u2 = ((float)(c2 >> 8) + 0.5f) * (1.f / 16777216.f), u3 = ((float)(c3 >> 8) + 0.5f) * (1.f / 16777216.f);
    const float r0 = sqrtf(-2.f * logf(u0)), r1 = sqrtf(-2.f * logf(u2));
    sincosf(6.283185307179586f * u1, &v[1], &v[0]);
    sincosf(6.283185307179586f * u3, &v[3], &v[2]);
    v[0] *= r0; v[1] *= r0; v[2] *= r1; v[3] *= r1;
}
