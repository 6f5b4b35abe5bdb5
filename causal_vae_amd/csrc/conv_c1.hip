// conv_c1.hip — the single-channel ends of the network (image -> 32 features, 32 features -> image): Cl == 1.
// These layers are HBM-bound (arithmetic intensity ~50 FLOP/B), so they are written as streaming
// kernels: the 1-channel halo tile and the whole (<= 8 KB) weight live in LDS as fp32, outputs leave as
// channel-contiguous rows.  The weight gradient is a [Cs x taps] = S^T · im2col(L) product with K = positions; it
// runs on the fp32 32x32x2 MFMA (one element per lane per operand, so no transposes) and leaves with fp32 atomics
// directly in the reference [Cs][1][taps] layout.
#include "common.h"

namespace {

template <int ND> struct TileC1;       // 256 output positions per workgroup
template <> struct TileC1<3> { static constexpr int TD = 4, TH = 8, TW = 8; };
template <> struct TileC1<2> { static constexpr int TD = 1, TH = 16, TW = 16; };

// -------------------------------------------------------------------------------------- down, Cl == 1
template <typename T, int ND, int CS>
__global__ __launch_bounds__(256) void down_c1_kernel(const T* __restrict__ L, const float* __restrict__ w, const float* __restrict__ bias,
                                                      const T* __restrict__ mask, T* __restrict__ S, int sd, int sh, int sw, int ld, int lh, int lw,
                                                      int tiles_h, int tiles_w, int act) {
    using TL = TileC1<ND>;
    constexpr int TD = TL::TD, TH = TL::TH, TW = TL::TW;
    constexpr int ID = (ND == 3) ? 2 * TD + 2 : 1, IH = 2 * TH + 2, IW = 2 * TW + 2;
    constexpr int NPOS = ID * IH * IW, TAPS = (ND == 3) ? 64 : 16, CH = CS / 2;
    __shared__ float halo[NPOS];
    __shared__ __attribute__((aligned(16))) float wl[TAPS * CS];      // [tap][cs]
    const int t = threadIdx.x, b = blockIdx.z;
    int tile = blockIdx.x;
    const int tw_i = tile % tiles_w; tile /= tiles_w;
    const int th_i = tile % tiles_h; tile /= tiles_h;
    const int o0d = tile * TD, o0h = th_i * TH, o0w = tw_i * TW;
    for (int i = t; i < TAPS * CS; i += 256) { const int tap = i / CS, cs = i % CS; wl[i] = w[cs * TAPS + tap]; }
    for (int pos = t; pos < NPOS; pos += 256) {
        const int x = pos % IW, y = pos / IW % IH, z = pos / (IW * IH);
        const int gz = (ND == 3) ? 2 * o0d - 1 + z : 0, gy = 2 * o0h - 1 + y, gx = 2 * o0w - 1 + x;
        const bool ok = gz >= 0 && gz < ld && gy >= 0 && gy < lh && gx >= 0 && gx < lw;
        halo[pos] = ok ? to_f32(L[(((size_t)b * ld + gz) * lh + gy) * lw + gx]) : 0.f;
    }
    __syncthreads();
    const int half = t & 1, pp = t >> 1;
    const int m0 = 2 * pp;                                  // two adjacent-w positions per thread
    const int w0 = m0 % TW, hh = m0 / TW % TH, d = m0 / (TW * TH);
    const int pb = ((2 * d) * IH + 2 * hh) * IW + 2 * w0;
    float acc0[CH], acc1[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) { acc0[c] = 0.f; acc1[c] = 0.f; }
#pragma unroll 4
    for (int tap = 0; tap < TAPS; ++tap) {
        const int kd = (ND == 3) ? (tap >> 4) : 0, kh = (tap >> 2) & 3, kw = tap & 3;
        const int off = (kd * IH + kh) * IW + kw;
        const float v0 = halo[pb + off], v1 = halo[pb + off + 2];
        const float4* wr = (const float4*)&wl[tap * CS + half * CH];
#pragma unroll
        for (int c4 = 0; c4 < CH / 4; ++c4) {
            const float4 ww = wr[c4];
            acc0[4 * c4 + 0] += v0 * ww.x; acc0[4 * c4 + 1] += v0 * ww.y; acc0[4 * c4 + 2] += v0 * ww.z; acc0[4 * c4 + 3] += v0 * ww.w;
            acc1[4 * c4 + 0] += v1 * ww.x; acc1[4 * c4 + 1] += v1 * ww.y; acc1[4 * c4 + 2] += v1 * ww.z; acc1[4 * c4 + 3] += v1 * ww.w;
        }
    }
    const int od = o0d + d, oh = o0h + hh;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int ow = o0w + w0 + q;
        if (od >= sd || oh >= sh || ow >= sw) continue;
        const size_t base = ((((size_t)b * sd + od) * sh + oh) * sw + ow) * CS + half * CH;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            float v = (q ? acc1[c] : acc0[c]) + (bias ? bias[half * CH + c] : 0.f);
            v = apply_act(v, act);
            if (mask && !(to_f32(mask[base + c]) > 0.f)) v = 0.f;
            S[base + c] = from_f32<T>(v);
        }
    }
}

// -------------------------------------------------------------------------------------- up, Cl == 1
// One thread per output voxel: 2 (s, k) pairs per strided dim -> 8 (3D) / 4 (2D) taps x CS channels.
template <typename T, int ND, int CS>
__global__ __launch_bounds__(256) void up_c1_kernel(const T* __restrict__ S, const float* __restrict__ w, const float* __restrict__ bias,
                                                    const T* __restrict__ mask, T* __restrict__ L, int B, int sd, int sh, int sw, int ld, int lh, int lw, int act) {
    constexpr int TAPS = (ND == 3) ? 64 : 16;
    __shared__ __attribute__((aligned(16))) float wl[TAPS * CS];      // [tap][cs]
    for (int i = threadIdx.x; i < TAPS * CS; i += 256) { const int tap = i / CS, cs = i % CS; wl[i] = w[cs * TAPS + tap]; }
    __syncthreads();
    const int64_t n = (int64_t)B * ld * lh * lw;
    const float b0 = bias ? bias[0] : 0.f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t rr = i;
        const int lx = (int)(rr % lw); rr /= lw;
        const int ly = (int)(rr % lh); rr /= lh;
        const int lz = (int)(rr % ld);
        const int b = (int)(rr / ld);
        float acc = 0.f;
#pragma unroll
        for (int a = 0; a < ((ND == 3) ? 2 : 1); ++a) {
            const int rz = lz & 1, sz = (ND == 3) ? (lz >> 1) - 1 + rz + a : 0, kd = (ND == 3) ? 3 - rz - 2 * a : 0;
            if (sz < 0 || sz >= sd) continue;
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
                const int ry = ly & 1, sy = (ly >> 1) - 1 + ry + bb, kh = 3 - ry - 2 * bb;
                if (sy < 0 || sy >= sh) continue;
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int rx = lx & 1, sx = (lx >> 1) - 1 + rx + c, kw = 3 - rx - 2 * c;
                    if (sx < 0 || sx >= sw) continue;
                    const T* sp = S + ((((size_t)b * sd + sz) * sh + sy) * sw + sx) * CS;
                    const float* wr = &wl[((kd * 4 + kh) * 4 + kw) * CS];
#pragma unroll
                    for (int c8 = 0; c8 < CS / 8; ++c8) {
                        __attribute__((aligned(16))) T v[8];
                        constexpr int NU = (8 * sizeof(T)) / 16;
#pragma unroll
                        for (int u = 0; u < NU; ++u) ((uint4*)v)[u] = ((const uint4*)(sp + 8 * c8))[u];
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc += to_f32(v[e]) * wr[8 * c8 + e];
                    }
                }
            }
        }
        float v = apply_act(acc + b0, act);
        if (mask && !(to_f32(mask[i]) > 0.f)) v = 0.f;
        L[i] = from_f32<T>(v);
    }
}

// -------------------------------------------------------------------------------------- wgrad, Cl == 1
template <typename T, int ND>
__global__ __launch_bounds__(256) void wgrad_c1_kernel(const T* __restrict__ S, const T* __restrict__ L, float* __restrict__ dW, int B, int sd, int sh, int sw,
                                                       int Cs, int ld, int lh, int lw, int tiles_d, int tiles_h, int tiles_w, int n_split) {
    constexpr int TD = (ND == 3) ? 4 : 1, TH = (ND == 3) ? 4 : 8, TW = (ND == 3) ? 8 : 16;     // 128 positions
    constexpr int ID = (ND == 3) ? 2 * TD + 2 : 1, IH = 2 * TH + 2, IW = 2 * TW + 2, NPOS = ID * IH * IW;
    constexpr int TAPS = (ND == 3) ? 64 : 16, NTS = (TAPS + 31) / 32;
    __shared__ float s_lds[128 * 32];
    __shared__ float l_lds[NPOS];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r = lane & 31, hk = lane >> 5;
    const int cs0 = blockIdx.y * 32;
    const int total = B * tiles_d * tiles_h * tiles_w;
    f32x16 acc[NTS];
#pragma unroll
    for (int s = 0; s < NTS; ++s)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[s][e] = 0.f;
    // per-lane tap offsets inside the halo tile (tap = ts*32 + r)
    int toff[NTS]; bool tvalid[NTS];
#pragma unroll
    for (int s = 0; s < NTS; ++s) {
        const int tap = s * 32 + r;
        tvalid[s] = tap < TAPS;
        const int kd = (ND == 3) ? (tap >> 4) & 3 : 0, kh = (tap >> 2) & 3, kw = tap & 3;
        toff[s] = (kd * IH + kh) * IW + kw;
    }
    for (int tile = blockIdx.x; tile < total; tile += n_split) {
        int tt = tile;
        const int tw_i = tt % tiles_w; tt /= tiles_w;
        const int th_i = tt % tiles_h; tt /= tiles_h;
        const int td_i = tt % tiles_d;
        const int b = tt / tiles_d;
        const int o0d = td_i * TD, o0h = th_i * TH, o0w = tw_i * TW;
        __syncthreads();
        for (int it = t; it < 128 * 32; it += 256) {
            const int c = it & 31, m = it >> 5;
            const int w = m % TW, hh = m / TW % TH, d = m / (TW * TH);
            const int od = o0d + d, oh = o0h + hh, ow = o0w + w;
            const bool ok = od < sd && oh < sh && ow < sw;
            s_lds[it] = ok ? to_f32(S[((((size_t)b * sd + od) * sh + oh) * sw + ow) * Cs + cs0 + c]) : 0.f;
        }
        for (int pos = t; pos < NPOS; pos += 256) {
            const int x = pos % IW, y = pos / IW % IH, z = pos / (IW * IH);
            const int gz = (ND == 3) ? 2 * o0d - 1 + z : 0, gy = 2 * o0h - 1 + y, gx = 2 * o0w - 1 + x;
            const bool ok = gz >= 0 && gz < ld && gy >= 0 && gy < lh && gx >= 0 && gx < lw;
            l_lds[pos] = ok ? to_f32(L[(((size_t)b * ld + gz) * lh + gy) * lw + gx]) : 0.f;
        }
        __syncthreads();
#pragma unroll 4
        for (int jj = 0; jj < 16; ++jj) {
            const int m = wave * 32 + 2 * jj + hk;
            const int w = m % TW, hh = m / TW % TH, d = m / (TW * TH);
            const int pb = ((2 * d) * IH + 2 * hh) * IW + 2 * w;
            const float a = s_lds[m * 32 + r];
#pragma unroll
            for (int s = 0; s < NTS; ++s) {
                const float bv = tvalid[s] ? l_lds[pb + toff[s]] : 0.f;
                acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[s], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int s = 0; s < NTS; ++s)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = (e & 3) + 8 * (e >> 2) + 4 * hk, tap = s * 32 + r;
            if (tap < TAPS) atomicAdd(&dW[(size_t)(cs0 + row) * TAPS + tap], acc[s][e]);
        }
}

}  // namespace

int cvae_conv_down_c1(const void* L, const float* w, const float* bias, const void* mask, void* S, int64_t B, int64_t sd, int64_t sh, int64_t sw,
                      int64_t Cs, int64_t ld, int64_t lh, int64_t lw, int nd, int dtype, int act, hipStream_t stream) {
    if (Cs != 32) return CVAE_E_UNSUPPORTED;
    const int th = (nd == 3) ? 8 : 16, tw = (nd == 3) ? 8 : 16, td = (nd == 3) ? 4 : 1;
    const int tiles_d = (int)((sd + td - 1) / td), tiles_h = (int)((sh + th - 1) / th), tiles_w = (int)((sw + tw - 1) / tw);
    dim3 grid((unsigned)(tiles_d * tiles_h * tiles_w), 1, (unsigned)B);
#define LAUNCH_DOWN_C1(T, ND)                                                                                                         \
    hipLaunchKernelGGL((down_c1_kernel<T, ND, 32>), grid, dim3(256), 0, stream, (const T*)L, w, bias, (const T*)mask, (T*)S, (int)sd, (int)sh, \
                       (int)sw, (int)ld, (int)lh, (int)lw, tiles_h, tiles_w, act)
    if (dtype == CVAE_BF16) { if (nd == 3) LAUNCH_DOWN_C1(bf16, 3); else LAUNCH_DOWN_C1(bf16, 2); }
    else { if (nd == 3) LAUNCH_DOWN_C1(float, 3); else LAUNCH_DOWN_C1(float, 2); }
#undef LAUNCH_DOWN_C1
    CVAE_CHECK_LAUNCH();
    return CVAE_OK;
}

int cvae_conv_up_c1(const void* S, const float* w, const float* bias, const void* mask, void* L, int64_t B, int64_t sd, int64_t sh, int64_t sw,
                    int64_t Cs, int64_t ld, int64_t lh, int64_t lw, int nd, int dtype, int act, hipStream_t stream) {
    if (Cs != 32) return CVAE_E_UNSUPPORTED;
    const int64_t n = B * ld * lh * lw;
    dim3 grid((unsigned)cvae_grid_1d(n, 256));
#define LAUNCH_UP_C1(T, ND)                                                                                                          \
    hipLaunchKernelGGL((up_c1_kernel<T, ND, 32>), grid, dim3(256), 0, stream, (const T*)S, w, bias, (const T*)mask, (T*)L, (int)B, (int)sd,   \
                       (int)sh, (int)sw, (int)ld, (int)lh, (int)lw, act)
    if (dtype == CVAE_BF16) { if (nd == 3) LAUNCH_UP_C1(bf16, 3); else LAUNCH_UP_C1(bf16, 2); }
    else { if (nd == 3) LAUNCH_UP_C1(float, 3); else LAUNCH_UP_C1(float, 2); }
#undef LAUNCH_UP_C1
    CVAE_CHECK_LAUNCH();
    return CVAE_OK;
}

int cvae_conv_wgrad_c1(const void* S, const void* L, float* dW, int64_t B, int64_t sd, int64_t sh, int64_t sw, int64_t Cs,
                       int64_t ld, int64_t lh, int64_t lw, int nd, int dtype, hipStream_t stream) {
    if (Cs % 32) return CVAE_E_UNSUPPORTED;
    const int taps = (nd == 3) ? 64 : 16;
    if (hipMemsetAsync(dW, 0, (size_t)Cs * taps * sizeof(float), stream) != hipSuccess) return CVAE_E_LAUNCH;
    const int td = (nd == 3) ? 4 : 1, th = (nd == 3) ? 4 : 8, tw = (nd == 3) ? 8 : 16;
    const int tiles_d = (int)((sd + td - 1) / td), tiles_h = (int)((sh + th - 1) / th), tiles_w = (int)((sw + tw - 1) / tw);
    const long long total = (long long)B * tiles_d * tiles_h * tiles_w;
    long long n_split = 1024 / (Cs / 32);
    if (n_split > total) n_split = total;
    if (n_split < 1) n_split = 1;
    dim3 grid((unsigned)n_split, (unsigned)(Cs / 32), 1);
#define LAUNCH_WG_C1(T, ND)                                                                                                           \
    hipLaunchKernelGGL((wgrad_c1_kernel<T, ND>), grid, dim3(256), 0, stream, (const T*)S, (const T*)L, dW, (int)B, (int)sd, (int)sh, (int)sw, \
                       (int)Cs, (int)ld, (int)lh, (int)lw, tiles_d, tiles_h, tiles_w, (int)n_split)
    if (dtype == CVAE_BF16) { if (nd == 3) LAUNCH_WG_C1(bf16, 3); else LAUNCH_WG_C1(bf16, 2); }
    else { if (nd == 3) LAUNCH_WG_C1(float, 3); else LAUNCH_WG_C1(float, 2); }
#undef LAUNCH_WG_C1
    CVAE_CHECK_LAUNCH();
    return CVAE_OK;
}
