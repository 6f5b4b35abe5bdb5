// losses.hip — fp32 reductions and small per-row kernels of the ELBO: reparameterise + KLD, SSE / BCE / weighted-MSE
// reconstruction terms, Gaussian NLL, softmax CE, uniform-KL, BatchNorm1d, Philox sampling, fused Adam, grad norm.
// Reduction pattern everywhere: float4 grid-stride loads -> per-thread partial -> wave64 xor-shuffle -> LDS across
// the 4 waves -> ONE plain store per block into the caller's scratch; a single-block finish launch adds the block sums in
// index order (cdna_hip_programming.md Guideline 12: "use the slab form when results must be bitwise reproducible").
// No float atomics anywhere: two runs on the same inputs give the same bits.
#include "common.h"

#define RED_BLOCK 256
#define RED_MAX_BLOCKS 1024
#define RED_MAX_OUT 2
extern "C" size_t cvae_reduce_workspace_bytes(void) { return (size_t)RED_MAX_BLOCKS * RED_MAX_OUT * sizeof(float); }
// Grid of a reduction over `items` work items: capped, and a single block when the caller gave no (or too little) scratch.
static inline int red_grid_ws(int64_t items, int block, int cap, int nout, const void* ws, size_t ws_bytes) {
    int g = cvae_grid_1d(items, block, cap < RED_MAX_BLOCKS ? cap : RED_MAX_BLOCKS);
    if (g > 1 && (!ws || ws_bytes < (size_t)g * nout * sizeof(float))) g = 1;
    return g;
}
// thread 0 of a block hands over its block sum: straight into *out when the launch is one block, else into slot blockIdx.x of row o
__device__ __forceinline__ void red_emit(float s, float* __restrict__ out, float* __restrict__ ws, int o) {
    if (gridDim.x == 1) out[o] += s;
    else ws[(size_t)o * gridDim.x + blockIdx.x] = s;
}
// out[o] += sum_b ws[o][b], b in index order (one block; NOUT rows)
template <int NOUT>
__global__ __launch_bounds__(RED_BLOCK) void red_finish_kernel(const float* __restrict__ ws, int nblocks, float* __restrict__ out) {
    __shared__ float red[RED_BLOCK / 64];
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
        float acc = 0.f;
        for (int i = threadIdx.x; i < nblocks; i += RED_BLOCK) acc += ws[(size_t)o * nblocks + i];
        const float s = block_sum(acc, red);
        if (threadIdx.x == 0) out[o] += s;
    }
}
#define RED_FINISH(NOUT, grid, ws, out, stream)                                                                      \
    do {                                                                                                              \
        if ((grid) > 1) {                                                                                             \
            hipLaunchKernelGGL((red_finish_kernel<NOUT>), dim3(1), dim3(RED_BLOCK), 0, (hipStream_t)(stream), (const float*)(ws), (int)(grid), out); \
            CVAE_CHECK_LAUNCH();                                                                                      \
        }                                                                                                             \
    } while (0)

// Generic two-input streaming reduction: F(a, b) -> float, summed.
template <typename F>
__global__ void reduce2_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, float* __restrict__ ws, int64_t n, F f) {
    __shared__ float red[RED_BLOCK / 64];
    float acc = 0.f;
    const int64_t n4 = n >> 2;
    const bool vec = ((((uintptr_t)a) | ((uintptr_t)b)) & 15) == 0;
    if (vec) {
        const float4* a4 = (const float4*)a;
        const float4* b4 = (const float4*)b;
        for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
            const float4 x = a4[i], y = b4[i];
            acc += f(x.x, y.x) + f(x.y, y.y) + f(x.z, y.z) + f(x.w, y.w);
        }
        for (int64_t i = (n4 << 2) + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) acc += f(a[i], b[i]);
    } else {
        for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) acc += f(a[i], b[i]);
    }
    const float s = block_sum(acc, red);
    if (threadIdx.x == 0) red_emit(s, out, ws, 0);
}

struct SseF { __device__ float operator()(float a, float b) const { const float d = a - b; return d * d; } };
struct SumF { __device__ float operator()(float a, float) const { return a; } };
struct SqF { __device__ float operator()(float a, float) const { return a * a; } };
struct BceF {   // -(x*max(log p, -100) + (1-x)*max(log(1-p), -100))  — aten binary_cross_entropy
    __device__ float operator()(float p, float x) const {
        const float lp = fmaxf(logf(p), -100.f), lq = fmaxf(log1pf(-p), -100.f);
        return -(x * lp + (1.f - x) * lq);
    }
};

#define LAUNCH_RED2(F, a, b, out, n, ws, wsb, stream)                                                                \
    do {                                                                                                              \
        if ((n) < 0) return CVAE_E_BADSHAPE;                                                                          \
        if ((n) == 0) return CVAE_OK;                                                                                 \
        if (!(a) || !(b) || !(out)) return CVAE_E_NULLPTR;                                                            \
        const int grid__ = red_grid_ws(((n) + 3) / 4, RED_BLOCK, RED_MAX_BLOCKS, 1, ws, wsb);                          \
        hipLaunchKernelGGL((reduce2_kernel<F>), dim3(grid__), dim3(RED_BLOCK), 0, (hipStream_t)(stream), a, b, out, (float*)(ws), n, F()); \
        CVAE_CHECK_LAUNCH();                                                                                          \
        RED_FINISH(1, grid__, ws, out, stream);                                                                       \
        return CVAE_OK;                                                                                               \
    } while (0)

extern "C" int cvae_sse_fwd(const float* a, const float* b, float* out, int64_t n, void* ws, size_t wsb, void* stream) { LAUNCH_RED2(SseF, a, b, out, n, ws, wsb, stream); }
extern "C" int cvae_sum_fwd(const float* x, float* out, int64_t n, void* ws, size_t wsb, void* stream) { LAUNCH_RED2(SumF, x, x, out, n, ws, wsb, stream); }
extern "C" int cvae_sqnorm(const float* g, float* out, int64_t n, void* ws, size_t wsb, void* stream) { LAUNCH_RED2(SqF, g, g, out, n, ws, wsb, stream); }
// *out += sum over a LIST of tensors of g^2 (the gradient norm of clip_grad_norm_) in ONE launch + one finish: every block sums one span of one
// tensor into its scratch slot, red_finish_kernel adds the slots in index order.  Per tensor this was two launches (~50 tensors: ~100 launches
// of < 5 us each in the vessel recipe's step).
#define SQM_MAX_TENSORS 64
struct SqTable {
    const float* g[SQM_MAX_TENSORS];
    long long n[SQM_MAX_TENSORS];
    int blk_start[SQM_MAX_TENSORS + 1];
    int count;
    long long span;
};
__global__ __launch_bounds__(RED_BLOCK) void sqnorm_multi_kernel(SqTable tb, float* __restrict__ out, float* __restrict__ ws) {
    __shared__ float red[RED_BLOCK / 64];
    int ti = 0;
    while (ti + 1 < tb.count && (int)blockIdx.x >= tb.blk_start[ti + 1]) ++ti;
    const float* __restrict__ g = tb.g[ti];
    const long long base = (long long)((int)blockIdx.x - tb.blk_start[ti]) * tb.span, end = min(tb.n[ti], base + tb.span);
    float acc = 0.f;
    if ((((uintptr_t)g) & 15) == 0 && (base & 3) == 0) {
        const long long v0 = base / 4, v1 = end / 4;
        for (long long i = v0 + threadIdx.x; i < v1; i += RED_BLOCK) { const float4 v = ((const float4*)g)[i]; acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w; }
        for (long long i = v1 * 4 + threadIdx.x; i < end; i += RED_BLOCK) acc += g[i] * g[i];
    } else {
        for (long long i = base + threadIdx.x; i < end; i += RED_BLOCK) acc += g[i] * g[i];
    }
    const float s = block_sum(acc, red);
    if (threadIdx.x == 0) red_emit(s, out, ws, 0);
}
extern "C" int cvae_sqnorm_multi(const float* const* g, const int64_t* n, int count, float* out, void* ws, size_t wsb, void* stream) {
    if (count < 0) return CVAE_E_BADSHAPE;
    if (count == 0) return CVAE_OK;
    if (!g || !n || !out) return CVAE_E_NULLPTR;
    for (int c0 = 0; c0 < count; c0 += SQM_MAX_TENSORS) {
        SqTable tb;
        const int cnt = (count - c0 < SQM_MAX_TENSORS) ? count - c0 : SQM_MAX_TENSORS;
        long long total = 0;
        for (int i = 0; i < cnt; ++i) { if (n[c0 + i] < 0) return CVAE_E_BADSHAPE; total += n[c0 + i]; }
        if (total == 0) continue;
        // spans of a multiple of 1024 elements, few enough blocks for the caller's scratch (one float per block) and for the finish pass
        long long cap = (ws && wsb >= sizeof(float)) ? (long long)(wsb / sizeof(float)) : 1;
        if (cap > RED_MAX_BLOCKS) cap = RED_MAX_BLOCKS;
        if (cap <= cnt) return CVAE_E_WORKSPACE;                  // at least one block per tensor
        tb.span = ((total + (cap - cnt) - 1) / (cap - cnt) + 1023) / 1024 * 1024;
        int blocks = 0, used = 0;
        for (int i = 0; i < cnt; ++i) {
            const int64_t ni = n[c0 + i];
            if (ni == 0) continue;
            if (!g[c0 + i]) return CVAE_E_NULLPTR;
            tb.g[used] = g[c0 + i]; tb.n[used] = ni; tb.blk_start[used] = blocks;
            blocks += (int)((ni + tb.span - 1) / tb.span);
            ++used;
        }
        tb.blk_start[used] = blocks; tb.count = used;
        hipLaunchKernelGGL(sqnorm_multi_kernel, dim3(blocks), dim3(RED_BLOCK), 0, (hipStream_t)stream, tb, out, (float*)ws);
        CVAE_CHECK_LAUNCH();
        RED_FINISH(1, blocks, ws, out, stream);
    }
    return CVAE_OK;
}
extern "C" int cvae_bce_fwd(const float* p, const float* x, float* out, int64_t n, void* ws, size_t wsb, void* stream) { LAUNCH_RED2(BceF, p, x, out, n, ws, wsb, stream); }

// Elementwise two-input map with a device-scalar upstream gradient.
template <typename F>
__global__ void map2_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ gout, float* __restrict__ o, int64_t n, F f, float scale) {
    const float g = (gout ? *gout : 1.f) * scale;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) o[i] = f(a[i], b[i]) * g;
}
struct SseB { __device__ float operator()(float a, float b) const { return 2.f * (a - b); } };
struct BceB {   // d/dp of BceF; aten: (p - x) / max((1-p)*p, 1e-12)
    __device__ float operator()(float p, float x) const { return (p - x) / fmaxf((1.f - p) * p, 1e-12f); }
};
#define LAUNCH_MAP2(F, a, b, g, o, n, scale, stream)                                                                        \
    do {                                                                                                              \
        if ((n) < 0) return CVAE_E_BADSHAPE;                                                                          \
        if ((n) == 0) return CVAE_OK;                                                                                 \
        if (!(a) || !(b) || !(o)) return CVAE_E_NULLPTR;                                                              \
        hipLaunchKernelGGL((map2_kernel<F>), dim3(cvae_grid_1d(n, 256)), dim3(256), 0, (hipStream_t)(stream), a, b, g, o, n, F(), scale); \
        CVAE_CHECK_LAUNCH();                                                                                          \
        return CVAE_OK;                                                                                               \
    } while (0)
extern "C" int cvae_sse_bwd(const float* a, const float* b, const float* gout, float scale, float* da, int64_t n, void* stream) { LAUNCH_MAP2(SseB, a, b, gout, da, n, scale, stream); }
extern "C" int cvae_bce_bwd(const float* p, const float* x, const float* gout, float* dp, int64_t n, void* stream) { LAUNCH_MAP2(BceB, p, x, gout, dp, n, 1.f, stream); }

// ------------------------------------------------------------------------------------- vessel recon terms
__device__ __forceinline__ float vessel_pos_weight(float sum_x, int64_t n) {
    const float pf = sum_x / ((float)n + 1e-6f);
    return fminf(fmaxf((1.f - pf) / (pf + 1e-6f), 1.f), 50.f);
}
__global__ void wmse_sparsity_fwd_kernel(const float* __restrict__ r, const float* __restrict__ x, const float* __restrict__ sum_x,
                                         float* __restrict__ out2, float* __restrict__ ws, int64_t n, int64_t n_pos) {
    __shared__ float red[RED_BLOCK / 64];
    const float pw = vessel_pos_weight(*sum_x, n_pos);
    float a0 = 0.f, a1 = 0.f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float rv = r[i], xv = x[i], d = rv - xv;
        a0 += d * d * (1.f + (pw - 1.f) * xv);
        a1 += (xv < 0.1f) ? fabsf(rv) : 0.f;
    }
    const float s0 = block_sum(a0, red);
    const float s1 = block_sum(a1, red);
    if (threadIdx.x == 0) { red_emit(s0, out2, ws, 0); red_emit(s1, out2, ws, 1); }
}
extern "C" int cvae_wmse_sparsity_fwd(const float* r, const float* x, const float* sum_x, float* out2, int64_t n, int64_t n_pos, void* ws, size_t wsb, void* stream) {
    if (n < 0 || n_pos < n) return CVAE_E_BADSHAPE;
    if (n == 0) return CVAE_OK;
    if (!r || !x || !sum_x || !out2) return CVAE_E_NULLPTR;
    const int grid = red_grid_ws(n, RED_BLOCK, RED_MAX_BLOCKS, 2, ws, wsb);
    hipLaunchKernelGGL(wmse_sparsity_fwd_kernel, dim3(grid), dim3(RED_BLOCK), 0, (hipStream_t)stream, r, x, sum_x, out2, (float*)ws, n, n_pos);
    CVAE_CHECK_LAUNCH();
    RED_FINISH(2, grid, ws, out2, stream);
    return CVAE_OK;
}
__global__ void wmse_sparsity_bwd_kernel(const float* __restrict__ r, const float* __restrict__ x, const float* __restrict__ sum_x,
                                         const float* __restrict__ g_recon, const float* __restrict__ g_sp, float* __restrict__ dr, int64_t n, int64_t n_pos) {
    const float pw = vessel_pos_weight(*sum_x, n_pos);
    const float gr = g_recon ? *g_recon : 0.f, gs = g_sp ? *g_sp : 0.f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float rv = r[i], xv = x[i];
        float g = gr * 2.f * (rv - xv) * (1.f + (pw - 1.f) * xv);
        if (xv < 0.1f) g += gs * ((rv > 0.f) ? 1.f : ((rv < 0.f) ? -1.f : 0.f));   // d|r|/dr = sign(r), 0 at 0 (aten)
        dr[i] = g;
    }
}
extern "C" int cvae_wmse_sparsity_bwd(const float* r, const float* x, const float* sum_x, const float* g_recon, const float* g_sp, float* dr, int64_t n, int64_t n_pos, void* stream) {
    if (n < 0 || n_pos < n) return CVAE_E_BADSHAPE;
    if (n == 0) return CVAE_OK;
    if (!r || !x || !sum_x || !dr) return CVAE_E_NULLPTR;
    hipLaunchKernelGGL(wmse_sparsity_bwd_kernel, dim3(cvae_grid_1d(n, 256)), dim3(256), 0, (hipStream_t)stream, r, x, sum_x, g_recon, g_sp, dr, n, n_pos);
    CVAE_CHECK_LAUNCH();
    return CVAE_OK;
}

// ------------------------------------------------------------------------------------- reparameterise + KLD
__global__ void reparam_kld_fwd_kernel(const float* __restrict__ mu, const float* __restrict__ logvar, const float* __restrict__ eps,
                                       float* __restrict__ z, float* __restrict__ kld, float* __restrict__ ws, int64_t n) {
    __shared__ float red[RED_BLOCK / 64];
    float acc = 0.f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float m = mu[i], lv = logvar[i];
        if (z) z[i] = m + eps[i] * expf(0.5f * lv);
        acc += 1.f + lv - m * m - expf(lv);
    }
    if (kld) {
        const float s = block_sum(acc, red);
        if (threadIdx.x == 0) red_emit(-0.5f * s, kld, ws, 0);
    }
}
extern "C" int cvae_reparam_kld_fwd(const float* mu, const float* logvar, const float* eps, float* z, float* kld, int64_t n, void* ws, size_t wsb, void* stream) {
    if (n < 0) return CVAE_E_BADSHAPE;
    if (n == 0) return CVAE_OK;
    if (!mu || !logvar || (z && !eps) || (!z && !kld)) return CVAE_E_NULLPTR;
    // without a KLD output nothing is reduced: any grid; with one, the block sums need scratch (or a single block)
    const int grid = kld ? red_grid_ws(n, RED_BLOCK, 256, 1, ws, wsb) : cvae_grid_1d(n, RED_BLOCK, 256);
    hipLaunchKernelGGL(reparam_kld_fwd_kernel, dim3(grid), dim3(RED_BLOCK), 0, (hipStream_t)stream, mu, logvar, eps, z, kld, (float*)ws, n);
    CVAE_CHECK_LAUNCH();
    if (kld) RED_FINISH(1, grid, ws, kld, stream);
    return CVAE_OK;
}
__global__ void reparam_kld_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ gkld, float gk_scale, const float* __restrict__ mu,
                                       const float* __restrict__ logvar, const float* __restrict__ eps, float* __restrict__ dmu,
                                       float* __restrict__ dlogvar, int64_t n) {
    const float gk = gkld ? *gkld * gk_scale : 0.f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float m = mu[i], lv = logvar[i];
        float gm = gk * m, gl = gk * 0.5f * (expf(lv) - 1.f);
        if (dz) { const float g = dz[i]; gm += g; gl += g * eps[i] * 0.5f * expf(0.5f * lv); }
        dmu[i] = gm;
        dlogvar[i] = gl;
    }
}
extern "C" int cvae_reparam_kld_bwd(const float* dz, const float* gkld, float gk_scale, const float* mu, const float* logvar, const float* eps,
                                    float* dmu, float* dlogvar, int64_t n, void* stream) {
    if (n < 0) return CVAE_E_BADSHAPE;
    if (n == 0) return CVAE_OK;
    if (!mu || !logvar || !dmu || !dlogvar || (dz && !eps)) return CVAE_E_NULLPTR;
    hipLaunchKernelGGL(reparam_kld_bwd_kernel, dim3(cvae_grid_1d(n, 256, 256)), dim3(256), 0, (hipStream_t)stream, dz, gkld, gk_scale, mu, logvar, eps, dmu, dlogvar, n);
    CVAE_CHECK_LAUNCH();
    return CVAE_OK;
}

// The latent head of a VAE step in one launch each way: h = [B][2 Z] rows (mu | logvar) exactly as the encoder's last Linear leaves them (no chunk copies),
// z1 = mu + eps1 exp(logvar / 2), optionally a second sample z2 from eps2 (the adversarial branch of mnist_test/01_baseline_causal_vae/train.py:78-81), and
// kld = -0.5 sum(1 + logvar - mu^2 - exp(logvar)) ASSIGNED (not accumulated) to *kld.  One workgroup: the reduction is a fixed tree, and a latent head is
// B x Z <= a few 10^4 elements.  The backward writes d h from d z1, d z2 and the KLD's incoming gradient: what autograd did with 3 kernels, 4 adds and a cat.
__global__ __launch_bounds__(1024) void latent_head_fwd_kernel(const float* __restrict__ h, const float* __restrict__ eps1, const float* __restrict__ eps2,
                                                               float* __restrict__ z1, float* __restrict__ z2, float* __restrict__ kld, int64_t B, int64_t Z) {
    __shared__ float red[16];
    const int n = (int)(B * Z), Zi = (int)Z;
    float acc = 0.f;
    for (int i0 = threadIdx.x; i0 < n; i0 += 4 * 1024) {     // four elements per pass: their loads are issued together, not one round trip each
        float m[4], lv[4], e1[4], e2[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * 1024, ic = i < n ? i : n - 1, b = ic / Zi, j = ic - b * Zi;
            m[u] = h[(size_t)b * 2 * Zi + j]; lv[u] = h[(size_t)b * 2 * Zi + Zi + j];
            e1[u] = z1 ? eps1[ic] : 0.f; e2[u] = z2 ? eps2[ic] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * 1024;
            if (i >= n) break;
            const float sd = expf(0.5f * lv[u]);
            if (z1) z1[i] = m[u] + e1[u] * sd;
            if (z2) z2[i] = m[u] + e2[u] * sd;
            acc += 1.f + lv[u] - m[u] * m[u] - expf(lv[u]);
        }
    }
    if (kld) {
        acc = wave_sum(acc);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            float s = 0.f;
            for (int w = 0; w < 16; ++w) s += red[w];
            *kld = -0.5f * s;
        }
    }
}
__global__ void latent_head_bwd_kernel(const float* __restrict__ dz1, const float* __restrict__ dz2, const float* __restrict__ gkld, const float* __restrict__ h,
                                       const float* __restrict__ eps1, const float* __restrict__ eps2, float* __restrict__ dh, int64_t B, int64_t Z) {
    const float gk = gkld ? *gkld : 0.f;
    const int64_t n = B * Z;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / Z, j = i - b * Z;
        const float m = h[b * 2 * Z + j], lv = h[b * 2 * Z + Z + j];
        float gm = gk * m, gl = gk * 0.5f * (expf(lv) - 1.f), ge = 0.f;
        if (dz1) { const float g = dz1[i]; gm += g; ge += g * eps1[i]; }
        if (dz2) { const float g = dz2[i]; gm += g; ge += g * eps2[i]; }
        gl += ge * 0.5f * expf(0.5f * lv);
        dh[b * 2 * Z + j] = gm;
        dh[b * 2 * Z + Z + j] = gl;
    }
}
extern "C" int cvae_latent_head_fwd(const float* h, const float* eps1, const float* eps2, float* z1, float* z2, float* kld, int64_t B, int64_t Z, void* stream) {
    if (B < 0 || Z <= 0 || B * Z > ((int64_t)1 << 24)) return CVAE_E_BADSHAPE;
    if (B == 0) return CVAE_OK;
    if (!h || (z1 && !eps1) || (z2 && !eps2) || (!z1 && !z2 && !kld)) return CVAE_E_NULLPTR;
    hipLaunchKernelGGL(latent_head_fwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, h, eps1, eps2, z1, z2, kld, B, Z);
    CVAE_CHECK_LAUNCH();
    return CVAE_OK;
}
extern "C" int cvae_latent_head_bwd(const float* dz1, const float* dz2, const float* gkld, const float* h, const float* eps1, const float* eps2, float* dh,
                                    int64_t B, int64_t Z, void* stream) {
    if (B < 0 || Z <= 0) return CVAE_E_BADSHAPE;
    if (B == 0) return CVAE_OK;
    if (!h || !dh || (dz1 && !eps1) || (dz2 && !eps2)) return CVAE_E_NULLPTR;
    hipLaunchKernelGGL(latent_head_bwd_kernel, dim3(cvae_grid_1d(B * Z, 256, 256)), dim3(256), 0, (hipStream_t)stream, dz1, dz2, gkld, h, eps1, eps2, dh, B, Z);
    CVAE_CHECK_LAUNCH();
    return CVAE_OK;
}

// out4 = {total, a, b, c} with total = a + wb * b + wc * c  (the ELBO of causal_cascade/train.py:16 from its three terms)
__global__ void combine3_kernel(float* __restrict__ out4, float wb, float wc) {
    if (threadIdx.x == 0 && blockIdx.x == 0) out4[0] = out4[1] + wb * out4[2] + wc * out4[3];
}
extern "C" int cvae_combine3(float* out4, float wb, float wc, void* stream) {
    if (!out4) return CVAE_E_NULLPTR;
    hipLaunchKernelGGL(combine3_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out4, wb, wc);
    CVAE_CHECK_LAUNCH();
    return CVAE_OK;
}

// ------------------------------------------------------------------------------------- Gaussian NLL
__global__ void gauss_nll_fwd_kernel(const float* __restrict__ m, const float* __restrict__ mu, const float* __restrict__ lv, float* __restrict__ out, float* __restrict__ ws, int64_t n) {
    __shared__ float red[RED_BLOCK / 64];
    float acc = 0.f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float d = m[i] - mu[i];
        acc += lv[i] + d * d / expf(lv[i]);
    }
    const float s = block_sum(acc, red);
    if (threadIdx.x == 0) red_emit(0.5f * s, out, ws, 0);
}
extern "C" int cvae_gauss_nll_fwd(const float* m, const float* mu, const float* lv, float* out, int64_t n, void* ws, size_t wsb, void* stream) {
    if (n < 0) return CVAE_E_BADSHAPE;
    if (n == 0) return CVAE_OK;
    if (!m || !mu || !lv || !out) return CVAE_E_NULLPTR;
    const int grid = red_grid_ws(n, RED_BLOCK, 256, 1, ws, wsb);
    hipLaunchKernelGGL(gauss_nll_fwd_kernel, dim3(grid), dim3(RED_BLOCK), 0, (hipStream_t)stream, m, mu, lv, out, (float*)ws, n);
    CVAE_CHECK_LAUNCH();
    RED_FINISH(1, grid, ws, out, stream);
    return CVAE_OK;
}
__global__ void gauss_nll_bwd_kernel(const float* __restrict__ m, const float* __restrict__ mu, const float* __restrict__ lv, const float* __restrict__ gout,
                                     float* __restrict__ dmu, float* __restrict__ dlv, int64_t n) {
    const float g = gout ? *gout : 1.f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float d = m[i] - mu[i], iv = 1.f / expf(lv[i]);
        dmu[i] = g * (-d * iv);
        dlv[i] = g * 0.5f * (1.f - d * d * iv);
    }
}
extern "C" int cvae_gauss_nll_bwd(const float* m, const float* mu, const float* lv, const float* gout, float* dmu, float* dlv, int64_t n, void* stream) {
    if (n < 0) return CVAE_E_BADSHAPE;
    if (n == 0) return CVAE_OK;
    if (!m || !mu || !lv || !dmu || !dlv) return CVAE_E_NULLPTR;
    hipLaunchKernelGGL(gauss_nll_bwd_kernel, dim3(cvae_grid_1d(n, 256, 256)), dim3(256), 0, (hipStream_t)stream, m, mu, lv, gout, dmu, dlv, n);
    CVAE_CHECK_LAUNCH();
    return CVAE_OK;
}

// ------------------------------------------------------------------------------------- softmax CE / uniform KL (rows of C <= 64 logits)
// One wave per row, lane = class: max and sum by xor-shuffle.
template <int MODE>   // 0: cross-entropy vs target; 1: KL(uniform || softmax)
__global__ void row_softmax_loss_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target, const float* __restrict__ gout,
                                        float* __restrict__ out, float* __restrict__ dlogits, float* __restrict__ ws, int64_t B, int64_t C) {
    __shared__ float red[4];
    const int lane = threadIdx.x & 63;
    const int64_t row0 = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6, nrow = ((int64_t)gridDim.x * blockDim.x) >> 6;
    float acc = 0.f;
    const float g = (dlogits && gout) ? *gout : 1.f;
    for (int64_t b = row0; b < B; b += nrow) {
        const float v = (lane < C) ? logits[b * C + lane] : -INFINITY;
        float mx = v;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        const float e = (lane < C) ? expf(v - mx) : 0.f;
        const float se = wave_sum(e);
        const float logp = v - mx - logf(se);             // log_softmax
        if (MODE == 0) {
            const int64_t t = target[b];
            if (dlogits) { if (lane < C) dlogits[b * C + lane] = g * (e / se - (lane == t ? 1.f : 0.f)) / (float)B; }
            else if (lane == t) acc += -logp / (float)B;
        } else {
            const float u = 1.f / (float)C;
            if (dlogits) { if (lane < C) dlogits[b * C + lane] = g * (e / se - u) / (float)B; }
            else if (lane < C) acc += u * (logf(u) - logp) / (float)B;   // kl_div(input=logp, target=u): u*(log u - logp), batchmean
        }
    }
    if (!dlogits) {                                          // wave sums -> block sum in wave order -> scratch slot (no atomics)
        const float s = block_sum(acc, red);
        if (threadIdx.x == 0) red_emit(s, out, ws, 0);
    }
}
static int row_loss(int mode, const float* logits, const int64_t* target, const float* gout, float* out, float* dl, int64_t B, int64_t C, void* ws, size_t wsb, void* stream) {
    if (B < 0 || C <= 0 || C > 64) return (C > 64) ? CVAE_E_UNSUPPORTED : CVAE_E_BADSHAPE;
    if (B == 0) return CVAE_OK;
    if (!logits || (mode == 0 && !target) || (!out && !dl)) return CVAE_E_NULLPTR;
    const int grid = dl ? cvae_grid_1d(B * 64, 256, 256) : red_grid_ws(B * 64, 256, 256, 1, ws, wsb);
    if (mode == 0) hipLaunchKernelGGL(row_softmax_loss_kernel<0>, dim3(grid), dim3(256), 0, (hipStream_t)stream, logits, target, gout, out, dl, (float*)ws, B, C);
    else hipLaunchKernelGGL(row_softmax_loss_kernel<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, logits, target, gout, out, dl, (float*)ws, B, C);
    CVAE_CHECK_LAUNCH();
    if (!dl) RED_FINISH(1, grid, ws, out, stream);
    return CVAE_OK;
}
extern "C" int cvae_softmax_ce_fwd(const float* logits, const int64_t* target, float* out, int64_t B, int64_t C, void* ws, size_t wsb, void* stream) { return row_loss(0, logits, target, nullptr, out, nullptr, B, C, ws, wsb, stream); }
extern "C" int cvae_softmax_ce_bwd(const float* logits, const int64_t* target, const float* gout, float* dl, int64_t B, int64_t C, void* stream) { return row_loss(0, logits, target, gout, nullptr, dl, B, C, nullptr, 0, stream); }
extern "C" int cvae_uniform_kl_fwd(const float* logits, float* out, int64_t B, int64_t C, void* ws, size_t wsb, void* stream) { return row_loss(1, logits, nullptr, nullptr, out, nullptr, B, C, ws, wsb, stream); }
extern "C" int cvae_uniform_kl_bwd(const float* logits, const float* gout, float* dl, int64_t B, int64_t C, void* stream) { return row_loss(1, logits, nullptr, gout, nullptr, dl, B, C, nullptr, 0, stream); }

// ------------------------------------------------------------------------------------- BatchNorm1d  (x [B, F], F <= a few hundred)
// One wave per feature, lanes stride the batch: two shuffle reductions (mean, then centred sum of squares — the
// two-pass form keeps fp32 accuracy at B = 4 where E[x^2]-E[x]^2 would cancel).
__global__ void bn1d_train_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ y,
                                      float* __restrict__ save_mean, float* __restrict__ save_rstd, float* __restrict__ rmean, float* __restrict__ rvar,
                                      int64_t B, int64_t F, float momentum, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t f = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
    if (f >= F) return;
    float s = 0.f;
    for (int64_t i = lane; i < B; i += 64) s += x[i * F + f];
    const float mean = wave_sum(s) / (float)B;
    float q = 0.f;
    for (int64_t i = lane; i < B; i += 64) { const float d = x[i * F + f] - mean; q += d * d; }
    q = wave_sum(q);
    const float var = q / (float)B, rstd = rsqrtf(var + eps);
    const float ww = w ? w[f] : 1.f, bb = b ? b[f] : 0.f;
    for (int64_t i = lane; i < B; i += 64) y[i * F + f] = (x[i * F + f] - mean) * rstd * ww + bb;
    if (lane == 0) {
        save_mean[f] = mean;
        save_rstd[f] = rstd;
        if (rmean) rmean[f] = (1.f - momentum) * rmean[f] + momentum * mean;
        if (rvar) rvar[f] = (1.f - momentum) * rvar[f] + momentum * (q / (float)(B - 1));
    }
}
extern "C" int cvae_bn1d_train_fwd(const float* x, const float* w, const float* b, float* y, float* save_mean, float* save_rstd,
                                   float* rmean, float* rvar, int64_t B, int64_t F, float momentum, float eps, void* stream) {
    if (B < 2 || F <= 0) return CVAE_E_BADSHAPE;      // nn.BatchNorm1d raises for B == 1 in train mode; the Python side raises ValueError first
    if (!x || !y || !save_mean || !save_rstd) return CVAE_E_NULLPTR;
    hipLaunchKernelGGL(bn1d_train_fwd_kernel, dim3((unsigned)((F * 64 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, w, b, y, save_mean, save_rstd, rmean, rvar, B, F, momentum, eps);
    CVAE_CHECK_LAUNCH();
    return CVAE_OK;
}
__global__ void bn1d_train_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ save_mean,
                                      const float* __restrict__ save_rstd, float* __restrict__ dx, float* __restrict__ dw, float* __restrict__ db, int64_t B, int64_t F) {
    const int lane = threadIdx.x & 63;
    const int64_t f = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
    if (f >= F) return;
    const float mean = save_mean[f], rstd = save_rstd[f], ww = w ? w[f] : 1.f;
    float s1 = 0.f, s2 = 0.f;
    for (int64_t i = lane; i < B; i += 64) { const float g = dy[i * F + f], xh = (x[i * F + f] - mean) * rstd; s1 += g; s2 += g * xh; }
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    const float inv = 1.f / (float)B;
    for (int64_t i = lane; i < B; i += 64) {
        const float g = dy[i * F + f], xh = (x[i * F + f] - mean) * rstd;
        dx[i * F + f] = ww * rstd * (g - s1 * inv - xh * s2 * inv);
    }
    if (lane == 0) { if (dw) dw[f] = s2; if (db) db[f] = s1; }
}
extern "C" int cvae_bn1d_train_bwd(const float* dy, const float* x, const float* w, const float* save_mean, const float* save_rstd,
                                   float* dx, float* dw, float* db, int64_t B, int64_t F, void* stream) {
    if (B < 2 || F <= 0) return CVAE_E_BADSHAPE;
    if (!dy || !x || !save_mean || !save_rstd || !dx) return CVAE_E_NULLPTR;
    hipLaunchKernelGGL(bn1d_train_bwd_kernel, dim3((unsigned)((F * 64 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy, x, w, save_mean, save_rstd, dx, dw, db, B, F);
    CVAE_CHECK_LAUNCH();
    return CVAE_OK;
}
// ---- BatchNorm1d with statistics that span data-parallel ranks (SyncBN): the same arithmetic cut at the two points where the batch
// sums are needed, so the host side can all-reduce them (2 x F floats forward, 2 x F backward) — a rank-local batch of 4 then
// normalises exactly like the reference's single-process batch of 32.  One wave per feature, as above.
// out[f] = sum_b x[b][f]                       (mean_sum == NULL)
// out[f] = sum_b (x[b][f] - mean_sum[f] * inv_n)^2   (mean_sum = the all-reduced sum, inv_n = 1 / global batch)
__global__ void bn1d_stats_kernel(const float* __restrict__ x, const float* __restrict__ mean_sum, float inv_n, float* __restrict__ out, int64_t B, int64_t F) {
    const int lane = threadIdx.x & 63;
    const int64_t f = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
    if (f >= F) return;
    const float mean = mean_sum ? mean_sum[f] * inv_n : 0.f;
    float s = 0.f;
    for (int64_t i = lane; i < B; i += 64) { const float d = x[i * F + f] - mean; s += mean_sum ? d * d : d; }
    s = wave_sum(s);
    if (lane == 0) out[f] = s;
}
extern "C" int cvae_bn1d_stats(const float* x, const float* mean_sum, float inv_n, float* out, int64_t B, int64_t F, void* stream) {
    if (B < 1 || F <= 0) return CVAE_E_BADSHAPE;
    if (!x || !out) return CVAE_E_NULLPTR;
    hipLaunchKernelGGL(bn1d_stats_kernel, dim3((unsigned)((F * 64 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, mean_sum, inv_n, out, B, F);
    CVAE_CHECK_LAUNCH();
    return CVAE_OK;
}
// y = (x - mean) * rstd * w + b with mean = mean_sum * inv_n, var = sqdev_sum * inv_n (both sums already reduced over ranks);
// running stats take the unbiased variance of the GLOBAL batch (n_global = 1 / inv_n).
__global__ void bn1d_apply_stats_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b, const float* __restrict__ mean_sum,
                                        const float* __restrict__ sqdev_sum, float inv_n, float* __restrict__ y, float* __restrict__ save_mean,
                                        float* __restrict__ save_rstd, float* __restrict__ rmean, float* __restrict__ rvar, int64_t B, int64_t F, float momentum,
                                        float eps, int update_running) {
    const int lane = threadIdx.x & 63;
    const int64_t f = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
    if (f >= F) return;
    const float mean = mean_sum[f] * inv_n, var = sqdev_sum[f] * inv_n, rstd = rsqrtf(var + eps);
    const float ww = w ? w[f] : 1.f, bb = b ? b[f] : 0.f;
    for (int64_t i = lane; i < B; i += 64) y[i * F + f] = (x[i * F + f] - mean) * rstd * ww + bb;
    if (lane == 0) {
        save_mean[f] = mean;
        save_rstd[f] = rstd;
        if (update_running && rmean) rmean[f] = (1.f - momentum) * rmean[f] + momentum * mean;
        const float n = 1.f / inv_n;
        if (update_running && rvar) rvar[f] = (1.f - momentum) * rvar[f] + momentum * (sqdev_sum[f] / (n - 1.f));
    }
}
extern "C" int cvae_bn1d_apply_stats(const float* x, const float* w, const float* b, const float* mean_sum, const float* sqdev_sum, float inv_n, float* y,
                                     float* save_mean, float* save_rstd, float* running_mean, float* running_var, int64_t B, int64_t F, float momentum,
                                     float eps, void* stream) {
    if (B < 1 || F <= 0 || !(inv_n > 0.f) || inv_n > 0.5f) return CVAE_E_BADSHAPE;       // a global batch of >= 2 samples
    if (!x || !mean_sum || !sqdev_sum || !y || !save_mean || !save_rstd) return CVAE_E_NULLPTR;
    hipLaunchKernelGGL(bn1d_apply_stats_kernel, dim3((unsigned)((F * 64 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, w, b, mean_sum, sqdev_sum, inv_n, y,
                       save_mean, save_rstd, running_mean, running_var, B, F, momentum, eps, 1);
    CVAE_CHECK_LAUNCH();
    return CVAE_OK;
}
// sums[f] = sum_b dy[b][f] (this rank's d beta), sums[F + f] = sum_b dy[b][f] * xhat[b][f] (this rank's d gamma)
__global__ void bn1d_bwd_sums_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ save_mean,
                                     const float* __restrict__ save_rstd, float* __restrict__ sums, int64_t B, int64_t F) {
    const int lane = threadIdx.x & 63;
    const int64_t f = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
    if (f >= F) return;
    const float mean = save_mean[f], rstd = save_rstd[f];
    float s1 = 0.f, s2 = 0.f;
    for (int64_t i = lane; i < B; i += 64) { const float g = dy[i * F + f]; s1 += g; s2 += g * (x[i * F + f] - mean) * rstd; }
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    if (lane == 0) { sums[f] = s1; sums[F + f] = s2; }
}
extern "C" int cvae_bn1d_bwd_sums(const float* dy, const float* x, const float* save_mean, const float* save_rstd, float* sums, int64_t B, int64_t F, void* stream) {
    if (B < 1 || F <= 0) return CVAE_E_BADSHAPE;
    if (!dy || !x || !save_mean || !save_rstd || !sums) return CVAE_E_NULLPTR;
    hipLaunchKernelGGL(bn1d_bwd_sums_kernel, dim3((unsigned)((F * 64 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy, x, save_mean, save_rstd, sums, B, F);
    CVAE_CHECK_LAUNCH();
    return CVAE_OK;
}
// dx = w * rstd * (dy - S1 / N - xhat * S2 / N) with S1, S2 the all-reduced sums and N the global batch
__global__ void bn1d_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ save_mean,
                                      const float* __restrict__ save_rstd, const float* __restrict__ sums, float inv_n, float* __restrict__ dx, int64_t B, int64_t F) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < B * F; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t f = i % F;
        const float rstd = save_rstd[f], xh = (x[i] - save_mean[f]) * rstd;
        dx[i] = (w ? w[f] : 1.f) * rstd * (dy[i] - sums[f] * inv_n - xh * sums[F + f] * inv_n);
    }
}
extern "C" int cvae_bn1d_bwd_apply(const float* dy, const float* x, const float* w, const float* save_mean, const float* save_rstd, const float* sums, float inv_n,
                                   float* dx, int64_t B, int64_t F, void* stream) {
    if (B < 1 || F <= 0 || !(inv_n > 0.f)) return CVAE_E_BADSHAPE;
    if (!dy || !x || !save_mean || !save_rstd || !sums || !dx) return CVAE_E_NULLPTR;
    hipLaunchKernelGGL(bn1d_bwd_apply_kernel, dim3(cvae_grid_1d(B * F, 256)), dim3(256), 0, (hipStream_t)stream, dy, x, w, save_mean, save_rstd, sums, inv_n, dx, B, F);
    CVAE_CHECK_LAUNCH();
    return CVAE_OK;
}
__global__ void bn1d_eval_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b, const float* __restrict__ rm,
                                     const float* __restrict__ rv, float* __restrict__ y, int64_t n, int64_t F, float eps) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t f = i % F;
        y[i] = (x[i] - rm[f]) / sqrtf(rv[f] + eps) * (w ? w[f] : 1.f) + (b ? b[f] : 0.f);
    }
}
extern "C" int cvae_bn1d_eval_fwd(const float* x, const float* w, const float* b, const float* rm, const float* rv, float* y, int64_t B, int64_t F, float eps, void* stream) {
    if (B < 0 || F <= 0) return CVAE_E_BADSHAPE;
    if (B == 0) return CVAE_OK;
    if (!x || !rm || !rv || !y) return CVAE_E_NULLPTR;
    hipLaunchKernelGGL(bn1d_eval_fwd_kernel, dim3(cvae_grid_1d(B * F, 256)), dim3(256), 0, (hipStream_t)stream, x, w, b, rm, rv, y, B * F, F, eps);
    CVAE_CHECK_LAUNCH();
    return CVAE_OK;
}

// ------------------------------------------------------------------------------------- Philox4x32-10 normals
// ADVANCE (single-block launches only): after every thread has read the call counter, thread 0 increments it — the draw and the
// "next call draws fresh numbers" bookkeeping in one launch instead of two.
// `subseq` selects one of 2^64 independent streams of the same key (Philox counter words 2 and 3): the Python side passes
// (rank << 32) | instance, so data-parallel ranks and the models of one process never share a stream.
template <bool ADVANCE>
__global__ void philox_normal_kernel(float* __restrict__ out, int64_t n, uint64_t seed, uint64_t offset, uint64_t subseq, int* __restrict__ call_dev) {
    if (call_dev) offset += ((uint64_t)(unsigned)(*call_dev)) << 24;       // device-side call counter: advances under graph replay
    if (ADVANCE) {
        __syncthreads();
        if (threadIdx.x == 0) *call_dev += 1;
    }
    const int64_t n4 = (n + 3) >> 2;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float v[4];
        philox_normal4(offset + (uint64_t)i, subseq, seed, v);
        for (int j = 0; j < 4; ++j) if (i * 4 + j < n) out[i * 4 + j] = v[j];
    }
}
extern "C" int cvae_philox_normal(float* out, int64_t n, uint64_t seed, uint64_t offset, uint64_t subsequence, const int* call_counter, void* stream) {
    if (n < 0) return CVAE_E_BADSHAPE;
    if (n == 0) return CVAE_OK;
    if (!out) return CVAE_E_NULLPTR;
    hipLaunchKernelGGL(philox_normal_kernel<false>, dim3(cvae_grid_1d((n + 3) / 4, 256)), dim3(256), 0, (hipStream_t)stream, out, n, seed, offset, subsequence, (int*)call_counter);
    CVAE_CHECK_LAUNCH();
    return CVAE_OK;
}
__global__ void add_int_kernel(int* c, int delta);
extern "C" int cvae_philox_normal_advance(float* out, int64_t n, uint64_t seed, uint64_t offset, uint64_t subsequence, int* call_counter, void* stream) {
    if (n < 0) return CVAE_E_BADSHAPE;
    if (!call_counter) return CVAE_E_NULLPTR;
    if (n > 0 && !out) return CVAE_E_NULLPTR;
    if (n > 0 && n <= 16384) {                               // small draws (the latent noise): one block does both
        hipLaunchKernelGGL(philox_normal_kernel<true>, dim3(1), dim3(256), 0, (hipStream_t)stream, out, n, seed, offset, subsequence, call_counter);
    } else {
        if (n > 0) hipLaunchKernelGGL(philox_normal_kernel<false>, dim3(cvae_grid_1d((n + 3) / 4, 256)), dim3(256), 0, (hipStream_t)stream, out, n, seed, offset, subsequence, call_counter);
        hipLaunchKernelGGL(add_int_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, call_counter, 1);
    }
    CVAE_CHECK_LAUNCH();
    return CVAE_OK;
}

// ------------------------------------------------------------------------------------- Adam + clip
// 28 B/param of HBM traffic (read p, g, m, v; write p, m, v) in float4 rows.
__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, float b1, float b2, float eps, float step, float rbc2) {
    m = b1 * m + (1.f - b1) * g;
    v = b2 * v + (1.f - b2) * g * g;
    p -= step * (m / (sqrtf(v) * rbc2 + eps));
}
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int64_t n,
                            float lr, float b1, float b2, float eps, float bc1, float bc2, const float* __restrict__ gscale) {
    const float gs = gscale ? *gscale : 1.f;
    const float step = lr / bc1, rbc2 = 1.f / sqrtf(bc2);
    const int64_t n4 = n >> 2;
    const bool vec = ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0;
    if (vec) {
        for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
            float4 P = ((float4*)p)[i], G = ((const float4*)g)[i], M = ((float4*)m)[i], V = ((float4*)v)[i];
            adam1(P.x, G.x * gs, M.x, V.x, b1, b2, eps, step, rbc2);
            adam1(P.y, G.y * gs, M.y, V.y, b1, b2, eps, step, rbc2);
            adam1(P.z, G.z * gs, M.z, V.z, b1, b2, eps, step, rbc2);
            adam1(P.w, G.w * gs, M.w, V.w, b1, b2, eps, step, rbc2);
            ((float4*)p)[i] = P; ((float4*)m)[i] = M; ((float4*)v)[i] = V;
        }
        for (int64_t i = (n4 << 2) + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
            adam1(p[i], g[i] * gs, m[i], v[i], b1, b2, eps, step, rbc2);
    } else {
        for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
            adam1(p[i], g[i] * gs, m[i], v[i], b1, b2, eps, step, rbc2);
    }
}
extern "C" int cvae_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps,
                              float bc1, float bc2, const float* gscale, void* stream) {
    if (n < 0 || bc1 <= 0.f || bc2 <= 0.f) return CVAE_E_BADSHAPE;
    if (n == 0) return CVAE_OK;
    if (!p || !g || !m || !v) return CVAE_E_NULLPTR;
    hipLaunchKernelGGL(adam_kernel, dim3(cvae_grid_1d((n + 3) / 4, 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, b1, b2, eps, bc1, bc2, gscale);
    CVAE_CHECK_LAUNCH();
    return CVAE_OK;
}
// Multi-tensor Adam: ONE launch for the whole parameter list (the pointer table travels as a kernel argument), each block
// owns a 1024-element span of one tensor (one float4 per stream per thread) and uses nontemporal loads and stores: every byte is
// touched once per step, so keeping it out of L2 leaves the cache to the activations (measured 5.2 vs 3.5 TB/s with 4096-spans).  `step_dev` (optional) is a device step counter: when given, the bias
// corrections are computed in-kernel so that a captured HIP graph replays with advancing corrections.
#define ADAM_MAX_TENSORS 64      // 2.8 KB of kernel arguments (limit 4 KB): the model's ~50 parameter tensors go in one launch
#define ADAM_SPAN 4096
#define ADAM_MULTI_SPAN 1024
struct AdamTable {
    float* p[ADAM_MAX_TENSORS];
    const float* g[ADAM_MAX_TENSORS];
    float* m[ADAM_MAX_TENSORS];
    float* v[ADAM_MAX_TENSORS];
    long long n[ADAM_MAX_TENSORS];
    int blk_start[ADAM_MAX_TENSORS + 1];
    int count;
};
template <int SPAN, bool NT = false>
__global__ __launch_bounds__(256) void adam_multi_kernel(AdamTable tb, float lr, float b1, float b2, float eps, float bc1, float bc2,
                                                         const int* __restrict__ step_dev, const float* __restrict__ gscale) {
    int ti = 0;
    while (ti + 1 < tb.count && (int)blockIdx.x >= tb.blk_start[ti + 1]) ++ti;
    if (step_dev) { const float st = (float)(*step_dev); bc1 = 1.f - powf(b1, st); bc2 = 1.f - powf(b2, st); }
    const float gs = gscale ? *gscale : 1.f;
    const float step = lr / bc1, rbc2 = 1.f / sqrtf(bc2);
    float* p = tb.p[ti]; const float* g = tb.g[ti]; float* m = tb.m[ti]; float* v = tb.v[ti];
    const long long n = tb.n[ti];
    const long long base = (long long)((int)blockIdx.x - tb.blk_start[ti]) * SPAN;
    const long long end = min(n, base + SPAN);
    const bool vec = ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0;
    if (vec && end - base == SPAN) {
#pragma unroll
        for (int k = 0; k < SPAN / (256 * 4); ++k) {
            const long long i = base / 4 + k * 256 + threadIdx.x;
            float4 P, G, M, V;
            if (NT) {
                typedef float f4v __attribute__((ext_vector_type(4)));
                f4v a = __builtin_nontemporal_load((f4v*)p + i), b = __builtin_nontemporal_load((const f4v*)g + i);
                f4v c = __builtin_nontemporal_load((f4v*)m + i), d = __builtin_nontemporal_load((f4v*)v + i);
                P = make_float4(a.x, a.y, a.z, a.w); G = make_float4(b.x, b.y, b.z, b.w); M = make_float4(c.x, c.y, c.z, c.w); V = make_float4(d.x, d.y, d.z, d.w);
            } else { P = ((float4*)p)[i]; G = ((const float4*)g)[i]; M = ((float4*)m)[i]; V = ((float4*)v)[i]; }
            adam1(P.x, G.x * gs, M.x, V.x, b1, b2, eps, step, rbc2);
            adam1(P.y, G.y * gs, M.y, V.y, b1, b2, eps, step, rbc2);
            adam1(P.z, G.z * gs, M.z, V.z, b1, b2, eps, step, rbc2);
            adam1(P.w, G.w * gs, M.w, V.w, b1, b2, eps, step, rbc2);
            if (NT) {
                typedef float f4v __attribute__((ext_vector_type(4)));
                f4v a = {P.x, P.y, P.z, P.w}, c = {M.x, M.y, M.z, M.w}, d = {V.x, V.y, V.z, V.w};
                __builtin_nontemporal_store(a, (f4v*)p + i); __builtin_nontemporal_store(c, (f4v*)m + i); __builtin_nontemporal_store(d, (f4v*)v + i);
            }
            else { ((float4*)p)[i] = P; ((float4*)m)[i] = M; ((float4*)v)[i] = V; }
        }
    } else {
        for (long long i = base + threadIdx.x; i < end; i += 256) adam1(p[i], g[i] * gs, m[i], v[i], b1, b2, eps, step, rbc2);
    }
}
extern "C" int cvae_adam_multi(float* const* p, const float* const* g, float* const* m, float* const* v, const int64_t* n, int count,
                               float lr, float b1, float b2, float eps, float bc1, float bc2, const int* step_dev, const float* gscale, void* stream) {
    if (count < 0) return CVAE_E_BADSHAPE;
    if (count == 0) return CVAE_OK;
    if (!p || !g || !m || !v || !n) return CVAE_E_NULLPTR;
    if (!step_dev && (bc1 <= 0.f || bc2 <= 0.f)) return CVAE_E_BADSHAPE;
    const int span = ADAM_MULTI_SPAN;
    for (int c0 = 0; c0 < count; c0 += ADAM_MAX_TENSORS) {
        AdamTable tb;
        const int cnt = (count - c0 < ADAM_MAX_TENSORS) ? count - c0 : ADAM_MAX_TENSORS;
        int blocks = 0, used = 0;
        for (int i = 0; i < cnt; ++i) {
            const int64_t ni = n[c0 + i];
            if (ni < 0) return CVAE_E_BADSHAPE;
            if (ni == 0) continue;
            if (!p[c0 + i] || !g[c0 + i] || !m[c0 + i] || !v[c0 + i]) return CVAE_E_NULLPTR;
            tb.p[used] = p[c0 + i]; tb.g[used] = g[c0 + i]; tb.m[used] = m[c0 + i]; tb.v[used] = v[c0 + i]; tb.n[used] = ni;
            tb.blk_start[used] = blocks;
            blocks += (int)((ni + span - 1) / span);
            ++used;
        }
        if (!used) continue;
        tb.blk_start[used] = blocks;
        tb.count = used;
        hipLaunchKernelGGL((adam_multi_kernel<ADAM_MULTI_SPAN, true>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, tb, lr, b1, b2, eps, bc1, bc2, step_dev, gscale);
        CVAE_CHECK_LAUNCH();
    }
    return CVAE_OK;
}
// Multi-tensor fp32 copy (pack gradients into the flat all-reduce bucket and back): one launch for the whole list.
struct CopyTable {
    const float* src[ADAM_MAX_TENSORS];
    float* dst[ADAM_MAX_TENSORS];
    long long n[ADAM_MAX_TENSORS];
    int blk_start[ADAM_MAX_TENSORS + 1];
    int count;
};
__global__ __launch_bounds__(256) void multi_copy_kernel(CopyTable tb) {
    int ti = 0;
    while (ti + 1 < tb.count && (int)blockIdx.x >= tb.blk_start[ti + 1]) ++ti;
    const float* s = tb.src[ti];
    float* d = tb.dst[ti];
    const long long base = (long long)((int)blockIdx.x - tb.blk_start[ti]) * ADAM_SPAN, end = min(tb.n[ti], base + ADAM_SPAN);
    if (((((uintptr_t)s) | ((uintptr_t)d)) & 15) == 0 && end - base == ADAM_SPAN) {
#pragma unroll
        for (int k = 0; k < ADAM_SPAN / (256 * 4); ++k) ((float4*)d)[base / 4 + k * 256 + threadIdx.x] = ((const float4*)s)[base / 4 + k * 256 + threadIdx.x];
    } else {
        for (long long i = base + threadIdx.x; i < end; i += 256) d[i] = s[i];
    }
}
extern "C" int cvae_multi_copy(const float* const* src, float* const* dst, const int64_t* n, int count, void* stream) {
    if (count < 0) return CVAE_E_BADSHAPE;
    if (count == 0) return CVAE_OK;
    if (!src || !dst || !n) return CVAE_E_NULLPTR;
    for (int c0 = 0; c0 < count; c0 += ADAM_MAX_TENSORS) {
        CopyTable tb;
        const int cnt = (count - c0 < ADAM_MAX_TENSORS) ? count - c0 : ADAM_MAX_TENSORS;
        int blocks = 0, used = 0;
        for (int i = 0; i < cnt; ++i) {
            const int64_t ni = n[c0 + i];
            if (ni < 0) return CVAE_E_BADSHAPE;
            if (ni == 0) continue;
            if (!src[c0 + i] || !dst[c0 + i]) return CVAE_E_NULLPTR;
            tb.src[used] = src[c0 + i]; tb.dst[used] = dst[c0 + i]; tb.n[used] = ni; tb.blk_start[used] = blocks;
            blocks += (int)((ni + ADAM_SPAN - 1) / ADAM_SPAN);
            ++used;
        }
        if (!used) continue;
        tb.blk_start[used] = blocks;
        tb.count = used;
        hipLaunchKernelGGL(multi_copy_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, tb);
        CVAE_CHECK_LAUNCH();
    }
    return CVAE_OK;
}
__global__ void add_int_kernel(int* c, int delta) { if (threadIdx.x == 0 && blockIdx.x == 0) *c += delta; }
extern "C" int cvae_counter_add(int* counter, int delta, void* stream) {
    if (!counter) return CVAE_E_NULLPTR;
    hipLaunchKernelGGL(add_int_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, counter, delta);
    CVAE_CHECK_LAUNCH();
    return CVAE_OK;
}

__global__ void scale_kernel(float* __restrict__ g, int64_t n, const float* __restrict__ scale) {
    const float s = *scale;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) g[i] *= s;
}
extern "C" int cvae_scale(float* g, int64_t n, const float* scale, void* stream) {
    if (n < 0) return CVAE_E_BADSHAPE;
    if (n == 0) return CVAE_OK;
    if (!g || !scale) return CVAE_E_NULLPTR;
    hipLaunchKernelGGL(scale_kernel, dim3(cvae_grid_1d(n, 256)), dim3(256), 0, (hipStream_t)stream, g, n, scale);
    CVAE_CHECK_LAUNCH();
    return CVAE_OK;
}
__global__ void clip_coef_kernel(const float* __restrict__ sq, float* __restrict__ scale, float max_norm) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *scale = fminf(1.f, max_norm / (sqrtf(*sq) + 1e-6f));
}
// out[0] = sum_i w[i] * *t[i] and out[1 + i] = w[i] * *t[i] (forward: the total and the weighted terms a training loop logs), or out[i] = w[i] * *g for
// every i (backward, g NULL = 1): the weighted sum of up to 8 scalar loss terms and its gradients as ONE launch each (composed from torch scalar ops
// the vessel recipe's total was ~20 launches of < 5 us).
struct Scalars8 { const float* t[8]; float w[8]; int n; };
__global__ void weighted_sum_kernel(Scalars8 a, const float* __restrict__ g, float* __restrict__ out, int backward) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (backward) {
        const float gv = g ? *g : 1.f;
        for (int i = 0; i < a.n; ++i) out[i] = a.w[i] * gv;
    } else {
        float s = 0.f;
        for (int i = 0; i < a.n; ++i) { const float v = a.w[i] * *a.t[i]; out[1 + i] = v; s += v; }     // index order: reproducible
        out[0] = s;
    }
}
extern "C" int cvae_weighted_sum(const float* const* terms, const float* weights, int count, const float* g, float* out, int backward, void* stream) {
    if (count < 1 || count > 8) return CVAE_E_BADSHAPE;
    if (!weights || !out || (!backward && !terms)) return CVAE_E_NULLPTR;
    Scalars8 a;
    a.n = count;
    for (int i = 0; i < count; ++i) {
        a.t[i] = backward ? nullptr : terms[i];
        if (!backward && !terms[i]) return CVAE_E_NULLPTR;
        a.w[i] = weights[i];
    }
    hipLaunchKernelGGL(weighted_sum_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, g, out, backward);
    CVAE_CHECK_LAUNCH();
    return CVAE_OK;
}
extern "C" int cvae_clip_coef(const float* sq, float* scale, float max_norm, void* stream) {
    if (!sq || !scale) return CVAE_E_NULLPTR;
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sq, scale, max_norm);
    CVAE_CHECK_LAUNCH();
    return CVAE_OK;
}
