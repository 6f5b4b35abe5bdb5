// small_dense.hip — nn.Linear (+ activation) for SMALL layers at LARGE batch: N * K <= 12288 weights, batch M > 16.
//
// The MLP heads of the MNIST model (mnist_test/01_baseline_causal_vae/models.py:24-37, 93-111: 10 -> 64 -> 64 -> 10, 10 -> 128 -> 12, 512 -> 20) at batch 1024
// are 20-odd products whose weight fits in LDS many times over.  On the 64 x 64-tile GEMM (linear.hip) each of them is a tile grid that is mostly padding
// (N = 10 .. 64 in a 64-wide tile), a split over the batch for the weight gradients, a slab sum and a separate bias column sum: 3-4 launches of a few
// microseconds for a few hundred kFLOP.  Here a layer is ONE launch forward, ONE for the data gradient and TWO for the weight + bias gradient:
//   fwd         y[m][n]  = act(sum_k x[m][k] W[n][k] + b[n])                                      workgroup = 16 batch rows, W and the rows in LDS
//   bwd_data    dx[m][k] = (sum_n g[m][n] W[n][k]) * in_act'(x_in[m][k]),  g = dy * act'(y)         the same; both activation gradients folded in
//   bwd_weight  part[wg][n][k] = sum_{m in wg's 16 rows} g[m][n] x[m][k],  part[wg][N K + n] = sum_m g[m][n];  then a fixed-order sum over the workgroups
// fp32 FMAs in a fixed order (no atomics): two runs give the same bits.  All global -> LDS traffic is batched (every load of a thread issued before its first
// LDS store: these kernels are a handful of dependent round trips long).
#include "common.h"

namespace {

constexpr int SD_ROWS = 16;            // batch rows per workgroup
constexpr int SD_MAX_NK = 12288;       // weights that fit: 48 KB of LDS
#ifndef CVAE_SD_MAX_DIM
#define CVAE_SD_MAX_DIM 128
#endif
constexpr int SD_MAX_DIM = CVAE_SD_MAX_DIM;   // beyond 128 inputs / outputs the per-thread dot products get long and the tiled GEMM wins (512 -> 20: 17 / 27 / 16 us here against 12 / 10 / 15)

// dst[r * pitch + c] = src[r * src_stride + c] for r < rows: every load of the thread is issued before its first LDS store.  A contiguous, 16-byte aligned
// source (the weight: src_stride == cols) goes as float4s, 12 per thread = the whole 48 KB in ONE pass of the workgroup.
__device__ __forceinline__ void sd_copy2d(float* __restrict__ dst, const float* __restrict__ src, int rows, int cols, int pitch, int64_t src_stride) {
    const int n = rows * cols;
    if (src_stride == cols && (n & 3) == 0 && (((uintptr_t)src) & 15) == 0) {
        constexpr int U = 12;
        const int n4 = n >> 2;
        for (int base = threadIdx.x; base < n4; base += 256 * U) {
            float4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = *(const float4*)(src + 4 * (size_t)min(base + 256 * u, n4 - 1));
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i4 = base + 256 * u;
                if (i4 < n4) {
                    const float e[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) { const int i = 4 * i4 + q, r = i / cols; dst[r * pitch + (i - r * cols)] = e[q]; }
                }
            }
        }
        return;
    }
    constexpr int U = 16;
    for (int base = threadIdx.x; base < n; base += 256 * U) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = min(base + 256 * u, n - 1), r = i / cols;
            v[u] = src[(int64_t)r * src_stride + (i - r * cols)];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = base + 256 * u;
            if (i < n) { const int r = i / cols; dst[r * pitch + (i - r * cols)] = v[u]; }
        }
    }
}

// g[r][n] = dy[r][n] * act'(y[r][n]) (act' from the activation's output; ACT_NONE: the copy) into LDS, rows beyond M zero
__device__ __forceinline__ void sd_load_g(float* __restrict__ gs, const float* __restrict__ dy, const float* __restrict__ y, int act, int64_t m0, int64_t M, int N,
                                          int64_t dy_stride, int64_t y_stride) {
    const int n = SD_ROWS * N;
    constexpr int U = 16;
    for (int base = threadIdx.x; base < n; base += 256 * U) {
        float v[U], yv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = min(base + 256 * u, n - 1), r = i / N, c = i - r * N;
            const int64_t m = min(m0 + r, M - 1);
            v[u] = dy[m * dy_stride + c];
            yv[u] = y ? y[m * y_stride + c] : 1.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = base + 256 * u;
            if (i < n) gs[i] = (m0 + i / N < M) ? (y ? v[u] * act_grad_from_out(yv[u], act) : v[u]) : 0.f;
        }
    }
}

__global__ __launch_bounds__(256) void sd_fwd_kernel(const float* __restrict__ x, const float* __restrict__ W, const float* __restrict__ b, float* __restrict__ y,
                                                     int64_t M, int K, int N, int64_t x_stride, int64_t y_stride, int act) {
    extern __shared__ float lds[];
    const int KP = K | 1;                                    // odd pitch: threads of a wave read different rows of W at the same k
    float* ws = lds;                                         // [N][KP]
    float* xs = ws + N * KP;                                 // [SD_ROWS][K]
    const int64_t m0 = (int64_t)blockIdx.x * SD_ROWS;
    const int rows = (int)min((int64_t)SD_ROWS, M - m0);
    sd_copy2d(xs, x + m0 * x_stride, rows, K, K, x_stride);
    for (int i = threadIdx.x + rows * K; i < SD_ROWS * K; i += 256) xs[i] = 0.f;
    sd_copy2d(ws, W, N, K, KP, K);
    __syncthreads();
    // thread = (output column n, group of 4 rows): one weight read feeds 4 FMAs (rows beyond `rows` hold zeros in xs and are not stored)
    for (int o = threadIdx.x; o < (SD_ROWS / 4) * N; o += 256) {
        const int rg = o / N, n = o - rg * N, r0 = 4 * rg;
        const float bv = b ? b[n] : 0.f;
        float acc[4] = {bv, bv, bv, bv};
        const float* wr = ws + n * KP;
        const float* xr = xs + r0 * K;
#pragma unroll 4
        for (int k = 0; k < K; ++k) {
            const float w = wr[k];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] += xr[j * K + k] * w;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) if (r0 + j < rows) y[(m0 + r0 + j) * y_stride + n] = apply_act(acc[j], act);
    }
}

__global__ __launch_bounds__(256) void sd_bwd_data_kernel(const float* __restrict__ dy, const float* __restrict__ W, float* __restrict__ dx, const float* __restrict__ y_act,
                                                          int act, const float* __restrict__ x_in, int in_act, int64_t M, int K, int N, int64_t dy_stride,
                                                          int64_t dx_stride, int64_t y_stride, int64_t x_stride) {
    extern __shared__ float lds[];
    float* ws = lds;                                         // [N][K]: threads of a wave walk k, consecutive
    float* gs = ws + N * K;                                  // [SD_ROWS][N]
    const int64_t m0 = (int64_t)blockIdx.x * SD_ROWS;
    const int rows = (int)min((int64_t)SD_ROWS, M - m0);
    sd_load_g(gs, dy, y_act, act, m0, M, N, dy_stride, y_stride);
    sd_copy2d(ws, W, N, K, K, K);
    __syncthreads();
    // thread = (input column k, group of 4 rows); the in_act operands of its 4 outputs are requested before the dot products
    for (int o = threadIdx.x; o < (SD_ROWS / 4) * K; o += 256) {
        const int rg = o / K, k = o - rg * K, r0 = 4 * rg;
        float xin[4] = {1.f, 1.f, 1.f, 1.f};
        if (x_in) {
#pragma unroll
            for (int j = 0; j < 4; ++j) xin[j] = x_in[min(m0 + r0 + j, M - 1) * x_stride + k];
        }
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        const float* gr = gs + r0 * N;
#pragma unroll 4
        for (int n = 0; n < N; ++n) {
            const float w = ws[n * K + k];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] += gr[j * N + n] * w;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (r0 + j < rows) dx[(m0 + r0 + j) * dx_stride + k] = x_in ? acc[j] * act_grad_from_out(xin[j], in_act) : acc[j];
    }
}

// part[wg][n * K + k] = sum over the workgroup's rows of g[m][n] x[m][k]; part[wg][N K + n] = sum of g[m][n]
__global__ __launch_bounds__(256) void sd_bwd_weight_partial_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ part,
                                                                    const float* __restrict__ y_act, int act, int64_t M, int K, int N, int64_t dy_stride,
                                                                    int64_t x_stride, int64_t y_stride) {
    extern __shared__ float lds[];
    float* xs = lds;                                         // [SD_ROWS][K]
    float* gs = xs + SD_ROWS * K;                            // [SD_ROWS][N]
    const int64_t m0 = (int64_t)blockIdx.x * SD_ROWS;
    const int rows = (int)min((int64_t)SD_ROWS, M - m0);
    sd_copy2d(xs, x + m0 * x_stride, rows, K, K, x_stride);
    for (int i = threadIdx.x + rows * K; i < SD_ROWS * K; i += 256) xs[i] = 0.f;      // (rows beyond M: g is zero there, keep x finite)
    sd_load_g(gs, dy, y_act, act, m0, M, N, dy_stride, y_stride);
    __syncthreads();
    float* out = part + (size_t)blockIdx.x * ((size_t)N * K + N);
    // thread = (k, group of 4 output rows n): one x read feeds 4 FMAs
    const int ng = (N + 3) / 4;
    for (int o = threadIdx.x; o < ng * K; o += 256) {
        const int g4 = o / K, k = o - g4 * K, n0 = 4 * g4;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < SD_ROWS; ++r) {
            const float xv = xs[r * K + k];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] += gs[r * N + min(n0 + j, N - 1)] * xv;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) if (n0 + j < N) out[(size_t)(n0 + j) * K + k] = acc[j];
    }
    for (int n = threadIdx.x; n < N; n += 256) {
        float acc = 0.f;
#pragma unroll
        for (int r = 0; r < SD_ROWS; ++r) acc += gs[r * N + n];
        out[(size_t)N * K + n] = acc;
    }
}

// dW[i] = sum_wg part[wg][i] (i < N K), db[n] = sum_wg part[wg][N K + n], wg in index order; 8 partials in flight per thread
__global__ __launch_bounds__(256) void sd_bwd_weight_finish_kernel(const float* __restrict__ part, float* __restrict__ dW, float* __restrict__ db, int nwg, int NK, int N) {
    const int i = blockIdx.x * 256 + threadIdx.x, tot = NK + N;
    if (i >= tot) return;
    float acc = 0.f;
    constexpr int U = 8;
    for (int g0 = 0; g0 < nwg; g0 += U) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = part[(size_t)min(g0 + u, nwg - 1) * tot + i];
#pragma unroll
        for (int u = 0; u < U; ++u) if (g0 + u < nwg) acc += v[u];
    }
    if (i < NK) dW[i] = acc;
    else if (db) db[i - NK] = acc;
}

bool sd_shape_ok(int64_t M, int64_t K, int64_t N) { return M > 16 && K >= 1 && N >= 1 && K <= SD_MAX_DIM && N <= SD_MAX_DIM && N * K <= SD_MAX_NK && M < ((int64_t)1 << 30); }
constexpr int SD_LDS_MAX = 112 * 1024;                   // 48 KB of weights + 16 rows of the two activations (+ pitch padding)
template <typename KERN> int sd_allow_lds(KERN kern, bool* done) {
    if (!*done) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SD_LDS_MAX) != hipSuccess) return CVAE_E_LAUNCH;
        *done = true;
    }
    return CVAE_OK;
}
bool sd_act_ok(int act) { return act >= CVAE_ACT_NONE && act <= CVAE_ACT_LEAKY02; }

}  // namespace

extern "C" int cvae_small_dense_supported(int64_t M, int64_t K, int64_t N) { return sd_shape_ok(M, K, N) ? 1 : 0; }

extern "C" size_t cvae_small_dense_workspace_bytes(int64_t M, int64_t K, int64_t N) {
    if (!sd_shape_ok(M, K, N)) return 0;
    return (size_t)((M + SD_ROWS - 1) / SD_ROWS) * (size_t)(N * K + N) * sizeof(float);
}

extern "C" int cvae_small_dense_fwd(const float* x, const float* W, const float* b, float* y, int64_t M, int64_t K, int64_t N, int64_t x_stride, int64_t y_stride, int act,
                                    void* stream) {
    if (!sd_shape_ok(M, K, N) || x_stride < K || y_stride < N || !sd_act_ok(act)) return CVAE_E_BADSHAPE;
    if (!x || !W || !y) return CVAE_E_NULLPTR;
    const size_t lds = sizeof(float) * ((size_t)N * (K | 1) + (size_t)SD_ROWS * K);
    static bool attr = false;
    if (sd_allow_lds(sd_fwd_kernel, &attr) != CVAE_OK) return CVAE_E_LAUNCH;
    hipLaunchKernelGGL(sd_fwd_kernel, dim3((unsigned)((M + SD_ROWS - 1) / SD_ROWS)), dim3(256), lds, (hipStream_t)stream, x, W, b, y, M, (int)K, (int)N, x_stride, y_stride, act);
    CVAE_CHECK_LAUNCH();
    return CVAE_OK;
}

extern "C" int cvae_small_dense_bwd_data(const float* dy, const float* W, float* dx, const float* y_act, int act, const float* x_in, int in_act, int64_t M, int64_t K,
                                         int64_t N, int64_t dy_stride, int64_t dx_stride, int64_t y_stride, int64_t x_stride, void* stream) {
    if (!sd_shape_ok(M, K, N) || dy_stride < N || dx_stride < K || !sd_act_ok(act) || !sd_act_ok(in_act)) return CVAE_E_BADSHAPE;
    if (!dy || !W || !dx) return CVAE_E_NULLPTR;
    if (act == CVAE_ACT_NONE) y_act = nullptr;
    if (in_act == CVAE_ACT_NONE) x_in = nullptr;
    if ((y_act && y_stride < N) || (x_in && x_stride < K)) return CVAE_E_BADSHAPE;
    const size_t lds = sizeof(float) * ((size_t)N * K + (size_t)SD_ROWS * N);
    static bool attr = false;
    if (sd_allow_lds(sd_bwd_data_kernel, &attr) != CVAE_OK) return CVAE_E_LAUNCH;
    hipLaunchKernelGGL(sd_bwd_data_kernel, dim3((unsigned)((M + SD_ROWS - 1) / SD_ROWS)), dim3(256), lds, (hipStream_t)stream, dy, W, dx, y_act, act, x_in, in_act, M, (int)K,
                       (int)N, dy_stride, dx_stride, y_stride, x_stride);
    CVAE_CHECK_LAUNCH();
    return CVAE_OK;
}

extern "C" int cvae_small_dense_bwd_weight(const float* dy, const float* x, float* dW, float* db, const float* y_act, int act, int64_t M, int64_t K, int64_t N,
                                           int64_t dy_stride, int64_t x_stride, int64_t y_stride, void* workspace, size_t workspace_bytes, void* stream) {
    if (!sd_shape_ok(M, K, N) || dy_stride < N || x_stride < K || !sd_act_ok(act)) return CVAE_E_BADSHAPE;
    if (!dy || !x || !dW || !workspace) return CVAE_E_NULLPTR;
    if (workspace_bytes < cvae_small_dense_workspace_bytes(M, K, N)) return CVAE_E_WORKSPACE;
    if (act == CVAE_ACT_NONE) y_act = nullptr;
    if (y_act && y_stride < N) return CVAE_E_BADSHAPE;
    const int nwg = (int)((M + SD_ROWS - 1) / SD_ROWS);
    const size_t lds = sizeof(float) * ((size_t)SD_ROWS * K + (size_t)SD_ROWS * N);
    hipLaunchKernelGGL(sd_bwd_weight_partial_kernel, dim3((unsigned)nwg), dim3(256), lds, (hipStream_t)stream, dy, x, (float*)workspace, y_act, act, M, (int)K, (int)N,
                       dy_stride, x_stride, y_stride);
    CVAE_CHECK_LAUNCH();
    const int tot = (int)(N * K + N);
    hipLaunchKernelGGL(sd_bwd_weight_finish_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, dW, db, nwg, (int)(N * K),
                       (int)N);
    CVAE_CHECK_LAUNCH();
    return CVAE_OK;
}
