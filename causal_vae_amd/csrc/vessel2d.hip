// vessel2d.hip — what CausalVesselVAE (vessel_analysis/00_core/models.py:9-166 of the reference) needs beyond the conv family:
//   * BatchNorm2d (train / eval) + activation on channels-last tensors, forward and backward;
//   * torch.clamp with its pass-through gradient mask;
//   * nn.Upsample(scale_factor=2, 'nearest') + nn.Conv2d(k3, s1, p1) expressed on the existing `up` kernels.
//
// Nearest x2 followed by a 3x3 / pad 1 cross-correlation is, per output parity, a 2-tap stencil on the low-resolution input:
//   out[2q]   = W3[0] S[q-1] + (W3[1] + W3[2]) S[q]         out[2q+1] = (W3[0] + W3[1]) S[q] + W3[2] S[q+1]        (per dimension)
// which is exactly the transposed k4/s2/p1 form L[l] = sum_{l = 2s - 1 + k} S[s] K4[k] with K4 = A W3, A = [[0,0,1],[0,1,1],[1,1,0],[1,0,0]]
// (zero padding of the conv == out-of-range S).  So the decoder runs on conv_up / conv_down / conv_wgrad unchanged with
// K4[cin][cout] = A W3[cout][cin] A^T, and dW3 = A^T dK4 A — two tiny kernels here.
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------ k3 <-> k4 weights
__constant__ const float A43[4][3] = {{0.f, 0.f, 1.f}, {0.f, 1.f, 1.f}, {1.f, 1.f, 0.f}, {1.f, 0.f, 0.f}};

__global__ __launch_bounds__(256) void conv3_to_k4_kernel(const float* __restrict__ w3, float* __restrict__ k4, int Cout, int Cin) {
    const int i = blockIdx.x * 256 + threadIdx.x;                     // (cin, cout) pair, k4 order
    if (i >= Cin * Cout) return;
    const int cin = i / Cout, cout = i - cin * Cout;
    float w[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) w[a][b] = w3[((size_t)cout * Cin + cin) * 9 + a * 3 + b];
#pragma unroll
    for (int kh = 0; kh < 4; ++kh)
#pragma unroll
        for (int kw = 0; kw < 4; ++kw) {
            float v = 0.f;
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) v += A43[kh][a] * A43[kw][b] * w[a][b];
            k4[(size_t)i * 16 + kh * 4 + kw] = v;
        }
}
__global__ __launch_bounds__(256) void k4_to_conv3_grad_kernel(const float* __restrict__ dk4, float* __restrict__ dw3, int Cout, int Cin) {
    const int i = blockIdx.x * 256 + threadIdx.x;                     // (cout, cin) pair, w3 order
    if (i >= Cin * Cout) return;
    const int cout = i / Cin, cin = i - cout * Cin;
    float g[4][4];
#pragma unroll
    for (int kh = 0; kh < 4; ++kh)
#pragma unroll
        for (int kw = 0; kw < 4; ++kw) g[kh][kw] = dk4[((size_t)cin * Cout + cout) * 16 + kh * 4 + kw];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            float v = 0.f;
#pragma unroll
            for (int kh = 0; kh < 4; ++kh)
#pragma unroll
                for (int kw = 0; kw < 4; ++kw) v += A43[kh][a] * A43[kw][b] * g[kh][kw];
            dw3[(size_t)i * 9 + a * 3 + b] = v;
        }
}

// ------------------------------------------------------------------------------------------------ clamp
__global__ void clamp_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, float lo, float hi, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) y[i] = fminf(fmaxf(x[i], lo), hi);
}
// torch.clamp backward: the gradient passes where lo <= x <= hi
__global__ void clamp_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g, float* __restrict__ dx, float lo, float hi, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dx[i] = (x[i] >= lo && x[i] <= hi) ? g[i] : 0.f;
}

// ------------------------------------------------------------------------------------------------ BatchNorm2d, channels-last [P][C]
// A lane owns 8 consecutive channels (one 16-byte bf16 / two 16-byte fp32 loads per position); a 256-thread block covers
// R = 256 / (C / 8) positions per pass.  Partial sums are combined in LDS and leave as one plain store per channel and block into
// row blockIdx.x of the caller's scratch ([gb][C] per output, gb <= 1024); the finalize launches add the rows in index order
// (no float atomics: bit-reproducible statistics and gradients).
template <typename T> __device__ __forceinline__ void load8(const T* p, float (&v)[8]) {
    __attribute__((aligned(16))) T raw[8];
    constexpr int NU = (8 * sizeof(T)) / 16;
#pragma unroll
    for (int u = 0; u < NU; ++u) ((uint4*)raw)[u] = ((const uint4*)p)[u];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = to_f32(raw[e]);
}
template <typename T> __device__ __forceinline__ void store8(T* p, const float (&v)[8]) {
    __attribute__((aligned(16))) T raw[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) raw[e] = from_f32<T>(v[e]);
    constexpr int NU = (8 * sizeof(T)) / 16;
#pragma unroll
    for (int u = 0; u < NU; ++u) ((uint4*)p)[u] = ((const uint4*)raw)[u];
}

// MODE 0: o0[blk] = sum x.   MODE 1: o0[blk] = sum (x - mean)^2.
// MODE 2: with g = dy * act'(y): o0[blk] = sum g, o1[blk] = sum g * (x - mean) * rstd     (dbeta, dgamma)
template <typename T, int MODE>
__global__ __launch_bounds__(256) void bn2d_reduce_kernel(const T* __restrict__ x, const T* __restrict__ dy, const T* __restrict__ y,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd, float* __restrict__ o0,
                                                          float* __restrict__ o1, int64_t P, int C, int act) {
    __shared__ float red[2][256][8 + 1];
    const int lanes = C >> 3, R = 256 / lanes, t = threadIdx.x;
    const int cl = t % lanes, rg = t / lanes, c0 = cl * 8;
    float a0[8], a1[8], mu[8], rs[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { a0[e] = 0.f; a1[e] = 0.f; mu[e] = (MODE >= 1) ? mean[c0 + e] : 0.f; rs[e] = (MODE == 2) ? rstd[c0 + e] : 0.f; }
    if (rg < R)
        for (int64_t p = (int64_t)blockIdx.x * R + rg; p < P; p += (int64_t)gridDim.x * R) {
            float xv[8];
            load8<T>(x + p * C + c0, xv);
            if (MODE == 0) {
#pragma unroll
                for (int e = 0; e < 8; ++e) a0[e] += xv[e];
            } else if (MODE == 1) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float dv = xv[e] - mu[e]; a0[e] += dv * dv; }
            } else {
                float gv[8], yv[8];
                load8<T>(dy + p * C + c0, gv);
                load8<T>(y + p * C + c0, yv);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float g = gv[e] * act_grad_from_out(yv[e], act);
                    a0[e] += g; a1[e] += g * (xv[e] - mu[e]) * rs[e];
                }
            }
        }
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[0][t][e] = a0[e]; red[1][t][e] = a1[e]; }
    __syncthreads();
    // thread c < C sums its channel over the R row groups
    for (int c = t; c < C; c += 256) {
        const int l = c >> 3, e = c & 7;
        float s0 = 0.f, s1 = 0.f;
        for (int q = 0; q < R; ++q) { s0 += red[0][q * lanes + l][e]; s1 += red[1][q * lanes + l][e]; }
        o0[(size_t)blockIdx.x * C + c] = s0;
        if (MODE == 2) o1[(size_t)blockIdx.x * C + c] = s1;
    }
}

// y = act((x - mean) * rstd * gamma + beta); block 0 also turns the two sums into mean / rstd bookkeeping when `finalize` is set:
// mean = sum / P (MODE stats pass 1), rstd = 1 / sqrt(sqdev / P + eps), running stats with the unbiased variance.
__device__ __forceinline__ float bn2d_row_sum(const float* __restrict__ part, int rows, int C, int c) {
    float s = 0.f;
    for (int r = 0; r < rows; ++r) s += part[(size_t)r * C + c];
    return s;
}
__global__ void bn2d_finalize_mean_kernel(const float* __restrict__ part, int rows, float* __restrict__ mean, int C, float invP) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < C) mean[c] = bn2d_row_sum(part, rows, C, c) * invP;
}
__global__ void bn2d_finalize_rstd_kernel(const float* __restrict__ part, int rows, const float* __restrict__ mean, float* __restrict__ rstd,
                                          float* __restrict__ running_mean, float* __restrict__ running_var, int C, float P, float eps, float momentum) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float var = bn2d_row_sum(part, rows, C, c) / P;
    if (running_mean) {
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean[c];
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * var * (P / (P - 1.f));
    }
    rstd[c] = 1.f / sqrtf(var + eps);
}
__global__ void bn2d_finalize_grads_kernel(const float* __restrict__ part_b, const float* __restrict__ part_g, int rows, float* __restrict__ dbeta,
                                           float* __restrict__ dgamma, int C) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    dbeta[c] = bn2d_row_sum(part_b, rows, C, c);
    dgamma[c] = bn2d_row_sum(part_g, rows, C, c);
}
template <typename T>
__global__ __launch_bounds__(256) void bn2d_apply_kernel(const T* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta, T* __restrict__ y, int64_t P, int C, int act) {
    const int lanes = C >> 3;
    const int64_t n8 = P * lanes;
    for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const int c0 = (int)(i % lanes) * 8;
        float v[8];
        load8<T>(x + i * 8, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = apply_act((v[e] - mean[c0 + e]) * rstd[c0 + e] * gamma[c0 + e] + beta[c0 + e], act);
        store8<T>(y + i * 8, v);
    }
}
// dx = gamma * rstd / P * (P g - dbeta - xhat dgamma),  g = dy * act'(y)
template <typename T>
__global__ __launch_bounds__(256) void bn2d_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ dy, const T* __restrict__ y,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                             const float* __restrict__ dgamma, const float* __restrict__ dbeta, T* __restrict__ dx,
                                                             int64_t P, int C, int act) {
    const int lanes = C >> 3;
    const int64_t n8 = P * lanes;
    const float fP = (float)P;
    for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const int c0 = (int)(i % lanes) * 8;
        float xv[8], gv[8], yv[8], o[8];
        load8<T>(x + i * 8, xv);
        load8<T>(dy + i * 8, gv);
        load8<T>(y + i * 8, yv);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = c0 + e;
            const float g = gv[e] * act_grad_from_out(yv[e], act), xh = (xv[e] - mean[c]) * rstd[c];
            o[e] = gamma[c] * rstd[c] / fP * (fP * g - dbeta[c] - xh * dgamma[c]);
        }
        store8<T>(dx + i * 8, o);
    }
}

static int64_t bn2d_blocks(int64_t P, int C) {
    const int lanes = C >> 3, R = 256 / lanes;
    int64_t gb = (P + (int64_t)R * 16 - 1) / ((int64_t)R * 16);
    if (gb > 1024) gb = 1024;
    if (gb < 1) gb = 1;
    return gb;
}
template <typename T>
int bn2d_fwd_t(const T* x, const float* gamma, const float* beta, T* y, float* mean, float* rstd, float* running_mean, float* running_var, int64_t P, int C,
               float momentum, float eps, int training, int act, float* ws, hipStream_t st) {
    const int lanes = C >> 3;
    const int64_t gb = bn2d_blocks(P, C);
    if (training) {
        hipLaunchKernelGGL((bn2d_reduce_kernel<T, 0>), dim3((unsigned)gb), dim3(256), 0, st, x, (const T*)nullptr, (const T*)nullptr, (const float*)nullptr,
                           (const float*)nullptr, ws, (float*)nullptr, P, C, 0);
        hipLaunchKernelGGL(bn2d_finalize_mean_kernel, dim3((C + 255) / 256), dim3(256), 0, st, (const float*)ws, (int)gb, mean, C, 1.f / (float)P);
        hipLaunchKernelGGL((bn2d_reduce_kernel<T, 1>), dim3((unsigned)gb), dim3(256), 0, st, x, (const T*)nullptr, (const T*)nullptr, (const float*)mean,
                           (const float*)nullptr, ws, (float*)nullptr, P, C, 0);
        hipLaunchKernelGGL(bn2d_finalize_rstd_kernel, dim3((C + 255) / 256), dim3(256), 0, st, (const float*)ws, (int)gb, (const float*)mean, rstd, running_mean,
                           running_var, C, (float)P, eps, momentum);
    }
    hipLaunchKernelGGL(bn2d_apply_kernel<T>, dim3(cvae_grid_1d(P * lanes, 256, 8192)), dim3(256), 0, st, x, (const float*)mean, (const float*)rstd, gamma, beta, y, P, C,
                       act);
    CVAE_CHECK_LAUNCH();
    return CVAE_OK;
}
template <typename T>
int bn2d_bwd_t(const T* x, const T* dy, const T* y, const float* gamma, const float* mean, const float* rstd, T* dx, float* dgamma, float* dbeta, int64_t P, int C,
               int act, float* ws, hipStream_t st) {
    const int lanes = C >> 3;
    const int64_t gb = bn2d_blocks(P, C);
    float* part_b = ws;
    float* part_g = ws + (size_t)gb * C;
    hipLaunchKernelGGL((bn2d_reduce_kernel<T, 2>), dim3((unsigned)gb), dim3(256), 0, st, x, dy, y, mean, rstd, part_b, part_g, P, C, act);
    hipLaunchKernelGGL(bn2d_finalize_grads_kernel, dim3((C + 255) / 256), dim3(256), 0, st, (const float*)part_b, (const float*)part_g, (int)gb, dbeta, dgamma, C);
    hipLaunchKernelGGL(bn2d_bwd_apply_kernel<T>, dim3(cvae_grid_1d(P * lanes, 256, 8192)), dim3(256), 0, st, x, dy, y, mean, rstd, gamma, (const float*)dgamma,
                       (const float*)dbeta, dx, P, C, act);
    CVAE_CHECK_LAUNCH();
    return CVAE_OK;
}

}  // namespace

extern "C" int cvae_conv3_to_k4(const float* w3, float* k4, int64_t Cout, int64_t Cin, void* stream) {
    if (Cout <= 0 || Cin <= 0 || Cout * Cin > ((int64_t)1 << 28)) return CVAE_E_BADSHAPE;
    if (!w3 || !k4) return CVAE_E_NULLPTR;
    hipLaunchKernelGGL(conv3_to_k4_kernel, dim3((unsigned)((Cout * Cin + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w3, k4, (int)Cout, (int)Cin);
    CVAE_CHECK_LAUNCH();
    return CVAE_OK;
}
extern "C" int cvae_k4_to_conv3_grad(const float* dk4, float* dw3, int64_t Cout, int64_t Cin, void* stream) {
    if (Cout <= 0 || Cin <= 0 || Cout * Cin > ((int64_t)1 << 28)) return CVAE_E_BADSHAPE;
    if (!dk4 || !dw3) return CVAE_E_NULLPTR;
    hipLaunchKernelGGL(k4_to_conv3_grad_kernel, dim3((unsigned)((Cout * Cin + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dk4, dw3, (int)Cout, (int)Cin);
    CVAE_CHECK_LAUNCH();
    return CVAE_OK;
}
extern "C" int cvae_clamp_fwd(const float* x, float* y, float lo, float hi, int64_t n, void* stream) {
    if (n < 0) return CVAE_E_BADSHAPE;
    if (n == 0) return CVAE_OK;
    if (!x || !y) return CVAE_E_NULLPTR;
    hipLaunchKernelGGL(clamp_fwd_kernel, dim3(cvae_grid_1d(n, 256)), dim3(256), 0, (hipStream_t)stream, x, y, lo, hi, n);
    CVAE_CHECK_LAUNCH();
    return CVAE_OK;
}
extern "C" int cvae_clamp_bwd(const float* x, const float* g, float* dx, float lo, float hi, int64_t n, void* stream) {
    if (n < 0) return CVAE_E_BADSHAPE;
    if (n == 0) return CVAE_OK;
    if (!x || !g || !dx) return CVAE_E_NULLPTR;
    hipLaunchKernelGGL(clamp_bwd_kernel, dim3(cvae_grid_1d(n, 256)), dim3(256), 0, (hipStream_t)stream, x, g, dx, lo, hi, n);
    CVAE_CHECK_LAUNCH();
    return CVAE_OK;
}
extern "C" size_t cvae_bn2d_workspace_bytes(int64_t P, int64_t C) {
    if (P <= 0 || C < 8 || C % 8 || C > 2048) return 0;
    return (size_t)2 * bn2d_blocks(P, (int)C) * C * sizeof(float);
}
extern "C" int cvae_bn2d_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, float* running_mean, float* running_var,
                             int64_t P, int64_t C, float momentum, float eps, int training, int act, int dtype, void* workspace, size_t workspace_bytes, void* stream) {
    if (P <= 0 || C < 8 || C % 8 || C > 2048 || (training && P < 2)) return CVAE_E_BADSHAPE;
    if (!x || !gamma || !beta || !y || !mean || !rstd) return CVAE_E_NULLPTR;
    if (training) {
        if (!workspace) return CVAE_E_NULLPTR;
        if (workspace_bytes < cvae_bn2d_workspace_bytes(P, C) / 2) return CVAE_E_WORKSPACE;
    }
    float* ws = (float*)workspace;
    if (dtype == CVAE_BF16) return bn2d_fwd_t<bf16>((const bf16*)x, gamma, beta, (bf16*)y, mean, rstd, running_mean, running_var, P, (int)C, momentum, eps, training, act, ws, (hipStream_t)stream);
    if (dtype == CVAE_F32) return bn2d_fwd_t<float>((const float*)x, gamma, beta, (float*)y, mean, rstd, running_mean, running_var, P, (int)C, momentum, eps, training, act, ws, (hipStream_t)stream);
    return CVAE_E_DTYPE;
}
extern "C" int cvae_bn2d_bwd(const void* x, const void* dy, const void* y, const float* gamma, const float* mean, const float* rstd, void* dx, float* dgamma,
                             float* dbeta, int64_t P, int64_t C, int act, int dtype, void* workspace, size_t workspace_bytes, void* stream) {
    if (P <= 0 || C < 8 || C % 8 || C > 2048) return CVAE_E_BADSHAPE;
    if (!x || !dy || !y || !gamma || !mean || !rstd || !dx || !dgamma || !dbeta || !workspace) return CVAE_E_NULLPTR;
    if (workspace_bytes < cvae_bn2d_workspace_bytes(P, C)) return CVAE_E_WORKSPACE;
    float* ws = (float*)workspace;
    if (dtype == CVAE_BF16) return bn2d_bwd_t<bf16>((const bf16*)x, (const bf16*)dy, (const bf16*)y, gamma, mean, rstd, (bf16*)dx, dgamma, dbeta, P, (int)C, act, ws, (hipStream_t)stream);
    if (dtype == CVAE_F32) return bn2d_bwd_t<float>((const float*)x, (const float*)dy, (const float*)y, gamma, mean, rstd, (float*)dx, dgamma, dbeta, P, (int)C, act, ws, (hipStream_t)stream);
    return CVAE_E_DTYPE;
}
