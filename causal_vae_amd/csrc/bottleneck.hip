// bottleneck.hip — the dense middle of CausalBioVAE (causal_cascade/models.py:57-79 of the reference) for small batches.
//
// Between the last encoder conv and the first decoder conv the model is ~25 tiny fp32 ops on a batch of <= 16 rows
// (pool, concat, enc_fc x2, fc_mu / fc_logvar, reparameterize, mechanism_net with BatchNorm1d, concat, dec_input) and as
// many again in backward.  Only two of them move real data: enc_fc.0 (a 512 x 16415 weight, 33.6 MB) and dec_input
// (16384 x 76, 5 MB).  As single launches each op pays ~5 us of launch / dependency latency for < 1 us of work, and the two big
// ones ran at < 2 TB/s.  Here every dependency LEVEL of that graph is one wide launch (a single workgroup walking the whole chain
// was tried first and is latency-bound: ~200 us), five forward and five backward:
//
//   forward   pool_cat_fwd        avg-pool + flatten + concat [feat, m, t]                                 -> xcat [M][K1]
//             skinny_fwd_partial  enc_fc.0 as a split-K weight stream -> partial [KS][M][N1]; one extra block runs the whole
//                                 mechanism_net (Linear, BatchNorm1d batch statistics + running stats, ReLU, Linear, ReLU, Linear)
//             fc2_fwd             bias + ReLU of the partials, enc_fc.2 + ReLU (one wave per row)           -> h1, h2
//             mulv_fwd            fc_mu, fc_logvar, reparameterize (one wave per latent)                    -> mu, logvar, zm = [z, m_hat]
//             dec_input_fwd       dec_input + bias, written channels-last in the conv dtype                -> [M][S][C]
//   backward  dec_input_bwd       dW, db of dec_input and per-cell partials of d(zm)
//             mulv_bwd            reparameterize / fc_mu / fc_logvar backward by weight columns             -> dh2
//             fc2_bwd             enc_fc.2 backward by weight columns                                       -> g1 = d(enc_fc.0 output), db1
//             skinny_bwd_colwise  enc_fc.0: dW (33.6 MB written) and the partial d(xcat) in one pass over W; one extra block runs
//                                 mechanism_net's backward (BatchNorm1d included)
//             pool_bwd            sums the partials, spreads them over the pooling windows, applies the ReLU mask
//
// All arithmetic is fp32 (as in the unfused path); M <= 16.
#include "common.h"

#define BN_MAXM 16

namespace {

struct TailDims { int M, N1, N2, Z, T, HM, DM, KS, P, NZ; }; // N1 = 512, N2 = 256, Z = latent, T = t_dim, HM = 64, DM = m_dim, NZ = d(zm) partial slots
struct TailParams {
    const float *b1, *W2, *b2, *Wmu, *bmu, *Wlv, *blv, *Wm0, *bm0, *gamma, *beta, *Wm3, *bm3, *Wm5, *bm5;
};
struct TailGrads {
    float *db1, *dW2, *db2, *dWmu, *dbmu, *dWlv, *dblv, *dWm0, *dbm0, *dgamma, *dbeta, *dWm3, *dbm3, *dWm5, *dbm5;
};
// saved activations (global), all [M][dim] row-major
struct TailSaved { float *h1, *h2, *mu, *logvar, *xhat, *invstd, *a1n, *a2, *m_hat, *zm; };
struct MechFwdArgs { const float* t_onehot; float *running_mean, *running_var; long long* num_batches_tracked; float momentum, bn_eps; int bn_training;
                     const float* sync_stats; int sync_ranks; };       // SyncBatchNorm: [ranks][2][HM] (sum, squared deviations from the rank's own mean) of every rank's batch
struct MechBwdArgs { const float *dzm_part, *g_mhat, *t_onehot; float *sync_dy, *sync_local; };      // SyncBatchNorm: the backward stops at the BatchNorm (see mech_bwd)
struct __attribute__((packed, aligned(4))) F4U { float x, y, z, w; };       // float4 at dword alignment
// streaming (nontemporal) forms of the dword-aligned 16-byte / 8-byte accesses, for the two passes over enc_fc.0's weight (-DCVAE_BN_NT=1 / 2 / 3: loads / stores / both)
#ifndef CVAE_BN_NT
#define CVAE_BN_NT 0
#endif
typedef float v4f_u __attribute__((ext_vector_type(4), aligned(4)));
typedef float v2f_u __attribute__((ext_vector_type(2), aligned(4)));
struct F2U;
template <typename V> __device__ __forceinline__ V ld_stream(const float* p) {
    if constexpr (CVAE_BN_NT & 1) {
        if constexpr (sizeof(V) == 16) { const v4f_u v = __builtin_nontemporal_load((const v4f_u*)p); V r; __builtin_memcpy(&r, &v, 16); return r; }
        else { const v2f_u v = __builtin_nontemporal_load((const v2f_u*)p); V r; __builtin_memcpy(&r, &v, 8); return r; }
    } else return *(const V*)p;
}
template <typename V> __device__ __forceinline__ void st_stream(float* p, const V& val) {
    if constexpr (CVAE_BN_NT & 2) {
        if constexpr (sizeof(V) == 16) { v4f_u v; __builtin_memcpy(&v, &val, 16); __builtin_nontemporal_store(v, (v4f_u*)p); }
        else { v2f_u v; __builtin_memcpy(&v, &val, 8); __builtin_nontemporal_store(v, (v2f_u*)p); }
    } else *(V*)p = val;
}

// global -> LDS copies of the level kernels: ALL of a thread's loads are issued before the first LDS store (U per pass, predicated on a clamped address).  The plain
// loop `for (i = tid; i < n; i += 256) dst[i] = src[i]` compiles to load / wait / store per iteration — a dependent global round trip (~1.5 us) per 256 elements, which
// is what these few-microsecond kernels then consist of.
template <int U>
__device__ __forceinline__ void copy_g2l(float* __restrict__ dst, const float* __restrict__ src, int n) {
    for (int base = threadIdx.x; base < n; base += 256 * U) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = src[min(base + 256 * u, n - 1)];
#pragma unroll
        for (int u = 0; u < U; ++u) if (base + 256 * u < n) dst[base + 256 * u] = v[u];
    }
}
// the same for the 64 consecutive weight rows of a dec_input block: n = 64 * K4 floats (K4 % 4 == 0) as float4 runs into rows of pitch KP
__device__ __forceinline__ void copy_rows_g2l(float* __restrict__ ws, const float* __restrict__ wsrc, int n, int K4, int KP) {
    constexpr int U = 8;                                     // 64 * 128 / 4 / 256: one pass up to K4 = 128
    for (int base = threadIdx.x * 4; base < n; base += 1024 * U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = *(const float4*)(wsrc + min(base + 1024 * u, n - 4));
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = base + 1024 * u;
            if (i < n) {
                const int r = i / K4, k = i - r * K4;
                float* d = ws + r * KP + k;
                d[0] = v[u].x; d[1] = v[u].y; d[2] = v[u].z; d[3] = v[u].w;
            }
        }
    }
}

__device__ __forceinline__ int pool_lo(int o, int in, int out) { return (o * in) / out; }
__device__ __forceinline__ int pool_hi(int o, int in, int out) { return ((o + 1) * in + out - 1) / out; }

// ------------------------------------------------------------------------------------------------ pool_cat_fwd
// grid (S + 1, M): block (s, b) averages window s of sample b for all C channels (coalesced along c) and writes
// xcat[b][c * S + s]; block (S, b) copies m and t behind the features.  Block (S, 0) can also draw the step's reparameterisation noise
// (cvae_philox_normal_advance's numbers and call-counter bump, without its launch).
struct NoiseDraw { float* eps; int n; uint64_t seed, subseq; int* counter; };
template <typename T>
__global__ __launch_bounds__(256) void pool_cat_fwd_kernel(const T* __restrict__ y, const float* __restrict__ m, float* __restrict__ t,
                                                           const long long* __restrict__ t_labels, float* __restrict__ xcat, int D, int H, int W, int C,
                                                           int OD, int OH, int OW, int m_dim, int t_dim, int K1, NoiseDraw nz) {
    const int S = OD * OH * OW, b = blockIdx.y, s = blockIdx.x;
    float* row = xcat + (size_t)b * K1;
    if (s == S) {
        if (nz.counter && b == 0) {                          // uniform over the block
            const uint64_t offset = ((uint64_t)(unsigned)(*nz.counter)) << 24;
            __syncthreads();                                 // every thread has read the counter
            if (threadIdx.x == 0) *nz.counter += 1;
            for (int i = threadIdx.x; i < (nz.n + 3) / 4; i += 256) {
                float v[4];
                philox_normal4(offset + (uint64_t)i, nz.subseq, nz.seed, v);
                for (int j = 0; j < 4; ++j) if (i * 4 + j < nz.n) nz.eps[i * 4 + j] = v[j];
            }
        }
        if (t_labels) {                                      // F.one_hot(t).float() made here: t[b][.] is an output of this block
            const long long lab = t_labels[b];
            for (int i = threadIdx.x; i < t_dim; i += 256) { const float v = (i == lab) ? 1.f : 0.f; t[b * t_dim + i] = v; row[C * S + m_dim + i] = v; }
            for (int i = threadIdx.x; i < m_dim; i += 256) row[C * S + i] = m[b * m_dim + i];
            return;
        }
        for (int i = threadIdx.x; i < m_dim + t_dim; i += 256) row[C * S + i] = i < m_dim ? m[b * m_dim + i] : t[b * t_dim + i - m_dim];
        return;
    }
    const int ow = s % OW, oh = (s / OW) % OH, od = s / (OW * OH);
    const int d0 = pool_lo(od, D, OD), d1 = pool_hi(od, D, OD), h0 = pool_lo(oh, H, OH), h1 = pool_hi(oh, H, OH), w0 = pool_lo(ow, W, OW), w1 = pool_hi(ow, W, OW);
    const float inv = 1.f / (float)((d1 - d0) * (h1 - h0) * (w1 - w0));
    const int nh = h1 - h0, nw = w1 - w0, nvox = (d1 - d0) * nh * nw;
    for (int c = threadIdx.x; c < C; c += 256) {
        float acc = 0.f;
        constexpr int U = 8;                                 // window voxels in flight (clamped index, predicated add: the order of the sum is the nested loops')
        for (int v0 = 0; v0 < nvox; v0 += U) {
            float v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int vi = min(v0 + u, nvox - 1), w = vi % nw, h = (vi / nw) % nh, d = vi / (nw * nh);
                v[u] = to_f32(y[((((size_t)b * D + d0 + d) * H + h0 + h) * W + w0 + w) * C + c]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) if (v0 + u < nvox) acc += v[u];
        }
        row[c * S + s] = acc * inv;
    }
}

#ifndef CVAE_BN_FWD_UNROLL
#define CVAE_BN_FWD_UNROLL 2
#endif
#ifndef CVAE_BN_BWD_UNROLL
#define CVAE_BN_BWD_UNROLL 8
#endif
// ------------------------------------------------------------------------------------------------ skinny_fwd_partial
// partial[ks][m][n] = sum_{k in slice ks} x[m][k] W[n][k].  grid (N / 4, KS), 256 threads: 4 weight rows per workgroup share one
// pass over the x slice; rows are read with coalesced 4-byte loads (K is odd in the model: rows are not 16-byte aligned).
__device__ void mech_fwd(const TailDims& d, const TailParams& p, const TailSaved& sv, const float* __restrict__ t_onehot, float* __restrict__ running_mean,
                         float* __restrict__ running_var, long long* __restrict__ num_batches_tracked, float momentum, float bn_eps, int bn_training, float* lds,
                         const float* __restrict__ sync_stats, int sync_ranks);
__device__ void mech_bwd(const TailDims& d, const TailParams& p, const TailGrads& gr, const TailSaved& sv, const float* __restrict__ dzm_part,
                         const float* __restrict__ g_mhat, const float* __restrict__ t_onehot, float* lds, float* __restrict__ sync_dy, float* __restrict__ sync_local);

// The launch carries one extra block row (blockIdx.y == KS): its block 0 runs mechanism_net's forward, which depends only on t and
// is hidden behind the weight stream.
template <int MT>
__global__ __launch_bounds__(256) void skinny_fwd_partial_kernel(const float* __restrict__ x, const float* __restrict__ Wt, float* __restrict__ partial,
                                                                 int M, int K, int N, int kslice, int KS, TailDims d, TailParams tp, TailSaved sv, MechFwdArgs ma) {
    extern __shared__ float dyn_lds[];
    if ((int)blockIdx.y == KS) {
        if (blockIdx.x == 0) mech_fwd(d, tp, sv, ma.t_onehot, ma.running_mean, ma.running_var, ma.num_batches_tracked, ma.momentum, ma.bn_eps, ma.bn_training, dyn_lds, ma.sync_stats, ma.sync_ranks);
        return;
    }
    constexpr int R = 4;
    const int n0 = blockIdx.x * R, ks = blockIdx.y;
    const int k0 = ks * kslice, k1 = min(K, k0 + kslice);
    float acc[R][MT];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int mm = 0; mm < MT; ++mm) acc[r][mm] = 0.f;
    const float* wr[R];
#pragma unroll
    for (int r = 0; r < R; ++r) wr[r] = Wt + (size_t)min(n0 + r, N - 1) * K;
    // 16-byte loads at dword alignment (rows of odd length are not 16-byte aligned); the < 4-element tail of the slice goes scalar
    const int kvec = k0 + ((k1 - k0) & ~3);
#pragma unroll CVAE_BN_FWD_UNROLL
    for (int k = k0 + 4 * threadIdx.x; k < kvec; k += 1024) {
        F4U xv[MT], wv[R];
#pragma unroll
        for (int mm = 0; mm < MT; ++mm) xv[mm] = mm < M ? *(const F4U*)(x + (size_t)mm * K + k) : F4U{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < R; ++r) wv[r] = ld_stream<F4U>(wr[r] + k);
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int mm = 0; mm < MT; ++mm) acc[r][mm] += wv[r].x * xv[mm].x + wv[r].y * xv[mm].y + wv[r].z * xv[mm].z + wv[r].w * xv[mm].w;
    }
    for (int k = kvec + threadIdx.x; k < k1; k += 256) {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int mm = 0; mm < MT; ++mm)
                if (mm < M) acc[r][mm] += wr[r][k] * x[(size_t)mm * K + k];
    }
    __shared__ float red[4][R * MT];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int mm = 0; mm < MT; ++mm) {
            const float sum = wave_sum(acc[r][mm]);
            if (lane == 0) red[wave][r * MT + mm] = sum;
        }
    __syncthreads();
    if (threadIdx.x < R * MT) {
        const int r = threadIdx.x / MT, mm = threadIdx.x % MT;
        if (mm < M && n0 + r < N)
            partial[((size_t)ks * M + mm) * N + n0 + r] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    }
}

// ------------------------------------------------------------------------------------------------ single-workgroup helpers
// All take LDS operands laid out [M][dim] and are called by every thread of the (1024-thread) workgroup; the caller places
// __syncthreads() between dependent stages.

// ys[m][n] = act(sum_k xs[m][k] W[n][k] + b[n]); also stored to `yg` (global, row stride N) when non-null.  One wave per row.
__device__ void wg_linear_fwd(const float* __restrict__ Wt, const float* __restrict__ bias, const float* xs, float* ys, float* __restrict__ yg,
                              int M, int K, int N, int act) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int n = wave; n < N; n += nw) {
        float acc[BN_MAXM];
#pragma unroll
        for (int m = 0; m < BN_MAXM; ++m) acc[m] = 0.f;
        for (int k = lane; k < K; k += 64) {
            const float w = Wt[(size_t)n * K + k];
#pragma unroll
            for (int m = 0; m < BN_MAXM; ++m)
                if (m < M) acc[m] += w * xs[m * K + k];
        }
#pragma unroll
        for (int m = 0; m < BN_MAXM; ++m) {
            if (m >= M) break;
            const float s = wave_sum(acc[m]);
            if (lane == 0) {
                const float v = apply_act(s + (bias ? bias[n] : 0.f), act);
                ys[m * N + n] = v;
                if (yg) yg[(size_t)m * N + n] = v;
            }
        }
    }
}

// dW[n][k] = sum_m gs[m][n] xs[m][k];  db[n] = sum_m gs[m][n];  dxs[m][k] = sum_n gs[m][n] W[n][k]  (dxs may be null).
// BatchNorm backward for one element: k (n dy - sum dy - xhat sum(dy xhat)), the one spelling both the rank-local and the SyncBatchNorm path use
__device__ inline float bn_bwd_dx(float k, float nb, float dy, float dbet, float xh, float dgam) { return k * fmaf(-xh, dgam, fmaf(nb, dy, -dbet)); }

__device__ void wg_linear_bwd(const float* __restrict__ Wt, float* __restrict__ dW, float* __restrict__ db, const float* gs, const float* xs, float* dxs,
                              int M, int K, int N) {
    const int T = blockDim.x;
    for (int i = threadIdx.x; i < N * K; i += T) {
        const int n = i / K, k = i - n * K;
        float acc = 0.f;
        for (int m = 0; m < M; ++m) acc += gs[m * N + n] * xs[m * K + k];
        dW[i] = acc;
    }
    if (db)
        for (int n = threadIdx.x; n < N; n += T) {
            float acc = 0.f;
            for (int m = 0; m < M; ++m) acc += gs[m * N + n];
            db[n] = acc;
        }
    if (dxs)
        for (int i = threadIdx.x; i < M * K; i += T) {         // thread (m, k): consecutive threads walk k, so W rows are read coalesced
            const int m = i / K, k = i - m * K;
            float acc = 0.f;
#pragma unroll 8
            for (int n = 0; n < N; ++n) acc += gs[m * N + n] * Wt[(size_t)n * K + k];
            dxs[i] = acc;
        }
}


// ------------------------------------------------------------------------------------------------ mechanism_net forward
// One 256-thread workgroup (an extra block of the enc_fc.0 launch, which hides it): Linear(t) -> BatchNorm1d (batch statistics,
// running-stat update) -> ReLU -> Linear -> ReLU -> Linear.  `lds` holds 4 * M * HM + M * T floats.
__device__ void mech_fwd(const TailDims& d, const TailParams& p, const TailSaved& sv, const float* __restrict__ t_onehot, float* __restrict__ running_mean,
                         float* __restrict__ running_var, long long* __restrict__ num_batches_tracked, float momentum, float bn_eps, int bn_training, float* lds,
                         const float* __restrict__ sync_stats, int sync_ranks) {
    const int M = d.M, T = blockDim.x, tid = threadIdx.x, HM = d.HM;
    float* ts = lds;                 // [M][T]
    float* a1s = ts + M * d.T;       // [M][HM]
    float* a2s = a1s + M * HM;       // [M][HM]
    for (int i = tid; i < M * d.T; i += T) ts[i] = t_onehot[i];
    __syncthreads();
    for (int i = tid; i < M * HM; i += T) {                  // mechanism_net.0: thread (m, j)
        const int m = i / HM, j = i - m * HM;
        float acc = p.bm0[j];
        for (int k = 0; k < d.T; ++k) acc += ts[m * d.T + k] * p.Wm0[j * d.T + k];
        a1s[i] = acc;
    }
    __syncthreads();
    for (int j = tid; j < HM; j += T) {                      // BatchNorm1d + ReLU
        float mean, var;
        if (bn_training) {
            float nb = (float)M;                             // samples behind the statistics
            if (sync_stats) {
                // the GLOBAL batch: every rank's (sum, squared deviations from its own mean) of M samples, gathered by the caller and combined
                // here in rank order (the same arithmetic on every rank: replicas stay bit-identical).  Chan's pairwise update, no E[x^2] - mean^2.
                nb = (float)M * (float)sync_ranks;
                mean = 0.f;
                for (int r = 0; r < sync_ranks; ++r) mean += sync_stats[(size_t)r * 2 * HM + j];
                mean /= nb;
                var = 0.f;
                for (int r = 0; r < sync_ranks; ++r) {
                    const float c = sync_stats[(size_t)r * 2 * HM + j] / (float)M - mean;
                    var += sync_stats[(size_t)r * 2 * HM + HM + j] + (float)M * c * c;
                }
                var /= nb;
            } else {
                mean = 0.f;
                for (int m = 0; m < M; ++m) mean += a1s[m * HM + j];
                mean /= (float)M;
                var = 0.f;
                for (int m = 0; m < M; ++m) { const float c = a1s[m * HM + j] - mean; var += c * c; }
                var /= (float)M;
            }
            if (running_mean) {
                running_mean[j] = (1.f - momentum) * running_mean[j] + momentum * mean;
                running_var[j] = (1.f - momentum) * running_var[j] + momentum * var * (nb / (nb - 1.f));
            }
        } else {
            mean = running_mean[j]; var = running_var[j];
        }
        const float is = 1.f / sqrtf(var + bn_eps);
        sv.invstd[j] = is;
        for (int m = 0; m < M; ++m) {
            const float xh = (a1s[m * HM + j] - mean) * is;
            sv.xhat[m * HM + j] = xh;
            const float v = fmaxf(xh * p.gamma[j] + p.beta[j], 0.f);
            a1s[m * HM + j] = v; sv.a1n[m * HM + j] = v;
        }
    }
    if (tid == 0 && bn_training && num_batches_tracked) *num_batches_tracked += 1;
    __syncthreads();
    for (int i = tid; i < M * HM; i += T) {                  // mechanism_net.3 + ReLU
        const int m = i / HM, j = i - m * HM;
        float acc = p.bm3[j];
#pragma unroll 8
        for (int k = 0; k < HM; ++k) acc += a1s[m * HM + k] * p.Wm3[j * HM + k];
        acc = fmaxf(acc, 0.f);
        a2s[i] = acc; sv.a2[i] = acc;
    }
    __syncthreads();
    const int K4 = d.Z + d.DM;
    for (int i = tid; i < M * d.DM; i += T) {                // mechanism_net.5 -> m_hat, zm[:, Z:]
        const int m = i / d.DM, j = i - m * d.DM;
        float acc = p.bm5[j];
#pragma unroll 8
        for (int k = 0; k < HM; ++k) acc += a2s[m * HM + k] * p.Wm5[j * HM + k];
        sv.m_hat[i] = acc;
        sv.zm[m * K4 + d.Z + j] = acc;
    }
}

// ------------------------------------------------------------------------------------------------ fc2_fwd
// h1 = relu(sum_ks partial + b1) (every workgroup rebuilds it in LDS from the L2-resident partials; block 0 also saves it),
// then one wave per row of enc_fc.2: h2[m][n] = relu(W2[n] . h1[m] + b2[n]).  grid N2 / 4, 256 threads.
__global__ __launch_bounds__(256) void fc2_fwd_kernel(TailDims d, TailParams p, TailSaved sv, const float* __restrict__ partial) {
    extern __shared__ float lds[];
    const int M = d.M, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* h1s = lds;                                        // [M][N1]
    // Dependent global round trips (~1.5 us each) are what these small kernels consist of: the first HO * 64 elements of this wave's
    // W2 row are requested before the h1 rebuild, so they travel together with the partial-sum loads.
    constexpr int HO = 8;
    const int n = blockIdx.x * 4 + wave;
    float wpre[HO];
#pragma unroll
    for (int j = 0; j < HO; ++j) wpre[j] = p.W2[(size_t)min(n, d.N2 - 1) * d.N1 + min(lane + 64 * j, d.N1 - 1)];
    // The struct's pointers carry no __restrict__, so a load placed after a global store waits for it: every loop below that both loads
    // and stores global memory batches its loads first (U iterations' worth in registers), then computes and stores.
    constexpr int U = 4;
    for (int base = tid; base < M * d.N1; base += 256 * U) {
        float v[U][8], bb[U];                                // KS <= 8 slabs per element
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ii = min(base + u * 256, M * d.N1 - 1);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) v[u][ks] = ks < d.KS ? partial[(size_t)ks * M * d.N1 + ii] : 0.f;
            bb[u] = p.b1[ii % d.N1];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = base + u * 256;
            if (i < M * d.N1) {
                float acc = bb[u];
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) acc += v[u][ks];
                acc = fmaxf(acc, 0.f);
                h1s[i] = acc;
                if (blockIdx.x == 0) sv.h1[i] = acc;
            }
        }
    }
    __syncthreads();
    if (n >= d.N2) return;
    float acc[BN_MAXM];
#pragma unroll
    for (int m = 0; m < BN_MAXM; ++m) acc[m] = 0.f;
#pragma unroll
    for (int j = 0; j < HO; ++j) {
        const int k = lane + 64 * j;
        if (k < d.N1) {
#pragma unroll
            for (int m = 0; m < BN_MAXM; ++m)
                if (m < M) acc[m] += wpre[j] * h1s[m * d.N1 + k];
        }
    }
    for (int k = lane + 64 * HO; k < d.N1; k += 64) {
        const float w = p.W2[(size_t)n * d.N1 + k];
#pragma unroll
        for (int m = 0; m < BN_MAXM; ++m)
            if (m < M) acc[m] += w * h1s[m * d.N1 + k];
    }
#pragma unroll
    for (int m = 0; m < BN_MAXM; ++m) {
        if (m >= M) break;
        const float sum = wave_sum(acc[m]);
        if (lane == 0) sv.h2[m * d.N2 + n] = fmaxf(sum + p.b2[n], 0.f);
    }
}

// ------------------------------------------------------------------------------------------------ mulv_fwd
// One wave per latent j: mu[:, j] = Wmu[j] . h2 + bmu[j], logvar likewise, z = mu + eps * exp(logvar / 2) -> zm[:, j].  grid Z / 4.
__global__ __launch_bounds__(256) void mulv_fwd_kernel(TailDims d, TailParams p, TailSaved sv, const float* __restrict__ eps) {
    extern __shared__ float lds[];
    const int M = d.M, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, K4 = d.Z + d.DM;
    float* h2s = lds;                                        // [M][N2]
    constexpr int HO = 4;                                    // this wave's first HO * 64 elements of both weight rows travel with the h2 loads
    const int j = blockIdx.x * 4 + wave;
    float wmp[HO], wlp[HO];
#pragma unroll
    for (int u = 0; u < HO; ++u) {
        const size_t o = (size_t)min(j, d.Z - 1) * d.N2 + min(lane + 64 * u, d.N2 - 1);
        wmp[u] = p.Wmu[o]; wlp[u] = p.Wlv[o];
    }
    copy_g2l<4>(h2s, sv.h2, M * d.N2);
    __syncthreads();
    if (j >= d.Z) return;
    float am[BN_MAXM], al[BN_MAXM];
#pragma unroll
    for (int m = 0; m < BN_MAXM; ++m) { am[m] = 0.f; al[m] = 0.f; }
#pragma unroll
    for (int u = 0; u < HO; ++u) {
        const int k = lane + 64 * u;
        if (k < d.N2) {
#pragma unroll
            for (int m = 0; m < BN_MAXM; ++m)
                if (m < M) { am[m] += wmp[u] * h2s[m * d.N2 + k]; al[m] += wlp[u] * h2s[m * d.N2 + k]; }
        }
    }
    for (int k = lane + 64 * HO; k < d.N2; k += 64) {
        const float wm = p.Wmu[(size_t)j * d.N2 + k], wl = p.Wlv[(size_t)j * d.N2 + k];
#pragma unroll
        for (int m = 0; m < BN_MAXM; ++m)
            if (m < M) { am[m] += wm * h2s[m * d.N2 + k]; al[m] += wl * h2s[m * d.N2 + k]; }
    }
#pragma unroll
    for (int m = 0; m < BN_MAXM; ++m) {
        if (m >= M) break;
        const float smu = wave_sum(am[m]), slv = wave_sum(al[m]);
        if (lane == 0) {
            const float mu = smu + p.bmu[j], lv = slv + p.blv[j];
            sv.mu[m * d.Z + j] = mu; sv.logvar[m * d.Z + j] = lv;
            sv.zm[m * K4 + j] = mu + eps[m * d.Z + j] * __expf(0.5f * lv);
        }
    }
}

// ------------------------------------------------------------------------------------------------ dec_input_fwd
// out[b][s][c] = sum_k zm[b][k] Wd[c * S + s][k] + bd[c * S + s].  grid C * S / 64, 256 threads: a block takes 64 CONSECUTIVE
// weight rows n = c * S + s (one contiguous run of 64 * K4 floats, float4 loads when K4 % 4 == 0) into LDS; thread (r, q) owns row r
// and a quarter of k.  The (small) output is what gets scattered instead of the weight reads.
template <typename T>
__global__ __launch_bounds__(256) void dec_input_fwd_kernel(const float* __restrict__ zm, const float* __restrict__ Wd, const float* __restrict__ bd,
                                                            T* __restrict__ out, int M, int K4, int S, int C) {
    extern __shared__ float lds[];
    const int KP = K4 | 1;                                   // odd row pitch: conflict-free column walks
    float* ws = lds;                                         // [64][KP]
    float* zs = ws + 64 * KP;                                // [M][K4]
    float* part = zs + M * K4;                               // [4][M][64]
    const int tid = threadIdx.x;
    const size_t row0 = (size_t)blockIdx.x * 64;
    const float* wsrc = Wd + row0 * K4;
    if ((K4 & 3) == 0) copy_rows_g2l(ws, wsrc, 64 * K4, K4, KP);
    else {
        for (int i = tid; i < 64 * K4; i += 256) { const int r = i / K4, k = i - r * K4; ws[r * KP + k] = wsrc[i]; }
    }
    copy_g2l<2>(zs, zm, M * K4);
    __syncthreads();
    const int r = tid & 63, q = tid >> 6;
    const int kq = (K4 + 3) / 4, ka = q * kq, kb = min(K4, ka + kq);
    float acc[BN_MAXM];
#pragma unroll
    for (int m = 0; m < BN_MAXM; ++m) acc[m] = 0.f;
    for (int k = ka; k < kb; ++k) {
        const float w = ws[r * KP + k];
#pragma unroll
        for (int m = 0; m < BN_MAXM; ++m)
            if (m < M) acc[m] += w * zs[m * K4 + k];
    }
#pragma unroll
    for (int m = 0; m < BN_MAXM; ++m)
        if (m < M) part[(q * M + m) * 64 + r] = acc[m];
    __syncthreads();
    for (int i = tid; i < M * 64; i += 256) {
        const int m = i >> 6, rr = i & 63;
        const int n = (int)row0 + rr, c = n / S, sc = n - c * S;
        const float v = part[(0 * M + m) * 64 + rr] + part[(1 * M + m) * 64 + rr] + part[(2 * M + m) * 64 + rr] + part[(3 * M + m) * 64 + rr] + bd[n];
        out[((size_t)m * S + sc) * C + c] = from_f32<T>(v);
    }
}

// ------------------------------------------------------------------------------------------------ dec_input_bwd
// g[b][n] = gcl[b][s][c] (n = c * S + s).  dWd[n][k] = sum_b g[b][n] zm[b][k]; dbd[n] = sum_b g[b][n];
// dzm[b][k] = sum over blocks of sum_{n in block} g[b][n] Wd[n][k].  A block owns R = 64 CONSECUTIVE rows n (C % 64 == 0), so its slice
// of Wd and dWd is one contiguous run of R * K4 floats (float4 when K4 % 4 == 0).  Each block leaves its M x K4 partial of d(zm) in its OWN
// slot of the scratch with plain stores (no float atomics: the gradients are bit-reproducible); the consumers add the slots in index
// order (load_dzm: 16-byte loads, the slots of one launch are L2-resident).
template <typename T>
__global__ __launch_bounds__(256) void dec_input_bwd_kernel(const T* __restrict__ gcl, const float* __restrict__ zm, const float* __restrict__ Wd,
                                                            float* __restrict__ dWd, float* __restrict__ dbd, float* __restrict__ dzm_acc,
                                                            int M, int K4, int S, int C) {
    extern __shared__ float lds[];
    constexpr int R = 64;
    const int KP = K4 | 1;
    float* ws = lds;                                         // [R][KP]
    float* zs = ws + R * KP;                                 // [M][K4]
    float* gs = zs + M * K4;                                 // [M][R]
    const int tid = threadIdx.x;
    const size_t row0 = (size_t)blockIdx.x * R;
    const float* wsrc = Wd + row0 * K4;
    float* wdst = dWd + row0 * K4;
    const bool v4 = (K4 & 3) == 0;                           // rows start 16-byte aligned
    {                                                        // this block's M x 64 gradient values first (the scattered side), then the contiguous runs
        constexpr int UG = (BN_MAXM * R) / 256;              // <= 4 values per thread
        float gv[UG];
#pragma unroll
        for (int u = 0; u < UG; ++u) {
            const int i = min(tid + 256 * u, M * R - 1);
            const int m = i / R, n = (int)row0 + (i - m * R), c = n / S, sc = n - c * S;
            gv[u] = to_f32(gcl[((size_t)m * S + sc) * C + c]);
        }
        if (v4) copy_rows_g2l(ws, wsrc, R * K4, K4, KP);
        else {
            for (int i = tid; i < R * K4; i += 256) { const int r = i / K4, k = i - r * K4; ws[r * KP + k] = wsrc[i]; }
        }
        copy_g2l<2>(zs, zm, M * K4);
#pragma unroll
        for (int u = 0; u < UG; ++u) if (tid + 256 * u < M * R) gs[tid + 256 * u] = gv[u];
    }
    __syncthreads();
    if (v4) {
        for (int i = tid * 4; i < R * K4; i += 1024) {
            const int r = i / K4, k = i - r * K4;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int m = 0; m < M; ++m) {
                const float gv = gs[m * R + r];
                const float* z = zs + m * K4 + k;
                a.x += gv * z[0]; a.y += gv * z[1]; a.z += gv * z[2]; a.w += gv * z[3];
            }
            *(float4*)(wdst + i) = a;
        }
    } else {
        for (int i = tid; i < R * K4; i += 256) {
            const int r = i / K4, k = i - r * K4;
            float acc = 0.f;
            for (int m = 0; m < M; ++m) acc += gs[m * R + r] * zs[m * K4 + k];
            wdst[i] = acc;
        }
    }
    for (int r = tid; r < R; r += 256) {
        float acc = 0.f;
        for (int m = 0; m < M; ++m) acc += gs[m * R + r];
        dbd[row0 + r] = acc;
    }
    float* dz = dzm_acc + (size_t)blockIdx.x * M * K4;
    for (int i = tid; i < M * K4; i += 256) {
        const int m = i / K4, k = i - m * K4;
        float acc = 0.f;
#pragma unroll 8
        for (int r = 0; r < R; ++r) acc += gs[m * R + r] * ws[r * KP + k];
        dz[i] = acc;
    }
}

// ------------------------------------------------------------------------------------------------ d(zm) -> LDS
__device__ __forceinline__ float* lds_align16(float* p) { return (float*)(((uintptr_t)p + 15) & ~(uintptr_t)15); }
// dzs[i] = sum_q slot[q][i]; `n` floats per slot (n <= 2048).  With n % 4 == 0 (K4 % 4 == 0) all threads take part: thread (i4, part) adds the slots
// of its contiguous part of the slot range for 4 consecutive elements (one 16-byte load per slot, 16 in flight), the parts meet in `scratch`
// (>= n * parts floats of LDS behind dzs ... the caller's dynamic LDS is sized for it) and are added in part order — a fixed summation order.
__device__ void load_dzm(const float* __restrict__ dzm_acc, float* dzs, int n, int nslots, float* scratch) {
    const int nt = blockDim.x, n4 = n >> 2;
    if ((n & 3) == 0 && n4 <= nt) {
        const int parts = nt / n4, i4 = threadIdx.x % n4, part = threadIdx.x / n4;
        if (part < parts) {
            const int q0 = (int)((long long)nslots * part / parts), q1 = (int)((long long)nslots * (part + 1) / parts);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            int q = q0;
            for (; q + 16 <= q1; q += 16) {
                float4 u[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) u[j] = *(const float4*)(dzm_acc + (size_t)(q + j) * n + 4 * i4);
#pragma unroll
                for (int j = 0; j < 16; ++j) { v.x += u[j].x; v.y += u[j].y; v.z += u[j].z; v.w += u[j].w; }
            }
            for (; q < q1; ++q) { const float4 u = *(const float4*)(dzm_acc + (size_t)q * n + 4 * i4); v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w; }
            *(float4*)(scratch + (size_t)part * n + 4 * i4) = v;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += nt) {
            float a = 0.f;
            for (int p = 0; p < parts; ++p) a += scratch[(size_t)p * n + i];
            dzs[i] = a;
        }
        return;                                              // the caller's next __syncthreads() publishes dzs
    }
    if ((n & 3) == 0) {
        // more 4-element groups than threads (M = 16: 16 x 76 floats = 304 groups): every thread takes several groups, 16 slot loads in flight each, slots
        // added in index order.  (The element-wise loop below issued one DEPENDENT 4-byte load per slot: 256 slots = 256 round trips, 299 us for
        // mulv_bwd at B = 16 — the other half of the 64^3 configuration's round-2 slowdown.)
        for (int i4 = threadIdx.x; i4 < n4; i4 += nt) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            int q = 0;
            for (; q + 16 <= nslots; q += 16) {
                float4 u[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) u[j] = *(const float4*)(dzm_acc + (size_t)(q + j) * n + 4 * i4);
#pragma unroll
                for (int j = 0; j < 16; ++j) { v.x += u[j].x; v.y += u[j].y; v.z += u[j].z; v.w += u[j].w; }
            }
            for (; q < nslots; ++q) { const float4 u = *(const float4*)(dzm_acc + (size_t)q * n + 4 * i4); v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w; }
            *(float4*)(dzs + 4 * i4) = v;
        }
        return;
    }
    for (int i = threadIdx.x; i < n; i += nt) {
        float v = 0.f;
        for (int q = 0; q < nslots; ++q) v += dzm_acc[(size_t)q * n + i];
        dzs[i] = v;
    }
}

// ------------------------------------------------------------------------------------------------ mechanism_net backward
// One 256-thread workgroup (an extra block of the enc_fc.0 backward launch, which hides it).  lds: M*K4 + 5*M*HM + M*T + M*DM.
__device__ void mech_bwd(const TailDims& d, const TailParams& p, const TailGrads& gr, const TailSaved& sv, const float* __restrict__ dzm_part,
                         const float* __restrict__ g_mhat, const float* __restrict__ t_onehot, float* lds, float* __restrict__ sync_dy, float* __restrict__ sync_local) {
    const int M = d.M, T = blockDim.x, tid = threadIdx.x, HM = d.HM, K4 = d.Z + d.DM;
    float* dzs = lds;                    // [M][K4]
    float* dmh = dzs + M * K4;           // [M][DM]
    float* a2s = dmh + M * d.DM;         // [M][HM]
    float* a1n = a2s + M * HM;           // [M][HM]
    float* da2 = a1n + M * HM;           // [M][HM]
    float* dy = da2 + M * HM;            // [M][HM]
    float* xh = dy + M * HM;             // [M][HM]
    float* ts = xh + M * HM;             // [M][T]
    load_dzm(dzm_part, dzs, M * K4, d.NZ, lds_align16(ts + M * d.T));    // scratch: 4 * blockDim.x floats behind the operands (mech_bwd_lds)
    for (int i = tid; i < M * HM; i += T) { a2s[i] = sv.a2[i]; a1n[i] = sv.a1n[i]; xh[i] = sv.xhat[i]; }
    for (int i = tid; i < M * d.T; i += T) ts[i] = t_onehot[i];
    __syncthreads();
    for (int i = tid; i < M * d.DM; i += T) {
        const int m = i / d.DM, j = i - m * d.DM;
        dmh[i] = dzs[m * K4 + d.Z + j] + (g_mhat ? g_mhat[i] : 0.f);
    }
    __syncthreads();
    wg_linear_bwd(p.Wm5, gr.dWm5, gr.dbm5, dmh, a2s, da2, M, HM, d.DM);
    __syncthreads();
    for (int i = tid; i < M * HM; i += T) da2[i] = a2s[i] > 0.f ? da2[i] : 0.f;
    __syncthreads();
    wg_linear_bwd(p.Wm3, gr.dWm3, gr.dbm3, da2, a1n, dy, M, HM, HM);
    __syncthreads();
    for (int j = tid; j < HM; j += T) {                      // ReLU mask, then BatchNorm1d backward (batch statistics)
        float dgam = 0.f, dbet = 0.f;
        for (int m = 0; m < M; ++m) {
            const float g = a1n[m * HM + j] > 0.f ? dy[m * HM + j] : 0.f;
            dy[m * HM + j] = g;
            dgam += g * xh[m * HM + j];
            dbet += g;
        }
        gr.dgamma[j] = dgam; gr.dbeta[j] = dbet;
        if (sync_local) {
            // SyncBatchNorm: d(input) needs sum(dy) and sum(dy * xhat) over the GLOBAL batch.  This rank's sums (they are also its share of d beta /
            // d gamma) and its masked dy leave here; the caller all-reduces the sums and mech_bn_finish_kernel does the rest (dWm0, dbm0).
            sync_local[j] = dbet; sync_local[HM + j] = dgam;
            for (int m = 0; m < M; ++m) sync_dy[m * HM + j] = dy[m * HM + j];
            continue;
        }
        const float k = p.gamma[j] * sv.invstd[j] / (float)M;
        for (int m = 0; m < M; ++m) dy[m * HM + j] = bn_bwd_dx(k, (float)M, dy[m * HM + j], dbet, xh[m * HM + j], dgam);
    }
    if (sync_local) return;                                  // uniform
    __syncthreads();
    wg_linear_bwd(p.Wm0, gr.dWm0, gr.dbm0, dy, ts, nullptr, M, d.T, HM);
}

// ------------------------------------------------------------------------------------------------ SyncBatchNorm (data-parallel ranks)
// Before the step's forward: this rank's statistics of mechanism_net.0's output, h[m][j] = bm0[j] + t_onehot[m] . Wm0[j] (the same
// accumulation order as mech_fwd, so the values are mech_fwd's), as (sum, squared deviations from the rank's own mean).  One workgroup.
__global__ __launch_bounds__(256) void mech_bn_local_stats_kernel(const float* __restrict__ Wm0, const float* __restrict__ bm0, const float* __restrict__ t_onehot,
                                                                   const long long* __restrict__ t_labels, float* __restrict__ out, int M, int T, int HM) {
    for (int j = threadIdx.x; j < HM; j += blockDim.x) {
        float h[BN_MAXM];
        float sum = 0.f;
        for (int m = 0; m < M; ++m) {
            float acc = bm0[j];
            if (t_labels) {
                const long long lab = t_labels[m];
                if (lab >= 0 && lab < T) acc += Wm0[j * T + lab];          // 1.0f * w, the other terms of the dot product are 0.0f * w: the same bits
            } else {
                for (int k = 0; k < T; ++k) acc += t_onehot[m * T + k] * Wm0[j * T + k];
            }
            h[m] = acc; sum += acc;
        }
        const float mean = sum / (float)M;
        float m2 = 0.f;
        for (int m = 0; m < M; ++m) { const float c = h[m] - mean; m2 += c * c; }
        out[j] = sum; out[HM + j] = m2;
    }
}

// After the backward's all-reduce of (sum dy, sum dy * xhat): d(BatchNorm input) with the GLOBAL sums, then mechanism_net.0's weight and bias
// gradients (this rank's share, like every other parameter gradient).  One workgroup; lds: 2 * M * HM + M * T floats.
__global__ __launch_bounds__(256) void mech_bn_finish_kernel(TailDims d, TailParams p, TailGrads gr, TailSaved sv, const float* __restrict__ t_onehot,
                                                              const float* __restrict__ sync_dy, const float* __restrict__ sums, int ranks) {
    extern __shared__ float dyn_lds[];
    const int M = d.M, T = blockDim.x, tid = threadIdx.x, HM = d.HM;
    float* dy = dyn_lds;                 // [M][HM]
    float* ts = dy + M * HM;             // [M][T]
    const float nb = (float)M * (float)ranks;
    for (int i = tid; i < M * HM; i += T) {
        const int j = i % HM;
        const float k = p.gamma[j] * sv.invstd[j] / nb;
        dy[i] = bn_bwd_dx(k, nb, sync_dy[i], sums[j], sv.xhat[i], sums[HM + j]);
    }
    for (int i = tid; i < M * d.T; i += T) ts[i] = t_onehot[i];
    __syncthreads();
    wg_linear_bwd(p.Wm0, gr.dWm0, gr.dbm0, dy, ts, nullptr, M, d.T, HM);
}

// ------------------------------------------------------------------------------------------------ mulv_bwd
// fc_mu / fc_logvar backward.  grid N2 / 16: workgroup b owns columns k in [16 b, 16 b + 16) of both weights: dWmu[:, k], dWlv[:, k]
// and dh2[:, k] = relu'(h2) (dmu . Wmu[:, k] + dlogvar . Wlv[:, k]); block 0 also writes the two bias gradients.
// dmu = dz + g_mu, dlogvar = dz * eps * exp(logvar / 2) / 2 + g_logvar with dz = d(zm)[:, :Z] summed from the partials.
__global__ __launch_bounds__(256) void mulv_bwd_kernel(TailDims d, TailParams p, TailGrads gr, TailSaved sv, const float* __restrict__ dzm_part,
                                                       const float* __restrict__ g_mu, const float* __restrict__ g_logvar, const float* __restrict__ eps,
                                                       float* __restrict__ dh2) {
    extern __shared__ float lds[];
    const int M = d.M, tid = threadIdx.x, Z = d.Z, K4 = d.Z + d.DM;
    float* dzs = lds;                    // [M][K4]
    float* dmu = dzs + M * K4;           // [M][Z]
    float* dlv = dmu + M * Z;            // [M][Z]
    float* h2c = dlv + M * Z;            // [M][16]  this block's columns of h2
    float* red = h2c + M * 16;           // [16][M][16]
    // 16 columns per block (N2 / 16 blocks: 4 blocks of 64 columns left the launch to 4 CUs).  Everything this block reads from global
    // memory is requested up front (its weight rows, the element-wise operands of the first 256 (m, j) pairs, d(zm), its h2 columns):
    // one round trip instead of four dependent ones.
    const int k0 = blockIdx.x * 16;
    const int kl = tid & 15, q = tid >> 4, k = k0 + kl, kk = min(k, d.N2 - 1);
    constexpr int U = 4;
    float wm0[U], wl0[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int nn = min(q + 16 * u, Z - 1);
        wm0[u] = p.Wmu[(size_t)nn * d.N2 + kk]; wl0[u] = p.Wlv[(size_t)nn * d.N2 + kk];
    }
    const int ie = min(tid, M * Z - 1);
    const float e_eps = eps[ie], e_lv = sv.logvar[ie], e_gm = g_mu ? g_mu[ie] : 0.f, e_gl = g_logvar ? g_logvar[ie] : 0.f;
    load_dzm(dzm_part, dzs, M * K4, d.NZ, lds_align16(red + 16 * M * 16));     // scratch: 4 * 256 floats behind `red` (launch LDS size)
    for (int i = tid; i < M * 16; i += 256) h2c[i] = (k0 + (i & 15) < d.N2) ? sv.h2[(i >> 4) * d.N2 + k0 + (i & 15)] : 0.f;
    __syncthreads();
    for (int i = tid; i < M * Z; i += 256) {
        const int m = i / Z, j = i - m * Z;
        const float dz = dzs[m * K4 + j];
        const bool first = i < 256;                          // i == tid: operands already in registers
        const float ve = first ? e_eps : eps[i], vl = first ? e_lv : sv.logvar[i];
        const float vgm = first ? e_gm : (g_mu ? g_mu[i] : 0.f), vgl = first ? e_gl : (g_logvar ? g_logvar[i] : 0.f);
        dmu[i] = dz + vgm;
        dlv[i] = dz * ve * 0.5f * __expf(0.5f * vl) + vgl;
    }
    __syncthreads();
    float acc[BN_MAXM];
#pragma unroll
    for (int m = 0; m < BN_MAXM; ++m) acc[m] = 0.f;
    if (k < d.N2) {
        for (int n0 = q; n0 < Z; n0 += 16 * U) {             // thread (k, q): rows n = q, q + 16, ..; loads of U rows first (see fc2_fwd)
            float wm[U], wl[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (n0 == q) { wm[u] = wm0[u]; wl[u] = wl0[u]; }
                else {
                    const int nn = min(n0 + 16 * u, Z - 1);
                    wm[u] = p.Wmu[(size_t)nn * d.N2 + k]; wl[u] = p.Wlv[(size_t)nn * d.N2 + k];
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int n = n0 + 16 * u;
                if (n < Z) {
                    float gm = 0.f, gl = 0.f;
#pragma unroll
                    for (int m = 0; m < BN_MAXM; ++m)
                        if (m < M) {
                            acc[m] += dmu[m * Z + n] * wm[u] + dlv[m * Z + n] * wl[u];
                            gm += dmu[m * Z + n] * h2c[m * 16 + kl];
                            gl += dlv[m * Z + n] * h2c[m * 16 + kl];
                        }
                    gr.dWmu[(size_t)n * d.N2 + k] = gm;
                    gr.dWlv[(size_t)n * d.N2 + k] = gl;
                }
            }
        }
    }
#pragma unroll
    for (int m = 0; m < BN_MAXM; ++m)
        if (m < M) red[(q * M + m) * 16 + kl] = acc[m];
    __syncthreads();
    for (int i = tid; i < M * 16; i += 256) {
        const int m = i >> 4, c = i & 15;
        if (k0 + c < d.N2) {
            float v = 0.f;
#pragma unroll
            for (int qq = 0; qq < 16; ++qq) v += red[(qq * M + m) * 16 + c];
            dh2[m * d.N2 + k0 + c] = h2c[i] > 0.f ? v : 0.f;
        }
    }
    if (blockIdx.x == 0)
        for (int n = tid; n < Z; n += 256) {
            float bm = 0.f, bl = 0.f;
            for (int m = 0; m < M; ++m) { bm += dmu[m * Z + n]; bl += dlv[m * Z + n]; }
            gr.dbmu[n] = bm; gr.dblv[n] = bl;
        }
}

// ------------------------------------------------------------------------------------------------ fc2_bwd
// enc_fc.2 backward.  grid N1 / 16: workgroup b owns columns k in [16 b, 16 b + 16) of W2: dW2[:, k], g1[:, k] = relu'(h1) (dh2 . W2[:, k])
// and db1[k] = sum_m g1[m][k]; block 0 also writes db2.  Thread (k, q) walks rows n = q, q + 16, .. (64-byte row segments).
__global__ __launch_bounds__(256) void fc2_bwd_kernel(TailDims d, TailParams p, TailGrads gr, TailSaved sv, const float* __restrict__ dh2, float* __restrict__ g1) {
    extern __shared__ float lds[];
    const int M = d.M, tid = threadIdx.x, N = d.N2, K = d.N1;
    float* dhs = lds;                    // [M][N2]
    float* h1c = dhs + M * N;            // [M][16]
    float* red = h1c + M * 16;           // [16][M][16]
    const int k0 = blockIdx.x * 16;
    const int kl = tid & 15, q = tid >> 4, k = k0 + kl;
    constexpr int U = 8;                                     // the first U rows of W2 travel with the dh2 / h1 loads (see fc2_fwd)
    float w0[U];
#pragma unroll
    for (int u = 0; u < U; ++u) w0[u] = p.W2[(size_t)min(q + 16 * u, N - 1) * K + min(k, K - 1)];
    copy_g2l<4>(dhs, dh2, M * N);
    for (int i = tid; i < M * 16; i += 256) h1c[i] = (k0 + (i & 15) < K) ? sv.h1[(i >> 4) * K + k0 + (i & 15)] : 0.f;
    __syncthreads();
    float acc[BN_MAXM];
#pragma unroll
    for (int m = 0; m < BN_MAXM; ++m) acc[m] = 0.f;
    if (k < K) {
        for (int n0 = q; n0 < N; n0 += 16 * U) {
            float wv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) wv[u] = (n0 == q) ? w0[u] : p.W2[(size_t)min(n0 + 16 * u, N - 1) * K + k];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int n = n0 + 16 * u;
                if (n < N) {
                    float dw = 0.f;
#pragma unroll
                    for (int m = 0; m < BN_MAXM; ++m)
                        if (m < M) { acc[m] += dhs[m * N + n] * wv[u]; dw += dhs[m * N + n] * h1c[m * 16 + kl]; }
                    gr.dW2[(size_t)n * K + k] = dw;
                }
            }
        }
    }
#pragma unroll
    for (int m = 0; m < BN_MAXM; ++m)
        if (m < M) red[(q * M + m) * 16 + kl] = acc[m];
    __syncthreads();
    if (tid < 16 && k0 + tid < K) {
        float bsum = 0.f;
        for (int m = 0; m < M; ++m) {
            float v = 0.f;
            for (int qq = 0; qq < 16; ++qq) v += red[(qq * M + m) * 16 + tid];
            v = h1c[m * 16 + tid] > 0.f ? v : 0.f;
            g1[m * K + k0 + tid] = v;
            bsum += v;
        }
        gr.db1[k0 + tid] = bsum;
    }
    if (blockIdx.x == 0)
        for (int n = tid; n < N; n += 256) {
            float b = 0.f;
            for (int m = 0; m < M; ++m) b += dhs[m * N + n];
            gr.db2[n] = b;
        }
}

// ------------------------------------------------------------------------------------------------ skinny_bwd_colwise
// enc_fc.0 backward in one pass over W: thread owns column k; for its slice of rows n it writes dW[n][k] = sum_m g[m][n] x[m][k]
// and accumulates dx[m][k] += g[m][n] W[n][k].  grid (ceil(K / 256), NS).  The partial dx of feature column f = c * S + s is
// stored cell-major, dxp[ns][m][s][c], so pool_bwd reads it coalesced along c (columns >= F — m and t — need no gradient).
struct __attribute__((packed, aligned(4))) F2U { float x, y; };                // float2 at dword alignment
template <int MT> struct SkinnyBwdCols { static constexpr int value = 4; using vec = F4U; };
template <> struct SkinnyBwdCols<16> { static constexpr int value = 2; using vec = F2U; };
template <int MT>
__global__ __launch_bounds__(256) void skinny_bwd_colwise_kernel(const float* __restrict__ g, const float* __restrict__ x, const float* __restrict__ Wt,
                                                                 float* __restrict__ dW, float* __restrict__ dxp, int M, int K, int N, int nslice, int F, int S,
                                                                 int NS, TailDims d, TailParams tp, TailGrads tg, TailSaved sv, MechBwdArgs mb) {
    extern __shared__ float dyn_lds[];
    if ((int)blockIdx.y == NS) {                             // extra block row: mechanism_net's backward, hidden behind the W stream
        if (blockIdx.x == 0) mech_bwd(d, tp, tg, sv, mb.dzm_part, mb.g_mhat, mb.t_onehot, dyn_lds, mb.sync_dy, mb.sync_local);
        return;
    }
    __shared__ float gs[MT][64];
    // thread owns columns k0 .. k0 + CPT - 1: 16-byte (8-byte) loads of W and stores of dW at 4-byte alignment (rows of odd length K are not
    // 16-byte aligned; global_load/store_dwordx4 take dword-aligned addresses), a quarter of the memory instructions of the scalar form.
    // CPT = 4 up to 8 rows; 16 rows (the 64^3 fp32 configuration, B = 16) keep 2 columns per thread: x and the dx accumulators are 2 x MT x CPT registers,
    // and with 4 columns the MT = 16 instance needed > 256 VGPRs (528 bytes of scratch per lane, one wave per SIMD: 494 us for this launch — 14 % of the
    // whole 64^3 step, the "3.09 -> 3.5 ms" of round 2's record; with 2 columns it fits in 150).
    constexpr int CPT = SkinnyBwdCols<MT>::value;
    const int k0 = (blockIdx.x * 256 + threadIdx.x) * CPT, ns = blockIdx.y;
    const int n0 = ns * nslice, n1 = min(N, n0 + nslice);
    const int nk = min(CPT, K - k0);                         // live columns of this thread (<= 0: none)
    float xv[MT][CPT], acc[MT][CPT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int c = 0; c < CPT; ++c) { xv[m][c] = (m < M && c < nk) ? x[(size_t)m * K + k0 + c] : 0.f; acc[m][c] = 0.f; }
    for (int nb = n0; nb < n1; nb += 64) {
        const int cnt = min(64, n1 - nb);
        __syncthreads();
        for (int i = threadIdx.x; i < MT * 64; i += 256) {
            const int m = i >> 6, j = i & 63;
            gs[m][j] = (m < M && j < cnt) ? g[(size_t)m * N + nb + j] : 0.f;
        }
        __syncthreads();
        if (nk == CPT) {
            constexpr int UNR = MT >= 16 ? 2 : CVAE_BN_BWD_UNROLL;      // an unrolled row holds MT values of g: 8 rows of 16 are 128 registers by themselves
#pragma unroll UNR
            for (int j = 0; j < cnt; ++j) {
                typename SkinnyBwdCols<MT>::vec w = ld_stream<typename SkinnyBwdCols<MT>::vec>(Wt + (size_t)(nb + j) * K + k0), dw;
                const float* wf = (const float*)&w;
                float* dwf = (float*)&dw;
#pragma unroll
                for (int c = 0; c < CPT; ++c) dwf[c] = 0.f;
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const float gv = gs[m][j];
#pragma unroll
                    for (int c = 0; c < CPT; ++c) { dwf[c] += gv * xv[m][c]; acc[m][c] += gv * wf[c]; }
                }
                st_stream<typename SkinnyBwdCols<MT>::vec>(dW + (size_t)(nb + j) * K + k0, dw);
            }
        } else if (nk > 0) {                                 // the row tail (K % CPT columns)
            for (int j = 0; j < cnt; ++j)
                for (int c = 0; c < nk; ++c) {
                    const float w = Wt[(size_t)(nb + j) * K + k0 + c];
                    float dw = 0.f;
#pragma unroll
                    for (int m = 0; m < MT; ++m) { dw += gs[m][j] * xv[m][c]; acc[m][c] += gs[m][j] * w; }
                    dW[(size_t)(nb + j) * K + k0 + c] = dw;
                }
        }
    }
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
        const int k = k0 + c;
        if (k < F) {
            const int ch = k / S, sp = k - ch * S, C = F / S;
#pragma unroll
            for (int m = 0; m < MT; ++m)
                if (m < M) dxp[(((size_t)ns * M + m) * S + sp) * C + ch] = acc[m][c];
        }
    }
}

// ------------------------------------------------------------------------------------------------ pool_bwd
// grid (S, M): block (s, b) sums the NS partials of cell s for all channels and writes value / |window| to every voxel of the
// window (windows must not overlap: D % OD == H % OH == W % OW == 0), zeroed where the pooled activation was not positive.
template <typename T>
__global__ __launch_bounds__(256) void pool_bwd_kernel(const float* __restrict__ dxp, const T* __restrict__ y, T* __restrict__ dy, int NS, int M,
                                                       int D, int H, int W, int C, int OD, int OH, int OW, int relu_mask) {
    const int S = OD * OH * OW, b = blockIdx.y, s = blockIdx.x;
    const int ow = s % OW, oh = (s / OW) % OH, od = s / (OW * OH);
    const int d0 = pool_lo(od, D, OD), d1 = pool_hi(od, D, OD), h0 = pool_lo(oh, H, OH), h1 = pool_hi(oh, H, OH), w0 = pool_lo(ow, W, OW), w1 = pool_hi(ow, W, OW);
    const int nh = h1 - h0, nw = w1 - w0, nvox = (d1 - d0) * nh * nw;
    const float inv = 1.f / (float)nvox;
    // 256 channels at a time: 4 thread groups split the NS partials (float4 = 4 channels per thread), LDS combines them, then the
    // window is written in 16-byte pieces (one piece of the ReLU mask read per piece written).
    __shared__ float4 part[4][64];
    __shared__ __attribute__((aligned(16))) float val[256];
    constexpr int EPP = 16 / sizeof(T);                      // channels per 16-byte piece
    const int l = threadIdx.x & 63, q = threadIdx.x >> 6;
    for (int c0 = 0; c0 < C; c0 += 256) {
        const int cw = min(256, C - c0);                     // multiple of 64
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (4 * l < cw) {
            constexpr int U = 8;                             // partials in flight (clamped index, predicated add: same order of the sum)
            for (int n0 = q; n0 < NS; n0 += 4 * U) {
                float4 v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) v[u] = *(const float4*)(dxp + (((size_t)min(n0 + 4 * u, NS - 1) * M + b) * S + s) * C + c0 + 4 * l);
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (n0 + 4 * u < NS) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
            }
        }
        part[q][l] = a;
        __syncthreads();
        if (q == 0) {
            const float4 b1 = part[1][l], b2 = part[2][l], b3 = part[3][l];
            *(float4*)(val + 4 * l) = make_float4((a.x + b1.x + b2.x + b3.x) * inv, (a.y + b1.y + b2.y + b3.y) * inv, (a.z + b1.z + b2.z + b3.z) * inv,
                                                  (a.w + b1.w + b2.w + b3.w) * inv);
        }
        __syncthreads();
        const int ppv = cw / EPP;                            // pieces per voxel
        for (int i = threadIdx.x; i < nvox * ppv; i += 256) {
            const int v = i / ppv, pc = (i - v * ppv) * EPP;
            const int ww = w0 + v % nw, hh = h0 + (v / nw) % nh, dd = d0 + v / (nw * nh);
            const size_t idx = ((((size_t)b * D + dd) * H + hh) * W + ww) * C + c0 + pc;
            union { uint4 u; T e[EPP]; } mk, o;
            if (relu_mask) mk.u = *(const uint4*)(y + idx);
#pragma unroll
            for (int e = 0; e < EPP; ++e) o.e[e] = from_f32<T>((!relu_mask || to_f32(mk.e[e]) > 0.f) ? val[pc + e] : 0.f);
            *(uint4*)(dy + idx) = o.u;
        }
        __syncthreads();
    }
}

size_t mech_fwd_lds(const TailDims& d) { return sizeof(float) * (size_t)d.M * (d.T + 2 * d.HM); }
size_t mech_bwd_lds(const TailDims& d) { return sizeof(float) * ((size_t)d.M * (d.Z + d.DM + d.DM + 5 * d.HM + d.T) + 4 * 256 + 4); }   // + load_dzm's scratch (16-byte aligned)

}  // namespace

// Host-side description of the bottleneck (plain C struct of the C ABI, see include/cvae_hip.h).
static bool dims_ok(const cvae_bottleneck_dims* q) {
    if (!q) return false;
    if (q->M < 1 || q->M > BN_MAXM) return false;
    if (q->C < 64 || q->C % 64 || q->D < 1 || q->H < 1 || q->W < 1 || q->OD < 1 || q->OH < 1 || q->OW < 1) return false;
    if (q->D % q->OD || q->H % q->OH || q->W % q->OW) return false;
    if (q->m_dim < 1 || q->t_dim < 1 || q->N1 < 4 || q->N2 < 1 || q->Z < 1 || q->HM < 1 || q->HM > 1024) return false;
    if (q->Z + q->m_dim > 128 || q->N1 > 4096 || q->N2 > 2048) return false;            // LDS / register budgets of the level kernels
    return true;
}
static TailDims tail_dims(const cvae_bottleneck_dims* q, int KS, int P) {
    return TailDims{(int)q->M, (int)q->N1, (int)q->N2, (int)q->Z, (int)q->t_dim, (int)q->HM, (int)q->m_dim, KS, P, (int)(q->C * q->OD * q->OH * q->OW / 64)};
}
// Split factors of the two passes over enc_fc.0's weight: the forward slices K into parts of >= 4096 columns (<= 8), the backward gives a
// workgroup 1024 columns x 16 rows (32 partial d(xcat) slabs).  Measured in the step (A/B of two builds in one session): halving both
// (2048 columns / 8 rows: twice the workgroups, half the bytes each) costs +10 us per step, the extra partial slabs outweigh the parallelism.
#ifndef CVAE_BN_FWD_SLICE
#define CVAE_BN_FWD_SLICE 4096
#endif
#ifndef CVAE_BN_BWD_ROWS
#define CVAE_BN_BWD_ROWS 16
#endif
static int fwd_ksplit(int64_t K1) { int ks = (int)(K1 / CVAE_BN_FWD_SLICE); return ks < 1 ? 1 : (ks > 8 ? 8 : ks); }
static int bwd_nsplit(int64_t N1) { int ns = (int)(N1 / CVAE_BN_BWD_ROWS); return ns < 1 ? 1 : (ns > 64 ? 64 : ns); }

extern "C" int cvae_bottleneck_sizes(const cvae_bottleneck_dims* q, int64_t* K1, int64_t* K4, int64_t* fwd_partial_floats, int64_t* dzm_partial_floats,
                                     int64_t* dx_partial_floats) {
    if (!dims_ok(q)) return CVAE_E_BADSHAPE;
    const int64_t S = q->OD * q->OH * q->OW, F = q->C * S, k1 = F + q->m_dim + q->t_dim, k4 = q->Z + q->m_dim;
    if (K1) *K1 = k1;
    if (K4) *K4 = k4;
    if (fwd_partial_floats) *fwd_partial_floats = (int64_t)fwd_ksplit(k1) * q->M * q->N1;
    if (dzm_partial_floats) *dzm_partial_floats = (F / 64) * q->M * k4;       // one slot per dec_input_bwd workgroup
    if (dx_partial_floats) *dx_partial_floats = (int64_t)bwd_nsplit(q->N1) * q->M * F;
    return CVAE_OK;
}

template <int MT>
static void launch_fwd_partial(const float* x, const float* W1, float* partial, int M, int K, int N, int KS, const TailDims& d, const TailParams& p, const TailSaved& sv,
                               const MechFwdArgs& ma, hipStream_t st) {
    const int kslice = (K + KS - 1) / KS;
    hipLaunchKernelGGL(skinny_fwd_partial_kernel<MT>, dim3((unsigned)((N + 3) / 4), (unsigned)(KS + 1)), dim3(256), mech_fwd_lds(d), st, x, W1, partial, M, K, N, kslice,
                       KS, d, p, sv, ma);
}
template <int MT>
static void launch_bwd_colwise(const float* g, const float* x, const float* W1, float* dW, float* dxp, int M, int K, int N, int NS, int F, int S, const TailDims& d,
                               const TailParams& p, const TailGrads& tg, const TailSaved& sv, const MechBwdArgs& mb, hipStream_t st) {
    const int nslice = (N + NS - 1) / NS;
    hipLaunchKernelGGL(skinny_bwd_colwise_kernel<MT>, dim3((unsigned)((K + 256 * SkinnyBwdCols<MT>::value - 1) / (256 * SkinnyBwdCols<MT>::value)), (unsigned)(NS + 1)), dim3(256), mech_bwd_lds(d), st, g, x, W1, dW, dxp, M, K, N,
                       nslice, F, S, NS, d, p, tg, sv, mb);
}

extern "C" int cvae_bottleneck_bn_local_stats(const float* Wm0, const float* bm0, const float* t_onehot, const int64_t* t_labels, float* local_stats, int64_t M, int64_t t_dim,
                                              int64_t HM, void* stream) {
    if (M < 1 || M > BN_MAXM || t_dim < 1 || HM < 1 || HM > 1024) return CVAE_E_BADSHAPE;
    if (!Wm0 || !bm0 || !local_stats || (!t_onehot && !t_labels)) return CVAE_E_NULLPTR;
    hipLaunchKernelGGL(mech_bn_local_stats_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, Wm0, bm0, t_onehot, (const long long*)t_labels, local_stats, (int)M, (int)t_dim, (int)HM);
    CVAE_CHECK_LAUNCH();
    return CVAE_OK;
}

extern "C" int cvae_bottleneck_fwd(const cvae_bottleneck_dims* q, const cvae_bottleneck_params* w, const void* y_cl, const float* m, float* t_onehot,
                                   const int64_t* t_labels, const float* eps, float* running_mean, float* running_var, long long* num_batches_tracked, float momentum, float bn_eps,
                                   int bn_training, float* xcat, float* partial, float* dzm_acc, const cvae_bottleneck_saved* sv, void* dec_cl, int dtype,
                                   void* stream) {
    return cvae_bottleneck_fwd_ex(q, w, y_cl, m, t_onehot, t_labels, (float*)eps, running_mean, running_var, num_batches_tracked, momentum, bn_eps, bn_training, xcat, partial,
                                  dzm_acc, sv, dec_cl, dtype, nullptr, 0, nullptr, stream);
}

extern "C" int cvae_bottleneck_fwd_sync(const cvae_bottleneck_dims* q, const cvae_bottleneck_params* w, const void* y_cl, const float* m, float* t_onehot,
                                        const int64_t* t_labels, const float* eps, float* running_mean, float* running_var, long long* num_batches_tracked, float momentum,
                                        float bn_eps, int bn_training, float* xcat, float* partial, float* dzm_acc, const cvae_bottleneck_saved* sv, void* dec_cl, int dtype,
                                        const float* bn_rank_stats, int bn_ranks, void* stream) {
    return cvae_bottleneck_fwd_ex(q, w, y_cl, m, t_onehot, t_labels, (float*)eps, running_mean, running_var, num_batches_tracked, momentum, bn_eps, bn_training, xcat, partial,
                                  dzm_acc, sv, dec_cl, dtype, bn_rank_stats, bn_ranks, nullptr, stream);
}

extern "C" int cvae_bottleneck_fwd_ex(const cvae_bottleneck_dims* q, const cvae_bottleneck_params* w, const void* y_cl, const float* m, float* t_onehot,
                                      const int64_t* t_labels, float* eps, float* running_mean, float* running_var, long long* num_batches_tracked, float momentum,
                                      float bn_eps, int bn_training, float* xcat, float* partial, float* dzm_acc, const cvae_bottleneck_saved* sv, void* dec_cl, int dtype,
                                      const float* bn_rank_stats, int bn_ranks, const cvae_bottleneck_noise* noise, void* stream) {
    if (!dims_ok(q)) return CVAE_E_BADSHAPE;
    if (noise && !noise->call_counter) return CVAE_E_NULLPTR;
    const NoiseDraw nz = noise ? NoiseDraw{eps, (int)(q->M * q->Z), noise->seed, noise->subsequence, noise->call_counter} : NoiseDraw{nullptr, 0, 0, 0, nullptr};
    if (bn_rank_stats && (bn_ranks < 1 || !bn_training)) return CVAE_E_BADSHAPE;
    if (dtype != CVAE_F32 && dtype != CVAE_BF16) return CVAE_E_DTYPE;
    if (!w || !sv || !y_cl || !m || !t_onehot || !eps || !xcat || !partial || !dec_cl) return CVAE_E_NULLPTR;
    if (bn_training && q->M * (bn_rank_stats ? bn_ranks : 1) < 2) return CVAE_E_BADSHAPE;
    if (!bn_training && (!running_mean || !running_var)) return CVAE_E_NULLPTR;
    hipStream_t st = (hipStream_t)stream;
    const int M = (int)q->M, S = (int)(q->OD * q->OH * q->OW), C = (int)q->C, F = C * S;
    const int K1 = F + (int)q->m_dim + (int)q->t_dim, K4 = (int)(q->Z + q->m_dim), KS = fwd_ksplit(K1);
    if (dtype == CVAE_BF16)
        hipLaunchKernelGGL(pool_cat_fwd_kernel<bf16>, dim3(S + 1, M), dim3(256), 0, st, (const bf16*)y_cl, m, t_onehot, (const long long*)t_labels, xcat, (int)q->D, (int)q->H, (int)q->W, C,
                           (int)q->OD, (int)q->OH, (int)q->OW, (int)q->m_dim, (int)q->t_dim, K1, nz);
    else
        hipLaunchKernelGGL(pool_cat_fwd_kernel<float>, dim3(S + 1, M), dim3(256), 0, st, (const float*)y_cl, m, t_onehot, (const long long*)t_labels, xcat, (int)q->D, (int)q->H, (int)q->W, C,
                           (int)q->OD, (int)q->OH, (int)q->OW, (int)q->m_dim, (int)q->t_dim, K1, nz);
    CVAE_CHECK_LAUNCH();
    const TailDims d = tail_dims(q, KS, 0);
    const TailParams p{w->b1, w->W2, w->b2, w->Wmu, w->bmu, w->Wlv, w->blv, w->Wm0, w->bm0, w->gamma, w->beta, w->Wm3, w->bm3, w->Wm5, w->bm5};
    const TailSaved s{sv->h1, sv->h2, sv->mu, sv->logvar, sv->xhat, sv->invstd, sv->a1n, sv->a2, sv->m_hat, sv->zm};
    const MechFwdArgs ma{t_onehot, running_mean, running_var, num_batches_tracked, momentum, bn_eps, bn_training, bn_rank_stats, bn_ranks};
    if (M <= 4) launch_fwd_partial<4>(xcat, w->W1, partial, M, K1, (int)q->N1, KS, d, p, s, ma, st);
    else if (M <= 8) launch_fwd_partial<8>(xcat, w->W1, partial, M, K1, (int)q->N1, KS, d, p, s, ma, st);
    else launch_fwd_partial<16>(xcat, w->W1, partial, M, K1, (int)q->N1, KS, d, p, s, ma, st);
    CVAE_CHECK_LAUNCH();
    hipLaunchKernelGGL(fc2_fwd_kernel, dim3((unsigned)((q->N2 + 3) / 4)), dim3(256), sizeof(float) * (size_t)M * q->N1, st, d, p, s, (const float*)partial);
    CVAE_CHECK_LAUNCH();
    hipLaunchKernelGGL(mulv_fwd_kernel, dim3((unsigned)((q->Z + 3) / 4)), dim3(256), sizeof(float) * (size_t)M * q->N2, st, d, p, s, eps);
    CVAE_CHECK_LAUNCH();
    const size_t lds_d = sizeof(float) * ((size_t)64 * (K4 | 1) + (size_t)M * K4 + (size_t)4 * M * 64);
    if (dtype == CVAE_BF16)
        hipLaunchKernelGGL(dec_input_fwd_kernel<bf16>, dim3(C * S / 64), dim3(256), lds_d, st, (const float*)sv->zm, w->Wd, w->bd, (bf16*)dec_cl, M, K4, S, C);
    else
        hipLaunchKernelGGL(dec_input_fwd_kernel<float>, dim3(C * S / 64), dim3(256), lds_d, st, (const float*)sv->zm, w->Wd, w->bd, (float*)dec_cl, M, K4, S, C);
    CVAE_CHECK_LAUNCH();
    return CVAE_OK;
}

extern "C" int cvae_bottleneck_bwd(const cvae_bottleneck_dims* q, const cvae_bottleneck_params* w, const cvae_bottleneck_grads* gr, const cvae_bottleneck_saved* sv,
                                   const void* g_dec_cl, const float* g_mu, const float* g_logvar, const float* g_mhat, const float* t_onehot, const float* eps,
                                   const float* xcat, const void* y_cl, int relu_mask, float* dzm_partial, float* g1, float* dx_partial, void* dy_cl, int dtype,
                                   void* stream) {
    return cvae_bottleneck_bwd_sync(q, w, gr, sv, g_dec_cl, g_mu, g_logvar, g_mhat, t_onehot, eps, xcat, y_cl, relu_mask, dzm_partial, g1, dx_partial, dy_cl, dtype, nullptr,
                                    nullptr, stream);
}

extern "C" int cvae_bottleneck_bn_bwd_finish(const cvae_bottleneck_dims* q, const cvae_bottleneck_params* w, const cvae_bottleneck_grads* gr, const cvae_bottleneck_saved* sv,
                                             const float* t_onehot, const float* bn_dy, const float* bn_sums, int bn_ranks, void* stream) {
    if (!dims_ok(q) || bn_ranks < 1) return CVAE_E_BADSHAPE;
    if (!w || !gr || !sv || !t_onehot || !bn_dy || !bn_sums) return CVAE_E_NULLPTR;
    const TailDims d = tail_dims(q, 0, 0);
    const TailParams p{w->b1, w->W2, w->b2, w->Wmu, w->bmu, w->Wlv, w->blv, w->Wm0, w->bm0, w->gamma, w->beta, w->Wm3, w->bm3, w->Wm5, w->bm5};
    const TailGrads g{gr->db1, gr->dW2, gr->db2, gr->dWmu, gr->dbmu, gr->dWlv, gr->dblv, gr->dWm0, gr->dbm0, gr->dgamma, gr->dbeta, gr->dWm3, gr->dbm3, gr->dWm5, gr->dbm5};
    const TailSaved s{sv->h1, sv->h2, sv->mu, sv->logvar, sv->xhat, sv->invstd, sv->a1n, sv->a2, sv->m_hat, sv->zm};
    hipLaunchKernelGGL(mech_bn_finish_kernel, dim3(1), dim3(256), sizeof(float) * (size_t)q->M * (size_t)(q->HM + q->t_dim), (hipStream_t)stream, d, p, g, s, t_onehot, bn_dy, bn_sums,
                       bn_ranks);
    CVAE_CHECK_LAUNCH();
    return CVAE_OK;
}

extern "C" int cvae_bottleneck_bwd_sync(const cvae_bottleneck_dims* q, const cvae_bottleneck_params* w, const cvae_bottleneck_grads* gr, const cvae_bottleneck_saved* sv,
                                        const void* g_dec_cl, const float* g_mu, const float* g_logvar, const float* g_mhat, const float* t_onehot, const float* eps,
                                        const float* xcat, const void* y_cl, int relu_mask, float* dzm_partial, float* g1, float* dx_partial, void* dy_cl, int dtype,
                                        float* bn_dy, float* bn_local_sums, void* stream) {
    if (!dims_ok(q)) return CVAE_E_BADSHAPE;
    if ((bn_dy == nullptr) != (bn_local_sums == nullptr)) return CVAE_E_NULLPTR;
    if (dtype != CVAE_F32 && dtype != CVAE_BF16) return CVAE_E_DTYPE;
    if (!w || !gr || !sv || !g_dec_cl || !t_onehot || !eps || !xcat || !y_cl || !dzm_partial || !g1 || !dx_partial || !dy_cl) return CVAE_E_NULLPTR;
    hipStream_t st = (hipStream_t)stream;
    const int M = (int)q->M, S = (int)(q->OD * q->OH * q->OW), C = (int)q->C, F = C * S;
    const int K1 = F + (int)q->m_dim + (int)q->t_dim, K4 = (int)(q->Z + q->m_dim), NS = bwd_nsplit(q->N1), P = S;
    const size_t lds_d = sizeof(float) * ((size_t)64 * (K4 | 1) + (size_t)M * K4 + (size_t)M * 64);
    if (dtype == CVAE_BF16)
        hipLaunchKernelGGL(dec_input_bwd_kernel<bf16>, dim3(F / 64), dim3(256), lds_d, st, (const bf16*)g_dec_cl, (const float*)sv->zm, w->Wd, gr->dWd, gr->dbd,
                           dzm_partial, M, K4, S, C);
    else
        hipLaunchKernelGGL(dec_input_bwd_kernel<float>, dim3(F / 64), dim3(256), lds_d, st, (const float*)g_dec_cl, (const float*)sv->zm, w->Wd, gr->dWd, gr->dbd,
                           dzm_partial, M, K4, S, C);
    CVAE_CHECK_LAUNCH();
    const TailDims d = tail_dims(q, 0, P);
    const TailParams p{w->b1, w->W2, w->b2, w->Wmu, w->bmu, w->Wlv, w->blv, w->Wm0, w->bm0, w->gamma, w->beta, w->Wm3, w->bm3, w->Wm5, w->bm5};
    const TailGrads g{gr->db1, gr->dW2, gr->db2, gr->dWmu, gr->dbmu, gr->dWlv, gr->dblv, gr->dWm0, gr->dbm0, gr->dgamma, gr->dbeta, gr->dWm3, gr->dbm3, gr->dWm5, gr->dbm5};
    const TailSaved s{sv->h1, sv->h2, sv->mu, sv->logvar, sv->xhat, sv->invstd, sv->a1n, sv->a2, sv->m_hat, sv->zm};
    float* dh2 = g1 + (size_t)M * q->N1;                     // second part of the g1 scratch
    hipLaunchKernelGGL(mulv_bwd_kernel, dim3((unsigned)((q->N2 + 15) / 16)), dim3(256), sizeof(float) * ((size_t)M * K4 + 2 * (size_t)M * q->Z + 17 * (size_t)M * 16 + 4 * 256 + 4), st,
                       d, p, g, s, (const float*)dzm_partial, g_mu, g_logvar, eps, dh2);
    CVAE_CHECK_LAUNCH();
    hipLaunchKernelGGL(fc2_bwd_kernel, dim3((unsigned)((q->N1 + 15) / 16)), dim3(256), sizeof(float) * ((size_t)M * q->N2 + 17 * (size_t)M * 16), st, d, p, g, s,
                       (const float*)dh2, g1);
    CVAE_CHECK_LAUNCH();
    const MechBwdArgs mb{dzm_partial, g_mhat, t_onehot, bn_dy, bn_local_sums};
    if (M <= 4) launch_bwd_colwise<4>(g1, xcat, w->W1, gr->dW1, dx_partial, M, K1, (int)q->N1, NS, F, S, d, p, g, s, mb, st);
    else if (M <= 8) launch_bwd_colwise<8>(g1, xcat, w->W1, gr->dW1, dx_partial, M, K1, (int)q->N1, NS, F, S, d, p, g, s, mb, st);
    else launch_bwd_colwise<16>(g1, xcat, w->W1, gr->dW1, dx_partial, M, K1, (int)q->N1, NS, F, S, d, p, g, s, mb, st);
    CVAE_CHECK_LAUNCH();
    if (dtype == CVAE_BF16)
        hipLaunchKernelGGL(pool_bwd_kernel<bf16>, dim3(S, M), dim3(256), 0, st, (const float*)dx_partial, (const bf16*)y_cl, (bf16*)dy_cl, NS, M, (int)q->D, (int)q->H,
                           (int)q->W, C, (int)q->OD, (int)q->OH, (int)q->OW, relu_mask);
    else
        hipLaunchKernelGGL(pool_bwd_kernel<float>, dim3(S, M), dim3(256), 0, st, (const float*)dx_partial, (const float*)y_cl, (float*)dy_cl, NS, M, (int)q->D, (int)q->H,
                           (int)q->W, C, (int)q->OD, (int)q->OH, (int)q->OW, relu_mask);
    CVAE_CHECK_LAUNCH();
    return CVAE_OK;
}
