// recon_loss.hip — the reconstruction end of the training step for the exact-2x resize (64^3 decoder output -> 128^3 volume):
// F.interpolate(x_dec, size, mode='trilinear', align_corners=False) followed by the sum-of-squares reconstruction term of
// loss_function (causal_cascade/models.py:84-87, train.py:5-17 of the reference).
//
// The resized volume is 8x the decoder output (33.5 MB fp32 per batch of 4); written, re-read by the loss, re-read by the loss
// backward, and its gradient written and re-read by the resize backward, it made ~240 MB of traffic around 2 MB of information.
// Here the 2x resize is recomputed where it is needed from the small tensor:
//   up2x_block_kernel<MODE 0>  recon = up(src)                     (model.forward: the volume a caller asked for; MODE 4: the same with nontemporal
//                                                                   stores, for outputs of 256 MB and more — the counterfactual sweep writes 2 GB)
//   up2x_block_kernel<MODE 1>  sum (up(src) - x)^2                 (ELBO forward without a backward to follow: reads x once, never writes the volume)
//   up2x_block_kernel<MODE 3>  the same sum AND t1 = U_w^T (up(src) - x)   (ELBO forward of a training step: x is read ONCE per step; the
//                              backward used to re-read it, 33.5 MB, in a launch of its own)
//   up2x_bwd_b_kernel          d src = 2 g U_d^T U_h^T t1          (ELBO backward; g = the incoming loss gradient, read on the device)
// A thread owns 4 consecutive source voxels along w and their 2 x 2 x 8 block of outputs; the 3 x 3 x 6 source neighbourhood is
// loaded once (1.7 loads per output instead of 8).  The tap weights are the values lin_tap() of the generic kernels produces and the
// interpolation keeps aten's association (w, then h, then d), so MODE 0 returns the same numbers as cvae_upsample_linear_fwd.
#include "common.h"

namespace {

struct LinTap { int i0, i1; float w0, w1; };
// torch upsample_linear, align_corners=False: src = scale * (dst + 0.5) - 0.5 clamped at 0, scale = in / out (float)
__device__ __forceinline__ LinTap lin_tap(int o, int in, float scale) {
    float s = scale * ((float)o + 0.5f) - 0.5f;
    if (s < 0.f) s = 0.f;
    LinTap t;
    t.i0 = (int)s;
    if (t.i0 > in - 1) t.i0 = in - 1;
    t.i1 = t.i0 + ((t.i0 < in - 1) ? 1 : 0);
    t.w1 = s - (float)t.i0;
    t.w0 = 1.f - t.w1;
    return t;
}
// weights with which outputs j0 .. j0 + 3 (j0 = 2 i - 1) of a 2x-resized axis read source index i
struct Win4 { int j0; float w[4]; };
__device__ __forceinline__ Win4 win4(int i, int in, int out, float scale, bool strided) {
    Win4 r;
    if (!strided) { r.j0 = i; r.w[0] = 1.f; r.w[1] = r.w[2] = r.w[3] = 0.f; return r; }
    r.j0 = 2 * i - 1;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int j = r.j0 + c;
        float wv = 0.f;
        if (j >= 0 && j < out) {
            const LinTap t = lin_tap(j, in, scale);
            wv = (t.i0 == i ? t.w0 : 0.f) + (t.i1 == i ? t.w1 : 0.f);
        }
        r.w[c] = wv;
    }
    return r;
}

__device__ __forceinline__ void load4_f32(const float* p, float* o) { const float4 v = *(const float4*)p; o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
__device__ __forceinline__ void load4_f32(const bf16* p, float* o) {
    const uint2 v = *(const uint2*)p;
    o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u); o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
}

#ifndef CVAE_UP2X_TILE
#define CVAE_UP2X_TILE 1
#endif
#ifndef CVAE_UP2X_STREAM_BYTES
#define CVAE_UP2X_STREAM_BYTES ((int64_t)256 << 20)          // cvae_up2x_fwd outputs from this size on use nontemporal stores (decode sweep: 707 -> 490 us with the vector loads)
#endif

// Exact 2x taps.  lin_tap() gives, for an even output o = 2 i: (i - 1, i) with weights (0.25, 0.75) — except o = 0: (0, 1) with
// weights (1, 0) — and for an odd output o = 2 i + 1: (i, i + 1) with weights (0.75, 0.25), the upper index clamped to in - 1.
// With source indices clamped on load the same values come out of a FIXED pattern: even -> (v[i-1], v[i]) x (0.25, 0.75), or x (0, 1)
// at o = 0 (0*v0 + 1*v0 == 1*v0 + 0*v1); odd -> (v[i], v[i+1]) x (0.75, 0.25).  No per-output index arithmetic is left.
__device__ __forceinline__ float2 even_w(int o) { return o == 0 ? make_float2(0.f, 1.f) : make_float2(0.25f, 0.75f); }

// MODE 0 / 4: dst = up(src).  MODE 1: acc_out[block] = sum (up(src) - xin)^2.  MODE 3: MODE 1 and dst = t1[b][od][oh][x] = sum_ow Ww(ow -> x) (up(src) - xin).
// D == d (2D tensors) leaves the depth axis untouched.
// The ELBO's finish inside the forward launch (round 3; it was a launch of its own: ~4.7 us of cold misses in series for a 1-block kernel,
// profiles/r03_launch_gap.txt).  Every workgroup leaves its partial sum and counts itself in on `ticket`; the workgroup whose add came LAST sums the
// partials — in index order, as elbo_finish_kernel does, so the loss keeps its bits — adds the small terms and writes out4.  Hand-off per
// MI355X_MICROARCH.md ("Valid forms", first table row): the partial is ONE 4-byte agent-scope (sc1) atomic store by the lane that then drains it
// (s_waitcnt vmcnt(0)) and makes the agent-scope ticket add; the last arriver — told by the value its add returned — passes a workgroup barrier and
// reads every partial with agent-scope (sc1) atomic loads, never through its L1.  The last arriver resets the ticket (a replayed graph finds it zero)
// and, when asked, bumps a device step counter (the optimizer's: one more single-block launch gone).
#define CVAE_ELBO_TICKET_GROUPS 32
struct ElboFinish {
    unsigned* ticket;                                        // CVAE_ELBO_TICKET_WORDS words; null: no in-launch finish (the caller runs elbo_finish_kernel)
    const float *m_hat, *m, *mu, *logvar;
    float* out4;
    int* bump;
    float gamma;
    int n_m, n_z;
};
struct SmallBwd {                                            // the ELBO's small terms, ridden along the backward launch
    const float *m_hat, *m, *mu, *logvar;
    float *d_mhat, *dmu, *dlv;
    float gamma;
    int n_m, n_z, main_blocks;
};
template <typename T, int MODE>
__global__ __launch_bounds__(256) void up2x_block_kernel(const T* __restrict__ src, const float* __restrict__ xin, float* __restrict__ dst,
                                                         float* __restrict__ acc_out, const float* __restrict__ gout, float gscale,
                                                         int B, int d, int h, int w, int D, int H, int W, SmallBwd sb, ElboFinish fin) {
    constexpr int J0 = (MODE == 3) ? -1 : 0, NJ = (MODE == 3) ? 10 : 8;     // outputs along w per row: local j <-> ow = ow0 + J0 + j
    const bool sdz = D != d;
    const float sw = (float)w / (float)W;
    const int wg = w >> 2, n = B * d * h * wg;
    const int i0 = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
    const bool live = i0 < n;
    const int i = live ? i0 : n - 1;                         // idle lanes of the last block shadow the last item: the lane exchange below needs every lane
    float sse = 0.f;
    {
        int r = i;
        const int xg = r % wg; r /= wg;
        const int y = r % h; r /= h;
        const int z = r % d;
        const int b = r / d;
        const int x0 = 4 * xg, ow0 = 8 * xg;
        // The target rows this thread compares with (2 x 2 output rows of 8) are requested FIRST, all of them, so that they travel while the source rows are
        // loaded and interpolated: with the loads inside the row loop every row was its own round trip (5 dependent trips per thread in a one-round grid:
        // 22 us for 52 MB).  The two neighbours a row needs beyond its 8 columns (MODE 3) come from the adjacent lanes; lanes 0 / 63 fetch them.
        float4 xa_[2][2], xb_[2][2];
        float xl_[2][2], xr_[2][2];
        if (MODE == 1 || MODE == 3) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int bq = 0; bq < 2; ++bq) {
                    const int od = sdz ? 2 * z + a : z, oh = 2 * y + bq;
                    const size_t rowoff = ((size_t)(b * D + od) * H + oh) * W;
                    xa_[a][bq] = *(const float4*)(xin + rowoff + ow0);
                    xb_[a][bq] = *(const float4*)(xin + rowoff + ow0 + 4);
                    if (MODE == 3) {
                        const float el = xin[rowoff + max(ow0 - 1, 0)], er = xin[rowoff + min(ow0 + 8, W - 1)];      // unconditional, clamped (see the source rows below)
                        const float lft = __shfl_up(xb_[a][bq].w, 1), rgt = __shfl_down(xa_[a][bq].x, 1);
                        xl_[a][bq] = (ow0 > 0) ? (lane > 0 ? lft : el) : 0.f;
                        xr_[a][bq] = (ow0 + 8 < W) ? (lane < 63 ? rgt : er) : 0.f;
                    }
                }
        }
        // w-interpolated rows P[zr][yr][j] of the 3 x 3 source rows around (z, y): zr <-> z - 1 + zr, yr <-> y - 1 + yr (clamped)
        float P[3][3][NJ];
        const float2 ew0 = even_w(ow0);
#pragma unroll
        for (int zr = 0; zr < 3; ++zr) {
            const int zi = sdz ? min(max(z - 1 + zr, 0), d - 1) : z;
#pragma unroll
            for (int yr = 0; yr < 3; ++yr) {
                const int yi = min(max(y - 1 + yr, 0), h - 1);
                const T* row = src + ((size_t)(b * d + zi) * h + yi) * w;
                float v[6];                                  // source x0 - 1 .. x0 + 4 (clamped): one vector load, the two ends from the neighbouring lanes
                load4_f32(row + x0, v + 1);
                // the two ends: from the neighbouring lanes, except at the ends of the wave — fetched by EVERY lane from a clamped index and selected (a load inside
                // a lane-dependent branch makes the compiler wait for everything in flight at each of the nine rows)
                const float el = to_f32(row[max(x0 - 1, 0)]), er = to_f32(row[min(x0 + 4, w - 1)]);
                const float lft = __shfl_up(v[4], 1), rgt = __shfl_down(v[1], 1);
                v[0] = (xg == 0) ? v[1] : (lane > 0 ? lft : el);
                v[5] = (xg == wg - 1) ? v[4] : (lane < 63 ? rgt : er);
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int jj = J0 + j;                   // ow = ow0 + jj, jj in [-1, 8]
                    if (jj & 1) {                            // odd: i = x0 + (jj - 1) / 2 -> v[(jj - 1) / 2 + 1], v[.. + 2]
                        const int c = (jj + 1) / 2;          // jj = -1 -> c = 0; jj = 7 -> c = 4
                        P[zr][yr][j] = 0.75f * v[c] + 0.25f * v[c + 1];
                    } else {                                 // even: i = x0 + jj / 2 -> v[jj / 2], v[jj / 2 + 1]
                        const int c = jj / 2;
                        const float2 ww = (jj == 0) ? ew0 : make_float2(0.25f, 0.75f);
                        P[zr][yr][j] = ww.x * v[c] + ww.y * v[c + 1];
                    }
                }
            }
        }
        Win4 wx[4];
        if (MODE == 3) {
#pragma unroll
            for (int q = 0; q < 4; ++q) wx[q] = win4(x0 + q, w, W, sw, true);
        }
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            if (!live || (!sdz && a == 1)) break;
            const int od = sdz ? 2 * z + a : z;
            // depth taps: rows (a, a + 1) with weights (dw.x, dw.y); on an unstrided depth axis all three rows hold plane z and dw = (1, 0)
            const int zr0 = a;
            const float2 dw = !sdz ? make_float2(1.f, 0.f) : (a == 0 ? even_w(od) : make_float2(0.75f, 0.25f));
#pragma unroll
            for (int bq = 0; bq < 2; ++bq) {
                const int oh = 2 * y + bq;
                const float2 hw = (bq == 0) ? even_w(oh) : make_float2(0.75f, 0.25f);
                float o[NJ];
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const float lo = hw.x * P[zr0][bq][j] + hw.y * P[zr0][bq + 1][j];
                    const float hi = hw.x * P[zr0 + 1][bq][j] + hw.y * P[zr0 + 1][bq + 1][j];
                    o[j] = dw.x * lo + dw.y * hi;
                }
                const size_t rowoff = ((size_t)(b * D + od) * H + oh) * W;
                if (MODE == 0 || MODE == 4) {
                    const f32x4 oa = {o[0], o[1], o[2], o[3]}, ob = {o[4], o[5], o[6], o[7]};
                    if (MODE == 4) {                         // a volume far beyond the L2 / MALL is written past them
                        __builtin_nontemporal_store(oa, (f32x4*)(dst + rowoff + ow0));
                        __builtin_nontemporal_store(ob, (f32x4*)(dst + rowoff + ow0 + 4));
                    } else {
                        *(f32x4*)(dst + rowoff + ow0) = oa;
                        *(f32x4*)(dst + rowoff + ow0 + 4) = ob;
                    }
                } else {
                    const float4 xa = xa_[a][bq], xb = xb_[a][bq];
                    const float xv[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
                    if (MODE == 1) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) { const float df = o[j] - xv[j]; sse += df * df; }
                    } else {
                        float g[10];
                        g[0] = (ow0 > 0) ? o[0] - xl_[a][bq] : 0.f;
#pragma unroll
                        for (int j = 0; j < 8; ++j) { g[1 + j] = o[1 + j] - xv[j]; sse += g[1 + j] * g[1 + j]; }
                        g[9] = (ow0 + 8 < W) ? o[9] - xr_[a][bq] : 0.f;
                        float t1v[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {                // outputs 2 (x0 + q) - 1 + c  ->  local g index 2 q + c
                            float rs = 0.f;
#pragma unroll
                            for (int c = 0; c < 4; ++c) if (wx[q].w[c] != 0.f) rs += wx[q].w[c] * g[2 * q + c];
                            t1v[q] = rs;
                        }
                        *(float4*)(dst + ((size_t)(b * D + od) * H + oh) * w + x0) = make_float4(t1v[0], t1v[1], t1v[2], t1v[3]);
                    }
                }
            }
        }
    }
    if (MODE == 1 || MODE == 3) {
        __shared__ float red[4];
        __shared__ unsigned is_last;
        sse = wave_sum(sse);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sse;
        __syncthreads();
        if (!fin.ticket) {
            if (threadIdx.x == 0) acc_out[blockIdx.x] = red[0] + red[1] + red[2] + red[3];     // per-block partial: summed in a fixed order by elbo_finish_kernel
            return;
        }
        if (threadIdx.x == 0) {
            __hip_atomic_store(acc_out + blockIdx.x, red[0] + red[1] + red[2] + red[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the partial has left this CU before the ticket says so
            // Two-level ticket: the ~1000 workgroups of this one-round grid finish nearly together, and that many RETURNING adds on ONE word serialise at
            // the memory side (measured: the step 18 us LONGER than with the finish launch).  Workgroup b counts in on word b % 32 of 32 (each on a line
            // of its own); the last arriver of a word counts in on the top word; the last arriver there has, transitively, every partial behind it.
            constexpr unsigned G = CVAE_ELBO_TICKET_GROUPS;
            const unsigned grp = blockIdx.x % G, pop = gridDim.x / G + (grp < gridDim.x % G ? 1u : 0u), ngrp = gridDim.x < G ? gridDim.x : G;
            unsigned last = 0u;
            if (__hip_atomic_fetch_add(fin.ticket + 32 * grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == pop - 1)
                last = __hip_atomic_fetch_add(fin.ticket + 32 * G, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ngrp - 1 ? 1u : 0u;
            is_last = last;
        }
        __syncthreads();
        if (!is_last) return;
        // ---- the last arriver: elbo_finish_kernel's sums, thread for thread ----
        __shared__ float fr[3][4];
        float sr = 0.f, sm = 0.f, sk = 0.f;
        for (int i0 = threadIdx.x; i0 < (int)gridDim.x; i0 += 256 * 8) {      // 8 sc1 loads in flight per lane, then added in index order (elbo_finish_kernel's order)
            float pv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) pv[u] = (i0 + 256 * u < (int)gridDim.x) ? __hip_atomic_load(acc_out + i0 + 256 * u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) if (i0 + 256 * u < (int)gridDim.x) sr += pv[u];
        }
        for (int i = threadIdx.x; i < fin.n_m; i += 256) { const float df = fin.m_hat[i] - fin.m[i]; sm += df * df; }
        for (int i = threadIdx.x; i < fin.n_z; i += 256) sk += 1.f + fin.logvar[i] - fin.mu[i] * fin.mu[i] - expf(fin.logvar[i]);
        sr = wave_sum(sr); sm = wave_sum(sm); sk = wave_sum(sk);
        if ((threadIdx.x & 63) == 0) { fr[0][threadIdx.x >> 6] = sr; fr[1][threadIdx.x >> 6] = sm; fr[2][threadIdx.x >> 6] = sk; }
        __syncthreads();
        if (threadIdx.x == 0) {
            const float recon = fr[0][0] + fr[0][1] + fr[0][2] + fr[0][3], ml = fr[1][0] + fr[1][1] + fr[1][2] + fr[1][3];
            const float kld = -0.5f * (fr[2][0] + fr[2][1] + fr[2][2] + fr[2][3]);
            fin.out4[0] = recon + fin.gamma * ml + kld; fin.out4[1] = recon; fin.out4[2] = ml; fin.out4[3] = kld;
            if (fin.bump) *fin.bump += 1;
        }
        if (threadIdx.x <= CVAE_ELBO_TICKET_GROUPS) __hip_atomic_store(fin.ticket + 32 * threadIdx.x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // every word back to zero
    }
}

// The large-output form of MODE 0 for 3D volumes (the counterfactual sweep resizes 240 x 64^3 to 128^3: 2 GB of fp32): a workgroup stages the
// 4 x 10 x 66 source values around a 2 x 8 x 64 source tile in LDS once (vector loads, edges clamped), every thread then produces the 2 x 2 output
// rows of two (z, y) source rows for one group of 4 output columns — one 16-byte store per row, 64 lanes = two complete 512-byte row segments per
// instruction.  Same taps and weights as up2x_block_kernel (even outputs (0.25, 0.75) of (i - 1, i), (0, 1) at o = 0; odd (0.75, 0.25) of (i, i + 1)).
// Measured on the sweep (2 GB out): 503 -> ~350 us, the rate of a plain 2 GB fill on this GPU (345-360 us, tools/probes/hbm_probe.hip); outputs of
// 256 MB and more are stored nontemporal (-20 us there).
template <typename T, bool NT>
__global__ __launch_bounds__(256) void up2x_tile_kernel(const T* __restrict__ src, float* __restrict__ dst, int d, int h, int w) {
    constexpr int TZ = 2, TY = 8, TX = 64, PZ = TZ + 2, PY = TY + 2, PITCH = TX + 4;     // 66 used; 68 keeps rows 16-byte aligned
    constexpr int VEC = 16 / sizeof(T), SEGS = TX / VEC;
    __shared__ __attribute__((aligned(16))) float tile[PZ * PY * PITCH];
    const int t = threadIdx.x;
    int blk = blockIdx.x;
    const int xt = blk % (w / TX); blk /= (w / TX);
    const int yt = blk % (h / TY); blk /= (h / TY);
    const int zt = blk % (d / TZ), b = blk / (d / TZ);
    const int x0 = xt * TX, y0 = yt * TY, z0 = zt * TZ;
    const T* sb = src + (size_t)b * d * h * w;
    // interior: PZ * PY rows of TX values as 16-byte vectors; edges: the two clamped neighbours of every row
    for (int i = t; i < PZ * PY * SEGS; i += 256) {
        const int seg = i % SEGS, row = i / SEGS, py = row % PY, pz = row / PY;
        const int zi = min(max(z0 - 1 + pz, 0), d - 1), yi = min(max(y0 - 1 + py, 0), h - 1);
        float v[VEC];
        if constexpr (sizeof(T) == 2) {
            const uint4 q = *(const uint4*)(sb + ((size_t)zi * h + yi) * w + x0 + seg * VEC);
            const uint32_t u[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[2 * e] = __uint_as_float(u[e] << 16); v[2 * e + 1] = __uint_as_float(u[e] & 0xffff0000u); }
        } else {
            const float4 q = *(const float4*)(sb + ((size_t)zi * h + yi) * w + x0 + seg * VEC);
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        }
        float* o = tile + row * PITCH + 1 + seg * VEC;
#pragma unroll
        for (int e = 0; e < VEC; ++e) o[e] = v[e];
    }
    if (t < 2 * PZ * PY) {
        const int side = t & 1, row = t >> 1, py = row % PY, pz = row / PY;
        const int zi = min(max(z0 - 1 + pz, 0), d - 1), yi = min(max(y0 - 1 + py, 0), h - 1);
        const int xi = side ? min(x0 + TX, w - 1) : max(x0 - 1, 0);
        tile[row * PITCH + (side ? TX + 1 : 0)] = to_f32(sb[((size_t)zi * h + yi) * w + xi]);
    }
    __syncthreads();
    const int xq = t & 31, yl = t >> 5;                       // output columns 4 (x0 / 2 .. ) : source x0 + 2 xq - 1 .. x0 + 2 xq + 2 = tile columns 2 xq .. 2 xq + 3
    const int ow0 = 2 * x0 + 4 * xq;
    const float2 ew0 = even_w(ow0);
    float P[PZ][3][4];
#pragma unroll
    for (int pz = 0; pz < PZ; ++pz)
#pragma unroll
        for (int yr = 0; yr < 3; ++yr) {
            const float* rp = tile + (pz * PY + yl + yr) * PITCH + 2 * xq;
            const float2 va = *(const float2*)rp, vb = *(const float2*)(rp + 2);
            const float v[4] = {va.x, va.y, vb.x, vb.y};
            P[pz][yr][0] = ew0.x * v[0] + ew0.y * v[1];
            P[pz][yr][1] = 0.75f * v[1] + 0.25f * v[2];
            P[pz][yr][2] = 0.25f * v[1] + 0.75f * v[2];
            P[pz][yr][3] = 0.75f * v[2] + 0.25f * v[3];
        }
    const int D = 2 * d, H = 2 * h, W = 2 * w;
#pragma unroll
    for (int zl = 0; zl < TZ; ++zl)
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int od = 2 * (z0 + zl) + a;
            const float2 dw = (a == 0) ? even_w(od) : make_float2(0.75f, 0.25f);
#pragma unroll
            for (int bq = 0; bq < 2; ++bq) {
                const int oh = 2 * (y0 + yl) + bq;
                const float2 hw = (bq == 0) ? even_w(oh) : make_float2(0.75f, 0.25f);
                f32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float lo = hw.x * P[zl + a][bq][j] + hw.y * P[zl + a][bq + 1][j];
                    const float hi = hw.x * P[zl + a + 1][bq][j] + hw.y * P[zl + a + 1][bq + 1][j];
                    o[j] = dw.x * lo + dw.y * hi;
                }
                f32x4* op = (f32x4*)(dst + (((size_t)b * D + od) * H + oh) * W + ow0);
                if (NT) __builtin_nontemporal_store(o, op); else *op = o;
            }
        }
}

// d src[b][z][y][x .. x + 3] = gs sum_{od, oh} Wd(od -> z) Wh(oh -> y) t1[b][od][oh][x ..], gs = 2 * (*gout) (1 when null).  One extra block
// (blockIdx.x == sb.main_blocks) writes the gradients of the ELBO's small terms.
template <typename T>
__global__ __launch_bounds__(256) void up2x_bwd_b_kernel(const float* __restrict__ t1, T* __restrict__ dsrc, const float* __restrict__ gout, int B, int d, int h, int w,
                                                         int D, int H, SmallBwd sb) {
    const float g = gout ? *gout : 1.f;
    if (sb.dmu && (int)blockIdx.x == sb.main_blocks) {
        // d m_hat = 2 g gamma (m_hat - m); d mu = g mu; d logvar = 0.5 g (exp(logvar) - 1)
        for (int i = threadIdx.x; i < sb.n_m; i += 256) sb.d_mhat[i] = 2.f * g * sb.gamma * (sb.m_hat[i] - sb.m[i]);
        for (int i = threadIdx.x; i < sb.n_z; i += 256) { sb.dmu[i] = g * sb.mu[i]; sb.dlv[i] = 0.5f * g * (expf(sb.logvar[i]) - 1.f); }
        return;
    }
    const float gs = 2.f * g;
    const float sd = (float)d / (float)D, sh = (float)h / (float)H;
    const int wg = w >> 2, n = B * d * h * wg;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int r = i;
    const int xg = r % wg; r /= wg;
    const int y = r % h; r /= h;
    const int z = r % d;
    const int b = r / d;
    const Win4 wz = win4(z, d, D, sd, D != d), wy = win4(y, h, H, sh, true);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    // all 16 rows of t1 this voxel group reads are requested first, unconditionally, from clamped row indices (a row with weight 0 — beyond the volume, or
    // the unstrided depth axis — is loaded and not used): with the loads inside the weight tests every row was a dependent round trip of a one-round grid
    float4 tv[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int bb = 0; bb < 4; ++bb)
            tv[a][bb] = *(const float4*)(t1 + ((size_t)(b * D + min(max(wz.j0 + a, 0), D - 1)) * H + min(max(wy.j0 + bb, 0), H - 1)) * w + 4 * xg);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        if (wz.w[a] == 0.f) continue;
        float pl[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
            if (wy.w[bb] == 0.f) continue;
            const float4 v = tv[a][bb];
            pl[0] += wy.w[bb] * v.x; pl[1] += wy.w[bb] * v.y; pl[2] += wy.w[bb] * v.z; pl[3] += wy.w[bb] * v.w;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] += wz.w[a] * pl[q];
    }
    T* o = dsrc + ((size_t)(b * d + z) * h + y) * w + 4 * xg;
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = from_f32<T>(gs * acc[q]);
}

// out4 = {loss, recon, m_loss, kld}: recon = sum of the per-block partial sums of up2x_block_kernel<MODE 1> (fixed order: the loss is
// bit-reproducible), m_loss = sum (m_hat - m)^2, kld = -0.5 sum (1 + logvar - mu^2 - exp(logvar)), loss = recon + gamma m_loss + kld.
__global__ __launch_bounds__(256) void elbo_finish_kernel(const float* __restrict__ partial, int n_partial, const float* __restrict__ m_hat,
                                                          const float* __restrict__ m, const float* __restrict__ mu, const float* __restrict__ logvar,
                                                          float gamma, float* __restrict__ out4, int n_m, int n_z) {
    __shared__ float red[3][4];
    float sr = 0.f, sm = 0.f, sk = 0.f;
    for (int i = threadIdx.x; i < n_partial; i += 256) sr += partial[i];
    for (int i = threadIdx.x; i < n_m; i += 256) { const float df = m_hat[i] - m[i]; sm += df * df; }
    for (int i = threadIdx.x; i < n_z; i += 256) sk += 1.f + logvar[i] - mu[i] * mu[i] - expf(logvar[i]);
    sr = wave_sum(sr); sm = wave_sum(sm); sk = wave_sum(sk);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = sr; red[1][threadIdx.x >> 6] = sm; red[2][threadIdx.x >> 6] = sk; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float recon = red[0][0] + red[0][1] + red[0][2] + red[0][3], ml = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        const float kld = -0.5f * (red[2][0] + red[2][1] + red[2][2] + red[2][3]);
        out4[0] = recon + gamma * ml + kld; out4[1] = recon; out4[2] = ml; out4[3] = kld;
    }
}

bool up2x_ok(int64_t B, int64_t d, int64_t h, int64_t w, int64_t D, int64_t H, int64_t W) {
    return B > 0 && d > 0 && h > 0 && w > 0 && (D == 2 * d || (D == 1 && d == 1)) && H == 2 * h && W == 2 * w && (w % 4) == 0 && B * D * H * W < (int64_t)1 << 31;
}

}  // namespace

extern "C" int cvae_up2x_supported(int64_t B, int64_t d, int64_t h, int64_t w, int64_t D, int64_t H, int64_t W) { return up2x_ok(B, d, h, w, D, H, W) ? 1 : 0; }

extern "C" int cvae_up2x_fwd(const void* src, float* dst, int64_t B, int64_t d, int64_t h, int64_t w, int64_t D, int64_t H, int64_t W, int dtype, void* stream) {
    if (!up2x_ok(B, d, h, w, D, H, W)) return CVAE_E_UNSUPPORTED;
    if (!src || !dst) return CVAE_E_NULLPTR;
    const unsigned grid = (unsigned)((B * d * h * (w / 4) + 255) / 256);
    const bool stream_out = B * D * H * W * 4 >= CVAE_UP2X_STREAM_BYTES;
#if CVAE_UP2X_TILE
    if (D == 2 * d && d % 2 == 0 && h % 8 == 0 && w % 64 == 0 && (dtype == CVAE_BF16 || dtype == CVAE_F32)) {      // rows of 64: the LDS-tiled form
        const dim3 tgrid((unsigned)(B * (d / 2) * (h / 8) * (w / 64)));
        hipStream_t st = (hipStream_t)stream;
#define UP2X_TILE(T, NT) hipLaunchKernelGGL((up2x_tile_kernel<T, NT>), tgrid, dim3(256), 0, st, (const T*)src, dst, (int)d, (int)h, (int)w)
        if (dtype == CVAE_BF16) { if (stream_out) UP2X_TILE(bf16, true); else UP2X_TILE(bf16, false); }
        else { if (stream_out) UP2X_TILE(float, true); else UP2X_TILE(float, false); }
#undef UP2X_TILE
        CVAE_CHECK_LAUNCH();
        return CVAE_OK;
    }
#endif
#define UP2X_FWD(T, MODE) hipLaunchKernelGGL((up2x_block_kernel<T, MODE>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const T*)src, nullptr, dst, nullptr, nullptr, 0.f, (int)B, (int)d, (int)h, (int)w, (int)D, (int)H, (int)W, SmallBwd{}, ElboFinish{})
    if (dtype == CVAE_BF16) { if (stream_out) UP2X_FWD(bf16, 4); else UP2X_FWD(bf16, 0); }
    else if (dtype == CVAE_F32) { if (stream_out) UP2X_FWD(float, 4); else UP2X_FWD(float, 0); }
#undef UP2X_FWD
    else return CVAE_E_DTYPE;
    CVAE_CHECK_LAUNCH();
    return CVAE_OK;
}

extern "C" int64_t cvae_elbo_up2x_partials(int64_t B, int64_t d, int64_t h, int64_t w) { return (B * d * h * (w / 4) + 255) / 256; }

extern "C" int cvae_elbo_up2x_fwd(const void* src, const float* x, const float* m_hat, const float* m, const float* mu, const float* logvar, float gamma,
                                  float* out4, float* partial, float* t1, void* ticket, int* bump, int64_t B, int64_t d, int64_t h, int64_t w, int64_t D, int64_t H, int64_t W,
                                  int64_t n_m, int64_t n_z, int dtype, void* stream) {
    if (!up2x_ok(B, d, h, w, D, H, W) || n_m < 0 || n_z < 0 || n_m > (1 << 24) || n_z > (1 << 24)) return CVAE_E_UNSUPPORTED;
    if (!src || !x || !m_hat || !m || !mu || !logvar || !out4 || !partial) return CVAE_E_NULLPTR;
    if (dtype != CVAE_BF16 && dtype != CVAE_F32) return CVAE_E_DTYPE;
    if (bump && !ticket) return CVAE_E_NULLPTR;              // the counter rides on the in-launch finish
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)cvae_elbo_up2x_partials(B, d, h, w);
    const ElboFinish fin{(unsigned*)ticket, m_hat, m, mu, logvar, out4, bump, gamma, (int)n_m, (int)n_z};
#define ELBO_FWD(T, MODE) hipLaunchKernelGGL((up2x_block_kernel<T, MODE>), dim3(grid), dim3(256), 0, st, (const T*)src, x, t1, partial, nullptr, 0.f, (int)B, (int)d, (int)h, (int)w, (int)D, (int)H, (int)W, SmallBwd{}, fin)
    if (dtype == CVAE_BF16) { if (t1) ELBO_FWD(bf16, 3); else ELBO_FWD(bf16, 1); }
    else { if (t1) ELBO_FWD(float, 3); else ELBO_FWD(float, 1); }
#undef ELBO_FWD
    CVAE_CHECK_LAUNCH();
    if (!ticket) {
        hipLaunchKernelGGL(elbo_finish_kernel, dim3(1), dim3(256), 0, st, (const float*)partial, (int)grid, m_hat, m, mu, logvar, gamma, out4, (int)n_m, (int)n_z);
        CVAE_CHECK_LAUNCH();
    }
    return CVAE_OK;
}

extern "C" int cvae_elbo_up2x_bwd(const float* t1, const float* m_hat, const float* m, const float* mu, const float* logvar, float gamma, const float* g_loss,
                                  void* dsrc, float* d_mhat, float* dmu, float* dlv, int64_t B, int64_t d, int64_t h, int64_t w, int64_t D, int64_t H, int64_t W,
                                  int64_t n_m, int64_t n_z, int dtype, void* stream) {
    if (!up2x_ok(B, d, h, w, D, H, W) || n_m < 0 || n_z < 0 || n_m > (1 << 24) || n_z > (1 << 24)) return CVAE_E_UNSUPPORTED;
    if (!t1 || !m_hat || !m || !mu || !logvar || !dsrc || !d_mhat || !dmu || !dlv) return CVAE_E_NULLPTR;
    if (dtype != CVAE_BF16 && dtype != CVAE_F32) return CVAE_E_DTYPE;
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)((B * d * h * (w / 4) + 255) / 256);
    const SmallBwd sb{m_hat, m, mu, logvar, d_mhat, dmu, dlv, gamma, (int)n_m, (int)n_z, (int)grid};
    if (dtype == CVAE_BF16) hipLaunchKernelGGL(up2x_bwd_b_kernel<bf16>, dim3(grid + 1), dim3(256), 0, st, t1, (bf16*)dsrc, g_loss, (int)B, (int)d, (int)h, (int)w, (int)D, (int)H, sb);
    else hipLaunchKernelGGL(up2x_bwd_b_kernel<float>, dim3(grid + 1), dim3(256), 0, st, t1, (float*)dsrc, g_loss, (int)B, (int)d, (int)h, (int)w, (int)D, (int)H, sb);
    CVAE_CHECK_LAUNCH();
    return CVAE_OK;
}
