// conv_c1.hip — the single-channel ends of the network (image -> 32 features, 32 features -> image): Cl == 1.
// These layers are HBM-bound on paper (arithmetic intensity ~50 FLOP/B); what actually bounded them was access width — a tile
// 8 positions wide reads 64-96 contiguous bytes per image row — so every kernel here tiles WIDE IN X (32 positions, 16 for the transposed conv)
// and a workgroup walks several tiles with the next halo in flight.
//   down   (image -> S)  : down_c1_vec_kernel (bf16 S; image read as stored, fp32 or bf16, in 16-byte vectors) / down_c1_kernel (element loads)
//   up     (S -> image)  : up_c1_mfma_kernel (bf16: v_mfma_f32_16x16x32_bf16 over the 3^nd neighbourhood; up_c1_mfma_walk_kernel walks z columns of tiles
//                          on large 3D launches and keeps the shared halo planes in LDS) / up_c1_kernel (scalar form, fp32)
//   wgrad                : wgrad_c1_kernel, [Cs x taps] = S^T . im2col(L) with K = positions on the MFMA (bf16 32x32x16 with transposing LDS reads, fp32
//                          32x32x2); persistent workgroups leave slabs of partial sums, wgrad_c1_finish_kernel adds them in index order (no atomics)
#include "common.h"

namespace {

template <int ND> struct TileC1;       // 256 output positions per workgroup
template <> struct TileC1<3> { static constexpr int TD = 4, TH = 8, TW = 8; };
template <> struct TileC1<2> { static constexpr int TD = 1, TH = 16, TW = 16; };

// -------------------------------------------------------------------------------------- down, Cl == 1
// S[pos][cs] = act(bias + sum_tap L[2 pos - 1 + k] W[cs][tap]) as a [256 positions x 64 taps] x [64 taps x 32 cs] MFMA product per
// workgroup: the A fragments are gathered straight out of the 1-channel halo tile in LDS (im2col is never written), the
// whole weight lives in registers as B fragments for the lifetime of the workgroup.  What remains is streaming:
// ~6.5 KB of halo in, 16 KB of outputs out per tile.
template <typename T> struct C1Ops;
template <> struct C1Ops<bf16> {
    // k-block = one depth tap (16 taps = kh x kw); lane (r, h) holds k = 8h + j  <->  kh = 2h + (j >> 2), kw = j & 3
    static constexpr int KB_TAPS = 16;
    struct BFrag { bf16x8 v; };
    static __device__ __forceinline__ void load_b(BFrag& f, const float* wrow, int kb, int h) {       // wrow = W[cs = r][.]
        const float4 a = *(const float4*)(wrow + kb * 16 + 8 * h), b = *(const float4*)(wrow + kb * 16 + 8 * h + 4);
        union { uint32_t u[4]; bf16x8 v; } o;
        o.u[0] = pack2_bf16(a.x, a.y); o.u[1] = pack2_bf16(a.z, a.w); o.u[2] = pack2_bf16(b.x, b.y); o.u[3] = pack2_bf16(b.z, b.w);
        f.v = o.v;
    }
    template <int IW>
    static __device__ __forceinline__ void mma_block(f32x16& acc, const bf16* halo, int pb, int h, const BFrag& bfr) {
        // two rows (kh = 2h, 2h+1) of 4 consecutive bf16: 8-byte groups at 4-byte alignment -> 2 x ds_read_b32 each
        const uint32_t* p0 = (const uint32_t*)(halo + pb + (2 * h) * IW);
        const uint32_t* p1 = (const uint32_t*)(halo + pb + (2 * h + 1) * IW);
        union { uint32_t u[4]; bf16x8 v; } a;
        a.u[0] = p0[0]; a.u[1] = p0[1]; a.u[2] = p1[0]; a.u[3] = p1[1];
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr.v, a.v, acc, 0, 0, 0);       // D = W x im2col^T: rows = channels, columns = positions
    }
};
template <> struct C1Ops<float> {
    // fp32: 32x32x2 MFMA, lane (r, h) feeds tap 2s + h at step s; a k-block is again 16 taps = 8 steps
    static constexpr int KB_TAPS = 16;
    struct BFrag { float v[8]; };
    static __device__ __forceinline__ void load_b(BFrag& f, const float* wrow, int kb, int h) {
#pragma unroll
        for (int s = 0; s < 8; ++s) f.v[s] = wrow[kb * 16 + 2 * s + h];
    }
    template <int IW>
    static __device__ __forceinline__ void mma_block(f32x16& acc, const float* halo, int pb, int h, const BFrag& bfr) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {                     // tap (within the block) = 2s + h: kh = s >> 1, kw = 2 (s & 1) + h
            const float a = halo[pb + (s >> 1) * IW + 2 * (s & 1) + h];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bfr.v[s], a, acc, 0, 0, 0);
        }
    }
};

template <typename T, int ND, int CS, int EPI>
__global__ __launch_bounds__(256) void down_c1_kernel(const T* __restrict__ L, const float* __restrict__ w, const float* __restrict__ bias,
                                                      const T* __restrict__ mask, T* __restrict__ S, int sd, int sh, int sw, int ld, int lh, int lw,
                                                      int tiles_h, int tiles_w, int act) {
    static_assert(CS == 32, "one 32-wide N sub-tile");
    using TL = TileC1<ND>;
    using OP = C1Ops<T>;
    constexpr int TD = TL::TD, TH = TL::TH, TW = TL::TW;
    constexpr int ID = (ND == 3) ? 2 * TD + 2 : 1, IH = 2 * TH + 2, IW = 2 * TW + 2;
    constexpr int NPOS = ID * IH * IW, TAPS = (ND == 3) ? 64 : 16, NKB = TAPS / 16;
    constexpr int HN = (NPOS + 255) / 256;
    __shared__ __attribute__((aligned(16))) T halo[NPOS + 8];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r = lane & 31, h = lane >> 5, b = blockIdx.z;
    int tile = blockIdx.x;
    const int tw_i = tile % tiles_w; tile /= tiles_w;
    const int th_i = tile % tiles_h; tile /= tiles_h;
    const int o0d = tile * TD, o0h = th_i * TH, o0w = tw_i * TW;
    // ---- all halo loads in flight at once, weights into B fragments meanwhile ----
    T hv[HN];
#pragma unroll
    for (int i = 0; i < HN; ++i) {
        const int pos = t + i * 256;
        const int x = pos % IW, y = pos / IW % IH, z = pos / (IW * IH);
        const int gz = (ND == 3) ? 2 * o0d - 1 + z : 0, gy = 2 * o0h - 1 + y, gx = 2 * o0w - 1 + x;
        const bool ok = (pos < NPOS) & (gz >= 0) & (gz < ld) & (gy >= 0) & (gy < lh) & (gx >= 0) & (gx < lw);
        const T v = L[(((size_t)b * ld + min(max(gz, 0), ld - 1)) * lh + min(max(gy, 0), lh - 1)) * lw + min(max(gx, 0), lw - 1)];    // clamped: unconditional load
        hv[i] = ok ? v : from_f32<T>(0.f);
    }
    typename OP::BFrag bfr[NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) OP::load_b(bfr[kb], w + (size_t)r * TAPS, kb, h);
#pragma unroll
    for (int i = 0; i < HN; ++i) if (t + i * 256 < NPOS) halo[t + i * 256] = hv[i];
    __syncthreads();
    float bv[2][8];                                          // bias of this lane's two 8-channel pieces (channels 16 j + 8 h + q)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
        if (bias) { b0 = *(const float4*)(bias + 16 * j + 8 * h); b1 = *(const float4*)(bias + 16 * j + 8 * h + 4); }
        bv[j][0] = b0.x; bv[j][1] = b0.y; bv[j][2] = b0.z; bv[j][3] = b0.w; bv[j][4] = b1.x; bv[j][5] = b1.y; bv[j][6] = b1.z; bv[j][7] = b1.w;
    }
    // ---- each wave: 2 sub-tiles of 32 positions ----
#pragma unroll
    for (int ms = 0; ms < 2; ++ms) {
        const int m = (wave * 2 + ms) * 32 + r;
        const int w0 = m % TW, hh = m / TW % TH, d = m / (TW * TH);
        const int pb = ((2 * d) * IH + 2 * hh) * IW + 2 * w0;
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) OP::template mma_block<IW>(acc, halo, pb + kb * IH * IW, h, bfr[kb]);
        // D rows are channels, columns positions: lane (r, h) holds channels (e & 3) + 8 (e >> 2) + 4 h of position r.  Two
        // v_permlane32_swap per register pair regroup them into channels 8h..8h+7 and 16+8h..23+8h: 16-byte stores and mask loads.
        const int mo = (wave * 2 + ms) * 32 + r;
        const int ow = o0w + mo % TW, oh = o0h + mo / TW % TH, od = o0d + mo / (TW * TH);
        const bool ok = od < sd && oh < sh && ow < sw;
        const size_t pidx = ((((size_t)b * sd + od) * sh + oh) * sw + ow) * CS;
        float v[2][8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const auto lo = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[i]), __float_as_uint(acc[4 + i]), false, false);
            const auto hi = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[8 + i]), __float_as_uint(acc[12 + i]), false, false);
            v[0][i] = __uint_as_float(lo[0]); v[0][4 + i] = __uint_as_float(lo[1]);
            v[1][i] = __uint_as_float(hi[0]); v[1][4 + i] = __uint_as_float(hi[1]);
        }
        if (ok) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int c = 16 * j + 8 * h;
                __attribute__((aligned(16))) T mv[8], ov[8];
                constexpr int NU = (8 * sizeof(T)) / 16;
                if (mask) {
#pragma unroll
                    for (int u = 0; u < NU; ++u) ((uint4*)mv)[u] = ((const uint4*)(mask + pidx + c))[u];
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    float x = apply_act_t<EPI>(v[j][q] + bv[j][q], act);
                    if (mask && !(to_f32(mv[q]) > 0.f)) x = 0.f;
                    ov[q] = from_f32<T>(x);
                }
#pragma unroll
                for (int u = 0; u < NU; ++u) ((uint4*)(S + pidx + c))[u] = ((const uint4*)ov)[u];
            }
        }
    }
}

// -------------------------------------------------------------------------------------- down, Cl == 1, image read in its own dtype
// The same product for a bf16 S with the 1-channel image read as it is stored (fp32 for the network input: no separate cast pass; bf16 for
// a gradient image) in 16-BYTE pieces: a halo row is the aligned window [2 o0w - EPV, 2 o0w + 2 TW + EPV) of EPV-element vectors (EPV = 4
// fp32 / 8 bf16), i.e. 6 (4) coalesced loads per row instead of 18 two-byte ones, converted to bf16 on the way into LDS.  The tap window of
// an output starts at an ODD element of that row image, so a lane reads 3 aligned dwords per row and funnel-shifts (v_alignbit) the 4
// bf16 it needs out of them.  Needs lw % EPV == 0 and a 16-byte aligned image (cvae_conv_image_supported).
// The layer moves 100 MB (128^3, B = 4) for 4 GFLOP, but its first form issued ~1000 instructions per wave for 8 MFMAs (64 positions) and ran at
// the VALU issue rate (45 us in the step): so MS = 4 sub-tiles of 32 positions per wave (512 per workgroup: the weight fragments, the bias and
// the staging index arithmetic are paid once per 128 positions of a wave), the bias rides in as the accumulators' initial value, the ReLU-mask
// code exists only in the MASKED instantiation, and the staging loop steps its (row, vector) coordinates instead of dividing.
// 3D tile of the single-channel weight gradient: 2 x 2 x 32 positions — its image rows are 72 contiguous elements, not 24
#define C1W_TD3 2
#define C1W_TH3 2
#define C1W_TW3 32
#ifndef CVAE_C1_MAX_WG
#define CVAE_C1_MAX_WG 1024
#endif
#ifndef CVAE_C1U_WALK
#define CVAE_C1U_WALK 1                 // up, Cl == 1, 3D: walk z columns when the launch has enough tiles
#endif
#ifndef CVAE_C1U_WALK_MIN_UNITS
#define CVAE_C1U_WALK_MIN_UNITS 1024
#endif
template <int ND, int MS> struct TileC1V;
template <int MS> struct TileC1V<3, MS> { static constexpr int TD = MS / 2, TH = 8, TW = 32; };   // wide in x: a halo row is 160-288 contiguous bytes, an output row 2 KB
template <int MS> struct TileC1V<2, MS> { static constexpr int TD = 1, TH = 8 * MS, TW = 16; };

// SIDE8 (the first layer of a training forward whose next conv runs on fp8 operands, causal_vae_amd/fp8.py): S is left a second time as fp8 (e4m3)
// codes of S * f8.dscale[0] (f8.dscale -> ONE device float, 1 / s_S) in f8.out8, and max |S| is recorded in f8.amax (common.h).
template <typename TL, int ND, int EPI, bool MASKED, int MS, bool SIDE8 = false>
__global__ __launch_bounds__(256) void down_c1_vec_kernel(const TL* __restrict__ L, const float* __restrict__ w, const float* __restrict__ bias,
                                                          const bf16* __restrict__ mask, bf16* __restrict__ S, int sd, int sh, int sw, int ld, int lh, int lw,
                                                          int tiles_d, int tiles_h, int tiles_w, int ntiles, int act, F8Side f8) {
    using TLE = TileC1V<ND, MS>;
    using OP = C1Ops<bf16>;
    constexpr int CS = 32;
    constexpr int TD = TLE::TD, TH = TLE::TH, TW = TLE::TW;
    static_assert(TD * TH * TW == 128 * MS, "4 waves x MS sub-tiles of 32 positions");
    constexpr int EPV = 16 / sizeof(TL), NV = (2 * TW + 2 * EPV) / EPV, IWP = NV * EPV;
    constexpr int ID = (ND == 3) ? 2 * TD + 2 : 1, IH = 2 * TH + 2, NROW = ID * IH, NVEC = NROW * NV;
    constexpr int TAPS = (ND == 3) ? 64 : 16, NKB = TAPS / 16, HN = (NVEC + 255) / 256;
    __shared__ __attribute__((aligned(16))) bf16 halo[NROW * IWP + 8];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r = lane & 31, h = lane >> 5;
    // A workgroup walks tiles blockIdx.x, + gridDim.x, ...: the halo vectors of the NEXT tile are requested before the current tile is multiplied
    // and stored.  With one tile per workgroup the whole chip read, then multiplied, then wrote in lockstep (one round of workgroups): 8 us of
    // reads, then 16 us of writes, neither using the other's HBM time (kbench: 34.7 us for 84 MB).
    uint4 hv[HN];
    auto issue_loads = [&](int tile) {
        const int tw_i = tile % tiles_w; tile /= tiles_w;
        const int th_i = tile % tiles_h; tile /= tiles_h;
        const int td_i = tile % tiles_d, b = tile / tiles_d;
        const int o0d = td_i * TD, o0h = th_i * TH, o0w = tw_i * TW;
        const int gx0 = 2 * o0w - EPV;
        constexpr int DJ = 256 % NV, DROW = 256 / NV, DY = DROW % IH, DZ = DROW / IH;
        int j = t % NV, row = t / NV;
        int y = row % IH, z = row / IH;
        const TL* Lb = L + (size_t)b * ld * lh * lw;
#pragma unroll
        for (int i = 0; i < HN; ++i) {                         // vector t + 256 i = (row, j): stepped, not divided
            const int gz = (ND == 3) ? 2 * o0d - 1 + z : 0, gy = 2 * o0h - 1 + y, gx = gx0 + j * EPV;
            const bool ok = (z < ID) & (gz >= 0) & (gz < ld) & (gy >= 0) & (gy < lh) & (gx >= 0) & (gx < lw);      // lw % EPV == 0: a vector is inside or outside as a whole
            const uint4 v = *(const uint4*)(Lb + ((size_t)min(max(gz, 0), ld - 1) * lh + min(max(gy, 0), lh - 1)) * lw + min(max(gx, 0), lw - EPV));   // clamped: unconditional load
            hv[i] = ok ? v : make_uint4(0u, 0u, 0u, 0u);
            j += DJ; if (j >= NV) { j -= NV; y += 1; }
            y += DY; if (y >= IH) { y -= IH; z += 1; }
            if (y >= IH) { y -= IH; z += 1; }
            z += DZ;
        }
    };
    auto store_lds = [&]() {
#pragma unroll
        for (int i = 0; i < HN; ++i) {
            const int it = t + i * 256;
            if (it < NVEC) {
                if constexpr (sizeof(TL) == 4) {
                    const float* f = (const float*)&hv[i];
                    *(uint2*)(halo + it * EPV) = make_uint2(pack2_bf16(f[0], f[1]), pack2_bf16(f[2], f[3]));           // row * IWP + j * EPV == it * EPV
                } else {
                    *(uint4*)(halo + it * EPV) = hv[i];
                }
            }
        }
    };
    int tile = blockIdx.x;
    issue_loads(tile);
    float amx = 0.f;
    const float o8s = (SIDE8 && f8.out8) ? f8.dscale[0] : 1.f;
    typename OP::BFrag bfr[NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) OP::load_b(bfr[kb], w + (size_t)r * TAPS, kb, h);
    // bias as the accumulators' initial value: D row (channel) of register e on lane (., h) is (e & 3) + 8 (e >> 2) + 4 h
    float4 bq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) bq[q] = bias ? *(const float4*)(bias + 8 * q + 4 * h) : make_float4(0.f, 0.f, 0.f, 0.f);
    store_lds();
    __syncthreads();
    while (true) {
        const int next = tile + (int)gridDim.x;
        const bool has_next = next < ntiles;
        if (has_next) issue_loads(next);
        int tt = tile;
        const int tw_i = tt % tiles_w; tt /= tiles_w;
        const int th_i = tt % tiles_h; tt /= tiles_h;
        const int td_i = tt % tiles_d, b = tt / tiles_d;
        const int o0d = td_i * TD, o0h = th_i * TH, o0w = tw_i * TW;
        const size_t tile_org = ((((size_t)b * sd + o0d) * sh + o0h) * sw + o0w) * CS;
#pragma unroll
        for (int ms = 0; ms < MS; ++ms) {
            const int m = (wave * MS + ms) * 32 + r;
            const int w0 = m % TW, hh = m / TW % TH, d = m / (TW * TH);
            // element index of the aligned dword that holds tap kw = 0 in its upper half: x = 2 w0 - 1 - gx0 = 2 w0 + EPV - 1 (odd)
            const int pb = ((2 * d) * IH + 2 * hh) * IWP + 2 * w0 + EPV - 2;
            f32x16 acc;
#pragma unroll
            for (int q = 0; q < 4; ++q) { acc[4 * q] = bq[q].x; acc[4 * q + 1] = bq[q].y; acc[4 * q + 2] = bq[q].z; acc[4 * q + 3] = bq[q].w; }
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                union { uint32_t u[4]; bf16x8 v; } a;
#pragma unroll
                for (int rr = 0; rr < 2; ++rr) {                 // rows kh = 2 h + rr
                    const uint32_t* p = (const uint32_t*)(halo + pb + (kb * IH + 2 * h + rr) * IWP);
                    const uint32_t d0 = p[0], d1 = p[1], d2 = p[2];
                    a.u[2 * rr] = __builtin_amdgcn_alignbit(d1, d0, 16);
                    a.u[2 * rr + 1] = __builtin_amdgcn_alignbit(d2, d1, 16);
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[kb].v, a.v, acc, 0, 0, 0);       // D = W x im2col^T: rows = channels, columns = positions
            }
            const int ow = o0w + w0, oh = o0h + hh, od = o0d + d;
            const bool ok = od < sd && oh < sh && ow < sw;
            const int loff = ((d * sh + hh) * sw + w0) * CS;      // 32-bit offset inside the tile; the tile's origin is uniform (scalar registers)
            // the activation on the MFMA's own registers, then two v_permlane32_swap per register pair regroup D (rows = channels) into channels
            // 8h..8h+7 and 16+8h..23+8h of this lane's position
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = apply_act_t<EPI>(acc[e], act);
            float v[2][8];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const auto lo = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[i]), __float_as_uint(acc[4 + i]), false, false);
                const auto hi = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[8 + i]), __float_as_uint(acc[12 + i]), false, false);
                v[0][i] = __uint_as_float(lo[0]); v[0][4 + i] = __uint_as_float(lo[1]);
                v[1][i] = __uint_as_float(hi[0]); v[1][4 + i] = __uint_as_float(hi[1]);
            }
            if (ok) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int c = 16 * j + 8 * h;
                    if constexpr (MASKED) {
                        if (f8.mask_bits) {                  // the position's 32-channel dword (F8Side, common.h) instead of 16 bytes of the saved activation
                            const unsigned mb = f8.mask_bits[(tile_org + loff) >> 5] >> c;
#pragma unroll
                            for (int q = 0; q < 8; ++q)
                                if (!((mb >> q) & 1u)) v[j][q] = 0.f;
                        } else {
                            __attribute__((aligned(16))) bf16 mv[8];
                            *(uint4*)mv = *(const uint4*)(mask + tile_org + loff + c);
#pragma unroll
                            for (int q = 0; q < 8; ++q)
                                if (!(to_f32(mv[q]) > 0.f)) v[j][q] = 0.f;
                        }
                    }
                    *(uint4*)(S + tile_org + loff + c) = make_uint4(pack2_bf16(v[j][0], v[j][1]), pack2_bf16(v[j][2], v[j][3]), pack2_bf16(v[j][4], v[j][5]), pack2_bf16(v[j][6], v[j][7]));
                }
            }
            if constexpr (SIDE8) {
                if (f8.bits_out) {                            // every lane takes part in the lane swap; lanes h = 0 store the position's dword
                    const unsigned dw = mask_bytes_to_dword(mask_byte_of(v[0]), mask_byte_of(v[1]));
                    if (ok && h == 0) f8.bits_out[(tile_org + loff) >> 5] = dw;
                }
                if (ok && f8.amax) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) amx = fmaxf(amx, fmaxf(fabsf(v[0][q]), fabsf(v[1][q])));
                }
                if (f8.out8) {                                // every lane takes part in the lane swap; `ok` only guards the store
                    const uint4 q16 = fp8_pair_to_16(pack8_fp8(v[0], o8s), pack8_fp8(v[1], o8s));
                    if (ok) *(uint4*)(f8.out8 + tile_org + loff + 16 * h) = q16;
                }
            }
        }
        if (!has_next) break;
        __syncthreads();                                     // every wave is done with this tile's LDS image
        store_lds();
        __syncthreads();
        tile = next;
    }
    if constexpr (SIDE8) {
        if (f8.amax) {
            __syncthreads();                                 // the halo image is spent: its first floats carry the per-wave maxima
            amax_publish_wg(f8.amax, amx, blockIdx.x, (float*)halo);
        }
    }
}

// -------------------------------------------------------------------------------------- up, Cl == 1
// One thread per (source voxel q, half of the CS channels): the 2 x 2 x 2 (3D) / 2 x 2 (2D) outputs around q read the 3^nd
// neighbourhood of q, so each neighbour's 16 channels are loaded ONCE (two 16-byte loads) and feed every output parity that
// uses it — 2.4x fewer loads than one thread per output voxel.  Weights sit in LDS as fp32 [tap][cs] and are read as
// wave-uniform broadcasts; the two channel halves of a voxel meet in a shuffle.
template <typename T, int ND, int CS>
__global__ __launch_bounds__(256) void up_c1_kernel(const T* __restrict__ S, const float* __restrict__ w, const float* __restrict__ bias,
                                                    const T* __restrict__ mask, T* __restrict__ L, int B, int sd, int sh, int sw, int ld, int lh, int lw, int act) {
    static_assert(CS == 32, "two 16-channel halves");
    constexpr int TAPS = (ND == 3) ? 64 : 16, NP = (ND == 3) ? 8 : 4;
    __shared__ __attribute__((aligned(16))) float wl[TAPS * CS];      // [tap][cs]
    for (int i = threadIdx.x; i < TAPS * CS; i += 256) { const int tap = i / CS, cs = i % CS; wl[i] = w[cs * TAPS + tap]; }
    __syncthreads();
    const int n = B * sd * sh * sw;                          // host bounds it below 2^30
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const int q = gid >> 1, hc = gid & 1;                    // lanes 2k, 2k+1: the two channel halves of voxel q
    const bool live = q < n;
    int rr = live ? q : 0;
    const int qx = rr % sw; rr /= sw;
    const int qy = rr % sh; rr /= sh;
    const int qz = rr % sd;
    const int b = rr / sd;
    float acc[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) acc[p] = 0.f;
    // output parity (pz, py, px) of voxel q: l = 2 q + p reads neighbour q + (p - 1 + a), a in {0, 1}, through tap k = 3 - p - 2 a
    // (per dimension).  Neighbour offset o = p - 1 + a in {-1, 0, 1}: o = -1 <- (p 0, a 0, k 3); o = 0 <- (p 0, a 1, k 1), (p 1, a 0, k 2);
    // o = +1 <- (p 1, a 1, k 0).
#pragma unroll
    for (int oz = (ND == 3 ? -1 : 0); oz <= (ND == 3 ? 1 : 0); ++oz) {
        const int z = qz + oz;
        if (ND == 3 && (z < 0 || z >= sd)) continue;
#pragma unroll
        for (int oy = -1; oy <= 1; ++oy) {
            const int y = qy + oy;
            if (y < 0 || y >= sh) continue;
#pragma unroll
            for (int ox = -1; ox <= 1; ++ox) {
                const int x = qx + ox;
                if (x < 0 || x >= sw) continue;
                const T* sp = S + ((((size_t)b * sd + z) * sh + y) * sw + x) * CS + 16 * hc;
                float sv[16];
                {
                    constexpr int NU = (16 * sizeof(T)) / 16;
                    __attribute__((aligned(16))) T raw[16];
#pragma unroll
                    for (int u = 0; u < NU; ++u) ((uint4*)raw)[u] = ((const uint4*)sp)[u];
#pragma unroll
                    for (int e = 0; e < 16; ++e) sv[e] = to_f32(raw[e]);
                }
                // every (parity, tap) pair per dimension that maps onto this offset
#pragma unroll
                for (int cz = 0; cz < (ND == 3 ? 2 : 1); ++cz) {
                    const int pz = (ND == 3) ? (oz == -1 ? 0 : (oz == 1 ? 1 : cz)) : 0;
                    const int kd = (ND == 3) ? (oz == -1 ? 3 : (oz == 1 ? 0 : (cz == 0 ? 1 : 2))) : 0;
                    if (ND == 3 && oz != 0 && cz == 1) continue;
#pragma unroll
                    for (int cy = 0; cy < 2; ++cy) {
                        const int py = oy == -1 ? 0 : (oy == 1 ? 1 : cy), kh = oy == -1 ? 3 : (oy == 1 ? 0 : (cy == 0 ? 1 : 2));
                        if (oy != 0 && cy == 1) continue;
#pragma unroll
                        for (int cx = 0; cx < 2; ++cx) {
                            const int px = ox == -1 ? 0 : (ox == 1 ? 1 : cx), kw = ox == -1 ? 3 : (ox == 1 ? 0 : (cx == 0 ? 1 : 2));
                            if (ox != 0 && cx == 1) continue;
                            const float* wr = &wl[((kd * 4 + kh) * 4 + kw) * CS + 16 * hc];
                            float a = 0.f;
#pragma unroll
                            for (int e = 0; e < 16; ++e) a += sv[e] * wr[e];
                            acc[(pz * 2 + py) * 2 + px] += a;
                        }
                    }
                }
            }
        }
    }
    const float b0 = bias ? bias[0] : 0.f;
#pragma unroll
    for (int p = 0; p < NP; ++p) acc[p] += __shfl_xor(acc[p], 1, 64);      // the other channel half
    if (!live) return;
    // lane hc writes the outputs with px == hc ... each lane stores half of the 2^nd outputs (x-adjacent pairs stay contiguous per py, pz)
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        if ((p & 1) != hc) continue;
        const int pz = (ND == 3) ? (p >> 2) : 0, py = (p >> 1) & 1, px = p & 1;
        const int lz = (ND == 3) ? 2 * qz + pz : 0, ly = 2 * qy + py, lx = 2 * qx + px;
        const size_t idx = (((size_t)b * ld + lz) * lh + ly) * lw + lx;
        float v = apply_act(acc[p] + b0, act);
        if (mask && !(to_f32(mask[idx]) > 0.f)) v = 0.f;
        L[idx] = from_f32<T>(v);
    }
}

// -------------------------------------------------------------------------------------- up, Cl == 1, bf16 on the MFMA
// The same product as a GEMM per source voxel q: out[2 q + p] = sum_{o in {-1,0,1}^nd} sum_c S[q + o][c] Wm[(o, c)][p], with
// Wm[(o, c)][p] = W[c][tap(p, o)] where parity p reaches neighbour o (per dimension o = p - 1 + a, a in {0, 1}, tap k = 3 - p - 2 a)
// and 0 elsewhere: M = voxels, K = 3^nd * 32, N = 2^nd parities.  8/27 of Wm is non-zero, which is cheaper than the 1024 LDS weight reads per
// thread of the scalar form above (30 -> 14 us at B = 4).  Round 2: v_mfma_f32_16x16x32_bf16 instead of 32x32x16 — the 8 parities fill half of
// its 16 rows instead of a quarter of 32, and one instruction takes all 32 channels of a neighbour: 27 MFMAs of 16 cycles per 16 voxels where
// there were 54 of 32 cycles per 32 (the layer is MFMA-bound at the 240-row decode sweep: 0.45 of its 1.2 ms) — and a workgroup builds its
// weight fragments ONCE and walks several tiles with the next halo in flight.
//   A (weights): lane l -> row l % 16 (parity; rows 8..15 re-read rows 0..7 and are ignored), k = 8 (l / 16) + j
//   B (voxels) : lane l -> column l % 16 (voxel), k = 8 (l / 16) + j            D: lane l -> column l % 16, rows 4 (l / 16) + r
// so lanes 0..15 end with parities 0..3 = the (py, px) block of plane pz = 0 of their voxel, lanes 16..31 with pz = 1 (3D); two 4-byte stores.
// LDS: the 32-channel halo as 4 planes of 16-byte pieces, plane pitch a multiple of 16 slots and 16 consecutive-x voxels per MFMA column set,
// so every ds_read_b128 lane group ({0-3, 12-15, 20-27}, ...) falls on 16 different bank slots; Wm^T as [neighbour][k group][parity] rows.
template <int ND> struct TileC1U;
template <> struct TileC1U<3> { static constexpr int TD = 2, TH = 8, TW = 16; };
template <> struct TileC1U<2> { static constexpr int TD = 1, TH = 16, TW = 16; };

// Wm^T fragments of the Cl == 1 up convolution, once per workgroup: one 16-byte row = 8 channels of (neighbour, k group, parity)
template <int ND>
__device__ inline void up_c1_build_wfr(bf16* wfr, const float* __restrict__ w, int t) {
    constexpr int NNB = (ND == 3) ? 27 : 9, TAPS = (ND == 3) ? 64 : 16, NPAR = (ND == 3) ? 8 : 4;
    for (int row = t; row < NNB * 4 * 8; row += 256) {
        const int p = row & 7, q = (row >> 3) & 3, nb = row >> 5;
        const int c0 = 8 * q;
        const int ox = nb % 3 - 1, oy = nb / 3 % 3 - 1, oz = (ND == 3) ? nb / 9 - 1 : 0;
        const int px = p & 1, py = (p >> 1) & 1, pz = (ND == 3) ? (p >> 2) : 0;
        const int ax = ox - px + 1, ay = oy - py + 1, az = oz - pz + 1;
        const bool valid = (p < NPAR) & (ax >= 0) & (ax <= 1) & (ay >= 0) & (ay <= 1) & ((ND == 2) | ((az >= 0) & (az <= 1)));
        const int kw = 3 - px - 2 * ax, kh = 3 - py - 2 * ay, kd = (ND == 3) ? 3 - pz - 2 * az : 0;
        uint4 o = make_uint4(0u, 0u, 0u, 0u);
        if (valid) {
            const float* wp = w + c0 * TAPS + (kd * 4 + kh) * 4 + kw;
            o = make_uint4(pack2_bf16(wp[0], wp[TAPS]), pack2_bf16(wp[2 * TAPS], wp[3 * TAPS]), pack2_bf16(wp[4 * TAPS], wp[5 * TAPS]), pack2_bf16(wp[6 * TAPS], wp[7 * TAPS]));
        }
        *(uint4*)(wfr + row * 8) = o;
    }
}

template <int ND, int EPI, bool MASKED>
__global__ __launch_bounds__(256) void up_c1_mfma_kernel(const bf16* __restrict__ S, const float* __restrict__ w, const float* __restrict__ bias,
                                                         const bf16* __restrict__ mask, bf16* __restrict__ L, int sd, int sh, int sw, int tiles_d, int tiles_h,
                                                         int tiles_w, int ntiles, int act) {
    using TLU = TileC1U<ND>;
    constexpr int TD = TLU::TD, TH = TLU::TH, TW = TLU::TW;     // 256 voxels
    static_assert(TW == 16 && TD * TH * TW == 256, "a column set is 16 consecutive-x voxels");
    constexpr int ID = (ND == 3) ? TD + 2 : 1, IH = TH + 2, IW = TW + 2, NPOS = ID * IH * IW, PPITCH = (NPOS + 15) / 16 * 16;
    constexpr int NNB = (ND == 3) ? 27 : 9, TAPS = (ND == 3) ? 64 : 16, NPAR = (ND == 3) ? 8 : 4;
    constexpr int HN = (NPOS * 4 + 255) / 256;
    __shared__ uint4 halo[4 * PPITCH];
    __shared__ __attribute__((aligned(16))) bf16 wfr[NNB * 4 * 8 * 8];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, col = lane & 15, kq = lane >> 4;
    uint4 hv[HN];
    auto issue_loads = [&](int tile) {                         // halo of `tile`: piece t % 4 of positions t / 4 + 64 i (stepped, not divided)
        const int tw_i = tile % tiles_w; tile /= tiles_w;
        const int th_i = tile % tiles_h; tile /= tiles_h;
        const int td_i = tile % tiles_d, b = tile / tiles_d;
        const int o0d = td_i * TD, o0h = th_i * TH, o0w = tw_i * TW;
        constexpr int DX = 64 % IW, DY = (64 / IW) % IH, DZ = 64 / (IW * IH);
        const int piece = t & 3, pos0 = t >> 2;
        int x = pos0 % IW, y = (pos0 / IW) % IH, z = pos0 / (IW * IH);
        const bf16* Sb = S + (size_t)b * sd * sh * sw * 32 + piece * 8;
#pragma unroll
        for (int i = 0; i < HN; ++i) {
            const int gz = (ND == 3) ? o0d - 1 + z : 0, gy = o0h - 1 + y, gx = o0w - 1 + x;
            const bool ok = (z < ID) & (gz >= 0) & (gz < sd) & (gy >= 0) & (gy < sh) & (gx >= 0) & (gx < sw);
            const uint4 v = *(const uint4*)(Sb + (((size_t)min(max(gz, 0), sd - 1) * sh + min(max(gy, 0), sh - 1)) * sw + min(max(gx, 0), sw - 1)) * 32);
            hv[i] = ok ? v : make_uint4(0u, 0u, 0u, 0u);
            x += DX; if (x >= IW) { x -= IW; y += 1; }
            y += DY; if (y >= IH) { y -= IH; z += 1; }
            if (y >= IH) { y -= IH; z += 1; }
            z += DZ;
        }
    };
    auto store_lds = [&]() {
#pragma unroll
        for (int i = 0; i < HN; ++i) {
            const int it = t + i * 256;
            if (it < NPOS * 4) halo[(it & 3) * PPITCH + (it >> 2)] = hv[i];
        }
    };
    int tile = blockIdx.x;
    issue_loads(tile);
    up_c1_build_wfr<ND>(wfr, w, t);
    const float bz = bias ? bias[0] : 0.f;
    const int ld = (ND == 3) ? 2 * sd : 1, lh = 2 * sh, lw = 2 * sw;
    store_lds();
    __syncthreads();
    // each wave: 4 column sets of 16 voxels (one x-row of the tile each): cs -> tile row (wave * 4 + cs) = (d, hh)
    int pbv[4];
#pragma unroll
    for (int cs = 0; cs < 4; ++cs) {
        const int rowi = wave * 4 + cs;
        pbv[cs] = ((rowi / TH) * IH + rowi % TH) * IW + col;
    }
    while (true) {
        const int next = tile + (int)gridDim.x;
        const bool has_next = next < ntiles;
        if (has_next) issue_loads(next);
        int tt = tile;
        const int tw_i = tt % tiles_w; tt /= tiles_w;
        const int th_i = tt % tiles_h; tt /= tiles_h;
        const int td_i = tt % tiles_d, b = tt / tiles_d;
        const int o0d = td_i * TD, o0h = th_i * TH, o0w = tw_i * TW;
        f32x4 acc[4];
#pragma unroll
        for (int cs = 0; cs < 4; ++cs) acc[cs] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int nb = 0; nb < NNB; ++nb) {
            const int off = (((ND == 3) ? nb / 9 : 0) * IH + nb / 3 % 3) * IW + nb % 3;
            union { uint4 u; bf16x8 v; } a, bv[4];
            a.u = *(const uint4*)(wfr + ((nb * 4 + kq) * 8 + (col & 7)) * 8);
#pragma unroll
            for (int cs = 0; cs < 4; ++cs) bv[cs].u = halo[kq * PPITCH + pbv[cs] + off];
#pragma unroll
            for (int cs = 0; cs < 4; ++cs) acc[cs] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, bv[cs].v, acc[cs], 0, 0, 0);
        }
        // ---- epilogue: lanes 0..15 hold (py, px) = (r >> 1, r & 1) of plane pz = 0, lanes 16..31 of pz = 1 (2D: lanes 0..15 only) ----
        if (kq < ((ND == 3) ? 2 : 1)) {
#pragma unroll
            for (int cs = 0; cs < 4; ++cs) {
                const int rowi = wave * 4 + cs;
                const int qz = o0d + rowi / TH, qy = o0h + rowi % TH, qx = o0w + col;
                if (qz >= sd || qy >= sh || qx >= sw) continue;
                const int lz = (ND == 3) ? 2 * qz + kq : 0;
#pragma unroll
                for (int py = 0; py < 2; ++py) {
                    const size_t idx = (((size_t)b * ld + lz) * lh + 2 * qy + py) * lw + 2 * qx;
                    float v0 = apply_act_t<EPI>(acc[cs][2 * py] + bz, act), v1 = apply_act_t<EPI>(acc[cs][2 * py + 1] + bz, act);
                    if constexpr (MASKED) {
                        union { uint32_t u; bf16 e[2]; } mk;
                        mk.u = *(const uint32_t*)(mask + idx);
                        if (!(to_f32(mk.e[0]) > 0.f)) v0 = 0.f;
                        if (!(to_f32(mk.e[1]) > 0.f)) v1 = 0.f;
                    }
                    *(uint32_t*)(L + idx) = pack2_bf16(v0, v1);
                }
            }
        }
        if (!has_next) break;
        __syncthreads();                                     // every wave is done with this tile's LDS image
        store_lds();
        __syncthreads();
        tile = next;
    }
}

// The 3D form for long z columns: a workgroup walks `walk` consecutive tiles along z and keeps the two halo planes a tile shares with the next one
// in LDS, so a step stages 2 planes instead of 4 (the halo is the kernel's L2 traffic: 2.8 source reads per voxel for a lone 2 x 8 x 16 tile, 1.4
// when walking).  LDS holds two half-buffers of 2 planes; tile j of a walk reads logical planes {0,1} from buffer j % 2 and {2,3} from the other.
// The host picks `walk` so that the launch still has >= ~1024 units (walk == 1 for the training batch: same traffic as up_c1_mfma_kernel).
// TIN = fp8 (the decode sweep's hand-off from an fp8 layer, cvae_conv_up_c1_fp8in): S holds e4m3 codes of activation / in_scale; a position's 32 channels are
// 32 bytes, fetched as four 8-byte pieces in the same thread -> (position, piece) map, widened to bf16 (exactly: 3 mantissa bits) on the way into LDS, and the
// accumulator is multiplied by in_scale before the bias (the product is linear in S).  Half the bytes of the layer's input, the same tap loop.
template <int EPI, bool MASKED, typename TIN = bf16>
__global__ __launch_bounds__(256) void up_c1_mfma_walk_kernel(const TIN* __restrict__ S, const float* __restrict__ w, const float* __restrict__ bias,
                                                              const bf16* __restrict__ mask, bf16* __restrict__ L, int sd, int sh, int sw, int tiles_d,
                                                              int tiles_h, int tiles_w, int walk, int segs, int nunits, int act, float in_scale) {
    constexpr bool IN8 = sizeof(TIN) == 1;
    using HV = typename std::conditional<IN8, uint2, uint4>::type;
    using TLU = TileC1U<3>;
    constexpr int TD = TLU::TD, TH = TLU::TH, TW = TLU::TW;
    static_assert(TD == 2 && TW == 16 && TD * TH * TW == 256, "two z planes per tile, 16 consecutive-x voxels per column set");
    constexpr int IH = TH + 2, IW = TW + 2, PLANE = IH * IW, HPOS = 2 * PLANE, PPITCH = (4 * PLANE + 15) / 16 * 16;
    constexpr int NNB = 27, HN = (HPOS * 4 + 255) / 256;
    __shared__ uint4 halo[4 * PPITCH];
    __shared__ __attribute__((aligned(16))) bf16 wfr[NNB * 4 * 8 * 8];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, col = lane & 15, kq = lane >> 4;
    HV hva[HN], hvb[HN];
    // two planes gz0, gz0 + 1 of the halo of tile column (b, o0h, o0w): piece t % 4 of positions t / 4 + 64 i
    auto issue_half = [&](HV (&hv)[HN], int b, int o0h, int o0w, int gz0) {
        constexpr int DX = 64 % IW, DY = (64 / IW) % IH, DZ = 64 / PLANE;
        const int piece = t & 3, pos0 = t >> 2;
        int x = pos0 % IW, y = (pos0 / IW) % IH, z = pos0 / PLANE;
        const TIN* Sb = S + (size_t)b * sd * sh * sw * 32 + piece * 8;
#pragma unroll
        for (int i = 0; i < HN; ++i) {
            const int gz = gz0 + z, gy = o0h - 1 + y, gx = o0w - 1 + x;
            const bool ok = (z < 2) & (gz >= 0) & (gz < sd) & (gy >= 0) & (gy < sh) & (gx >= 0) & (gx < sw);
            const HV v = *(const HV*)(Sb + (((size_t)min(max(gz, 0), sd - 1) * sh + min(max(gy, 0), sh - 1)) * sw + min(max(gx, 0), sw - 1)) * 32);
            hv[i] = ok ? v : HV{};                           // the fp8 code 0 is +0.0
            x += DX; if (x >= IW) { x -= IW; y += 1; }
            y += DY; if (y >= IH) { y -= IH; z += 1; }
            if (y >= IH) { y -= IH; z += 1; }
            z += DZ;
        }
    };
    auto store_half = [&](const HV (&hv)[HN], int buf) {
#pragma unroll
        for (int i = 0; i < HN; ++i) {
            const int it = t + i * 256;
            if constexpr (IN8) {
                if (it < HPOS * 4) halo[(it & 3) * PPITCH + buf * HPOS + (it >> 2)] = fp8x8_to_bf16x8(hv[i]);
            } else {
                if (it < HPOS * 4) halo[(it & 3) * PPITCH + buf * HPOS + (it >> 2)] = hv[i];
            }
        }
    };
    auto decode = [&](int unit, int& b, int& td0, int& td1, int& o0h, int& o0w) {
        const int tw_i = unit % tiles_w; unit /= tiles_w;
        const int th_i = unit % tiles_h; unit /= tiles_h;
        const int seg = unit % segs; b = unit / segs;
        td0 = seg * walk; td1 = min(tiles_d, td0 + walk);
        o0h = th_i * TH; o0w = tw_i * TW;
    };
    int unit = blockIdx.x, b, td, td_end, o0h, o0w;
    decode(unit, b, td, td_end, o0h, o0w);
    issue_half(hva, b, o0h, o0w, td * TD - 1);
    issue_half(hvb, b, o0h, o0w, td * TD + 1);
    up_c1_build_wfr<3>(wfr, w, t);
    const float bz = bias ? bias[0] : 0.f;
    const int ld = 2 * sd, lh = 2 * sh, lw = 2 * sw;
    store_half(hva, 0);
    store_half(hvb, 1);
    __syncthreads();
    // each wave: 4 column sets of 16 voxels (one x-row of the tile each): cs -> tile row (wave * 4 + cs) = (d, hh); d is the same for the wave
    const int dz_own = (wave * 4) / TH;
    int pb2[4];
#pragma unroll
    for (int cs = 0; cs < 4; ++cs) pb2[cs] = kq * PPITCH + ((wave * 4 + cs) % TH) * IW + col;
    int par = 0;                                             // buffer that holds logical planes 0, 1
    while (true) {
        const bool same = td + 1 < td_end;
        const int next_unit = unit + (int)gridDim.x;
        const bool has_next = next_unit < nunits;
        int nb_ = 0, ntd = 0, ntd_end = 0, nh = 0, nw = 0;
        if (same) issue_half(hva, b, o0h, o0w, (td + 1) * TD + 1);
        else if (has_next) {
            decode(next_unit, nb_, ntd, ntd_end, nh, nw);
            issue_half(hva, nb_, nh, nw, ntd * TD - 1);
            issue_half(hvb, nb_, nh, nw, ntd * TD + 1);
        }
        int zoff[3];
#pragma unroll
        for (int dz = 0; dz < 3; ++dz) zoff[dz] = ((dz_own + dz + 2 * par) & 3) * PLANE;
        f32x4 acc[4];
#pragma unroll
        for (int cs = 0; cs < 4; ++cs) acc[cs] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int nb = 0; nb < NNB; ++nb) {
            const int off = (nb / 3 % 3) * IW + nb % 3;
            union { uint4 u; bf16x8 v; } a, bv[4];
            a.u = *(const uint4*)(wfr + ((nb * 4 + kq) * 8 + (col & 7)) * 8);
#pragma unroll
            for (int cs = 0; cs < 4; ++cs) bv[cs].u = halo[zoff[nb / 9] + pb2[cs] + off];
#pragma unroll
            for (int cs = 0; cs < 4; ++cs) acc[cs] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, bv[cs].v, acc[cs], 0, 0, 0);
        }
        // ---- epilogue: lanes 0..15 hold (py, px) = (r >> 1, r & 1) of plane pz = 0, lanes 16..31 of pz = 1 ----
        if (kq < 2) {
            const int o0d = td * TD;
#pragma unroll
            for (int cs = 0; cs < 4; ++cs) {
                const int rowi = wave * 4 + cs;
                const int qz = o0d + rowi / TH, qy = o0h + rowi % TH, qx = o0w + col;
                if (qz >= sd || qy >= sh || qx >= sw) continue;
                const int lz = 2 * qz + kq;
#pragma unroll
                for (int py = 0; py < 2; ++py) {
                    const size_t idx = (((size_t)b * ld + lz) * lh + 2 * qy + py) * lw + 2 * qx;
                    float a0 = acc[cs][2 * py], a1 = acc[cs][2 * py + 1];
                    if constexpr (IN8) { a0 *= in_scale; a1 *= in_scale; }
                    float v0 = apply_act_t<EPI>(a0 + bz, act), v1 = apply_act_t<EPI>(a1 + bz, act);
                    if constexpr (MASKED) {
                        union { uint32_t u; bf16 e[2]; } mk;
                        mk.u = *(const uint32_t*)(mask + idx);
                        if (!(to_f32(mk.e[0]) > 0.f)) v0 = 0.f;
                        if (!(to_f32(mk.e[1]) > 0.f)) v1 = 0.f;
                    }
                    *(uint32_t*)(L + idx) = pack2_bf16(v0, v1);
                }
            }
        }
        if (!same && !has_next) break;
        __syncthreads();                                     // every wave is done with this tile's LDS image
        if (same) { store_half(hva, par); par ^= 1; td += 1; }   // the new planes replace logical {0,1}; old {2,3} become the next tile's {0,1}
        else {
            store_half(hva, 0); store_half(hvb, 1); par = 0;
            unit = next_unit; b = nb_; td = ntd; td_end = ntd_end; o0h = nh; o0w = nw;
        }
        __syncthreads();
    }
}

// -------------------------------------------------------------------------------------- wgrad, Cl == 1
// dW[cs][tap] = sum_pos S[pos][cs] * L[2 pos - 1 + k(tap)]  ==  S^T [32 x K] . im2col(L) [K x taps],  K = positions.
//   bf16: A fragments (S^T) by the transposing LDS read; B fragments gathered from the 1-channel halo (8 stride-2
//         elements per lane); v_mfma_f32_32x32x16_bf16.  A third accumulator S^T . ones yields the bias gradient.
//   fp32: v_mfma_f32_32x32x2_f32, one element per lane per operand.
// Each workgroup walks `total / n_split` tiles of 128 positions and leaves ONE slab of partial sums with plain stores; wgrad_c1_finish_kernel adds
// the slabs in index order into [Cs][1][taps] (no atomics).
// TL: the dtype the 1-channel image L is STORED in (fp32 for the network input: read directly, rounded to T on the way into LDS).
// VEC (bf16 compute only): the halo rows of L are fetched as aligned 16-byte vectors ([2 o0w - EPV, 2 o0w + 2 TW + EPV), EPV = 4 fp32 / 8 bf16
// elements: 3 loads per thread and tile instead of 8 element loads) and scattered into the four LDS planes element by element; needs
// lw % EPV == 0 and a 16-byte aligned L (image_vec_ok).
template <typename T, typename TL, int ND, bool LSUM, bool VEC>
__global__ __launch_bounds__(256) void wgrad_c1_kernel(const T* __restrict__ S, const TL* __restrict__ L, float* __restrict__ ws, bool want_bias, int B, int sd, int sh, int sw,
                                                       int Cs, int ld, int lh, int lw, int tiles_d, int tiles_h, int tiles_w, int n_split, float* __restrict__ lsum_ws) {
    constexpr int TD = (ND == 3) ? C1W_TD3 : 1, TH = (ND == 3) ? C1W_TH3 : 8, TW = (ND == 3) ? C1W_TW3 : 16;     // 128 positions; 3D: wide in x (see TileC1V)
    constexpr int ID = (ND == 3) ? 2 * TD + 2 : 1, IH = 2 * TH + 2, IW = 2 * TW + 2, NPOS = ID * IH * IW;
    constexpr int TAPS = (ND == 3) ? 64 : 16, NTS = (TAPS + 31) / 32;
    constexpr int NU = (8 * sizeof(T)) / 16, HN = (NPOS + 255) / 256;
    // bf16: the 1-channel halo is kept as four planes — x parity (the taps step by 2 in x) times a copy shifted by one slot — so that
    // the 8 stride-2 elements a lane gathers for one B fragment (tap kw, positions w0 .. w0 + 7) are ONE aligned 16-byte read from plane
    // (kw & 1, kw >> 1) instead of eight 2-byte reads.  fp32 keeps the linear image (one element per lane per MFMA there).
    constexpr int LROWS = ID * IH, LPITCH = ((IW / 2 + 1 + 7) / 8) * 8, LPLANE = LROWS * LPITCH;
    constexpr int L_ELEMS = sizeof(T) == 2 ? 4 * LPLANE : NPOS + 8;
    __shared__ __attribute__((aligned(16))) T s_lds[128 * 32];
    __shared__ __attribute__((aligned(16))) T l_lds[L_ELEMS];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r = lane & 31, hk = lane >> 5;
    const int cs0 = blockIdx.y * 32;
    const int total = B * tiles_d * tiles_h * tiles_w;
    f32x16 acc[NTS], accb;
#pragma unroll
    for (int s = 0; s < NTS; ++s)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[s][e] = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) accb[e] = 0.f;
    int toff[NTS]; bool tvalid[NTS];                        // per-lane tap offsets inside the halo tile (tap = ts*32 + r)
#pragma unroll
    for (int s = 0; s < NTS; ++s) {
        const int tap = s * 32 + r;
        tvalid[s] = tap < TAPS;
        const int kd = (ND == 3) ? (tap >> 4) & 3 : 0, kh = (tap >> 2) & 3, kw = tap & 3;
        toff[s] = tvalid[s] ? (sizeof(T) == 2 ? ((kw & 1) * 2 + (kw >> 1)) * LPLANE + (kd * IH + kh) * LPITCH : (kd * IH + kh) * IW + kw) : 0;
    }
    // Software pipeline over the workgroup's tiles: the global loads of tile i+1 are in flight (in registers) while tile i
    // is consumed from LDS, so the HBM latency hides under the barriers, LDS traffic and MFMAs of the previous tile.
    constexpr int EPV = 16 / sizeof(TL), NV = (2 * TW + 2 * EPV) / EPV, NVEC = LROWS * NV, HNV = VEC ? (NVEC + 255) / 256 : 1;
    uint4 sv[2][NU];
    T hv[VEC ? 1 : HN];
    uint4 hvv[HNV];
    // ConvTranspose bias gradient (the plain sum of L): every L element lies in the non-halo part of exactly one tile (host checks
    // L == 2 S), so the channel block 0 workgroups add up what they load anyway; wgrad_c1_finish sums the per-workgroup partials.
    const bool want_lsum = LSUM && blockIdx.y == 0;
    float lacc = 0.f;
    auto issue = [&](int tile) {
        int tt = tile;
        const int tw_i = tt % tiles_w; tt /= tiles_w;
        const int th_i = tt % tiles_h; tt /= tiles_h;
        const int td_i = tt % tiles_d;
        const int b = tt / tiles_d;
        const int o0d = td_i * TD, o0h = th_i * TH, o0w = tw_i * TW;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int it = t + i * 256, piece = it & 3, m = it >> 2;
            const int w = m % TW, hh = m / TW % TH, d = m / (TW * TH);
            const int od = o0d + d, oh = o0h + hh, ow = o0w + w;
            const bool ok = od < sd && oh < sh && ow < sw;
            const T* src = S + (ok ? ((((size_t)b * sd + od) * sh + oh) * sw + ow) * Cs + cs0 + piece * 8 : 0);
#pragma unroll
            for (int u = 0; u < NU; ++u) sv[i][u] = ok ? ((const uint4*)src)[u] : make_uint4(0, 0, 0, 0);
        }
        if constexpr (VEC) {
#pragma unroll
            for (int i = 0; i < HNV; ++i) {
                const int it = min(t + i * 256, NVEC - 1), row = it / NV, j = it % NV;
                const int y = row % IH, z = row / IH;
                const int gz = (ND == 3) ? 2 * o0d - 1 + z : 0, gy = 2 * o0h - 1 + y, gx = 2 * o0w - EPV + j * EPV;
                const bool ok = (gz >= 0) & (gz < ld) & (gy >= 0) & (gy < lh) & (gx >= 0) & (gx < lw);
                const uint4 v = *(const uint4*)(L + (((size_t)b * ld + min(max(gz, 0), ld - 1)) * lh + min(max(gy, 0), lh - 1)) * lw + min(max(gx, 0), lw - EPV));
                hvv[i] = ok ? v : make_uint4(0u, 0u, 0u, 0u);
                if (LSUM) {
                    if (want_lsum && ok && (ND == 2 || (z >= 1 && z <= 2 * TD)) && y >= 1 && y <= 2 * TH && t + i * 256 < NVEC) {
                        const TL* ev = (const TL*)&hvv[i];
#pragma unroll
                        for (int e = 0; e < EPV; ++e) {
                            const int x = j * EPV + e - (EPV - 1);                      // column inside the tile's halo row
                            if (x >= 1 && x <= 2 * TW) lacc += to_f32(from_f32<T>(to_f32(ev[e])));
                        }
                    }
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < HN; ++i) {
                const int pos = t + i * 256;
                const int x = pos % IW, y = pos / IW % IH, z = pos / (IW * IH);
                const int gz = (ND == 3) ? 2 * o0d - 1 + z : 0, gy = 2 * o0h - 1 + y, gx = 2 * o0w - 1 + x;
                const bool ok = pos < NPOS && gz >= 0 && gz < ld && gy >= 0 && gy < lh && gx >= 0 && gx < lw;
                hv[i] = ok ? from_f32<T>(to_f32(L[(((size_t)b * ld + gz) * lh + gy) * lw + gx])) : from_f32<T>(0.f);
                if (LSUM) {
                    const bool inner = (ND == 2 || (z >= 1 && z <= 2 * TD)) && y >= 1 && y <= 2 * TH && x >= 1 && x <= 2 * TW;
                    if (want_lsum && ok && inner) lacc += to_f32(hv[i]);
                }
            }
        }
    };
    if ((int)blockIdx.x < total) issue(blockIdx.x);
    for (int tile = blockIdx.x; tile < total; tile += n_split) {
        __syncthreads();                                    // the previous tile's readers are done with LDS
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int it = t + i * 256;
#pragma unroll
            for (int u = 0; u < NU; ++u) ((uint4*)(s_lds + it * 8))[u] = sv[i][u];       // row-major [pos][32 cs], 16-byte stores
        }
        if constexpr (VEC) {
#pragma unroll
            for (int i = 0; i < HNV; ++i) {
                const int it = t + i * 256, row = it / NV, j = it % NV;
                if (it < NVEC) {
                    const TL* ev = (const TL*)&hvv[i];
#pragma unroll
                    for (int e = 0; e < EPV; ++e) {
                        const int x = j * EPV + e - (EPV - 1), xi = x >> 1;
                        if (x >= 0 && x < IW) {
                            const T v = from_f32<T>(to_f32(ev[e]));
                            T* pl = l_lds + (x & 1) * 2 * LPLANE + row * LPITCH;
                            pl[xi] = v;                                      // copy 0: slot xi
                            if (xi > 0) pl[LPLANE + xi - 1] = v;             // copy 1: shifted left by one slot
                        }
                    }
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < HN; ++i) {
                const int pos = t + i * 256;
                if (pos < NPOS) {
                    if constexpr (sizeof(T) == 2) {
                        const int x = pos % IW, row = pos / IW, xi = x >> 1;
                        T* pl = l_lds + (x & 1) * 2 * LPLANE + row * LPITCH;
                        pl[xi] = hv[i];                                          // copy 0: slot xi
                        if (xi > 0) pl[LPLANE + xi - 1] = hv[i];                 // copy 1: shifted left by one slot
                    } else {
                        l_lds[pos] = hv[i];
                    }
                }
            }
        }
        __syncthreads();
        if (tile + n_split < total) issue(tile + n_split);
        if constexpr (sizeof(T) == 2) {
            typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
            const int gq = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3, colblk = gq & 1;     // gq >> 1 == hk
            bf16x8 ones;
#pragma unroll
            for (int j = 0; j < 8; ++j) ones[j] = (bf16)1.0f;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int m0 = (wave * 2 + c) * 16 + 8 * hk;                                // 8 consecutive-w positions m0 .. m0+7
                bf16x8 a;
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(s_lds + (m0 + 4 * jj + q) * 32 + 16 * colblk + 4 * p));
                    a[4 * jj + 0] = v[0]; a[4 * jj + 1] = v[1]; a[4 * jj + 2] = v[2]; a[4 * jj + 3] = v[3];
                }
                if (want_bias) accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, ones, accb, 0, 0, 0);
                const int w = m0 % TW, hh = m0 / TW % TH, d = m0 / (TW * TH);
                const int pb = ((2 * d) * IH + 2 * hh) * LPITCH + w;            // w is a multiple of 8: 16-byte aligned in every plane
#pragma unroll
                for (int s = 0; s < NTS; ++s) {
                    union { uint4 u; bf16x8 v; } bv;
                    bv.u = *(const uint4*)(l_lds + pb + toff[s]);
                    if (!tvalid[s]) bv.u = make_uint4(0u, 0u, 0u, 0u);
                    acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bv.v, acc[s], 0, 0, 0);
                }
            }
        } else {
#pragma unroll 4
            for (int jj = 0; jj < 16; ++jj) {
                const int m = wave * 32 + 2 * jj + hk;
                const int w = m % TW, hh = m / TW % TH, d = m / (TW * TH);
                const int pb = ((2 * d) * IH + 2 * hh) * IW + 2 * w;
                const float a = s_lds[m * 32 + r];
                if (want_bias) accb = __builtin_amdgcn_mfma_f32_32x32x2f32(a, 1.0f, accb, 0, 0, 0);
#pragma unroll
                for (int s = 0; s < NTS; ++s) {
                    const float bv = tvalid[s] ? l_lds[pb + toff[s]] : 0.f;
                    acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[s], 0, 0, 0);
                }
            }
        }
    }
    // ---- the 4 waves hold partial sums over different positions: combine them in LDS and store ONE partial slab per
    // workgroup with plain stores; wgrad_c1_finish sums the slabs (no atomics: fp32 atomics of hundreds of workgroups onto
    // the same 8 KB run at ~1/14 of the spread-out rate, MI355X_MICROARCH.md "Global float atomics") ----
    constexpr int RW = NTS * 32 + 1;
    __shared__ float red[4][32 * RW];
#pragma unroll
    for (int s = 0; s < NTS; ++s)
#pragma unroll
        for (int e = 0; e < 16; ++e) red[wave][((e & 3) + 8 * (e >> 2) + 4 * hk) * RW + s * 32 + r] = acc[s][e];
    if (r == 0) {
#pragma unroll
        for (int e = 0; e < 16; ++e) red[wave][((e & 3) + 8 * (e >> 2) + 4 * hk) * RW + NTS * 32] = accb[e];
    }
    __syncthreads();
    float* slab = ws + ((size_t)blockIdx.y * n_split + blockIdx.x) * (32 * RW);
    for (int i = t; i < 32 * RW; i += 256) slab[i] = red[0][i] + red[1][i] + red[2][i] + red[3][i];
    if (LSUM && want_lsum) {                                 // block-uniform
        __shared__ float lred[4];
        const float v = wave_sum(lacc);
        if (lane == 0) lred[wave] = v;
        __syncthreads();
        if (t == 0) lsum_ws[blockIdx.x] = lred[0] + lred[1] + lred[2] + lred[3];
    }
}

// Sum the per-workgroup slabs: block (x, y) owns 16 outputs of channel block y; its 256 threads are 16 outputs x 16 slab groups, every
// group walks n_split / 16 slabs, LDS adds the groups.  Plain stores into dW / dbias: no zero-fill, no atomics, fixed summation order.
template <int ND>
__global__ __launch_bounds__(256) void wgrad_c1_finish_kernel(const float* __restrict__ ws, float* __restrict__ dW, float* __restrict__ dbias, int n_split,
                                                              const float* __restrict__ lsum_ws, float* __restrict__ dbias_l, int main_blocks) {
    constexpr int TAPS = (ND == 3) ? 64 : 16, NTS = (TAPS + 31) / 32, RW = NTS * 32 + 1;
    if ((int)blockIdx.x >= main_blocks) {                    // one extra block (channel block 0 only): the L-side bias = sum of the partial sums
        if (blockIdx.y != 0 || !dbias_l) return;
        __shared__ float lred[4];
        float v = 0.f;
        for (int i = threadIdx.x; i < n_split; i += 256) v += lsum_ws[i];
        v = wave_sum(v);
        if ((threadIdx.x & 63) == 0) lred[threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0) dbias_l[0] = lred[0] + lred[1] + lred[2] + lred[3];
        return;
    }
    __shared__ float part[16][17];
    const int ol = threadIdx.x & 15, q = threadIdx.x >> 4, o = blockIdx.x * 16 + ol;
    const float* base = ws + (size_t)blockIdx.y * n_split * (32 * RW);
    float acc = 0.f;
    if (o < 32 * RW) {
#pragma unroll 8
        for (int x = q; x < n_split; x += 16) acc += base[(size_t)x * (32 * RW) + o];
    }
    part[q][ol] = acc;
    __syncthreads();
    if (q == 0 && o < 32 * RW) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) v += part[k][ol];
        const int row = o / RW, col = o % RW, cs = blockIdx.y * 32 + row;
        if (col == NTS * 32) { if (dbias) dbias[cs] = v; }
        else if (col < TAPS) dW[(size_t)cs * TAPS + col] = v;
    }
}


}  // namespace

// 1 when the vector-load form can read this image: whole 16-byte vectors per row and a 16-byte aligned base
static bool image_vec_ok(const void* L, int64_t lw, int l_dtype) {
    const int epv = l_dtype == CVAE_BF16 ? 8 : 4;
    return lw % epv == 0 && lw >= epv && (((uintptr_t)L) & 15) == 0;
}
int cvae_conv_down_c1(const void* L, int l_dtype, const float* w, const float* bias, const void* mask, void* S, int64_t B, int64_t sd, int64_t sh, int64_t sw,
                      int64_t Cs, int64_t ld, int64_t lh, int64_t lw, int nd, int dtype, int act, hipStream_t stream, F8Side f8) {
    if (Cs != 32) return CVAE_E_UNSUPPORTED;
    const int epi = CVAE_EPI_OF(act);
    const bool side8 = f8.out8 || f8.amax || f8.bits_out;
    if (side8 && !(dtype == CVAE_BF16 && image_vec_ok(L, lw, l_dtype) && !mask && !f8.mask_bits)) return CVAE_E_UNSUPPORTED;
    if (f8.mask_bits && !(dtype == CVAE_BF16 && image_vec_ok(L, lw, l_dtype))) return CVAE_E_UNSUPPORTED;
    if (dtype == CVAE_BF16 && image_vec_ok(L, lw, l_dtype)) {          // bf16 output: the 16-byte-load form, image in its own dtype
        // 512 positions per workgroup once that still leaves ~4 workgroups per CU, else 256
        const bool big = B * ((sd + 1) / 2) * ((sh + 7) / 8) * ((sw + 31) / 32) >= 1024 || nd == 2;
        const int ms = (nd == 3 && !big) ? 2 : 4;
        const int td = (nd == 3) ? ms / 2 : 1, th = (nd == 3) ? 8 : 8 * ms, tw = (nd == 3) ? 32 : 16;
        const int tiles_d = (int)((sd + td - 1) / td), tiles_h = (int)((sh + th - 1) / th), tiles_w = (int)((sw + tw - 1) / tw);
        const long long ntiles_ll = (long long)B * tiles_d * tiles_h * tiles_w;
        if (ntiles_ll > 0x7fffffff) return CVAE_E_BADSHAPE;
        const int ntiles = (int)ntiles_ll;
        dim3 grid((unsigned)(ntiles < CVAE_C1_MAX_WG ? ntiles : CVAE_C1_MAX_WG), 1, 1);       // <= 4 resident workgroups per CU; each walks ntiles / grid tiles
#define LAUNCH_DOWN_VEC__(TLT, ND, EPI, MASKED, MS, SIDE)                                                                                              \
    hipLaunchKernelGGL((down_c1_vec_kernel<TLT, ND, EPI, MASKED, MS, SIDE>), grid, dim3(256), 0, stream, (const TLT*)L, w, bias, (const bf16*)mask, (bf16*)S, (int)sd, (int)sh, \
                       (int)sw, (int)ld, (int)lh, (int)lw, tiles_d, tiles_h, tiles_w, ntiles, act, f8)
#define LAUNCH_DOWN_VEC_(TLT, ND, EPI, MS) do { if (mask || f8.mask_bits) LAUNCH_DOWN_VEC__(TLT, ND, EPI, true, MS, false); else if (side8) LAUNCH_DOWN_VEC__(TLT, ND, EPI, false, MS, true); \
                                                else LAUNCH_DOWN_VEC__(TLT, ND, EPI, false, MS, false); } while (0)
#define LAUNCH_DOWN_VEC(TLT, ND, MS)                                                                                                      \
    do { if (epi == 0) LAUNCH_DOWN_VEC_(TLT, ND, 0, MS); else if (epi == 1) LAUNCH_DOWN_VEC_(TLT, ND, 1, MS); else LAUNCH_DOWN_VEC_(TLT, ND, 2, MS); } while (0)
        if (l_dtype == CVAE_F32) { if (nd == 2) LAUNCH_DOWN_VEC(float, 2, 4); else if (ms == 4) LAUNCH_DOWN_VEC(float, 3, 4); else LAUNCH_DOWN_VEC(float, 3, 2); }
        else { if (nd == 2) LAUNCH_DOWN_VEC(bf16, 2, 4); else if (ms == 4) LAUNCH_DOWN_VEC(bf16, 3, 4); else LAUNCH_DOWN_VEC(bf16, 3, 2); }
#undef LAUNCH_DOWN_VEC
#undef LAUNCH_DOWN_VEC_
#undef LAUNCH_DOWN_VEC__
        CVAE_CHECK_LAUNCH();
        return CVAE_OK;
    }
    const int th = (nd == 3) ? 8 : 16, tw = (nd == 3) ? 8 : 16, td = (nd == 3) ? 4 : 1;
    const int tiles_d = (int)((sd + td - 1) / td), tiles_h = (int)((sh + th - 1) / th), tiles_w = (int)((sw + tw - 1) / tw);
    dim3 grid((unsigned)(tiles_d * tiles_h * tiles_w), 1, (unsigned)B);
    if (l_dtype != dtype) return CVAE_E_UNSUPPORTED;         // the element-wise form reads the image in the compute dtype
#define LAUNCH_DOWN_C1_(T, ND, EPI)                                                                                                   \
    hipLaunchKernelGGL((down_c1_kernel<T, ND, 32, EPI>), grid, dim3(256), 0, stream, (const T*)L, w, bias, (const T*)mask, (T*)S, (int)sd, (int)sh, \
                       (int)sw, (int)ld, (int)lh, (int)lw, tiles_h, tiles_w, act)
#define LAUNCH_DOWN_C1(T, ND)                                                                                                         \
    do { if (epi == 0) LAUNCH_DOWN_C1_(T, ND, 0); else if (epi == 1) LAUNCH_DOWN_C1_(T, ND, 1); else LAUNCH_DOWN_C1_(T, ND, 2); } while (0)
    if (dtype == CVAE_BF16) { if (nd == 3) LAUNCH_DOWN_C1(bf16, 3); else LAUNCH_DOWN_C1(bf16, 2); }
    else { if (nd == 3) LAUNCH_DOWN_C1(float, 3); else LAUNCH_DOWN_C1(float, 2); }
#undef LAUNCH_DOWN_C1
#undef LAUNCH_DOWN_C1_
    CVAE_CHECK_LAUNCH();
    return CVAE_OK;
}

int cvae_conv_up_c1(const void* S, const float* w, const float* bias, const void* mask, void* L, int64_t B, int64_t sd, int64_t sh, int64_t sw,
                    int64_t Cs, int64_t ld, int64_t lh, int64_t lw, int nd, int dtype, int act, hipStream_t stream, long long walk_units_arg) {
    const long long walk_min_units = walk_units_arg > 0 ? walk_units_arg : CVAE_C1U_WALK_MIN_UNITS;     // per call (cvae_conv_up_variant), no process-wide state
    if (Cs != 32) return CVAE_E_UNSUPPORTED;
    const int64_t n = B * sd * sh * sw;                     // one thread per (source voxel, channel half)
    if (n >= ((int64_t)1 << 30) || (nd == 3 && ld != 2 * sd) || lh != 2 * sh || lw != 2 * sw) return CVAE_E_UNSUPPORTED;   // exact 2x only (depth counts in 3D)
    if (dtype == CVAE_BF16) {                               // MFMA form
        const int td = (nd == 3) ? 2 : 1, th = (nd == 3) ? 8 : 16, tw = 16;                 // TileC1U
        const int tiles_d = (int)((sd + td - 1) / td), tiles_h = (int)((sh + th - 1) / th), tiles_w = (int)((sw + tw - 1) / tw);
        const long long ntiles_ll = (long long)B * tiles_d * tiles_h * tiles_w;
        if (ntiles_ll > 0x7fffffff) return CVAE_E_BADSHAPE;
        const int ntiles = (int)ntiles_ll;
        dim3 mgrid((unsigned)(ntiles < CVAE_C1_MAX_WG ? ntiles : CVAE_C1_MAX_WG), 1, 1);      // a workgroup walks ntiles / grid tiles with one set of weight fragments
        const int epi = CVAE_EPI_OF(act);
#if CVAE_C1U_WALK
        if (nd == 3 && ntiles >= 2 * walk_min_units && tiles_d > 1) {   // long z columns: walk them, the shared halo planes stay in LDS
            int walk = (int)(ntiles / walk_min_units);
            if (walk > tiles_d) walk = tiles_d;
            const int segs = (tiles_d + walk - 1) / walk;
            const int nunits = (int)B * segs * tiles_h * tiles_w;
            dim3 wgrid((unsigned)(nunits < CVAE_C1_MAX_WG ? nunits : CVAE_C1_MAX_WG), 1, 1);
#define LAUNCH_UP_WALK_(EPI, MASKED) hipLaunchKernelGGL((up_c1_mfma_walk_kernel<EPI, MASKED>), wgrid, dim3(256), 0, stream, (const bf16*)S, w, bias, (const bf16*)mask, (bf16*)L, (int)sd, (int)sh, (int)sw, tiles_d, tiles_h, tiles_w, walk, segs, nunits, act, 1.f)
#define LAUNCH_UP_WALK(EPI) do { if (mask) LAUNCH_UP_WALK_(EPI, true); else LAUNCH_UP_WALK_(EPI, false); } while (0)
            if (epi == 0) LAUNCH_UP_WALK(0); else if (epi == 1) LAUNCH_UP_WALK(1); else LAUNCH_UP_WALK(2);
#undef LAUNCH_UP_WALK
#undef LAUNCH_UP_WALK_
            CVAE_CHECK_LAUNCH();
            return CVAE_OK;
        }
#endif
#define LAUNCH_UP_MFMA_(ND, EPI, MASKED) hipLaunchKernelGGL((up_c1_mfma_kernel<ND, EPI, MASKED>), mgrid, dim3(256), 0, stream, (const bf16*)S, w, bias, (const bf16*)mask, (bf16*)L, (int)sd, (int)sh, (int)sw, tiles_d, tiles_h, tiles_w, ntiles, act)
#define LAUNCH_UP_MFMA(ND, EPI) do { if (mask) LAUNCH_UP_MFMA_(ND, EPI, true); else LAUNCH_UP_MFMA_(ND, EPI, false); } while (0)
        if (nd == 3) { if (epi == 0) LAUNCH_UP_MFMA(3, 0); else if (epi == 1) LAUNCH_UP_MFMA(3, 1); else LAUNCH_UP_MFMA(3, 2); }
        else { if (epi == 0) LAUNCH_UP_MFMA(2, 0); else if (epi == 1) LAUNCH_UP_MFMA(2, 1); else LAUNCH_UP_MFMA(2, 2); }
#undef LAUNCH_UP_MFMA
#undef LAUNCH_UP_MFMA_
        CVAE_CHECK_LAUNCH();
        return CVAE_OK;
    }
    dim3 grid((unsigned)((2 * n + 255) / 256));
#define LAUNCH_UP_C1(T, ND)                                                                                                          \
    hipLaunchKernelGGL((up_c1_kernel<T, ND, 32>), grid, dim3(256), 0, stream, (const T*)S, w, bias, (const T*)mask, (T*)L, (int)B, (int)sd,   \
                       (int)sh, (int)sw, (int)ld, (int)lh, (int)lw, act)
    if (nd == 3) LAUNCH_UP_C1(float, 3); else LAUNCH_UP_C1(float, 2);      // bf16 took the MFMA form above
#undef LAUNCH_UP_C1
    CVAE_CHECK_LAUNCH();
    return CVAE_OK;
}

// The single-channel ConvTranspose3d end fed by fp8 (e4m3) codes: L (bf16) = act(in_scale * (S8 (*) w) + bias).  3D, Cs == 32, exact 2x.
int cvae_conv_up_c1_fp8in_impl(const void* S8, const float* w, const float* bias, void* L, float in_scale, int64_t B, int64_t sd, int64_t sh, int64_t sw, int64_t Cs, int act,
                          hipStream_t stream) {
    if (Cs != 32) return CVAE_E_UNSUPPORTED;
    if (B * sd * sh * sw >= ((int64_t)1 << 30)) return CVAE_E_UNSUPPORTED;
    const int td = 2, th = 8, tw = 16;                       // TileC1U<3>
    const int tiles_d = (int)((sd + td - 1) / td), tiles_h = (int)((sh + th - 1) / th), tiles_w = (int)((sw + tw - 1) / tw);
    const long long ntiles = (long long)B * tiles_d * tiles_h * tiles_w;
    if (ntiles > 0x7fffffff) return CVAE_E_BADSHAPE;
    int walk = (int)(ntiles / CVAE_C1U_WALK_MIN_UNITS);
    if (walk < 1) walk = 1;
    if (walk > tiles_d) walk = tiles_d;
    const int segs = (tiles_d + walk - 1) / walk;
    const int nunits = (int)B * segs * tiles_h * tiles_w;
    dim3 wgrid((unsigned)(nunits < CVAE_C1_MAX_WG ? nunits : CVAE_C1_MAX_WG), 1, 1);
    const int epi = CVAE_EPI_OF(act);
#define LAUNCH_UP_WALK8(EPI) hipLaunchKernelGGL((up_c1_mfma_walk_kernel<EPI, false, fp8>), wgrid, dim3(256), 0, stream, (const fp8*)S8, w, bias, (const bf16*)nullptr, (bf16*)L, (int)sd, (int)sh, (int)sw, tiles_d, tiles_h, tiles_w, walk, segs, nunits, act, in_scale)
    if (epi == 0) LAUNCH_UP_WALK8(0); else if (epi == 1) LAUNCH_UP_WALK8(1); else LAUNCH_UP_WALK8(2);
#undef LAUNCH_UP_WALK8
    CVAE_CHECK_LAUNCH();
    return CVAE_OK;
}

size_t cvae_conv_wgrad_c1_workspace_bytes(int64_t Cs, int nd) {
    const int rw = ((nd == 3) ? 64 : 32) + 1;
    return ((size_t)2048 * 32 * rw + 2048) * sizeof(float);  // (Cs/32) * n_split <= 2048 slabs of [32][rw], then <= 2048 partial sums of L
}

int cvae_conv_wgrad_c1(const void* S, const void* L, int l_dtype, float* dW, float* dbias, float* dbias_l, void* workspace, size_t workspace_bytes, int64_t B, int64_t sd, int64_t sh,
                       int64_t sw, int64_t Cs, int64_t ld, int64_t lh, int64_t lw, int nd, int dtype, hipStream_t stream) {
    if (l_dtype != dtype && !(dtype == CVAE_BF16 && l_dtype == CVAE_F32)) return CVAE_E_UNSUPPORTED;     // mixed form: fp32 image, bf16 gradient
    // dbias: per-channel sum of S (Conv bias) or NULL; dbias_l: sum of L (ConvTranspose bias, one value; needs L == 2 S) or NULL
    if (dbias_l && (lh != 2 * sh || lw != 2 * sw || (nd == 3 && ld != 2 * sd))) return CVAE_E_UNSUPPORTED;
    if (Cs % 32 || Cs > 1024 * 32) return CVAE_E_UNSUPPORTED;
    if (!workspace) return CVAE_E_NULLPTR;
    if (workspace_bytes < cvae_conv_wgrad_c1_workspace_bytes(Cs, nd)) return CVAE_E_WORKSPACE;
    const int td = (nd == 3) ? C1W_TD3 : 1, th = (nd == 3) ? C1W_TH3 : 8, tw = (nd == 3) ? C1W_TW3 : 16;
    const int tiles_d = (int)((sd + td - 1) / td), tiles_h = (int)((sh + th - 1) / th), tiles_w = (int)((sw + tw - 1) / tw);
    const long long total = (long long)B * tiles_d * tiles_h * tiles_w;
    // 2 workgroups per CU (256 CUs): measured in the step at 256 / 384 / 512 / 640 / 768 / 1024 / 2048 workgroups: 58 / 45 / 36 / 49 / 43 / 39 / 44 us
    // for enc1 — whole multiples of the CU count, and as few slabs as keep the loads in flight; slabs leave with plain stores
#ifdef CVAE_TUNE                                             // tuning builds only (make EXTRA=-DCVAE_TUNE); clamped to the validated workspace
    static const int c1_wgs = getenv("CVAE_TUNE_C1_SLABS") ? (atoi(getenv("CVAE_TUNE_C1_SLABS")) > 2048 ? 2048 : atoi(getenv("CVAE_TUNE_C1_SLABS"))) : 512;
#else
    constexpr int c1_wgs = 512;
#endif
    long long n_split = c1_wgs / (Cs / 32);
    if (n_split > total) n_split = total;
    if (n_split < 1) n_split = 1;
    dim3 grid((unsigned)n_split, (unsigned)(Cs / 32), 1);
    float* ws = (float*)workspace;
    const int rw = ((nd == 3) ? 64 : 32) + 1;
    float* lsum_ws = dbias_l ? ws + (size_t)2048 * 32 * rw : nullptr;
    const bool vec = image_vec_ok(L, lw, l_dtype);
#define LAUNCH_WG_C1(T, ND)                                                                                                           \
    if (lsum_ws) LAUNCH_WG_C1_(T, ND, true); else LAUNCH_WG_C1_(T, ND, false)
#define LAUNCH_WG_C1__(T, TLT, ND, LS, VEC)                                                                                           \
    hipLaunchKernelGGL((wgrad_c1_kernel<T, TLT, ND, LS, VEC>), grid, dim3(256), 0, stream, (const T*)S, (const TLT*)L, ws, dbias != nullptr, (int)B, (int)sd, (int)sh, \
                       (int)sw, (int)Cs, (int)ld, (int)lh, (int)lw, tiles_d, tiles_h, tiles_w, (int)n_split, lsum_ws)
#define LAUNCH_WG_C1_(T, ND, LS)                                                                                                      \
    do {                                                                                                                              \
        constexpr bool can_vec = sizeof(T) == 2;                                                                                      \
        if (l_dtype == dtype) { if (can_vec && vec) LAUNCH_WG_C1__(T, T, ND, LS, can_vec); else LAUNCH_WG_C1__(T, T, ND, LS, false); } \
        else { if (can_vec && vec) LAUNCH_WG_C1__(T, float, ND, LS, can_vec); else LAUNCH_WG_C1__(T, float, ND, LS, false); }          \
    } while (0)
    if (dtype == CVAE_BF16) { if (nd == 3) { LAUNCH_WG_C1(bf16, 3); } else { LAUNCH_WG_C1(bf16, 2); } }
    else { if (nd == 3) { LAUNCH_WG_C1(float, 3); } else { LAUNCH_WG_C1(float, 2); } }
#undef LAUNCH_WG_C1
#undef LAUNCH_WG_C1_
#undef LAUNCH_WG_C1__
    CVAE_CHECK_LAUNCH();
    const int main_blocks = (32 * rw + 15) / 16;
    dim3 fgrid((unsigned)(main_blocks + (dbias_l ? 1 : 0)), (unsigned)(Cs / 32), 1);
    if (nd == 3) hipLaunchKernelGGL(wgrad_c1_finish_kernel<3>, fgrid, dim3(256), 0, stream, ws, dW, dbias, (int)n_split, lsum_ws, dbias_l, main_blocks);
    else hipLaunchKernelGGL(wgrad_c1_finish_kernel<2>, fgrid, dim3(256), 0, stream, ws, dW, dbias, (int)n_split, lsum_ws, dbias_l, main_blocks);
    CVAE_CHECK_LAUNCH();
    return CVAE_OK;
}
