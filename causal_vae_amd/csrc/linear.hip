// linear.hip — fp32 nn.Linear forward / backward-data / backward-weight on the exact-fp32 MFMA
// (v_mfma_f32_16x16x4_f32: bitwise an fmaf chain, MI355X_MICROARCH.md "Matrix cores").
//
// One strided GEMM kernel serves the three products by choosing strides, so no operand is ever transposed in HBM:
//     C[m, n] (+)= sum_k A[m*sam + k*sak] * B[k*sbk + n*sbn]
//   forward     : A = x  (sam = ldx, sak = 1) , B = W^T (sbk = 1,   sbn = K)  -> y  [M, N]
//   bwd-data    : A = dy (sam = ldy, sak = 1) , B = W   (sbk = K,   sbn = 1)  -> dx [M, K]
//   bwd-weight  : A = dy^T (sam = 1, sak = ldy), B = x  (sbk = ldx, sbn = 1)  -> dW [N, K]
// Workgroup = 256 threads = 2x2 waves, tile 64x64x16, each wave 32x32 as 2x2 MFMA tiles.  The 3D model's
// enc_fc[0] is a 16415x512 weight read by a batch of 4: pure weight streaming, so K is split across
// workgroups until the grid covers the chip: every split leaves its partial tile in a slab of the caller's workspace
// ([split][M][N], plain stores) and one finish launch adds the slabs in index order, then bias and activation — no float
// atomics, so results are bit-reproducible.  Without (enough) workspace the same kernels run unsplit.
// Opt-in bf16-operand forms (Linear.math = bfloat16: fp32 in memory, rounded on the way into LDS, fp32 accumulate and output): gemm_bf16_kernel (the same
// 64 x 64 tiling) and, for products with M, N, K >= 128, gemm_bf16_t128_kernel (128 x 128 tiles, vector operand fetch, transposed LDS reads).
#include "common.h"
#include <type_traits>

// Optional last step of an epilogue: C[m][n] *= act'(src[m][n]) with act' taken from the activation's OUTPUT (common.h act_grad_from_out) — the data gradient of a
// linear layer whose input was the previous layer's activation output leaves the GEMM already multiplied by that activation's derivative (one elementwise
// launch fewer per layer; the same fp32 product the separate pass would form).
struct EpiMul { const float* src; int64_t ld; int act; };
__device__ __forceinline__ float epi_mul(float v, const EpiMul& em, int64_t gm, int64_t gn) {
    return em.src ? v * act_grad_from_out(em.src[gm * em.ld + gn], em.act) : v;
}

#define LT 64
#define LK 16
#define AS_STRIDE 17   // As[64][17]  : A-fragment reads (lane -> row) conflict-free
#define BS_STRIDE 80   // Bs[16][80]  : B-fragment reads (two k rows per 32-lane half) conflict-free

__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, const float* __restrict__ Bm, float* __restrict__ C,
                                                       const float* __restrict__ bias, int64_t M, int64_t N, int64_t K,
                                                       int64_t sam, int64_t sak, int64_t sbk, int64_t sbn, int64_t ldc,
                                                       int64_t k_per_split, int act, float* __restrict__ slabs, EpiMul em) {
    __shared__ float As[LT * AS_STRIDE];
    __shared__ float Bs[LK * BS_STRIDE];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int64_t m0 = (int64_t)blockIdx.y * LT, n0 = (int64_t)blockIdx.x * LT;
    const int64_t kbeg = (int64_t)blockIdx.z * k_per_split;
    const int64_t kend = min(K, kbeg + k_per_split);
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const bool a_k_contig = (sak == 1), b_n_contig = (sbn == 1);
    // The tile of step k0 + LK is requested into registers before the tile of step k0 is multiplied: without that every k-step paid a full
    // global round trip (~2.3 us per 16 k measured: 150 us for ANY weight gradient over a 1024-row batch, however small the layer).
    float ra[4], rb[4];
    auto fetch = [&](int64_t k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int m, k;
            if (a_k_contig) { k = t & 15; m = (t >> 4) + 16 * i; } else { m = t & 63; k = (t >> 6) + 4 * i; }
            const int64_t gm = m0 + m, gk = k0 + k;
            ra[i] = (gm < M && gk < kend) ? A[gm * sam + gk * sak] : 0.f;
            int n, kb;
            if (b_n_contig) { n = t & 63; kb = (t >> 6) + 4 * i; } else { kb = t & 15; n = (t >> 4) + 16 * i; }
            const int64_t gn = n0 + n, gkb = k0 + kb;
            rb[i] = (gn < N && gkb < kend) ? Bm[gkb * sbk + gn * sbn] : 0.f;
        }
    };
    if (kbeg < kend) fetch(kbeg);
    for (int64_t k0 = kbeg; k0 < kend; k0 += LK) {
        // ---- stage A tile [64 m][16 k] and B tile [16 k][64 n] (4 elements per thread each) ----
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int m, k;
            if (a_k_contig) { k = t & 15; m = (t >> 4) + 16 * i; } else { m = t & 63; k = (t >> 6) + 4 * i; }
            As[m * AS_STRIDE + k] = ra[i];
            int n, kb;
            if (b_n_contig) { n = t & 63; kb = (t >> 6) + 4 * i; } else { kb = t & 15; n = (t >> 4) + 16 * i; }
            Bs[kb * BS_STRIDE + n] = rb[i];
        }
        __syncthreads();
        if (k0 + LK < kend) fetch(k0 + LK);
#pragma unroll
        for (int kk = 0; kk < LK / 4; ++kk) {
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = As[(wm * 32 + i * 16 + (lane & 15)) * AS_STRIDE + kk * 4 + (lane >> 4)];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = Bs[(kk * 4 + (lane >> 4)) * BS_STRIDE + wn * 32 + j * 16 + (lane & 15)];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    // ---- epilogue: C/D map of 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg ----
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t gm = m0 + wm * 32 + i * 16 + (lane >> 4) * 4 + r, gn = n0 + wn * 32 + j * 16 + (lane & 15);
                if (gm < M && gn < N) {
                    float v = acc[i][j][r];
                    if (slabs) slabs[((size_t)blockIdx.z * M + gm) * N + gn] = v;          // split-K partial: slab blockIdx.z
                    else C[gm * ldc + gn] = epi_mul(apply_act(v + (bias ? bias[gn] : 0.f), act), em, gm, gn);
                }
            }
}

// The same GEMM with bf16 operands on v_mfma_f32_32x32x16_bf16 (fp32 in memory, rounded to bf16 on the way into LDS, fp32 accumulate, fp32 out):
// 16x the MFMA rate of the exact-fp32 form for the large MNIST linears (1024 x 3158 x 512), where a config named bf16 ran 68 % of its step in
// fp32 GEMMs.  Same tile (64 x 64, 4 waves of 32 x 32), same split-K slabs and strides, 32 k per stage; opt-in per model (Linear.math).
#define LK16 32
#define S16 40           // LDS row pitch in bf16: 80 B = 5 x 16 B, so the 16-byte fragment reads of 16 consecutive rows fall in 16 different bank slots
__global__ __launch_bounds__(256) void gemm_bf16_kernel(const float* __restrict__ A, const float* __restrict__ Bm, float* __restrict__ C,
                                                        const float* __restrict__ bias, int64_t M, int64_t N, int64_t K,
                                                        int64_t sam, int64_t sak, int64_t sbk, int64_t sbn, int64_t ldc,
                                                        int64_t k_per_split, int act, float* __restrict__ slabs, EpiMul em) {
    __shared__ __attribute__((aligned(16))) bf16 As[LT * S16];
    __shared__ __attribute__((aligned(16))) bf16 Bs[LT * S16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int64_t m0 = (int64_t)blockIdx.y * LT, n0 = (int64_t)blockIdx.x * LT;
    const int64_t kbeg = (int64_t)blockIdx.z * k_per_split;
    const int64_t kend = min(K, kbeg + k_per_split);
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const bool a_k_contig = (sak == 1), b_n_contig = (sbn == 1);
    float ra[8], rb[8];
    auto fetch = [&](int64_t k0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int m, k;
            if (a_k_contig) { k = t & 31; m = (t >> 5) + 8 * i; } else { m = t & 63; k = (t >> 6) + 4 * i; }
            const int64_t gm = m0 + m, gk = k0 + k;
            ra[i] = (gm < M && gk < kend) ? A[gm * sam + gk * sak] : 0.f;
            int n, kb;
            if (b_n_contig) { n = t & 63; kb = (t >> 6) + 4 * i; } else { kb = t & 31; n = (t >> 5) + 8 * i; }
            const int64_t gn = n0 + n, gkb = k0 + kb;
            rb[i] = (gn < N && gkb < kend) ? Bm[gkb * sbk + gn * sbn] : 0.f;
        }
    };
    if (kbeg < kend) fetch(kbeg);
    for (int64_t k0 = kbeg; k0 < kend; k0 += LK16) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int m, k;
            if (a_k_contig) { k = t & 31; m = (t >> 5) + 8 * i; } else { m = t & 63; k = (t >> 6) + 4 * i; }
            As[m * S16 + k] = (bf16)ra[i];
            int n, kb;
            if (b_n_contig) { n = t & 63; kb = (t >> 6) + 4 * i; } else { kb = t & 31; n = (t >> 5) + 8 * i; }
            Bs[n * S16 + kb] = (bf16)rb[i];
        }
        __syncthreads();
        if (k0 + LK16 < kend) fetch(k0 + LK16);
#pragma unroll
        for (int kk = 0; kk < LK16 / 16; ++kk) {
            const bf16x8 a = *(const bf16x8*)(As + (wm * 32 + r) * S16 + kk * 16 + 8 * h);
            const bf16x8 b = *(const bf16x8*)(Bs + (wn * 32 + r) * S16 + kk * 16 + 8 * h);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    // D[i][j]: lane (r, h) holds column j = r, rows i = (e & 3) + 8 (e >> 2) + 4 h
    const int64_t gn = n0 + wn * 32 + r;
    if (gn < N) {
        const float bv = (bias && !slabs) ? bias[gn] : 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int64_t gm = m0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (gm < M) {
                if (slabs) slabs[((size_t)blockIdx.z * M + gm) * N + gn] = acc[e];
                else C[gm * ldc + gn] = epi_mul(apply_act(acc[e] + bv, act), em, gm, gn);
            }
        }
    }
}

#ifndef CVAE_GEMM_T128
#define CVAE_GEMM_T128 1
#endif
#ifndef CVAE_GEMM_T128_DEPTH
#define CVAE_GEMM_T128_DEPTH 2
#endif
#ifndef CVAE_GEMM_T128_BK
#define CVAE_GEMM_T128_BK 32
#endif
#ifndef CVAE_GEMM_T128_WGS
#define CVAE_GEMM_T128_WGS 256
#endif
// The large-product form of the bf16-operand GEMM (M, N >= 128): 128 x 128 tiles, 4 waves of 64 x 64 (2 x 2 MFMA tiles: every fragment read feeds two
// MFMAs), fp32 operands fetched as 16-byte vectors along whichever dimension is contiguous and written to LDS as 8-byte bf16 quads:
//   * an operand that is k-contiguous in memory (x, W in the forward) gets a [row][k] image (pitch BK + 8) read with ds_read_b128;
//   * an operand that is row-contiguous (g^T and x in the weight gradient, W in the data gradient) keeps its memory order in a [k][row] image
//     (pitch 160: the four k-rows of a transposed read fall in four different 64-byte bank groups) and is read with ds_read_b64_tr_b16 — no
//     scalar transposing stores.
// gemm_bf16_kernel above moved 4-byte loads and 2-byte LDS stores, 64 x 64 tiles: 80 TFLOP/s on 1024 x 3158 x 512 (38-42 us per product).
struct __attribute__((packed, aligned(4))) F4U { float x, y, z, w; };       // float4 at dword alignment (rows of odd length)
// out[e] = e < nvalid ? L[e + sh] : 0 (sh in 0..3; e + sh <= 3 wherever e < nvalid): the fix-up of a vector that was loaded from a start clamped into the row
__device__ __forceinline__ F4U shift_sel(F4U L, int sh, int nvalid) {
    F4U o;
    o.x = sh == 0 ? L.x : (sh == 1 ? L.y : (sh == 2 ? L.z : L.w));
    o.y = sh == 0 ? L.y : (sh == 1 ? L.z : (sh == 2 ? L.w : 0.f));
    o.z = sh == 0 ? L.z : (sh == 1 ? L.w : 0.f);
    o.w = sh == 0 ? L.w : 0.f;
    if (nvalid < 1) o.x = 0.f;
    if (nvalid < 2) o.y = 0.f;
    if (nvalid < 3) o.z = 0.f;
    if (nvalid < 4) o.w = 0.f;
    return o;
}
template <bool A_KC, bool B_KC, int BK>
__global__ __launch_bounds__(256) void gemm_bf16_t128_kernel(const float* __restrict__ A, const float* __restrict__ Bm, float* __restrict__ C,
                                                             const float* __restrict__ bias, int64_t M, int64_t N, int64_t K,
                                                             int64_t sam, int64_t sak, int64_t sbk, int64_t sbn, int64_t ldc,
                                                             int64_t k_per_split, int act, float* __restrict__ slabs, int tn, int tm, int n_slow, EpiMul em) {
    constexpr int BT = 128, PR = BK + 8, PT = 160, NL = BK / 8;      // NL: 16-byte loads per thread per operand per stage
    constexpr int IMG = (BT * PR > BK * PT) ? BT * PR : BK * PT;
    __shared__ __attribute__((aligned(16))) bf16 As[IMG];
    __shared__ __attribute__((aligned(16))) bf16 Bs[IMG];
    typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r = lane & 31, h = lane >> 5;
    const int gq = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3, colblk = gq & 1;     // transposed-read roles (T10): h == gq >> 1
    const int wm = wave >> 1, wn = wave & 1;
    // 1-D grid, XCD-aware (blocks b and b + 8 share an XCD and its L2): every XCD gets a contiguous run of the order (split, slow tile index, fast tile
    // index), i.e. the workgroups of one K slice — which re-read the same A rows and B rows — sit behind one L2.  In grid order the 4 n-tiles sharing an
    // A tile landed on 4 different XCDs and the forward moved 106 MB through the fabric for 19 MB of operands.  The larger operand's tile index is the
    // slow one (its tiles are fetched once per XCD).
    int lid;
    { const int nb = (int)gridDim.x, qq = nb >> 3, rr = nb & 7, xx = blockIdx.x & 7; lid = (xx < rr ? xx * (qq + 1) : rr * (qq + 1) + (xx - rr) * qq) + ((int)blockIdx.x >> 3); }
    const int tiles = tn * tm, split = lid / tiles, tl = lid - split * tiles;
    const int bx = n_slow ? tl / tm : tl % tn, by = n_slow ? tl % tm : tl / tn;
    const int64_t m0 = (int64_t)by * BT, n0 = (int64_t)bx * BT;
    const int64_t kbeg = (int64_t)split * k_per_split;
    const int64_t kend = min(K, kbeg + k_per_split);
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    // DEPTH stages of both operands in flight in registers: a workgroup has ~10 stages of 32 k and one workgroup sits on a CU, so with a single
    // stage in flight every stage cost a full global round trip (2 us per stage measured: 26 us for the 1024 x 3158 x 512 forward)
    constexpr int DEPTH = CVAE_GEMM_T128_DEPTH;
    F4U ra[DEPTH][NL], rb[DEPTH][NL];
    // one operand's stage: P[row * s_row + k * s_k], rows row0 .. row0 + 127 (< R), k in [k0, k0 + BK) (< kend)
    // Every load is an unconditional 16-byte vector from an address clamped into the operand (rows to R - 1 / R - 4, k to K - 4 / kend - 1); what the
    // clamp moved or what lies past kend is fixed up in registers when the stage is written to LDS (stash) — and only there, behind block-uniform
    // branches that contain no load.  No branch surrounds a load and the stage loop is straight-line code: with guarded loads or conditional fetches
    // the compiler's s_waitcnt insertion fell back to vmcnt(0) before every LDS write (one global round trip per stage whatever the prefetch depth).
    // Rows past R need no zeroing: they only reach outputs that are never stored.  Offsets are 32-bit (the host checks the operands' extents) and
    // the row part is hoisted: a wave64 VALU instruction costs 4 cycles, and 64-bit address arithmetic plus unconditional fix-up selects for 8
    // vectors per stage had the loop VALU-bound (1.2 us per stage for 256 cycles of MFMA).
    constexpr int QR = BK / 4, RS = 256 / QR;                // k-contiguous image: thread -> rows t / QR + RS i, k = 4 (t % QR)
    const int kq = 4 * (t % QR), kl = NL * (t >> 5);         // row-contiguous image: thread -> rows 4 (t % 32) .. + 3, k = kl + j
    const int Ki = (int)K, kendi = (int)kend;
    int rowA[NL], rowB[NL];                                  // hoisted row parts (k-contiguous: one per vector; row-contiguous: [0] only)
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        rowA[i] = A_KC ? (int)(min(m0 + t / QR + RS * i, M - 1) * sam) : (int)min(m0 + 4 * (t & 31), M - 4);
        rowB[i] = B_KC ? (int)(min(n0 + t / QR + RS * i, N - 1) * sbn) : (int)min(n0 + 4 * (t & 31), N - 4);
    }
    auto fetch = [&](auto kc, const float* __restrict__ P, const int (&rowo)[NL], int s_k, int k0, F4U (&rg)[NL]) {
        if constexpr (decltype(kc)::value) {
            const int d = min(k0 + kq, Ki - 4);
#pragma unroll
            for (int i = 0; i < NL; ++i) rg[i] = *(const F4U*)(P + (rowo[i] + d));
        } else {
#pragma unroll
            for (int j = 0; j < NL; ++j) rg[j] = *(const F4U*)(P + (min(k0 + kl + j, kendi - 1) * s_k + rowo[0]));
        }
    };
    auto stash = [&](auto kc, bf16* img, const F4U (&rg)[NL], int row0, int R, int k0) {
        const bool kpart = k0 + BK > kendi;                  // block-uniform: only the last stage of the last split, or a padding stage
        if constexpr (decltype(kc)::value) {
            const int gk = k0 + kq, sh = gk - min(gk, Ki - 4), nv = max(0, min(4, kendi - gk));
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                F4U v = rg[i];
                if (kpart) v = shift_sel(v, sh, nv);
                *(uint2*)(img + (t / QR + RS * i) * PR + kq) = make_uint2(pack2_bf16(v.x, v.y), pack2_bf16(v.z, v.w));
            }
        } else {
            const bool redge = row0 + BT > R;                // block-uniform: the last row tile of an extent that is not a multiple of 128
            const int g = row0 + 4 * (t & 31), sh = g - min(g, R - 4);
#pragma unroll
            for (int j = 0; j < NL; ++j) {
                F4U v = rg[j];
                if (redge) v = shift_sel(v, sh, 4);
                if (kpart && k0 + kl + j >= kendi) v = F4U{0.f, 0.f, 0.f, 0.f};
                *(uint2*)(img + (kl + j) * PT + 4 * (t & 31)) = make_uint2(pack2_bf16(v.x, v.y), pack2_bf16(v.z, v.w));
            }
        }
    };
    // fragment of rows rb0 .. rb0 + 31, k-step kk (16 k): lane (r, h) gets row r, k = 16 kk + 8 h .. + 7
    auto frag = [&](auto kc, const bf16* img, int rb0, int kk) -> bf16x8 {
        if constexpr (decltype(kc)::value) return *(const bf16x8*)(img + (rb0 + r) * PR + kk * 16 + 8 * h);
        else {
            bf16x8 o;
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(img + (kk * 16 + 8 * h + 4 * jj + q) * PT + rb0 + 16 * colblk + 4 * p));
                o[4 * jj + 0] = v[0]; o[4 * jj + 1] = v[1]; o[4 * jj + 2] = v[2]; o[4 * jj + 3] = v[3];
            }
            return o;
        }
    };
    const std::integral_constant<bool, A_KC> akc;
    const std::integral_constant<bool, B_KC> bkc;
    const int sak_i = (int)sak, sbk_i = (int)sbk, m0i = (int)m0, n0i = (int)n0, Mi = (int)M, Ni = (int)N;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) { fetch(akc, A, rowA, sak_i, (int)kbeg + d * BK, ra[d]); fetch(bkc, Bm, rowB, sbk_i, (int)kbeg + d * BK, rb[d]); }
    // DEPTH stages per trip, no exits inside: a stage past kend stashes zeros (its loads were clamped) and multiplies them
    for (int k0 = (int)kbeg; k0 < kendi; k0 += DEPTH * BK) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int ks = k0 + d * BK;
            stash(akc, As, ra[d], m0i, Mi, ks);
            stash(bkc, Bs, rb[d], n0i, Ni, ks);
            __syncthreads();
            fetch(akc, A, rowA, sak_i, ks + DEPTH * BK, ra[d]);
            fetch(bkc, Bm, rowB, sbk_i, ks + DEPTH * BK, rb[d]);
#pragma unroll
            for (int kk = 0; kk < BK / 16; ++kk) {
                bf16x8 a[2], b[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) { a[i] = frag(akc, As, wm * 64 + i * 32, kk); b[i] = frag(bkc, Bs, wn * 64 + i * 32, kk); }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
            __syncthreads();
        }
    }
    // Epilogue through LDS, 16 tile rows at a time: the MFMA leaves a lane one column of 16 rows, so direct stores are 4-byte pieces in 128-byte runs
    // (6-11 us of a 28 us product by ablation: 13-17 MB of output per product); staged, every lane stores 16 bytes and a wave two 512-byte rows.
    // D: lane (r, h) holds column r, rows (e & 3) + 8 (e >> 2) + 4 h of each 32 x 32 tile.
    constexpr int SPITCH = 132;                              // floats: 528 B rows keep the 16-byte reads aligned and spread the 4-row step over the banks
    static_assert(16 * SPITCH * 4 <= IMG * 2, "the staging rows fit the A image");
    float* stage = (float*)As;
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
        const int pwm = pass >> 2, pi = (pass >> 1) & 1, half = pass & 1;
        if (wm == pwm) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int eh = 0; eh < 2; ++eh)
#pragma unroll
                    for (int el = 0; el < 4; ++el)
                        stage[(el + 4 * h + 8 * eh) * SPITCH + wn * 64 + j * 32 + r] = acc[pi][j][4 * (2 * half + eh) + el];
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int lr = (t >> 5) + 8 * u, c4 = t & 31;
            const int64_t gm = m0 + pwm * 64 + pi * 32 + 16 * half + lr, gn = n0 + 4 * c4;
            const float4 v4 = *(const float4*)(stage + lr * SPITCH + 4 * c4);
            float v[4] = {v4.x, v4.y, v4.z, v4.w};
            if (gm < M && gn < N) {
#ifndef CVAE_GEMM_T128_NOSTORE
                float* dst = slabs ? slabs + ((size_t)split * M + gm) * N + gn : C + gm * ldc + gn;
                if (!slabs) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e] + ((bias && gn + e < N) ? bias[gn + e] : 0.f), act);
                    if (em.src) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (gn + e < N) v[e] = epi_mul(v[e], em, gm, gn + e);
                    }
                }
                if (gn + 3 < N) *(F4U*)dst = F4U{v[0], v[1], v[2], v[3]};
                else {
#pragma unroll
                    for (int e = 0; e < 3; ++e) if (gn + e < N) dst[e] = v[e];
                }
#else
                if (v[0] == 123.456f) C[0] = 1.f;            // probe build: the epilogue's stores removed
#endif
            }
        }
        __syncthreads();
    }
}
// split-K of the 128-tile form: until ~CVAE_GEMM_T128_WGS workgroups exist, >= 4 stages per split
static int64_t gemm_t128_splits(int64_t M, int64_t N, int64_t K, int64_t* k_per_split_out) {
    const int64_t tiles = ((M + 127) / 128) * ((N + 127) / 128), stages = (K + CVAE_GEMM_T128_BK - 1) / CVAE_GEMM_T128_BK;
    int64_t splits = CVAE_GEMM_T128_WGS / tiles;
    if (splits > stages / 4) splits = stages / 4;
    if (splits < 1) splits = 1;
    const int64_t kps = ((stages + splits - 1) / splits) * CVAE_GEMM_T128_BK;
    if (k_per_split_out) *k_per_split_out = kps;
    return (K + kps - 1) / kps;
}
static bool gemm_t128_ok(int64_t M, int64_t N, int64_t sam, int64_t sak, int64_t sbk, int64_t sbn) {
    return CVAE_GEMM_T128 && M >= 128 && N >= 128 && (sak == 1 || sam == 1) && (sbk == 1 || sbn == 1);     // (the caller adds K >= 128: a one-stage product is all output)
}

// C[m][c] = act(sum_s slabs[s][m][c] + bias[c]), s in index order
__global__ void slab_sum_bias_act_kernel(const float* __restrict__ slabs, int splits, float* __restrict__ C, const float* __restrict__ bias, int64_t M, int64_t N,
                                         int64_t ldc, int act, EpiMul em = EpiMul{nullptr, 0, 0}) {
    const int64_t n = M * N;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = i / N, c = i - m * N;
        float v = 0.f;
        constexpr int U = 8;                                 // slab loads in flight (clamped index, predicated add: same order of the sum)
        for (int s0 = 0; s0 < splits; s0 += U) {
            float sv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) sv[u] = slabs[(size_t)min(s0 + u, splits - 1) * n + i];
#pragma unroll
            for (int u = 0; u < U; ++u) if (s0 + u < splits) v += sv[u];
        }
        C[m * ldc + c] = epi_mul(apply_act(v + (bias ? bias[c] : 0.f), act), em, m, c);
    }
}

// split-K factor of gemm_f32 for an [M][N] result over K: until ~2 workgroups per CU exist, keeping >= 4 K-steps (64 k) per split.  A small result
// over a long reduction (the weight gradients of the MLP heads over a 1024-row batch: 1 - 8 tiles, K = 1024) is otherwise ONE workgroup walking
// the whole reduction.
#ifndef CVAE_GEMM_SPLIT_MIN_KSTEPS
#define CVAE_GEMM_SPLIT_MIN_KSTEPS 8
#endif
static int64_t gemm_splits(int64_t M, int64_t N, int64_t K, int64_t* k_per_split_out) {
    const int64_t tm = (M + LT - 1) / LT, tn = (N + LT - 1) / LT;
    int64_t splits = 1;
    const int64_t ksteps = (K + LK - 1) / LK;
    if (tm * tn < 512 && ksteps >= CVAE_GEMM_SPLIT_MIN_KSTEPS) {
        splits = 512 / (tm * tn);
        if (splits > ksteps / 4) splits = ksteps / 4;
        if (splits < 1) splits = 1;
        if (splits > 1024) splits = 1024;
    }
    const int64_t k_per_split = ((ksteps + splits - 1) / splits) * LK;
    if (k_per_split_out) *k_per_split_out = k_per_split;
    return (K + k_per_split - 1) / k_per_split;
}
static int gemm_f32(const float* A, const float* Bm, float* C, const float* bias, int64_t M, int64_t N, int64_t K,
                    int64_t sam, int64_t sak, int64_t sbk, int64_t sbn, int64_t ldc, int act, float* ws, size_t ws_bytes, hipStream_t stream, bool bf16_math = false,
                    EpiMul em = EpiMul{nullptr, 0, 0}) {
    if (M < 0 || N <= 0 || K <= 0 || ldc < N) return CVAE_E_BADSHAPE;
    if (M == 0) return CVAE_OK;
    if (!A || !Bm || !C) return CVAE_E_NULLPTR;
    if (bf16_math && K >= 128 && gemm_t128_ok(M, N, sam, sak, sbk, sbn) && (M * sam + K * sak) < ((int64_t)1 << 31) && (N * sbn + K * sbk) < ((int64_t)1 << 31)) {      // (32-bit element offsets inside the kernel)
        int64_t kps;
        int64_t sp = gemm_t128_splits(M, N, K, &kps);
        if (sp > 1 && (!ws || ws_bytes < (size_t)sp * M * N * sizeof(float))) { sp = 1; kps = ((K + CVAE_GEMM_T128_BK - 1) / CVAE_GEMM_T128_BK) * CVAE_GEMM_T128_BK; }
        float* sl = sp > 1 ? ws : nullptr;
        const int64_t tn128 = (N + 127) / 128, tm128 = (M + 127) / 128;
        if (tn128 * tm128 * sp > 0x7fffffff) return CVAE_E_BADSHAPE;
        const dim3 grid((unsigned)(tn128 * tm128 * sp));
        const bool akc = sak == 1, bkc = sbk == 1;
        const int n_slow = N > M;
#define T128(AK, BKC) hipLaunchKernelGGL((gemm_bf16_t128_kernel<AK, BKC, CVAE_GEMM_T128_BK>), grid, dim3(256), 0, stream, A, Bm, C, bias, M, N, K, sam, sak, sbk, sbn, ldc, kps, act, sl, (int)tn128, (int)tm128, n_slow, sl ? EpiMul{nullptr, 0, 0} : em)
        if (akc) { if (bkc) T128(true, true); else T128(true, false); } else { if (bkc) T128(false, true); else T128(false, false); }
#undef T128
        CVAE_CHECK_LAUNCH();
        if (sl) {
            hipLaunchKernelGGL(slab_sum_bias_act_kernel, dim3(cvae_grid_1d(M * N, 256)), dim3(256), 0, stream, (const float*)sl, (int)sp, C, bias, M, N, ldc, act, em);
            CVAE_CHECK_LAUNCH();
        }
        return CVAE_OK;
    }
    const int64_t tm = (M + LT - 1) / LT, tn = (N + LT - 1) / LT;
    if (tm > 65535) return CVAE_E_BADSHAPE;
    int64_t k_per_split;
    int64_t splits = gemm_splits(M, N, K, &k_per_split);
    if (splits > 1 && (!ws || ws_bytes < (size_t)splits * M * N * sizeof(float))) { splits = 1; k_per_split = ((K + LK - 1) / LK) * LK; }
    float* slabs = splits > 1 ? ws : nullptr;
    dim3 grid((unsigned)tn, (unsigned)tm, (unsigned)splits);
    const EpiMul emk = slabs ? EpiMul{nullptr, 0, 0} : em;   // split-K: the slab sum applies it
    if (bf16_math) hipLaunchKernelGGL(gemm_bf16_kernel, grid, dim3(256), 0, stream, A, Bm, C, bias, M, N, K, sam, sak, sbk, sbn, ldc, k_per_split, act, slabs, emk);
    else hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(256), 0, stream, A, Bm, C, bias, M, N, K, sam, sak, sbk, sbn, ldc, k_per_split, act, slabs, emk);
    CVAE_CHECK_LAUNCH();
    if (slabs) {
        hipLaunchKernelGGL(slab_sum_bias_act_kernel, dim3(cvae_grid_1d(M * N, 256)), dim3(256), 0, stream, (const float*)slabs, (int)splits, C, bias, M, N, ldc, act, em);
        CVAE_CHECK_LAUNCH();
    }
    return CVAE_OK;
}

// ------------------------------------------------------------------------------------------------ skinny (M <= 16)
// At the model's batch sizes (4 per GPU) the three products of the 16415 x 512 encoder layer are pure weight streaming
// (33.6 MB each): the MFMA tile would be 94 % padding and its LDS round trip only adds latency.  These kernels read or
// write every weight element exactly once with fully coalesced dword rows and keep the M partial sums in registers.
#define SK_M 16

// y[m][n] (+)= sum_{k in chunk} x[m][k] W[n][k].  One wave per (row n, k-chunk); lanes stride k.
__global__ __launch_bounds__(256) void linear_fwd_skinny_kernel(const float* __restrict__ x, const float* __restrict__ W, const float* __restrict__ bias,
                                                                float* __restrict__ y, int M, int64_t K, int64_t N, int64_t ldx, int64_t ldy,
                                                                int64_t kchunk, int act, float* __restrict__ slabs) {
    const int lane = threadIdx.x & 63;
    const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const int64_t k0 = (int64_t)blockIdx.y * kchunk, k1 = min(K, k0 + kchunk);
    float acc[SK_M];
#pragma unroll
    for (int m = 0; m < SK_M; ++m) acc[m] = 0.f;
    const float* wr = W + n * K;
    int64_t k = k0 + lane;
    for (; k + 192 < k1; k += 256) {                         // 4 independent 256-byte rows of W (and of each x row) in flight
        const float w0 = wr[k], w1 = wr[k + 64], w2 = wr[k + 128], w3 = wr[k + 192];
#pragma unroll
        for (int m = 0; m < SK_M; ++m)
            if (m < M) {
                const float* xr = x + m * ldx + k;
                acc[m] += w0 * xr[0] + w1 * xr[64] + w2 * xr[128] + w3 * xr[192];
            }
    }
    for (; k < k1; k += 64) {
        const float w = wr[k];
#pragma unroll
        for (int m = 0; m < SK_M; ++m) if (m < M) acc[m] += w * x[m * ldx + k];
    }
#pragma unroll
    for (int m = 0; m < SK_M; ++m) {
        if (m >= M) break;
        const float s = wave_sum(acc[m]);
        if (lane == 0) {
            if (slabs) slabs[((size_t)blockIdx.y * M + m) * N + n] = s;                   // k-chunk partial: slab blockIdx.y
            else y[m * ldy + n] = apply_act(s + (bias ? bias[n] : 0.f), act);
        }
    }
}
// dx[m][k] = sum_{n in chunk} dy[m][n] W[n][k] (one n-chunk: straight into dx; several: slab blockIdx.y of [chunk][M][K], summed by
// slab_sum_bias_act_kernel).  One thread per k (coalesced W rows), blockIdx.y walks n-chunks.
__global__ __launch_bounds__(256) void linear_bwd_data_skinny_kernel(const float* __restrict__ dy, const float* __restrict__ W, float* __restrict__ dx,
                                                                     int M, int64_t K, int64_t N, int64_t ldy, int64_t ldx, int64_t nchunk,
                                                                     const float* __restrict__ yact, int act, float* __restrict__ slabs) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t n0 = (int64_t)blockIdx.y * nchunk, n1 = min(N, n0 + nchunk);
    __shared__ float gs[SK_M][64];                          // g = dy * act'(y) of one 64-wide n sub-chunk
    float acc[SK_M];
#pragma unroll
    for (int m = 0; m < SK_M; ++m) acc[m] = 0.f;
    for (int64_t nb = n0; nb < n1; nb += 64) {
        const int cnt = (int)min((int64_t)64, n1 - nb);
        __syncthreads();
        for (int i = threadIdx.x; i < M * 64; i += 256) {
            const int m = i >> 6, j = i & 63;
            float g = 0.f;
            if (j < cnt) { g = dy[m * ldy + nb + j]; if (yact) g *= act_grad_from_out(yact[m * ldy + nb + j], act); }
            gs[m][j] = g;
        }
        __syncthreads();
        if (k < K) {
#pragma unroll 8
            for (int j = 0; j < cnt; ++j) {
                const float w = W[(nb + j) * K + k];
#pragma unroll
                for (int m = 0; m < SK_M; ++m) if (m < M) acc[m] += w * gs[m][j];      // LDS broadcast
            }
        }
    }
    if (k < K) {
#pragma unroll
        for (int m = 0; m < SK_M; ++m)
            if (m < M) {
                if (slabs) slabs[((size_t)blockIdx.y * M + m) * K + k] = acc[m];
                else dx[m * ldx + k] = acc[m];
            }
    }
}
// dW[n][k] = sum_m dy[m][n] x[m][k]: one thread per element of the flattened [N*K] weight (coalesced stores).
__global__ __launch_bounds__(256) void linear_bwd_weight_skinny_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dW,
                                                                       float* __restrict__ db, int M, int64_t K, int64_t N, int64_t ldy, int64_t ldx,
                                                                       const float* __restrict__ yact, int act) {
    const int64_t total = N * K;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t n = i / K, k = i - n * K;
        float acc = 0.f, bsum = 0.f;
#pragma unroll
        for (int m = 0; m < SK_M; ++m)
            if (m < M) {
                const float g = dy[m * ldy + n] * (yact ? act_grad_from_out(yact[m * ldy + n], act) : 1.f);
                acc += g * x[m * ldx + k];
                bsum += g;
            }
        dW[i] = acc;
        if (db && k == 0) db[n] = bsum;                       // the bias gradient rides along (column sums of dy)
    }
}

// Row-per-block form for wide layers (K >= 256): g[m] = dy[m][n] * act'(y[m][n]) is block-uniform, each thread owns k.
__global__ __launch_bounds__(256) void linear_bwd_weight_rows_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dW,
                                                                     float* __restrict__ db, int M, int64_t K, int64_t N, int64_t ldy, int64_t ldx,
                                                                     const float* __restrict__ yact, int act) {
    const int64_t n = blockIdx.y;
    float g[SK_M], bsum = 0.f;
#pragma unroll
    for (int m = 0; m < SK_M; ++m) {
        g[m] = 0.f;
        if (m < M) { g[m] = dy[m * ldy + n]; if (yact) g[m] *= act_grad_from_out(yact[m * ldy + n], act); bsum += g[m]; }
    }
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < K; k += (int64_t)gridDim.x * 256) {
        float acc = 0.f;
#pragma unroll
        for (int m = 0; m < SK_M; ++m) if (m < M) acc += g[m] * x[m * ldx + k];
        dW[n * K + k] = acc;
    }
    if (db && blockIdx.x == 0 && threadIdx.x == 0) db[n] = bsum;
}

// k-chunks of the skinny forward: enough waves in flight to cover HBM latency
static int64_t skinny_fwd_chunks(int64_t K, int64_t N, int64_t* kchunk_out) {
    int64_t chunks = (4096 + N - 1) / N;
    if (chunks > K / 512) chunks = K / 512;
    if (chunks < 1) chunks = 1;
    const int64_t kchunk = ((K + chunks - 1) / chunks + 63) / 64 * 64;
    if (kchunk_out) *kchunk_out = kchunk;
    return (K + kchunk - 1) / kchunk;
}
// n-chunks of the skinny backward-data
static int64_t skinny_bwd_chunks(int64_t K, int64_t N, bool wide, int64_t* nchunk_out) {
    const int64_t kb = (K + 255) / 256;
    int64_t chunks;
    if (wide) {
        chunks = (2048 + kb - 1) / kb;                        // >= 8 waves per SIMD worth of blocks: the W stream is latency-bound per wave
        if (chunks > N / 16) chunks = N / 16;
    } else {
        chunks = N / 8; if (chunks > 256) chunks = 256;       // tiny W: spread the n reduction over many blocks
    }
    if (chunks < 1) chunks = 1;
    const int64_t nchunk = (N + chunks - 1) / chunks;
    if (nchunk_out) *nchunk_out = nchunk;
    return (N + nchunk - 1) / nchunk;
}
extern "C" size_t cvae_linear_workspace_bytes(int64_t M, int64_t K, int64_t N, int op) {
    if (M <= 0 || K <= 0 || N <= 0) return 0;
    int64_t splits = 1, elems = 0;
    // (the bf16-operand products may take the 128-tile kernel, whose split count differs: the larger of the two needs is reported)
    auto both = [](int64_t m, int64_t n, int64_t k) { const int64_t a = gemm_splits(m, n, k, nullptr), b = (m >= 128 && n >= 128 && k >= 128) ? gemm_t128_splits(m, n, k, nullptr) : 1; return a > b ? a : b; };
    if (op == 0) {                                           // forward: y [M][N] over K
        if (M <= SK_M && K >= 64) splits = skinny_fwd_chunks(K, N, nullptr); else splits = both(M, N, K);
        elems = M * N;
    } else if (op == 1) {                                    // backward-data: dx [M][K] over N
        if (M <= SK_M) splits = skinny_bwd_chunks(K, N, K >= 1024, nullptr); else splits = both(M, K, N);
        elems = M * K;
    } else if (op == 2) {                                    // backward-weight: dW [N][K] over M (+ the bias column sums)
        if (M > SK_M) {
            splits = both(N, K, M);
            const size_t cs = cvae_channel_sum_workspace_bytes(M, N, CVAE_F32);
            const size_t sl = splits > 1 ? (size_t)splits * N * K * sizeof(float) : 0;
            return sl > cs ? sl : cs;
        }
        return 0;
    }
    return splits > 1 ? (size_t)splits * elems * sizeof(float) : 0;
}
extern "C" int cvae_linear_fwd(const float* x, const float* W, const float* b, float* y, int64_t M, int64_t K, int64_t N,
                               int64_t x_stride, int64_t y_stride, int act, void* workspace, size_t workspace_bytes, void* stream) {
    if (x_stride < K) return CVAE_E_BADSHAPE;
    if (M > 0 && M <= SK_M && N > 0 && K >= 64 && y_stride >= N) {
        if (!x || !W || !y) return CVAE_E_NULLPTR;
        hipStream_t st = (hipStream_t)stream;
        int64_t kchunk;
        int64_t chunks = skinny_fwd_chunks(K, N, &kchunk);
        if (chunks > 1 && (!workspace || workspace_bytes < (size_t)chunks * M * N * sizeof(float))) { chunks = 1; kchunk = (K + 63) / 64 * 64; }
        float* slabs = chunks > 1 ? (float*)workspace : nullptr;
        dim3 grid((unsigned)((N + 3) / 4), (unsigned)chunks);
        hipLaunchKernelGGL(linear_fwd_skinny_kernel, grid, dim3(256), 0, st, x, W, b, y, (int)M, K, N, x_stride, y_stride, kchunk, act, slabs);
        CVAE_CHECK_LAUNCH();
        if (slabs) {
            hipLaunchKernelGGL(slab_sum_bias_act_kernel, dim3(cvae_grid_1d(M * N, 256)), dim3(256), 0, st, (const float*)slabs, (int)chunks, y, b, M, N, y_stride, act);
            CVAE_CHECK_LAUNCH();
        }
        return CVAE_OK;
    }
    return gemm_f32(x, W, y, b, M, N, K, x_stride, 1, 1, K, y_stride, act, (float*)workspace, workspace_bytes, (hipStream_t)stream);
}
extern "C" int cvae_linear_bwd_data(const float* dy, const float* W, float* dx, int64_t M, int64_t K, int64_t N,
                                    int64_t dy_stride, int64_t dx_stride, const float* y_act, int act, void* workspace, size_t workspace_bytes, void* stream) {
    if (dy_stride < N) return CVAE_E_BADSHAPE;
    if (act == CVAE_ACT_NONE) y_act = nullptr;
    if (y_act && !(M > 0 && M <= SK_M)) return CVAE_E_UNSUPPORTED;      // fused activation gradient: skinny path only
    // skinny path: wide layers (K >= 1024) always; narrow ones when the activation gradient is fused (materialise-free)
    if (M > 0 && M <= SK_M && N > 0 && dx_stride >= K && (K >= 1024 || y_act)) {
        if (!dy || !W || !dx) return CVAE_E_NULLPTR;
        hipStream_t st = (hipStream_t)stream;
        const int64_t kb = (K + 255) / 256;
        int64_t nchunk;
        int64_t chunks = skinny_bwd_chunks(K, N, K >= 1024, &nchunk);
        if (chunks > 1 && (!workspace || workspace_bytes < (size_t)chunks * M * K * sizeof(float))) { chunks = 1; nchunk = N; }
        float* slabs = chunks > 1 ? (float*)workspace : nullptr;
        hipLaunchKernelGGL(linear_bwd_data_skinny_kernel, dim3((unsigned)kb, (unsigned)chunks), dim3(256), 0, st, dy, W, dx, (int)M, K, N, dy_stride, dx_stride, nchunk, y_act, act, slabs);
        CVAE_CHECK_LAUNCH();
        if (slabs) {
            hipLaunchKernelGGL(slab_sum_bias_act_kernel, dim3(cvae_grid_1d(M * K, 256)), dim3(256), 0, st, (const float*)slabs, (int)chunks, dx, (const float*)nullptr, M, K, dx_stride, CVAE_ACT_NONE);
            CVAE_CHECK_LAUNCH();
        }
        return CVAE_OK;
    }
    return gemm_f32(dy, W, dx, nullptr, M, K, N, dy_stride, 1, K, 1, dx_stride, CVAE_ACT_NONE, (float*)workspace, workspace_bytes, (hipStream_t)stream);
}
extern "C" int cvae_linear_bwd_weight(const float* dy, const float* x, float* dW, float* db, int64_t M, int64_t K, int64_t N,
                                      int64_t dy_stride, int64_t x_stride, const float* y_act, int act, void* workspace, size_t workspace_bytes, void* stream) {
    if (dy_stride < N || x_stride < K || M <= 0) return CVAE_E_BADSHAPE;
    if (act == CVAE_ACT_NONE) y_act = nullptr;
    if (y_act && M > SK_M) return CVAE_E_UNSUPPORTED;
    int rc = CVAE_OK;
    if (M <= SK_M && N > 0 && K > 0) {
        if (!dy || !x || !dW) return CVAE_E_NULLPTR;
        if (K >= 256 && N <= 65535) {
            int64_t gx = (K + 1023) / 1024;                  // each thread ~4 columns
            hipLaunchKernelGGL(linear_bwd_weight_rows_kernel, dim3((unsigned)gx, (unsigned)N), dim3(256), 0, (hipStream_t)stream, dy, x, dW, db, (int)M, K, N, dy_stride, x_stride, y_act, act);
        } else {
            hipLaunchKernelGGL(linear_bwd_weight_skinny_kernel, dim3(cvae_grid_1d(N * K, 256, 16384)), dim3(256), 0, (hipStream_t)stream, dy, x, dW, db, (int)M, K, N, dy_stride, x_stride, y_act, act);
        }
        CVAE_CHECK_LAUNCH();
        return CVAE_OK;
    } else {
        rc = gemm_f32(dy, x, dW, nullptr, N, K, M, 1, dy_stride, x_stride, 1, K, CVAE_ACT_NONE, (float*)workspace, workspace_bytes, (hipStream_t)stream);
    }
    if (rc != CVAE_OK) return rc;
    if (db) {
        if (dy_stride != N) return CVAE_E_UNSUPPORTED;
        return cvae_channel_sum(dy, db, M, N, CVAE_F32, workspace, workspace_bytes, stream);    // stream order: the GEMM's slabs are consumed by then
    }
    return CVAE_OK;
}

// ---- the three linear products with bf16 MFMA operands (gemm_bf16_kernel): batches above the skinny range only; fp32 in, fp32 out, the bias
// gradient stays an fp32 column sum.  Workspace: cvae_linear_workspace_bytes (same tiles and split-K as the fp32 form). ----
extern "C" int cvae_linear_fwd_bf16(const float* x, const float* W, const float* b, float* y, int64_t M, int64_t K, int64_t N, int64_t x_stride, int64_t y_stride,
                                    int act, void* workspace, size_t workspace_bytes, void* stream) {
    if (x_stride < K || y_stride < N) return CVAE_E_BADSHAPE;
    if (M <= SK_M) return CVAE_E_UNSUPPORTED;
    return gemm_f32(x, W, y, b, M, N, K, x_stride, 1, 1, K, y_stride, act, (float*)workspace, workspace_bytes, (hipStream_t)stream, true);
}
extern "C" int cvae_linear_bwd_data_bf16(const float* dy, const float* W, float* dx, int64_t M, int64_t K, int64_t N, int64_t dy_stride, int64_t dx_stride,
                                         void* workspace, size_t workspace_bytes, void* stream) {
    if (dy_stride < N || dx_stride < K) return CVAE_E_BADSHAPE;
    if (M <= SK_M) return CVAE_E_UNSUPPORTED;
    return gemm_f32(dy, W, dx, nullptr, M, K, N, dy_stride, 1, K, 1, dx_stride, CVAE_ACT_NONE, (float*)workspace, workspace_bytes, (hipStream_t)stream, true);
}
// dx = (dy . W) * act'(x_in): the data gradient through this layer AND through the activation that produced this layer's input x_in [M][K] (act' from the
// activation's output, i.e. from x_in itself) in the GEMM's epilogue — the previous layer then skips its own activation-gradient pass.  Batches above the skinny range.
extern "C" int cvae_linear_bwd_data_inact(const float* dy, const float* W, float* dx, int64_t M, int64_t K, int64_t N, int64_t dy_stride, int64_t dx_stride,
                                          const float* x_in, int64_t x_stride, int in_act, int bf16_math, void* workspace, size_t workspace_bytes, void* stream) {
    if (dy_stride < N || dx_stride < K || x_stride < K) return CVAE_E_BADSHAPE;
    if (M <= SK_M) return CVAE_E_UNSUPPORTED;
    if (in_act != CVAE_ACT_NONE && !x_in) return CVAE_E_NULLPTR;
    const EpiMul em = (in_act == CVAE_ACT_NONE) ? EpiMul{nullptr, 0, 0} : EpiMul{x_in, x_stride, in_act};
    return gemm_f32(dy, W, dx, nullptr, M, K, N, dy_stride, 1, K, 1, dx_stride, CVAE_ACT_NONE, (float*)workspace, workspace_bytes, (hipStream_t)stream, bf16_math != 0, em);
}
extern "C" int cvae_linear_bwd_weight_bf16(const float* dy, const float* x, float* dW, float* db, int64_t M, int64_t K, int64_t N, int64_t dy_stride,
                                           int64_t x_stride, void* workspace, size_t workspace_bytes, void* stream) {
    if (dy_stride < N || x_stride < K) return CVAE_E_BADSHAPE;
    if (M <= SK_M) return CVAE_E_UNSUPPORTED;
    const int rc = gemm_f32(dy, x, dW, nullptr, N, K, M, 1, dy_stride, x_stride, 1, K, CVAE_ACT_NONE, (float*)workspace, workspace_bytes, (hipStream_t)stream, true);
    if (rc != CVAE_OK) return rc;
    if (db) {
        if (dy_stride != N) return CVAE_E_UNSUPPORTED;
        return cvae_channel_sum(dy, db, M, N, CVAE_F32, workspace, workspace_bytes, stream);
    }
    return CVAE_OK;
}
