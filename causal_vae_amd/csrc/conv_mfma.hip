// conv_mfma.hip — the k4/s2/p1 convolution family on the CDNA4 matrix cores, channels-last.
//
// Every strided conv of the model relates a SMALL tensor S [B, sd, sh, sw, Cs] and a LARGE tensor L [B, ld, lh, lw, Cl]
// (l = 2 s - 1 + k, k = 0..3 per strided dim; nd = 2 has one depth tap).  Three products cover all six ops
// (include/cvae_hip.h): down (S from L), up (L from S), wgrad (dW from S and L).
//
// down / up  — implicit GEMM, M = output positions, N = output channels, K = (tap, input channel):
//   * a workgroup owns a spatial tile of M (3D: 4x4x8 or 4x8x8 positions; 2D: 8x16 or 16x16) and loads the INPUT
//     HALO of that tile ONCE per 16-channel chunk into LDS with coalesced 16-byte rows; all 64 (down) / 8 (up, per
//     output parity) taps then read their A fragments straight out of that tile — the im2col matrix is never
//     materialised and each input byte crosses L2->LDS ~1.8x instead of 8x.
//   * `up` is the transposed conv in its parity form: the 2x2x2 output parity classes are 8 independent k2/s1
//     sub-convolutions (8 taps each), so no zero-insertion FLOPs are spent.
//   * B (weights, pre-packed [tap][K/16][N][16] by cvae_conv_pack_weight): bf16 tiles of 2 x 2 waves fetch their fragments per wave straight
//     from the packed global panels into a register ring ("BD"); the other forms stream them through a double-buffered LDS panel, 4 taps
//     per barrier.  A second `up` kernel (conv_up_full_kernel) stages the halo of ALL input channels once and walks the output parities
//     inside the workgroup; it serves the large grids of the decode sweep.
//   * bf16: v_mfma_f32_32x32x16_bf16 (fp32 accumulate);  fp32: v_mfma_f32_32x32x2_f32 (exact fp32 fmaf chain);  fp8: v_mfma_scale_f32_32x32x64_f8f6f4.
//   * epilogue fuses bias + ReLU/Sigmoid (forward use) or the ReLU mask of the saved activation (backward use).
// wgrad — M = Cs, N = Cl, K = positions.  Both operands are [position][channel] in memory, i.e. K-strided: bf16 uses
//   the gfx950 transposing LDS read (ds_read_b64_tr_b16) to build K-contiguous fragments; fp32's 32x32x2 MFMA takes
//   one element per lane and needs no transpose.  Every workgroup leaves ONE fp32 slab [kh][kw][64 cs][32 cl] of partial sums with
//   plain stores; wgrad_reduce_kernel adds the slabs in index order and writes the reference [Cs][Cl][taps] layout (no atomics).
#include "common.h"
#include <cstdlib>
#include <type_traits>

// Development aid (make EXTRA=-DCVAE_STAMP, tools/stamp_probe.py): thread 0 of every workgroup of conv_data_kernel records the
// shader clock at its phase boundaries into a device array that cvae_debug_stamps() copies out.  Not compiled by default.
#ifdef CVAE_STAMP
#define CVAE_STAMP_SLOTS 32
#define CVAE_STAMP_WGS 8192
__device__ unsigned long long g_stamp[(size_t)CVAE_STAMP_WGS * CVAE_STAMP_SLOTS];
#define STAMP(i)                                                                                                         \
    do {                                                                                                                 \
        if (t == 0 && stamp_wg < CVAE_STAMP_WGS) g_stamp[(size_t)stamp_wg * CVAE_STAMP_SLOTS + (i)] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define STAMP(i)
#endif

namespace {

#ifndef CVAE_BDIRECT
#define CVAE_BDIRECT 1                  // bf16 data kernels fetch their weight fragments per wave (BD, below): 0.867 -> 0.832 ms/step at 128^3 B=4
#endif
#ifndef CVAE_WG_TILES
#define CVAE_WG_TILES 16
#endif
#ifndef CVAE_WG_MIN_WG
#define CVAE_WG_MIN_WG 64
#endif
#ifndef CVAE_KSPLIT_WAVES
#define CVAE_KSPLIT_WAVES 1             // bf16 64-channel tiles: 2 x 2 waves = (K split) x (N sub-tile) instead of (M half) x (N sub-tile): 0.793 -> 0.785 ms/step
#endif
#ifndef CVAE_BD_GS
#define CVAE_BD_GS 8
#endif
#ifndef CVAE_BD_HPRE
#define CVAE_BD_HPRE 0
#endif
#ifndef CVAE_APIPE
#define CVAE_APIPE 3
#endif
#ifndef CVAE_UPFULL
#define CVAE_UPFULL 1
#endif
#ifndef CVAE_UPFULL_WIDE
#define CVAE_UPFULL_WIDE 0
#endif
#ifndef CVAE_UPFULL_MIN_GRID
#define CVAE_UPFULL_MIN_GRID 2048        // (tiles x channel blocks x batch) from which conv_up_full_kernel is used: many rounds of one workgroup per CU (the 240-row decode
#endif                                   // sweep: -12 % on its 64 -> 32 channel layer); at the training step's 512 it measured +-0 against conv_data_kernel<UP>
#ifndef CVAE_UPFULL_MIN_WG
#define CVAE_UPFULL_MIN_WG 512
#endif

struct ConvGeom {
    int B;
    int sd, sh, sw, Cs;
    int ld, lh, lw, Cl;
    int tiles_d, tiles_h, tiles_w;   // tiling of the M grid (S grid for down / wgrad, q grid for up)
};

// ---------------------------------------------------------------------------------------------- fragments
// f8x2: TWO consecutive fp8 (e4m3) channels as one 2-byte element.  With it the fp8 product is, byte for byte, the bf16 kernel on a tensor of
// Cin / 2 "elements": a 16-byte piece is 16 channels, a halo stage (two pieces per position) is 32 channels, the packed weight panels are
// [tap][Cin / 32][N][32 fp8] = [tap][Cin' / 16][N][16 elements] — the same LDS images, the same conflict-free layout, the same staging.  The one
// difference is the MFMA: v_mfma_scale_f32_32x32x64_f8f6f4 (block-scaled; unit block scales give plain per-tensor fp8) multiplies K = 64
// per instruction at twice the bf16 rate per FLOP, so TWO k-steps (two taps of a 32-channel stage) feed ONE instruction: registers 0-3 of
// both operands hold k-step s, registers 4-7 k-step s + 1.  The order of K inside an instruction does not matter to a dot product as
// long as both operands use the same one, which the symmetric A / B operand maps guarantee.
struct f8x2 { unsigned short v; };
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
template <typename T> struct IsF8 { static constexpr bool value = false; };
template <> struct IsF8<f8x2> { static constexpr bool value = true; };
template <typename T> struct Frag;
template <> struct Frag<bf16> { bf16x8 v; };
template <> struct Frag<float> { float v[8]; };
template <> struct Frag<f8x2> { i32x4 v; };

__device__ __forceinline__ void lds_load(Frag<bf16>& f, const char* p) { f.v = *(const bf16x8*)p; }
__device__ __forceinline__ void lds_load(Frag<float>& f, const char* p) {
    const float4 a = *(const float4*)p, b = *(const float4*)(p + 16);
    f.v[0] = a.x; f.v[1] = a.y; f.v[2] = a.z; f.v[3] = a.w; f.v[4] = b.x; f.v[5] = b.y; f.v[6] = b.z; f.v[7] = b.w;
}
__device__ __forceinline__ void lds_load(Frag<f8x2>& f, const char* p) { f.v = *(const i32x4*)p; }
#define CVAE_E8M0_ONE 0x7F7F7F7F        // block scale 2^0 in every byte: the scaled instruction then is a plain fp8 x fp8 product
// lane (r = lane & 31, h = lane >> 5) holds elements k = 8h .. 8h+7 of its row/column in both precisions
__device__ __forceinline__ void mma(f32x16& acc, const Frag<bf16>& a, const Frag<bf16>& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma(f32x16& acc, const Frag<float>& a, const Frag<float>& b) {
#pragma unroll
    for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[j], b.v[j], acc, 0, 0, 0);
}
// What ONE MFMA consumes: a k-step's fragment — or, for fp8, the fragments of TWO k-steps as one 8-register operand, put together where they are
// LOADED (two 16-byte reads into the halves of one register tuple).  Joining single-step fragments in front of each MFMA instead made the
// register allocator juggle 4-register values into 8-register tuples: 256 VGPRs + 59 spilled on the `down` form, a 20 k-cycle epilogue.
struct Frag2 { i32x8 v; };
template <typename T> struct StepFrag { using type = Frag<T>; static constexpr int PW = 1; };
template <> struct StepFrag<f8x2> { using type = Frag2; static constexpr int PW = 2; };
template <typename F>
__device__ __forceinline__ void load_step(F& f, const char* p0, const char*) { lds_load(f, p0); }
__device__ __forceinline__ void load_step(Frag2& f, const char* p0, const char* p1) {
    const i32x4 lo = *(const i32x4*)p0, hi = *(const i32x4*)p1;
    f.v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ void mma(f32x16& acc, const Frag2& a, const Frag2& b) {
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a.v, b.v, acc, 0, 0, 0, CVAE_E8M0_ONE, 0, CVAE_E8M0_ONE);
}

template <int ND, int BM> struct Tile;
template <> struct Tile<3, 128> { static constexpr int TD = 4, TH = 4, TW = 8; };
template <> struct Tile<3, 256> { static constexpr int TD = 4, TH = 8, TW = 8; };
template <> struct Tile<2, 128> { static constexpr int TD = 1, TH = 8, TW = 16; };
template <> struct Tile<2, 256> { static constexpr int TD = 1, TH = 16, TW = 16; };

// ---- bank-conflict-free halo layout (tools/lds_layout_search.py) ---------------------------------------------------
// ds_read_b128 is served in fixed 16-lane groups {0-3,12-15,20-27} / {4-11,16-19,28-31}; a group is conflict-free when
// its 16 fragments fall in 16 different 16-byte slots of the 256-byte bank row.  Two free choices make that true for
// every tap: (a) WHICH output position each MFMA row (lane) stands for inside its 32-position sub-tile — any bijection
// works as long as the epilogue uses the same one — and (b) the halo row pitch RS (in 16-byte slots), with the
// stride-2 (down) halo split into even-x / odd-x planes so that consecutive outputs read consecutive slots.
// Found by exhaustive search over bit permutations and pitches: 3D sub-tile 4(h) x 8(w), 2D sub-tile 2(h) x 16(w).
template <int ND> struct SubTile;
template <> struct SubTile<3> {
    static constexpr int SH = 4, SW = 8;
    __device__ static __forceinline__ int w_of(int r) { return ((r >> 2) & 1) | (((r >> 3) & 1) << 1) | ((r & 1) << 2); }
    __device__ static __forceinline__ int h_of(int r) { return ((r >> 4) & 1) | (((r >> 1) & 1) << 1); }
};
template <> struct SubTile<2> {
    static constexpr int SH = 2, SW = 16;
    __device__ static __forceinline__ int w_of(int r) { return ((r >> 2) & 1) | (((r >> 3) & 1) << 1) | ((r & 1) << 2) | (((r >> 1) & 1) << 3); }
    __device__ static __forceinline__ int h_of(int r) { return (r >> 4) & 1; }
};
template <int ND, bool UP> struct HaloPitch;                 // slots per halo row (per x-parity plane when !UP)
template <> struct HaloPitch<3, false> { static constexpr int RS = 10; };
template <> struct HaloPitch<3, true> { static constexpr int RS = 12; };
template <> struct HaloPitch<2, false> { static constexpr int RS = 18; };
template <> struct HaloPitch<2, true> { static constexpr int RS = 20; };

// One 8-element fragment piece (8 B fp8 / 16 B bf16 / 32 B fp32) in flight between a global load and its LDS store.  Staging
// is always written as "issue ALL loads of a tile, then store them": the loads overlap each other (and, for the weight
// panels, the MFMA work placed between the two halves) instead of paying one memory latency per piece.
template <typename T> struct PieceW { using type = uint4; static constexpr int N = (8 * sizeof(T)) / 16; };
template <> struct PieceW<fp8> { using type = uint2; static constexpr int N = 1; };       // 8 fp8 codes: an OUTPUT piece (8 channels of one position)
__device__ __forceinline__ uint4 piece_sel(bool ok, uint4 v) { return make_uint4(ok ? v.x : 0u, ok ? v.y : 0u, ok ? v.z : 0u, ok ? v.w : 0u); }
__device__ __forceinline__ uint2 piece_sel(bool ok, uint2 v) { return make_uint2(ok ? v.x : 0u, ok ? v.y : 0u); }
template <typename T> struct Piece { typename PieceW<T>::type v[PieceW<T>::N]; };
template <typename T>
__device__ __forceinline__ void piece_load(Piece<T>& p, const T* src, bool ok) {
    using W = typename PieceW<T>::type;
#pragma unroll
    for (int u = 0; u < PieceW<T>::N; ++u) {
        // `src` is always a readable address (callers pass the tensor base when !ok): load unconditionally and select, so the
        // compiler emits one straight-line global_load per piece instead of an exec-masked branch around each of them
        p.v[u] = piece_sel(ok, ((const W*)src)[u]);
    }
}
template <typename T>
__device__ __forceinline__ void piece_load_raw(Piece<T>& p, const T* src) {
    using W = typename PieceW<T>::type;
#pragma unroll
    for (int u = 0; u < PieceW<T>::N; ++u) p.v[u] = ((const W*)src)[u];
}
template <typename T>
__device__ __forceinline__ void piece_store_sel(const Piece<T>& p, bool ok, char* dst) {     // zero when !ok (decided at store time)
    using W = typename PieceW<T>::type;
#pragma unroll
    for (int u = 0; u < PieceW<T>::N; ++u) ((W*)dst)[u] = piece_sel(ok, p.v[u]);
}
template <typename T>
__device__ __forceinline__ void piece_store(const Piece<T>& p, char* dst) {
    using W = typename PieceW<T>::type;
#pragma unroll
    for (int u = 0; u < PieceW<T>::N; ++u) ((W*)dst)[u] = p.v[u];
}

// XCD-aware block -> tile map (cdna_hip_programming.md T1, bijective form): blocks b and b+8 share an XCD (and its L2),
// so each XCD gets a contiguous run of spatially adjacent tiles whose halos overlap.  Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int b, int n) {
    const int q = n >> 3, r = n & 7, x = b & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}

// ---------------------------------------------------------------------------------------------- down / up
// EPI: 0 = bias only, 1 = bias + ReLU, 2 = generic activation code
// KH: 16-channel k-steps per staged halo (1, or 2 for `up` in bf16: its halo box is small enough to hold 32 channels, which halves
// the stage / barrier count per MFMA and fetches 64 contiguous bytes per position instead of 32).
// TO: output (and mask) dtype, = T except on the fp8 inference path, where an fp8 x fp8 product leaves as fp8 (next fp8 layer) or bf16
// (the layer in front of the single-channel kernel): out = act(acc * acc_scale + bias) * out_scale, acc_scale = s_in * s_w the product of the
// operands' per-tensor scales, out_scale = 1 / s_out.
// BD ("B direct"): every wave fetches its weight fragments straight from the packed global panels into a two-group register ring (the panels are
// laid out so that one fragment of a wave is 1 KB contiguous) instead of all waves staging them through LDS: no weight traffic on the LDS
// pipe, which the activation fragments already load to ~2/3 of the MFMA time, and no barrier inside a channel chunk's tap loop.
// TS ("K split", BD only): TS = 2 wave groups of WM x WN waves each own every second k-step (odd / even taps for KH = 1, the two 16-channel halves
// of a stage for KH = 2) of the WHOLE tile and add their accumulators through LDS once, before the epilogue.  With WM = 1 no two waves fetch the
// same weight fragment, and a fetched fragment feeds MI = 4 MFMAs: half the bytes per MFMA on the vector-memory path the BD tap loop is bound by.
template <typename T, int ND, bool UP, int WM, int WN, int MI, int NI, int EPI, int KH = 1, typename TO = T, bool BD = false, int TS = 1, int XB = 1>
__global__ __launch_bounds__(WM * WN * TS * 64, BD ? 2 : 1) void conv_data_kernel(const T* __restrict__ in, const T* __restrict__ wp, const float* __restrict__ bias,
                                                                  const TO* __restrict__ mask, TO* __restrict__ out, ConvGeom g, int act,
                                                                  float* __restrict__ ws, int ksplit, float acc_scale, float out_scale, F8Side f8) {
    constexpr bool F8 = IsF8<T>::value;
    static_assert(!F8 || sizeof(TO) <= 2, "fp8 products leave as bf16 or as fp8 codes");
    static_assert(TS == 1 || (TS == 2 && BD && MI % 2 == 0), "the K split needs the per-wave weight fetch and an even number of M sub-tiles");
    constexpr int NT = WM * WN * TS * 64;
    constexpr int BM = WM * MI * 32, BN = WN * NI * 32;
    using TL = Tile<ND, BM>;
    constexpr int TD = TL::TD, TH = TL::TH, TW = TL::TW;
    constexpr int STR = UP ? 1 : 2;
    // UP: an output parity class reads q - 1 + pr + {0, 1} per dimension, so its halo box is (T + 1)^nd with the origin shifted by the
    // parity — 405 instead of the parity-independent 600 positions for 4 x 8 x 8 tiles (the halo loads are the largest single cost of
    // the `up` launches: 29 of 78 us on enc2's backward-data by ablation).
    constexpr int ID = (ND == 3) ? (UP ? TD + 1 : 2 * TD + 2) : 1;
    // XB == 2: a layer at most TW / 2 wide puts two samples side by side in x (the 4^3 decoder input would leave half of every MFMA row tile
    // empty): sample s owns tile columns [s TW / 2, (s + 1) TW / 2) and its own HWS halo columns, so a row of the halo is two sample rows with
    // the zero padding of each in place; a tile column w reads slot w + s (+ tap), which is one more term in the per-lane base.
    static_assert(XB == 1 || XB == 2, "one sample per tile, or two side by side in x");
    constexpr int IH = UP ? TH + 1 : 2 * TH + 2, IW = (UP ? TW + 1 : 2 * TW + 2) + (XB - 1) * (UP ? 1 : 2), HWS = IW / XB;
    constexpr int NPOS = ID * IH * IW;
    constexpr int FB = 8 * sizeof(T);                    // bytes of one fragment piece (8 channels)
    constexpr int NG = UP ? (ND == 3 ? 2 : 1) : (ND == 3 ? 16 : 4);   // tap groups of 4
    using ST = SubTile<ND>;
    constexpr int RS = HaloPitch<ND, UP>::RS, NROWS = ID * IH;
    constexpr int PLANE = (UP ? 1 : 2) * NROWS * RS;     // slots of one k-half plane (down: even-x rows then odd-x rows)
    static_assert(RS >= (UP ? IW : IW / 2), "halo pitch too small");
    constexpr int HALO_BYTES = 2 * KH * PLANE * FB;           // planes: (k-step, k-half)
    // slot of halo position (z, y, x) inside a k-half plane
    auto hslot = [](int z, int y, int x) -> int {
        return UP ? (z * IH + y) * RS + x : ((x & 1) * NROWS + z * IH + y) * RS + (x >> 1);
    };
    constexpr int BT_BYTES = BD ? 0 : 4 * KH * 2 * BN * FB;       // one B buffer: [4 taps][KH k-steps][2 halves][BN]
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* halo = smem;
    char* bt = smem + HALO_BYTES;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
#ifdef CVAE_STAMP
    const unsigned stamp_wg = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    if (t == 0 && stamp_wg < CVAE_STAMP_WGS) {
        g_stamp[(size_t)stamp_wg * CVAE_STAMP_SLOTS + 30] = wall_clock64();
        g_stamp[(size_t)stamp_wg * CVAE_STAMP_SLOTS + 29] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | __builtin_amdgcn_s_getreg(63492);
    }
#endif
    STAMP(0);
    const int ts = wave / (WM * WN), wv = wave % (WM * WN);      // K-split group (0 when TS == 1)
    const int wm = wv / WN, wn = wv % WN;
    const int r = lane & 31, h = lane >> 5;
    const int b = blockIdx.z * XB;
    const int Cin = (UP ? g.Cs : g.Cl) / (F8 ? 2 : 1), Cout = UP ? g.Cl : g.Cs;      // fp8: input channels counted in 2-channel elements
    const int nblocks = Cout / BN;
    constexpr int NPAR = UP ? (ND == 3 ? 8 : 4) : 1;
    // blockIdx.y = (ks * NPAR + par) * nblocks + nb; par: output parity class (UP only); ks: split-K slice of the channel chunks
    const int nb = blockIdx.y % nblocks, par = (blockIdx.y / nblocks) % NPAR, ks = blockIdx.y / (nblocks * NPAR);
    const int prd = (UP && ND == 3) ? ((par >> 2) & 1) : 0, prh = UP ? ((par >> 1) & 1) : 0, prw = UP ? (par & 1) : 0;
    const int n0 = nb * BN;
    int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tw_i = tile % g.tiles_w; tile /= g.tiles_w;
    const int th_i = tile % g.tiles_h; tile /= g.tiles_h;
    const int td_i = tile;
    const int o0d = td_i * TD, o0h = th_i * TH, o0w = tw_i * TW;       // tile origin in the M grid
    // input dims
    const int in_d = UP ? g.sd : g.ld, in_h = UP ? g.sh : g.lh, in_w = UP ? g.sw : g.lw;
    const int g0d = (ND == 3) ? (UP ? o0d - 1 + prd : 2 * o0d - 1) : 0;
    const int g0h = UP ? o0h - 1 + prh : 2 * o0h - 1, g0w = UP ? o0w - 1 + prw : 2 * o0w - 1;
    const int nchunks = Cin / (16 * KH);                    // stages; the packed weights are indexed in 16-channel chunks (nch16)
    const int nch16 = Cin / 16;

    // per-lane halo base position of each M sub-tile row
    // sub-tile ms of the workgroup tile covers d = ms / HB, h in [(ms % HB) * SH, +SH), all of w (SW == TW)
    static_assert(ST::SW == TW && TH % ST::SH == 0, "sub-tile must tile the workgroup tile");
    constexpr int HB = TH / ST::SH;
    int pbase[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int ms = wm * MI + mi;
        const int w = ST::w_of(r), hh = (ms % HB) * ST::SH + ST::h_of(r), d = ms / HB;
        pbase[mi] = (UP ? (d * IH + hh) * RS + w : ((2 * d) * IH + 2 * hh) * RS + w) + ((XB == 2 && w >= TW / 2) ? 1 : 0);
    }
    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

    auto tap_halo_off = [&](int grp, int j) -> int {
        if (!UP) {
            const int kd = (ND == 3) ? (grp >> 2) : 0, kh = (ND == 3) ? (grp & 3) : grp;
            return ((j & 1) * NROWS + kd * IH + kh) * RS + (j >> 1);      // kw = j: x-parity plane j & 1, slot shift j >> 1
        } else {
            const int a = (ND == 3) ? grp : 0, bb = j >> 1, c = j & 1;
            return (a * IH + bb) * RS + c;                                 // the halo origin already carries the parity
        }
    };
    auto tap_weight_idx = [&](int grp, int j) -> int {
        if (!UP) return grp * 4 + j;
        const int a = (ND == 3) ? grp : 0, bb = j >> 1, c = j & 1;
        const int kd = (ND == 3) ? (3 - prd - 2 * a) : 0, kh = 3 - prh - 2 * bb, kw = 3 - prw - 2 * c;
        return (kd * 4 + kh) * 4 + kw;
    };
    // ---- halo staging plan: the (position, half) pieces this thread moves are the same for every channel chunk ----
    constexpr int PPP = 2 * KH;                            // 8-channel pieces per position per stage
    constexpr int HN = (NPOS * PPP + NT - 1) / NT;
    int hoff[HN];                                          // element offset of the piece at chunk 0, or -1 (zero fill)
    int hdst[HN];                                          // its LDS byte offset, or -1 (past the halo box)
    {
        // piece t + i NT = (position t / PPP + i PSTEP, half t % PPP): the position's (x, y, z) is stepped, not divided (the divisions were
        // ~3 k cycles at the head of every workgroup, 10-15 % of the lifetime of the small layers' workgroups)
        static_assert(NT % PPP == 0, "a position's pieces stay in one pass");
        constexpr int PSTEP = NT / PPP, DX = PSTEP % IW, DY = (PSTEP / IW) % IH, DZ = PSTEP / (IW * IH);
        const int half = t % PPP, pos0 = t / PPP;
        int x = pos0 % IW, y = (pos0 / IW) % IH, z = pos0 / (IW * IH);
#pragma unroll
        for (int i = 0; i < HN; ++i) {
            const int sx = (XB == 2 && x >= HWS) ? 1 : 0;   // sample of this halo column (its columns restart at the sample's own left padding)
            const int gz = g0d + z, gy = g0h + y, gx = g0w + x - sx * HWS;
            const bool inbox = z < ID;
            const bool ok = inbox & (gz >= 0) & (gz < in_d) & (gy >= 0) & (gy < in_h) & (gx >= 0) & (gx < in_w) & (b + sx < g.B);
            hoff[i] = ok ? ((((sx * in_d + gz) * in_h + gy) * in_w + gx) * Cin + 8 * half) : -1;
            hdst[i] = inbox ? (half * PLANE + hslot(z, y, x)) * FB : -1;
            x += DX; if (x >= IW) { x -= IW; y += 1; }
            y += DY; if (y >= IH) { y -= IH; z += 1; }
            if (y >= IH) { y -= IH; z += 1; }
            z += DZ;
        }
    }
    const T* in_b = in + (size_t)b * in_d * in_h * in_w * Cin;
    constexpr int BP = (4 * KH * 2 * BN) / NT;             // weight pieces per thread per tap group
    static_assert((4 * KH * 2 * BN) % NT == 0, "weight panel must divide evenly over the workgroup");
    auto load_b = [&](Piece<T> (&pb)[BP], int chunk, int grp) {
#pragma unroll
        for (int i = 0; i < BP; ++i) {
            const int it = t + i * NT, half = it & 1, n = (it >> 1) % BN, kk = it / (2 * BN) % KH, j = it / (2 * BN * KH);
            const int wt = tap_weight_idx(grp, j);
            piece_load<T>(pb[i], wp + (((size_t)wt * nch16 + chunk * KH + kk) * Cout + n0 + n) * 16 + 8 * half, true);
        }
    };
    auto store_b = [&](const Piece<T> (&pb)[BP], int buf) {
#pragma unroll
        for (int i = 0; i < BP; ++i) {
            const int it = t + i * NT, half = it & 1, n = (it >> 1) % BN, kk = it / (2 * BN) % KH, j = it / (2 * BN * KH);
            piece_store<T>(pb[i], bt + buf * BT_BYTES + (((j * KH + kk) * 2 + half) * BN + n) * FB);
        }
    };

    // ---- BD: k-steps of a chunk in the order the LDS form walks them (tap group, tap, k-step), cut into NGRP groups of GS steps; group g + 1
    // (or the next chunk's group 0) is in flight while group g feeds the MFMAs ----
    // With TS = 2 a wave walks its OWN steps u = 0 .. STEPS - 1 <-> stage step 2 u + ts.  The ts part never enters the loops: for KH = 1 it is the
    // tap's low bit (down: the odd-x halo plane and the next weight tap; up: one slot to the right and weight tap kw - 2), for KH = 2 the second
    // 16-channel half of the stage — a constant offset of this wave's LDS and weight base addresses.
    constexpr int STEPS = NG * 4 * KH / TS, GS = STEPS >= 64 ? CVAE_BD_GS : (STEPS >= 16 ? (MI >= 4 ? 4 : 8) : STEPS / 2), NGRP = STEPS / GS;      // MI = 4: a step is 4 MFMAs, 4 steps are as long as 8
    static_assert(!BD || (NGRP % 2 == 0 && GS * NGRP == STEPS && (GS * TS) % KH == 0), "BD walks the groups in pairs");
    static_assert(TS == 1 || KH <= 2, "K split: one or two k-steps per stage");
    const long long w_tap = (long long)nch16 * Cout * 16;     // elements between two taps of the packed panels
    const long long w_ts = (TS == 1) ? 0 : (KH == 2 ? (long long)ts * Cout * 16 : (UP ? -2 * ts * w_tap : ts * w_tap));
    const int a_ts = (TS == 1) ? 0 : (KH == 2 ? ts * 2 * PLANE : (UP ? ts : ts * NROWS * RS));
    const T* wl = wp + ((size_t)(n0 + wn * NI * 32 + r)) * 16 + 8 * h + w_ts;
    const char* halo_a = halo + (size_t)a_ts * FB;
    constexpr int PW = StepFrag<T>::PW;                       // k-steps per MFMA (fp8: 2)
    using SF = typename StepFrag<T>::type;
    constexpr int GSX = GS / PW;                              // MFMAs (per accumulator) of a weight group
    static_assert(GS % PW == 0, "a weight group holds whole MFMA steps");
    SF qa[BD ? GSX : 1][NI], qb[BD ? GSX : 1][NI];
    auto load_q = [&](SF (&q)[BD ? GSX : 1][NI], int chunk, int gidx) {
#pragma unroll
        for (int ix = 0; ix < GSX; ++ix) {
            const T* wsrc[2];
#pragma unroll
            for (int u = 0; u < PW; ++u) {
                const int i = ix * PW + u;
                const int kk = (i * TS) % KH, tj = gidx * (GS * TS / KH) + (i * TS) / KH;
                const int wt = tap_weight_idx(tj >> 2, tj & 3);
                wsrc[u] = wl + (size_t)(wt * nch16 + chunk * KH + kk) * Cout * 16;
            }
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) load_step(q[ix][ni], (const char*)(wsrc[0] + (size_t)ni * 32 * 16), (const char*)(wsrc[PW - 1] + (size_t)ni * 32 * 16));
        }
    };
    STAMP(1);
    const int chunk_per = nchunks / ksplit;                 // host guarantees ksplit divides nchunks
    // what the epilogue needs from memory — this lane's bias values and ReLU-mask pieces — is requested now, not in the epilogue, where each was
    // an exposed global round trip at the end of every workgroup
    constexpr int MO = MI / TS;                               // M sub-tiles this wave finishes (TS = 2: the other half goes to its partner wave)
    const int mi0 = ts * MO;
    float bpre[NI][2][8];
    unsigned mbw[MO][NI];                                    // ReLU mask of this lane's channels, as bits of the position's 32-channel block dword
    const bool masked = !F8 && (mask || f8.mask_bits);
    const int out_d = UP ? g.ld : g.sd, out_h = UP ? g.lh : g.sh, out_w = UP ? g.lw : g.sw;
    if (ksplit == 1) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int c = n0 + (wn * NI + ni) * 32 + 16 * j + 8 * h;
#pragma unroll
                for (int q = 0; q < 8; ++q) bpre[ni][j][q] = bias ? bias[c + q] : 0.f;
            }
        if (masked) {                                        // fp8 products are forward products: no mask
#pragma unroll
            for (int mo = 0; mo < MO; ++mo) {
                const int mi = mi0 + mo, ms = wm * MI + mi;
                const int wt = ST::w_of(r), sx = (XB == 2 && wt >= TW / 2) ? 1 : 0, w = wt - sx * (TW / 2);
                const int hh = (ms % HB) * ST::SH + ST::h_of(r), d = ms / HB;
                int od, oh, ow;
                if (UP) { od = (ND == 3) ? 2 * (o0d + d) + prd : 0; oh = 2 * (o0h + hh) + prh; ow = 2 * (o0w + w) + prw; }
                else { od = o0d + d; oh = o0h + hh; ow = o0w + w; }
                const bool ok = od < out_d && oh < out_h && ow < out_w && b + sx < g.B;
                const size_t pidx = ok ? ((((size_t)(b + sx) * out_d + od) * out_h + oh) * out_w + ow) * Cout : 0;
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    if (f8.mask_bits) {                      // one dword per (position, 32-channel block) instead of two 16-byte pieces of the saved activation
                        mbw[mo][ni] = f8.mask_bits[(pidx + n0 + (wn * NI + ni) * 32) >> 5];
                    } else {                                 // the activation itself as the mask (callers without the bit form): turned into bits here
                        unsigned wbits = 0;
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            Piece<TO> mp;
                            piece_load_raw<TO>(mp, mask + pidx + n0 + (wn * NI + ni) * 32 + 16 * j + 8 * h);
                            const TO* mv = (const TO*)&mp;
#pragma unroll
                            for (int q = 0; q < 8; ++q) wbits |= (to_f32(mv[q]) > 0.f ? 1u : 0u) << (16 * j + 8 * h + q);
                        }
                        mbw[mo][ni] = wbits;
                    }
                }
            }
        }
    }
    // UP with 32-channel stages (long K loops on small grids): the NEXT stage's halo is requested right after this stage's LDS image is
    // complete and lands under the tap loop (-5 %).  Elsewhere the prefetch loses: DOWN stages 14 pieces per thread (registers), and the
    // Cin = 64 `up` launches fill the chip, where the co-resident workgroups already hide the stage (+4 % measured).
    constexpr bool HPRE = (UP && KH == 2) || (BD && CVAE_BD_HPRE);
    // Groups gp (weights in qa) and gp + 1 (qb) as ONE run of 2 GS k-steps; the group after them goes back into qa once qa is spent.  The activation
    // fragments are software-pipelined by hand: the ds_reads of step i + APD are issued in front of the MFMAs of step i (APD + 1 register slots), so an
    // MFMA never waits for a read issued right before it — left to itself the compiler emits read / s_waitcnt / MFMA per step and the loop runs at
    // LDS latency (~35 % of the MFMA rate by the stamp probes, one wave per SIMD).
    auto bd_pair = [&](int chunk, int gp) {
        // in MFMA steps (fp8: one step = two k-steps).  MI = 4: 4 reads per k-step, 2 k-steps ahead is as many in flight
        constexpr int NS = 2 * GSX, APW = (MI >= 4) ? 2 / PW : (CVAE_APIPE + PW - 1) / PW, APD = APW < NS ? APW : NS - 1;
        load_q(qb, chunk, gp + 1);
        SF ar[APD + 1][MI];
        auto lda = [&](int slot, int ix) {
            int off[2];
#pragma unroll
            for (int u = 0; u < PW; ++u) {
                const int i = ix * PW + u;
                const int gidx = gp + i / GS, ii = i % GS;
                const int kk = (ii * TS) % KH, tj = gidx * (GS * TS / KH) + (ii * TS) / KH;
                off[u] = (kk * 2 + h) * PLANE + tap_halo_off(tj >> 2, tj & 3);
            }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) load_step(ar[slot][mi], halo_a + (size_t)(off[0] + pbase[mi]) * FB, halo_a + (size_t)(off[PW - 1] + pbase[mi]) * FB);
        };
#pragma unroll
        for (int d = 0; d < APD; ++d) lda(d, d);
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            if (i + APD < NS) lda((i + APD) % (APD + 1), i + APD);
            if (i == GSX) {
                const bool wrap = gp + 2 >= NGRP;
                if (!wrap || chunk + 1 < (ks + 1) * chunk_per) load_q(qa, wrap ? chunk + 1 : chunk, wrap ? 0 : gp + 2);
            }
            __builtin_amdgcn_sched_barrier(0);             // keep the reads where they are written: the scheduler would sink them back to their uses
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) mma(acc[mi][ni], (i < GSX ? qa[i % GSX] : qb[i % GSX])[ni], ar[i % (APD + 1)][mi]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto bd_chunk = [&](int chunk) {
        if constexpr (NGRP > 2) {                           // 3D down: 64 taps; rolled, or the unrolled LDS reads spill
#pragma unroll 1
            for (int gp = 0; gp < NGRP; gp += 2) bd_pair(chunk, gp);
        } else {
            bd_pair(chunk, 0);
        }
    };
    if constexpr (BD) load_q(qa, ks * chunk_per, 0);
    Piece<T> hp[HN];
    if (HPRE) {
#pragma unroll
        for (int i = 0; i < HN; ++i) piece_load<T>(hp[i], in_b + (hoff[i] < 0 ? 0 : hoff[i]) + ks * chunk_per * (16 * KH), hoff[i] >= 0);
    }
    for (int chunk = ks * chunk_per; chunk < (ks + 1) * chunk_per; ++chunk) {
        {
            Piece<T> pb0[BP];
            if (!HPRE) {
#pragma unroll
                for (int i = 0; i < HN; ++i) piece_load<T>(hp[i], in_b + (hoff[i] < 0 ? 0 : hoff[i]) + chunk * (16 * KH), hoff[i] >= 0);
            }
            if constexpr (!BD) load_b(pb0, chunk, 0);
            __syncthreads();                               // previous chunk's readers are done with halo + B buffers
            if (chunk - ks * chunk_per < 8) STAMP(2 + 3 * (chunk - ks * chunk_per));
#pragma unroll
            for (int i = 0; i < HN; ++i)
                if (hdst[i] >= 0) piece_store<T>(hp[i], halo + hdst[i]);
            if constexpr (!BD) store_b(pb0, 0);
        }
        __syncthreads();
        if (chunk - ks * chunk_per < 8) STAMP(3 + 3 * (chunk - ks * chunk_per));
        if (HPRE && chunk + 1 < (ks + 1) * chunk_per) {
#pragma unroll
            for (int i = 0; i < HN; ++i) piece_load<T>(hp[i], in_b + (hoff[i] < 0 ? 0 : hoff[i]) + (chunk + 1) * (16 * KH), hoff[i] >= 0);
        }
        if constexpr (BD) {
            bd_chunk(chunk);
        } else {
            // Weight panels ride a 2-deep ring: the panel of group g+2 is loaded into registers at the start of group g and
            // stored to LDS at the end of group g+1, so every panel load has two groups of MFMA work to land (one group is
            // shorter than the L2 latency).  The loop is unrolled by two so the register sets pbA / pbB stay static.
            auto taps = [&](int grp, const char* btb) {
    #pragma unroll
                for (int sp = 0; sp < 4 * KH; sp += PW) {   // k-steps (tap j, k-step kk); fp8 feeds one K = 64 instruction per pair
                    SF a[MI], bf[NI];
                    int aoff[2], boff[2];
    #pragma unroll
                    for (int u = 0; u < PW; ++u) {
                        const int j = (sp + u) / KH, kk = (sp + u) % KH;
                        aoff[u] = (kk * 2 + h) * PLANE + tap_halo_off(grp, j);
                        boff[u] = ((j * KH + kk) * 2 + h) * BN;
                    }
    #pragma unroll
                    for (int mi = 0; mi < MI; ++mi) load_step(a[mi], halo + (size_t)(aoff[0] + pbase[mi]) * FB, halo + (size_t)(aoff[PW - 1] + pbase[mi]) * FB);
    #pragma unroll
                    for (int ni = 0; ni < NI; ++ni) load_step(bf[ni], btb + (boff[0] + (wn * NI + ni) * 32 + r) * FB, btb + (boff[PW - 1] + (wn * NI + ni) * 32 + r) * FB);
    #pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
    #pragma unroll
                        for (int ni = 0; ni < NI; ++ni) mma(acc[mi][ni], bf[ni], a[mi]);     // D = W^T x X^T: rows = channels (see epilogue)
                }
            };
            Piece<T> pbA[BP], pbB[BP];
            if (NG > 1) load_b(pbA, chunk, 1);
    #pragma unroll 1
            for (int grp = 0; grp < NG; grp += 2) {
                if (grp + 2 < NG) load_b(pbB, chunk, grp + 2);
                taps(grp, bt);
                if (grp + 1 < NG) store_b(pbA, 1);
                __syncthreads();
                if (grp + 1 < NG) {
                    if (grp + 3 < NG) load_b(pbA, chunk, grp + 3);
                    taps(grp + 1, bt + BT_BYTES);
                    if (grp + 2 < NG) store_b(pbB, 0);
                    __syncthreads();
                }
            }
        }
        if (chunk - ks * chunk_per < 8) STAMP(4 + 3 * (chunk - ks * chunk_per));
    }
    STAMP(26);

    // ---- epilogue.  The MFMAs ran with the WEIGHT fragment as the A operand, so D rows are output channels and D columns are
    // positions: lane (r, h) holds, for position r of each M sub-tile, channels (e & 3) + 8 (e >> 2) + 4 h of each 32-channel N
    // sub-tile.  Two v_permlane32_swap per register pair regroup them so that the lane owns channels 8h..8h+7 and 16+8h..23+8h:
    // two 8-channel pieces, each ONE 16-byte (bf16) store and ONE 16-byte mask load instead of eight 2-byte ones.
    // the exchange and the epilogue index the accumulators with ts: written once as a generic lambda and called with the wave's ts as a compile-time
    // constant (a run-time index would put the accumulator array in scratch memory)
    float amx = 0.f;                                          // fp8 side channel: largest |result| this lane stored
    const float accs = (F8 && f8.dscale) ? f8.dscale[0] : acc_scale, o8s = (F8 && f8.dscale) ? f8.dscale[1] : out_scale;
    auto finish = [&](auto TSV) {
        constexpr int tsc = decltype(TSV)::value, mi0c = tsc * MO;
        if constexpr (TS == 2) {
            // each wave hands the accumulators of the partner's M sub-tiles over through LDS (the halo is spent) and adds what the partner hands it
            __syncthreads();
            float4* xb = (float4*)smem;
            const int pw = wave ^ (WM * WN);
#pragma unroll
            for (int mo = 0; mo < MO; ++mo)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int e4 = 0; e4 < 4; ++e4) {
                        const f32x16& a = acc[(1 - tsc) * MO + mo][ni];
                        xb[(((size_t)wave * MO + mo) * NI + ni) * 4 * 64 + e4 * 64 + lane] = make_float4(a[4 * e4], a[4 * e4 + 1], a[4 * e4 + 2], a[4 * e4 + 3]);
                    }
            __syncthreads();
#pragma unroll
            for (int mo = 0; mo < MO; ++mo)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int e4 = 0; e4 < 4; ++e4) {
                        const float4 v = xb[(((size_t)pw * MO + mo) * NI + ni) * 4 * 64 + e4 * 64 + lane];
                        f32x16& a = acc[mi0c + mo][ni];
                        a[4 * e4] += v.x; a[4 * e4 + 1] += v.y; a[4 * e4 + 2] += v.z; a[4 * e4 + 3] += v.w;
                    }
        }
#pragma unroll
    for (int mo = 0; mo < MO; ++mo) {
        const int mi = mi0c + mo;
        const int ms = wm * MI + mi;                                            // same lane -> position map as pbase
        const int wt = ST::w_of(r), sx = (XB == 2 && wt >= TW / 2) ? 1 : 0, w = wt - sx * (TW / 2);
        const int hh = (ms % HB) * ST::SH + ST::h_of(r), d = ms / HB;
        int od, oh, ow;
        if (UP) { od = (ND == 3) ? 2 * (o0d + d) + prd : 0; oh = 2 * (o0h + hh) + prh; ow = 2 * (o0w + w) + prw; }
        else { od = o0d + d; oh = o0h + hh; ow = o0w + w; }
        const bool ok = od < out_d && oh < out_h && ow < out_w && b + sx < g.B;
        const size_t pidx = ((((size_t)(b + sx) * out_d + od) * out_h + oh) * out_w + ow) * Cout;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            float v[2][8];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const auto lo = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[mi][ni][i]), __float_as_uint(acc[mi][ni][4 + i]), false, false);
                const auto hi = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[mi][ni][8 + i]), __float_as_uint(acc[mi][ni][12 + i]), false, false);
                v[0][i] = __uint_as_float(lo[0]); v[0][4 + i] = __uint_as_float(lo[1]);
                v[1][i] = __uint_as_float(hi[0]); v[1][4 + i] = __uint_as_float(hi[1]);
            }
            if (ksplit > 1) {
                // split-K: this workgroup saw only its slice of the input channels; leave the raw fp32 partial sums in slab ks of
                // the workspace ([ks][B][positions][Cout]); conv_splitk_finish_kernel adds the slabs, bias, activation and mask.
                if (ok) {
                    float* wrow = ws + (size_t)ks * g.B * out_d * out_h * out_w * Cout + pidx;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int c = n0 + (wn * NI + ni) * 32 + 16 * j + 8 * h;
                        *(float4*)(wrow + c) = make_float4(v[j][0], v[j][1], v[j][2], v[j][3]);
                        *(float4*)(wrow + c + 4) = make_float4(v[j][4], v[j][5], v[j][6], v[j][7]);
                    }
                }
                continue;
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int c = n0 + (wn * NI + ni) * 32 + 16 * j + 8 * h;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    float x = (F8 ? v[j][q] * accs : v[j][q]) + bpre[ni][j][q];
                    if (EPI == 1) x = relu_f32(x);
                    else if (EPI == 2) x = apply_act(x, act);
                    v[j][q] = x;
                }
                if (!ok) continue;
                if (masked) {
                    const unsigned mb = mbw[mo][ni] >> (16 * j + 8 * h);
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        if (!((mb >> q) & 1u)) v[j][q] = 0.f;
                }
                if constexpr (F8) {
                    if (f8.amax) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) amx = fmaxf(amx, fabsf(v[j][q]));
                    }
                }
                if constexpr (sizeof(TO) == 2 && sizeof(T) == 2) {           // bf16: one v_cvt_pk_bf16_f32 per pair
                    *(uint4*)(out + pidx + c) = make_uint4(pack2_bf16(v[j][0], v[j][1]), pack2_bf16(v[j][2], v[j][3]), pack2_bf16(v[j][4], v[j][5]), pack2_bf16(v[j][6], v[j][7]));
                } else if constexpr (F8) {                                   // fp8 codes only (the inference chain between two fp8 layers): below, 16 bytes per lane
                } else {
                    Piece<TO> op;
                    TO* ov = (TO*)&op;
#pragma unroll
                    for (int q = 0; q < 8; ++q) ov[q] = from_f32<TO>(v[j][q]);
                    piece_store<TO>(op, (char*)(out + pidx + c));
                }
            }
            if (f8.bits_out) {                               // uniform: every lane takes part in the lane swap; lanes h = 0 store the block's dword
                const unsigned dw = mask_bytes_to_dword(mask_byte_of(v[0]), mask_byte_of(v[1]));
                if (ok && h == 0) f8.bits_out[(pidx + n0 + (wn * NI + ni) * 32) >> 5] = dw;
            }
            if constexpr (F8) {
                // the fp8 copy of this 32-channel block: 16 bytes per lane (all lanes take part in the lane swap; `ok` only guards the store)
                fp8* o8 = sizeof(TO) == 1 ? (fp8*)out : f8.out8;
                if (o8) {
                    const uint4 q16 = fp8_pair_to_16(pack8_fp8(v[0], o8s), pack8_fp8(v[1], o8s));
                    if (ok) *(uint4*)(o8 + pidx + n0 + (wn * NI + ni) * 32 + 16 * h) = q16;
                }
            }
        }
    }
    };
    if constexpr (TS == 2) {
        if (ts == 0) finish(std::integral_constant<int, 0>{}); else finish(std::integral_constant<int, 1>{});
    } else {
        finish(std::integral_constant<int, 0>{});
    }
    if constexpr (F8) {
        if (f8.amax && ksplit == 1) {                          // uniform over the workgroup
            __syncthreads();                                   // the LDS image (halo / accumulator exchange) is spent
            amax_publish_wg(f8.amax, amx, blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), (float*)smem);
        }
    }
#ifdef CVAE_STAMP
    __builtin_amdgcn_s_waitcnt(0);                         // vmcnt(0): the stores have left the wave
    STAMP(27);
    if (t == 0 && stamp_wg < CVAE_STAMP_WGS) g_stamp[(size_t)stamp_wg * CVAE_STAMP_SLOTS + 31] = wall_clock64();
#endif
}

// out[p][c] = act(acc_scale * sum_ks ws[ks][p][c] + bias[c]) (* mask > 0): 8 channels per thread, 16-byte bf16 stores.  acc_scale is 1 except behind an
// fp8 product (x * 1.0f is exact: the other dtypes' bits do not change), whose side channel (second fp8 output, amax) is served here too.
template <typename T, int EPI>
__global__ __launch_bounds__(256) void conv_splitk_finish_kernel(const float* __restrict__ ws, const float* __restrict__ bias, const T* __restrict__ mask,
                                                                  T* __restrict__ out, int64_t total, int Cout, int ksplit, int act, float acc_scale, F8Side f8) {
    const int64_t i8 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
    float amx = 0.f;
    if (i8 < total) {
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = 0.f;
        constexpr int U = 8;                                 // slab loads in flight (clamped index, predicated add: same order of the sum)
        for (int k0 = 0; k0 < ksplit; k0 += U) {
            float4 a[U], b[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float* p = ws + (size_t)min(k0 + u, ksplit - 1) * total + i8;
                a[u] = *(const float4*)p; b[u] = *(const float4*)(p + 4);
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (k0 + u < ksplit) { v[0] += a[u].x; v[1] += a[u].y; v[2] += a[u].z; v[3] += a[u].w; v[4] += b[u].x; v[5] += b[u].y; v[6] += b[u].z; v[7] += b[u].w; }
        }
        const float accs = f8.dscale ? f8.dscale[0] : acc_scale;
        const int c = (int)(i8 % Cout);
        Piece<T> mp, op;
        unsigned mb = 0xffu;
        if (f8.mask_bits) mb = ((const unsigned char*)f8.mask_bits)[i8 >> 3];
        else if (mask) {
            piece_load_raw<T>(mp, mask + i8);
            const T* mv = (const T*)&mp;
            mb = 0;
#pragma unroll
            for (int q = 0; q < 8; ++q) mb |= (to_f32(mv[q]) > 0.f ? 1u : 0u) << q;
        }
        T* ov = (T*)&op;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            float x = v[q] * accs + (bias ? bias[c + q] : 0.f);
            x = apply_act_t<EPI>(x, act);
            if (!((mb >> q) & 1u)) x = 0.f;
            v[q] = x;
            ov[q] = from_f32<T>(x);
            amx = fmaxf(amx, fabsf(x));
        }
        piece_store<T>(op, (char*)(out + i8));
        if (f8.bits_out) ((unsigned char*)f8.bits_out)[i8 >> 3] = (unsigned char)mask_byte_of(v);
        if (f8.out8) *(uint2*)(f8.out8 + i8) = pack8_fp8(v, f8.dscale ? f8.dscale[1] : 1.f);
    }
    __shared__ float red[4];
    if (f8.amax) amax_publish_wg(f8.amax, amx, blockIdx.x, red);
}

#ifndef CVAE_XPAIR
#define CVAE_XPAIR 1                    // two samples per tile (XB = 2) for layers at most half a tile wide
#endif
#ifndef CVAE_XPAIR_MIN_WGS
#define CVAE_XPAIR_MIN_WGS 2048
#endif
#ifndef CVAE_XPAIR_2D
#define CVAE_XPAIR_2D 1                 // the same for 2D layers, `down` and `up`
#endif
#ifndef CVAE_XPAIR_2D_MIN_WGS
#define CVAE_XPAIR_2D_MIN_WGS 512
#endif
// Kernel-form selection of one `up` launch.  The library picks by launch size (-1 / 0 = automatic); cvae_conv_up_variant and cvae_conv_fp8 let a caller
// force a form for ONE call — the tests run every narrow case through both forms that way.  No process-wide state.
struct UpVariant {
    int upfull = -1;            // conv_up_full_kernel: -1 by grid size (CVAE_UPFULL_MIN_GRID), 0 never, 1 whenever the shape fits it
    int xpair = -1;             // two samples per tile for layers at most half a tile wide: -1 by grid size (CVAE_XPAIR_MIN_WGS), 0 never, 1 always
    long long walk_units = 0;   // single-channel output layer: 0 = by launch size (CVAE_C1U_WALK_MIN_UNITS), > 0 = that many units
};

// Split-K factor for a launch of `nwg` workgroups over `nchunks` channel chunks: the layers with 8^3 / 4^3 grids fill a fraction
// of the 256 CUs with one long serial K loop each; slicing K puts ~2 workgroups on every CU.  Largest divisor of nchunks <= target.
// Only `down` splits: an `up` launch already has 8 (4) parity classes per tile and its output is 8x (4x) its input, so the fp32 slabs
// cost more than the shorter K loop saves (measured: enc4 backward-data 23 -> 34 us, dec2 forward 12 -> 22 us with split-K).
static int pick_ksplit(bool up, long long nwg, int nchunks) {
    // a launch that already has one workgroup per CU stays whole: enc3's forward (256 workgroups) took 39 us unsplit against 34 + 8 (finish) split
    if (up || nwg >= 256 || nwg < 1 || nchunks < 2) return 1;
    long long target = (512 + nwg - 1) / nwg;
    if (target > 16) target = 16;
    int best = 1;
    for (int d = 1; d <= nchunks && d <= target; ++d)
        if (nchunks % d == 0) best = d;
    return best;
}

template <typename T, int ND, bool UP, int WM, int WN, int MI, int NI, int EPI, int KH = 1, typename TO = T, bool BD = (CVAE_BDIRECT && sizeof(T) == 2 && WM <= 2), int TS = 1, int XB = 1>
int launch_data_epi(const void* in, const void* wp, const float* bias, const void* mask, void* out, ConvGeom g, int act, void* workspace,
                    size_t workspace_bytes, hipStream_t stream, float acc_scale = 1.f, float out_scale = 1.f, F8Side f8 = F8Side{nullptr, nullptr, nullptr},
                    UpVariant var = UpVariant{}) {
    constexpr int BM = WM * MI * 32, BN = WN * NI * 32;
    using TL = Tile<ND, BM>;
    constexpr int ID = (ND == 3) ? (UP ? TL::TD + 1 : 2 * TL::TD + 2) : 1;
    constexpr int IH = UP ? TL::TH + 1 : 2 * TL::TH + 2, IW = UP ? TL::TW + 1 : 2 * TL::TW + 2;
    constexpr int FB = 8 * sizeof(T);
    static_assert(IW >= 0, "");
    constexpr size_t LDS_MAIN = (size_t)2 * KH * (UP ? 1 : 2) * ID * IH * HaloPitch<ND, UP>::RS * FB + (BD ? 0 : (size_t)2 * 4 * KH * 2 * BN * FB);
    constexpr size_t LDS_X = TS == 2 ? (size_t)WM * WN * TS * (MI / 2) * NI * 16 * 64 * sizeof(float) : 0;      // the K split's accumulator exchange
    constexpr size_t LDS = LDS_MAIN > LDS_X ? LDS_MAIN : LDS_X;
    static_assert(LDS <= 160 * 1024, "LDS tile exceeds the 160 KiB of a CDNA4 CU");
    const int md = UP ? ((ND == 3) ? (g.ld + 1) / 2 : 1) : g.sd, mh = UP ? (g.lh + 1) / 2 : g.sh, mw = UP ? (g.lw + 1) / 2 : g.sw;
    if constexpr (CVAE_XPAIR && XB == 1 && sizeof(T) <= 2 && ((UP && ND == 3 && (TS == 2 || IsF8<T>::value)) || (ND == 2 && CVAE_XPAIR_2D))) {
        // a layer at most half a tile wide, on a launch that fills the chip several times over (the decode sweep's 4^3 -> 8^3 layer; round 3: the 7 x 7 grids of the
        // MNIST model at batch 1024, where one image fills 49 of a tile's 128 positions, in both directions): two samples per tile
        const long long wgs = (long long)((md + TL::TD - 1) / TL::TD) * ((mh + TL::TH - 1) / TL::TH) * ((UP ? g.Cl : g.Cs) / BN) * (UP ? (ND == 3 ? 8 : 4) : 1) * g.B;
        if (mw <= TL::TW / 2 && g.B >= 2 && (var.xpair == 1 || (var.xpair < 0 && wgs >= (ND == 3 ? CVAE_XPAIR_MIN_WGS : CVAE_XPAIR_2D_MIN_WGS))))
            return launch_data_epi<T, ND, UP, WM, WN, MI, NI, EPI, KH, TO, BD, TS, 2>(in, wp, bias, mask, out, g, act, workspace, workspace_bytes, stream, acc_scale, out_scale, f8, var);
    }
    auto kern = conv_data_kernel<T, ND, UP, WM, WN, MI, NI, EPI, KH, TO, BD, TS, XB>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS) != hipSuccess) return CVAE_E_LAUNCH;
        attr_set = true;
    }
    g.tiles_d = (md + TL::TD - 1) / TL::TD; g.tiles_h = (mh + TL::TH - 1) / TL::TH; g.tiles_w = (mw + TL::TW - 1) / TL::TW;
    const int Cout = UP ? g.Cl : g.Cs, Cin = (UP ? g.Cs : g.Cl) / (IsF8<T>::value ? 2 : 1);
    const int npar = UP ? ((ND == 3) ? 8 : 4) : 1;
    const long long tiles = (long long)g.tiles_d * g.tiles_h * g.tiles_w;
    long long gy = (long long)(Cout / BN) * npar;
    const int64_t total = (int64_t)g.B * (UP ? (int64_t)g.ld * g.lh * g.lw : (int64_t)g.sd * g.sh * g.sw) * Cout;
    int ksplit = pick_ksplit(UP, tiles * gy * g.B, Cin / (16 * KH));
    if (!workspace || workspace_bytes < (size_t)ksplit * total * sizeof(float)) ksplit = 1;     // no (or too small a) workspace: unsplit
    if (IsF8<T>::value && sizeof(TO) != 2) ksplit = 1;       // the finish pass writes bf16
    gy *= ksplit;
    if (gy > 65535 || g.B > 65535) return CVAE_E_BADSHAPE;
    dim3 grid((unsigned)tiles, (unsigned)gy, (unsigned)((g.B + XB - 1) / XB));
    hipLaunchKernelGGL(kern, grid, dim3(WM * WN * TS * 64), LDS, stream, (const T*)in, (const T*)wp, bias, (const TO*)mask, (TO*)out, g, act, (float*)workspace, ksplit,
                       acc_scale, out_scale, f8);
    CVAE_CHECK_LAUNCH();
    if constexpr (sizeof(TO) != 1) if (ksplit > 1) {
        hipLaunchKernelGGL((conv_splitk_finish_kernel<TO, EPI>), dim3((unsigned)((total / 8 + 255) / 256)), dim3(256), 0, stream, (const float*)workspace, bias,
                           (const TO*)mask, (TO*)out, total, Cout, ksplit, act, IsF8<T>::value ? acc_scale : 1.f, f8);
        CVAE_CHECK_LAUNCH();
    }
    return CVAE_OK;
}

template <typename T, int ND, bool UP, int WM, int WN, int MI, int NI, int TS = 1>
int launch_data(const void* in, const void* wp, const float* bias, const void* mask, void* out, ConvGeom g, int act, void* ws, size_t wsb, hipStream_t stream, UpVariant var = UpVariant{}, F8Side side = F8Side{nullptr, nullptr, nullptr}) {
    constexpr bool BDX = CVAE_BDIRECT && sizeof(T) == 2 && WM <= 2;
    // 32-channel stages pay where the K loop is long and the grid small (measured: Cin 256: -12 %, 128: -5 %, 64: +2 %)
    if (UP && sizeof(T) == 2 && g.Cs >= 128 && (g.Cs % 32) == 0) {
        constexpr int KH2 = (UP && sizeof(T) == 2) ? 2 : 1;
        if (act == CVAE_ACT_NONE) return launch_data_epi<T, ND, UP, WM, WN, MI, NI, 0, KH2, T, BDX, TS>(in, wp, bias, mask, out, g, act, ws, wsb, stream, 1.f, 1.f, side, var);
        if (act == CVAE_ACT_RELU) return launch_data_epi<T, ND, UP, WM, WN, MI, NI, 1, KH2, T, BDX, TS>(in, wp, bias, mask, out, g, act, ws, wsb, stream, 1.f, 1.f, side, var);
        return launch_data_epi<T, ND, UP, WM, WN, MI, NI, 2, KH2, T, BDX, TS>(in, wp, bias, mask, out, g, act, ws, wsb, stream, 1.f, 1.f, side, var);
    }
    if (act == CVAE_ACT_NONE) return launch_data_epi<T, ND, UP, WM, WN, MI, NI, 0, 1, T, BDX, TS>(in, wp, bias, mask, out, g, act, ws, wsb, stream, 1.f, 1.f, side, var);
    if (act == CVAE_ACT_RELU) return launch_data_epi<T, ND, UP, WM, WN, MI, NI, 1, 1, T, BDX, TS>(in, wp, bias, mask, out, g, act, ws, wsb, stream, 1.f, 1.f, side, var);
    return launch_data_epi<T, ND, UP, WM, WN, MI, NI, 2, 1, T, BDX, TS>(in, wp, bias, mask, out, g, act, ws, wsb, stream, 1.f, 1.f, side, var);
}

// Workspace the split-K path of launch_data would use for this geometry (0: the launch fills the chip without it).
template <int ND, bool UP, int BM, int BN>
size_t data_workspace_bytes(const ConvGeom& g) {
    using TL = Tile<ND, BM>;
    const int md = UP ? ((ND == 3) ? (g.ld + 1) / 2 : 1) : g.sd, mh = UP ? (g.lh + 1) / 2 : g.sh, mw = UP ? (g.lw + 1) / 2 : g.sw;
    const long long tiles = (long long)((md + TL::TD - 1) / TL::TD) * ((mh + TL::TH - 1) / TL::TH) * ((mw + TL::TW - 1) / TL::TW);
    const int Cout = UP ? g.Cl : g.Cs, Cin = UP ? g.Cs : g.Cl;
    const int npar = UP ? ((ND == 3) ? 8 : 4) : 1;
    const int ksplit = pick_ksplit(UP, tiles * (Cout / BN) * npar * g.B, Cin / 16);
    if (ksplit <= 1) return 0;
    return (size_t)ksplit * g.B * (UP ? (size_t)g.ld * g.lh * g.lw : (size_t)g.sd * g.sh * g.sw) * Cout * sizeof(float);
}


// ---------------------------------------------------------------------------------------------- up, whole-K halo ("up_full")
// The transposed conv has only 2^nd taps per output parity, so one 16- or 32-channel stage of conv_data_kernel<UP> feeds 16-32 MFMAs per wave
// between two barriers and a global->LDS round trip: the layers with few input channels spend more time staging than multiplying (stamp probes:
// ~3.7 k cycles per stage for 2.7 k cycles of taps on dec1's forward; enc2's backward-data ran at 480 TFLOP/s).  Here the (T + 2)^nd halo of the
// q tile is staged ONCE for ALL input channels (Cin = 16 KCH) and serves every parity class the workgroup owns (`ppw` of the 2^nd; the rest go to
// sibling workgroups when the grid is small).  The weights of a parity come through LDS in half panels (2^nd / 2 taps x all k-steps, two
// buffers, ONE barrier per half panel, fetched two half panels ahead: a per-wave fetch of the same fragments by every wave ran into the
// L2 -> L1 bandwidth, 22 B/clk/CU).  What the stamp probes and the ISA showed about the inner loop, with one wave per SIMD:
//   * the compiler sinks every ds_read to just in front of the MFMA that uses it (read / s_waitcnt / MFMA per step): the reads are issued
//     CVAE_APIPE steps ahead by hand and pinned with sched_barrier;
//   * with in-order issue every non-MFMA instruction between two MFMAs costs issue time the MFMA pipe idles for once there are more than
//     ~6 per MFMA: the LDS image is position-major ([halo slot][8-channel piece], pitch NPC + 1 pieces: the odd pitch keeps the reads
//     conflict-free exactly as the plane-major image did, and the pieces of one position, stored by consecutive lanes, hit different banks),
//     so that every read of a parity is `ds_read_b128 base, offset:imm` off two per-parity base registers — no address VALU in the loop;
//   * bias and the ReLU mask of a parity are fetched before its taps, not in the epilogue.
template <typename T, int ND, int WM, int WN, int MI, int NI, int KCH, int EPI, typename TO = T>
__global__ __launch_bounds__(WM * WN * 64) void conv_up_full_kernel(const T* __restrict__ in, const T* __restrict__ wp, const float* __restrict__ bias,
                                                                     const TO* __restrict__ mask, TO* __restrict__ out, ConvGeom g, int act, int ppw,
                                                                     float acc_scale, float out_scale) {
    constexpr int NT = WM * WN * 64;
    constexpr int BM = WM * MI * 32, BN = WN * NI * 32;
    using TL = Tile<ND, BM>;
    constexpr int TD = TL::TD, TH = TL::TH, TW = TL::TW;
    constexpr int ID = (ND == 3) ? TD + 2 : 1, IH = TH + 2, IW = TW + 2, NPOS = ID * IH * IW;
    constexpr int FB = 8 * sizeof(T);
    using ST = SubTile<ND>;
    constexpr int RS = HaloPitch<ND, true>::RS, NROWS = ID * IH, NSLOT = NROWS * RS;
    static_assert(RS >= IW, "halo pitch too small");
    constexpr int NPC = 2 * KCH, SPITCH = NPC + 1;           // 8-channel pieces per position; slot pitch in pieces (odd)
    constexpr int NTAP = (ND == 3) ? 8 : 4;                  // taps per parity class (= number of parity classes)
    constexpr int HTAP = NTAP / 2, PT = KCH * 2 * BN, HP_PIECES = HTAP * PT, HP_BYTES = HP_PIECES * FB, HPP = HP_PIECES / NT;
    static_assert(PT % NT == 0, "a tap's weight pieces must be a whole number of workgroup passes");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* halo = smem;
    char* wbuf = smem + (size_t)NSLOT * SPITCH * FB;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
#ifdef CVAE_STAMP
    const unsigned stamp_wg = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    if (t == 0 && stamp_wg < CVAE_STAMP_WGS) {
        g_stamp[(size_t)stamp_wg * CVAE_STAMP_SLOTS + 30] = wall_clock64();
        g_stamp[(size_t)stamp_wg * CVAE_STAMP_SLOTS + 29] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | __builtin_amdgcn_s_getreg(63492);
    }
#endif
    STAMP(0);
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 31, h = lane >> 5;
    const int b = blockIdx.z;
    const int Cin = g.Cs, Cout = g.Cl;
    const int nblocks = Cout / BN;
    const int nb = blockIdx.y % nblocks, pg = blockIdx.y / nblocks;
    const int n0 = nb * BN;
    int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tw_i = tile % g.tiles_w; tile /= g.tiles_w;
    const int th_i = tile % g.tiles_h; tile /= g.tiles_h;
    const int o0d = tile * TD, o0h = th_i * TH, o0w = tw_i * TW;
    const int g0d = (ND == 3) ? o0d - 1 : 0, g0h = o0h - 1, g0w = o0w - 1;
    static_assert(ST::SW == TW && TH % ST::SH == 0, "sub-tile must tile the workgroup tile");
    constexpr int HB = TH / ST::SH;
    int pbase[MI];                                           // halo slot of this lane's position in each M sub-tile, parity (0, 0, 0), tap (0, 0, 0)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int ms = wm * MI + mi;
        pbase[mi] = ((ms / HB) * IH + (ms % HB) * ST::SH + ST::h_of(r)) * RS + ST::w_of(r);
    }
    const int par0 = pg * ppw, nhp = 2 * ppw;
    auto tap_abc = [](int tap, int& a, int& bb, int& c) { a = (ND == 3) ? (tap >> 2) : 0; bb = (tap >> 1) & 1; c = tap & 1; };
    // ---- weight half panels: hpi -> (parity par0 + hpi / 2, taps (hpi & 1) * HTAP ..); LDS image [tap][k-step][half][n] ----
    const unsigned w_lane = (unsigned)((((t / (2 * BN)) * Cout + (t >> 1) % BN) * 16 + 8 * (t & 1)) * sizeof(T));     // the thread's piece inside a pass
    const int w_slot = (t / (2 * BN)) * (2 * BN) + (t & 1) * BN + (t >> 1) % BN;
    struct HalfPanel { Piece<T> p[HPP]; };                   // by value: as reference parameters of the lambdas the two register sets ended up in scratch
    auto load_hp = [&](int hpi) -> HalfPanel {
        HalfPanel wr;
        const int par = par0 + (hpi >> 1), th = hpi & 1;
        const int prd = (ND == 3) ? ((par >> 2) & 1) : 0, prh = (par >> 1) & 1, prw = par & 1;
#pragma unroll
        for (int i = 0; i < HPP; ++i) {
            const int tl = (i * NT) / PT, k0 = ((i * NT) % PT) / (2 * BN);       // pass i: tap tl of the half, k-steps k0 .. k0 + NT / (2 BN) - 1
            int a, bb, c;
            tap_abc(th * HTAP + tl, a, bb, c);
            const int kd = (ND == 3) ? (3 - prd - 2 * a) : 0, kh = 3 - prh - 2 * bb, kw = 3 - prw - 2 * c;
            const T* wu = wp + ((size_t)(((kd * 4 + kh) * 4 + kw) * KCH + k0) * Cout + n0) * 16;     // uniform
            piece_load_raw<T>(wr.p[i], (const T*)((const char*)wu + w_lane));
        }
        return wr;
    };
    auto store_hp = [&](const HalfPanel& wr, int buf) {
#pragma unroll
        for (int i = 0; i < HPP; ++i) piece_store<T>(wr.p[i], wbuf + (size_t)buf * HP_BYTES + (size_t)(i * NT + w_slot) * FB);
    };
    f32x16 acc[MI][NI];
    const char* abase[MI];                                   // per parity: LDS address of (lane position + parity shift, piece h)
    auto compute_hp = [&](const char* wb, int th) {          // HTAP * KCH k-steps, both operands from LDS, reads APD steps ahead of their MFMAs
        constexpr int NS = HTAP * KCH, APD = CVAE_APIPE < NS ? CVAE_APIPE : NS - 1;
        Frag<T> ar[APD + 1][MI], br[APD + 1][NI];
        const char* bb0 = wb + (size_t)(h * BN + wn * NI * 32 + r) * FB;
        auto ld = [&](int slot, int i) {
            const int tl = i / KCH, kk = i % KCH;
            int a, bb, c;
            tap_abc(tl, a, bb, c);                            // th * HTAP + tl: the half only moves the first tap coordinate (a in 3D, bb in 2D)
            const int tapc = (ND == 3) ? ((a + th) * IH + bb) * RS + c : (bb + th) * RS + c;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) lds_load(ar[slot][mi], abase[mi] + (size_t)(tapc * SPITCH + 2 * kk) * FB);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) lds_load(br[slot][ni], bb0 + (size_t)((tl * KCH + kk) * 2 * BN + ni * 32) * FB);
        };
#pragma unroll
        for (int d = 0; d < APD; ++d) ld(d, d);
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            if (i + APD < NS) ld((i + APD) % (APD + 1), i + APD);
            __builtin_amdgcn_sched_barrier(0);              // keep the reads where they are written: the scheduler would sink them back to their uses
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) mma(acc[mi][ni], br[i % (APD + 1)][ni], ar[i % (APD + 1)][mi]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    STAMP(1);
    HalfPanel wra = load_hp(0), wrb = wra;
    // ---- stage the whole halo (all channels), once: thread t moves piece t % NPC of positions t / NPC + i * (NT / NPC) ----
    {
        static_assert(NT % NPC == 0, "pieces of a position must stay in one pass");
        constexpr int PSTEP = NT / NPC, HN = (NPOS + PSTEP - 1) / PSTEP;
        constexpr int DX = PSTEP % IW, DY = (PSTEP / IW) % IH, DZ = PSTEP / (IW * IH);
        const T* in_b = in + (size_t)b * g.sd * g.sh * g.sw * Cin;
        const int pc = t % NPC, pos0 = t / NPC;
        int x = pos0 % IW, y = (pos0 / IW) % IH, z = pos0 / (IW * IH);
        Piece<T> hp[HN];
        int hs[HN];
#pragma unroll
        for (int i = 0; i < HN; ++i) {
            const int gz = g0d + z, gy = g0h + y, gx = g0w + x;
            const bool in_tile = z < ID;
            const bool ok = in_tile & (gz >= 0) & (gz < g.sd) & (gy >= 0) & (gy < g.sh) & (gx >= 0) & (gx < g.sw);
            piece_load<T>(hp[i], in_b + (ok ? (((size_t)gz * g.sh + gy) * g.sw + gx) * Cin + 8 * pc : 0), ok);
            hs[i] = in_tile ? ((z * IH + y) * RS + x) * SPITCH + pc : -1;
            x += DX; if (x >= IW) { x -= IW; y += 1; }
            y += DY; if (y >= IH) { y -= IH; z += 1; }
            z += DZ;
        }
        STAMP(2);
#pragma unroll
        for (int i = 0; i < HN; ++i)
            if (hs[i] >= 0) piece_store<T>(hp[i], halo + (size_t)hs[i] * FB);
    }
    store_hp(wra, 0);
    wra = load_hp(1);
    __syncthreads();
    STAMP(3);

    const int out_d = g.ld, out_h = g.lh, out_w = g.lw;
    Piece<TO> mpre[MI][NI][2];
    float bpre[NI][2][8];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = n0 + (wn * NI + ni) * 32 + 16 * j + 8 * h;
#pragma unroll
            for (int q = 0; q < 8; ++q) bpre[ni][j][q] = bias ? bias[c + q] : 0.f;
        }
    auto parity_begin = [&](int par) {                       // accumulators, LDS base of the parity's reads, its mask pieces
        const int prd = (ND == 3) ? ((par >> 2) & 1) : 0, prh = (par >> 1) & 1, prw = par & 1;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            abase[mi] = halo + (size_t)((pbase[mi] + (prd * IH + prh) * RS + prw) * SPITCH + h) * FB;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;
        }
        if (!mask) return;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int ms = wm * MI + mi;
            const int w = ST::w_of(r), hh = (ms % HB) * ST::SH + ST::h_of(r), d = ms / HB;
            const int od = (ND == 3) ? 2 * (o0d + d) + prd : 0, oh = 2 * (o0h + hh) + prh, ow = 2 * (o0w + w) + prw;
            const bool ok = od < out_d && oh < out_h && ow < out_w;
            const size_t pidx = ok ? ((((size_t)b * out_d + od) * out_h + oh) * out_w + ow) * Cout : 0;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int j = 0; j < 2; ++j) piece_load_raw<TO>(mpre[mi][ni][j], mask + pidx + n0 + (wn * NI + ni) * 32 + 16 * j + 8 * h);
        }
    };
    auto epilogue = [&](int par) {                           // conv_data_kernel's: permlane32_swap regroup, 16-byte stores
        const int prd = (ND == 3) ? ((par >> 2) & 1) : 0, prh = (par >> 1) & 1, prw = par & 1;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int ms = wm * MI + mi;
            const int w = ST::w_of(r), hh = (ms % HB) * ST::SH + ST::h_of(r), d = ms / HB;
            const int od = (ND == 3) ? 2 * (o0d + d) + prd : 0, oh = 2 * (o0h + hh) + prh, ow = 2 * (o0w + w) + prw;
            const bool ok = od < out_d && oh < out_h && ow < out_w;
            const size_t pidx = ((((size_t)b * out_d + od) * out_h + oh) * out_w + ow) * Cout;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                float v[2][8];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const auto lo = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[mi][ni][i]), __float_as_uint(acc[mi][ni][4 + i]), false, false);
                    const auto hi = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[mi][ni][8 + i]), __float_as_uint(acc[mi][ni][12 + i]), false, false);
                    v[0][i] = __uint_as_float(lo[0]); v[0][4 + i] = __uint_as_float(lo[1]);
                    v[1][i] = __uint_as_float(hi[0]); v[1][4 + i] = __uint_as_float(hi[1]);
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int c = n0 + (wn * NI + ni) * 32 + 16 * j + 8 * h;
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        float x = (sizeof(T) == 1 ? v[j][q] * acc_scale : v[j][q]) + bpre[ni][j][q];
                        if (EPI == 1) x = relu_f32(x);
                        else if (EPI == 2) x = apply_act(x, act);
                        v[j][q] = x;
                    }
                    if (!ok) continue;
                    if (mask) {
                        const TO* mv = (const TO*)&mpre[mi][ni][j];
#pragma unroll
                        for (int q = 0; q < 8; ++q)
                            if (!(to_f32(mv[q]) > 0.f)) v[j][q] = 0.f;
                    }
                    Piece<TO> op;
                    TO* ov = (TO*)&op;
#pragma unroll
                    for (int q = 0; q < 8; ++q) ov[q] = from_f32<TO>(sizeof(T) == 1 ? v[j][q] * out_scale : v[j][q]);
                    piece_store<TO>(op, (char*)(out + pidx + c));
                }
            }
        }
    };
    // half panels in pairs = one parity per iteration; panel hpi + 1 sits in wra, hpi + 2 is requested into wrb at the top
    for (int hpi = 0; hpi < nhp; hpi += 2) {
        const int par = par0 + (hpi >> 1);
        if (hpi + 2 < nhp) wrb = load_hp(hpi + 2);
        parity_begin(par);
        compute_hp(wbuf, 0);
        store_hp(wra, 1);
        __syncthreads();
        if (hpi + 3 < nhp) wra = load_hp(hpi + 3);
        compute_hp(wbuf + HP_BYTES, 1);
        if (hpi + 2 < nhp) store_hp(wrb, 0);
        STAMP(4 + hpi);
        epilogue(par);
        STAMP(5 + hpi);
        __syncthreads();
    }
#ifdef CVAE_STAMP
    __builtin_amdgcn_s_waitcnt(0);
    STAMP(27);
    if (t == 0 && stamp_wg < CVAE_STAMP_WGS) g_stamp[(size_t)stamp_wg * CVAE_STAMP_SLOTS + 31] = wall_clock64();
#endif
}

template <typename T, int ND, int WM, int WN, int MI, int NI, int KCH> constexpr size_t up_full_lds_bytes() {
    using TL = Tile<ND, WM * MI * 32>;
    constexpr int ID = (ND == 3) ? TL::TD + 2 : 1, IH = TL::TH + 2;
    return ((size_t)(ID * IH * HaloPitch<ND, true>::RS) * (2 * KCH + 1) + (size_t)2 * (((ND == 3) ? 8 : 4) / 2) * KCH * 2 * (WN * NI * 32)) * 8 * sizeof(T);
}

// Launch of conv_up_full_kernel; CVAE_E_UNSUPPORTED when this (tile, channel count) pair does not fit the LDS (the caller falls back to launch_data<UP>).
template <typename T, int ND, int WM, int WN, int MI, int NI, int KCH, typename TO = T>
int launch_up_full(const void* in, const void* wp, const float* bias, const void* mask, void* out, ConvGeom g, int act, hipStream_t stream,
                   UpVariant var, float acc_scale = 1.f, float out_scale = 1.f) {
    constexpr size_t LDS = up_full_lds_bytes<T, ND, WM, WN, MI, NI, KCH>();
    if constexpr (LDS > 160 * 1024 || (KCH * 2 * WN * NI * 32) % (WM * WN * 64) != 0) {
        return CVAE_E_UNSUPPORTED;
    } else {
        constexpr int BM = WM * MI * 32, BN = WN * NI * 32;
        using TL = Tile<ND, BM>;
        const int md = (ND == 3) ? (g.ld + 1) / 2 : 1, mh = (g.lh + 1) / 2, mw = (g.lw + 1) / 2;
        g.tiles_d = (md + TL::TD - 1) / TL::TD; g.tiles_h = (mh + TL::TH - 1) / TL::TH; g.tiles_w = (mw + TL::TW - 1) / TL::TW;
        const long long tiles = (long long)g.tiles_d * g.tiles_h * g.tiles_w;
        const int npar = (ND == 3) ? 8 : 4, nblocks = g.Cl / BN;
        if (var.upfull == 0 || (var.upfull < 0 && tiles * nblocks * g.B < CVAE_UPFULL_MIN_GRID)) return CVAE_E_UNSUPPORTED;
        // parity classes per workgroup: all of them (one halo stage per tile) once the grid has ~2 workgroups per CU without splitting them
        int psplit = 1;
        while (psplit < npar && tiles * nblocks * g.B * psplit < CVAE_UPFULL_MIN_WG) psplit *= 2;
        const long long gy = (long long)nblocks * psplit;
        if (gy > 65535 || g.B > 65535) return CVAE_E_BADSHAPE;
        const int e = (act == CVAE_ACT_NONE) ? 0 : (act == CVAE_ACT_RELU ? 1 : 2);
        dim3 grid((unsigned)tiles, (unsigned)gy, (unsigned)g.B), block(WM * WN * 64);
#define UPFULL_LAUNCH(EPI)                                                                                                                          \
        {                                                                                                                                           \
            auto kern = conv_up_full_kernel<T, ND, WM, WN, MI, NI, KCH, EPI, TO>;                                                                   \
            static bool attr_set = false;                                                                                                           \
            if (!attr_set) {                                                                                                                        \
                if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS) != hipSuccess) return CVAE_E_LAUNCH; \
                attr_set = true;                                                                                                                    \
            }                                                                                                                                       \
            hipLaunchKernelGGL(kern, grid, block, LDS, stream, (const T*)in, (const T*)wp, bias, (const TO*)mask, (TO*)out, g, act, npar / psplit,   \
                               acc_scale, out_scale);                                                                                               \
        }
        if (e == 0) UPFULL_LAUNCH(0) else if (e == 1) UPFULL_LAUNCH(1) else UPFULL_LAUNCH(2)
#undef UPFULL_LAUNCH
        CVAE_CHECK_LAUNCH();
        return CVAE_OK;
    }
}

// `up` through the whole-K kernel when the input channel count is one it is built for (64 / 128 / 256 where the halo fits): CVAE_E_UNSUPPORTED otherwise.
template <typename T, int ND, int WM, int WN, int MI, int NI, typename TO = T>
int try_up_full(const void* in, const void* wp, const float* bias, const void* mask, void* out, const ConvGeom& g, int act, hipStream_t stream,
                UpVariant var, float acc_scale = 1.f, float out_scale = 1.f) {
    if constexpr (!CVAE_UPFULL || (WN * NI > 1 && !CVAE_UPFULL_WIDE)) {      // the 64-channel-tile form measured slower than conv_data_kernel<UP> with BD: not instantiated
        return CVAE_E_UNSUPPORTED;
    } else {
        if (g.Cs == 64) return launch_up_full<T, ND, WM, WN, MI, NI, 4, TO>(in, wp, bias, mask, out, g, act, stream, var, acc_scale, out_scale);
        if (g.Cs == 128) return launch_up_full<T, ND, WM, WN, MI, NI, 8, TO>(in, wp, bias, mask, out, g, act, stream, var, acc_scale, out_scale);
        if (g.Cs == 256) return launch_up_full<T, ND, WM, WN, MI, NI, 16, TO>(in, wp, bias, mask, out, g, act, stream, var, acc_scale, out_scale);
    }
    return CVAE_E_UNSUPPORTED;
}

// ---------------------------------------------------------------------------------------------- weight packing
// CH: input channels per panel chunk — 16 (bf16 / fp32: one MFMA k-step) or 32 (fp8: the f8x2 element is two channels, so a chunk of 16
// elements is 32 channels; see Frag<f8x2>)
template <typename T, int CH = 16>
__global__ void pack_weight_kernel(const float* __restrict__ w, T* __restrict__ out, int Cs, int Cl, int taps, int for_up, float mul = 1.f) {
    const int64_t n = (int64_t)Cs * Cl * taps;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        // i enumerates the OUTPUT so writes are contiguous
        const int e = (int)(i % CH);
        int64_t rr = i / CH;
        int cs, cl, tap;
        if (!for_up) { cs = (int)(rr % Cs); rr /= Cs; const int ch = (int)(rr % (Cl / CH)); tap = (int)(rr / (Cl / CH)); cl = ch * CH + e; }
        else { cl = (int)(rr % Cl); rr /= Cl; const int ch = (int)(rr % (Cs / CH)); tap = (int)(rr / (Cs / CH)); cs = ch * CH + e; }
        out[i] = from_f32<T>(sizeof(T) == 1 ? w[((int64_t)cs * Cl + cl) * taps + tap] * mul : w[((int64_t)cs * Cl + cl) * taps + tap]);
    }
}

// fp8 panels of several layers in one launch, scales read from DEVICE memory (a captured training step re-reads them on every replay), and the
// largest |w| of every layer recorded for the next step's scale (delayed scaling: cvae_fp8_scale_update).
#define F8PACK_MAX 12
struct F8PackTable {
    const float* w[F8PACK_MAX];
    fp8* out[F8PACK_MAX];
    const float* inv_scale[F8PACK_MAX];      // device: 1 / s_w
    unsigned* amax[F8PACK_MAX];              // device: CVAE_AMAX_SLOTS words, or null
    int Cs[F8PACK_MAX], Cl[F8PACK_MAX], for_up[F8PACK_MAX], blk_start[F8PACK_MAX + 1];
    int count, taps;
};
__global__ __launch_bounds__(256) void pack_weight_fp8_multi_kernel(F8PackTable tb) {
    constexpr int CH = 32;
    int ti = 0;
    while (ti + 1 < tb.count && (int)blockIdx.x >= tb.blk_start[ti + 1]) ++ti;
    const int Cs = tb.Cs[ti], Cl = tb.Cl[ti], taps = tb.taps, for_up = tb.for_up[ti];
    const float* w = tb.w[ti];
    fp8* out = tb.out[ti];
    const float mul = tb.inv_scale[ti][0];
    const int64_t n = (int64_t)Cs * Cl * taps;
    const int nb = tb.blk_start[ti + 1] - tb.blk_start[ti];
    float amx = 0.f;
    for (int64_t i = (((int64_t)blockIdx.x - tb.blk_start[ti]) * 256 + threadIdx.x) * 4; i < n; i += (int64_t)nb * 1024) {      // 4 consecutive codes per thread: one dword store
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t iu = i + u;
            const int e = (int)(iu % CH);
            int64_t rr = iu / CH;
            int cs, cl, tap;
            if (!for_up) { cs = (int)(rr % Cs); rr /= Cs; const int ch = (int)(rr % (Cl / CH)); tap = (int)(rr / (Cl / CH)); cl = ch * CH + e; }
            else { cl = (int)(rr % Cl); rr /= Cl; const int ch = (int)(rr % (Cs / CH)); tap = (int)(rr / (Cs / CH)); cs = ch * CH + e; }
            v[u] = w[((int64_t)cs * Cl + cl) * taps + tap];
            amx = fmaxf(amx, fabsf(v[u]));
            v[u] = __builtin_amdgcn_fmed3f(v[u] * mul, -CVAE_FP8_MAX, CVAE_FP8_MAX);
        }
        int c = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0, false);
        c = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], c, true);
        *(int*)(out + i) = c;
    }
    __shared__ float red[4];
    if (tb.amax[ti]) amax_publish_wg(tb.amax[ti], amx, blockIdx.x, red);
}

// Delayed scaling, once per step: every tracked tensor i gets scale[i] = headroom * amax_i / 448 from the largest value recorded since the last
// call (its slots are cleared; a tensor that recorded nothing keeps its scale), then every fp8 layer l its pair {s_in * s_w, 1 / s_out}.
#define F8LAYER_MAX 16
struct F8LayerIdx { int in[F8LAYER_MAX], w[F8LAYER_MAX], out[F8LAYER_MAX]; int count; };
// One workgroup per tracked tensor reduces that tensor's record (a single workgroup reading all of them ran at one CU's load rate: 11.6 us for 12 tensors).
// The per-layer pairs need EVERY scale: the workgroup whose ticket add comes last computes them.  Hand-off per MI355X_MICROARCH.md (visibility, valid forms):
// every scale is written by an agent-scope (sc1) atomic store, the writing lane drains it (s_waitcnt vmcnt(0)) before its agent-scope ticket add, and the last
// arriver — told by the value its add returned — reads the scales with agent-scope atomic loads, never through its L1.  `ticket` is one device word that the
// last arriver resets, so a replayed graph finds it zero.
__global__ __launch_bounds__(1024) void fp8_scale_update_kernel(unsigned* __restrict__ amax, float* __restrict__ scale, float* __restrict__ inv_scale, int n, float headroom,
                                                                F8LayerIdx li, float* __restrict__ dscale, unsigned* __restrict__ ticket) {
    static_assert(CVAE_AMAX_SLOTS == 4096, "one uint4 per thread");
    __shared__ float red[16];
    __shared__ unsigned last;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, i = blockIdx.x;
    uint4* a4 = (uint4*)amax + (size_t)i * (CVAE_AMAX_SLOTS / 4);
    const uint4 v = a4[t];
    a4[t] = make_uint4(0u, 0u, 0u, 0u);
    float a = fmaxf(fmaxf(__uint_as_float(v.x), __uint_as_float(v.y)), fmaxf(__uint_as_float(v.z), __uint_as_float(v.w)));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a = fmaxf(a, __shfl_xor(a, o, 64));
    if (lane == 0) red[wave] = a;
    __syncthreads();
    if (t == 0) {
        float m = 0.f;
        for (int w_ = 0; w_ < 16; ++w_) m = fmaxf(m, red[w_]);
        if (m > 0.f) {
            const float s_ = headroom * m / CVAE_FP8_MAX;
            __hip_atomic_store(scale + i, s_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(inv_scale + i, 1.f / s_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the stores have left this CU before the ticket says so
        const unsigned got = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = (got == (unsigned)n - 1u) ? 1u : 0u;
    }
    __syncthreads();
    if (last && li.count > 0) {
        if (t < li.count) {
            const float si = __hip_atomic_load(scale + li.in[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float sw = __hip_atomic_load(scale + li.w[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            dscale[2 * t] = si * sw;
            dscale[2 * t + 1] = li.out[t] >= 0 ? 1.f / __hip_atomic_load(scale + li.out[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
        }
    }
    if (last && t == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// max |x| of a tensor into CVAE_AMAX_SLOTS words (calibration of the first step's scales; the training step records its amax in the producers' epilogues)
__global__ __launch_bounds__(256) void absmax_kernel(const void* __restrict__ src, int dtype, int64_t n, unsigned* __restrict__ slots) {
    float amx = 0.f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        amx = fmaxf(amx, fabsf(dtype == CVAE_BF16 ? to_f32(((const bf16*)src)[i]) : ((const float*)src)[i]));
    __shared__ float red[4];
    amax_publish_wg(slots, amx, blockIdx.x, red);
}

// All conv weights of a model packed in ONE launch (the table rides in the kernel arguments): the per-step re-pack of the
// fp32 masters is ~12 tiny tensors, i.e. pure launch latency when issued one by one.
#define PACK_MAX 24
struct PackTable {
    const float* w[PACK_MAX];
    void* out[PACK_MAX];
    int Cs[PACK_MAX], Cl[PACK_MAX], for_up[PACK_MAX], blk_start[PACK_MAX + 1];
    int count, taps;
};
template <typename T>
__global__ __launch_bounds__(256) void pack_weight_multi_kernel(PackTable tb) {
    int ti = 0;
    while (ti + 1 < tb.count && (int)blockIdx.x >= tb.blk_start[ti + 1]) ++ti;
    const int Cs = tb.Cs[ti], Cl = tb.Cl[ti], taps = tb.taps, for_up = tb.for_up[ti];
    const float* w = tb.w[ti];
    T* out = (T*)tb.out[ti];
    const int64_t n = (int64_t)Cs * Cl * taps;
    const int nb = tb.blk_start[ti + 1] - tb.blk_start[ti];
    for (int64_t i = ((int64_t)blockIdx.x - tb.blk_start[ti]) * 256 + threadIdx.x; i < n; i += (int64_t)nb * 256) {
        const int e = (int)(i & 15);
        int64_t rr = i >> 4;
        int cs, cl, tap;
        if (!for_up) { cs = (int)(rr % Cs); rr /= Cs; const int ch = (int)(rr % (Cl / 16)); tap = (int)(rr / (Cl / 16)); cl = ch * 16 + e; }
        else { cl = (int)(rr % Cl); rr /= Cl; const int ch = (int)(rr % (Cs / 16)); tap = (int)(rr / (Cs / 16)); cs = ch * 16 + e; }
        out[i] = from_f32<T>(w[((int64_t)cs * Cl + cl) * taps + tap]);
    }
}

// Both directions of every weight in one launch, through an LDS transpose: a block takes a 16 cs x 16 cl x 16 taps tile of one fp32
// master (64-byte runs, float4 loads), and writes the `down` panel ([tap][cl / 16][cs][16 cl]) and the `up` panel
// ([tap][cs / 16][cl][16 cs]) in 512-byte (bf16) runs — the gather form above reads every source line 16 times.  3D weights
// (64 taps) take 4 blocks per (cs, cl) tile: ~1300 small blocks for the model instead of ~330 long ones.
#define PAIR_MAX 16
struct PairTable {
    const float* w[PAIR_MAX];
    void* down[PAIR_MAX];
    void* up[PAIR_MAX];
    int Cs[PAIR_MAX], Cl[PAIR_MAX], blk_start[PAIR_MAX + 1];
    int count;
    // fp8 training forward: f8dir 1 / 2 = the `down` / `up` panel of this weight is written as fp8 codes of w * *inv_scale ([tap][C_in / 32][C_out][32],
    // into f8out) INSTEAD of its bf16 panel (the forward product reads the fp8 one; the other direction serves the bf16 backward pass); amax records max |w|
    int f8dir[PAIR_MAX];
    fp8* f8out[PAIR_MAX];
    const float* inv_scale[PAIR_MAX];
    unsigned* amax[PAIR_MAX];
};
template <typename T, int TAPS>
__global__ __launch_bounds__(256) void pack_weight_pairs_kernel(PairTable tb) {
    constexpr int TT = 16, TSPLIT = TAPS / TT, TP = TT + 1;
    __shared__ float tile[16 * 17 * TP];                     // [(cs * 17 + cl)][tap]
    int ti = 0;
    while (ti + 1 < tb.count && (int)blockIdx.x >= tb.blk_start[ti + 1]) ++ti;
    const int Cs = tb.Cs[ti], Cl = tb.Cl[ti];
    int blk = (int)blockIdx.x - tb.blk_start[ti];
    const int tq = blk % TSPLIT; blk /= TSPLIT;
    const int nclt = Cl / 16, ncst = Cs / 16;
    const int cst = blk / nclt, clt = blk % nclt, cs0 = cst * 16, cl0 = clt * 16;
    const float* w = tb.w[ti];
    const int f8dir = tb.f8dir[ti];
    float amx = 0.f;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int i = threadIdx.x + it * 256, f = i & 3, cl = (i >> 2) & 15, cs = i >> 6;
        const float4 v = *(const float4*)(w + ((size_t)(cs0 + cs) * Cl + cl0 + cl) * TAPS + tq * TT + 4 * f);
        float* d = tile + (cs * 17 + cl) * TP + 4 * f;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        amx = fmaxf(fmaxf(amx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    __syncthreads();
    T* od = (T*)tb.down[ti];
    T* ou = (T*)tb.up[ti];
    const float mul8 = f8dir ? tb.inv_scale[ti][0] : 1.f;
    // per tap both panels are 256 contiguous elements ([16][16]); a thread writes 8 of them (16 bytes bf16) for one of 8 taps at a time
    const int tg = threadIdx.x >> 5, e0 = (threadIdx.x & 31) * 8, a = e0 >> 4, b0 = e0 & 15;
#pragma unroll
    for (int tl = tg; tl < TT; tl += 8) {
        const int tap = tq * TT + tl;
        __attribute__((aligned(16))) T vd[8], vu[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            vd[q] = from_f32<T>(tile[(a * 17 + b0 + q) * TP + tl]);              // down: (cs = a, cl = b0 + q)
            vu[q] = from_f32<T>(tile[((b0 + q) * 17 + a) * TP + tl]);            // up:   (cl = a, cs = b0 + q)
        }
        T* pd = od + ((size_t)(tap * nclt + clt) * Cs + cs0 + a) * 16 + b0;
        T* pu = ou + ((size_t)(tap * ncst + cst) * Cl + cl0 + a) * 16 + b0;
        constexpr int NU = (8 * sizeof(T)) / 16;
        if (f8dir != 1) {
#pragma unroll
            for (int u = 0; u < NU; ++u) ((uint4*)pd)[u] = ((const uint4*)vd)[u];
        }
        if (f8dir != 2) {
#pragma unroll
            for (int u = 0; u < NU; ++u) ((uint4*)pu)[u] = ((const uint4*)vu)[u];
        }
        if (f8dir) {                                         // 32-channel chunks: this 16-wide tile is one half of a chunk row
            float v8[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v8[q] = f8dir == 1 ? tile[(a * 17 + b0 + q) * TP + tl] : tile[((b0 + q) * 17 + a) * TP + tl];
            fp8* p8 = f8dir == 1 ? tb.f8out[ti] + ((size_t)(tap * (Cl / 32) + (cl0 >> 5)) * Cs + cs0 + a) * 32 + (cl0 & 16) + b0
                                 : tb.f8out[ti] + ((size_t)(tap * (Cs / 32) + (cs0 >> 5)) * Cl + cl0 + a) * 32 + (cs0 & 16) + b0;
            *(uint2*)p8 = pack8_fp8(v8, mul8);
        }
    }
    if (f8dir && tb.amax[ti]) {                              // uniform over the workgroup
        __syncthreads();
        amax_publish_wg(tb.amax[ti], amx, blockIdx.x, tile);
    }
}

// ---------------------------------------------------------------------------------------------- wgrad
// Workgroup: 8 waves; block of 64 cs x 32 cl; one depth tap kd (3D) / all taps (2D); wave w owns kh = w & 3, kw = 0..3 and
// the 32-row half sg = w >> 2 of the cs block: 4 accumulator tiles (64 VGPRs), so two workgroups (16 waves) fit a CU with the
// next tile's global loads held in registers during the MFMA phase.
// Several layers can share ONE launch (cvae_conv_wgrad_multi: the deferred weight gradients of a whole backward pass): the grid is the
// concatenation of the per-layer grids and a workgroup finds its layer in the table that rides in the kernel arguments.  The small layers
// (a handful of tiles each, latency-bound on their own) then run in the shadow of the large ones instead of each paying a launch.
#define WG_MULTI_MAX 8
struct WgradEntry {
    const void* S; const void* L; float* ws; float* bias_ws;
    ConvGeom g;
    int n_split, cb, tg, bias_mode;
    int xb;             // 2D layers at most TW / 2 wide: two samples per tile, side by side in x (the 7 x 7 maps of the MNIST model fill 49 of a tile's 128 positions)
};
struct WgradTable { WgradEntry e[WG_MULTI_MAX]; int blk_start[WG_MULTI_MAX + 1]; int count; };

template <typename T, int ND>
__global__ __launch_bounds__(512, sizeof(T) == 2 ? 4 : 2) void conv_wgrad_kernel(WgradTable tb) {
    int ti = 0;
    while (ti + 1 < tb.count && (int)blockIdx.x >= tb.blk_start[ti + 1]) ++ti;
    const T* __restrict__ S = (const T*)tb.e[ti].S;
    const T* __restrict__ L = (const T*)tb.e[ti].L;
    float* __restrict__ ws = tb.e[ti].ws;
    float* __restrict__ bias_ws = tb.e[ti].bias_ws;
    const ConvGeom g = tb.e[ti].g;
    const int n_split = tb.e[ti].n_split, bias_mode = tb.e[ti].bias_mode, grid_y = tb.e[ti].cb;
    const int xb = (ND == 2) ? tb.e[ti].xb : 0;
    // the layer's own grid (n_split, cb, tg), x fastest
    int lb = (int)blockIdx.x - tb.blk_start[ti];
    const int bx = lb % n_split; lb /= n_split;
    const int by = lb % grid_y, bz = lb / grid_y;
    using TL = Tile<ND, 128>;
    constexpr int NT = 512;
    constexpr int TD = TL::TD, TH = TL::TH, TW = TL::TW;
    // L tile: TD planes (one kd) x IH x IW positions x 32 cl.  2D: two more columns, so that two samples' halos (2 (TW / 2) + 2 columns each) fit side by side (xb)
    constexpr int IW1 = 2 * TW + 2, IW = IW1 + (ND == 2 ? 2 : 0), IH = 2 * TH + 2;
    constexpr int SROW = 64 * sizeof(T), LROW = 32 * sizeof(T);
    constexpr int S_BYTES = 128 * SROW;
    constexpr int FB = 8 * sizeof(T);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_lds = smem;
    char* l_lds = smem + S_BYTES;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int cl_blocks = g.Cl / 32;
    const int cs0 = (by / cl_blocks) * 64, cl0 = (by % cl_blocks) * 32;
    const int kd = (ND == 3) ? bz : 0;
    const int kh = wave & 3, sg = wave >> 2;
    // Bank-conflict-free LDS images for the transposing reads.  One 32-lane group of ds_read_b64_tr_b16 reads 4 k-rows x 64
    // bytes = the whole 256-byte bank row, provided the 4 rows sit in 4 different 64-byte quarters.  The k-rows of a group are 4
    // consecutive output positions m .. m+3 along w.  S image (128-byte rows): the 64-byte halves of row m are swapped when bit 1
    // of m is set.  L image (64-byte rows): the stride-2 input columns x = 2 w + kw are split into an even-x and an odd-x plane,
    // so consecutive w are consecutive rows of plane kw & 1 and every tap is a constant row offset (kw >> 1).
    constexpr int LHALF = IW / 2, LPLANE = TD * IH * LHALF;  // rows per line / per parity plane
    auto s_byte = [](int m, int c) -> int { return m * SROW + ((((c >> 3) ^ (((m >> 1) & 1) << 2)) << 3) + (c & 7)) * (int)sizeof(T); };
    auto l_row = [](int line, int x) -> int { return (x & 1) * LPLANE + line * LHALF + (x >> 1); };
    const int tiles_per_b = g.tiles_d * g.tiles_h * g.tiles_w, total_tiles = (xb ? (g.B + 1) / 2 : g.B) * tiles_per_b;
#ifdef CVAE_STAMP
    const unsigned stamp_wg = blockIdx.x;
    if (t == 0 && stamp_wg < CVAE_STAMP_WGS) {
        g_stamp[(size_t)stamp_wg * CVAE_STAMP_SLOTS + 30] = wall_clock64();
        g_stamp[(size_t)stamp_wg * CVAE_STAMP_SLOTS + 29] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | __builtin_amdgcn_s_getreg(63492);
    }
    int stamp_it = 0;
#endif
    STAMP(0);

    f32x16 acc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[k][e] = 0.f;

    // Bias gradients ride along (no separate channel-sum pass over the gradient tensor): bias_mode 1 = per-channel sum of S (Conv
    // layers), done by the (cl block 0, kd 0) workgroups from their S tiles; bias_mode 2 = per-channel sum of L (ConvTranspose
    // layers), done by the (cs block 0, kd 1 | 2) workgroups from the non-halo part of their L tiles (kd = 1 sees the even planes,
    // kd = 2 the odd ones; in 2D the single tap group sees the one plane).  Partials go to bias_ws, wgrad_reduce_kernel sums them.
    const bool bias_s = bias_mode == 1 && (by % cl_blocks) == 0 && kd == 0;
    const bool bias_l = bias_mode == 2 && (by / cl_blocks) == 0 && (ND == 2 || kd == 1 || kd == 2);
    float bacc = 0.f;
    // S tile: 128 positions x 64 channels; L tile: planes lz = 2 (o0d + d) - 1 + kd, rows 2 o0h - 1 + y, cols 2 o0w - 1 + x.
    // Both LDS images are piece-linear (piece `it` at byte 16 it / 32 it).  The global loads of tile i+1 are issued right after
    // tile i has been stored to LDS, so they are in flight during tile i's MFMA phase; out-of-range pieces are loaded from a
    // clamped (valid) address and zeroed when they are stored, which keeps the loads unconditional and straight-line.
    // Addresses are 32-bit offsets inside one sample (geom_ok bounds the sample size) on a wave-uniform 64-bit base.
    constexpr int SN = (128 * 8) / NT, LNPOS = TD * IH * IW, LN = (LNPOS * 4 + NT - 1) / NT;
    Piece<T> sp[SN], lp[LN];
    unsigned okmask = 0;
    const int s_sample = g.sd * g.sh * g.sw * g.Cs, l_sample = g.ld * g.lh * g.lw * g.Cl;
    auto load_tile = [&](int tile) {
        int tt = tile;
        const int tw_i = tt % g.tiles_w; tt /= g.tiles_w;
        const int th_i = tt % g.tiles_h; tt /= g.tiles_h;
        const int td_i = tt % g.tiles_d;
        const int b = (tt / g.tiles_d) * (1 + xb);            // xb: samples b (tile columns 0 .. TW / 2 - 1) and b + 1 (the other half)
        const int o0d = td_i * TD, o0h = th_i * TH, o0w = tw_i * TW;
        const T* Sb = S + (size_t)b * s_sample + cs0;
        const T* Lb = L + (size_t)b * l_sample + cl0;
        const bool b1ok = b + 1 < g.B;
        okmask = 0;
        int tz = t;
        asm volatile("" : "+v"(tz));                         // opaque: keeps the per-piece index math inside the call (hoisted, it spills)
#pragma unroll
        for (int i = 0; i < SN; ++i) {
            const int it = tz + i * NT, piece = it & 7, m = it >> 3;
            const int w = m % TW, hh = m / TW % TH, d = m / (TW * TH);
            const int sx = (xb && w >= TW / 2) ? 1 : 0;
            const int od = o0d + d, oh = o0h + hh, ow = o0w + w - sx * (TW / 2);
            const bool ok = (od < g.sd) & (oh < g.sh) & (ow < g.sw) & (!sx | b1ok);
            const unsigned off = ((min(od, g.sd - 1) * g.sh + min(oh, g.sh - 1)) * g.sw + min(ow, g.sw - 1)) * g.Cs + piece * 8 + ((sx && b1ok) ? (unsigned)s_sample : 0u);
            piece_load_raw<T>(sp[i], Sb + off);
            okmask |= (unsigned)ok << i;
        }
#pragma unroll
        for (int i = 0; i < LN; ++i) {
            const int it = tz + i * NT, piece = it & 3, pos = min(it >> 2, LNPOS - 1);
            const int x = pos % IW, y = pos / IW % IH, d = pos / (IW * IH);
            const int sx = (xb && x >= IW / 2) ? 1 : 0;
            const int lz = (ND == 3) ? 2 * (o0d + d) - 1 + kd : 0, ly = 2 * o0h - 1 + y, lx = 2 * o0w - 1 + x - sx * (IW / 2);
            const bool ok = (lz >= 0) & (lz < g.ld) & (ly >= 0) & (ly < g.lh) & (lx >= 0) & (lx < g.lw) & (!sx | b1ok) & (xb | (x < IW1));
            const unsigned off = ((min(max(lz, 0), g.ld - 1) * g.lh + min(max(ly, 0), g.lh - 1)) * g.lw + min(max(lx, 0), g.lw - 1)) * g.Cl + piece * 8 + ((sx && b1ok) ? (unsigned)l_sample : 0u);
            piece_load_raw<T>(lp[i], Lb + off);
            okmask |= (unsigned)ok << (SN + i);
        }
    };
    if (bx < total_tiles) load_tile(bx);
    for (int tile = bx; tile < total_tiles; tile += n_split) {
        __syncthreads();                                     // the previous tile's readers are done with both images
#ifdef CVAE_STAMP
        if (stamp_it < 8) STAMP(2 + 3 * stamp_it);
#endif
        {
            int tz = t;
            asm volatile("" : "+v"(tz));                     // as in load_tile: recompute, do not keep 8 offsets live across the MFMA phase
#pragma unroll
            for (int i = 0; i < SN; ++i) {
                const int it = tz + i * NT;
                piece_store_sel<T>(sp[i], (okmask >> i) & 1, s_lds + s_byte(it >> 3, (it & 7) * 8));
            }
#pragma unroll
            for (int i = 0; i < LN; ++i) {
                const int it = tz + i * NT, pos = it >> 2;
                if (it < LNPOS * 4) piece_store_sel<T>(lp[i], (okmask >> (SN + i)) & 1, l_lds + (l_row(pos / IW, pos % IW) * 4 + (it & 3)) * FB);
            }
        }
        __syncthreads();
#ifdef CVAE_STAMP
        if (stamp_it < 8) STAMP(3 + 3 * stamp_it);
#endif
        if (tile + n_split < total_tiles) load_tile(tile + n_split);
        if (bias_s) {                                        // thread (c, pg): 16 of the tile's 128 positions of channel c
            const int c = t & 63, pg = t >> 6;
#pragma unroll 4
            for (int j = 0; j < 16; ++j) bacc += to_f32(*(const T*)(s_lds + s_byte(pg * 16 + j, c)));
        } else if (bias_l) {                                 // thread (c, pg): 32 of the 512 owned (non-halo) positions of channel c
            const int c = t & 31, pg = t >> 5;
#pragma unroll 4
            for (int j = 0; j < 32; ++j) {
                const int pp = pg + 16 * j;
                const int xx = pp % (2 * TW), yy = pp / (2 * TW) % (2 * TH), dd = pp / (4 * TW * TH);
                const int xs = xx + 1 + ((xb && xx >= TW) ? 2 : 0);           // xb: the second sample's own columns start two columns further right
                bacc += to_f32(*(const T*)(l_lds + l_row(dd * IH + yy + 1, xs) * LROW + c * (int)sizeof(T)));
            }
        }
        if constexpr (sizeof(T) == 2) {
            // bf16: transposing LDS reads.  Lane i of 16-lane group gq supplies the address of k-row q = i>>2,
            // 4 channels at 4*(i&3); it receives channel i of the 4 k-rows  (cdna_hip_programming.md T10).
            const int gq = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
            const int hk = gq >> 1, colblk = gq & 1;
            typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
            // k-row of (c, jj): position m = 16 c + r0 with r0 = 8 hk + 4 jj + q.  The lane part (w, hh) = (r0 % TW, r0 / TW) and
            // the c part (hh, d) = (16 c / TW % TH, 16 c / (TW TH)) add without carries, so with the loop fully unrolled every
            // LDS address below is one per-lane base plus a compile-time offset.
            const char* abase[2];
            const char* bbase[2];
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int r0 = 8 * hk + 4 * jj + q;
                abase[jj] = s_lds + s_byte(r0, sg * 32 + 16 * colblk + 4 * p);
                bbase[jj] = l_lds + l_row(2 * (r0 / TW) + kh, 2 * (r0 % TW) + ((xb && (r0 % TW) >= TW / 2) ? 2 : 0)) * LROW + (16 * colblk + 4 * p) * 2;
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int hh_c = (16 * c / TW) % TH, d_c = (16 * c) / (TW * TH);
                bf16x8 a;
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(abase[jj] + 16 * c * SROW));
                    a[4 * jj + 0] = v[0]; a[4 * jj + 1] = v[1]; a[4 * jj + 2] = v[2]; a[4 * jj + 3] = v[3];
                }
#pragma unroll
                for (int kw = 0; kw < 4; ++kw) {
                    bf16x8 bv;
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const char* bp = bbase[jj] + (((kw & 1) * LPLANE + (d_c * IH + 2 * hh_c) * LHALF + (kw >> 1)) * LROW);
                        const bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)bp);
                        bv[4 * jj + 0] = v[0]; bv[4 * jj + 1] = v[1]; bv[4 * jj + 2] = v[2]; bv[4 * jj + 3] = v[3];
                    }
                    acc[kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bv, acc[kw], 0, 0, 0);
                }
            }
        } else {
            // fp32: 32x32x2 MFMA, lane (r, h) feeds A[r][k = h], B[k = h][r]: one element each, no transpose needed
            const int r = lane & 31, hk = lane >> 5;
#pragma unroll 1
            for (int mm = 0; mm < 128; mm += 2) {
                const int m = mm + hk;
                const int w = m % TW, hh = m / TW % TH, d = m / (TW * TH);
                const int line = d * IH + 2 * hh + kh;
                const float a = *(const float*)(s_lds + s_byte(m, sg * 32 + r));
#pragma unroll
                for (int kw = 0; kw < 4; ++kw) {
                    const float bv = *(const float*)(l_lds + l_row(line, 2 * w + kw + ((xb && w >= TW / 2) ? 2 : 0)) * LROW + r * 4);
                    acc[kw] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[kw], 0, 0, 0);
                }
            }
        }
#ifdef CVAE_STAMP
        if (stamp_it < 8) STAMP(4 + 3 * stamp_it);
        ++stamp_it;
#endif
    }
    STAMP(26);
    // ---- write-out: this workgroup's partial sums leave as ONE slab [kh][kw][64 cs][32 cl] of plain 128-byte-row stores;
    // wgrad_reduce_kernel sums the slabs (fp32 atomics here cost more than the MFMA phase: 67 MB of adds at < 1 TB/s) ----
    if (bias_s || bias_l) {                                   // block-uniform
        __syncthreads();                                     // the last tile's readers are done with the S image: reuse it as scratch
        float* red = (float*)s_lds;
        red[t] = bacc;
        __syncthreads();
        if (bias_s && t < 64) {
            float v = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) v += red[q * 64 + t];
            bias_ws[((size_t)(by / cl_blocks) * n_split + bx) * 64 + t] = v;
        }
        if (bias_l && t < 32) {
            float v = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) v += red[q * 32 + t];
            const int half = (ND == 3) ? kd - 1 : 0, nhalf = (ND == 3) ? 2 : 1;
            bias_ws[(((size_t)(by % cl_blocks) * nhalf + half) * n_split + bx) * 32 + t] = v;
        }
    }
    const int col = lane & 31, hq = lane >> 5;
    float* slab = ws + ((size_t)(bz * grid_y + by) * n_split + bx) * 32768;
    // slab layout [kh][64 cs][32 cl][kw]: the lane owns the four kw values of (cs row, cl col), so they leave as ONE 16-byte store (32 lanes = 512 contiguous
    // bytes) and wgrad_reduce_kernel reads them back as one 16-byte load per slab — a quarter of the memory instructions of the [kh][kw][cs][cl] form on both sides
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int row = sg * 32 + (e & 3) + 8 * (e >> 2) + 4 * hq;
        *(float4*)(slab + ((kh * 64 + row) * 32 + col) * 4) = make_float4(acc[0][e], acc[1][e], acc[2][e], acc[3][e]);
    }
#ifdef CVAE_STAMP
    __builtin_amdgcn_s_waitcnt(0);
    STAMP(27);
    if (t == 0 && stamp_wg < CVAE_STAMP_WGS) g_stamp[(size_t)stamp_wg * CVAE_STAMP_SLOTS + 31] = wall_clock64();
#endif
}

// dW[cs][cl][kd][kh][0..3] = sum over the n_split slabs of group (kd, channel block).  One thread per (kd, kh, cs, cl)
// sums the 4 kw values (a 16-byte store into the reference layout); 4 thread groups split the slab range, LDS combines.
// Blocks past the dW range sum the bias partials: block j handles 64 channels, 4 thread groups split the partial rows.
struct WgradReduceEntry {
    const float* ws; float* dW; const float* bias_ws; float* dbias;
    int Cs, Cl, n_split, cb, dw_blocks, bias_n, bias_rows, bias_width;
};
struct WgradReduceTable { WgradReduceEntry e[WG_MULTI_MAX]; int blk_start[WG_MULTI_MAX + 1]; int count, taps; };
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(WgradReduceTable tb) {
    int ti = 0;
    while (ti + 1 < tb.count && (int)blockIdx.x >= tb.blk_start[ti + 1]) ++ti;
    const float* __restrict__ ws = tb.e[ti].ws;
    float* __restrict__ dW = tb.e[ti].dW;
    const float* __restrict__ bias_ws = tb.e[ti].bias_ws;
    float* __restrict__ dbias = tb.e[ti].dbias;
    const int Cl = tb.e[ti].Cl, taps = tb.taps, n_split = tb.e[ti].n_split, cb = tb.e[ti].cb, dw_blocks = tb.e[ti].dw_blocks;
    const int bias_n = tb.e[ti].bias_n, bias_rows = tb.e[ti].bias_rows, bias_width = tb.e[ti].bias_width;
    const int lbx = (int)blockIdx.x - tb.blk_start[ti];
    __shared__ float4 part[4][64];
    if (lbx >= dw_blocks) {
        // bias_ws is [channel group][bias_rows][bias_width]; channel = group * bias_width + lane
        __shared__ float bred[4][64];
        const int ch = (lbx - dw_blocks) * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
        float v = 0.f;
        if (ch < bias_n) {
            const float* base = bias_ws + (size_t)(ch / bias_width) * bias_rows * bias_width + (ch % bias_width);
            for (int r = q; r < bias_rows; r += 4) v += base[(size_t)r * bias_width];
        }
        bred[q][threadIdx.x & 63] = v;
        __syncthreads();
        if (q == 0 && ch < bias_n) dbias[ch] = bred[0][threadIdx.x] + bred[1][threadIdx.x] + bred[2][threadIdx.x] + bred[3][threadIdx.x];
        return;
    }
    const int cl_blocks = Cl / 32;
    // lbx enumerates (group = kd * cb + block, kh, cs row pair) ; 64 threads = 2 cs rows x 32 cl
    int bi = lbx;
    const int rowpair = bi % 32; bi /= 32;
    const int kh = bi % 4; bi /= 4;
    const int grp = bi;                                      // kd * cb + channel block
    const int kd = grp / cb, blk = grp % cb;
    const int lane64 = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int row = rowpair * 2 + (lane64 >> 5), col = lane64 & 31;
    const float* base = ws + (size_t)grp * n_split * 32768 + ((kh * 64 + row) * 32 + col) * 4;      // slab layout [kh][cs][cl][kw]
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    // U slabs' loads are issued before the first add (clamped slab index, predicated add: the order of the sum is unchanged).  The plain loop compiled to
    // load / wait / add per slab: one dependent round trip per slab, 16-64 of them per thread.
    constexpr int U = 8;
    for (int x0 = q; x0 < n_split; x0 += 4 * U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = *(const float4*)(base + (size_t)min(x0 + 4 * u, n_split - 1) * 32768);
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (x0 + 4 * u < n_split) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
    }
    part[q][lane64] = a;
    __syncthreads();
    if (q == 0) {
        const float4 b = part[1][lane64], c = part[2][lane64], d = part[3][lane64];
        a.x += b.x + c.x + d.x; a.y += b.y + c.y + d.y; a.z += b.z + c.z + d.z; a.w += b.w + c.w + d.w;
        const int cs = (blk / cl_blocks) * 64 + row, cl = (blk % cl_blocks) * 32 + col;
        *(float4*)(dW + ((size_t)cs * Cl + cl) * taps + kd * 16 + kh * 4) = a;
    }
}

#define WGRAD_MAX_WG 512
#ifndef CVAE_WGRAD_XPAIR
#define CVAE_WGRAD_XPAIR 1
#endif
template <typename T, int ND> constexpr size_t wgrad_lds_bytes() {
    using TL = Tile<ND, 128>;
    return (size_t)128 * 64 * sizeof(T) + (size_t)TL::TD * (2 * TL::TH + 2) * (2 * TL::TW + 2 + (ND == 2 ? 2 : 0)) * 32 * sizeof(T);
}
// Launch geometry of one layer's weight gradient: fills the two table entries, returns the workgroup counts of the main and the reduce pass.
template <typename T, int ND>
int plan_wgrad(const void* S, const void* L, float* ws, float* dW, float* dbias, int bias_mode, ConvGeom g, WgradEntry* me, WgradReduceEntry* re, int* main_blocks,
               int* reduce_blocks, long long n_split_req = 0) {
    using TL = Tile<ND, 128>;
    g.tiles_d = (g.sd + TL::TD - 1) / TL::TD; g.tiles_h = (g.sh + TL::TH - 1) / TL::TH; g.tiles_w = (g.sw + TL::TW - 1) / TL::TW;
    // 2D layers at most half a tile wide (7 x 7, 4 x 4 maps): two samples per tile — a property of the layer's shape alone, like the slab count below
    const int xb = (CVAE_WGRAD_XPAIR && ND == 2 && g.sw <= TL::TW / 2 && g.B >= 2) ? 1 : 0;
    const long long total_tiles = (long long)(xb ? (g.B + 1) / 2 : g.B) * g.tiles_d * g.tiles_h * g.tiles_w;
    const int cb = (g.Cs / 64) * (g.Cl / 32), tg = (ND == 3) ? 4 : 1;
    // each workgroup ends with a 128 KB slab: ~2 workgroups per CU at most, and >= 4 tiles of work per slab
    long long n_split = WGRAD_MAX_WG / ((long long)cb * tg);   // in-step scan of 256 / 512 / 768 / 1024: 234 / 231 / 248 / 253 us for the six launches + reductions
    // >= 4 tiles per slab, except that a small layer may go down to one tile per slab until it has one workgroup per CU (in-step scan:
    // dec1 29 -> 22 us, dec2 26 -> 20 us incl. their reductions; more slabs than that only lengthen the reduction)
    long long floor_split = 256 / ((long long)cb * tg);
    if (floor_split > total_tiles) floor_split = total_tiles;
    long long by_tiles = total_tiles / 4;
    if (by_tiles < floor_split) by_tiles = floor_split;
    if (n_split > by_tiles) n_split = by_tiles;
#ifdef CVAE_TUNE                                             // tuning builds only (make EXTRA=-DCVAE_TUNE, tools/kbench.py): never in the shipped library
    if (const char* e = getenv("CVAE_TUNE_WGRAD_NSPLIT")) n_split = atoll(e);
    if (n_split * cb * tg > WGRAD_MAX_WG && n_split > 1) n_split = WGRAD_MAX_WG / ((long long)cb * tg) > 0 ? WGRAD_MAX_WG / ((long long)cb * tg) : 1;   // stay inside the validated workspace
#endif
    {   // ~CVAE_WG_TILES tiles of 128 positions per slab, at least CVAE_WG_MIN_WG workgroups: a function of THIS layer only, the same whether the layer
        // is launched alone (cvae_conv_wgrad) or as one entry of a grouped launch (cvae_conv_wgrad_multi) — the summation order, and with it every bit
        // of the gradient, does not depend on how the caller batches its launches
        long long req = (total_tiles + CVAE_WG_TILES - 1) / CVAE_WG_TILES;
        const long long groups = (long long)cb * tg;
        if (req * groups < CVAE_WG_MIN_WG) req = (CVAE_WG_MIN_WG + groups - 1) / groups;
        if (req < n_split) n_split = req;
    }
    if (n_split_req > 0 && n_split_req < n_split) n_split = n_split_req;
    if (n_split < 1) n_split = 1;
    if (n_split > total_tiles) n_split = total_tiles;
    if ((long long)cb * tg * n_split > (1 << 24)) return CVAE_E_BADSHAPE;
    // bias partials live behind the slabs: [channel group][rows][width] with (mode 1) 64-wide groups of cs, n_split rows;
    // (mode 2) 32-wide groups of cl, (2 plane parities in 3D) * n_split rows
    const long long wgs = (long long)cb * tg * n_split;
    float* bias_ws = ws + (size_t)(wgs > WGRAD_MAX_WG ? wgs : WGRAD_MAX_WG) * 32768;
    if (!dbias) bias_mode = 0;
    *me = WgradEntry{S, L, ws, bias_ws, g, (int)n_split, cb, tg, bias_mode, xb};
    *main_blocks = (int)wgs;
    const int dw_blocks = cb * tg * 4 * 32;
    const int bias_n = bias_mode == 1 ? g.Cs : (bias_mode == 2 ? g.Cl : 0), bias_width = bias_mode == 1 ? 64 : 32;
    const int bias_rows = (int)n_split * ((bias_mode == 2 && ND == 3) ? 2 : 1);
    *re = WgradReduceEntry{ws, dW, bias_ws, dbias, g.Cs, g.Cl, (int)n_split, cb, dw_blocks, bias_n, bias_rows, bias_width};
    *reduce_blocks = dw_blocks + (bias_n + 63) / 64;
    return CVAE_OK;
}
template <typename T, int ND>
int launch_wgrad_tables(WgradTable& mt, WgradReduceTable& rt, hipStream_t stream) {
    constexpr size_t LDS = wgrad_lds_bytes<T, ND>();
    static_assert(LDS <= 160 * 1024, "LDS tile exceeds the 160 KiB of a CDNA4 CU");
    auto kern = conv_wgrad_kernel<T, ND>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS) != hipSuccess) return CVAE_E_LAUNCH;
        attr_set = true;
    }
    rt.taps = (ND == 3) ? 64 : 16;
    hipLaunchKernelGGL(kern, dim3((unsigned)mt.blk_start[mt.count]), dim3(512), LDS, stream, mt);
    CVAE_CHECK_LAUNCH();
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)rt.blk_start[rt.count]), dim3(256), 0, stream, rt);
    CVAE_CHECK_LAUNCH();
    return CVAE_OK;
}
template <typename T, int ND>
int launch_wgrad(const void* S, const void* L, float* ws, float* dW, float* dbias, int bias_mode, ConvGeom g, hipStream_t stream) {
    WgradTable mt;
    WgradReduceTable rt;
    int mb, rb;
    const int rc = plan_wgrad<T, ND>(S, L, ws, dW, dbias, bias_mode, g, &mt.e[0], &rt.e[0], &mb, &rb);
    if (rc != CVAE_OK) return rc;
    mt.count = rt.count = 1;
    mt.blk_start[0] = rt.blk_start[0] = 0;
    mt.blk_start[1] = mb; rt.blk_start[1] = rb;
    return launch_wgrad_tables<T, ND>(mt, rt, stream);
}

bool geom_ok(int64_t B, int64_t sd, int64_t sh, int64_t sw, int64_t Cs, int64_t ld, int64_t lh, int64_t lw, int64_t Cl, int nd) {
    if (nd != 2 && nd != 3) return false;
    if (B < 0 || sd <= 0 || sh <= 0 || sw <= 0 || Cs <= 0 || ld <= 0 || lh <= 0 || lw <= 0 || Cl <= 0) return false;
    if (B > 65535 || sd > 32767 || sh > 32767 || sw > 32767 || ld > 65535 || lh > 65535 || lw > 65535) return false;
    if (nd == 2 && (sd != 1 || ld != 1)) return false;
    if (sd * sh * sw * Cs >= (int64_t(1) << 31) || ld * lh * lw * Cl >= (int64_t(1) << 31)) return false;   // 32-bit offsets inside a sample
    auto pair_ok = [](int64_t s, int64_t l) { return s == l / 2; };        // floor((l + 2 - 4) / 2) + 1 == l / 2
    if (nd == 3 && !pair_ok(sd, ld)) return false;
    return pair_ok(sh, lh) && pair_ok(sw, lw) && lh >= 2 && lw >= 2 && (nd == 2 || ld >= 2);
}

}  // namespace

// C1 kernels live in conv_c1.hip
int cvae_conv_down_c1(const void* L, int l_dtype, const float* w, const float* bias, const void* mask, void* S, int64_t B, int64_t sd, int64_t sh, int64_t sw,
                      int64_t Cs, int64_t ld, int64_t lh, int64_t lw, int nd, int dtype, int act, hipStream_t stream, F8Side f8 = F8Side{nullptr, nullptr, nullptr});
int cvae_conv_up_c1(const void* S, const float* w, const float* bias, const void* mask, void* L, int64_t B, int64_t sd, int64_t sh, int64_t sw,
                    int64_t Cs, int64_t ld, int64_t lh, int64_t lw, int nd, int dtype, int act, hipStream_t stream, long long walk_units = 0);
int cvae_conv_up_c1_fp8in_impl(const void* S8, const float* w, const float* bias, void* L, float in_scale, int64_t B, int64_t sd, int64_t sh, int64_t sw, int64_t Cs, int act,
                               hipStream_t stream);
size_t cvae_conv_wgrad_c1_workspace_bytes(int64_t Cs, int nd);
int cvae_conv_wgrad_c1(const void* S, const void* L, int l_dtype, float* dW, float* dbias, float* dbias_l, void* workspace, size_t workspace_bytes, int64_t B, int64_t sd, int64_t sh,
                       int64_t sw, int64_t Cs, int64_t ld, int64_t lh, int64_t lw, int nd, int dtype, hipStream_t stream);

#ifdef CVAE_STAMP
extern "C" int cvae_debug_stamps(unsigned long long* host, size_t count) {
    if (count > (size_t)CVAE_STAMP_WGS * CVAE_STAMP_SLOTS) return CVAE_E_BADSHAPE;
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamp), count * sizeof(unsigned long long)) == hipSuccess ? CVAE_OK : CVAE_E_LAUNCH;
}
#endif

extern "C" size_t cvae_conv_packed_weight_bytes(int64_t Cs, int64_t Cl, int nd, int dtype) {
    const int64_t taps = (nd == 3) ? 64 : 16;
    return (size_t)(Cs * Cl * taps) * (dtype == CVAE_BF16 ? 2 : 4);
}

extern "C" int cvae_conv_pack_weight(const float* w, void* packed, int64_t Cs, int64_t Cl, int nd, int for_up, int dtype, void* stream) {
    if ((nd != 2 && nd != 3) || Cs <= 0 || Cl <= 0) return CVAE_E_BADSHAPE;
    if ((!for_up && Cl % 16) || (for_up && (Cs % 16 || Cl % 32))) return CVAE_E_UNSUPPORTED;      // what cvae_conv_up_fp8 accepts
    if (!w || !packed) return CVAE_E_NULLPTR;
    const int taps = (nd == 3) ? 64 : 16;
    const int64_t n = Cs * Cl * taps;
    if (dtype == CVAE_BF16) hipLaunchKernelGGL(pack_weight_kernel<bf16>, dim3(cvae_grid_1d(n, 256)), dim3(256), 0, (hipStream_t)stream, w, (bf16*)packed, (int)Cs, (int)Cl, taps, for_up);
    else if (dtype == CVAE_F32) hipLaunchKernelGGL(pack_weight_kernel<float>, dim3(cvae_grid_1d(n, 256)), dim3(256), 0, (hipStream_t)stream, w, (float*)packed, (int)Cs, (int)Cl, taps, for_up);
    else return CVAE_E_DTYPE;
    CVAE_CHECK_LAUNCH();
    return CVAE_OK;
}

extern "C" int cvae_conv_pack_weights(const float* const* w, void* const* packed, const int64_t* Cs, const int64_t* Cl, const int* for_up,
                                       int count, int nd, int dtype, void* stream) {
    if ((nd != 2 && nd != 3) || count < 0) return CVAE_E_BADSHAPE;
    if (count == 0) return CVAE_OK;
    if (!w || !packed || !Cs || !Cl || !for_up) return CVAE_E_NULLPTR;
    if (dtype != CVAE_BF16 && dtype != CVAE_F32) return CVAE_E_DTYPE;
    for (int c0 = 0; c0 < count; c0 += PACK_MAX) {
        PackTable tb;
        const int cnt = (count - c0 < PACK_MAX) ? count - c0 : PACK_MAX;
        tb.taps = (nd == 3) ? 64 : 16;
        int blocks = 0;
        for (int i = 0; i < cnt; ++i) {
            const int64_t cs = Cs[c0 + i], cl = Cl[c0 + i];
            if (cs <= 0 || cl <= 0) return CVAE_E_BADSHAPE;
            if ((!for_up[c0 + i] && cl % 16) || (for_up[c0 + i] && cs % 16)) return CVAE_E_UNSUPPORTED;
            if (!w[c0 + i] || !packed[c0 + i]) return CVAE_E_NULLPTR;
            tb.w[i] = w[c0 + i]; tb.out[i] = packed[c0 + i]; tb.Cs[i] = (int)cs; tb.Cl[i] = (int)cl; tb.for_up[i] = for_up[c0 + i];
            tb.blk_start[i] = blocks;
            int64_t nb = (cs * cl * tb.taps + 256 * 8 - 1) / (256 * 8);
            if (nb > 1024) nb = 1024;
            if (nb < 1) nb = 1;
            blocks += (int)nb;
        }
        tb.blk_start[cnt] = blocks;
        tb.count = cnt;
        if (dtype == CVAE_BF16) hipLaunchKernelGGL(pack_weight_multi_kernel<bf16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, tb);
        else hipLaunchKernelGGL(pack_weight_multi_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, tb);
        CVAE_CHECK_LAUNCH();
    }
    return CVAE_OK;
}

// channel counts an fp8 product accepts (cvae_conv_fp8): conv C_in % 32, C_out % 64; ConvTranspose C_in % 32, C_out % 32, C_out > 1
static bool fp8_pack_ok(int64_t Cs, int64_t Cl, int for_up) { return for_up ? (Cs % 32 == 0 && Cl % 32 == 0 && Cl > 1) : (Cl % 32 == 0 && Cs % 64 == 0); }
extern "C" int cvae_conv_pack_weight_pairs_f8(const float* const* w, void* const* packed_down, void* const* packed_up, const int64_t* Cs, const int64_t* Cl,
                                              const int* f8dir, void* const* f8out, const float* const* inv_scale_dev, void* const* amax_slots,
                                              int count, int nd, int dtype, void* stream) {
    if ((nd != 2 && nd != 3) || count < 0) return CVAE_E_BADSHAPE;
    if (count == 0) return CVAE_OK;
    if (!w || !packed_down || !packed_up || !Cs || !Cl) return CVAE_E_NULLPTR;
    if (dtype != CVAE_BF16 && dtype != CVAE_F32) return CVAE_E_DTYPE;
    const int tsplit = (nd == 3) ? 4 : 1;                    // blocks per (cs, cl) tile: 16 taps each
    for (int c0 = 0; c0 < count; c0 += PAIR_MAX) {
        PairTable tb;
        const int cnt = (count - c0 < PAIR_MAX) ? count - c0 : PAIR_MAX;
        long long blocks = 0;
        for (int i = 0; i < cnt; ++i) {
            const int64_t cs = Cs[c0 + i], cl = Cl[c0 + i];
            if (cs <= 0 || cl <= 0) return CVAE_E_BADSHAPE;
            if (cs % 16 || cl % 16) return CVAE_E_UNSUPPORTED;
            const int fd = f8dir ? f8dir[c0 + i] : 0;
            if (fd < 0 || fd > 2) return CVAE_E_BADSHAPE;
            if (fd) {
                if (dtype != CVAE_BF16) return CVAE_E_DTYPE;
                if (!fp8_pack_ok(cs, cl, fd == 2)) return CVAE_E_UNSUPPORTED;
                if (!f8out || !f8out[c0 + i] || !inv_scale_dev || !inv_scale_dev[c0 + i]) return CVAE_E_NULLPTR;
            }
            if (!w[c0 + i] || (fd != 1 && !packed_down[c0 + i]) || (fd != 2 && !packed_up[c0 + i])) return CVAE_E_NULLPTR;
            tb.w[i] = w[c0 + i]; tb.down[i] = packed_down[c0 + i]; tb.up[i] = packed_up[c0 + i]; tb.Cs[i] = (int)cs; tb.Cl[i] = (int)cl;
            tb.f8dir[i] = fd; tb.f8out[i] = fd ? (fp8*)f8out[c0 + i] : nullptr; tb.inv_scale[i] = fd ? inv_scale_dev[c0 + i] : nullptr;
            tb.amax[i] = (fd && amax_slots) ? (unsigned*)amax_slots[c0 + i] : nullptr;
            tb.blk_start[i] = (int)blocks;
            blocks += (cs / 16) * (cl / 16) * tsplit;
            if (blocks > (1 << 30)) return CVAE_E_BADSHAPE;
        }
        tb.blk_start[cnt] = (int)blocks;
        tb.count = cnt;
        const dim3 grid((unsigned)blocks);
        hipStream_t st = (hipStream_t)stream;
        if (dtype == CVAE_BF16) {
            if (nd == 3) hipLaunchKernelGGL((pack_weight_pairs_kernel<bf16, 64>), grid, dim3(256), 0, st, tb);
            else hipLaunchKernelGGL((pack_weight_pairs_kernel<bf16, 16>), grid, dim3(256), 0, st, tb);
        } else {
            if (nd == 3) hipLaunchKernelGGL((pack_weight_pairs_kernel<float, 64>), grid, dim3(256), 0, st, tb);
            else hipLaunchKernelGGL((pack_weight_pairs_kernel<float, 16>), grid, dim3(256), 0, st, tb);
        }
        CVAE_CHECK_LAUNCH();
    }
    return CVAE_OK;
}
extern "C" int cvae_conv_pack_weight_pairs(const float* const* w, void* const* packed_down, void* const* packed_up, const int64_t* Cs, const int64_t* Cl,
                                           int count, int nd, int dtype, void* stream) {
    return cvae_conv_pack_weight_pairs_f8(w, packed_down, packed_up, Cs, Cl, nullptr, nullptr, nullptr, nullptr, count, nd, dtype, stream);
}

#define GEOM_INIT() ConvGeom g{(int)B, (int)sd, (int)sh, (int)sw, (int)Cs, (int)ld, (int)lh, (int)lw, (int)Cl, 0, 0, 0}

extern "C" size_t cvae_conv_data_workspace_bytes(int64_t B, int64_t sd, int64_t sh, int64_t sw, int64_t Cs,
                                                  int64_t ld, int64_t lh, int64_t lw, int64_t Cl, int nd, int for_up) {
    if (!geom_ok(B, sd, sh, sw, Cs, ld, lh, lw, Cl, nd) || B == 0 || Cl == 1) return 0;
    GEOM_INIT();
    if (!for_up) {
        if (Cl % 16 || Cs % 64) return 0;
        return nd == 3 ? data_workspace_bytes<3, false, 128, 64>(g) : data_workspace_bytes<2, false, 128, 64>(g);
    }
    if (Cs % 16 || Cl % 32) return 0;
    if (Cl % 64 == 0) return nd == 3 ? data_workspace_bytes<3, true, 128, 64>(g) : data_workspace_bytes<2, true, 128, 64>(g);
    return nd == 3 ? data_workspace_bytes<3, true, 256, 32>(g) : data_workspace_bytes<2, true, 256, 32>(g);
}

static int conv_down_impl(const void* L, const void* w, const float* bias, const void* mask, void* S,
                          int64_t B, int64_t sd, int64_t sh, int64_t sw, int64_t Cs,
                          int64_t ld, int64_t lh, int64_t lw, int64_t Cl, int nd, int dtype, int act,
                          void* workspace, size_t workspace_bytes, void* stream, F8Side side, UpVariant var = UpVariant{}) {
    if (!geom_ok(B, sd, sh, sw, Cs, ld, lh, lw, Cl, nd)) return CVAE_E_BADSHAPE;
    if (dtype != CVAE_F32 && dtype != CVAE_BF16) return CVAE_E_DTYPE;
    if (B == 0) return CVAE_OK;
    if (!L || !w || !S) return CVAE_E_NULLPTR;
    hipStream_t st = (hipStream_t)stream;
    if (Cl == 1) return cvae_conv_down_c1(L, dtype, (const float*)w, bias, mask, S, B, sd, sh, sw, Cs, ld, lh, lw, nd, dtype, act, st, side);
    if (Cl % 16 || Cs % 64) return CVAE_E_UNSUPPORTED;
    GEOM_INIT();
#if CVAE_KSPLIT_WAVES
    if (dtype == CVAE_BF16) return nd == 3 ? launch_data<bf16, 3, false, 1, 2, 4, 1, 2>(L, w, bias, mask, S, g, act, workspace, workspace_bytes, st, var, side)
                                           : launch_data<bf16, 2, false, 1, 2, 4, 1, 2>(L, w, bias, mask, S, g, act, workspace, workspace_bytes, st, var, side);
#endif
    if (dtype == CVAE_BF16) return nd == 3 ? launch_data<bf16, 3, false, 2, 2, 2, 1>(L, w, bias, mask, S, g, act, workspace, workspace_bytes, st, var, side)
                                           : launch_data<bf16, 2, false, 2, 2, 2, 1>(L, w, bias, mask, S, g, act, workspace, workspace_bytes, st, var, side);
    return nd == 3 ? launch_data<float, 3, false, 2, 2, 2, 1>(L, w, bias, mask, S, g, act, workspace, workspace_bytes, st, var, side)
                   : launch_data<float, 2, false, 2, 2, 2, 1>(L, w, bias, mask, S, g, act, workspace, workspace_bytes, st, var, side);
}

extern "C" int cvae_conv_down(const void* L, const void* w, const float* bias, const void* mask, void* S,
                              int64_t B, int64_t sd, int64_t sh, int64_t sw, int64_t Cs,
                              int64_t ld, int64_t lh, int64_t lw, int64_t Cl, int nd, int dtype, int act,
                              void* workspace, size_t workspace_bytes, void* stream) {
    return conv_down_impl(L, w, bias, mask, S, B, sd, sh, sw, Cs, ld, lh, lw, Cl, nd, dtype, act, workspace, workspace_bytes, stream, F8Side{nullptr, nullptr, nullptr});
}
// cvae_conv_down with the two-samples-per-tile form forced on (1) or off (0) for this call (the tests run every narrow case through both)
extern "C" int cvae_conv_down_variant(const void* L, const void* w, const float* bias, const void* mask, void* S,
                                      int64_t B, int64_t sd, int64_t sh, int64_t sw, int64_t Cs,
                                      int64_t ld, int64_t lh, int64_t lw, int64_t Cl, int nd, int dtype, int act,
                                      void* workspace, size_t workspace_bytes, int xpair, void* stream) {
    if (xpair < -1 || xpair > 1) return CVAE_E_BADSHAPE;
    UpVariant var;
    var.xpair = xpair;
    return conv_down_impl(L, w, bias, mask, S, B, sd, sh, sw, Cs, ld, lh, lw, Cl, nd, dtype, act, workspace, workspace_bytes, stream, F8Side{nullptr, nullptr, nullptr}, var);
}
// cvae_conv_down / cvae_conv_up with ReLU masks as BITS (F8Side, common.h): mask_bits replaces `mask` (1 bit per element of the result instead of the saved
// activation: 1/16 of the bytes the backward launch reads for it), relu_bits_out receives the mask of THIS launch's result for the backward pass to come.
static bool bits_ok(int64_t Cout) { return Cout % 32 == 0; }
extern "C" int cvae_conv_down_bits(const void* L, const void* w, const float* bias, const void* mask_bits, void* S, void* relu_bits_out,
                                   int64_t B, int64_t sd, int64_t sh, int64_t sw, int64_t Cs,
                                   int64_t ld, int64_t lh, int64_t lw, int64_t Cl, int nd, int dtype, int act,
                                   void* workspace, size_t workspace_bytes, void* stream) {
    if ((mask_bits || relu_bits_out) && !bits_ok(Cs)) return CVAE_E_UNSUPPORTED;
    if (Cl == 1 && mask_bits && !(dtype == CVAE_BF16)) return CVAE_E_UNSUPPORTED;
    F8Side side{nullptr, nullptr, nullptr};
    side.mask_bits = (const unsigned*)mask_bits; side.bits_out = (unsigned*)relu_bits_out;
    return conv_down_impl(L, w, bias, nullptr, S, B, sd, sh, sw, Cs, ld, lh, lw, Cl, nd, dtype, act, workspace, workspace_bytes, stream, side);
}
// ---- the single-channel image end of the network, image read in the dtype it is stored in (no cast pass in front of the first conv) ----
extern "C" int cvae_conv_image_supported(const void* L, int64_t lw, int l_dtype, int dtype) {
    if (l_dtype == dtype) return 1;
    if (!(dtype == CVAE_BF16 && l_dtype == CVAE_F32)) return 0;
    return lw % 4 == 0 && lw >= 4 && (((uintptr_t)L) & 15) == 0;
}
extern "C" int cvae_conv_down_image(const void* L, int l_dtype, const float* w, const float* bias, const void* mask, void* S,
                                    int64_t B, int64_t sd, int64_t sh, int64_t sw, int64_t Cs, int64_t ld, int64_t lh, int64_t lw, int nd, int dtype, int act, void* stream) {
    if (!geom_ok(B, sd, sh, sw, Cs, ld, lh, lw, 1, nd)) return CVAE_E_BADSHAPE;
    if ((dtype != CVAE_F32 && dtype != CVAE_BF16) || (l_dtype != CVAE_F32 && l_dtype != CVAE_BF16)) return CVAE_E_DTYPE;
    if (B == 0) return CVAE_OK;
    if (!L || !w || !S) return CVAE_E_NULLPTR;
    if (!cvae_conv_image_supported(L, lw, l_dtype, dtype)) return CVAE_E_UNSUPPORTED;
    return cvae_conv_down_c1(L, l_dtype, w, bias, mask, S, B, sd, sh, sw, Cs, ld, lh, lw, nd, dtype, act, (hipStream_t)stream);
}
extern "C" int cvae_conv_down_image_f8(const void* L, int l_dtype, const float* w, const float* bias, void* S, void* S8, const float* inv_scale_dev, void* amax_slots,
                                       void* relu_bits_out, int64_t B, int64_t sd, int64_t sh, int64_t sw, int64_t Cs, int64_t ld, int64_t lh, int64_t lw, int nd, int act,
                                       void* stream) {
    if (!geom_ok(B, sd, sh, sw, Cs, ld, lh, lw, 1, nd)) return CVAE_E_BADSHAPE;
    if (l_dtype != CVAE_F32 && l_dtype != CVAE_BF16) return CVAE_E_DTYPE;
    if (B == 0) return CVAE_OK;
    if (!L || !w || !S || (S8 && !inv_scale_dev)) return CVAE_E_NULLPTR;
    if (!cvae_conv_image_supported(L, lw, l_dtype, CVAE_BF16) && l_dtype != CVAE_BF16) return CVAE_E_UNSUPPORTED;
    F8Side side{inv_scale_dev, (fp8*)S8, (unsigned*)amax_slots};
    side.bits_out = (unsigned*)relu_bits_out;
    return cvae_conv_down_c1(L, l_dtype, w, bias, nullptr, S, B, sd, sh, sw, Cs, ld, lh, lw, nd, CVAE_BF16, act, (hipStream_t)stream, side);
}
extern "C" int cvae_conv_wgrad_image(const void* S, const void* L, int l_dtype, float* dW, float* dbias, void* workspace, size_t workspace_bytes,
                                     int64_t B, int64_t sd, int64_t sh, int64_t sw, int64_t Cs, int64_t ld, int64_t lh, int64_t lw, int nd, int dtype, void* stream) {
    if (!geom_ok(B, sd, sh, sw, Cs, ld, lh, lw, 1, nd)) return CVAE_E_BADSHAPE;
    if ((dtype != CVAE_F32 && dtype != CVAE_BF16) || (l_dtype != CVAE_F32 && l_dtype != CVAE_BF16)) return CVAE_E_DTYPE;
    if (!dW) return CVAE_E_NULLPTR;
    const int taps = (nd == 3) ? 64 : 16;
    if (B == 0) {
        if (dbias && hipMemsetAsync(dbias, 0, (size_t)Cs * sizeof(float), (hipStream_t)stream) != hipSuccess) return CVAE_E_LAUNCH;
        return hipMemsetAsync(dW, 0, (size_t)Cs * taps * sizeof(float), (hipStream_t)stream) == hipSuccess ? CVAE_OK : CVAE_E_LAUNCH;
    }
    if (!S || !L) return CVAE_E_NULLPTR;
    return cvae_conv_wgrad_c1(S, L, l_dtype, dW, dbias, nullptr, workspace, workspace_bytes, B, sd, sh, sw, Cs, ld, lh, lw, nd, dtype, (hipStream_t)stream);
}

static int conv_up_impl(const void* S, const void* w, const float* bias, const void* mask, void* L,
                        int64_t B, int64_t sd, int64_t sh, int64_t sw, int64_t Cs,
                        int64_t ld, int64_t lh, int64_t lw, int64_t Cl, int nd, int dtype, int act,
                        void* workspace, size_t workspace_bytes, void* stream, UpVariant var, F8Side side = F8Side{nullptr, nullptr, nullptr}) {
    if (!geom_ok(B, sd, sh, sw, Cs, ld, lh, lw, Cl, nd)) return CVAE_E_BADSHAPE;
    if (dtype != CVAE_F32 && dtype != CVAE_BF16) return CVAE_E_DTYPE;
    if (B == 0) return CVAE_OK;
    if (!S || !w || !L) return CVAE_E_NULLPTR;
    hipStream_t st = (hipStream_t)stream;
    if (Cl == 1) return cvae_conv_up_c1(S, (const float*)w, bias, mask, L, B, sd, sh, sw, Cs, ld, lh, lw, nd, dtype, act, st, var.walk_units);
    if (Cs % 16 || Cl % 32) return CVAE_E_UNSUPPORTED;
    GEOM_INIT();
    const bool wide = (Cl % 64) == 0;     // N tile 64 (2x2 waves, 128 rows) else N tile 32 (4x1 waves, 256 rows)
    if (dtype == CVAE_BF16) {
        int rc = CVAE_E_UNSUPPORTED;
        if (side.mask_bits || side.bits_out) {}      // the whole-K kernel knows the tensor form of the mask only (inference sweeps: no mask at all)
        else if (nd == 3) rc = wide ? try_up_full<bf16, 3, 2, 2, 2, 1>(S, w, bias, mask, L, g, act, st, var) : try_up_full<bf16, 3, 4, 1, 2, 1>(S, w, bias, mask, L, g, act, st, var);
        else rc = wide ? try_up_full<bf16, 2, 2, 2, 2, 1>(S, w, bias, mask, L, g, act, st, var) : try_up_full<bf16, 2, 4, 1, 2, 1>(S, w, bias, mask, L, g, act, st, var);
        if (rc != CVAE_E_UNSUPPORTED) return rc;
#if CVAE_KSPLIT_WAVES
        if (wide) return nd == 3 ? launch_data<bf16, 3, true, 1, 2, 4, 1, 2>(S, w, bias, mask, L, g, act, workspace, workspace_bytes, st, var, side)
                                 : launch_data<bf16, 2, true, 1, 2, 4, 1, 2>(S, w, bias, mask, L, g, act, workspace, workspace_bytes, st, var, side);
#endif
        if (nd == 3) return wide ? launch_data<bf16, 3, true, 2, 2, 2, 1>(S, w, bias, mask, L, g, act, workspace, workspace_bytes, st, var, side) : launch_data<bf16, 3, true, 4, 1, 2, 1>(S, w, bias, mask, L, g, act, workspace, workspace_bytes, st, var, side);
        return wide ? launch_data<bf16, 2, true, 2, 2, 2, 1>(S, w, bias, mask, L, g, act, workspace, workspace_bytes, st, var, side) : launch_data<bf16, 2, true, 4, 1, 2, 1>(S, w, bias, mask, L, g, act, workspace, workspace_bytes, st, var, side);
    }
    if (nd == 3) return wide ? launch_data<float, 3, true, 2, 2, 2, 1>(S, w, bias, mask, L, g, act, workspace, workspace_bytes, st, var, side) : launch_data<float, 3, true, 4, 1, 2, 1>(S, w, bias, mask, L, g, act, workspace, workspace_bytes, st, var, side);
    return wide ? launch_data<float, 2, true, 2, 2, 2, 1>(S, w, bias, mask, L, g, act, workspace, workspace_bytes, st, var, side) : launch_data<float, 2, true, 4, 1, 2, 1>(S, w, bias, mask, L, g, act, workspace, workspace_bytes, st, var, side);
}

extern "C" int cvae_conv_up(const void* S, const void* w, const float* bias, const void* mask, void* L,
                            int64_t B, int64_t sd, int64_t sh, int64_t sw, int64_t Cs,
                            int64_t ld, int64_t lh, int64_t lw, int64_t Cl, int nd, int dtype, int act,
                            void* workspace, size_t workspace_bytes, void* stream) {
    return conv_up_impl(S, w, bias, mask, L, B, sd, sh, sw, Cs, ld, lh, lw, Cl, nd, dtype, act, workspace, workspace_bytes, stream, UpVariant{});
}
extern "C" int cvae_conv_up_bits(const void* S, const void* w, const float* bias, const void* mask_bits, void* L, void* relu_bits_out,
                                 int64_t B, int64_t sd, int64_t sh, int64_t sw, int64_t Cs,
                                 int64_t ld, int64_t lh, int64_t lw, int64_t Cl, int nd, int dtype, int act,
                                 void* workspace, size_t workspace_bytes, void* stream) {
    if ((mask_bits || relu_bits_out) && (!bits_ok(Cl) || Cl == 1)) return CVAE_E_UNSUPPORTED;
    F8Side side{nullptr, nullptr, nullptr};
    side.mask_bits = (const unsigned*)mask_bits; side.bits_out = (unsigned*)relu_bits_out;
    return conv_up_impl(S, w, bias, nullptr, L, B, sd, sh, sw, Cs, ld, lh, lw, Cl, nd, dtype, act, workspace, workspace_bytes, stream, UpVariant{}, side);
}
extern "C" int cvae_conv_up_variant(const void* S, const void* w, const float* bias, const void* mask, void* L,
                                    int64_t B, int64_t sd, int64_t sh, int64_t sw, int64_t Cs,
                                    int64_t ld, int64_t lh, int64_t lw, int64_t Cl, int nd, int dtype, int act,
                                    void* workspace, size_t workspace_bytes, int upfull, int xpair, int64_t c1_walk_units, void* stream) {
    if (upfull < -1 || upfull > 1 || xpair < -1 || xpair > 1 || c1_walk_units < 0 || c1_walk_units >= ((int64_t)1 << 30)) return CVAE_E_BADSHAPE;
    UpVariant var;
    var.upfull = upfull; var.xpair = xpair; var.walk_units = c1_walk_units;
    return conv_up_impl(S, w, bias, mask, L, B, sd, sh, sw, Cs, ld, lh, lw, Cl, nd, dtype, act, workspace, workspace_bytes, stream, var);
}

extern "C" size_t cvae_conv_wgrad_workspace_bytes(int64_t Cs, int64_t Cl, int nd) {
    if (Cl == 1) return cvae_conv_wgrad_c1_workspace_bytes(Cs, nd);
    const int64_t groups = (Cs / 64) * (Cl / 32) * ((nd == 3) ? 4 : 1);             // slab groups (kd, channel block)
    const int64_t wgs = groups > WGRAD_MAX_WG ? groups : WGRAD_MAX_WG;             // groups * n_split <= max(WGRAD_MAX_WG, groups)
    const int64_t bias_floats = 2 * wgs * 64;                                      // bias partials: < 2 rows of 64 floats per workgroup
    return (size_t)(wgs * 32768 + bias_floats) * sizeof(float);
}

extern "C" int cvae_conv_wgrad(const void* S, const void* L, float* dW, float* dbias, int dbias_side, void* workspace, size_t workspace_bytes,
                               int64_t B, int64_t sd, int64_t sh, int64_t sw, int64_t Cs,
                               int64_t ld, int64_t lh, int64_t lw, int64_t Cl, int nd, int dtype, void* stream) {
    if (!geom_ok(B, sd, sh, sw, Cs, ld, lh, lw, Cl, nd)) return CVAE_E_BADSHAPE;
    if (dtype != CVAE_F32 && dtype != CVAE_BF16) return CVAE_E_DTYPE;
    if (!dW) return CVAE_E_NULLPTR;
    if (dbias_side != 0 && dbias_side != 1) return CVAE_E_BADSHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int taps = (nd == 3) ? 64 : 16;
    if (B == 0) {
        if (dbias && hipMemsetAsync(dbias, 0, (size_t)(dbias_side ? Cl : Cs) * sizeof(float), st) != hipSuccess) return CVAE_E_LAUNCH;
        return hipMemsetAsync(dW, 0, (size_t)Cs * Cl * taps * sizeof(float), st) == hipSuccess ? CVAE_OK : CVAE_E_LAUNCH;
    }
    if (!S || !L) return CVAE_E_NULLPTR;
    if (Cl == 1) {
        float* dbias_l = nullptr;
        if (dbias && dbias_side == 1) {                      // ConvTranspose to one channel: its bias gradient is the plain sum of L
            if (lh == 2 * sh && lw == 2 * sw && (nd != 3 || ld == 2 * sd)) dbias_l = dbias;     // fused: the kernel reads all of L anyway
            else {
                const int rcb = cvae_channel_sum(L, dbias, B * ld * lh * lw, 1, dtype, workspace, workspace_bytes, stream);   // scratch shared in stream order
                if (rcb != CVAE_OK) return rcb;
            }
            dbias = nullptr;
        }
        return cvae_conv_wgrad_c1(S, L, dtype, dW, dbias, dbias_l, workspace, workspace_bytes, B, sd, sh, sw, Cs, ld, lh, lw, nd, dtype, st);   // S-side bias sum fused (S^T . ones)
    }
    if (Cs % 64 || Cl % 32) return CVAE_E_UNSUPPORTED;
    int bias_mode = dbias ? (dbias_side ? 2 : 1) : 0;
    if (bias_mode == 2 && (lh != 2 * sh || lw != 2 * sw || (nd == 3 && ld != 2 * sd))) {
        // an odd L extent leaves a plane / row outside every tile's non-halo part: sum L separately
        const int rcb = cvae_channel_sum(L, dbias, B * ld * lh * lw, Cl, dtype, workspace, workspace_bytes, stream);   // scratch shared in stream order
        if (rcb != CVAE_OK) return rcb;
        bias_mode = 0;
    }
    const size_t need = cvae_conv_wgrad_workspace_bytes(Cs, Cl, nd);
    if (!workspace) return CVAE_E_NULLPTR;
    if (workspace_bytes < need) return CVAE_E_WORKSPACE;
    GEOM_INIT();
    float* db = bias_mode ? dbias : nullptr;
    if (dtype == CVAE_BF16)
        return nd == 3 ? launch_wgrad<bf16, 3>(S, L, (float*)workspace, dW, db, bias_mode, g, st) : launch_wgrad<bf16, 2>(S, L, (float*)workspace, dW, db, bias_mode, g, st);
    return nd == 3 ? launch_wgrad<float, 3>(S, L, (float*)workspace, dW, db, bias_mode, g, st) : launch_wgrad<float, 2>(S, L, (float*)workspace, dW, db, bias_mode, g, st);
}

template <typename T, int ND>
static int wgrad_multi_t(int count, const void* const* S, const void* const* L, float* const* dW, float* const* dbias, const int* dbias_side, void* const* workspace,
                         const int64_t* dims, hipStream_t st) {
    WgradTable mt;
    WgradReduceTable rt;
    int mb = 0, rb = 0;
    // Together the layers need ~2-3 workgroups per CU, not 2 each: every workgroup ends with a 128 KB slab, and 6 x 512 slabs (384 MB) no longer fit
    // the 256 MB Infinity Cache between the main pass and the reduction (measured: the grouped reduction 77 us against 64 us for six separate
    // ones).  The slab count of a layer depends on THAT layer only — ~CVAE_WG_TILES tiles of 128 positions per slab, at least CVAE_WG_MIN_WG
    // workgroups — never on what else rides in the launch: the split-backward capture flushes decoder and encoder separately, and its
    // gradients must have the summation order (the bits) of the single-launch step.
    using TLm = Tile<ND, 128>;
    // longest workgroups first (tiles per slab, descending): the launch is ~2.5 rounds of workgroups, and a long one started last is the tail
    long long req[WG_MULTI_MAX], key[WG_MULTI_MAX];
    int order[WG_MULTI_MAX];
    for (int i = 0; i < count; ++i) {
        const int64_t* d = dims + 9 * i;
        const long long tiles = d[0] * ((d[1] + TLm::TD - 1) / TLm::TD) * ((d[2] + TLm::TH - 1) / TLm::TH) * ((d[3] + TLm::TW - 1) / TLm::TW);
        const long long groups = (d[4] / 64) * (d[8] / 32) * ((ND == 3) ? 4 : 1);            // workgroups per slab index
        req[i] = (tiles + CVAE_WG_TILES - 1) / CVAE_WG_TILES;
        if (req[i] * groups < CVAE_WG_MIN_WG) req[i] = (CVAE_WG_MIN_WG + groups - 1) / groups;
        if (req[i] > tiles) req[i] = tiles;
        key[i] = (tiles + req[i] - 1) / req[i];
        int k = i;
        while (k > 0 && key[order[k - 1]] < key[i]) { order[k] = order[k - 1]; --k; }
        order[k] = i;
    }
    for (int k = 0; k < count; ++k) {
        const int i = order[k];
        const int64_t* d = dims + 9 * i;
        ConvGeom g{(int)d[0], (int)d[1], (int)d[2], (int)d[3], (int)d[4], (int)d[5], (int)d[6], (int)d[7], (int)d[8], 0, 0, 0};
        const int bias_mode = dbias[i] ? (dbias_side[i] ? 2 : 1) : 0;
        int m1, r1;
        const int rc = plan_wgrad<T, ND>(S[i], L[i], (float*)workspace[i], dW[i], dbias[i], bias_mode, g, &mt.e[k], &rt.e[k], &m1, &r1, req[i]);
        if (rc != CVAE_OK) return rc;
        mt.blk_start[k] = mb; rt.blk_start[k] = rb;
        mb += m1; rb += r1;
    }
    mt.blk_start[count] = mb; rt.blk_start[count] = rb;
    mt.count = rt.count = count;
    return launch_wgrad_tables<T, ND>(mt, rt, st);
}

extern "C" int cvae_conv_wgrad_multi(int count, const void* const* S, const void* const* L, float* const* dW, float* const* dbias, const int* dbias_side,
                                     void* const* workspace, const size_t* workspace_bytes, const int64_t* dims, int nd, int dtype, void* stream) {
    if (count < 0 || count > WG_MULTI_MAX || (nd != 2 && nd != 3)) return CVAE_E_BADSHAPE;
    if (count == 0) return CVAE_OK;
    if (dtype != CVAE_F32 && dtype != CVAE_BF16) return CVAE_E_DTYPE;
    if (!S || !L || !dW || !dbias || !dbias_side || !workspace || !workspace_bytes || !dims) return CVAE_E_NULLPTR;
    for (int i = 0; i < count; ++i) {
        const int64_t* d = dims + 9 * i;
        const int64_t B = d[0], sd = d[1], sh = d[2], sw = d[3], Cs = d[4], ld = d[5], lh = d[6], lw = d[7], Cl = d[8];
        if (!geom_ok(B, sd, sh, sw, Cs, ld, lh, lw, Cl, nd) || B == 0) return CVAE_E_BADSHAPE;
        if (Cl == 1 || Cs % 64 || Cl % 32) return CVAE_E_UNSUPPORTED;                       // the single-channel layers have their own kernels
        if (dbias_side[i] != 0 && dbias_side[i] != 1) return CVAE_E_BADSHAPE;
        if (dbias[i] && dbias_side[i] == 1 && (lh != 2 * sh || lw != 2 * sw || (nd == 3 && ld != 2 * sd))) return CVAE_E_UNSUPPORTED;   // odd L extent: cvae_conv_wgrad sums L separately
        if (!S[i] || !L[i] || !dW[i] || !workspace[i]) return CVAE_E_NULLPTR;
        if (workspace_bytes[i] < cvae_conv_wgrad_workspace_bytes(Cs, Cl, nd)) return CVAE_E_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    if (dtype == CVAE_BF16)
        return nd == 3 ? wgrad_multi_t<bf16, 3>(count, S, L, dW, dbias, dbias_side, workspace, dims, st) : wgrad_multi_t<bf16, 2>(count, S, L, dW, dbias, dbias_side, workspace, dims, st);
    return nd == 3 ? wgrad_multi_t<float, 3>(count, S, L, dW, dbias, dbias_side, workspace, dims, st) : wgrad_multi_t<float, 2>(count, S, L, dW, dbias, dbias_side, workspace, dims, st);
}

// ---------------------------------------------------------------------------------------------- fp8 (e4m3) products
// BASELINE.json configs[4].  Conv / ConvTranspose products with C_in >= 32 and C_out > 1 on fp8 operands with per-tensor scales and fp32
// accumulation, on the block-scaled CDNA4 MFMA v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales (K = 64 per instruction, twice the
// bf16 FLOPs per clock).  conv_data_kernel<f8x2, ..>: the bf16 kernel's tiles, staging and epilogue on half the operand bytes (see Frag<f8x2>).
//   * inference (the counterfactual decode sweep): static scales from a calibration pass, fp8 codes travel between fp8 layers;
//   * training forward (the step of causal_cascade/train.py:19-39 with fp8 conv inputs): scales live on the device and follow the tensors with
//     one step of delay (cvae_fp8_scale_update), every producer leaves its result twice — bf16 for the backward pass, fp8 for the next layer —
//     and records its amax; the backward pass runs the bf16 kernels on the bf16 copies.
__global__ void quantize_fp8_kernel(const void* __restrict__ src, int src_dtype, fp8* __restrict__ dst, int64_t n, float inv_scale, const float* __restrict__ inv_scale_dev,
                                    unsigned* __restrict__ amax) {
    const float mul = inv_scale_dev ? inv_scale_dev[0] : inv_scale;
    float amx = 0.f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = src_dtype == CVAE_BF16 ? to_f32(((const bf16*)src)[i]) : ((const float*)src)[i];
        amx = fmaxf(amx, fabsf(v));
        dst[i] = from_f32<fp8>(v * mul);
    }
    __shared__ float red[4];
    if (amax) amax_publish_wg(amax, amx, blockIdx.x, red);
}
extern "C" int cvae_quantize_fp8(const void* src, int src_dtype, void* dst, int64_t n, float inv_scale, void* stream) {
    if (n < 0 || !(inv_scale > 0.f)) return CVAE_E_BADSHAPE;
    if (src_dtype != CVAE_F32 && src_dtype != CVAE_BF16) return CVAE_E_DTYPE;
    if (n == 0) return CVAE_OK;
    if (!src || !dst) return CVAE_E_NULLPTR;
    hipLaunchKernelGGL(quantize_fp8_kernel, dim3(cvae_grid_1d(n, 256)), dim3(256), 0, (hipStream_t)stream, src, src_dtype, (fp8*)dst, n, inv_scale, (const float*)nullptr, (unsigned*)nullptr);
    CVAE_CHECK_LAUNCH();
    return CVAE_OK;
}
extern "C" int cvae_quantize_fp8_dev(const void* src, int src_dtype, void* dst, int64_t n, const float* inv_scale_dev, void* amax_slots, void* stream) {
    if (n < 0) return CVAE_E_BADSHAPE;
    if (src_dtype != CVAE_F32 && src_dtype != CVAE_BF16) return CVAE_E_DTYPE;
    if (n == 0) return CVAE_OK;
    if (!src || !dst || !inv_scale_dev) return CVAE_E_NULLPTR;
    hipLaunchKernelGGL(quantize_fp8_kernel, dim3(cvae_grid_1d(n, 256)), dim3(256), 0, (hipStream_t)stream, src, src_dtype, (fp8*)dst, n, 1.f, inv_scale_dev, (unsigned*)amax_slots);
    CVAE_CHECK_LAUNCH();
    return CVAE_OK;
}
extern "C" int cvae_absmax(const void* src, int dtype, int64_t n, void* amax_slots, void* stream) {
    if (n < 0) return CVAE_E_BADSHAPE;
    if (dtype != CVAE_F32 && dtype != CVAE_BF16) return CVAE_E_DTYPE;
    if (n == 0) return CVAE_OK;
    if (!src || !amax_slots) return CVAE_E_NULLPTR;
    hipLaunchKernelGGL(absmax_kernel, dim3(cvae_grid_1d(n, 256)), dim3(256), 0, (hipStream_t)stream, src, dtype, n, (unsigned*)amax_slots);
    CVAE_CHECK_LAUNCH();
    return CVAE_OK;
}
extern "C" int cvae_conv_pack_weight_fp8(const float* w, void* packed, int64_t Cs, int64_t Cl, int nd, int for_up, float inv_scale, void* stream) {
    if ((nd != 2 && nd != 3) || Cs <= 0 || Cl <= 0 || !(inv_scale > 0.f)) return CVAE_E_BADSHAPE;
    if (!fp8_pack_ok(Cs, Cl, for_up)) return CVAE_E_UNSUPPORTED;      // what cvae_conv_fp8 accepts
    if (!w || !packed) return CVAE_E_NULLPTR;
    const int taps = (nd == 3) ? 64 : 16;
    const int64_t n = Cs * Cl * taps;
    hipLaunchKernelGGL((pack_weight_kernel<fp8, 32>), dim3(cvae_grid_1d(n, 256)), dim3(256), 0, (hipStream_t)stream, w, (fp8*)packed, (int)Cs, (int)Cl, taps, for_up, inv_scale);
    CVAE_CHECK_LAUNCH();
    return CVAE_OK;
}
extern "C" int cvae_conv_pack_weights_fp8(const float* const* w, void* const* packed, const int64_t* Cs, const int64_t* Cl, const int* for_up,
                                           const float* const* inv_scale_dev, void* const* amax_slots, int count, int nd, void* stream) {
    if ((nd != 2 && nd != 3) || count < 0 || count > F8PACK_MAX) return CVAE_E_BADSHAPE;
    if (count == 0) return CVAE_OK;
    if (!w || !packed || !Cs || !Cl || !for_up || !inv_scale_dev) return CVAE_E_NULLPTR;
    F8PackTable tb;
    tb.taps = (nd == 3) ? 64 : 16;
    int blocks = 0;
    for (int i = 0; i < count; ++i) {
        if (Cs[i] <= 0 || Cl[i] <= 0) return CVAE_E_BADSHAPE;
        if (!fp8_pack_ok(Cs[i], Cl[i], for_up[i])) return CVAE_E_UNSUPPORTED;
        if (!w[i] || !packed[i] || !inv_scale_dev[i]) return CVAE_E_NULLPTR;
        tb.w[i] = w[i]; tb.out[i] = (fp8*)packed[i]; tb.inv_scale[i] = inv_scale_dev[i]; tb.amax[i] = amax_slots ? (unsigned*)amax_slots[i] : nullptr;
        tb.Cs[i] = (int)Cs[i]; tb.Cl[i] = (int)Cl[i]; tb.for_up[i] = for_up[i];
        tb.blk_start[i] = blocks;
        int64_t nb = (Cs[i] * Cl[i] * tb.taps + 4095) / 4096;
        if (nb > 1024) nb = 1024;
        blocks += (int)nb;
    }
    tb.blk_start[count] = blocks;
    tb.count = count;
    hipLaunchKernelGGL(pack_weight_fp8_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, tb);
    CVAE_CHECK_LAUNCH();
    return CVAE_OK;
}
extern "C" int cvae_fp8_scale_update(void* amax_slots, float* scale, float* inv_scale, int n, float headroom, const int* layer_in, const int* layer_w, const int* layer_out,
                                     int n_layers, float* dscale, void* ticket, void* stream) {
    if (n < 0 || n > 4096 || n_layers < 0 || n_layers > F8LAYER_MAX || !(headroom > 0.f)) return CVAE_E_BADSHAPE;
    if (n == 0) return CVAE_OK;
    if (!amax_slots || !scale || !inv_scale || !ticket || (n_layers && (!layer_in || !layer_w || !layer_out || !dscale))) return CVAE_E_NULLPTR;
    F8LayerIdx li;
    li.count = n_layers;
    for (int l = 0; l < n_layers; ++l) {
        if (layer_in[l] < 0 || layer_in[l] >= n || layer_w[l] < 0 || layer_w[l] >= n || layer_out[l] >= n) return CVAE_E_BADSHAPE;
        li.in[l] = layer_in[l]; li.w[l] = layer_w[l]; li.out[l] = layer_out[l];
    }
    hipLaunchKernelGGL(fp8_scale_update_kernel, dim3(n), dim3(1024), 0, (hipStream_t)stream, (unsigned*)amax_slots, scale, inv_scale, n, headroom, li, dscale, (unsigned*)ticket);
    CVAE_CHECK_LAUNCH();
    return CVAE_OK;
}

// One fp8 product.  up = 0: S = conv(L) ("down"), up = 1: L = convT(S).  out_dtype CVAE_BF16 (bf16 result; `out8` may ask for a second copy as fp8
// codes) or CVAE_FP8 (codes only).  Scales: by value (acc_scale = s_in * s_w, out8_inv_scale = 1 / s_out8), or, when `dscale` is not null,
// read from the device pair {acc_scale, out8_inv_scale} at run time.
template <int ND, bool UP, typename TO>
static int conv_fp8_t(const void* in, const void* w, const float* bias, void* out, ConvGeom g, int act, float acc_scale, float out_scale, F8Side f8, void* ws, size_t wsb, hipStream_t st,
                      UpVariant var) {
    const int Cout = UP ? g.Cl : g.Cs;
    const bool wide = (Cout % 64) == 0;
    // the bf16 launches' tile shapes: 64-channel tiles as (K split) x (N sub-tile) waves with the per-wave weight fetch, 32-channel tiles as 4 x 1 waves on LDS panels
#ifndef CVAE_F8_FORM
#define CVAE_F8_FORM 0      // 64-channel tiles: 0 = (K split) x (N sub-tile) waves with the per-wave weight fetch (the bf16 form), 1 = 2 x 2 waves on LDS weight panels,
#endif                      // 2 = 2 x 2 waves with the per-wave fetch
#define F8L(WM, WN, MI, TS, BDX, EPI) launch_data_epi<f8x2, ND, UP, WM, WN, MI, 1, EPI, 1, TO, BDX, TS>(in, w, bias, nullptr, out, g, act, ws, wsb, st, acc_scale, out_scale, f8, var)
#define F8E(WM, WN, MI, TS, BDX) (act == CVAE_ACT_NONE ? F8L(WM, WN, MI, TS, BDX, 0) : (act == CVAE_ACT_RELU ? F8L(WM, WN, MI, TS, BDX, 1) : F8L(WM, WN, MI, TS, BDX, 2)))
    if (wide) {
        // measured (rocprofv3 device durations, 4 x 128^3 shapes, profiles/r03_fp8_forms.txt): the `down` products want their weights on LDS panels — the
        // per-wave fetch form needs > 256 VGPRs with 8-register fp8 operands (59 spilled; enc2 57 us against 36 us) — the `up` products keep the bf16
        // launch's (K split) x (N sub-tile) waves with the per-wave fetch (dec1 9.9 vs 11.2 us, dec2 8.1 vs 8.4 us)
        if constexpr (UP && ND == 3 && CVAE_F8_FORM == 0) return F8E(1, 2, 4, 2, true);
        else return F8E(2, 2, 2, 1, (CVAE_F8_FORM == 2));
    }
    if constexpr (UP) return F8E(4, 1, 2, 1, false);
    else return CVAE_E_UNSUPPORTED;
#undef F8E
#undef F8L
}
extern "C" int cvae_conv_fp8(int up, const void* in8, const void* w8, const float* bias, void* out, int out_dtype, void* out8, const float* dscale, float acc_scale,
                             float out8_inv_scale, void* amax_slots, int64_t B, int64_t sd, int64_t sh, int64_t sw, int64_t Cs, int64_t ld, int64_t lh, int64_t lw, int64_t Cl,
                             int nd, int act, void* workspace, size_t workspace_bytes, int xpair, void* relu_bits_out, void* stream) {
    if (!geom_ok(B, sd, sh, sw, Cs, ld, lh, lw, Cl, nd) || (up != 0 && up != 1) || xpair < -1 || xpair > 1) return CVAE_E_BADSHAPE;
    if (out_dtype != CVAE_FP8 && out_dtype != CVAE_BF16) return CVAE_E_DTYPE;
    if (!dscale && (!(acc_scale > 0.f) || ((out_dtype == CVAE_FP8 || out8) && !(out8_inv_scale > 0.f)))) return CVAE_E_BADSHAPE;
    if (out_dtype == CVAE_FP8 && out8) return CVAE_E_BADSHAPE;        // codes-only output: there is no second copy to ask for
    if (B == 0) return CVAE_OK;
    if (!in8 || !w8 || !out) return CVAE_E_NULLPTR;
    if (!fp8_pack_ok(Cs, Cl, up)) return CVAE_E_UNSUPPORTED;
    if (up ? (lh != 2 * sh || lw != 2 * sw || (nd == 3 && ld != 2 * sd)) : false) return CVAE_E_UNSUPPORTED;   // forward products only: exact 2x extents
    GEOM_INIT();
    hipStream_t st = (hipStream_t)stream;
    F8Side f8{dscale, (fp8*)out8, (unsigned*)amax_slots};
    f8.bits_out = (unsigned*)relu_bits_out;
    UpVariant var;
    var.xpair = xpair;
#define F8D(ND_, UP_) (out_dtype == CVAE_FP8 ? conv_fp8_t<ND_, UP_, fp8>(in8, w8, bias, out, g, act, acc_scale, out8_inv_scale, f8, nullptr, 0, st, var) \
                                             : conv_fp8_t<ND_, UP_, bf16>(in8, w8, bias, out, g, act, acc_scale, out8_inv_scale, f8, workspace, workspace_bytes, st, var))
    if (nd == 3) return up ? F8D(3, true) : F8D(3, false);
    return up ? F8D(2, true) : F8D(2, false);
#undef F8D
}
// the inference entry point of round 2, kept: static scales by value, result as bf16 or as codes
extern "C" int cvae_conv_up_fp8(const void* S, const void* w, const float* bias, void* L, int out_dtype, float acc_scale, float out_inv_scale,
                                int64_t B, int64_t sd, int64_t sh, int64_t sw, int64_t Cs, int64_t ld, int64_t lh, int64_t lw, int64_t Cl, int nd, int act,
                                void* stream) {
    return cvae_conv_fp8(1, S, w, bias, L, out_dtype, nullptr, nullptr, acc_scale, out_dtype == CVAE_FP8 ? out_inv_scale : 1.f, nullptr, B, sd, sh, sw, Cs, ld, lh, lw, Cl, nd, act,
                         nullptr, 0, -1, nullptr, stream);
}

extern "C" int cvae_conv_up_c1_fp8in(const void* S8, const float* w, const float* bias, void* L, float in_scale, int64_t B, int64_t sd, int64_t sh, int64_t sw, int64_t Cs, int nd,
                                     int act, void* stream) {
    if (!S8 || !w || !L) return CVAE_E_NULLPTR;
    if (nd != 3 || B < 1 || sd < 1 || sh < 1 || sw < 1) return CVAE_E_UNSUPPORTED;
    if (act < CVAE_ACT_NONE || act > CVAE_ACT_LEAKY02) return CVAE_E_BADSHAPE;
    return cvae_conv_up_c1_fp8in_impl(S8, w, bias, L, in_scale, B, sd, sh, sw, Cs, act, (hipStream_t)stream);
}
