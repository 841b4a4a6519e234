// MFMA tile engine for gfx950 (CDNA4): one templated kernel, pluggable operand loaders and
// epilogues.  Every dense contraction of the iS-DQN step (implicit-GEMM convolutions,
// dense layers, their data- and weight-gradients) is an instance of
//
//      C[m][n] = sum_k A(m, k) * B(n, k)
//
// with fp32 data in HBM, bf16 MFMA (v_mfma_f32_16x16x32_bf16) and fp32 accumulation.
//
// Operand staging: global -> registers (8-element chunks fetched by a problem-specific
// loader functor: im2col gathers, uint8 frame decode, ...) -> bf16 split -> LDS, double
// buffered, one barrier per 32-deep K step.  An operand is staged in one of two images:
//   ROW : LDS tile [rows][32 k]  (k contiguous)  -> fragments by ds_read_b128
//   TR  : LDS tile [32 k][rows]  (rows contiguous: the contraction index is the slow
//         one in memory, as in every weight-gradient) -> fragments by the gfx950
//         transposing read ds_read_b64_tr_b16, so no operand is ever transposed in HBM.
//
// Precision (PASSES): 1 = plain bf16; 3 = split-bf16 (x = hi + lo, both bf16):
//   hi*hi + lo*hi + hi*lo  -> ~2^-17 relative, fp32-class results at 3/16 of the fp32
//   MFMA cost instead of 16/16;  2 = same with an operand B that is exact in bf16
//   (uint8 pixels), so its lo plane is skipped.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include "common.h"

namespace isdqn {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int GEMM_BK = 32;
constexpr int GEMM_THREADS = 256;

// Workgroup ids are dealt round-robin over the 8 XCDs (each with its own 4 MiB L2).  This bijective remap gives
// every XCD a CONTIGUOUS range of logical tile ids, so tiles that share an operand panel (consecutive logical
// ids) hit the same L2 instead of each re-fetching the panel from the Infinity Cache.  Speed only.
__device__ __forceinline__ int xcd_remap(int bid, int n_blocks) {
    const int xcd = bid & 7, q = n_blocks >> 3, r = n_blocks & 7;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

#define LDS_AS __attribute__((address_space(3)))

__device__ __forceinline__ void split8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        __bf16 h = (__bf16)v[i];
        hi[i] = h;
        lo[i] = (__bf16)(v[i] - (float)h);
    }
}
__device__ __forceinline__ void round8(const float (&v)[8], bf16x8& hi) {
#pragma unroll
    for (int i = 0; i < 8; ++i) hi[i] = (__bf16)v[i];
}

// "S8" split storage (round 2): a tensor whose MFMA consumers need it as bf16 hi + lo is stored pre-split by its
// producer.  Every aligned group of 8 consecutive elements keeps its 32 bytes, but they hold the 8 hi halves (16 B)
// followed by the 8 lo halves (16 B) instead of 8 fp32 values: addresses, chunk loads (two dwordx4) and footprints are
// those of the fp32 tensor, and a consumer's staging is a plain copy -- no conversion VALU in any K loop or image fill.
// hi + lo is exactly what split8() of the fp32 value gives, so results are bit-identical to splitting at the consumer.
// Used for the weight mirror (written by the optimizer), the activations and the pre-activation gradients.
__device__ __forceinline__ void s8_unpack(const float (&raw)[8], bf16x8& hi, bf16x8& lo) {
    const f32x4 h = {raw[0], raw[1], raw[2], raw[3]}, l = {raw[4], raw[5], raw[6], raw[7]};
    hi = __builtin_bit_cast(bf16x8, h);
    lo = __builtin_bit_cast(bf16x8, l);
}
__device__ __forceinline__ void s8_unpack_hi(const float (&raw)[8], bf16x8& hi) {
    const f32x4 h = {raw[0], raw[1], raw[2], raw[3]};
    hi = __builtin_bit_cast(bf16x8, h);
}
// staging of one 8-element chunk into the hi (+ lo) planes, from either storage
template <bool S8>
__device__ __forceinline__ void chunk_planes(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
    if constexpr (S8) s8_unpack(v, hi, lo);
    else split8(v, hi, lo);
}
template <bool S8>
__device__ __forceinline__ void chunk_hi(const float (&v)[8], bf16x8& hi) {
    if constexpr (S8) s8_unpack_hi(v, hi);
    else round8(v, hi);
}
// one whole group from 8 fp32 values (group = float pointer to the 32-byte group)
__device__ __forceinline__ void s8_store_group(float* group, const float (&v)[8]) {
    bf16x8 hi, lo;
    split8(v, hi, lo);
    reinterpret_cast<bf16x8*>(group)[0] = hi;
    reinterpret_cast<bf16x8*>(group)[1] = lo;
}
// four consecutive elements c0 .. c0+3 (c0 % 4 == 0) of the row starting at `row`: one half of a group, two 8-byte stores
__device__ __forceinline__ void s8_store_quad(float* row, int c0, float v0, float v1, float v2, float v3) {
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
    const float v[4] = {v0, v1, v2, v3};
    bf16x4 hi, lo;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const __bf16 h = (__bf16)v[i];
        hi[i] = h;
        lo[i] = (__bf16)(v[i] - (float)h);
    }
    char* g = reinterpret_cast<char*>(row + (c0 & ~7)) + (c0 & 4) * 2;
    *reinterpret_cast<bf16x4*>(g) = hi;
    *reinterpret_cast<bf16x4*>(g + 16) = lo;
}

// The MFMA accumulator layout gives lane rows 2q and 2q+1 (lanes 16 apart) the two halves c0 = 8g and 8g+4 of one group.
// Two v_permlane16_swap exchange them so that the even row holds the group's 8 hi halves and the odd row its 8 lo halves:
// every lane then stores 16 contiguous bytes and a wave store covers whole 32-byte groups, like the fp32 float4 store
// did (two 8-byte stores per lane leave 16-byte holes in every store instruction: measured 2.5 % of the step).
// Must be executed by both lanes of a pair (same pixel, channels c0 and c0 ^ 4).
__device__ __forceinline__ void s8_store_quad_paired(float* row, int c0, float v0, float v1, float v2, float v3) {
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    const float v[4] = {v0, v1, v2, v3};
    bf16x4 hi, lo;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const __bf16 h = (__bf16)v[i];
        hi[i] = h;
        lo[i] = (__bf16)(v[i] - (float)h);
    }
    const u32x2 h = __builtin_bit_cast(u32x2, hi), l = __builtin_bit_cast(u32x2, lo);
    const auto r0 = __builtin_amdgcn_permlane16_swap(h[0], l[0], false, false);
    const auto r1 = __builtin_amdgcn_permlane16_swap(h[1], l[1], false, false);
    char* g = reinterpret_cast<char*>(row + (c0 & ~7)) + (c0 & 4) * 4;  // even row: the hi half, odd row: the lo half
    *reinterpret_cast<u32x4*>(g) = u32x4{r0[0], r1[0], r0[1], r1[1]};
}

// two adjacent elements c0, c0+1 (c0 even) / one element of the row starting at `row`
__device__ __forceinline__ void s8_store_pair(float* row, int c0, float v0, float v1) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    const __bf16 h0 = (__bf16)v0, h1 = (__bf16)v1;
    char* g = reinterpret_cast<char*>(row + (c0 & ~7)) + (c0 & 7) * 2;
    *reinterpret_cast<bf16x2*>(g) = bf16x2{h0, h1};
    *reinterpret_cast<bf16x2*>(g + 16) = bf16x2{(__bf16)(v0 - (float)h0), (__bf16)(v1 - (float)h1)};
}
__device__ __forceinline__ void s8_store_elem(float* row, int c, float v) {
    const __bf16 h = (__bf16)v;
    __bf16* g = reinterpret_cast<__bf16*>(row + (c & ~7)) + (c & 7);
    g[0] = h;
    g[8] = (__bf16)(v - (float)h);
}

// LDS image geometry (in bf16 elements)
//   ROW images [rows][32 k]: pitch 48 elements = 96 bytes, (pitch / 16) = 2 (mod 4): the 16-lane groups of ds_read_b128 hit
//     16 distinct bank quads (conv_img.h ConvImgTraits); the 80-byte pitch of round 1 gave 2-way conflicts on every read.
//   TR images [32 k][rows]: a ds_read_b64_tr_b16 is served in two 32-lane groups; the lanes of one group address four k rows
//     of two 8-row k groups, 8 bytes (2 banks) each at the same 4 column offsets.  With the k rows stored in natural order the
//     two k groups sit 8 pitches apart and hit the same banks whatever the pitch (2-way, scripts/lds_conflicts.py).  The rows
//     are therefore stored in the order tr_row() -- the eight rows one group reads become neighbours -- with a pitch of
//     ROWS + 16 elements (an odd multiple of 32 bytes): conflict-free.  TR_PAD = 8 keeps round 1's pitch (still 2-way) for
//     the one kernel whose four workgroups per CU would not fit otherwise.
//   Stash writes (ds_write_b128: 8 contiguous lanes per LDS cycle, 32 four-byte banks): with the 96-byte pitch the chunks of
//     rows r and r + 1 overlap in 8 banks, those of rows r and r + 2 do not -- stash_row() hands the lane quads of a ROW image
//     the rows of every block of four in the order 0, 2, 1, 3 (row counts are multiples of 4).
__device__ __forceinline__ constexpr int stash_row(int chunk) {  // row of 8-element chunk `chunk` (4 chunks per row)
    return ((chunk >> 4) << 2) + (((chunk >> 2) & 1) << 1) + ((chunk >> 3) & 1);
}
__device__ __forceinline__ constexpr int tr_row(int kk) {  // LDS row of k row kk (kk < 32) of a TR image
    return 16 * (kk >> 4) + 8 * ((kk >> 2) & 1) + 2 * (kk & 3) + ((kk >> 3) & 1);
}
template <int ROWS, bool TR, int TR_PAD = 16>
struct TileGeom {
    static constexpr int PITCH = TR ? (ROWS + TR_PAD) : (GEMM_BK + 16);
    static constexpr int ELEMS = TR ? GEMM_BK * PITCH : ROWS * PITCH;
    static constexpr int CHUNKS = ROWS * (GEMM_BK / 8);  // 8-element chunks per K step (same for both images)
    static constexpr int PER_THREAD = (CHUNKS + GEMM_THREADS - 1) / GEMM_THREADS;
};

// acc += A * B on the matrix pipe (v_mfma_f32_16x16x32_bf16), through the BUILTIN, so that hipcc's hazard recogniser
// sees every MFMA and pads its dependencies itself (VALU / v_accvgpr_write -> MFMA operand: 2 wait states; MFMA result ->
// any other reader: passes + 4; scripts/isa_lint.py re-checks that table on the final assembly).  Round 1 issued the
// MFMAs through inline asm with hand-placed s_nops because "tiles lacking one MFMA pass" were blamed on compiler-renamed
// accumulators; round 2 traced the only reproducible instability to SLP-vectorised packed-fp32 epilogue code instead
// (the library is built with -fno-slp-vectorize), and the builtin form is bit-stable in the same
// hunts (0 differing steps in 3000 + 300 + 300) -- so the asm, its pads and the "two accumulator sets" rule are gone.
__device__ __forceinline__ void mfma_init(f32x4& acc) { acc = f32x4{0.f, 0.f, 0.f, 0.f}; }
__device__ __forceinline__ void mfma_acc(f32x4& acc, const bf16x8& a, const bf16x8& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
}

// Fragment of 16 rows x 32 k for lane `lane`: element j <-> (row = row0 + (lane&15), k = 8*(lane>>4) + j).
template <bool TR, int PITCH>
__device__ __forceinline__ bf16x8 read_frag(const __bf16* tile, int row0, int lane) {
    if constexpr (!TR) {
        return *reinterpret_cast<const bf16x8*>(tile + (row0 + (lane & 15)) * PITCH + (lane >> 4) * 8);
    } else {
        // ds_read_b64_tr_b16: per 16-lane group a 4(k) x 16(rows) block; lane 4q+p supplies the address of
        // k-row q, rows 4p..4p+3; lane i receives row i of the 4 k-rows.
        const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
        const __bf16* a0 = tile + (16 * (g >> 1) + 2 * q + (g & 1)) * PITCH + row0 + 4 * p;  // tr_row(8 * g + q)
        const __bf16* a1 = a0 + 8 * PITCH;                                                   // tr_row(8 * g + q + 4)
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)a0);
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)a1);
        s16x8 r = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(bf16x8, r);
    }
}

// A "problem" P supplies:
//   static constexpr int BM, BN, WM, WN;   block tile and wave grid (WM*WN == 4)
//   static constexpr bool A_TR, B_TR;      LDS image per operand
//   static constexpr int PASSES;           1, 2 or 3
//   struct Tile { int m0, n0, k0, k1; ... }            k range in elements, k0 % 32 == 0
//   __device__ bool tile(int block, Tile&) const
//   ACtx / BCtx + a_ctx(tile, fixed) / b_ctx(tile, fixed)
//        ROW image: fixed = absolute row (m0 + r),      varying = absolute k of the chunk (8 along k)
//        TR  image: fixed = absolute first row of the chunk (8 along rows), varying = absolute k
//   __device__ void load_a(const Tile&, const ACtx&, int varying, float (&v)[8]) const   (same for b)
//   __device__ void epilogue(const Tile&, f32x4 (&acc)[MT][NT], int m_wave, int n_wave, int lane) const
//        acc[mt][nt][r] <-> C[m_wave + mt*16 + (lane>>4)*4 + r][n_wave + nt*16 + (lane&15)]
// K groups (problems that declare `static constexpr int KG = 2`): two groups of four waves own the same output tile and
// take alternate K steps, each with its own pair of LDS stages; they swap one row tile of partial sums through LDS and
// each runs the epilogue on one (see conv_fwd_img_kernel).  For tilings that put one workgroup on a CU.
template <class P, class = void>
struct KGroupsOf { static constexpr int value = 1; };
template <class P>
struct KGroupsOf<P, std::void_t<decltype(P::KG)>> { static constexpr int value = P::KG; };
// TR-image pitch pads (problems that declare `static constexpr int TR_PAD_A / TR_PAD_B`; see TileGeom)
template <class P, class = void>
struct TrPadAOf { static constexpr int value = 16; };
template <class P>
struct TrPadAOf<P, std::void_t<decltype(P::TR_PAD_A)>> { static constexpr int value = P::TR_PAD_A; };
template <class P, class = void>
struct TrPadBOf { static constexpr int value = 16; };
template <class P>
struct TrPadBOf<P, std::void_t<decltype(P::TR_PAD_B)>> { static constexpr int value = P::TR_PAD_B; };
// operands stored S8 (problems that declare `static constexpr bool A_S8 / B_S8`): staged by copy
template <class P, class = void>
struct AS8Of { static constexpr bool value = false; };
template <class P>
struct AS8Of<P, std::void_t<decltype(P::A_S8)>> { static constexpr bool value = P::A_S8; };
template <class P, class = void>
struct BS8Of { static constexpr bool value = false; };
template <class P>
struct BS8Of<P, std::void_t<decltype(P::B_S8)>> { static constexpr bool value = P::B_S8; };

template <class P>
struct GemmTraits {
    static constexpr int BM = P::BM, BN = P::BN, WM = P::WM, WN = P::WN;
    static constexpr int MT = BM / WM / 16, NT = BN / WN / 16;
    static constexpr int PASSES = P::PASSES;
    static constexpr int A_PLANES = PASSES >= 2 ? 2 : 1;
    static constexpr int B_PLANES = PASSES >= 3 ? 2 : 1;
    using GA = TileGeom<BM, P::A_TR, TrPadAOf<P>::value>;
    using GB = TileGeom<BN, P::B_TR, TrPadBOf<P>::value>;
    static constexpr int STAGE_ELEMS = A_PLANES * GA::ELEMS + B_PLANES * GB::ELEMS;
    static constexpr int KG = KGroupsOf<P>::value;
    static constexpr int PIPE_BYTES = KG * 2 * STAGE_ELEMS * 2;
    static constexpr int XCHG_BYTES = KG > 1 ? KG * 4 * NT * 4 * 64 * 4 : 0;  // one row tile of every wave, both directions
    static constexpr int EPI_BYTES = P::EPI_LDS_BYTES + XCHG_BYTES;
    static constexpr int LDS_BYTES = PIPE_BYTES > EPI_BYTES ? PIPE_BYTES : EPI_BYTES;
};

template <class P>
__global__ __launch_bounds__(GEMM_THREADS * KGroupsOf<P>::value) void gemm_kernel(const P p) {
    ISDQN_EMPTY_KERNEL_RETURN
    using T = GemmTraits<P>;
    constexpr int KG = T::KG;
    using GA = typename T::GA;
    using GB = typename T::GB;
    constexpr int MT = T::MT, NT = T::NT, PASSES = T::PASSES;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    __bf16* smem = reinterpret_cast<__bf16*>(smem_raw);

    typename P::Tile tile;
    if (!p.tile((int)blockIdx.x, tile)) return;

    const int kg = KG > 1 ? (int)threadIdx.x / GEMM_THREADS : 0;
    const int tid = (int)threadIdx.x - kg * GEMM_THREADS;  // (group-local)
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / P::WN, wn = wave % P::WN;
    const int wave_m = wm * (MT * 16), wave_n = wn * (NT * 16);

    // ---- per-thread staging assignments (k-invariant part) ----
    typename P::ACtx actx[GA::PER_THREAD];
    typename P::BCtx bctx[GB::PER_THREAD];
    int a_lds[GA::PER_THREAD], a_var[GA::PER_THREAD];
    int b_lds[GB::PER_THREAD], b_var[GB::PER_THREAD];
    bool a_on[GA::PER_THREAD], b_on[GB::PER_THREAD];
#pragma unroll
    for (int i = 0; i < GA::PER_THREAD; ++i) {
        int c = tid + i * GEMM_THREADS;
        a_on[i] = c < GA::CHUNKS;
        if (!a_on[i]) c = 0;
        if constexpr (!P::A_TR) {
            int r = stash_row(c), kc = c & 3;
            actx[i] = p.a_ctx(tile, tile.m0 + r);
            a_var[i] = kc * 8;
            a_lds[i] = r * GA::PITCH + kc * 8;
        } else {
            int kk = c / (T::BM / 8), rc = c % (T::BM / 8);
            actx[i] = p.a_ctx(tile, tile.m0 + rc * 8);
            a_var[i] = kk;
            a_lds[i] = tr_row(kk) * GA::PITCH + rc * 8;
        }
    }
#pragma unroll
    for (int i = 0; i < GB::PER_THREAD; ++i) {
        int c = tid + i * GEMM_THREADS;
        b_on[i] = c < GB::CHUNKS;
        if (!b_on[i]) c = 0;
        if constexpr (!P::B_TR) {
            int r = stash_row(c), kc = c & 3;
            bctx[i] = p.b_ctx(tile, tile.n0 + r);
            b_var[i] = kc * 8;
            b_lds[i] = r * GB::PITCH + kc * 8;
        } else {
            int kk = c / (T::BN / 8), rc = c % (T::BN / 8);
            bctx[i] = p.b_ctx(tile, tile.n0 + rc * 8);
            b_var[i] = kk;
            b_lds[i] = tr_row(kk) * GB::PITCH + rc * 8;
        }
    }

    float sa[GA::PER_THREAD][8], sb[GB::PER_THREAD][8];

    auto fetch = [&](int k) {
#pragma unroll
        for (int i = 0; i < GA::PER_THREAD; ++i)
            if (GA::CHUNKS % GEMM_THREADS == 0 || a_on[i]) p.load_a(tile, actx[i], k + a_var[i], sa[i]);
#pragma unroll
        for (int i = 0; i < GB::PER_THREAD; ++i)
            if (GB::CHUNKS % GEMM_THREADS == 0 || b_on[i]) p.load_b(tile, bctx[i], k + b_var[i], sb[i]);
    };
    auto stash = [&](int stage) {
        __bf16* base = smem + (kg * 2 + stage) * T::STAGE_ELEMS;
        __bf16* a_hi = base;
        __bf16* a_lo = base + GA::ELEMS;
        __bf16* b_hi = base + T::A_PLANES * GA::ELEMS;
        __bf16* b_lo = b_hi + GB::ELEMS;
#pragma unroll
        for (int i = 0; i < GA::PER_THREAD; ++i) {
            if (GA::CHUNKS % GEMM_THREADS != 0 && !a_on[i]) continue;
            bf16x8 hi, lo;
            if constexpr (PASSES >= 2) {
                chunk_planes<AS8Of<P>::value>(sa[i], hi, lo);
                *reinterpret_cast<bf16x8*>(a_lo + a_lds[i]) = lo;
            } else {
                chunk_hi<AS8Of<P>::value>(sa[i], hi);
            }
            *reinterpret_cast<bf16x8*>(a_hi + a_lds[i]) = hi;
        }
#pragma unroll
        for (int i = 0; i < GB::PER_THREAD; ++i) {
            if (GB::CHUNKS % GEMM_THREADS != 0 && !b_on[i]) continue;
            bf16x8 hi, lo;
            if constexpr (PASSES >= 3) {
                chunk_planes<BS8Of<P>::value>(sb[i], hi, lo);
                *reinterpret_cast<bf16x8*>(b_lo + b_lds[i]) = lo;
            } else {
                chunk_hi<BS8Of<P>::value>(sb[i], hi);
            }
            *reinterpret_cast<bf16x8*>(b_hi + b_lds[i]) = hi;
        }
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) mfma_init(acc[mt][nt]);

    auto compute = [&](int stage) {
        const __bf16* base = smem + (kg * 2 + stage) * T::STAGE_ELEMS;
        const __bf16* a_hi = base;
        const __bf16* a_lo = base + GA::ELEMS;
        const __bf16* b_hi = base + T::A_PLANES * GA::ELEMS;
        const __bf16* b_lo = b_hi + GB::ELEMS;
        bf16x8 fa_hi[MT], fa_lo[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            fa_hi[mt] = read_frag<P::A_TR, GA::PITCH>(a_hi, wave_m + mt * 16, lane);
            if constexpr (PASSES >= 2) fa_lo[mt] = read_frag<P::A_TR, GA::PITCH>(a_lo, wave_m + mt * 16, lane);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            bf16x8 fb_hi = read_frag<P::B_TR, GB::PITCH>(b_hi, wave_n + nt * 16, lane);
            bf16x8 fb_lo;
            if constexpr (PASSES >= 3) fb_lo = read_frag<P::B_TR, GB::PITCH>(b_lo, wave_n + nt * 16, lane);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if constexpr (PASSES >= 3)
                    mfma_acc(acc[mt][nt], fa_hi[mt], fb_lo);
                if constexpr (PASSES >= 2)
                    mfma_acc(acc[mt][nt], fa_lo[mt], fb_hi);
                mfma_acc(acc[mt][nt], fa_hi[mt], fb_hi);
            }
        }
    };

    // ---- main loop: one barrier per K step, loads of step s+1 in flight under the MFMAs of step s ----
    // (K groups: group g takes steps g, g + KG, ...; a position past the last step loads zeros)
    const int nsteps = ((tile.k1 - tile.k0 + GEMM_BK - 1) / GEMM_BK + KG - 1) / KG;
    if (nsteps > 0) {
        fetch(tile.k0 + kg * GEMM_BK);
        stash(0);
        __syncthreads();
        for (int s = 0; s < nsteps; ++s) {
            const bool more = s + 1 < nsteps;
            if (more) fetch(tile.k0 + ((s + 1) * KG + kg) * GEMM_BK);
            compute(s & 1);
            if (more) stash((s + 1) & 1);
            __syncthreads();
        }
    }
    if constexpr (KG > 1) {
        // group g finalizes row tile mt = g of every wave: hand the other one to the partner (behind the epilogue's
        // own LDS area), add the partner's, and run the epilogue on a one-row-tile view
        static_assert(KG == 2 && MT == 2, "one row tile per K group");
        float* red = reinterpret_cast<float*>(smem_raw + P::EPI_LDS_BYTES);  // [group][wave][nt][r][lane]
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float give = kg == 0 ? acc[1][nt][r] : acc[0][nt][r];
                red[(((kg * 4 + wave) * NT + nt) * 4 + r) * 64 + lane] = give;
            }
        __syncthreads();
        f32x4 mine[1][NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float got = red[((((1 - kg) * 4 + wave) * NT + nt) * 4 + r) * 64 + lane];
                mine[0][nt][r] = (kg == 0 ? acc[0][nt][r] : acc[1][nt][r]) + got;
            }
        p.epilogue(tile, mine, tile.m0 + wave_m + kg * 16, tile.n0 + wave_n, lane, smem_raw);
    } else {
        p.epilogue(tile, acc, tile.m0 + wave_m, tile.n0 + wave_n, lane, smem_raw);
    }
}

template <class P>
static int launch_gemm(const P& p, int n_blocks, hipStream_t stream) {
    using T = GemmTraits<P>;
    static LdsConfigured configured;
    int rc = ensure_dynamic_lds(&gemm_kernel<P>, T::LDS_BYTES, configured);
    if (rc) return rc;
    ISDQN_REPORT_OCCUPANCY((&gemm_kernel<P>), GEMM_THREADS * T::KG, T::LDS_BYTES, n_blocks);
    if (n_blocks <= 0) return ISDQN_OK;
    hipLaunchKernelGGL(gemm_kernel<P>, dim3(n_blocks), dim3(GEMM_THREADS * T::KG), T::LDS_BYTES, stream, p);
    ISDQN_HIP_CHECK(hipGetLastError());
    return ISDQN_OK;
}

// --------------------------------------------------------------------------------------------
// Generic operand loaders over row-major fp32 matrices
// --------------------------------------------------------------------------------------------
__device__ __forceinline__ void zero8(float (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = 0.f;
}
// Sum over the 16 lanes of a DPP row (lanes 16q .. 16q+15), result in every lane of the row.  Four v_add_f32 with
// DPP modifiers (quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_ror:4, row_ror:8) instead of four ds_bpermute round
// trips through the LDS crossbar: the epilogues that reduce LayerNorm statistics and per-channel partial sums over
// the 16 pixel lanes of an MFMA accumulator issue hundreds of these per wave.
__device__ __forceinline__ float row16_sum(float x) {
    auto dpp = [](float v, auto ctrl) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), decltype(ctrl)::value, 0xF, 0xF, true));
    };
    x += dpp(x, std::integral_constant<int, 0xB1>{});   // quad_perm [1,0,3,2]
    x += dpp(x, std::integral_constant<int, 0x4E>{});   // quad_perm [2,3,0,1]
    x += dpp(x, std::integral_constant<int, 0x124>{});  // row_ror:4
    x += dpp(x, std::integral_constant<int, 0x128>{});  // row_ror:8
    return x;
}

// Loaders are BRANCH-FREE and MASK-FREE.  A load inside a divergent `if` makes hipcc branch around it and wait
// for it (s_waitcnt at the join); a select on the loaded registers right after the load (v = ok ? v : 0) is no
// better: it is a use of the data, so the compiler waits (vmcnt(0)) at the load site and a "prefetch" issued
// before a K step's MFMA work becomes a synchronous round trip.  Out-of-range chunks are therefore read from a
// block of zeros (`zero_chunk()`): the select happens on the ADDRESS, before the load, and the data registers are
// not touched until the consumer converts them for LDS.
static __device__ const float isdqn_zero_block[16] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f,
                                                      0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
__device__ __forceinline__ const float* zero_chunk() { return isdqn_zero_block; }

// Bounds-checked development build (-DISDQN_BOUNDS, tests/test_gpu_bounds.py): every global LOAD of the operand loaders, the image
// fills, the frame-id table and the prefetching epilogues is checked against the byte extents the caller registered
// (isdqn_debug_bounds_set: the exact extents of every tensor it hands to the library) plus the zero block.  The first load
// outside all of them is recorded (address + site number) instead of going unnoticed: an out-of-range read whose value is never
// used computes right results and only faults when the tensor happens to end a mapped segment (round 2's fc abort).
#if defined(ISDQN_BOUNDS)
struct BoundsTable {
    int n, bad, bad_site, pad;
    unsigned long long bad_addr;
    unsigned long long lo[64], hi[64];
};
static __device__ BoundsTable isdqn_bounds;
__device__ __forceinline__ void bounds_check(const void* p, int bytes, int site) {
    const unsigned long long a = (unsigned long long)p, b = a + (unsigned)bytes;
    const unsigned long long z = (unsigned long long)isdqn_zero_block;
    if (a >= z && b <= z + sizeof(isdqn_zero_block)) return;
    const int n = isdqn_bounds.n;
    if (n <= 0) return;
    bool ok = false;
    for (int i = 0; i < n; ++i) ok |= (a >= isdqn_bounds.lo[i] && b <= isdqn_bounds.hi[i]);
    if (!ok && atomicCAS(&isdqn_bounds.bad, 0, 1) == 0) {
        isdqn_bounds.bad_addr = a;
        isdqn_bounds.bad_site = site;
    }
}
#define ISDQN_BOUNDS_CHECK(p, bytes, site) bounds_check((const void*)(p), (int)(bytes), (site))
#else
#define ISDQN_BOUNDS_CHECK(p, bytes, site) ((void)0)
#endif
__device__ __forceinline__ void mask8(bool ok, float (&v)[8]) {  // for data that is already being consumed
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = ok ? v[i] : 0.f;
}
// Every pointer these loaders see is device global memory; saying so keeps the loads `global_load` even when the
// address was selected between a kernel argument and the zero block (a generic pointer would become `flat_load`,
// which counts on both wait counters and completes out of order, so every wait turns into vmcnt(0) lgkmcnt(0)).
#define ISDQN_GLOBAL __attribute__((address_space(1)))
// Streaming ("nontemporal") 16-byte accesses for the big cold streams of the step: the optimizer's moments and master weights
// (96 MB read once + written once by the fused-Adam GEMM) and the weight-gradient slabs' stores.  They do not claim cache lines the
// kernels running beside them re-read: c2 +2 - 4 %.  NOT for anything a later kernel finds in L2 / the Infinity Cache: streaming the
// forward's activation reads, the weight-gradient fills, the slab reads or the small tensors' moments cost 0.5 - 1.5 %, frames and
// pre-activations made no difference (profiles/round4/cache_policy_ab.txt).
__device__ __forceinline__ f32x4 nt_load4(const float* p) { return __builtin_nontemporal_load((const ISDQN_GLOBAL f32x4*)p); }
__device__ __forceinline__ void nt_store4(float* p, const f32x4& v) { __builtin_nontemporal_store(v, (ISDQN_GLOBAL f32x4*)p); }
// optax.adam on one element (isdqn.py:46, 85-86; optax scale_by_adam + scale(-lr)):  m, v updated in place, returns the new parameter.
// `inv_c1` / `inv_c2` = 1 / (1 - b^t), one IEEE division per thread; the square root and the reciprocal of (sqrt(v_hat) + eps) are the
// hardware's 1-ulp v_sqrt_f32 / v_rcp_f32: the three IEEE divisions and the refined square root per element they replace were 66 vector
// instructions per element in the fused-Adam GEMM's epilogue (two thirds of that kernel's instruction stream).  The update differs
// from the all-IEEE form by < 3e-7 of itself, i.e. < 2e-11 absolute at lr = 6.25e-5 -- below one ulp of any parameter above 2e-4.
__device__ __forceinline__ float adam_element(float& m, float& v, float p, float g, float b1, float b2, float lr, float eps, float inv_c1,
                                              float inv_c2) {
    m = b1 * m + (1.f - b1) * g;
    v = b2 * v + (1.f - b2) * g * g;
#if defined(ISDQN_ADAM_IEEE)
    return p - lr * ((m * inv_c1) / (sqrtf(v * inv_c2) + eps));
#else
    const float den = __builtin_amdgcn_sqrtf(v * inv_c2) + eps;
    return p - lr * ((m * inv_c1) * __builtin_amdgcn_rcpf(den));
#endif
}
__device__ __forceinline__ void load8_aligned(const float* p, float (&v)[8]) {
    ISDQN_BOUNDS_CHECK(p, 32, 1);
    const ISDQN_GLOBAL f32x4* gp = (const ISDQN_GLOBAL f32x4*)p;
    const f32x4 a = gp[0];
    const f32x4 b = gp[1];
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
    v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
}
__device__ __forceinline__ unsigned long long load_u64_unaligned(const uint8_t* p) {
    typedef unsigned long long __attribute__((aligned(1))) u64_u;
    ISDQN_BOUNDS_CHECK(p, 8, 2);
    return *(const ISDQN_GLOBAL u64_u*)p;
}

// Matrix X[rows][ld] (row-major).  `inner` bounds the contiguous index, `outer` the strided one.
// ROW image: fixed = row (outer), varying = first k of the chunk (inner).
// TR  image: fixed = first row of the chunk (inner!), varying = k (outer).
// `aligned8` promises base 16-B aligned, ld % 4 == 0 and inner % 8 == 0 (chunks never straddle the bound).
struct MatSrc {
    const float* base;
    int ld, outer, inner;
    int aligned8;
    // ALIGNED is a compile-time choice: a runtime flag keeps both bodies (and their waits) in the K loop
    template <bool ALIGNED = true>
    __device__ __forceinline__ void load(int o, int i0, float (&v)[8]) const {
        const bool ok = (o < outer) && (i0 < inner);
        if constexpr (ALIGNED) {
            load8_aligned(ok ? base + (int64_t)o * ld + i0 : zero_chunk(), v);
        } else {
            const float* p = base + (ok ? (int64_t)o * ld + i0 : (int64_t)0);
            const int last = ok ? inner - 1 - i0 : 0;  // last valid element offset in this chunk
            ISDQN_BOUNDS_CHECK(p, 4 * ((last < 7 ? last : 7) + 1), 3);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float x = p[j <= last ? j : last];
                v[j] = (ok && j <= last) ? x : 0.f;
            }
        }
    }
};

}  // namespace isdqn
