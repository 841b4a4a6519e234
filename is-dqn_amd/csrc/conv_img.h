// Image-resident forward convolution for gfx950.
//
// The generic engine (gemm_core.h) re-gathers every im2col chunk from global memory in every K step.  A
// convolution reads each input pixel (k/s)^2 times, and in the iS-DQN torso a whole input image is small
// (84x84x4 uint8 = 28 KB, 21x21x32 / 11x11x64 fp32 = 56 / 31 KB), so here a workgroup loads the input rows its
// 128 output pixels need ONCE -- coalesced 16-byte loads of frame rows / channel-last pixels -- converts them
// to bf16 (hi [+ lo]) and keeps them in LDS with the SAME zero border the SAME padding implies.  The MFMA B
// fragments of every tap are then plain ds_reads at a per-lane offset into that image; only the weight
// K-slices stream through the double-buffered LDS stage.  Epilogue: bias + LayerNorm(channels) + ReLU, fused
// (dqn.py:55-58, 62-65, 69-72), same as ConvFwd.
//
//   M = output channels (MT*16 >= cout_p), N = 128 output pixels of ONE image (4 waves x 32), K = taps*cin_p.
#pragma once
#include "net_problems.h"

namespace isdqn {

// Which pixel a fragment column holds.  The B fragments of the image-resident kernels are ds_read_b128s at per-lane pixel
// addresses; the hardware serves the columns {0-3, 12-15} of one k chunk together with the columns {4-11} of the next chunk
// (MI355X_MICROARCH.md, LDS), and with a pixel pitch of (pitch / 16) = 2 (mod 4) each of the two sets is conflict-free exactly
// when its 8 pixels are 8 consecutive positions of one image row.  So (a) column c of a 16-column tile holds position
// column_slot(c) of the tile's 16 pixels -- the two sets become the tile's first and second 8 positions -- and (b) where the
// whole image is resident the positions walk the pixel grid in 8-wide strips (row-major inside a strip), so that an aligned
// run of 8 never wraps around a row.  Row-major 16-pixel tiles wrapped in every tile of an 11-pixel row: conflict factor 2.0
// on these reads, 1.4 with this order (scripts/lds_conflicts.py; the epilogues are per pixel, any order serves them).
__device__ __forceinline__ int column_slot(int c16) { return c16 < 4 ? 8 + c16 : (c16 < 12 ? c16 - 4 : c16); }
struct PixelOrder {
    int strips;   // 0: positions are row-major pixel indices
    int n_full;   // positions inside the full 8-wide strips: H * 8 * (W / 8)
    int full_w;   // 8 * (W / 8)
    FastDiv d_h8, d_rem, d_w;  // by H * 8, by W % 8 (1 if none), by W
    PixelOrder() : strips(0), n_full(0), full_w(0) {}
    PixelOrder(int H, int W, bool use_strips) : strips(use_strips ? 1 : 0), n_full(H * 8 * (W / 8)), full_w(8 * (W / 8)) {
        d_h8 = FastDiv((uint32_t)(H * 8));
        d_rem = FastDiv((uint32_t)(W % 8 ? W % 8 : 1));
        d_w = FastDiv((uint32_t)W);
    }
    // position n (< H * W) -> pixel (y, x)
    __device__ __forceinline__ void map(int n, int& y, int& x) const {
        uint32_t a, b;
        if (!strips) {
            d_w.divmod((uint32_t)n, a, b);
            y = (int)a; x = (int)b;
        } else if (n < n_full) {
            d_h8.divmod((uint32_t)n, a, b);  // a: strip, b: position inside it
            y = (int)(b >> 3); x = (int)(a * 8 + (b & 7));
        } else {
            d_rem.divmod((uint32_t)(n - n_full), a, b);
            y = (int)a; x = full_w + (int)b;
        }
    }
};

struct ConvImgParams {
    ConvGeom g;
    MatSrc W;                // [cout_p][K], S8 mirror of the fp32 weights
    const float* in;         // NHWC input activations, S8 (if !U8)
    FrameSrc fs;             // uint8 frames (if U8)
    const float *bias, *gamma, *beta;
    float scale;
    float* act;
    float* z;
    int n_img, z_img;
    int tiles_per_img;       // ceil(npix / 128)
    int R, Wp;               // local rows / padded width of the LDS image
    int PP;                  // fp32 layers: pixel pitch in elements (cin_p + pad: spreads consecutive pixels over LDS banks)
    int plane_elems;         // bf16 elements of one precision plane of the image
    int ablate;              // profiling only (env ISDQN_ABLATE): 1 skip fill, 2 skip K loop, 4 skip epilogue
    FastDiv d_chunk, d_Wp, d_R;  // fill index math: chunks per pixel (fp32) or per row (uint8), padded width, local rows
    PixelOrder order;        // fp32 layers: which output pixel a fragment column holds (strips when the image is one tile)
    long long* stamps;       // profiling only (isdqn_debug_set_stamps): [workgroup][8] s_memtime / s_memrealtime at phase boundaries
};

template <int MT, int PASSES, bool U8>
struct ConvImgTraits {
    static constexpr int BM = MT * 16;
    static constexpr int A_PLANES = PASSES >= 2 ? 2 : 1;
    static constexpr int B_PLANES = (PASSES >= 3) ? 2 : 1;
    // Weight stage image [BM rows][32 k], row pitch 48 elements = 96 bytes.  ds_read_b128 is served in 16-lane groups
    // ({0-3,12-15,20-27}, ...), each lane covering 4 of the 64 banks; with fragment lane l reading row l&15, chunk
    // l>>4, the 16 lanes of a group hit 16 distinct bank quads exactly when (pitch/16) = 2 (mod 4).  The generic
    // engine's 80-byte pitch gives 2-way conflicts on these reads, which the four waves of the workgroup (all reading
    // the SAME weight fragments) pay four times per K step.
    struct GA {
        static constexpr int PITCH = 48;
        static constexpr int ELEMS = BM * PITCH;
        static constexpr int CHUNKS = BM * (GEMM_BK / 8);
    };
    static constexpr int A_STAGE = A_PLANES * GA::ELEMS;
};

// Workgroup id -> (image, tile of the image) such that the tiles of one image run on the SAME XCD (hardware places
// workgroup w on XCD w mod 8): the tiles of an image re-read the same input rows (overlapping frame rows of the first
// layer, the whole dz image for the stride classes of a data gradient), and on one XCD the second reader finds them in
// that XCD's L2.  Groups of 8 images: id = group * 8T + t * 8 + xcd  ->  image group * 8 + xcd, tile t.  The tail group
// (n_img not a multiple of 8) keeps the plain order.
__device__ __forceinline__ void xcd_image_tile(int wg, int n_img, int T, int& j, int& tile) {
    const int full = (n_img / 8) * 8 * T;
    if (T > 1 && wg < full) {
        const int grp = wg / (8 * T), r = wg - grp * 8 * T;
        j = grp * 8 + (r & 7);
        tile = r >> 3;
    } else {
        j = wg / T;
        tile = wg - j * T;
    }
}

// Copy a dense channel-last S8 image [H][W][C] (C % 8 == 0, `src` = its first element) into an LDS image whose local pixel
// (lr, xl), lr < Rl, xl < Wl, is source pixel (y0 + lr, x0 + xl) -- zeros outside the source (SAME padding / gradient
// borders) -- at img[(lr * Wl + xl) * PP + c] (+ plane_elems for the lo plane).  The 8-channel chunks are dealt round-robin
// to the NTHR threads and a thread WALKS its chunks: coordinates advance by additions and one carry per level instead of
// two divisions per chunk (the fills of these kernels are bound by instruction issue: ~250 instructions per chunk before,
// a third of a forward workgroup's life).  BATCH chunks are requested before the first one is written.
template <int NTHR, int PLANES, int BATCH>
__device__ __forceinline__ void fill_image_s8(__bf16* img, int plane_elems, const float* src, int H, int W, int C, int y0, int x0,
                                              int Rl, int Wl, int PP, int tid, const FastDiv& d_cpp, const FastDiv& d_Wl) {
    const int cpp = C >> 3, n_chunks = Rl * Wl * cpp;
    uint32_t pix_u, cc_u, lr_u, xl_u;
    d_cpp.divmod((uint32_t)tid, pix_u, cc_u);
    d_Wl.divmod(pix_u, lr_u, xl_u);
    int cc = (int)cc_u, lr = (int)lr_u, xl = (int)xl_u, pix = (int)pix_u;
    // per-round advance (wave-uniform): NTHR chunks = dpix pixels + dcc chunks; dpix pixels = dpy rows + dpx columns
    uint32_t dpix_u, dcc_u, dpy_u, dpx_u;
    d_cpp.divmod((uint32_t)NTHR, dpix_u, dcc_u);
    d_Wl.divmod(dpix_u, dpy_u, dpx_u);
    const int dpix = (int)dpix_u, dcc = (int)dcc_u, dpy = (int)dpy_u, dpx = (int)dpx_u;
    for (int c0 = tid; c0 - tid < n_chunks; ) {  // (uniform trip count: every thread runs the same number of rounds)
        float v[BATCH][8];
        int dst[BATCH];
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const bool on = c0 < n_chunks;
            const int iy = y0 + lr, ix = x0 + xl;
            const bool ok = on && iy >= 0 && iy < H && ix >= 0 && ix < W;
            load8_aligned(ok ? src + ((iy * W + ix) * C + cc * 8) : zero_chunk(), v[u]);
            dst[u] = on ? pix * PP + cc * 8 : -1;
            // advance to this thread's next chunk
            c0 += NTHR;
            cc += dcc;
            const int carry_c = cc >= cpp ? 1 : 0;
            cc -= carry_c ? cpp : 0;
            pix += dpix + carry_c;
            xl += dpx + carry_c;
            lr += dpy;
            const int carry_x = xl >= Wl ? 1 : 0;
            xl -= carry_x ? Wl : 0;
            lr += carry_x;
        }
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            bf16x8 hi, lo;  // S8 source: staged by copy
            if constexpr (PLANES >= 2) {
                s8_unpack(v[u], hi, lo);
                if (dst[u] >= 0) *reinterpret_cast<bf16x8*>(img + plane_elems + dst[u]) = lo;
            } else {
                s8_unpack_hi(v[u], hi);
            }
            if (dst[u] >= 0) *reinterpret_cast<bf16x8*>(img + dst[u]) = hi;
        }
    }
}

// The same copy in two halves for a workgroup that stages one image after another: `request` issues the loads of ALL chunks of
// this thread (at most N; the caller checks that the image fits N * NTHR chunks) and touches none of them, `commit` converts and
// writes them to LDS.  Between the two the registers travel under whatever the workgroup computes on the image it already holds
// (barriers wait for LDS traffic only, and nothing else in those kernels' K steps loads from global memory: the requests stay
// the newest loads until the commit).
template <int N>
struct ImagePrefetch {
    float v[N][8];
    int dst[N];
};
template <int NTHR, int N>
__device__ __forceinline__ void request_image_s8(ImagePrefetch<N>& r, const float* src, int H, int W, int C, int y0, int x0, int Rl, int Wl,
                                                 int PP, int tid, const FastDiv& d_cpp, const FastDiv& d_Wl) {
    const int cpp = C >> 3, n_chunks = Rl * Wl * cpp;
    uint32_t pix_u, cc_u, lr_u, xl_u;
    d_cpp.divmod((uint32_t)tid, pix_u, cc_u);
    d_Wl.divmod(pix_u, lr_u, xl_u);
    int cc = (int)cc_u, lr = (int)lr_u, xl = (int)xl_u, pix = (int)pix_u;
    uint32_t dpix_u, dcc_u, dpy_u, dpx_u;
    d_cpp.divmod((uint32_t)NTHR, dpix_u, dcc_u);
    d_Wl.divmod(dpix_u, dpy_u, dpx_u);
    const int dpix = (int)dpix_u, dcc = (int)dcc_u, dpy = (int)dpy_u, dpx = (int)dpx_u;
    int c0 = tid;
#pragma unroll
    for (int u = 0; u < N; ++u) {
        const bool on = c0 < n_chunks;
        const int iy = y0 + lr, ix = x0 + xl;
        const bool ok = on && iy >= 0 && iy < H && ix >= 0 && ix < W;
        load8_aligned(ok ? src + ((iy * W + ix) * C + cc * 8) : zero_chunk(), r.v[u]);
        r.dst[u] = on ? pix * PP + cc * 8 : -1;
        c0 += NTHR;
        cc += dcc;
        const int carry_c = cc >= cpp ? 1 : 0;
        cc -= carry_c ? cpp : 0;
        pix += dpix + carry_c;
        xl += dpx + carry_c;
        lr += dpy;
        const int carry_x = xl >= Wl ? 1 : 0;
        xl -= carry_x ? Wl : 0;
        lr += carry_x;
    }
}
template <int PLANES, int N>
__device__ __forceinline__ void commit_image_s8(__bf16* img, int plane_elems, const ImagePrefetch<N>& r) {
#pragma unroll
    for (int u = 0; u < N; ++u) {
        bf16x8 hi, lo;
        if constexpr (PLANES >= 2) {
            s8_unpack(r.v[u], hi, lo);
            if (r.dst[u] >= 0) *reinterpret_cast<bf16x8*>(img + plane_elems + r.dst[u]) = lo;
        } else {
            s8_unpack_hi(r.v[u], hi);
        }
        if (r.dst[u] >= 0) *reinterpret_cast<bf16x8*>(img + r.dst[u]) = hi;
    }
}

// uint8 frames -> the planar bf16 LDS image img[c][lr][Wp] (zero border included), 8 columns per chunk.
// Position-major: a thread owns (row, chunk-of-8-columns) positions and walks the `stack` planes of each, so the divisions,
// the border tests and the in-frame offset are computed once per position instead of once per chunk -- with four workgroups
// per CU filling at the same time this index arithmetic (integer multiplies are quarter rate), not the loads, was the cost
// of the fill.  All loads of a batch are issued before the first one is converted.
template <int NTHR, int STACK_MAX = 4>
__device__ __forceinline__ void fill_frames_u8(__bf16* img, const FrameSrc& fs, int j, int row_base, int x0, int R, int Wp, int tid,
                                               const FastDiv& d_cpr) {
    constexpr int PB = 2;  // positions per thread and batch: PB * stack loads in flight
    const int cpr = Wp / 8, per_plane = R * cpr, plane = R * Wp;
    const uint8_t* base[STACK_MAX];
#pragma unroll
    for (int c = 0; c < STACK_MAX; ++c) {
        const int id = c < fs.stack ? fs.frame_id(j, c) : -1;
        base[c] = id >= 0 ? fs.frames + (int64_t)id * fs.stride : nullptr;
    }
    const uint8_t* zeros = reinterpret_cast<const uint8_t*>(zero_chunk());
    for (int pb = 0; pb < per_plane; pb += NTHR * PB) {
        unsigned long long raw[PB][STACK_MAX];
        int sh[PB], dst[PB];
#pragma unroll
        for (int k = 0; k < PB; ++k) {  // loads only: nothing here touches the loaded registers
            const int pos = pb + k * NTHR + tid;
            const bool on = pos < per_plane;
            uint32_t lr, cx;
            d_cpr.divmod((uint32_t)(on ? pos : 0), lr, cx);
            const int iy = row_base + (int)lr, ix0 = (int)cx * 8 + x0;
            const bool ok = on && iy >= 0 && iy < fs.H && ix0 > -8 && ix0 < fs.W;
            const int ixc = min(max(ix0, 0), fs.W - 8);
            const int off = iy * fs.W + ixc;
            sh[k] = ix0 - ixc;
            dst[k] = on ? (int)lr * Wp + (int)cx * 8 : -1;
#pragma unroll
            for (int c = 0; c < STACK_MAX; ++c)
                raw[k][c] = load_u64_unaligned((ok && base[c] != nullptr) ? base[c] + off : zeros);
        }
#pragma unroll
        for (int k = 0; k < PB; ++k)
#pragma unroll
            for (int c = 0; c < STACK_MAX; ++c) {
                if (c >= fs.stack) continue;
                float v[8];
                FrameSrc::patch8_cvt(raw[k][c], sh[k], v);
                bf16x8 hi;
                round8(v, hi);
                if (dst[k] >= 0) *reinterpret_cast<bf16x8*>(img + c * plane + dst[k]) = hi;
            }
    }
}


// Four waves; each owns all MT channel tiles of 32 output pixels.  (An eight-wave variant -- two waves per SIMD sharing
// the image, channel halves per wave -- was measured slower: barrier-locked waves do not overlap each other.)
// The uint8 first layer (MT = 2: 2048 workgroups of 38 KB LDS at the headline size) is asked to fit four workgroups per
// CU (<= 128 registers): at three, its 8 workgroups per CU run as 3 + 3 + 2.
// KG = 2 (layers whose image leaves room for one workgroup per CU only): eight waves, two groups of four.  Both groups
// own the same 128 pixels x all channels and split the K steps (group g takes loop positions g, g+2, ...), each with
// its own pair of weight stages; group 1's accumulators are added to group 0's through LDS before the epilogue.  Two
// waves per SIMD interleave where one left every stall exposed (fill, K loop), and -- unlike splitting channels or
// pixels between eight waves -- the LDS reads per MFMA stay the same.
template <int MT, int PASSES, bool U8, int KG = 1>
__global__ __launch_bounds__(GEMM_THREADS * KG, (U8 && MT == 2 && PASSES == 2) ? 4 : 1) void conv_fwd_img_kernel(const ConvImgParams p) {
    ISDQN_EMPTY_KERNEL_RETURN
    using T = ConvImgTraits<MT, PASSES, U8>;
    using GA = typename T::GA;
    constexpr int NT = 2;
    constexpr int NTHR = GEMM_THREADS;            // threads of one K group (weight staging, fragments, epilogue)
    constexpr int NTHR_ALL = GEMM_THREADS * KG;   // all threads (image fill)
    constexpr int MTW = MT;  // channel tiles per wave
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    __bf16* smem = reinterpret_cast<__bf16*>(smem_raw);
    const int tid_all = threadIdx.x, kg = KG > 1 ? tid_all / GEMM_THREADS : 0;
    __bf16* a_stage = smem + kg * 2 * T::A_STAGE;        // 2 stages of weight K-slices (per K group)
    __bf16* img = smem + KG * 2 * T::A_STAGE;            // B_PLANES planes of the input tile
    const ConvGeom& g = p.g;

    const int tid = tid_all - kg * GEMM_THREADS, lane = tid & 63, wave = tid >> 6;  // (group-local)
#if defined(ISDQN_DEV)
#define ISDQN_STAMP(i)                                                                               \
    if (p.stamps != nullptr && threadIdx.x == 0) {                                                  \
        p.stamps[(int64_t)blockIdx.x * 8 + (i)] = (long long)__builtin_amdgcn_s_memtime();           \
        if ((i) == 0) p.stamps[(int64_t)blockIdx.x * 8 + 7] = (long long)__builtin_amdgcn_s_memrealtime(); \
    }
#define ISDQN_ABLATED(bit) ((p.ablate & (bit)) != 0)
#else
#define ISDQN_STAMP(i)
#define ISDQN_ABLATED(bit) (false)
#endif
    ISDQN_STAMP(0);
    if (ISDQN_ABLATED(8)) return;  // (development: the cost of launching this grid with this LDS / register footprint)
    constexpr int mt0 = 0;
    int j, tile;
    xcd_image_tile((int)blockIdx.x, p.n_img, p.tiles_per_img, j, tile);
    const int p0 = tile * 128;
    const int oy_min = (int)g.d_wout.div((uint32_t)p0);
    const int row_base = oy_min * g.stride - g.pad;  // global input row of local row 0

    // Epilogue parameters (bias, gamma, beta of up to 64 channels): one float per thread is requested now and parked
    // in LDS once the image fill has drained, so the epilogue neither waits on global memory nor do the values
    // occupy registers through the K loop (a few VGPRs decide whether two workgroups share a CU).
    const int grp = lane >> 4;
    __shared__ __attribute__((aligned(16))) float s_par[3][64];
    float par_v = 0.f;
    {
        const int which = tid >> 6, ch = tid & 63;
        const float* src = which == 0 ? p.bias : which == 1 ? p.gamma : p.beta;
        const bool ok = kg == 0 && tid < 192 && ch < g.cout_p && src != nullptr;
        ISDQN_BOUNDS_CHECK(ok ? src + ch : zero_chunk(), 4, 13);
        par_v = *(const ISDQN_GLOBAL float*)(ok ? src + ch : zero_chunk());
    }

    const int nsteps = (g.K + GEMM_BK - 1) / GEMM_BK;
    const int k_last = g.K - 8;  // last valid chunk start (weights are zero-filled past K, B only has to stay finite)
    // ---------------- weight K-slice staging (A operand, ROW image), as in the generic engine ----------------
    constexpr int A_PER = (GA::CHUNKS + NTHR - 1) / NTHR;
    int a_row[A_PER], a_var[A_PER], a_lds[A_PER];
    bool a_on[A_PER];
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
        int c = tid + i * NTHR;
        a_on[i] = c < GA::CHUNKS;
        if (!a_on[i]) c = 0;
        a_row[i] = stash_row(c);  // (lane quads take the rows of a block of four in the order 0, 2, 1, 3: gemm_core.h)
        a_var[i] = (c & 3) * 8;
        a_lds[i] = a_row[i] * GA::PITCH + (c & 3) * 8;
    }
    // Weight slices are fetched PF steps ahead into a ring of register sets: with one workgroup per CU nothing
    // else hides the L2 round trip, and one K step of MFMA work is shorter than it.
    constexpr int PF = 4;
    float sa[PF][A_PER][8];
    auto fetch = [&](int slot, int k) {
#pragma unroll
        for (int i = 0; i < A_PER; ++i) p.W.load(a_row[i], k + a_var[i], sa[slot][i]);
    };
    auto stash = [&](int slot, int stage) {
        __bf16* a_hi = a_stage + stage * T::A_STAGE;
        __bf16* a_lo = a_hi + GA::ELEMS;
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            if (GA::CHUNKS % NTHR != 0 && !a_on[i]) continue;  // (whole multiples: no guard, no branch in the K loop)
            bf16x8 hi, lo;  // the weights come from the S8 mirror: staged by copy
            if constexpr (PASSES >= 2) {
                s8_unpack(sa[slot][i], hi, lo);
                *reinterpret_cast<bf16x8*>(a_lo + a_lds[i]) = lo;
            } else {
                s8_unpack_hi(sa[slot][i], hi);
            }
            *reinterpret_cast<bf16x8*>(a_hi + a_lds[i]) = hi;
        }
    };

    const int nsteps_p = ((nsteps + KG - 1) / KG + PF - 1) / PF * PF;  // loop positions of one K group
    const int rot = ISDQN_ABLATED(16) ? 0 : (int)((blockIdx.x >> 3) % (unsigned)nsteps);
    auto slice = [&](int s) {  // K step handled at this group's loop position s; positions past nsteps read zeros
        const int gs = s * KG + kg;
        const int k = gs + rot;
        return gs < nsteps ? (k >= nsteps ? k - nsteps : k) : nsteps_p * KG;
    };
    // the first PF weight slices are requested BEFORE the image fill: they travel under it
#pragma unroll
    for (int d = 0; d < PF; ++d) fetch(d, slice(d) * GEMM_BK);

    // ---------------- stage the input rows of this tile into LDS (zero border included) ----------------
    // FILL_BATCH chunk loads are issued back to back before the first one is consumed: a plain
    // load -> convert -> store loop is one L2/HBM round trip per iteration.
    constexpr int FILL_BATCH = (U8 ? 8 : 12) / KG;  // chunks in flight per thread: the 21x21x32 image (10.6 chunks) in ONE round trip
    if (ISDQN_ABLATED(1)) {
    } else if constexpr (U8) {
#if defined(ISDQN_DEV)
        if (ISDQN_ABLATED(32)) {  // experiment: no id-table hop (frame slots j*stack + c; wrong results, same traffic)
            FrameSrc direct = p.fs;
            direct.ids = nullptr;
            fill_frames_u8<NTHR_ALL>(img, direct, j, row_base, -g.pad, p.R, p.Wp, tid_all, p.d_chunk);
        } else
#endif
        fill_frames_u8<NTHR_ALL>(img, p.fs, j, row_base, -g.pad, p.R, p.Wp, tid_all, p.d_chunk);
    } else {
        // channel-last: img[lr][xp][PP], chunk = 8 channels of one padded pixel
        fill_image_s8<NTHR_ALL, T::B_PLANES, FILL_BATCH>(img, p.plane_elems, p.in + (int64_t)j * g.hin * g.win * g.cin_p, g.hin, g.win,
                                                         g.cin_p, row_base, -g.pad, p.R, p.Wp, p.PP, tid_all, p.d_chunk, p.d_Wp);
    }

    ISDQN_STAMP(1);  // fill loads consumed, LDS image written (this wave)
    // ---------------- per-lane patch origins of the two 16-pixel column tiles of this wave ----------------
    int b_org[NT];  // element offset of the patch origin inside one image plane
    int out_pix[NT];  // output pixel of this lane's column (-1: past the image)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        int pp = p0 + wave * 32 + nt * 16 + (U8 ? (lane & 15) : column_slot(lane & 15));
        const bool in_img = pp < g.npix;
        pp = in_img ? pp : g.npix - 1;  // lanes past the image compute a duplicate pixel; never stored
        int oy, ox;
        if constexpr (U8) {
            uint32_t oy_u, ox_u;
            g.d_wout.divmod((uint32_t)pp, oy_u, ox_u);
            oy = (int)oy_u; ox = (int)ox_u;
        } else {
            p.order.map(pp, oy, ox);
        }
        out_pix[nt] = in_img ? oy * g.wout + ox : -1;
        const int ly0 = oy * g.stride - g.pad - row_base;  // >= 0 by construction
        const int lx0 = ox * g.stride;                     // padded column of tap kx = 0
        b_org[nt] = U8 ? (ly0 * p.Wp + lx0) : (ly0 * p.Wp + lx0) * p.PP;
    }

    f32x4 acc[MTW][NT];
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) mfma_init(acc[mt][nt]);


    // Fragments of one K step: MTW weight tiles and NT pixel tiles, each hi (+ lo).  Two sets alternate so that the
    // ds_reads of step s+1 are in flight while the MFMAs of step s run (one wave per SIMD: nothing else overlaps them).
    struct Frags {
        bf16x8 a_hi[MTW], a_lo[MTW], b_hi[NT], b_lo[NT];
    };
    auto read_frags = [&](int stage, int kk, Frags& f) {
        const __bf16* a_hi = a_stage + stage * T::A_STAGE;
        const __bf16* a_lo = a_hi + GA::ELEMS;
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) {
            f.a_hi[mt] = read_frag<false, GA::PITCH>(a_hi, (mt0 + mt) * 16, lane);
            if constexpr (PASSES >= 2) f.a_lo[mt] = read_frag<false, GA::PITCH>(a_lo, (mt0 + mt) * 16, lane);
        }
        // tap / channel of this lane's 8-element chunk
        int kq = kk * GEMM_BK + grp * 8;
        kq = kq < k_last ? kq : k_last;
        int tap_off;
        if constexpr (U8) {
            const int c = kq >> 6, ky = (kq >> 3) & 7;
            tap_off = (c * p.R + ky) * p.Wp;
        } else {
            uint32_t tap, ci, ky, kx;
            g.d_cinp.divmod((uint32_t)kq, tap, ci);
            g.d_ksz.divmod(tap, ky, kx);
            tap_off = ((int)ky * p.Wp + (int)kx) * p.PP + (int)ci;
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const __bf16* src = img + b_org[nt] + tap_off;
            if constexpr (U8) {
                // 8-byte aligned (padded column = 4*ox), two ds_read_b64
                typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
                bf16x4 h0 = *reinterpret_cast<const bf16x4*>(src);
                bf16x4 h1 = *reinterpret_cast<const bf16x4*>(src + 4);
                f.b_hi[nt] = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
            } else {
                f.b_hi[nt] = *reinterpret_cast<const bf16x8*>(src);
                if constexpr (PASSES >= 3) f.b_lo[nt] = *reinterpret_cast<const bf16x8*>(src + p.plane_elems);
            }
        }
    };
    auto mfma_step = [&](const Frags& f) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt) {
                if constexpr (PASSES >= 3)
                    mfma_acc(acc[mt][nt], f.a_hi[mt], f.b_lo[nt]);
                if constexpr (PASSES >= 2)
                    mfma_acc(acc[mt][nt], f.a_lo[mt], f.b_hi[nt]);
                mfma_acc(acc[mt][nt], f.a_hi[mt], f.b_hi[nt]);
            }
    };

    // The loop body is straight-line code: a fetch or a stash under a condition ends in a control-flow merge, where
    // the compiler must choose one wait count that is safe on both paths, which turns the counted vmcnt(N) of the
    // register ring into vmcnt(0) right behind the fetch.  So the step count is padded to a multiple of PF; slices
    // past K read the zero block (MatSrc::load) and add nothing.
    //
    // Every workgroup streams the SAME weight slices.  Slice s is the 128-byte column block s of every weight row
    // (rows are K*4 bytes apart), so the workgroups of one XCD, started together, would all pull the same few L2
    // lines -- one or two L2 channels -- at the same moment.  Each workgroup therefore walks the K steps from its
    // own starting slice; workgroups b, b+8, b+16, ... share an XCD, hence the rotation by b/8.
    static_assert(PF % 2 == 0 && PF >= 4, "the fragment sets alternate with the step parity; steps 0..3 are pre-fetched");
    Frags fr[2];
    if (kg == 0 && tid < 192) s_par[tid >> 6][tid & 63] = par_v;
    stash(0, 0);  // steps 0 and 1 fill the two LDS stages; their ring slots take steps PF and PF+1
    stash(1, 1);
    fetch(0, slice(PF) * GEMM_BK);
    fetch(1, slice(PF + 1) * GEMM_BK);
    __syncthreads();  // image and the first two weight slices visible
    ISDQN_STAMP(2);
    if (!ISDQN_ABLATED(2)) {
        // One barrier per K step, and everything between two barriers is independent, so the compiler is free to
        // interleave it: the fragment reads of step s+1 (stage (s+1)&1, written during step s-1), the MFMAs of step s
        // (fragments read during step s-1), the conversion of slice s+2 into stage s&1 (whose readers finished before
        // the last barrier) and the request for slice s+2+PF.
        read_frags(0, slice(0), fr[0]);
        __syncthreads();  // every wave has read stage 0 before step 0 overwrites it with slice 2
        // Hand-interleaved K step (fp32 layers, 24 MFMAs per step).  The MFMAs are opaque asm to the scheduler (rules
        // above), which leaves them back to back with every other piece of the step in front of or behind the block:
        // 384 cycles of matrix pipe in an ~880-cycle step.  A 16x16x32 MFMA holds the pipe for 16 cycles but the issue
        // port for 4, so each one is followed by ONE micro-op of the other pieces -- a fragment read of step s+1, the
        // hi/lo conversion of one element of slice s+2, its two LDS writes, the request for slice s+2+PF -- and a
        // sched_barrier pins that order.  Per accumulator the pass order (hi.lo, lo.hi, hi.hi) is the old one: same bits.
        constexpr bool INTERLEAVED = !U8 && PASSES == 3 && MTW == 4 && NT == 2 && A_PER == 1 &&
                                     GA::CHUNKS == NTHR;
        auto tap_offset_of = [&](int kk) {  // image offset of this lane's 8-channel chunk of K step kk (fp32 layers)
            int kq = kk * GEMM_BK + grp * 8;
            kq = kq < k_last ? kq : k_last;
            uint32_t tap, ci, ky, kx;
            g.d_cinp.divmod((uint32_t)kq, tap, ci);
            g.d_ksz.divmod(tap, ky, kx);
            return ((int)ky * p.Wp + (int)kx) * p.PP + (int)ci;
        };
        int tap_next = INTERLEAVED ? tap_offset_of(slice(1)) : 0;
        for (int s0 = 0; s0 < nsteps_p; s0 += PF) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int s = s0 + u;
                if constexpr (INTERLEAVED) {
                    const Frags& fc = fr[u & 1];
                    Frags& fn = fr[(u + 1) & 1];
                    const __bf16* na_hi = a_stage + ((s + 1) & 1) * T::A_STAGE;
                    const __bf16* na_lo = na_hi + GA::ELEMS;
                    const int tap_off = tap_next;  // of slice(s + 1): computed in the last slot of the previous step
                    const int slot = (u + 2) % PF;  // (a constant once the step loop is unrolled)
                    __bf16* st_hi = a_stage + (s & 1) * T::A_STAGE + a_lds[0];
                    __bf16* st_lo = st_hi + GA::ELEMS;
                    bf16x8 c_hi, c_lo;
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < 24; ++j) {
                        const int pass = j >> 3, nt = (j >> 2) & 1, mt = j & 3;
                        mfma_acc(acc[mt][nt], pass == 1 ? fc.a_lo[mt] : fc.a_hi[mt], pass == 0 ? fc.b_lo[nt] : fc.b_hi[nt]);
                        if (j < 4) fn.a_hi[j] = read_frag<false, GA::PITCH>(na_hi, j * 16, lane);
                        else if (j < 8) fn.a_lo[j - 4] = read_frag<false, GA::PITCH>(na_lo, (j - 4) * 16, lane);
                        else if (j < 10) fn.b_hi[j - 8] = *reinterpret_cast<const bf16x8*>(img + b_org[j - 8] + tap_off);
                        else if (j < 12) fn.b_lo[j - 10] = *reinterpret_cast<const bf16x8*>(img + b_org[j - 10] + tap_off + p.plane_elems);
                        else if (j == 12) s8_unpack(sa[slot][0], c_hi, c_lo);  // S8 mirror: no conversion
                        else if (j < 16) {
                        } else if (j == 16) *reinterpret_cast<bf16x8*>(st_lo) = c_lo;
                        else if (j == 17) *reinterpret_cast<bf16x8*>(st_hi) = c_hi;
                        else if (j == 18) fetch(slot, slice(s + 2 + PF) * GEMM_BK);
                        else if (j == 19) tap_next = tap_offset_of(slice(s + 2));  // (nothing trails the last MFMAs)
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    __syncthreads();
                    continue;
                }
                read_frags((s + 1) & 1, slice(s + 1), fr[(u + 1) & 1]);
                mfma_step(fr[u & 1]);
                stash((u + 2) % PF, s & 1);
                fetch((u + 2) % PF, slice(s + 2 + PF) * GEMM_BK);
                // Issue order for the block: a 16x16x32 MFMA occupies the matrix pipe for 16 cycles but the issue port
                // for 8, so every MFMA is followed by one LDS operation or two vector instructions of the other
                // pieces; left alone, the compiler emits the MFMAs back to back and the rest serially behind them.
                constexpr int N_MFMA = MTW * NT * (PASSES >= 3 ? 3 : PASSES);
#pragma unroll
                for (int i = 0; i < N_MFMA; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
                    if (i < N_MFMA / 2) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // 1 DS read/write
                    __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);  // 2 VALU
                }
                __syncthreads();
            }
        }
    }
    ISDQN_STAMP(3);  // K loop done
    if constexpr (KG > 1) {
        // The two groups hold partial sums of the same tile.  Group g finalizes pixel tile nt = g of every wave: each
        // wave hands the OTHER tile to its partner through LDS (the image is dead: the K loop ended with a barrier).
        static_assert(KG == 2 && NT == 2, "one pixel tile per K group");
        float* red = reinterpret_cast<float*>(img);  // [group][wave][mt][r][lane]: conflict-free, 4 KB per wave
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float give = kg == 0 ? acc[mt][1][r] : acc[mt][0][r];
                red[(((kg * 4 + wave) * MTW + mt) * 4 + r) * 64 + lane] = give;
            }
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float got = red[((((1 - kg) * 4 + wave) * MTW + mt) * 4 + r) * 64 + lane];
                if (kg == 0) acc[mt][0][r] += got;
                else acc[mt][1][r] += got;
            }
    }
    if (ISDQN_ABLATED(4)) {
        if (acc[0][0][0] == 12345.678f) p.act[0] = 1.f;  // keep the accumulators alive
        return;
    }

    // ---------------- epilogue: bias + LayerNorm over channels + ReLU (same math as ConvFwd::epilogue) -------------
    float bi[MTW][4], ga[MTW][4], be[MTW][4];
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
        const int ch0 = ((mt0 + mt) * 16 + grp * 4) & 63;
        const float4 b4 = *reinterpret_cast<const float4*>(&s_par[0][ch0]);
        const float4 g4 = *reinterpret_cast<const float4*>(&s_par[1][ch0]);
        const float4 e4 = *reinterpret_cast<const float4*>(&s_par[2][ch0]);
        bi[mt][0] = b4.x; bi[mt][1] = b4.y; bi[mt][2] = b4.z; bi[mt][3] = b4.w;
        ga[mt][0] = g4.x; ga[mt][1] = g4.y; ga[mt][2] = g4.z; ga[mt][3] = g4.w;
        be[mt][0] = e4.x; be[mt][1] = e4.y; be[mt][2] = e4.z; be[mt][3] = e4.w;
    }
    const float inv_c = 1.0f / (float)g.cout;
    float zv[NT][MTW][4], s1[NT], s2[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        s1[nt] = s2[nt] = 0.f;
        if (KG > 1 && nt != kg) continue;  // the partner group finalizes this tile
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int ch = (mt0 + mt) * 16 + grp * 4 + r;
                float zz = ch < g.cout ? acc[mt][nt][r] * p.scale + bi[mt][r] : 0.f;
                zv[nt][mt][r] = zz;
                s1[nt] += zz;
                s2[nt] += zz * zz;
            }
        if (p.gamma != nullptr) {
            s1[nt] += __shfl_xor(s1[nt], 16); s1[nt] += __shfl_xor(s1[nt], 32);
            s2[nt] += __shfl_xor(s2[nt], 16); s2[nt] += __shfl_xor(s2[nt], 32);
        }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        if (KG > 1 && nt != kg) continue;
        float mean = 0.f, rstd = 1.f;
        if (p.gamma != nullptr) {
            mean = s1[nt] * inv_c;
            float var = fmaxf(s2[nt] * inv_c - mean * mean, 0.f);
            rstd = rsqrtf(var + 1e-6f);
        }
        if (out_pix[nt] >= 0) {
            const int64_t pix = (int64_t)j * g.npix + out_pix[nt];
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt) {
                int ch0 = (mt0 + mt) * 16 + grp * 4;
                if (ch0 >= g.cout_p) continue;
                float4 a, zq;
                float* ap = &a.x;
                float* zp = &zq.x;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float y = p.gamma != nullptr ? (zv[nt][mt][r] - mean) * (rstd * ga[mt][r]) + be[mt][r] : zv[nt][mt][r];
                    ap[r] = (ch0 + r < g.cout) ? fmaxf(y, 0.f) : 0.f;
                    zp[r] = zv[nt][mt][r];
                }
                s8_store_quad_paired(p.act + pix * g.cout_p, ch0, a.x, a.y, a.z, a.w);  // activations: S8 (consumers stage by copy)
                if (j < p.z_img) *reinterpret_cast<float4*>(p.z + pix * g.cout_p + ch0) = zq;
            }
        }
    }
    ISDQN_STAMP(4);  // epilogue stores issued
#if defined(ISDQN_DEV)
    if (p.stamps != nullptr) {
        __builtin_amdgcn_s_waitcnt(0);
        ISDQN_STAMP(5);  // stores retired
    }
#endif
#undef ISDQN_STAMP
#undef ISDQN_ABLATED
}

template <int MT, int PASSES, bool U8, int KG = 1>
static int launch_conv_fwd_img(const ConvImgParams& p, hipStream_t st) {
    using T = ConvImgTraits<MT, PASSES, U8>;
    const int lds = (KG * 2 * T::A_STAGE + T::B_PLANES * p.plane_elems) * 2;
    static LdsConfigured configured;
    if (int rc = ensure_dynamic_lds(&conv_fwd_img_kernel<MT, PASSES, U8, KG>, lds, configured)) return rc;
    ISDQN_REPORT_OCCUPANCY((&conv_fwd_img_kernel<MT, PASSES, U8, KG>), GEMM_THREADS * KG, lds, p.n_img * p.tiles_per_img);
    hipLaunchKernelGGL((conv_fwd_img_kernel<MT, PASSES, U8, KG>), dim3(p.n_img * p.tiles_per_img), dim3(GEMM_THREADS * KG), lds,
                       st, p);
    ISDQN_HIP_CHECK(hipGetLastError());
    return ISDQN_OK;
}

// LDS bytes the image-resident kernel would need for this layer (host-side feasibility check)
static inline int conv_img_geometry(const ConvGeom& g, bool u8, int stack, int b_planes, int mt, int a_planes, int& R,
                                    int& Wp, int& plane_elems, int pixel_pitch = 0) {
    int rows_per_tile = g.npix <= 128 ? g.hout : ((128 + g.wout - 2) / g.wout + 1);
    if (rows_per_tile > g.hout) rows_per_tile = g.hout;
    R = g.stride * (rows_per_tile - 1) + g.ksz;
    if (u8) {
        Wp = ((g.wout - 1) * g.stride + g.ksz + 7) / 8 * 8;  // padded width, room for the right-most patch
        plane_elems = stack * R * Wp;
    } else {
        Wp = (g.wout - 1) * g.stride + g.ksz;           // = win + pad_lo + pad_hi
        plane_elems = R * Wp * (pixel_pitch ? pixel_pitch : g.cin_p);
    }
    const int a_stage = a_planes * (mt * 16) * 48;  // ConvImgTraits::GA::PITCH
    return (2 * a_stage + b_planes * plane_elems) * 2;
}


// =====================================================================================================
// Image-resident weight gradient
// =====================================================================================================
//   dW[co][k'] = sum_images sum_pix dz[pix][co] * in[pix shifted by the tap of k'][ci of k']
// A workgroup owns a group of images and a slice of the k' columns (a few taps).  Per image it stages the
// dz image ([pix][cout_p]) and the input image (zero-bordered, as in the forward kernel) in LDS once; the
// contraction index (pixels) is the slow index of BOTH images, so every MFMA fragment is a transposing
// ds_read_b64_tr_b16 at a per-lane pixel address -- no im2col gather from HBM, and the dz fragments of a
// K step are reused by all taps.  Accumulators stay in registers across the images of the group; one fp32
// slab per image group goes to HBM (reduced by the Adam kernel).
struct ConvWgradImgParams {
    ConvGeom g;
    const float* dz;     // [n_img][npix][cout_p]
    const float* in;     // NHWC input activations, S8 (if !U8)
    FrameSrc fs;         // uint8 frames (if U8)
    float* slabs;        // [n_img_groups][cout_p][K]
    float scale;
    int n_img, G;        // images per workgroup
    int n_col_groups, NC;  // k' columns per workgroup (multiple of 64)
    int R, Wp;           // input image: local rows (whole image), padded width
    int PPin;            // fp32 layers: pixel pitch of the input image in elements (cin_p + pad, see conv_wgrad_img_kernel)
    int in_plane;        // elements of one precision plane of the input image
    int PA, npix_pad;    // dz image: row pitch (elements), rows padded to a multiple of 32
    int dz_plane;        // elements of one precision plane of the dz image
    FastDiv d_ncg;         // by n_col_groups
    FastDiv d_chunk, d_Wp, d_R, d_dzchunk, d_npixpad;  // fill index math (see ConvImgParams); d_npixpad: by npix_pad
    int prefetch;          // the next image of the group is requested into registers under this image's K steps (set by the launcher)
    long long* stamps;     // profiling only (isdqn_debug_set_stamps "wgrad:<layer>"), as in ConvImgParams
};
// chunks per thread the prefetching form holds (dz image / input image): the 64-channel layers of the headline network on 256 and
// 512 threads (11 x 11 x 64 dz = 4 / 2; 13 x 13 x 64 input = 6, 24 x 24 x 32 input on 512 threads = 5)
// (first layer, uint8 input: the 21 x 21 x 32 dz image = 7 chunks per thread of 256)
template <int WV> struct WgradPrefetch { static constexpr int DZ = 1024 / (64 * WV), IN = WV == 4 ? 6 : 5, DZ_U8 = 1792 / (64 * WV); };

template <bool U8, int PASSES, int NTHR = GEMM_THREADS>
__device__ __forceinline__ void fill_input_image(__bf16* img, int plane_elems, const ConvGeom& g, const FrameSrc& fs,
                                                 const float* in, int j, int row_base, int R, int Wp, int PP, int tid,
                                                 const FastDiv& d_chunk, const FastDiv& d_Wp, const FastDiv& d_R) {
    constexpr int FILL_BATCH = 8;
    if constexpr (U8) {
        fill_frames_u8<NTHR>(img, fs, j, row_base, -g.pad, R, Wp, tid, d_chunk);
    } else {
        fill_image_s8<NTHR, (PASSES >= 3 ? 2 : 1), FILL_BATCH>(img, plane_elems, in + (int64_t)j * g.hin * g.win * g.cin_p, g.hin, g.win, g.cin_p,
                                                               row_base, -g.pad, R, Wp, PP, tid, d_chunk, d_Wp);
    }
}

// one transposed fragment: 8 k-rows given by two per-lane row addresses (rows q and q+4 of the lane's 8-row group)
__device__ __forceinline__ bf16x8 tr_frag(const __bf16* a0, const __bf16* a1) {
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)a0);
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)a1);
    s16x8 r = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, r);
}

// WV waves (4 or 8): a column tile is owned by exactly one wave, so eight waves only re-deal the tiles (NTW is per
// wave; tiles past the last column are computed on column 0 and never stored).  The kernel stages two images per
// workgroup and runs 4 K steps on each: with one wave per SIMD the two fills are most of its time.
template <int MT, int NTW, int PASSES, bool U8, int WV = 4>
__global__ __launch_bounds__(64 * WV) void conv_wgrad_img_kernel(const ConvWgradImgParams p) {
    ISDQN_EMPTY_KERNEL_RETURN
    constexpr int NTHR = 64 * WV;
    constexpr int A_PLANES = PASSES >= 2 ? 2 : 1;
    constexpr int B_PLANES = (PASSES >= 3) ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    __bf16* dzi = reinterpret_cast<__bf16*>(smem_raw);     // A_PLANES planes [npix_pad][PA]
    __bf16* img = dzi + A_PLANES * p.dz_plane;             // B_PLANES planes of the input image
    const ConvGeom& g = p.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#if defined(ISDQN_DEV)  // phase stamps of the first two images of the group: 0 start, 1 / 3 image staged, 2 / 4 its K steps done, 5 slab stored
#define WG_STAMP(i)                                                                                   \
    if (p.stamps != nullptr && threadIdx.x == 0) {                                                   \
        p.stamps[(int64_t)blockIdx.x * 8 + (i)] = (long long)__builtin_amdgcn_s_memtime();            \
        if ((i) == 0) p.stamps[(int64_t)blockIdx.x * 8 + 7] = (long long)__builtin_amdgcn_s_memrealtime(); \
    }
#else
#define WG_STAMP(i)
#endif
    WG_STAMP(0);
    uint32_t ig_u, cg_u;
    p.d_ncg.divmod(blockIdx.x, ig_u, cg_u);
    const int cg = (int)cg_u, ig = (int)ig_u;
    const int j0 = ig * p.G, j1 = min(p.n_img, j0 + p.G);
    const int grp = lane >> 4, li = lane & 15, q = li >> 2, pq = li & 3;

    // column tiles of this wave: global k' column of tile t = cg*NC + (wave + WV*t)*16
    int col_off[NTW];  // element offset inside the input image contributed by the tile's tap / channel / lane part
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        const int c0 = cg * p.NC + (wave + WV * t) * 16;  // (>= K for surplus tiles of an uneven deal)
        if constexpr (U8) {
            const int c = c0 >> 6, ky0 = (c0 >> 3) & 7;  // the 16 columns are (ky0, kx 0..7), (ky0+1, kx 0..7)
            col_off[t] = (c * p.R + ky0 + (pq >> 1)) * p.Wp + 4 * (pq & 1);
        } else {
            uint32_t tap, ci, ky, kx;
            g.d_cinp.divmod((uint32_t)(c0 < g.K ? c0 : 0), tap, ci);
            g.d_ksz.divmod(tap, ky, kx);
            col_off[t] = ((int)ky * p.Wp + (int)kx) * p.PPin + (int)ci + 4 * pq;
        }
    }

    f32x4 acc[MT][NTW];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < NTW; ++t) mfma_init(acc[mt][t]);

    const int nsteps = p.npix_pad / GEMM_BK;
    auto k_steps = [&]() {
        for (int ks = 0; ks < nsteps; ++ks) {
            // The two 4-row blocks of this lane's 8-pixel group.  The contraction order is free, so k row 8*grp + 4*h + q of
            // the step takes pixel tr_row(.) of its 32: the eight k rows one 32-lane group of a ds_read_b64_tr_b16 addresses
            // are then eight CONSECUTIVE pixels, which a pixel pitch of an odd multiple of 32 bytes (dz: PA; input: stride *
            // PPin) spreads over all 64 banks.  In natural order the two k groups of a read sat 8 pixels apart on the same
            // banks and the unpadded 128-byte input pixels made that 4-way (scripts/lds_conflicts.py; 60 % of the LDS cycles
            // of these kernels were bank conflicts, profiles/round2).
            int pixoff[2], arow[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                int kp = ks * GEMM_BK + tr_row(8 * grp + 4 * h + q);
                arow[h] = kp * p.PA + 4 * pq;
                kp = kp < g.npix ? kp : g.npix - 1;  // padded pixels: dz rows are zero, the input only has to be finite
                uint32_t oy, ox;
                g.d_wout.divmod((uint32_t)kp, oy, ox);
                pixoff[h] = U8 ? ((int)oy * g.stride * p.Wp + (int)ox * g.stride)
                               : ((int)oy * g.stride * p.Wp + (int)ox * g.stride) * p.PPin;
            }
            bf16x8 fa_hi[MT], fa_lo[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                fa_hi[mt] = tr_frag(dzi + arow[0] + mt * 16, dzi + arow[1] + mt * 16);
                if constexpr (PASSES >= 2)
                    fa_lo[mt] = tr_frag(dzi + p.dz_plane + arow[0] + mt * 16, dzi + p.dz_plane + arow[1] + mt * 16);
            }
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
                const __bf16* b0 = img + pixoff[0] + col_off[t];
                const __bf16* b1 = img + pixoff[1] + col_off[t];
                bf16x8 fb_hi = tr_frag(b0, b1), fb_lo;
                if constexpr (PASSES >= 3) fb_lo = tr_frag(b0 + p.in_plane, b1 + p.in_plane);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    if constexpr (PASSES >= 3)
                        mfma_acc(acc[mt][t], fa_hi[mt], fb_lo);
                    if constexpr (PASSES >= 2)
                        mfma_acc(acc[mt][t], fa_lo[mt], fb_hi);
                    mfma_acc(acc[mt][t], fa_hi[mt], fb_hi);
                }
            }
        }
    };
    bool staged = false;
    if constexpr (!U8) {
        if (p.prefetch) {  // (uniform) two or more images per group and both images fit the register budget
            staged = true;
            ImagePrefetch<WgradPrefetch<WV>::DZ> ra;
            ImagePrefetch<WgradPrefetch<WV>::IN> rb;
            auto request = [&](int j) {
                request_image_s8<NTHR>(ra, p.dz + (int64_t)j * g.npix * g.cout_p, 1, g.npix, g.cout_p, 0, 0, 1, p.npix_pad, p.PA, tid,
                                       p.d_dzchunk, p.d_npixpad);
                request_image_s8<NTHR>(rb, p.in + (int64_t)j * g.hin * g.win * g.cin_p, g.hin, g.win, g.cin_p, -g.pad, -g.pad, p.R, p.Wp,
                                       p.PPin, tid, p.d_chunk, p.d_Wp);
            };
            request(j0);
            for (int j = j0; j < j1; ++j) {
                __syncthreads();  // previous image fully consumed
                commit_image_s8<A_PLANES>(dzi, p.dz_plane, ra);
                commit_image_s8<B_PLANES>(img, p.in_plane, rb);
                __syncthreads();
                if (j == j0) { WG_STAMP(1); } else if (j == j0 + 1) { WG_STAMP(3); }
                if (j + 1 < j1) request(j + 1);  // travels under this image's K steps
                k_steps();
                if (j == j0) { WG_STAMP(2); } else if (j == j0 + 1) { WG_STAMP(4); }
            }
        }
    }
    if (!staged)
    for (int j = j0; j < j1; ++j) {
        __syncthreads();  // previous image fully consumed
        // ---- dz image: [npix_pad][PA], rows >= npix are zero (one "row" of npix_pad pixels for the walker) ----
        bool dz_done = false;
        if constexpr (U8) {
            if (p.prefetch) {  // (uniform) the dz chunks are requested in front of the frame fill: they travel with its frame-id round trip
                ImagePrefetch<WgradPrefetch<WV>::DZ_U8> ra;  // instead of costing one of their own (13.9 k cycles to stage an image before)
                request_image_s8<NTHR>(ra, p.dz + (int64_t)j * g.npix * g.cout_p, 1, g.npix, g.cout_p, 0, 0, 1, p.npix_pad, p.PA, tid,
                                       p.d_dzchunk, p.d_npixpad);
                fill_input_image<U8, PASSES, NTHR>(img, p.in_plane, g, p.fs, p.in, j, -g.pad, p.R, p.Wp, p.PPin, tid, p.d_chunk, p.d_Wp, p.d_R);
                commit_image_s8<A_PLANES>(dzi, p.dz_plane, ra);
                dz_done = true;
            }
        }
        if (!dz_done) {
            fill_image_s8<NTHR, A_PLANES, 8>(dzi, p.dz_plane, p.dz + (int64_t)j * g.npix * g.cout_p, 1, g.npix, g.cout_p, 0, 0, 1, p.npix_pad,
                                             p.PA, tid, p.d_dzchunk, p.d_npixpad);
            fill_input_image<U8, PASSES, NTHR>(img, p.in_plane, g, p.fs, p.in, j, -g.pad, p.R, p.Wp, p.PPin, tid, p.d_chunk, p.d_Wp, p.d_R);
        }
        __syncthreads();
        if (j == j0) { WG_STAMP(1); } else if (j == j0 + 1) { WG_STAMP(3); }
        k_steps();
        if (j == j0) { WG_STAMP(2); } else if (j == j0 + 1) { WG_STAMP(4); }
    }

    float* slab = p.slabs + (int64_t)ig * g.cout_p * g.K;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const int col = cg * p.NC + (wave + WV * t) * 16 + li;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = mt * 16 + grp * 4 + r;
#if !defined(ISDQN_NO_STREAMING)
                if (row < g.cout_p && col < g.K) __builtin_nontemporal_store(acc[mt][t][r] * p.scale, (ISDQN_GLOBAL float*)(slab + (int64_t)row * g.K + col));
#else
                if (row < g.cout_p && col < g.K) slab[(int64_t)row * g.K + col] = acc[mt][t][r] * p.scale;
#endif
            }
        }
#if defined(ISDQN_DEV)
    if (p.stamps != nullptr) {
        __builtin_amdgcn_s_waitcnt(0);
        WG_STAMP(5);
    }
#endif
#undef WG_STAMP
}

template <int MT, int NTW, int PASSES, bool U8, int WV = 4>
static int launch_conv_wgrad_img(const ConvWgradImgParams& p, int n_img_groups, hipStream_t st) {
    constexpr int A_PLANES = PASSES >= 2 ? 2 : 1;
    constexpr int B_PLANES = (PASSES >= 3) ? 2 : 1;
    const int lds = (A_PLANES * p.dz_plane + B_PLANES * p.in_plane) * 2;
    static LdsConfigured configured;
    if (int rc = ensure_dynamic_lds(&conv_wgrad_img_kernel<MT, NTW, PASSES, U8, WV>, lds, configured)) return rc;
    ISDQN_REPORT_OCCUPANCY((&conv_wgrad_img_kernel<MT, NTW, PASSES, U8, WV>), 64 * WV, lds, n_img_groups * p.n_col_groups);
    ConvWgradImgParams q = p;
    q.prefetch = 0;
#if !defined(ISDQN_WGRAD_NO_PREFETCH)
    if (!U8 && p.G >= 2 && p.npix_pad * (p.g.cout_p / 8) <= WgradPrefetch<WV>::DZ * 64 * WV &&
        p.R * p.Wp * (p.g.cin_p / 8) <= WgradPrefetch<WV>::IN * 64 * WV)
        q.prefetch = 1;
#if !defined(ISDQN_WGRAD_NO_U8_DZ)
    if (U8 && p.npix_pad * (p.g.cout_p / 8) <= WgradPrefetch<WV>::DZ_U8 * 64 * WV) q.prefetch = 1;
#endif
#endif
    hipLaunchKernelGGL((conv_wgrad_img_kernel<MT, NTW, PASSES, U8, WV>), dim3(n_img_groups * p.n_col_groups),
                       dim3(64 * WV), lds, st, q);
    ISDQN_HIP_CHECK(hipGetLastError());
    return ISDQN_OK;
}


// =====================================================================================================
// Image-resident data gradient with the LayerNorm + ReLU backward of the layer below fused in
// =====================================================================================================
//   da[iy,ix,ci] = sum_{jy,jx,co} dz[(iy+pad-ky)/s, (ix+pad-kx)/s, co] * W[co][ky][kx][ci],  ky = py + s*jy
// One workgroup = (image, stride-parity class, 128-pixel tile of that class).  The dz image of the layer
// ([hout][wout][cout_p], zero border of T-1 pixels top/left and `bh` bottom/right) sits in LDS; the B fragments
// of all taps are ds_reads at a per-lane offset; only the weight K-slices stream (TR image: the contraction
// index (tap, co) is the slow one of W[co][tap][ci]).  M = input channels, so each input pixel's channels sit in
// 4 lanes x MT*4 registers and the backward of  a = relu(LN(z))  of the layer below runs in the epilogue:
//   xhat = (z-mean)*rstd ; dy = da*[xhat*gamma+beta > 0] ; g = dy*gamma
//   dz_in = rstd*(g - mean(g) - xhat*mean(g*xhat)) ; dgamma += dy*xhat ; dbeta += dy ; dbias += dz_in
// `da` never goes to HBM.  Per-workgroup partial sums of (dgamma, dbeta, dbias) go to part[wg][3][cin_p].
struct ConvDgradImgParams {
    ConvGeom g;          // geometry of THIS conv layer (dz is its output gradient, da its input gradient)
    const float* W;      // [cout_p][taps][cin_p], S8 mirror
    const float* dz;     // [n_img][hout][wout][cout_p]
    const float* z_in;   // pre-LayerNorm output of the layer below  [n_img][hin][win][cin_p]
    const float *gamma, *beta;  // LayerNorm of the layer below (nullptr: no LayerNorm, ReLU only)
    int c_in;            // true channel count of the layer below
    float* dz_in;        // [n_img][hin][win][cin_p]
    float* part;         // [n_wg][3][cin_p]
    int n_img, T, Kc;    // taps per dim per class, K of a class = T*T*cout_p
    int Hd, Wd, PPd, bt; // padded dz image rows / cols, pixel pitch (elements), top/left border
    FastDiv d_chunk, d_Wd, d_T;  // fill index math (chunks per pixel, padded width) and taps per axis of a class
    long long* stamps;           // profiling only (isdqn_debug_set_stamps), as in ConvImgParams
    int dz_plane;
    int tiles_per_img;   // sum over classes of ceil(class pixels / tile pixels), tile = 64 * NT
    int cls_tile_start[5];
    PixelOrder cls_order[4];  // per class: position inside the class -> (row, column) of the class grid (8-wide strips)
    int n_classes;
};

// (second launch-bound: two waves per SIMD, i.e. at most 256 registers -- the kernel sits right at that edge and two
// workgroups per CU, its own or a weight-gradient one, are worth more than the last two registers)
// NT = 16-pixel column tiles per wave: a workgroup covers 64 * NT pixels of a class.  NT = 1 halves the tile so that a batch
// whose images would otherwise leave one workgroup (four waves) per CU launches two per image (net_plan.h: dgi_tile_pix).
template <int MT, int PASSES, int NT = 2>
__global__ __launch_bounds__(GEMM_THREADS, 2) void conv_dgrad_img_kernel(const ConvDgradImgParams p) {
    ISDQN_EMPTY_KERNEL_RETURN
    constexpr int BM = MT * 16;
    constexpr int A_PLANES = PASSES >= 2 ? 2 : 1;
    constexpr int B_PLANES = PASSES >= 3 ? 2 : 1;
    using GA = TileGeom<BM, true>;  // weights: TR image [32 k][BM ci]
    constexpr int A_STAGE = A_PLANES * GA::ELEMS;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    __bf16* a_stage = reinterpret_cast<__bf16*>(smem_raw);
    __bf16* img = a_stage + 2 * A_STAGE;
    __shared__ float s_part[4][3][64];
    const ConvGeom& g = p.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, grp = lane >> 4;
#if defined(ISDQN_DEV)
#define ISDQN_STAMP(i)                                                                               \
    if (p.stamps != nullptr && threadIdx.x == 0) {                                                  \
        p.stamps[(int64_t)blockIdx.x * 8 + (i)] = (long long)__builtin_amdgcn_s_memtime();           \
        if ((i) == 0) p.stamps[(int64_t)blockIdx.x * 8 + 7] = (long long)__builtin_amdgcn_s_memrealtime(); \
    }
#define ISDQN_ABLATED(bit) ((p.ablate & (bit)) != 0)
#else
#define ISDQN_STAMP(i)
#define ISDQN_ABLATED(bit) (false)
#endif
    ISDQN_STAMP(0);

    int j, tl;
    xcd_image_tile((int)blockIdx.x, p.n_img, p.tiles_per_img, j, tl);
    int cls = 0;
    while (cls + 1 < p.n_classes && tl >= p.cls_tile_start[cls + 1]) ++cls;
    const int q0 = (tl - p.cls_tile_start[cls]) * (64 * NT);  // first class-local pixel of this tile
    const int smask = g.stride - 1;  // (stride is a power of two: shifts and masks instead of divisions)
    const int cy = cls >> g.stride_sh, cx = cls & smask;
    const int Ha = (g.hin - cy + g.stride - 1) >> g.stride_sh, Wb = (g.win - cx + g.stride - 1) >> g.stride_sh;
    const int n_cls_pix = Ha * Wb;
    const int py = (cy + g.pad) & smask, px = (cx + g.pad) & smask;

    // LayerNorm parameters of the layer below for the epilogue: requested now (one float per thread), parked in LDS
    // after the fill (see conv_fwd_img_kernel)
    __shared__ __attribute__((aligned(16))) float s_gb[2][64];
    float par_v = 0.f;
    {
        const int which = tid >> 6, ch = tid & 63;
        const float* src = which == 0 ? p.gamma : p.beta;
        const bool ok = tid < 128 && ch < g.cin_p && p.gamma != nullptr;
        ISDQN_BOUNDS_CHECK(ok ? src + ch : zero_chunk(), 4, 13);
        par_v = *(const ISDQN_GLOBAL float*)(ok ? src + ch : zero_chunk());
    }

    const int nsteps = (p.Kc + GEMM_BK - 1) / GEMM_BK;
    const int k_last = p.Kc - 8;
    // ---- weight K-slice staging: TR image [32 k][BM ci], chunk = 8 consecutive ci of one (tap, co) ----
    constexpr int A_PER = GA::PER_THREAD;
    int a_ci0[A_PER], a_kk[A_PER], a_lds[A_PER];
    bool a_on[A_PER];
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
        int c = tid + i * GEMM_THREADS;
        a_on[i] = c < GA::CHUNKS;
        if (!a_on[i]) c = 0;
        const int kk = c / (BM / 8), rc = c % (BM / 8);
        a_ci0[i] = rc * 8; a_kk[i] = kk;
        a_lds[i] = tr_row(kk) * GA::PITCH + rc * 8;
    }
    constexpr int PF = 4;  // weight slices in flight (register ring), as in conv_fwd_img_kernel
    float sa[PF][A_PER][8];
    auto fetch = [&](int slot, int k0) {
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const int k = k0 + a_kk[i];
            const bool ok = (k < p.Kc) && (a_ci0[i] < g.cin_p);
            uint32_t jt, co;
            g.d_coutp.divmod(ok ? (uint32_t)k : 0u, jt, co);
            uint32_t jy_u, jx_u;
            p.d_T.divmod(jt, jy_u, jx_u);
            const int jy = (int)jy_u, jx = (int)jx_u;
            const int ky = py + g.stride * jy, kx = px + g.stride * jx;
            load8_aligned(ok ? p.W + ((int64_t)co * g.K + (ky * g.ksz + kx) * g.cin_p + a_ci0[i]) : zero_chunk(), sa[slot][i]);
        }
    };
    auto stash = [&](int slot, int stage) {
        __bf16* a_hi = a_stage + stage * A_STAGE;
        __bf16* a_lo = a_hi + GA::ELEMS;
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            if (GA::CHUNKS % GEMM_THREADS != 0 && !a_on[i]) continue;  // (whole multiples: no guard, no branch in the K loop)
            bf16x8 hi, lo;  // the weights come from the S8 mirror: staged by copy
            if constexpr (PASSES >= 2) {
                s8_unpack(sa[slot][i], hi, lo);
                *reinterpret_cast<bf16x8*>(a_lo + a_lds[i]) = lo;
            } else {
                s8_unpack_hi(sa[slot][i], hi);
            }
            *reinterpret_cast<bf16x8*>(a_hi + a_lds[i]) = hi;
        }
    };

    // K-step order of this workgroup (workgroups of one XCD start at different slices) and the first PF weight
    // slices, requested before the image fill so that they travel under it
    const int nsteps_p = (nsteps + PF - 1) / PF * PF;
    const int rot = (int)((blockIdx.x >> 3) % (unsigned)nsteps);
    auto slice = [&](int s) {
        const int k = s + rot;
        return s < nsteps ? (k >= nsteps ? k - nsteps : k) : nsteps_p;
    };
#pragma unroll
    for (int d = 0; d < PF; ++d) fetch(d, slice(d) * GEMM_BK);

    // ---- dz image of this sample into LDS (zero border of bt pixels top / left, the rest bottom / right) ----
    fill_image_s8<GEMM_THREADS, B_PLANES, 8>(img, p.dz_plane, p.dz + (int64_t)j * g.hout * g.wout * g.cout_p, g.hout, g.wout, g.cout_p,
                                             -p.bt, -p.bt, p.Hd, p.Wd, p.PPd, tid, p.d_chunk, p.d_Wd);

    ISDQN_STAMP(1);  // dz image staged (this wave)
    // ---- per-lane pixel of the two column tiles ----
    int b_org[NT], pix_iy[NT], pix_ix[NT];
    bool pix_ok[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        int q = q0 + wave * (16 * NT) + nt * 16 + column_slot(lane & 15);
        pix_ok[nt] = q < n_cls_pix;
        q = pix_ok[nt] ? q : n_cls_pix - 1;
        int a, b;
        p.cls_order[cls].map(q, a, b);
        const int iy = cy + g.stride * a, ix = cx + g.stride * b;
        pix_iy[nt] = iy; pix_ix[nt] = ix;
        const int oyb = (iy + g.pad - py) >> g.stride_sh, oxb = (ix + g.pad - px) >> g.stride_sh;  // (non-negative multiples of the stride)
        b_org[nt] = ((oyb + p.bt) * p.Wd + oxb + p.bt) * p.PPd;
    }

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) mfma_init(acc[mt][nt]);

    struct Frags {
        bf16x8 a_hi[MT], a_lo[MT], b_hi[NT], b_lo[NT];
    };
    auto read_frags = [&](int stage, int kk, Frags& f) {
        const __bf16* a_hi = a_stage + stage * A_STAGE;
        const __bf16* a_lo = a_hi + GA::ELEMS;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            f.a_hi[mt] = read_frag<true, GA::PITCH>(a_hi, mt * 16, lane);
            if constexpr (PASSES >= 2) f.a_lo[mt] = read_frag<true, GA::PITCH>(a_lo, mt * 16, lane);
        }
        int kq = kk * GEMM_BK + grp * 8;
        kq = kq < k_last ? kq : k_last;
        uint32_t jt, co;
        g.d_coutp.divmod((uint32_t)kq, jt, co);
        uint32_t jy_u, jx_u;
        p.d_T.divmod(jt, jy_u, jx_u);
        const int jy = (int)jy_u, jx = (int)jx_u;
        const int tap_off = -(jy * p.Wd + jx) * p.PPd + (int)co;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const __bf16* src = img + b_org[nt] + tap_off;
            f.b_hi[nt] = *reinterpret_cast<const bf16x8*>(src);
            if constexpr (PASSES >= 3) f.b_lo[nt] = *reinterpret_cast<const bf16x8*>(src + p.dz_plane);
        }
    };
    auto mfma_step = [&](const Frags& f) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if constexpr (PASSES >= 3)
                    mfma_acc(acc[mt][nt], f.a_hi[mt], f.b_lo[nt]);
                if constexpr (PASSES >= 2)
                    mfma_acc(acc[mt][nt], f.a_lo[mt], f.b_hi[nt]);
                mfma_acc(acc[mt][nt], f.a_hi[mt], f.b_hi[nt]);
            }
    };

    // pre-activations of the layer below for the epilogue: requested now, they arrive under the K loop
    float zpre[NT][MT][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int64_t pixel = ((int64_t)j * g.hin + pix_iy[nt]) * g.win + pix_ix[nt];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int ch0 = mt * 16 + grp * 4;
            ISDQN_BOUNDS_CHECK(ch0 < g.cin_p ? p.z_in + pixel * g.cin_p + ch0 : zero_chunk(), 16, 12);
            const ISDQN_GLOBAL f32x4* zp = (const ISDQN_GLOBAL f32x4*)(ch0 < g.cin_p ? p.z_in + pixel * g.cin_p + ch0 : zero_chunk());
            const f32x4 zq = *zp;
            zpre[nt][mt][0] = zq[0]; zpre[nt][mt][1] = zq[1]; zpre[nt][mt][2] = zq[2]; zpre[nt][mt][3] = zq[3];
        }
    }

    // main loop: one barrier per K step, independent pieces between barriers (see conv_fwd_img_kernel)
    static_assert(PF % 2 == 0 && PF >= 4, "the fragment sets alternate with the step parity; steps 0..3 are pre-fetched");
    Frags fr[2];
    if (tid < 128) s_gb[tid >> 6][tid & 63] = par_v;
    stash(0, 0);
    stash(1, 1);
    fetch(0, slice(PF) * GEMM_BK);
    fetch(1, slice(PF + 1) * GEMM_BK);
    __syncthreads();
    ISDQN_STAMP(2);
    read_frags(0, slice(0), fr[0]);
    __syncthreads();  // every wave has read stage 0 before step 0 overwrites it with slice 2
    // hand-interleaved K step for the 64-channel layer (12 * NT MFMAs per step), as in conv_fwd_img_kernel
    constexpr bool INTERLEAVED = PASSES == 3 && MT == 4 && A_PER == 1 && GA::CHUNKS == GEMM_THREADS;
    auto tap_offset_of = [&](int kk) {  // dz-image offset of this lane's 8-channel chunk of K step kk
        int kq = kk * GEMM_BK + grp * 8;
        kq = kq < k_last ? kq : k_last;
        uint32_t jt, co, jy_u, jx_u;
        g.d_coutp.divmod((uint32_t)kq, jt, co);
        p.d_T.divmod(jt, jy_u, jx_u);
        return -((int)jy_u * p.Wd + (int)jx_u) * p.PPd + (int)co;
    };
    int tap_next = INTERLEAVED ? tap_offset_of(slice(1)) : 0;
    for (int s0 = 0; s0 < nsteps_p; s0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int s = s0 + u;
            if constexpr (INTERLEAVED) {
                const Frags& fc = fr[u & 1];
                Frags& fn = fr[(u + 1) & 1];
                const __bf16* na_hi = a_stage + ((s + 1) & 1) * A_STAGE;
                const __bf16* na_lo = na_hi + GA::ELEMS;
                const int tap_off = tap_next;  // of slice(s + 1)
                const int slot = (u + 2) % PF;  // (a constant once the step loop is unrolled)
                __bf16* st_hi = a_stage + (s & 1) * A_STAGE + a_lds[0];
                __bf16* st_lo = st_hi + GA::ELEMS;
                bf16x8 c_hi, c_lo;
                constexpr int NM = 12 * NT;
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int jm = 0; jm < NM; ++jm) {
                    const int pass = jm / (4 * NT), nt = (jm >> 2) % NT, mt = jm & 3;
                    mfma_acc(acc[mt][nt], pass == 1 ? fc.a_lo[mt] : fc.a_hi[mt], pass == 0 ? fc.b_lo[nt] : fc.b_hi[nt]);
                    if (jm < 4) fn.a_hi[jm] = read_frag<true, GA::PITCH>(na_hi, jm * 16, lane);
                    else if (jm < 8) fn.a_lo[jm - 4] = read_frag<true, GA::PITCH>(na_lo, (jm - 4) * 16, lane);
                    else if (jm < 8 + NT) fn.b_hi[jm - 8] = *reinterpret_cast<const bf16x8*>(img + b_org[jm - 8] + tap_off);
                    else if (jm < 8 + 2 * NT) fn.b_lo[jm - 8 - NT] = *reinterpret_cast<const bf16x8*>(img + b_org[jm - 8 - NT] + tap_off + p.dz_plane);
                    else if constexpr (NT == 2) {
                        if (jm == 12) s8_unpack(sa[slot][0], c_hi, c_lo);  // S8 mirror: no conversion
                        else if (jm == 16) *reinterpret_cast<bf16x8*>(st_lo) = c_lo;
                        else if (jm == 17) *reinterpret_cast<bf16x8*>(st_hi) = c_hi;
                        else if (jm == 18) fetch(slot, slice(s + 2 + PF) * GEMM_BK);
                        else if (jm == 19) tap_next = tap_offset_of(slice(s + 2));
                    } else {
                        if (jm == 10) {
                            s8_unpack(sa[slot][0], c_hi, c_lo);
                            *reinterpret_cast<bf16x8*>(st_lo) = c_lo;
                            *reinterpret_cast<bf16x8*>(st_hi) = c_hi;
                        } else if (jm == 11) {
                            fetch(slot, slice(s + 2 + PF) * GEMM_BK);
                            tap_next = tap_offset_of(slice(s + 2));
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                __syncthreads();
                continue;
            }
            read_frags((s + 1) & 1, slice(s + 1), fr[(u + 1) & 1]);
            mfma_step(fr[u & 1]);
            stash((u + 2) % PF, s & 1);
            fetch((u + 2) % PF, slice(s + 2 + PF) * GEMM_BK);
            constexpr int N_MFMA = MT * NT * (PASSES >= 3 ? 3 : PASSES);
#pragma unroll
            for (int i = 0; i < N_MFMA; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
                if (i < N_MFMA / 2) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // 1 DS read/write
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);  // 2 VALU
            }
            __syncthreads();
        }
    }

    ISDQN_STAMP(3);  // K loop done
    // ---- epilogue: LayerNorm + ReLU backward of the layer below, per input pixel (column) ----
    float ga[MT][4], be[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int ch0 = (mt * 16 + grp * 4) & 63;
        const float4 g4 = *reinterpret_cast<const float4*>(&s_gb[0][ch0]);
        const float4 e4 = *reinterpret_cast<const float4*>(&s_gb[1][ch0]);
        ga[mt][0] = g4.x; ga[mt][1] = g4.y; ga[mt][2] = g4.z; ga[mt][3] = g4.w;
        be[mt][0] = e4.x; be[mt][1] = e4.y; be[mt][2] = e4.z; be[mt][3] = e4.w;
    }
    float dg[MT][4], db[MT][4], dbias[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) dg[mt][r] = db[mt][r] = dbias[mt][r] = 0.f;
    const float inv_c = 1.f / (float)p.c_in;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int64_t pixel = ((int64_t)j * g.hin + pix_iy[nt]) * g.win + pix_ix[nt];
        float zv[MT][4];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) zv[mt][r] = zpre[nt][mt][r];
        float out[MT][4];
        if (p.gamma != nullptr) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool ok = (mt * 16 + grp * 4 + r) < p.c_in;
                    s1 += ok ? zv[mt][r] : 0.f;
                    s2 += ok ? zv[mt][r] * zv[mt][r] : 0.f;
                }
            s1 += __shfl_xor(s1, 16); s1 += __shfl_xor(s1, 32);
            s2 += __shfl_xor(s2, 16); s2 += __shfl_xor(s2, 32);
            const float mean = s1 * inv_c;
            const float rstd = rsqrtf(fmaxf(s2 * inv_c - mean * mean, 0.f) + 1e-6f);
            float xh[MT][4], gg[MT][4], m1 = 0.f, m2 = 0.f;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool ok = pix_ok[nt] && (mt * 16 + grp * 4 + r) < p.c_in;
                    xh[mt][r] = (zv[mt][r] - mean) * rstd;
                    const float y = xh[mt][r] * ga[mt][r] + be[mt][r];
                    const float dy = (ok && y > 0.f) ? acc[mt][nt][r] : 0.f;
                    dg[mt][r] += dy * xh[mt][r];
                    db[mt][r] += dy;
                    gg[mt][r] = dy * ga[mt][r];
                    m1 += gg[mt][r];
                    m2 += gg[mt][r] * xh[mt][r];
                }
            m1 += __shfl_xor(m1, 16); m1 += __shfl_xor(m1, 32);
            m2 += __shfl_xor(m2, 16); m2 += __shfl_xor(m2, 32);
            m1 *= inv_c;
            m2 *= inv_c;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool ok = pix_ok[nt] && (mt * 16 + grp * 4 + r) < p.c_in;
                    out[mt][r] = ok ? rstd * (gg[mt][r] - m1 - xh[mt][r] * m2) : 0.f;
                    dbias[mt][r] += out[mt][r];
                }
        } else {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool ok = pix_ok[nt] && (mt * 16 + grp * 4 + r) < p.c_in;
                    out[mt][r] = (ok && zv[mt][r] > 0.f) ? acc[mt][nt][r] : 0.f;
                    dbias[mt][r] += out[mt][r];
                }
        }
        if (pix_ok[nt]) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int ch0 = mt * 16 + grp * 4;
                if (ch0 < g.cin_p)  // dz of the layer below: S8 (lane rows 2q / 2q+1 hold the two halves of a group)
                    s8_store_quad_paired(p.dz_in + pixel * g.cin_p, ch0, out[mt][0], out[mt][1], out[mt][2], out[mt][3]);
            }
        }
    }
    // ---- partial sums: over the 16 pixel lanes of a group, then over the 4 waves ----
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            dg[mt][r] = row16_sum(dg[mt][r]);
            db[mt][r] = row16_sum(db[mt][r]);
            dbias[mt][r] = row16_sum(dbias[mt][r]);
            if ((lane & 15) == 0) {
                const int ch = mt * 16 + grp * 4 + r;
                s_part[wave][0][ch] = dg[mt][r];
                s_part[wave][1][ch] = db[mt][r];
                s_part[wave][2][ch] = dbias[mt][r];
            }
        }
    __syncthreads();
    for (int i = tid; i < 3 * g.cin_p; i += GEMM_THREADS) {
        const int which = (i >= g.cin_p) + (i >= 2 * g.cin_p), c = i - which * g.cin_p;
        p.part[((int64_t)blockIdx.x * 3 + which) * g.cin_p + c] =
            s_part[0][which][c] + s_part[1][which][c] + s_part[2][which][c] + s_part[3][which][c];
    }
    ISDQN_STAMP(4);  // epilogue stores issued
#if defined(ISDQN_DEV)
    if (p.stamps != nullptr) {
        __builtin_amdgcn_s_waitcnt(0);
        ISDQN_STAMP(5);
    }
#endif
#undef ISDQN_STAMP
#undef ISDQN_ABLATED
}

// Deterministic reduction of per-workgroup partial rows: out[c] = sum_r part[r][c].  One workgroup per 8
// columns; 32 row-lanes per column stride over the rows with 8 independent loads in flight, then a fixed
// shuffle/LDS tree (same order every run).  Up to REDUCE_MAX_JOBS independent reductions share one launch (only
// Adam reads the results, so the backward pass collects them and issues them once, off the dgrad chain).
constexpr int REDUCE_MAX_JOBS = 4;
struct ReduceJobs {
    const float* part[REDUCE_MAX_JOBS];
    float* out[REDUCE_MAX_JOBS];
    int n_rows[REDUCE_MAX_JOBS], width[REDUCE_MAX_JOBS];
    int block_start[REDUCE_MAX_JOBS + 1];  // first workgroup of each job
    int n;
};
__global__ __launch_bounds__(256) void reduce_rows_kernel(const ReduceJobs jobs) {
    ISDQN_EMPTY_KERNEL_RETURN
    __shared__ float s_red[32][8];
    int job = 0;
#pragma unroll
    for (int i = 1; i < REDUCE_MAX_JOBS; ++i)
        if (i < jobs.n && (int)blockIdx.x >= jobs.block_start[i]) job = i;
    const float* __restrict__ part = jobs.part[job];
    float* __restrict__ out = jobs.out[job];
    const int n_rows = jobs.n_rows[job], width = jobs.width[job];
    const int col = ((int)blockIdx.x - jobs.block_start[job]) * 8 + (threadIdx.x & 7), rl = threadIdx.x >> 3;  // rl in [0, 32)
    float s = 0.f;
    if (col < width) {
        int r = rl;
        for (; r + 7 * 32 < n_rows; r += 8 * 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = part[(int64_t)(r + u * 32) * width + col];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; r < n_rows; r += 32) s += part[(int64_t)r * width + col];
    }
    s_red[rl][threadIdx.x & 7] = s;
    __syncthreads();
    if (rl == 0 && col < width) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) t += s_red[k][threadIdx.x & 7];
        out[col] = t;
    }
}

template <int MT, int PASSES, int NT = 2>
static int launch_conv_dgrad_img(const ConvDgradImgParams& p, hipStream_t st) {
    constexpr int A_PLANES = PASSES >= 2 ? 2 : 1;
    constexpr int B_PLANES = PASSES >= 3 ? 2 : 1;
    using GA = TileGeom<MT * 16, true>;
    const int lds = (2 * A_PLANES * GA::ELEMS + B_PLANES * p.dz_plane) * 2;
    static LdsConfigured configured;
    if (int rc = ensure_dynamic_lds(&conv_dgrad_img_kernel<MT, PASSES, NT>, lds, configured)) return rc;
    ISDQN_REPORT_OCCUPANCY((&conv_dgrad_img_kernel<MT, PASSES, NT>), GEMM_THREADS, lds, p.n_img * p.tiles_per_img);
    hipLaunchKernelGGL((conv_dgrad_img_kernel<MT, PASSES, NT>), dim3(p.n_img * p.tiles_per_img), dim3(GEMM_THREADS), lds, st, p);
    ISDQN_HIP_CHECK(hipGetLastError());
    return ISDQN_OK;
}

}  // namespace isdqn
