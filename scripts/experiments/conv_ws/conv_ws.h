// Weights-stationary convolution kernels for gfx950 (round 3).
//
// The image-resident kernels (conv_img.h) give every workgroup one LDS-resident input image (bf16 hi + lo planes: as many
// bytes as the fp32 image) and stream the WEIGHTS through a barrier-synchronised LDS ring.  Their three phases -- image
// fill, K loop, epilogue -- run strictly one after the other, every workgroup of a launch in the same phase, at one or two
// workgroups per CU (the image is most of the 160 KB): profiles/round2 show K loops at ~55 % matrix-pipe utilisation that
// are only 40-50 % of a kernel's life.
//
// For the 64-output-channel layers of the torso the roles can be swapped.  The whole weight matrix of a layer, split bf16
// hi + lo, is 128-144 KB: it fits the LDS of a CU ONCE, read-only, for the whole launch.  Then
//   * one workgroup per CU of up to 16 waves loads it at kernel start (the only barrier of the kernel);
//   * every wave owns 16-pixel x all-channel output tiles of its own and streams their activation fragments STRAIGHT from
//     global memory into MFMA operand registers: a lane's fragment (8 consecutive channels of one pixel at one tap) is the
//     32 contiguous bytes of one S8 group (gemm_core.h), so there is no im2col gather, no LDS image, no conversion;
//   * waves never synchronise again: four waves per SIMD sit in different K steps / in their epilogues, so the fragment
//     loads of one hide under the MFMAs of the others, and there is no fill phase and no per-K-step barrier to line them up.
// Activations are re-read once per tap from L1 / L2 (4-9x the unique bytes, all cache hits: an image is 30-60 KB) instead of
// once from HBM into LDS; the matrix pipe, not L2 bandwidth, stays the binding resource.
//
// Work assignment is static (tile t -> workgroup t / WAVES ... no queues, no atomics): results are run-to-run identical.
#pragma once
#include "conv_img.h"

namespace isdqn {

// LDS image of the weights: [K step][plane][row (output channel) 0 .. MT*16)][32 k], 64 bytes per row.  A fragment read
// (ds_read_b128, lane l: row l & 15, 16-byte chunk l >> 4) is served in the 16-lane groups of MI355X_MICROARCH.md
// ({0-3, 12-15, 20-27}, ...); with 64-byte rows the 16-byte chunk c of row r is stored in slot c ^ X[(r >> 2) & 3],
// X = {0, 2, 3, 1}: every service group then touches 16 different bank quads (checked exhaustively: scripts/r3/ws_swizzle.py).
__device__ __forceinline__ constexpr int ws_slot(int row, int chunk) {
    return chunk ^ ((0x1320 >> (((row >> 2) & 3) * 4)) & 3);  // X = {0, 2, 3, 1}
}

struct ConvWsParams {
    ConvGeom g;
    const float* W;     // S8 mirror [cout_p][K]
    const float* in;    // S8 activations [n_img][hin][win][cin_p]
    const float *bias, *gamma, *beta;
    float* act;         // S8 [n_img][npix][cout_p]
    float* z;           // fp32 pre-LayerNorm values of the first z_img images
    int n_img, z_img;
    int tiles_per_img;  // ceil(npix / (16 * NT))
    int n_tiles;
    unsigned in_bytes;  // extent of `in` (buffer bounds: fragments of taps outside the image read zeros through an offset past it)
    FastDiv d_tpi;      // by tiles_per_img
    long long* stamps;  // development builds (isdqn_debug_set_stamps): [wave][8] s_memtime at phase boundaries
    int ablate;         // development builds (env ISDQN_ABLATE): 1 no weight fill, 2 no fragment loads, 4 no stores, 8 no MFMA / LDS reads, 16 return at once, 32 prologue only
};

typedef __attribute__((ext_vector_type(4))) unsigned ws_u32x4;
constexpr unsigned WS_OOB = 0x80000000u;  // byte offset past every activation tensor (they are < 2 GB): the buffer load returns zeros

// Forward convolution + bias + LayerNorm(channels) + ReLU (dqn.py:62-65, 69-72), cin_p = 32 or 64 so that every 32-deep K
// step lies inside one tap.  WAVES waves per workgroup, one workgroup per CU; a wave owns NT 16-pixel tiles at a time.
//
// The K loop is lean on purpose: with several waves per SIMD every instruction of a wave costs issue cycles of all of them
// (ablation, profiles/round3: the first version spent 10 of its 24 us on address arithmetic).  Per step a lane executes the
// MFMAs, the LDS fragment reads, two buffer loads per pixel tile and ~4 vector instructions:
//   * the tap / channel offset of a step is wave-uniform: scalar arithmetic;
//   * the lane's own offset is fixed per tile; whether its pixel exists at a tap is one bit of a per-tile mask;
//   * a fragment outside the image is not a branch and not an address select on a zero block: the buffer load gets an offset
//     beyond the tensor and the hardware returns zeros.
template <int MT, int PASSES, int WAVES, int NT>
__global__ __launch_bounds__(64 * WAVES) void conv_fwd_ws_kernel(const ConvWsParams p) {
    constexpr int PL = PASSES >= 2 ? 2 : 1;       // weight planes (hi [+ lo])
    constexpr int ROWS = MT * 16;
    constexpr int STEP_BYTES = PL * ROWS * 64;     // one K step of the LDS image
    constexpr int NTHR = 64 * WAVES;
    constexpr int PF = NT == 1 ? 3 : 4;           // K steps of activation fragments in flight (registers)
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    __shared__ __attribute__((aligned(16))) float s_par[3][64];
    const ConvGeom& g = p.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, grp = lane >> 4, li = lane & 15;
    const int nsteps = g.K >> 5;
    const int spt_sh = g.cin_p == 64 ? 1 : 0;  // K steps per tap: 1 or 2
#if defined(ISDQN_DEV)
#define WS_ABLATED(bit) ((p.ablate & (bit)) != 0)
#else
#define WS_ABLATED(bit) (false)
#endif
#if defined(ISDQN_DEV)
#define WS_STAMP(i)                                                                                                       \
    if (p.stamps != nullptr && lane == 0) {                                                                               \
        p.stamps[((int64_t)blockIdx.x * WAVES + wave) * 8 + (i)] = (long long)__builtin_amdgcn_s_memtime();                 \
        if ((i) == 0) p.stamps[((int64_t)blockIdx.x * WAVES + wave) * 8 + 7] = (long long)__builtin_amdgcn_s_memrealtime(); \
    }
#else
#define WS_STAMP(i)
#endif
    WS_STAMP(0);
    if (WS_ABLATED(16)) return;  // (launch cost of this grid / LDS footprint alone)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), (short)0, (int)p.in_bytes, 0x00020000);

    // ---- per-tile lane state: byte offset of (patch origin, channel chunk grp), validity mask over the taps ----
    unsigned lane_base[NT], tapmask[NT];
    int64_t out_pix[NT];
    bool pix_ok[NT];
    int img_j = 0;
    auto setup = [&](int t) {
        uint32_t j_u, tl_u;
        p.d_tpi.divmod((uint32_t)t, j_u, tl_u);
        img_j = (int)j_u;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int p0 = ((int)tl_u * NT + nt) * 16 + li;
            pix_ok[nt] = p0 < g.npix;
            const int pp = pix_ok[nt] ? p0 : g.npix - 1;  // lanes past the image compute a duplicate pixel; never stored
            uint32_t oy_u, ox_u;
            g.d_wout.divmod((uint32_t)pp, oy_u, ox_u);
            const int iy0 = (int)oy_u * g.stride - g.pad, ix0 = (int)ox_u * g.stride - g.pad;
            out_pix[nt] = (int64_t)img_j * g.npix + pp;
            lane_base[nt] = (unsigned)((((int64_t)img_j * g.hin + iy0) * g.win + ix0) * g.cin_p + grp * 8) * 4u;  // (wraps where the
            // origin lies outside the image: only used together with a tap whose pixel exists, and then the sum is in range)
            unsigned xm = 0, m = 0;
            for (int k = 0; k < g.ksz; ++k) xm |= ((unsigned)(ix0 + k) < (unsigned)g.win ? 1u : 0u) << k;
            for (int k = 0; k < g.ksz; ++k) m |= ((unsigned)(iy0 + k) < (unsigned)g.hin ? xm : 0u) << (k * g.ksz);
            tapmask[nt] = WS_ABLATED(2) ? 0u : m;
        }
    };
    ws_u32x4 bh[PF][NT], bl[PF][NT];
    auto fetch = [&](int slot, int s) {  // s is wave-uniform: everything but the last two lines is scalar
        const bool son = s < nsteps;
        const int tap = son ? s >> spt_sh : 0;
        uint32_t ky_u, kx_u;
        g.d_ksz.divmod((uint32_t)tap, ky_u, kx_u);
        const unsigned step_off = (unsigned)((((int)ky_u * g.win + (int)kx_u) << (5 + spt_sh)) + ((s - (tap << spt_sh)) << 5)) * 4u;
        const unsigned bit = son ? 1u << tap : 0u;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const unsigned off = (tapmask[nt] & bit) ? lane_base[nt] + step_off : WS_OOB;
            bh[slot][nt] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
            if constexpr (PASSES >= 3) bl[slot][nt] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 16, 0, 0);
        }
    };

    int t = (int)blockIdx.x * WAVES + wave;
    const int t_stride = (int)gridDim.x * WAVES;
    if (t < p.n_tiles) {  // the first tile's fragments travel under the weight fill
        setup(t);
#pragma unroll
        for (int d = 0; d < PF; ++d) fetch(d, d);
    }

    WS_STAMP(1);  // first tile's fragments requested
    // ---- weights -> LDS, once: 32-byte S8 groups of the mirror, hi half to plane 0, lo half to plane 1 ----
    if (!WS_ABLATED(1)) {
        const int groups_per_row = g.K >> 3, n_groups = g.cout_p * groups_per_row;
        const FastDiv d_gpr((uint32_t)groups_per_row);
        constexpr int BATCH = 5;
        for (int c0 = tid; c0 < n_groups; c0 += NTHR * BATCH) {
            float v[BATCH][8];
            int dst[BATCH];
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int c = c0 + u * NTHR;
                const bool on = c < n_groups;
                uint32_t row, gq;
                d_gpr.divmod((uint32_t)(on ? c : 0), row, gq);
                load8_aligned(on ? p.W + (int64_t)row * g.K + gq * 8 : zero_chunk(), v[u]);
                dst[u] = on ? (int)(gq >> 2) * STEP_BYTES + (int)row * 64 + ws_slot((int)row, (int)(gq & 3)) * 16 : -1;
            }
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                if (dst[u] < 0) continue;
                bf16x8 hi, lo;
                if constexpr (PASSES >= 2) {
                    s8_unpack(v[u], hi, lo);
                    *reinterpret_cast<bf16x8*>(smem_raw + dst[u] + ROWS * 64) = lo;
                } else {
                    s8_unpack_hi(v[u], hi);
                }
                *reinterpret_cast<bf16x8*>(smem_raw + dst[u]) = hi;
            }
        }
        // rows cout_p .. ROWS of a channel tile that is not full are never written: they only feed accumulator rows that the
        // epilogue drops, and an accumulator row depends on its own A row only
    }
    if (tid < 192) {
        const int which = tid >> 6, ch = tid & 63;
        const float* src = which == 0 ? p.bias : which == 1 ? p.gamma : p.beta;
        s_par[which][ch] = (src != nullptr && ch < g.cout) ? src[ch] : (which == 1 ? 1.f : 0.f);
    }
    __syncthreads();  // the only barrier: the LDS image is read-only from here on
    WS_STAMP(2);

    const float inv_c = 1.0f / (float)g.cout;
    // this lane's A-fragment address inside K step 0 of the LDS image (plane 0, channel tile 0)
    const char* a_lane = smem_raw + li * 64 + ws_slot(li, grp) * 16;  // (ws_slot depends on the row inside the tile only)

    if (WS_ABLATED(32)) return;  // (prologue alone)
    while (t < p.n_tiles) {
        f32x4 acc[MT][NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) mfma_init(acc[mt][nt]);

        for (int s0 = 0; s0 < nsteps; s0 += PF) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int s = s0 + u;  // (steps past nsteps multiply zero fragments: no branch in the loop body)
                const char* a_step = a_lane + (s < nsteps ? s : 0) * STEP_BYTES;
                bf16x8 b_hi[NT], b_lo[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    b_hi[nt] = __builtin_bit_cast(bf16x8, bh[u][nt]);
                    if constexpr (PASSES >= 3) b_lo[nt] = __builtin_bit_cast(bf16x8, bl[u][nt]);
                }
                fetch(u, s + PF);
                if (WS_ABLATED(8)) {
                    acc[0][0][0] += (float)b_hi[0][0];
                    continue;
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const bf16x8 a_hi = *reinterpret_cast<const bf16x8*>(a_step + mt * 1024);
                    bf16x8 a_lo;
                    if constexpr (PASSES >= 2) a_lo = *reinterpret_cast<const bf16x8*>(a_step + mt * 1024 + ROWS * 64);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        if constexpr (PASSES >= 3) mfma_acc(acc[mt][nt], a_hi, b_lo[nt]);
                        if constexpr (PASSES >= 2) mfma_acc(acc[mt][nt], a_lo, b_hi[nt]);
                        mfma_acc(acc[mt][nt], a_hi, b_hi[nt]);
                    }
                }
            }
        }
        WS_STAMP(3);  // K loop done (the last tile's, where a wave takes several)
        // what the epilogue needs of this tile, then the next tile's first fragments are requested: they travel under the epilogue
        int64_t cur_pix[NT];
        bool cur_ok[NT];
        const bool cur_z = img_j < p.z_img;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) { cur_pix[nt] = out_pix[nt]; cur_ok[nt] = pix_ok[nt]; }
        t += t_stride;
        if (t < p.n_tiles) {
            setup(t);
#pragma unroll
            for (int d = 0; d < PF; ++d) fetch(d, d);
        }

        // ---- epilogue: bias + LayerNorm over channels + ReLU (same math as conv_fwd_img_kernel); its parameters are read
        //      from LDS here rather than held in 48 registers through the K loop ----
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            float zv[MT][4], s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const float4 b4 = *reinterpret_cast<const float4*>(&s_par[0][mt * 16 + grp * 4]);
                const float bi[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ch = mt * 16 + grp * 4 + r;
                    const float zz = ch < g.cout ? acc[mt][nt][r] + bi[r] : 0.f;
                    zv[mt][r] = zz;
                    s1 += zz;
                    s2 += zz * zz;
                }
            }
            float mean = 0.f, rstd = 1.f;
            if (p.gamma != nullptr) {
                s1 += __shfl_xor(s1, 16); s1 += __shfl_xor(s1, 32);
                s2 += __shfl_xor(s2, 16); s2 += __shfl_xor(s2, 32);
                mean = s1 * inv_c;
                rstd = rsqrtf(fmaxf(s2 * inv_c - mean * mean, 0.f) + 1e-6f);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int ch0 = mt * 16 + grp * 4;
                const float4 g4 = *reinterpret_cast<const float4*>(&s_par[1][ch0]);
                const float4 e4 = *reinterpret_cast<const float4*>(&s_par[2][ch0]);
                const float ga[4] = {g4.x, g4.y, g4.z, g4.w}, be[4] = {e4.x, e4.y, e4.z, e4.w};
                float a[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float y = p.gamma != nullptr ? (zv[mt][r] - mean) * (rstd * ga[r]) + be[r] : zv[mt][r];
                    a[r] = (ch0 + r < g.cout) ? fmaxf(y, 0.f) : 0.f;
                }
                if (ch0 < g.cout_p) {
                    // (all 64 lanes take part in the pair exchange; lanes of pixels past the image store nothing)
                    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
                    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
                    bf16x4 hi, lo;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const __bf16 h = (__bf16)a[i];
                        hi[i] = h;
                        lo[i] = (__bf16)(a[i] - (float)h);
                    }
                    const u32x2 h2 = __builtin_bit_cast(u32x2, hi), l2 = __builtin_bit_cast(u32x2, lo);
                    const auto r0 = __builtin_amdgcn_permlane16_swap(h2[0], l2[0], false, false);
                    const auto r1 = __builtin_amdgcn_permlane16_swap(h2[1], l2[1], false, false);
                    if (cur_ok[nt] && !WS_ABLATED(4)) {
                        char* gp = reinterpret_cast<char*>(p.act + cur_pix[nt] * g.cout_p + (ch0 & ~7)) + (ch0 & 4) * 4;
                        *reinterpret_cast<ws_u32x4*>(gp) = ws_u32x4{r0[0], r1[0], r0[1], r1[1]};
                        if (cur_z)
                            *reinterpret_cast<float4*>(p.z + cur_pix[nt] * g.cout_p + ch0) = float4{zv[mt][0], zv[mt][1], zv[mt][2], zv[mt][3]};
                    }
                }
            }
        }
    }
    WS_STAMP(4);  // epilogue stores issued
#if defined(ISDQN_DEV)
    if (p.stamps != nullptr) {
        __builtin_amdgcn_s_waitcnt(0);
        WS_STAMP(5);  // stores retired
    }
#endif
}

#undef WS_ABLATED
#undef WS_STAMP

// LDS bytes of the weight image; 0 if the layer does not qualify (host-side check)
static inline int conv_ws_lds_bytes(const ConvGeom& g, int passes) {
    if ((g.cin_p != 32 && g.cin_p != 64) || g.cout_p > 64 || g.K % 32 != 0 || g.ksz > 4) return 0;
    const int mt = g.cout_p <= 32 ? 2 : 4;
    return (g.K / 32) * (passes >= 2 ? 2 : 1) * mt * 16 * 64;
}

template <int MT, int PASSES, int WAVES, int NT>
static int launch_conv_fwd_ws(ConvWsParams p, int lds, hipStream_t st) {
    static LdsConfigured configured;
    if (int rc = ensure_dynamic_lds(&conv_fwd_ws_kernel<MT, PASSES, WAVES, NT>, lds, configured)) return rc;
    p.tiles_per_img = ceil_div(p.g.npix, 16 * NT);
    p.n_tiles = p.n_img * p.tiles_per_img;
    p.d_tpi = FastDiv((uint32_t)p.tiles_per_img);
    const int n_cu = 256;
    const int grid = min(n_cu, ceil_div(p.n_tiles, WAVES));
    ISDQN_REPORT_OCCUPANCY((&conv_fwd_ws_kernel<MT, PASSES, WAVES, NT>), 64 * WAVES, lds, grid);
    hipLaunchKernelGGL((conv_fwd_ws_kernel<MT, PASSES, WAVES, NT>), dim3(grid), dim3(64 * WAVES), lds, st, p);
    ISDQN_HIP_CHECK(hipGetLastError());
    return ISDQN_OK;
}

// =====================================================================================================
// Weights-stationary data gradient, fused with the LayerNorm + ReLU backward of the layer below (32-channel input layers)
// =====================================================================================================
//   da[j, iy, ix, ci] = sum_{taps of the pixel's stride class, co} dz[j, oy, ox, co] * W[co][ky][kx][ci]     (as ConvDgrad)
// The image-resident kernel of this job (conv_dgrad_img_kernel<2, 3>: data gradient of the stride-2 layer + LayerNorm backward
// of the first layer) is the least efficient kernel of the step: 40 us for 2 GFLOP -- four workgroups per image, each staging
// the whole dz image for 0.6 us of MFMA work.  Here the weights of ALL stride classes (64 x 16 taps x 32 ci, hi + lo: 128 KB)
// are LDS-resident per CU in the transposed-read layout, and a wave streams the dz fragments of its 16 input pixels from
// global memory: 8 K steps x 6 MFMAs per tile, then the LayerNorm backward of those 16 pixels in registers.  A wave takes
// several tiles and keeps the (dgamma, dbeta, dbias) partial sums in registers across them: one partial row per workgroup.
struct ConvDgradWsParams {
    ConvGeom g;          // geometry of THIS conv layer (dz is its output gradient, da its input gradient)
    const float* W;      // [cout_p][taps][cin_p], S8 mirror
    const float* dz;     // S8 [n_img][hout][wout][cout_p]
    const float* z_in;   // pre-LayerNorm output of the layer below [n_img][hin][win][cin_p]
    const float *gamma, *beta;
    int c_in;
    float* dz_in;        // S8 [n_img][hin][win][cin_p]
    float* part;         // [grid][3][cin_p]
    int n_img, T;        // taps per axis of a class
    int n_classes, tiles_per_img, n_tiles;
    int cls_tile_start[5];
    FastDiv cls_d_w[4];  // per class: by its grid width Wb
    FastDiv d_tpi, d_T;  // by tiles_per_img, by T
    unsigned dz_bytes;
    long long* stamps;
};

// LDS image: [class][K step][plane][32 k rows in tr_row order][MT * 16 ci], 64-byte rows (MT = 2).  A transposing read
// (ds_read_b64_tr_b16) is served in two 32-lane groups, each covering 8 consecutive LDS rows x one 32-byte window; with 64-byte
// rows, rows r and r + 4 would share banks, so the 32-byte half of a row is flipped where (r >> 2) & 1: conflict-free.
template <int PASSES, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void conv_dgrad_ws_kernel(const ConvDgradWsParams p) {
    constexpr int MT = 2, PL = PASSES >= 2 ? 2 : 1, RB = MT * 32;  // bytes per LDS row
    constexpr int STEP_BYTES = PL * 32 * RB;
    constexpr int NTHR = 64 * WAVES;
    constexpr int PF = 3;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    __shared__ __attribute__((aligned(16))) float s_gb[2][64];
    __shared__ float s_part[WAVES][3][MT * 16];
    const ConvGeom& g = p.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, grp = lane >> 4, li = lane & 15;
    const int T = p.T, spt_sh = g.cout_p == 64 ? 1 : 0, ksteps = (T * T) << spt_sh;  // K steps of one class
    const int smask = g.stride - 1;
#if defined(ISDQN_DEV)
#define WS_STAMP(i)                                                                                                       \
    if (p.stamps != nullptr && lane == 0) {                                                                               \
        p.stamps[((int64_t)blockIdx.x * WAVES + wave) * 8 + (i)] = (long long)__builtin_amdgcn_s_memtime();                 \
        if ((i) == 0) p.stamps[((int64_t)blockIdx.x * WAVES + wave) * 8 + 7] = (long long)__builtin_amdgcn_s_memrealtime(); \
    }
#else
#define WS_STAMP(i)
#endif
    WS_STAMP(0);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dz), (short)0, (int)p.dz_bytes, 0x00020000);

    // ---- weights of every class -> LDS (once).  One 32-byte S8 group of the mirror = (co, tap, 8 ci): hi -> plane 0, lo -> plane 1 ----
    {
        const int cig = g.cin_p >> 3, n_taps = g.ksz * g.ksz;  // ci groups per (co, tap)
        const int n_groups = g.cout_p * n_taps * cig;
        const FastDiv d_cig((uint32_t)cig), d_taps((uint32_t)n_taps), d_ksz((uint32_t)g.ksz);
        constexpr int BATCH = 4;
        for (int c0 = tid; c0 < n_groups; c0 += NTHR * BATCH) {
            float v[BATCH][8];
            int dst[BATCH];
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int c = c0 + u * NTHR;
                const bool on = c < n_groups;
                uint32_t ct, gq, co, tap, ky, kx;
                d_cig.divmod((uint32_t)(on ? c : 0), ct, gq);
                d_taps.divmod(ct, co, tap);
                d_ksz.divmod(tap, ky, kx);
                load8_aligned(on ? p.W + (int64_t)co * g.K + tap * g.cin_p + gq * 8 : zero_chunk(), v[u]);
                // class of this tap: (ky mod s, kx mod s) = (py, px); tap index inside the class (jy, jx) = (ky / s, kx / s)
                const int py = (int)ky & smask, px = (int)kx & smask, jy = (int)ky >> g.stride_sh, jx = (int)kx >> g.stride_sh;
                // the class whose pixels use parity (py, px): py = (cy + pad) mod s  ->  cy = (py - pad) mod s
                const int cy = (py - g.pad) & smask, cx = (px - g.pad) & smask, cls = cy * g.stride + cx;
                const int ks = ((jy * T + jx) << spt_sh) + ((int)co >> 5), R = tr_row((int)co & 31);
                dst[u] = on ? (cls * ksteps + ks) * STEP_BYTES + R * RB + ((((int)gq * 16)) ^ (32 * ((R >> 2) & 1))) : -1;
            }
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                if (dst[u] < 0) continue;
                bf16x8 hi, lo;
                if constexpr (PASSES >= 2) {
                    s8_unpack(v[u], hi, lo);
                    *reinterpret_cast<bf16x8*>(smem_raw + dst[u] + 32 * RB) = lo;
                } else {
                    s8_unpack_hi(v[u], hi);
                }
                *reinterpret_cast<bf16x8*>(smem_raw + dst[u]) = hi;
            }
        }
        if (tid < 128) {
            const int which = tid >> 6, ch = tid & 63;
            const float* src = which == 0 ? p.gamma : p.beta;
            s_gb[which][ch] = (src != nullptr && ch < p.c_in) ? src[ch] : (which == 0 ? 1.f : 0.f);
        }
    }
    __syncthreads();
    WS_STAMP(1);

    // this lane's transposed-read addresses inside one plane of one K step: k rows tr_row(8 * grp + q) and that + 8 LDS rows
    const int q4 = li >> 2, p4 = li & 3;
    const int R0 = 16 * (grp >> 1) + 2 * q4 + (grp & 1);
    const int a_flip = 32 * ((R0 >> 2) & 1);  // (the same for row R0 + 8)
    const char* a_lane = smem_raw + R0 * RB + 8 * p4;  // + (mt * 32) ^ a_flip

    float dg[MT][4], db[MT][4], dbias[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) dg[mt][r] = db[mt][r] = dbias[mt][r] = 0.f;
    const float inv_c = 1.f / (float)p.c_in;

    for (int t = (int)blockIdx.x * WAVES + wave; t < p.n_tiles; t += (int)gridDim.x * WAVES) {
        uint32_t j_u, tl_u;
        p.d_tpi.divmod((uint32_t)t, j_u, tl_u);
        const int j = (int)j_u, tl = (int)tl_u;
        int cls = 0;
        while (cls + 1 < p.n_classes && tl >= p.cls_tile_start[cls + 1]) ++cls;
        const int cy = cls >> g.stride_sh, cx = cls & smask;
        const int Ha = (g.hin - cy + g.stride - 1) >> g.stride_sh, Wb = (g.win - cx + g.stride - 1) >> g.stride_sh;
        const int py = (cy + g.pad) & smask, px = (cx + g.pad) & smask;
        const int qpix = (tl - p.cls_tile_start[cls]) * 16 + li;
        const bool pix_ok = qpix < Ha * Wb;
        uint32_t a_u, b_u;
        p.cls_d_w[cls].divmod((uint32_t)(pix_ok ? qpix : Ha * Wb - 1), a_u, b_u);
        const int iy = cy + g.stride * (int)a_u, ix = cx + g.stride * (int)b_u;
        const int oyb = (iy + g.pad - py) >> g.stride_sh, oxb = (ix + g.pad - px) >> g.stride_sh;
        const int64_t pixel = ((int64_t)j * g.hin + iy) * g.win + ix;
        const unsigned lane_base = (unsigned)((((int64_t)j * g.hout + oyb) * g.wout + oxb) * g.cout_p + grp * 8) * 4u;
        unsigned tapmask = 0;
        for (int jy = 0; jy < T; ++jy)
            for (int jx = 0; jx < T; ++jx)
                tapmask |= (((unsigned)(oyb - jy) < (unsigned)g.hout && (unsigned)(oxb - jx) < (unsigned)g.wout) ? 1u : 0u) << (jy * T + jx);

        // pre-activations of the layer below for the epilogue: requested now, they arrive under the K loop
        f32x4 zq[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) zq[mt] = *(const ISDQN_GLOBAL f32x4*)(p.z_in + pixel * g.cin_p + mt * 16 + grp * 4);

        ws_u32x4 bh[PF], bl[PF];
        auto fetch = [&](int slot, int s) {
            const bool son = s < ksteps;
            const int tap = son ? s >> spt_sh : 0;
            uint32_t jy_u, jx_u;
            p.d_T.divmod((uint32_t)tap, jy_u, jx_u);
            const unsigned step_off = (unsigned)(-(((int)jy_u * g.wout + (int)jx_u) * g.cout_p) + ((s - (tap << spt_sh)) << 5)) * 4u;
            const unsigned bit = son ? 1u << tap : 0u;
            const unsigned off = (tapmask & bit) ? lane_base + step_off : WS_OOB;
            bh[slot] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
            if constexpr (PASSES >= 3) bl[slot] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 16, 0, 0);
        };
#pragma unroll
        for (int d = 0; d < PF; ++d) fetch(d, d);

        f32x4 acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) mfma_init(acc[mt]);
        const char* a_cls = a_lane + cls * ksteps * STEP_BYTES;
        for (int s0 = 0; s0 < ksteps; s0 += PF) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int s = s0 + u;
                const char* a_step = a_cls + (s < ksteps ? s : 0) * STEP_BYTES;
                const bf16x8 b_hi = __builtin_bit_cast(bf16x8, bh[u]);
                bf16x8 b_lo;
                if constexpr (PASSES >= 3) b_lo = __builtin_bit_cast(bf16x8, bl[u]);
                fetch(u, s + PF);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const char* a0 = a_step + ((mt * 32) ^ a_flip);
                    const bf16x8 a_hi = tr_frag(reinterpret_cast<const __bf16*>(a0), reinterpret_cast<const __bf16*>(a0 + 8 * RB));
                    if constexpr (PASSES >= 3) mfma_acc(acc[mt], a_hi, b_lo);
                    if constexpr (PASSES >= 2) {
                        const bf16x8 a_lo = tr_frag(reinterpret_cast<const __bf16*>(a0 + 32 * RB), reinterpret_cast<const __bf16*>(a0 + 40 * RB));
                        mfma_acc(acc[mt], a_lo, b_hi);
                    }
                    mfma_acc(acc[mt], a_hi, b_hi);
                }
            }
        }

        // ---- epilogue: LayerNorm + ReLU backward of the layer below for this lane's pixel (as conv_dgrad_img_kernel) ----
        float zv[MT][4], ga[MT][4], be[MT][4], out[MT][4];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const float4 g4 = *reinterpret_cast<const float4*>(&s_gb[0][mt * 16 + grp * 4]);
            const float4 e4 = *reinterpret_cast<const float4*>(&s_gb[1][mt * 16 + grp * 4]);
            ga[mt][0] = g4.x; ga[mt][1] = g4.y; ga[mt][2] = g4.z; ga[mt][3] = g4.w;
            be[mt][0] = e4.x; be[mt][1] = e4.y; be[mt][2] = e4.z; be[mt][3] = e4.w;
#pragma unroll
            for (int r = 0; r < 4; ++r) zv[mt][r] = zq[mt][r];
        }
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
                    const bool ok = pix_ok && (mt * 16 + grp * 4 + r) < p.c_in;
                    xh[mt][r] = (zv[mt][r] - mean) * rstd;
                    const float y = xh[mt][r] * ga[mt][r] + be[mt][r];
                    const float dy = (ok && y > 0.f) ? acc[mt][r] : 0.f;
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
                    const bool ok = pix_ok && (mt * 16 + grp * 4 + r) < p.c_in;
                    out[mt][r] = ok ? rstd * (gg[mt][r] - m1 - xh[mt][r] * m2) : 0.f;
                    dbias[mt][r] += out[mt][r];
                }
        } else {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool ok = pix_ok && (mt * 16 + grp * 4 + r) < p.c_in;
                    out[mt][r] = (ok && zv[mt][r] > 0.f) ? acc[mt][r] : 0.f;
                    dbias[mt][r] += out[mt][r];
                }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int ch0 = mt * 16 + grp * 4;
            // dz of the layer below: S8 (lane rows 2q / 2q+1 hold the two halves of a group); every lane takes part in the exchange
            typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
            typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
            bf16x4 hi, lo;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const __bf16 h = (__bf16)out[mt][i];
                hi[i] = h;
                lo[i] = (__bf16)(out[mt][i] - (float)h);
            }
            const u32x2 h2 = __builtin_bit_cast(u32x2, hi), l2 = __builtin_bit_cast(u32x2, lo);
            const auto r0 = __builtin_amdgcn_permlane16_swap(h2[0], l2[0], false, false);
            const auto r1 = __builtin_amdgcn_permlane16_swap(h2[1], l2[1], false, false);
            if (pix_ok && ch0 < g.cin_p) {
                char* gp = reinterpret_cast<char*>(p.dz_in + pixel * g.cin_p + (ch0 & ~7)) + (ch0 & 4) * 4;
                *reinterpret_cast<ws_u32x4*>(gp) = ws_u32x4{r0[0], r1[0], r0[1], r1[1]};
            }
        }
    }
    WS_STAMP(2);
    // ---- partial sums: over the 16 pixel lanes of a group, then over the waves (fixed order) ----
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            dg[mt][r] = row16_sum(dg[mt][r]);
            db[mt][r] = row16_sum(db[mt][r]);
            dbias[mt][r] = row16_sum(dbias[mt][r]);
            if (li == 0) {
                const int ch = mt * 16 + grp * 4 + r;
                s_part[wave][0][ch] = dg[mt][r];
                s_part[wave][1][ch] = db[mt][r];
                s_part[wave][2][ch] = dbias[mt][r];
            }
        }
    __syncthreads();
    for (int i = tid; i < 3 * g.cin_p; i += NTHR) {
        const int which = (i >= g.cin_p) + (i >= 2 * g.cin_p), c = i - which * g.cin_p;
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) sum += s_part[w][which][c];
        p.part[((int64_t)blockIdx.x * 3 + which) * g.cin_p + c] = sum;
    }
    WS_STAMP(3);
#undef WS_STAMP
}

template <int PASSES>
static int launch_conv_dgrad_ws(const ConvDgradWsParams& p, int lds, int grid, hipStream_t st) {
    constexpr int WAVES = 16;
    static LdsConfigured configured;
    if (int rc = ensure_dynamic_lds(&conv_dgrad_ws_kernel<PASSES, WAVES>, lds, configured)) return rc;
    ISDQN_REPORT_OCCUPANCY((&conv_dgrad_ws_kernel<PASSES, WAVES>), 64 * WAVES, lds, grid);
    hipLaunchKernelGGL((conv_dgrad_ws_kernel<PASSES, WAVES>), dim3(grid), dim3(64 * WAVES), lds, st, p);
    ISDQN_HIP_CHECK(hipGetLastError());
    return ISDQN_OK;
}

}  // namespace isdqn
