// First convolution (uint8 frames, architectures/dqn.py:51-58), two pixel tiles per workgroup with the second tile's frame rows
// PREFETCHED INTO REGISTERS under the first tile's K loop and epilogue.
//
// Why (profiles/round2/phase_stamps_c2.txt): conv_fwd_img_kernel<2, 2, true> runs 2048 workgroups of 9.5 us at four per
// CU -- two rounds -- and HALF of a workgroup's life is its image fill: the frame-id row, then eight 8-byte pixel loads per thread,
// two dependent round trips with nothing of its own to overlap them.  Here a workgroup owns tiles 2t and 2t + 1 of one image (the
// frame ids and base pointers are shared), requests tile 2t + 1's pixels right behind tile 2t's first barrier and converts them into
// the (then dead) LDS image after tile 2t's epilogue: one exposed fill per workgroup instead of two, one dispatch round instead
// of two (1024 workgroups at the headline size = the chip's four per CU).
//
// The arithmetic is conv_fwd_img_kernel's, statement for statement (same K-step rotation per (image, tile), same pass order, same
// epilogue): the outputs are bit-identical (tests/test_gpu_network.py::test_first_layer_pair_kernel_is_bit_identical_to_the_one_tile_kernel).
// uint8 layers of four stacked frames (K = 256: eight K steps) and at most 32 output channels (MT = 2) only; measured at the
// headline size: 28.2 -> 26.2 us, the step +1 % (profiles/round3/ab_u8_pair.txt, phase stamps: stamps_u8_pair.txt).
#pragma once
#include "conv_img.h"

namespace isdqn {

constexpr int U8P_PB = 2;  // positions per thread: one batch covers the whole tile image (R * Wp / 8 <= 2 * 256 positions)

template <int PASSES, int NSTEPS>
__global__ __launch_bounds__(GEMM_THREADS, 4) void conv_fwd_u8_pair_kernel(const ConvImgParams p) {
    ISDQN_EMPTY_KERNEL_RETURN
    constexpr int MT = 2, NT = 2, MTW = MT, NTHR = GEMM_THREADS, STACK_MAX = 4;
    using T = ConvImgTraits<MT, PASSES, true>;
    using GA = typename T::GA;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    __bf16* smem = reinterpret_cast<__bf16*>(smem_raw);
    __bf16* a_stage = smem;
    __bf16* img = smem + 2 * T::A_STAGE;
    const ConvGeom& g = p.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, grp = lane >> 4;
#if defined(ISDQN_DEV)
#define U8P_STAMP(i)                                                                                 \
    if (p.stamps != nullptr && threadIdx.x == 0) {                                                  \
        p.stamps[(int64_t)blockIdx.x * 8 + (i)] = (long long)__builtin_amdgcn_s_memtime();           \
        if ((i) == 0) p.stamps[(int64_t)blockIdx.x * 8 + 7] = (long long)__builtin_amdgcn_s_memrealtime(); \
    }
#else
#define U8P_STAMP(i)
#endif
    U8P_STAMP(0);
    const int T2 = p.tiles_per_img >> 1;  // tile pairs per image
    int j, tp;
    xcd_image_tile((int)blockIdx.x, p.n_img, T2, j, tp);

    __shared__ __attribute__((aligned(16))) float s_par[3][64];
    float par_v = 0.f;
    {
        const int which = tid >> 6, ch = tid & 63;
        const float* src = which == 0 ? p.bias : which == 1 ? p.gamma : p.beta;
        const bool ok = tid < 192 && ch < g.cout_p && src != nullptr;
        ISDQN_BOUNDS_CHECK(ok ? src + ch : zero_chunk(), 4, 13);
        par_v = *(const ISDQN_GLOBAL float*)(ok ? src + ch : zero_chunk());
    }
    constexpr int nsteps = NSTEPS;  // K / 32 = 2 * stacked frames, a compile-time constant: the K loop is straight-line code
    const int k_last = g.K - 8;
    constexpr int A_PER = (GA::CHUNKS + NTHR - 1) / NTHR;
    int a_row[A_PER], a_var[A_PER], a_lds[A_PER];
    bool a_on[A_PER];
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
        int c = tid + i * NTHR;
        a_on[i] = c < GA::CHUNKS;
        if (!a_on[i]) c = 0;
        a_row[i] = stash_row(c);
        a_var[i] = (c & 3) * 8;
        a_lds[i] = a_row[i] * GA::PITCH + (c & 3) * 8;
    }
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
            if (GA::CHUNKS % NTHR != 0 && !a_on[i]) continue;
            bf16x8 hi, lo;
            if constexpr (PASSES >= 2) {
                s8_unpack(sa[slot][i], hi, lo);
                *reinterpret_cast<bf16x8*>(a_lo + a_lds[i]) = lo;
            } else {
                s8_unpack_hi(sa[slot][i], hi);
            }
            *reinterpret_cast<bf16x8*>(a_hi + a_lds[i]) = hi;
        }
    };
    static_assert(NSTEPS >= PF + 2 && NSTEPS % 2 == 0, "ring of four slices + two staged ones");

    // ---- frame rows of one tile: request (registers only) / convert into the LDS image ----
    const FrameSrc& fs = p.fs;
    const uint8_t* base[STACK_MAX];
#pragma unroll
    for (int c = 0; c < STACK_MAX; ++c) {
        const int id = c < fs.stack ? fs.frame_id(j, c) : -1;
        base[c] = id >= 0 ? fs.frames + (int64_t)id * fs.stride : nullptr;
    }
    const uint8_t* zeros = reinterpret_cast<const uint8_t*>(zero_chunk());
    const int cpr = p.Wp / 8, per_plane = p.R * cpr, plane = p.R * p.Wp;
    unsigned long long raw[U8P_PB][STACK_MAX];  // the only registers a request keeps alive (shift and LDS offset are recomputed)
    auto request = [&](int row_base) {  // loads only: nothing here touches the loaded registers
#pragma unroll
        for (int k = 0; k < U8P_PB; ++k) {
            const int pos = k * NTHR + tid;
            const bool on = pos < per_plane;
            uint32_t lr, cx;
            p.d_chunk.divmod((uint32_t)(on ? pos : 0), lr, cx);
            const int iy = row_base + (int)lr, ix0 = (int)cx * 8 - g.pad;
            const bool ok = on && iy >= 0 && iy < fs.H && ix0 > -8 && ix0 < fs.W;
            const int ixc = min(max(ix0, 0), fs.W - 8);
            const int off = iy * fs.W + ixc;
#pragma unroll
            for (int c = 0; c < STACK_MAX; ++c) raw[k][c] = load_u64_unaligned((ok && base[c] != nullptr) ? base[c] + off : zeros);
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int k = 0; k < U8P_PB; ++k) {
            const int pos = k * NTHR + tid;
            const bool on = pos < per_plane;
            uint32_t lr, cx;
            p.d_chunk.divmod((uint32_t)(on ? pos : 0), lr, cx);
            const int ix0 = (int)cx * 8 - g.pad;
            const int sh = ix0 - min(max(ix0, 0), fs.W - 8);
            const int dst = (int)lr * p.Wp + (int)cx * 8;
#pragma unroll
            for (int c = 0; c < STACK_MAX; ++c) {
                if (c >= fs.stack) continue;
                float v[8];
                FrameSrc::patch8_cvt(raw[k][c], sh, v);
                bf16x8 hi;
                round8(v, hi);
                if (on) *reinterpret_cast<bf16x8*>(img + c * plane + dst) = hi;
            }
        }
    };
    auto tile_rows = [&](int tile) {  // global input row of local row 0 of this tile's image
        const int oy_min = (int)g.d_wout.div((uint32_t)(tile * 128));
        return oy_min * g.stride - g.pad;
    };

    struct Frags {
        bf16x8 a_hi[MTW], a_lo[MTW], b_hi[NT];
    };

    // The two tiles are two copies of straight-line code (compile-time `it`): a request under a run-time condition would end in a
    // control-flow merge, where the compiler has to assume the larger number of loads in flight for every later wait.
    auto do_tile = [&](auto it_c) {
        constexpr int it = decltype(it_c)::value;
        const int tile = 2 * tp + it;
        const int p0 = tile * 128;
        const int row_base = tile_rows(tile);
        // the K walk of conv_fwd_img_kernel's workgroup for (image j, tile): blockIdx / 8 there is (j / 8) * tiles + tile for the
        // images of the full groups of eight (xcd_image_tile), the plain workgroup index / 8 in the tail group
        const int full = (p.n_img / 8) * 8;
        const int old_wg8 = j < full ? (j >> 3) * p.tiles_per_img + tile : (j * p.tiles_per_img + tile) >> 3;
        const int rot = (int)((unsigned)old_wg8 % (unsigned)nsteps);
        auto slice = [&](int s) {  // (s < nsteps wherever it is called)
            const int k = s + rot;
            return k >= nsteps ? k - nsteps : k;
        };
#pragma unroll
        for (int d = 0; d < PF; ++d) fetch(d, slice(d) * GEMM_BK);
        if constexpr (it == 0) request(row_base);
        commit();  // (it == 1: the rows requested under the previous tile's K loop)
        U8P_STAMP(it == 0 ? 1 : 4);  // tile image written

        int b_org[NT], out_pix[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            int pp = p0 + wave * 32 + nt * 16 + (lane & 15);
            const bool in_img = pp < g.npix;
            pp = in_img ? pp : g.npix - 1;
            uint32_t oy_u, ox_u;
            g.d_wout.divmod((uint32_t)pp, oy_u, ox_u);
            const int oy = (int)oy_u, ox = (int)ox_u;
            out_pix[nt] = in_img ? oy * g.wout + ox : -1;
            const int ly0 = oy * g.stride - g.pad - row_base;
            const int lx0 = ox * g.stride;
            b_org[nt] = ly0 * p.Wp + lx0;
        }
        f32x4 acc[MTW][NT];
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) mfma_init(acc[mt][nt]);

        auto read_frags = [&](int stage, int kk, Frags& f) {
            const __bf16* a_hi = a_stage + stage * T::A_STAGE;
            const __bf16* a_lo = a_hi + GA::ELEMS;
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt) {
                f.a_hi[mt] = read_frag<false, GA::PITCH>(a_hi, mt * 16, lane);
                if constexpr (PASSES >= 2) f.a_lo[mt] = read_frag<false, GA::PITCH>(a_lo, mt * 16, lane);
            }
            int kq = kk * GEMM_BK + grp * 8;
            kq = kq < k_last ? kq : k_last;
            const int c = kq >> 6, ky = (kq >> 3) & 7;
            const int tap_off = (c * p.R + ky) * p.Wp;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const __bf16* src = img + b_org[nt] + tap_off;
                typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
                bf16x4 h0 = *reinterpret_cast<const bf16x4*>(src);
                bf16x4 h1 = *reinterpret_cast<const bf16x4*>(src + 4);
                f.b_hi[nt] = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
            }
        };
        auto mfma_step = [&](const Frags& f) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) {
                    if constexpr (PASSES >= 2) mfma_acc(acc[mt][nt], f.a_lo[mt], f.b_hi[nt]);
                    mfma_acc(acc[mt][nt], f.a_hi[mt], f.b_hi[nt]);
                }
        };
        Frags fr[2];
        if constexpr (it == 0) {
            if (tid < 192) s_par[tid >> 6][tid & 63] = par_v;
        }
        stash(0, 0);
        stash(1, 1);
        fetch(0, slice(PF) * GEMM_BK);
        fetch(1, slice(PF + 1) * GEMM_BK);
        __syncthreads();  // image and the first two weight slices visible
        read_frags(0, slice(0), fr[0]);
        __syncthreads();
        // conv_fwd_img_kernel's K loop, unrolled over the NSTEPS steps: no slice past the last one is requested (there: the zero
        // block), so after step NSTEPS - PF - 3 the loop issues no load and after step NSTEPS - 3 it waits for none.
        // vmcnt retires IN ORDER: the partner tile's rows may only be requested once every load this tile still waits for is
        // OLDER than them -- i.e. behind the stash of the last weight slice -- or the next wait of the ring would wait for
        // them too (the first version requested them in front of the loop: 27.3 us, exactly the two-round kernel's time).
#pragma unroll
        for (int s = 0; s < NSTEPS; ++s) {
            if (s + 1 < NSTEPS) read_frags((s + 1) & 1, slice(s + 1), fr[(s + 1) & 1]);
            mfma_step(fr[s & 1]);
            if (s + 2 < NSTEPS) stash((s + 2) % PF, s & 1);
            if (s + 2 + PF < NSTEPS) fetch((s + 2) % PF, slice(s + 2 + PF) * GEMM_BK);
            if constexpr (it == 0) {
                if (s == NSTEPS - 3) request(tile_rows(tile + 1));  // (the last stash is behind us: 2 steps + the epilogue cover the trip)
            }
            constexpr int N_MFMA = MTW * NT * (PASSES >= 3 ? 3 : PASSES);
#pragma unroll
            for (int i = 0; i < N_MFMA; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (i < N_MFMA / 2) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
            }
            __syncthreads();
        }

        U8P_STAMP(it == 0 ? 2 : 5);  // K loop done
        // ---- epilogue: bias + LayerNorm over channels + ReLU (conv_fwd_img_kernel's, KG = 1) ----
        float bi[MTW][4], ga[MTW][4], be[MTW][4];
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) {
            const int ch0 = (mt * 16 + grp * 4) & 63;
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
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int ch = mt * 16 + grp * 4 + r;
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
                    int ch0 = mt * 16 + grp * 4;
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
                    s8_store_quad_paired(p.act + pix * g.cout_p, ch0, a.x, a.y, a.z, a.w);
                    // (a nontemporal store of z -- not read before the backward -- was measured: 26.5 -> 28.0 us)
                    if (j < p.z_img) *reinterpret_cast<float4*>(p.z + pix * g.cout_p + ch0) = zq;
                }
            }
        }
        U8P_STAMP(it == 0 ? 3 : 6);  // epilogue stores issued
        // (the K loop ended with a barrier: every wave is done with the LDS image, the next tile's commit may overwrite it; the
        // weight stages are rewritten by that tile's stash(0, 0) / stash(1, 1), whose readers finished before the same barrier)
    };
    do_tile(std::integral_constant<int, 0>{});
    do_tile(std::integral_constant<int, 1>{});
#undef U8P_STAMP
}

template <int PASSES, int NSTEPS>
static int launch_conv_fwd_u8_pair(const ConvImgParams& p, hipStream_t st) {
    using T = ConvImgTraits<2, PASSES, true>;
    const int lds = (2 * T::A_STAGE + T::B_PLANES * p.plane_elems) * 2;
    static LdsConfigured configured;
    if (int rc = ensure_dynamic_lds(&conv_fwd_u8_pair_kernel<PASSES, NSTEPS>, lds, configured)) return rc;
    const int grid = p.n_img * (p.tiles_per_img / 2);
    ISDQN_REPORT_OCCUPANCY((&conv_fwd_u8_pair_kernel<PASSES, NSTEPS>), GEMM_THREADS, lds, grid);
    hipLaunchKernelGGL((conv_fwd_u8_pair_kernel<PASSES, NSTEPS>), dim3(grid), dim3(GEMM_THREADS), lds, st, p);
    ISDQN_HIP_CHECK(hipGetLastError());
    return ISDQN_OK;
}

}  // namespace isdqn
