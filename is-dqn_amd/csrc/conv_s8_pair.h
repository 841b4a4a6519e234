// Second convolution of the Nature torso (21x21x32 S8 activations -> 11x11x64, 4x4 / 2; architectures/dqn.py:62-65), TWO IMAGES per
// workgroup with the second image PREFETCHED INTO REGISTERS under the first one's last K steps, partial-sum exchange and epilogue.
//
// conv_fwd_img_kernel<4, 3, false, 2> holds one 512-thread workgroup per CU (its 24 x 27 x 40 hi + lo image is 117 KB of LDS with the
// weight stages), so the 512 images of a step are two strictly serial rounds, each paying its own image fill (7.1 k of a workgroup's
// 28.3 k cycles, profiles/round2/phase_stamps_c2.txt) with nothing to overlap it.  Same remedy as conv_u8_pair.h: the workgroup owns
// images 2w and 2w + 1 and requests image 2w + 1 (six 32-byte chunks per thread, 48 registers) behind the LAST weight-slice wait of
// image 2w's K loop -- vmcnt retires in order, so any later ring wait would wait for the request too -- which needs the K loop as
// straight-line code: K = 512 is a compile-time eight positions per K group, no padded fetches.
//
// The arithmetic is conv_fwd_img_kernel's, statement for statement (same K rotation per image, same hand-interleaved step, same
// exchange and epilogue): bit-identical outputs (tests/test_gpu_network.py against the -DISDQN_NO_U8_PAIR build, which leaves both
// pair kernels out).  Only for MT = 4, three passes, two K groups, one pixel tile per image, an image of at most 6 x 512 chunks and an
// even image count.  Measured at the headline size: 25.7 -> 24.6 us, the step +0.6 % (profiles/round3/ab_s8_pair.txt).
#pragma once
#include "conv_img.h"

namespace isdqn {

constexpr int S8P_BATCH = 6;  // chunks per thread: the whole LDS image (R * Wp * cin_p / 8 chunks) in one batch of 512 threads

template <int NPOS>
__global__ __launch_bounds__(GEMM_THREADS * 2, 1) void conv_fwd_s8_pair_kernel(const ConvImgParams p) {
    ISDQN_EMPTY_KERNEL_RETURN
    constexpr int MT = 4, PASSES = 3, KG = 2, NT = 2, MTW = MT, NTHR = GEMM_THREADS, NTHR_ALL = GEMM_THREADS * KG;
    using T = ConvImgTraits<MT, PASSES, false>;
    using GA = typename T::GA;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    __bf16* smem = reinterpret_cast<__bf16*>(smem_raw);
    const int tid_all = threadIdx.x, kg = tid_all / GEMM_THREADS;
    __bf16* a_stage = smem + kg * 2 * T::A_STAGE;
    __bf16* img = smem + KG * 2 * T::A_STAGE;
    const ConvGeom& g = p.g;
    const int tid = tid_all - kg * GEMM_THREADS, lane = tid & 63, wave = tid >> 6, grp = lane >> 4;
    const int row_base = -g.pad;  // one pixel tile per image: local row 0 is input row -pad

    __shared__ __attribute__((aligned(16))) float s_par[3][64];
    float par_v = 0.f;
    {
        const int which = tid >> 6, ch = tid & 63;
        const float* src = which == 0 ? p.bias : which == 1 ? p.gamma : p.beta;
        const bool ok = kg == 0 && tid < 192 && ch < g.cout_p && src != nullptr;
        ISDQN_BOUNDS_CHECK(ok ? src + ch : zero_chunk(), 4, 13);
        par_v = *(const ISDQN_GLOBAL float*)(ok ? src + ch : zero_chunk());
    }
    constexpr int nsteps = NPOS * KG;  // K / 32
    const int k_last = g.K - 8;
    static_assert(GA::CHUNKS == NTHR, "one weight chunk per thread and K step");
    const int a_row = stash_row(tid), a_var = (tid & 3) * 8, a_lds = a_row * GA::PITCH + (tid & 3) * 8;
    constexpr int PF = 4;
    static_assert(NPOS >= PF + 2, "ring of four slices + two staged ones");
    float sa[PF][1][8];
    auto fetch = [&](int slot, int k) { p.W.load(a_row, k + a_var, sa[slot][0]); };
    auto stash = [&](int slot, int stage) {
        __bf16* a_hi = a_stage + stage * T::A_STAGE;
        __bf16* a_lo = a_hi + GA::ELEMS;
        bf16x8 hi, lo;
        s8_unpack(sa[slot][0], hi, lo);
        *reinterpret_cast<bf16x8*>(a_lo + a_lds) = lo;
        *reinterpret_cast<bf16x8*>(a_hi + a_lds) = hi;
    };

    // ---- one image: request its chunks (registers only) / copy them into the LDS image (fill_image_s8's walk, one batch) ----
    const int cpp = g.cin_p >> 3, n_chunks = p.R * p.Wp * cpp;
    uint32_t dpix_u, dcc_u, dpy_u, dpx_u;
    p.d_chunk.divmod((uint32_t)NTHR_ALL, dpix_u, dcc_u);
    p.d_Wp.divmod(dpix_u, dpy_u, dpx_u);
    const int dpix = (int)dpix_u, dcc = (int)dcc_u, dpy = (int)dpy_u, dpx = (int)dpx_u;
    float rv[S8P_BATCH][8];
    auto walk = [&](auto&& visit) {  // visit(u, on, lr, xl, cc, pix) for this thread's chunks tid_all, tid_all + 512, ...
        uint32_t pix_u, cc_u, lr_u, xl_u;
        p.d_chunk.divmod((uint32_t)tid_all, pix_u, cc_u);
        p.d_Wp.divmod(pix_u, lr_u, xl_u);
        int cc = (int)cc_u, lr = (int)lr_u, xl = (int)xl_u, pix = (int)pix_u, c0 = tid_all;
#pragma unroll
        for (int u = 0; u < S8P_BATCH; ++u) {
            visit(u, c0 < n_chunks, lr, xl, cc, pix);
            c0 += NTHR_ALL;
            cc += dcc;
            const int carry_c = cc >= cpp ? 1 : 0;
            cc -= carry_c ? cpp : 0;
            pix += dpix + carry_c;
            xl += dpx + carry_c;
            lr += dpy;
            const int carry_x = xl >= p.Wp ? 1 : 0;
            xl -= carry_x ? p.Wp : 0;
            lr += carry_x;
        }
    };
    auto request = [&](int j) {
        const float* src = p.in + (int64_t)j * g.hin * g.win * g.cin_p;
        walk([&](int u, bool on, int lr, int xl, int cc, int) {
            const int iy = row_base + lr, ix = xl - g.pad;
            const bool ok = on && iy >= 0 && iy < g.hin && ix >= 0 && ix < g.win;
            load8_aligned(ok ? src + ((iy * g.win + ix) * g.cin_p + cc * 8) : zero_chunk(), rv[u]);
        });
    };
    auto commit = [&]() {
        walk([&](int u, bool on, int, int, int cc, int pix) {
            bf16x8 hi, lo;
            s8_unpack(rv[u], hi, lo);
            if (on) {
                const int dst = pix * p.PP + cc * 8;
                *reinterpret_cast<bf16x8*>(img + p.plane_elems + dst) = lo;
                *reinterpret_cast<bf16x8*>(img + dst) = hi;
            }
        });
    };

    // per-lane patch origins of the two 16-pixel column tiles of this wave (the same for both images)
    int b_org[NT], out_pix[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        int pp = wave * 32 + nt * 16 + column_slot(lane & 15);
        const bool in_img = pp < g.npix;
        pp = in_img ? pp : g.npix - 1;
        int oy, ox;
        p.order.map(pp, oy, ox);
        out_pix[nt] = in_img ? oy * g.wout + ox : -1;
        const int ly0 = oy * g.stride - g.pad - row_base;
        const int lx0 = ox * g.stride;
        b_org[nt] = (ly0 * p.Wp + lx0) * p.PP;
    }
    struct Frags {
        bf16x8 a_hi[MTW], a_lo[MTW], b_hi[NT], b_lo[NT];
    };

    auto do_image = [&](auto it_c) {
        constexpr int it = decltype(it_c)::value;
        // Which two images: both on the XCD that PRODUCED them.  The first layer's workgroups for image i run on XCD i mod 8
        // (xcd_image_tile), this kernel's workgroup b on XCD b mod 8 (round-robin dispatch), and the next layer's workgroup for
        // image i on XCD i mod 8 again: with images 16 k + x and 16 k + 8 + x (x = b mod 8, k = b / 8) a workgroup reads rows
        // that are still in its own XCD's L2 and leaves its output where the next kernel looks for it, instead of 2 b and
        // 2 b + 1, whose producers and consumers sit on other XCDs (the fill then waits for the Infinity Cache).  Speed only.
#if defined(ISDQN_PAIR_PLAIN_ORDER)
        const bool by_xcd = false;
#else
        const bool by_xcd = (p.n_img & 15) == 0;
#endif
        const int bx = (int)blockIdx.x;
        const int j_first = by_xcd ? ((bx >> 3) << 4) + (bx & 7) : 2 * bx;
        const int j_second = by_xcd ? j_first + 8 : j_first + 1;
        const int j = it == 0 ? j_first : j_second;
        const int rot = (int)((unsigned)(j >> 3) % (unsigned)nsteps);  // conv_fwd_img_kernel: (blockIdx.x >> 3) % nsteps with blockIdx.x = j
        auto slice = [&](int s) {  // K step of this group's loop position s (s < NPOS wherever it is called)
            const int k = s * KG + kg + rot;
            return k >= nsteps ? k - nsteps : k;
        };
        auto tap_offset_of = [&](int kk) {
            int kq = kk * GEMM_BK + grp * 8;
            kq = kq < k_last ? kq : k_last;
            uint32_t tap, ci, ky, kx;
            g.d_cinp.divmod((uint32_t)kq, tap, ci);
            g.d_ksz.divmod(tap, ky, kx);
            return ((int)ky * p.Wp + (int)kx) * p.PP + (int)ci;
        };
#pragma unroll
        for (int d = 0; d < PF; ++d) fetch(d, slice(d) * GEMM_BK);
        if constexpr (it == 0) request(j);
        commit();  // (it == 1: requested under image 2w's last K steps)

        f32x4 acc[MTW][NT];
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) mfma_init(acc[mt][nt]);
        Frags fr[2];
        if constexpr (it == 0) {
            if (kg == 0 && tid < 192) s_par[tid >> 6][tid & 63] = par_v;
        }
        stash(0, 0);
        stash(1, 1);
        fetch(0, slice(PF) * GEMM_BK);
        fetch(1, slice(PF + 1) * GEMM_BK);
        __syncthreads();  // image and the first two weight slices visible
        {  // read_frags(0, slice(0), fr[0])
            const __bf16* a_hi = a_stage;
            const __bf16* a_lo = a_hi + GA::ELEMS;
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt) {
                fr[0].a_hi[mt] = read_frag<false, GA::PITCH>(a_hi, mt * 16, lane);
                fr[0].a_lo[mt] = read_frag<false, GA::PITCH>(a_lo, mt * 16, lane);
            }
            const int tap_off = tap_offset_of(slice(0));
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const __bf16* src = img + b_org[nt] + tap_off;
                fr[0].b_hi[nt] = *reinterpret_cast<const bf16x8*>(src);
                fr[0].b_lo[nt] = *reinterpret_cast<const bf16x8*>(src + p.plane_elems);
            }
        }
        __syncthreads();  // every wave has read stage 0 before step 0 overwrites it with slice 2
        int tap_next = tap_offset_of(slice(1));
#pragma unroll
        for (int s = 0; s < NPOS; ++s) {
            const Frags& fc = fr[s & 1];
            Frags& fn = fr[(s + 1) & 1];
            const __bf16* na_hi = a_stage + ((s + 1) & 1) * T::A_STAGE;
            const __bf16* na_lo = na_hi + GA::ELEMS;
            const int tap_off = tap_next;  // of slice(s + 1)
            const int slot = (s + 2) % PF;
            __bf16* st_hi = a_stage + (s & 1) * T::A_STAGE + a_lds;
            __bf16* st_lo = st_hi + GA::ELEMS;
            const bool nxt = s + 1 < NPOS, stg = s + 2 < NPOS, ftc = s + 2 + PF < NPOS;  // (constants once the loop is unrolled)
            bf16x8 c_hi, c_lo;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 24; ++q) {
                const int pass = q >> 3, nt = (q >> 2) & 1, mt = q & 3;
                mfma_acc(acc[mt][nt], pass == 1 ? fc.a_lo[mt] : fc.a_hi[mt], pass == 0 ? fc.b_lo[nt] : fc.b_hi[nt]);
                if (q < 4) { if (nxt) fn.a_hi[q] = read_frag<false, GA::PITCH>(na_hi, q * 16, lane); }
                else if (q < 8) { if (nxt) fn.a_lo[q - 4] = read_frag<false, GA::PITCH>(na_lo, (q - 4) * 16, lane); }
                else if (q < 10) { if (nxt) fn.b_hi[q - 8] = *reinterpret_cast<const bf16x8*>(img + b_org[q - 8] + tap_off); }
                else if (q < 12) { if (nxt) fn.b_lo[q - 10] = *reinterpret_cast<const bf16x8*>(img + b_org[q - 10] + tap_off + p.plane_elems); }
                else if (q == 12) { if (stg) s8_unpack(sa[slot][0], c_hi, c_lo); }
                else if (q < 16) {
                } else if (q == 16) { if (stg) *reinterpret_cast<bf16x8*>(st_lo) = c_lo; }
                else if (q == 17) { if (stg) *reinterpret_cast<bf16x8*>(st_hi) = c_hi; }
                else if (q == 18) { if (ftc) fetch(slot, slice(s + 2 + PF) * GEMM_BK); }
                else if (q == 19) { if (s + 2 < NPOS) tap_next = tap_offset_of(slice(s + 2)); }
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (it == 0) {
                if (s == NPOS - 3) request(j_second);  // behind the last weight-slice wait: two positions, the exchange and the epilogue cover the trip
            }
            __syncthreads();
        }
        // ---- the two groups hold partial sums of the same tile: group g finalizes pixel tile nt = g (conv_fwd_img_kernel) ----
        {
            float* red = reinterpret_cast<float*>(img);  // (the image is dead: the K loop ended with a barrier)
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
        // ---- epilogue: bias + LayerNorm over channels + ReLU ----
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
            if (nt != kg) continue;  // the partner group finalizes this tile
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
            if (nt != kg) continue;
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
                    if (j < p.z_img) *reinterpret_cast<float4*>(p.z + pix * g.cout_p + ch0) = zq;
                }
            }
        }
        if constexpr (it == 0) __syncthreads();  // every wave is done with the exchange area before image 2w + 1 is copied over it
    };
    do_image(std::integral_constant<int, 0>{});
    do_image(std::integral_constant<int, 1>{});
}

template <int NPOS>
static int launch_conv_fwd_s8_pair(const ConvImgParams& p, hipStream_t st) {
    using T = ConvImgTraits<4, 3, false>;
    const int lds = (2 * 2 * T::A_STAGE + T::B_PLANES * p.plane_elems) * 2;
    static LdsConfigured configured;
    if (int rc = ensure_dynamic_lds(&conv_fwd_s8_pair_kernel<NPOS>, lds, configured)) return rc;
    ISDQN_REPORT_OCCUPANCY((&conv_fwd_s8_pair_kernel<NPOS>), GEMM_THREADS * 2, lds, p.n_img / 2);
    hipLaunchKernelGGL((conv_fwd_s8_pair_kernel<NPOS>), dim3(p.n_img / 2), dim3(GEMM_THREADS * 2), lds, st, p);
    ISDQN_HIP_CHECK(hipGetLastError());
    return ISDQN_OK;
}

}  // namespace isdqn
