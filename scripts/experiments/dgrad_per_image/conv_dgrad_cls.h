// Data gradient of a STRIDED convolution + LayerNorm / ReLU backward of the layer below, ONE workgroup per image.
//
// conv_dgrad_img_kernel (conv_img.h) gives every (image, stride class) its own workgroup: for the 4 x 4 / 2 layer that is four
// workgroups per image which each stage the SAME dz image (round-4 stamps, B = 256, the kernel alone on the chip: fill 6.7 k of a
// workgroup's 22.8 k cycles, 8-step K loop 6.1 k, epilogue 6.3 k -- 96 MFMAs per wave, 7 % of the workgroup's life).  Here the eight
// waves of a workgroup stage the dz image of their image ONCE and then work as two groups of four: group g runs the stride classes
// (tiles) g, g + 2, ... one after the other, each with its own pair of weight stages; the barriers are the workgroup's, so both
// groups walk their K loops in step (every class has the same K).  The next tile's first weight slices are requested before the
// current tile's epilogue and travel under it.
//
// The arithmetic of a tile is conv_dgrad_img_kernel's, statement for statement (same K rotation per (image, tile) as the
// workgroup that used to own it, same pass order, same epilogue), and the per-tile partial sums go to the row that workgroup
// wrote, so reduce_rows sees the same rows in the same order: the results are bit-identical to the one-class kernel
// (tests/test_gpu_network.py holds the two builds to the same bits).
#pragma once
#include "conv_img.h"

namespace isdqn {

template <int MT, int PASSES>
__global__ __launch_bounds__(2 * GEMM_THREADS) void conv_dgrad_cls_kernel(const ConvDgradImgParams p) {
    ISDQN_EMPTY_KERNEL_RETURN
    constexpr int NT = 2;
    constexpr int BM = MT * 16;
    constexpr int A_PLANES = PASSES >= 2 ? 2 : 1;
    constexpr int B_PLANES = PASSES >= 3 ? 2 : 1;
    using GA = TileGeom<BM, true>;  // weights: TR image [32 k][BM ci]
    constexpr int A_STAGE = A_PLANES * GA::ELEMS;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tid_wg = threadIdx.x;
    const int half = tid_wg >> 8;             // wave group 0 / 1
    const int tid = tid_wg & (GEMM_THREADS - 1);  // thread inside the group: everything below is the one-class kernel's `tid`
    __bf16* a_stage = reinterpret_cast<__bf16*>(smem_raw) + half * 2 * A_STAGE;
    __bf16* img = reinterpret_cast<__bf16*>(smem_raw) + 2 * 2 * A_STAGE;
    __shared__ float s_part[2][4][3][64];
    __shared__ __attribute__((aligned(16))) float s_gb[2][64];
    const ConvGeom& g = p.g;
    const int lane = tid & 63, wave = tid >> 6, grp = lane >> 4;

    const int j = (int)blockIdx.x;  // image
    const int T_img = p.tiles_per_img;
    const int smask = g.stride - 1;

    // LayerNorm parameters of the layer below: one float per thread of the first 128, parked in LDS behind the fill
    float par_v = 0.f;
    {
        const int which = tid_wg >> 6, ch = tid_wg & 63;
        const float* src = which == 0 ? p.gamma : p.beta;
        const bool ok = tid_wg < 128 && ch < g.cin_p && p.gamma != nullptr;
        ISDQN_BOUNDS_CHECK(ok ? src + ch : zero_chunk(), 4, 13);
        par_v = *(const ISDQN_GLOBAL float*)(ok ? src + ch : zero_chunk());
    }

    const int nsteps = (p.Kc + GEMM_BK - 1) / GEMM_BK;
    const int k_last = p.Kc - 8;
    // ---- weight K-slice staging of this group: TR image [32 k][BM ci], chunk = 8 consecutive ci of one (tap, co) ----
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
    constexpr int PF = 4;  // weight slices in flight (register ring)
    float sa[PF][A_PER][8];
    int py = 0, px = 0;  // tap parity of the current class
    auto fetch = [&](int slot, int k0) {
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const int k = k0 + a_kk[i];
            const bool ok = (k < p.Kc) && (a_ci0[i] < g.cin_p);
            uint32_t jt, co;
            g.d_coutp.divmod(ok ? (uint32_t)k : 0u, jt, co);
            uint32_t jy_u, jx_u;
            p.d_T.divmod(jt, jy_u, jx_u);
            const int ky = py + g.stride * (int)jy_u, kx = px + g.stride * (int)jx_u;
            load8_aligned(ok ? p.W + ((int64_t)co * g.K + (ky * g.ksz + kx) * g.cin_p + a_ci0[i]) : zero_chunk(), sa[slot][i]);
        }
    };
    auto stash = [&](int slot, int stage) {
        __bf16* a_hi = a_stage + stage * A_STAGE;
        __bf16* a_lo = a_hi + GA::ELEMS;
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            if (GA::CHUNKS % GEMM_THREADS != 0 && !a_on[i]) continue;
            bf16x8 hi, lo;  // S8 mirror: staged by copy
            if constexpr (PASSES >= 2) {
                s8_unpack(sa[slot][i], hi, lo);
                *reinterpret_cast<bf16x8*>(a_lo + a_lds[i]) = lo;
            } else {
                s8_unpack_hi(sa[slot][i], hi);
            }
            *reinterpret_cast<bf16x8*>(a_hi + a_lds[i]) = hi;
        }
    };

    // class, K rotation and first weight slices of tile tl (the rotation of the workgroup that owned (j, tl) in the one-class kernel)
    const int nsteps_p = (nsteps + PF - 1) / PF * PF;
    int rot = 0, cls = 0, old_wg = 0;
    auto slice = [&](int s) {
        const int k = s + rot;
        return s < nsteps ? (k >= nsteps ? k - nsteps : k) : nsteps_p;
    };
    auto begin_tile = [&](int tl) {
        cls = 0;
        while (cls + 1 < p.n_classes && tl >= p.cls_tile_start[cls + 1]) ++cls;
        const int full = (p.n_img / 8) * 8;  // inverse of xcd_image_tile
        old_wg = (T_img > 1 && j < full) ? (j >> 3) * 8 * T_img + tl * 8 + (j & 7) : j * T_img + tl;
        rot = (int)(((unsigned)old_wg >> 3) % (unsigned)nsteps);
        const int cy = cls >> g.stride_sh, cx = cls & smask;
        py = (cy + g.pad) & smask;
        px = (cx + g.pad) & smask;
#pragma unroll
        for (int d = 0; d < PF; ++d) fetch(d, slice(d) * GEMM_BK);
    };
    begin_tile(half);

    // ---- dz image of this sample into LDS, by all eight waves ----
    fill_image_s8<2 * GEMM_THREADS, B_PLANES, 4>(img, p.dz_plane, p.dz + (int64_t)j * g.hout * g.wout * g.cout_p, g.hout, g.wout, g.cout_p,
                                                 -p.bt, -p.bt, p.Hd, p.Wd, p.PPd, tid_wg, p.d_chunk, p.d_Wd);
    if (tid_wg < 128) s_gb[tid_wg >> 6][tid_wg & 63] = par_v;

    struct Frags {
        bf16x8 a_hi[MT], a_lo[MT], b_hi[NT], b_lo[NT];
    };
    for (int tl = half; tl < T_img; tl += 2) {
        const int cy = cls >> g.stride_sh, cx = cls & smask;
        const int Ha = (g.hin - cy + g.stride - 1) >> g.stride_sh, Wb = (g.win - cx + g.stride - 1) >> g.stride_sh;
        const int n_cls_pix = Ha * Wb;
        const int q0 = (tl - p.cls_tile_start[cls]) * 128;  // first class-local pixel of this tile
        // ---- per-lane pixel of the two column tiles ----
        int b_org[NT], pix_iy[NT], pix_ix[NT];
        bool pix_ok[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            int q = q0 + wave * 32 + nt * 16 + column_slot(lane & 15);
            pix_ok[nt] = q < n_cls_pix;
            q = pix_ok[nt] ? q : n_cls_pix - 1;
            int a, b;
            p.cls_order[cls].map(q, a, b);
            const int iy = cy + g.stride * a, ix = cx + g.stride * b;
            pix_iy[nt] = iy; pix_ix[nt] = ix;
            const int oyb = (iy + g.pad - py) >> g.stride_sh, oxb = (ix + g.pad - px) >> g.stride_sh;
            b_org[nt] = ((oyb + p.bt) * p.Wd + oxb + p.bt) * p.PPd;
        }
        f32x4 acc[MT][NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) mfma_init(acc[mt][nt]);

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
            const int tap_off = -((int)jy_u * p.Wd + (int)jx_u) * p.PPd + (int)co;
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
                    if constexpr (PASSES >= 3) mfma_acc(acc[mt][nt], f.a_hi[mt], f.b_lo[nt]);
                    if constexpr (PASSES >= 2) mfma_acc(acc[mt][nt], f.a_lo[mt], f.b_hi[nt]);
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

        // ---- K loop: one barrier per step (the workgroup's: both groups walk their loops together) ----
        static_assert(PF % 2 == 0 && PF >= 4, "the fragment sets alternate with the step parity; steps 0..3 are pre-fetched");
        Frags fr[2];
        stash(0, 0);
        stash(1, 1);
        fetch(0, slice(PF) * GEMM_BK);
        fetch(1, slice(PF + 1) * GEMM_BK);
        __syncthreads();  // (first tile: the dz image and the LayerNorm parameters too)
        read_frags(0, slice(0), fr[0]);
        __syncthreads();  // every wave has read stage 0 before step 0 overwrites it with slice 2
        for (int s0 = 0; s0 < nsteps_p; s0 += PF) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int s = s0 + u;
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

        // the next tile of this group: class constants, and its first weight slices travel under this tile's epilogue
        const int tl_here = tl, cls_here = cls, old_wg_here = old_wg;
        (void)cls_here;
        if (tl + 2 < T_img) begin_tile(tl + 2);

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
        // ---- partial sums of this tile: over the 16 pixel lanes of a group, then over the 4 waves; one row per (image, tile) ----
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                dg[mt][r] = row16_sum(dg[mt][r]);
                db[mt][r] = row16_sum(db[mt][r]);
                dbias[mt][r] = row16_sum(dbias[mt][r]);
                if ((lane & 15) == 0) {
                    const int ch = mt * 16 + grp * 4 + r;
                    s_part[half][wave][0][ch] = dg[mt][r];
                    s_part[half][wave][1][ch] = db[mt][r];
                    s_part[half][wave][2][ch] = dbias[mt][r];
                }
            }
        __syncthreads();
        for (int i = tid; i < 3 * g.cin_p; i += GEMM_THREADS) {
            const int which = (i >= g.cin_p) + (i >= 2 * g.cin_p), c = i - which * g.cin_p;
            p.part[((int64_t)old_wg_here * 3 + which) * g.cin_p + c] =
                s_part[half][0][which][c] + s_part[half][1][which][c] + s_part[half][2][which][c] + s_part[half][3][which][c];
        }
        (void)tl_here;
        // (the next tile's stash / K-loop barriers order its s_part writes behind these reads: at least nsteps_p + 2 barriers)
    }
}

template <int MT, int PASSES>
static int launch_conv_dgrad_cls(const ConvDgradImgParams& p, hipStream_t st) {
    constexpr int A_PLANES = PASSES >= 2 ? 2 : 1;
    constexpr int B_PLANES = PASSES >= 3 ? 2 : 1;
    using GA = TileGeom<MT * 16, true>;
    const int lds = (2 * 2 * A_PLANES * GA::ELEMS + B_PLANES * p.dz_plane) * 2;
    static LdsConfigured configured;
    if (int rc = ensure_dynamic_lds(&conv_dgrad_cls_kernel<MT, PASSES>, lds, configured)) return rc;
    ISDQN_REPORT_OCCUPANCY((&conv_dgrad_cls_kernel<MT, PASSES>), 2 * GEMM_THREADS, lds, p.n_img);
    hipLaunchKernelGGL((conv_dgrad_cls_kernel<MT, PASSES>), dim3(p.n_img), dim3(2 * GEMM_THREADS), lds, st, p);
    ISDQN_HIP_CHECK(hipGetLastError());
    return ISDQN_OK;
}
static inline int conv_dgrad_cls_lds_bytes(int mt, int passes, int dz_plane) {
    return (2 * 2 * (passes >= 2 ? 2 : 1) * 32 * (mt * 16 + 16) + (passes >= 3 ? 2 : 1) * dz_plane) * 2;
}

}  // namespace isdqn
