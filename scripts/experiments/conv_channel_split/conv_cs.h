// Image-resident convolution, CHANNEL-SPLIT waves that run free (round 4).
//
// conv_fwd_img_kernel (conv_img.h) splits a workgroup's tile by PIXELS: every wave needs every weight, so the weight K-slices
// are staged through LDS and the four (or eight) waves meet at a barrier in every K step -- with one or two workgroups per CU
// nothing covers a wave that waits there (round-4 stamps of the same structure in the data gradient: 1 080 cycles per K step for
// 384 cycles of MFMA).  Here a wave owns ONE 16-channel tile and ALL pixel tiles of its image:
//   * its weight fragment of a K step is 16 rows x 32 k of the S8 mirror = one 32-byte group per lane, loaded straight into the
//     MFMA A registers (hi = first 16 bytes, lo = the rest) through a ring of register sets -- no weight stage in LDS, nobody else
//     needs these bytes, so no barrier in the K loop at all: after the image fill the waves run free;
//   * the pixel fragments come from the LDS image as before (per-lane pixel address + tap offset): 8 tiles x (hi, lo) per step.
// Per K step and wave: one 32-byte global load per lane, 16 ds_read_b128, 24 MFMAs (the pixel-split wave: 12 + 8 reads, 2 LDS
// writes, a barrier).  LDS reads per workgroup-step: 64 KB (48 KB before), still under the matrix pipe's 384 cycles at 256 B/clk.
// The LayerNorm statistics of a pixel now span the four waves: partial sums (16 channels each) meet in LDS once, in the epilogue.
//
// KG = 2 (layers whose image leaves room for one workgroup per CU: the 21x21x32 input of the 4x4/2 layer): eight waves, the two
// groups of four take alternate K steps -- still without a barrier in the loop -- swap half of their pixel tiles' partial sums
// through the dead image and finalize four tiles each.
#pragma once
#include "conv_img.h"

namespace isdqn {

template <int PASSES, int KG>
__global__ __launch_bounds__(GEMM_THREADS * KG, 2) void conv_fwd_cs_kernel(const ConvImgParams p) {
    ISDQN_EMPTY_KERNEL_RETURN
    constexpr int NTC = 8;                 // pixel tiles of 16: one image of up to 128 output pixels per workgroup
    constexpr int NTF = NTC / KG;          // tiles a K group finalizes
    constexpr int B_PLANES = PASSES >= 3 ? 2 : 1;
    constexpr int NTHR_ALL = GEMM_THREADS * KG;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    __bf16* img = reinterpret_cast<__bf16*>(smem_raw);
    __shared__ __attribute__((aligned(16))) float s_par[3][64];
    __shared__ __attribute__((aligned(16))) float s_stat[KG][4][NTF * 16][2];  // [group][wave][pixel of the group's tiles][s1, s2]
    const ConvGeom& g = p.g;
    const int tid_all = threadIdx.x, kg = KG > 1 ? tid_all / GEMM_THREADS : 0;
    const int tid = tid_all - kg * GEMM_THREADS, lane = tid & 63, wave = tid >> 6, grp = lane >> 4;
    const int j = (int)blockIdx.x;  // image
#if defined(ISDQN_DEV)
#define ISDQN_STAMP(i)                                                                               \
    if (p.stamps != nullptr && threadIdx.x == 0) {                                                  \
        p.stamps[(int64_t)blockIdx.x * 8 + (i)] = (long long)__builtin_amdgcn_s_memtime();           \
        if ((i) == 0) p.stamps[(int64_t)blockIdx.x * 8 + 7] = (long long)__builtin_amdgcn_s_memrealtime(); \
    }
#else
#define ISDQN_STAMP(i)
#endif
    ISDQN_STAMP(0);

    // epilogue parameters: one float per thread requested now, parked in LDS behind the fill
    float par_v = 0.f;
    {
        const int which = tid >> 6, ch = tid & 63;
        const float* src = which == 0 ? p.bias : which == 1 ? p.gamma : p.beta;
        const bool ok = kg == 0 && tid < 192 && ch < g.cout_p && src != nullptr;
        ISDQN_BOUNDS_CHECK(ok ? src + ch : zero_chunk(), 4, 13);
        par_v = *(const ISDQN_GLOBAL float*)(ok ? src + ch : zero_chunk());
    }

    const int nsteps = (g.K + GEMM_BK - 1) / GEMM_BK;
    const int k_last = g.K - 8;
    // ---- this lane's weight row and the ring of fragments in flight ----
    const int co = wave * 16 + (lane & 15);
    constexpr int PFW = 4;
    float wa[PFW][8];
    const int nsteps_p = ((nsteps + KG - 1) / KG + PFW - 1) / PFW * PFW;  // loop positions of one K group
    const int rot = (int)((blockIdx.x >> 3) % (unsigned)nsteps);          // workgroups of one XCD start at different slices
    auto slice = [&](int s) {  // K step at this group's loop position s; positions past the last step read zeros
        const int gs = s * KG + kg;
        const int k = gs + rot;
        return gs < nsteps ? (k >= nsteps ? k - nsteps : k) : nsteps_p * KG;
    };
    auto fetch = [&](int slot, int kk) { p.W.load(co, kk * GEMM_BK + grp * 8, wa[slot]); };
#pragma unroll
    for (int d = 0; d < PFW; ++d) fetch(d, slice(d));  // travel under the image fill

    // ---- the whole input image into LDS (zero border included) ----
    constexpr int FILL_BATCH = 12 / KG;
    fill_image_s8<NTHR_ALL, B_PLANES, FILL_BATCH>(img, p.plane_elems, p.in + (int64_t)j * g.hin * g.win * g.cin_p, g.hin, g.win, g.cin_p,
                                                  -g.pad, -g.pad, p.R, p.Wp, p.PP, tid_all, p.d_chunk, p.d_Wp);
    if (kg == 0 && tid < 192) s_par[tid >> 6][tid & 63] = par_v;
    ISDQN_STAMP(1);  // fill loads consumed, LDS image written (this wave)

    // ---- per-lane patch origin of every pixel tile ----
    int b_org[NTC], out_pix[NTC];
#pragma unroll
    for (int nt = 0; nt < NTC; ++nt) {
        int pp = nt * 16 + column_slot(lane & 15);
        const bool in_img = pp < g.npix;
        pp = in_img ? pp : g.npix - 1;  // lanes past the image compute a duplicate pixel; never stored
        int oy, ox;
        p.order.map(pp, oy, ox);
        out_pix[nt] = in_img ? oy * g.wout + ox : -1;
        b_org[nt] = (oy * g.stride * p.Wp + ox * g.stride) * p.PP;  // (local row 0 is input row -pad)
    }
    auto tap_offset_of = [&](int kk) {  // image offset of this lane's 8-channel chunk of K step kk
        int kq = kk * GEMM_BK + grp * 8;
        kq = kq < k_last ? kq : k_last;
        uint32_t tap, ci, ky, kx;
        g.d_cinp.divmod((uint32_t)kq, tap, ci);
        g.d_ksz.divmod(tap, ky, kx);
        return ((int)ky * p.Wp + (int)kx) * p.PP + (int)ci;
    };

    f32x4 acc[NTC];
#pragma unroll
    for (int nt = 0; nt < NTC; ++nt) mfma_init(acc[nt]);

    __syncthreads();  // the image (and the epilogue parameters) are in LDS: from here to the end of the K loop no wave waits for another
    ISDQN_STAMP(2);

    // Pixel fragments of one K step: 8 tiles x (hi, lo).  Two sets alternate: the 16 reads of step s + 1 are issued between the MFMAs
    // of step s (one read behind each of the first 16), so a step's fragments have had a whole step to land when its MFMAs start.
    struct BFrags {
        bf16x8 hi[NTC], lo[NTC];
    };
    auto read_b = [&](int tap_off, BFrags& f) {
#pragma unroll
        for (int nt = 0; nt < NTC; ++nt) {
            const __bf16* src = img + b_org[nt] + tap_off;
            f.hi[nt] = *reinterpret_cast<const bf16x8*>(src);
            if constexpr (PASSES >= 3) f.lo[nt] = *reinterpret_cast<const bf16x8*>(src + p.plane_elems);
        }
    };
    static_assert(PFW % 2 == 0, "the fragment sets alternate with the step parity");
    BFrags fb[2];
    read_b(tap_offset_of(slice(0)), fb[0]);
    for (int s0 = 0; s0 < nsteps_p; s0 += PFW) {
#pragma unroll
        for (int u = 0; u < PFW; ++u) {
            const int s = s0 + u;
            bf16x8 a_hi, a_lo;
            if constexpr (PASSES >= 2) s8_unpack(wa[u], a_hi, a_lo);
            else s8_unpack_hi(wa[u], a_hi);
            const BFrags& fc = fb[u & 1];
            read_b(tap_offset_of(slice(s + 1)), fb[(u + 1) & 1]);
#pragma unroll
            for (int nt = 0; nt < NTC; ++nt) {
                if constexpr (PASSES >= 3) mfma_acc(acc[nt], a_hi, fc.lo[nt]);
                if constexpr (PASSES >= 2) mfma_acc(acc[nt], a_lo, fc.hi[nt]);
                mfma_acc(acc[nt], a_hi, fc.hi[nt]);
            }
            fetch(u, slice(s + PFW));  // refill the slot this step consumed
            constexpr int N_MFMA = NTC * (PASSES >= 3 ? 3 : PASSES);
            constexpr int N_DS = NTC * B_PLANES;
#pragma unroll
            for (int i = 0; i < N_MFMA; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
                if (i < N_DS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // 1 DS read
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);  // 2 VALU (tap offsets, addresses)
            }
        }
    }

    ISDQN_STAMP(3);  // K loop done (this wave)
    if constexpr (KG > 1) {
        // The two groups hold partial sums of all eight tiles: group g keeps tiles [4g, 4g + 4) and hands the other four to its
        // partner through the image area (dead once every wave is past its last fragment read).
        __syncthreads();
        float* red = reinterpret_cast<float*>(img);  // [group][wave][tile][r][lane]
#pragma unroll
        for (int t = 0; t < NTF; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                red[(((kg * 4 + wave) * NTF + t) * 4 + r) * 64 + lane] = kg == 0 ? acc[NTF + t][r] : acc[t][r];  // (static indices: a runtime-indexed register array lives in scratch)
        __syncthreads();
#pragma unroll
        for (int t = 0; t < NTF; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r)
            {
                const float got = red[((((1 - kg) * 4 + wave) * NTF + t) * 4 + r) * 64 + lane];
                if (kg == 0) acc[t][r] += got;
                else acc[NTF + t][r] += got;
            }
    }

    // ---- epilogue: bias + LayerNorm over the 64 channels of a pixel (four waves x 16) + ReLU ----
    const int ch0 = wave * 16 + grp * 4;  // this lane's four channels of every pixel
    float bi[4], ga[4], be[4];
    {
        const float4 b4 = *reinterpret_cast<const float4*>(&s_par[0][ch0]);
        const float4 g4 = *reinterpret_cast<const float4*>(&s_par[1][ch0]);
        const float4 e4 = *reinterpret_cast<const float4*>(&s_par[2][ch0]);
        bi[0] = b4.x; bi[1] = b4.y; bi[2] = b4.z; bi[3] = b4.w;
        ga[0] = g4.x; ga[1] = g4.y; ga[2] = g4.z; ga[3] = g4.w;
        be[0] = e4.x; be[1] = e4.y; be[2] = e4.z; be[3] = e4.w;
    }
    const float inv_c = 1.0f / (float)g.cout;
    float zv[NTF][4];
#pragma unroll
    for (int t = 0; t < NTF; ++t) {
        const int nt = kg * NTF + t;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float av = KG == 1 ? acc[t][r] : (kg == 0 ? acc[t][r] : acc[(KG - 1) * NTF + t][r]);
            const float zz = ch0 + r < g.cout ? av * p.scale + bi[r] : 0.f;
            zv[t][r] = zz;
            s1 += zz;
            s2 += zz * zz;
        }
        if (p.gamma != nullptr) {
            s1 += __shfl_xor(s1, 16); s1 += __shfl_xor(s1, 32);
            s2 += __shfl_xor(s2, 16); s2 += __shfl_xor(s2, 32);
            if (grp == 0) *reinterpret_cast<float2*>(&s_stat[kg][wave][t * 16 + (lane & 15)][0]) = make_float2(s1, s2);
        }
    }
    if (p.gamma != nullptr) __syncthreads();  // (uniform: every thread of the workgroup takes the same branch)
#pragma unroll
    for (int t = 0; t < NTF; ++t) {
        const int nt = kg * NTF + t;
        float mean = 0.f, rstd = 1.f;
        if (p.gamma != nullptr) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {  // fixed order: the same bits on every wave and every run
                const float2 q = *reinterpret_cast<const float2*>(&s_stat[kg][w][t * 16 + (lane & 15)][0]);
                s1 += q.x;
                s2 += q.y;
            }
            mean = s1 * inv_c;
            rstd = rsqrtf(fmaxf(s2 * inv_c - mean * mean, 0.f) + 1e-6f);
        }
        const int opix = KG == 1 ? out_pix[t] : (kg == 0 ? out_pix[t] : out_pix[(KG - 1) * NTF + t]);
        if (opix >= 0 && ch0 < g.cout_p) {
            const int64_t pix = (int64_t)j * g.npix + opix;
            float a[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float y = p.gamma != nullptr ? (zv[t][r] - mean) * (rstd * ga[r]) + be[r] : zv[t][r];
                a[r] = (ch0 + r < g.cout) ? fmaxf(y, 0.f) : 0.f;
            }
            s8_store_quad_paired(p.act + pix * g.cout_p, ch0, a[0], a[1], a[2], a[3]);  // activations: S8
            if (j < p.z_img) *reinterpret_cast<float4*>(p.z + pix * g.cout_p + ch0) = float4{zv[t][0], zv[t][1], zv[t][2], zv[t][3]};
        }
    }
    ISDQN_STAMP(4);  // epilogue stores issued
#if defined(ISDQN_DEV)
    if (p.stamps != nullptr) {
        __builtin_amdgcn_s_waitcnt(0);
        ISDQN_STAMP(5);
    }
#endif
#undef ISDQN_STAMP
}

template <int PASSES, int KG>
static int launch_conv_fwd_cs(const ConvImgParams& p, int lds, hipStream_t st) {
    static LdsConfigured configured;
    if (int rc = ensure_dynamic_lds(&conv_fwd_cs_kernel<PASSES, KG>, lds, configured)) return rc;
    ISDQN_REPORT_OCCUPANCY((&conv_fwd_cs_kernel<PASSES, KG>), GEMM_THREADS * KG, lds, p.n_img);
    hipLaunchKernelGGL((conv_fwd_cs_kernel<PASSES, KG>), dim3(p.n_img), dim3(GEMM_THREADS * KG), lds, st, p);
    ISDQN_HIP_CHECK(hipGetLastError());
    return ISDQN_OK;
}

}  // namespace isdqn
