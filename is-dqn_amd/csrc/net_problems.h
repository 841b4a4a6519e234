// Problem definitions (operand loaders + epilogues) that instantiate the MFMA tile engine for
// every contraction of the iS-DQN gradient step.  See gemm_core.h for the engine contract.
//
// Data layouts in HBM (all fp32 unless noted):
//   frames      uint8 [slot][H*W]                     single frames of the device replay
//   activations [image][y][x][c padded to 8]          channel-last, like the reference (NHWC)
//   conv kernel [out][ky][kx][in padded to 8]         (first conv: [out][plane][ky][kx])
//   dense kernel[out][in]
// Forward GEMMs put the OUTPUT CHANNELS on the MFMA rows (M) and pixels on the columns (N):
// a pixel's channels then sit in 4 lanes x (MT*4) registers, so LayerNorm over channels
// (dqn.py:56-57: nn.LayerNorm() normalises the last axis only) is two wave shuffles.
#pragma once
#include "gemm_core.h"

namespace isdqn {

// ---------------------------------------------------------------------------------------------
// uint8 frame access: image j, plane c -> frame slot through the id table
// ---------------------------------------------------------------------------------------------
struct FrameSrc {
    const uint8_t* frames;
    int64_t stride;
    const int* ids;
    int stack;
    int paired_B;  // > 0: learn layout ids[b][2*stack] and images [0,B) = states, [B,2B) = next states
    int H, W;
    int id_pitch = 0, id_off = 0;  // unpaired: image j reads ids[j * id_pitch + id_off + c] (0: rows of `stack` ids) -- one half of a learn-layout table
    __device__ __forceinline__ int frame_id(int j, int c) const {
#if defined(ISDQN_DEV)
        if (ids == nullptr) return j * stack + c;  // (development experiment, conv_img.h)
#endif
        if (paired_B > 0) {
            const int* q = j < paired_B ? ids + (int64_t)j * 2 * stack + c : ids + (int64_t)(j - paired_B) * 2 * stack + stack + c;
            ISDQN_BOUNDS_CHECK(q, 4, 4);
            return *q;
        }
        ISDQN_BOUNDS_CHECK(ids + (int64_t)j * (id_pitch ? id_pitch : stack) + id_off + c, 4, 4);
        return ids[(int64_t)j * (id_pitch ? id_pitch : stack) + id_off + c];
    }
    // 8 horizontally adjacent pixels (ix0 .. ix0+7) of row iy of frame `id`, zero outside the frame; exact in
    // bf16.  Branch-free: one unaligned 8-byte load from a clamped position, then a 64-bit shift moves the
    // bytes into place and shifts zeros in for the columns that fall outside the row (needs W >= 8).
    // Two-phase form: `patch8_raw` only issues the (unaligned) 8-byte load -- out-of-image patches read a block of
    // zeros, so nothing touches the loaded registers -- and `patch8_cvt` shifts the border in and converts.  Callers
    // that prefetch keep `raw` and `d` and convert when they stage to LDS.
    __device__ __forceinline__ void patch8_raw(int id, int iy, int ix0, unsigned long long& raw, int& d) const {
        const bool ok = (id >= 0) && (iy >= 0) && (iy < H) && (ix0 > -8) && (ix0 < W);
        const int ixc = min(max(ix0, 0), W - 8);
        const uint8_t* row = ok ? frames + ((int64_t)id * stride + (int64_t)iy * W + ixc)
                                : reinterpret_cast<const uint8_t*>(zero_chunk());
        raw = load_u64_unaligned(row);   // one (unaligned) global_load_dwordx2
        d = ix0 - ixc;                   // < 0: left border, > 0: right border
    }
    static __device__ __forceinline__ void patch8_cvt(unsigned long long u, int d, float (&v)[8]) {
        u = (u >> (8 * max(d, 0))) << (8 * max(-d, 0));  // one of the two shifts is by zero (no branch)
        const unsigned lo = (unsigned)u, hi = (unsigned)(u >> 32);
        v[0] = (float)(lo & 0xff); v[1] = (float)((lo >> 8) & 0xff);
        v[2] = (float)((lo >> 16) & 0xff); v[3] = (float)(lo >> 24);
        v[4] = (float)(hi & 0xff); v[5] = (float)((hi >> 8) & 0xff);
        v[6] = (float)((hi >> 16) & 0xff); v[7] = (float)(hi >> 24);
    }
    __device__ __forceinline__ void patch8_id(int id, int iy, int ix0, float (&v)[8]) const {
        unsigned long long u;
        int d;
        patch8_raw(id, iy, ix0, u, d);
        patch8_cvt(u, d, v);
    }
    __device__ __forceinline__ void patch8(int j, int c, int iy, int ix0, float (&v)[8]) const {
        patch8_id(frame_id(j, c), iy, ix0, v);
    }
};

// ---------------------------------------------------------------------------------------------
// Plain GEMM over row-major matrices (dense forward / data-grad / weight-grad, head, self-test)
//   C[split][m][n] = sum_{k in split} A(m,k) * B(n,k)
// ---------------------------------------------------------------------------------------------
// AL: both sources promise 16-B aligned, 8-element-granular rows (everything except caller-provided fc
// observations and the self-test); A2: operand A comes in two row blocks (fc: state / next_state).
// Adam applied in the epilogue (optax.adam, isdqn.py:46, 85-86): used by weight-gradient GEMMs whose tile holds
// the complete gradient (no split-K), so the gradient of the big dense kernel is never written to HBM.
struct AdamFuse {
    float *p, *m, *v;        // parameter tensor and its moments, same [M][ldc] layout as C
    const float* consts;     // {1 - b1^t, 1 - b2^t}
    float lr, b1, b2, eps;
    float* grad_out;         // optional: also store the raw gradient (tests)
    float* mirror;           // S8 mirror of p (same layout): the updated parameters are written in both forms
    int update;              // 0: gradient only (grad_out), p / m / v / mirror untouched
};

// S8M: bit 0 / bit 1 = operand A / B is stored S8 (gemm_core.h): staged by copy; such an operand is always chunk-aligned.
template <int BM_, int BN_, int WM_, int WN_, bool ATR, bool BTR, int PASSES_, bool AL = true, bool A2PART = false,
          bool ADAM = false, int S8M = 0>
struct PlainGemm {
    static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_, PASSES = PASSES_;
    static constexpr bool A_TR = ATR, B_TR = BTR;
    static constexpr bool A_S8 = (S8M & 1) != 0, B_S8 = (S8M & 2) != 0;
    // the fused-Adam GEMM lives off four workgroups per CU: only one of its two TR images gets the wider conflict-free pitch
    static constexpr int TR_PAD_B = ADAM ? 8 : 16;
    MatSrc A, B;       // ROW: outer = rows, inner = K ; TR: outer = K, inner = rows
    const float* A2;   // optional second part of A (ROW only): rows >= a_split come from A2
    int a_split;
    float* C;
    int ldc, M, N, K;
    int tiles_m, tiles_n, splits, steps_per_split;
    int64_t slab_stride;
    AdamFuse adam;
    static constexpr int EPI_LDS_BYTES = ADAM ? BM_ * (BN_ + 4) * 4 : 0;
    struct Tile { int m0, n0, k0, k1, split; };
    struct ACtx { int fixed; };
    struct BCtx { int fixed; };
    __device__ __forceinline__ bool tile(int bid, Tile& t) const {
        bid = xcd_remap(bid, tiles_m * tiles_n * splits);  // m-tiles of one (n-tile, split) share an XCD
        int tm = bid % tiles_m, rest = bid / tiles_m;
        int tn = rest % tiles_n;
        t.split = rest / tiles_n;
#if defined(ISDQN_ADAM_ROWMAJOR)
        if constexpr (ADAM) {  // experiment: n-tiles fastest, so that the workgroups of an XCD stream adjacent pieces of the same p/m/v rows
            tn = bid % tiles_n;
            tm = bid / tiles_n;
        }
#endif
        if (t.split >= splits) return false;
        t.m0 = tm * BM;
        t.n0 = tn * BN;
        t.k0 = t.split * steps_per_split * GEMM_BK;
        t.k1 = min(K, t.k0 + steps_per_split * GEMM_BK);
        return true;
    }
    __device__ __forceinline__ ACtx a_ctx(const Tile&, int fixed) const { return ACtx{fixed}; }
    __device__ __forceinline__ BCtx b_ctx(const Tile&, int fixed) const { return BCtx{fixed}; }
    __device__ __forceinline__ void load_a(const Tile&, const ACtx& c, int var, float (&v)[8]) const {
        if constexpr (!ATR) {
            if constexpr (A2PART) {
                // A.outer counts the rows of BOTH parts.  Each part bounds its own rows: tile rows past the last real
                // row (a 128-row tile over 2B = 64 rows) would otherwise pass `row - a_split < A.outer` and read up to
                // a_split rows beyond the end of the caller's second matrix -- results never stored, but a page fault
                // when that matrix ends a mapped segment (round 2's intermittent abort in the fc learn step).
                const bool second = c.fixed >= a_split;
                MatSrc s = A;
                s.base = second ? A2 : A.base;
                s.outer = second ? A.outer - a_split : min(A.outer, a_split);
                s.load<AL || A_S8>(second ? c.fixed - a_split : c.fixed, var, v);
            } else {
                A.load<AL || A_S8>(c.fixed, var, v);
            }
        } else {
            A.load<AL || A_S8>(var, c.fixed, v);
        }
    }
    __device__ __forceinline__ void load_b(const Tile&, const BCtx& c, int var, float (&v)[8]) const {
        if constexpr (!BTR) B.load<AL || B_S8>(c.fixed, var, v);
        else B.load<AL || B_S8>(var, c.fixed, v);
    }
    template <int MT, int NT>
    __device__ __forceinline__ void epilogue(const Tile& t, f32x4 (&acc)[MT][NT], int m_wave, int n_wave, int lane, char* smem) const {
        if constexpr (ADAM) {
            // The MFMA accumulator layout gives each lane 4 rows x 1 column: straight to memory that is 64-byte
            // pieces of p/m/v rows.  Stage the gradient tile in LDS and let every thread own float4s of a row,
            // so parameters and moments move as full coalesced rows (16 B per lane, 512 B per row of the tile).
            constexpr int LD = BN + 4;
            float* tg = reinterpret_cast<float*>(smem);
            const int lm = m_wave - t.m0, ln = n_wave - t.n0;
            // The parameters and moments of this thread's float4s do not depend on the gradient: ALL of them are requested
            // first (12 loads of 16 B in flight per thread instead of 3, one round trip instead of four -- the kernel
            // lives off streaming p / m / v), and arrive while the gradient tile goes through LDS.
            constexpr int N_IT = BM * (BN / 4) / GEMM_THREADS;
            static_assert(BM * (BN / 4) % GEMM_THREADS == 0, "whole float4 rounds");
            const int tid = threadIdx.x;
            f32x4 pm[N_IT], pv[N_IT], pp[N_IT];
            int64_t off[N_IT];
            bool on[N_IT];
#pragma unroll
            for (int it = 0; it < N_IT; ++it) {
                const int i = tid + it * GEMM_THREADS;
                const int rl = i / (BN / 4), c4 = (i % (BN / 4)) * 4;
                const int row = t.m0 + rl, col = t.n0 + c4;
                on[it] = row < M && col < N;  // N and ldc are multiples of 4: a float4 never straddles the edge
                off[it] = on[it] ? (int64_t)row * ldc + col : 0;
                ISDQN_BOUNDS_CHECK(adam.m + off[it], 16, 10);
                ISDQN_BOUNDS_CHECK(adam.v + off[it], 16, 10);
                ISDQN_BOUNDS_CHECK(adam.p + off[it], 16, 10);
#if !defined(ISDQN_NO_STREAMING)
                // moments and master weights are read once and written once per step: streamed past the caches (nt), so that the
                // 96 MB they move do not evict what the data-gradient chain beside this kernel re-reads
                pm[it] = __builtin_nontemporal_load((const ISDQN_GLOBAL f32x4*)(adam.m + off[it]));
                pv[it] = __builtin_nontemporal_load((const ISDQN_GLOBAL f32x4*)(adam.v + off[it]));
                pp[it] = __builtin_nontemporal_load((const ISDQN_GLOBAL f32x4*)(adam.p + off[it]));
#else
                pm[it] = *(const ISDQN_GLOBAL f32x4*)(adam.m + off[it]);
                pv[it] = *(const ISDQN_GLOBAL f32x4*)(adam.v + off[it]);
                pp[it] = *(const ISDQN_GLOBAL f32x4*)(adam.p + off[it]);
#endif
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        tg[(lm + mt * 16 + (lane >> 4) * 4 + r) * LD + ln + nt * 16 + (lane & 15)] = acc[mt][nt][r];
            __syncthreads();
            const float inv_c1 = 1.f / adam.consts[0], inv_c2 = 1.f / adam.consts[1];
#pragma unroll
            for (int it = 0; it < N_IT; ++it) {
                const int i = tid + it * GEMM_THREADS;
                const int rl = i / (BN / 4), c4 = (i % (BN / 4)) * 4;
                const float4 g = *reinterpret_cast<const float4*>(tg + rl * LD + c4);
                const float* gp = &g.x;
                f32x4 nm, nv, np;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float mm = pm[it][r], vv = pv[it][r];
                    np[r] = adam_element(mm, vv, pp[it][r], gp[r], adam.b1, adam.b2, adam.lr, adam.eps, inv_c1, inv_c2);
                    nm[r] = mm;
                    nv[r] = vv;
                }
                if (on[it] && adam.grad_out) *reinterpret_cast<float4*>(adam.grad_out + off[it]) = g;
                if (on[it] && adam.update) {
#if !defined(ISDQN_NO_STREAMING)
                    __builtin_nontemporal_store(nm, (ISDQN_GLOBAL f32x4*)(adam.m + off[it]));
                    __builtin_nontemporal_store(nv, (ISDQN_GLOBAL f32x4*)(adam.v + off[it]));
                    __builtin_nontemporal_store(np, (ISDQN_GLOBAL f32x4*)(adam.p + off[it]));
#else
                    *reinterpret_cast<f32x4*>(adam.m + off[it]) = nm;
                    *reinterpret_cast<f32x4*>(adam.v + off[it]) = nv;
                    *reinterpret_cast<f32x4*>(adam.p + off[it]) = np;
#endif
                    s8_store_quad(adam.mirror, (int)off[it], np[0], np[1], np[2], np[3]);  // (tensor sizes < 2^31)
                }
            }
            return;
        }
        float* c = C + (int64_t)t.split * slab_stride;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                int col = n_wave + nt * 16 + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int row = m_wave + mt * 16 + (lane >> 4) * 4 + r;
                    if (row < M && col < N) c[(int64_t)row * ldc + col] = acc[mt][nt][r];
                }
            }
    }
};

// ---------------------------------------------------------------------------------------------
// Data gradient of the first dense layer fused with the LayerNorm + ReLU backward of the conv layer below.
//   da[b][pix*64 + c] = sum_o dz[b][o] * W[o][pix*64 + c]          (A = dz ROW, B = W TR)
// A 128 x 64 tile is 128 samples x the 64 channels of ONE pixel, and with 4 waves stacked along M every wave holds
// complete channel rows: the LayerNorm statistics of a row are 4 in-lane values + a 16-lane shuffle.  The epilogue
// reads z of the conv output, writes dz of the conv layer directly (da never goes to HBM) and emits the
// workgroup's partial sums of (dgamma, dbeta, dbias).
// ---------------------------------------------------------------------------------------------
// BM_: 128 or 64 rows per workgroup (64 doubles the workgroup count: the 3136-column problem has only 49 column tiles)
template <int PASSES_, int BM_ = 128, int KG_ = 1>
struct DenseDgradLN {
    static constexpr int BM = BM_, BN = 64, WM = 4, WN = 1, PASSES = PASSES_;
    static constexpr int KG = KG_;  // K groups (gemm_core.h): the epilogue then sees one row tile per wave, 8 waves
    static constexpr bool A_TR = false, B_TR = true;
    static constexpr bool A_S8 = true;  // dz of the dense layer: S8
    static constexpr bool B_S8 = true;  // the weights come from the S8 mirror
    static constexpr int TR_PITCH = 68;  // floats per staged row: 64 channels + 4 (bank spread of the transposed reads)
    static constexpr int SP_BYTES = 4 * KG_ * 3 * 64 * 4;                  // per-wave partial sums
    static constexpr int EPI_LDS_BYTES = SP_BYTES + 4 * KG_ * 16 * TR_PITCH * 4;  // + one 16-row x 64-channel tile per wave
    MatSrc A, B;
    const float* z;            // [M][ldc] pre-LayerNorm conv output (flat [b][pix][64])
    const float *gamma, *beta; // nullptr: ReLU only
    float* dz_out;             // [M][ldc]
    float* part;               // [n_workgroups][3][64]
    int ldc, M, N, K, c_in;
    int tiles_m, tiles_n;
    struct Tile { int m0, n0, k0, k1, wg; };
    struct ACtx { int fixed; };
    struct BCtx { int fixed; };
    __device__ __forceinline__ bool tile(int bid, Tile& t) const {
        t.wg = bid;
        bid = xcd_remap(bid, tiles_m * tiles_n);
        t.m0 = (bid % tiles_m) * BM;
        t.n0 = (bid / tiles_m) * BN;
        t.k0 = 0; t.k1 = K;
        return bid < tiles_m * tiles_n;
    }
    __device__ __forceinline__ ACtx a_ctx(const Tile&, int fixed) const { return ACtx{fixed}; }
    __device__ __forceinline__ BCtx b_ctx(const Tile&, int fixed) const { return BCtx{fixed}; }
    __device__ __forceinline__ void load_a(const Tile&, const ACtx& c, int var, float (&v)[8]) const { A.load(c.fixed, var, v); }
    __device__ __forceinline__ void load_b(const Tile&, const BCtx& c, int var, float (&v)[8]) const { B.load(var, c.fixed, v); }
    template <int MT, int NT>
    __device__ __forceinline__ void epilogue(const Tile& t, f32x4 (&acc)[MT][NT], int m_wave, int n_wave, int lane, char* smem) const {
        static_assert(NT == 4, "one wave must hold all 64 channels of a row");
        const int grp = lane >> 4, li = lane & 15, wave = (m_wave - t.m0) / (MT * 16);
        float ga[NT], be[NT], dg[NT], db[NT], dbias[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int ch = nt * 16 + li;
            const bool ok = ch < c_in;
            ga[nt] = (ok && gamma) ? gamma[ch] : 1.f;
            be[nt] = (ok && gamma) ? beta[ch] : 0.f;
            dg[nt] = db[nt] = dbias[nt] = 0.f;
        }
        const float inv_c = 1.f / (float)c_in;
        // dz of the conv layer is stored S8 (groups of 8 channels of one row), but a lane owns channels li, 16 + li, ...:
        // each wave transposes its rows through its own 16-row LDS tile and stores whole 32-byte groups
        float* tr = reinterpret_cast<float*>(smem + SP_BYTES) + wave * 16 * TR_PITCH;
        // all pre-activations of this lane first (MT*4*NT independent loads in flight), then the arithmetic: loaded
        // inside the row loop, each row would wait for its own round trip
        float zall[MT][4][NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m_wave + mt * 16 + grp * 4 + r;
                const int64_t base = (int64_t)(row < M ? row : 0) * ldc + t.n0;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    ISDQN_BOUNDS_CHECK(z + base + nt * 16 + li, 4, 11);
                    zall[mt][r][nt] = *(const ISDQN_GLOBAL float*)(z + base + nt * 16 + li);
                }
            }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m_wave + mt * 16 + grp * 4 + r;
                const bool row_ok = row < M;
                const int64_t base = (int64_t)(row_ok ? row : 0) * ldc + t.n0;
                float zv[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) zv[nt] = zall[mt][r][nt];
                float out[NT];
                if (gamma != nullptr) {
                    float s1 = 0.f, s2 = 0.f;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const bool ok = (nt * 16 + li) < c_in;
                        s1 += ok ? zv[nt] : 0.f;
                        s2 += ok ? zv[nt] * zv[nt] : 0.f;
                    }
                    s1 = row16_sum(s1);
                    s2 = row16_sum(s2);
                    const float mean = s1 * inv_c;
                    const float rstd = rsqrtf(fmaxf(s2 * inv_c - mean * mean, 0.f) + 1e-6f);
                    float xh[NT], gg[NT], m1 = 0.f, m2 = 0.f;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const bool ok = row_ok && (nt * 16 + li) < c_in;
                        xh[nt] = (zv[nt] - mean) * rstd;
                        const float y = xh[nt] * ga[nt] + be[nt];
                        const float dy = (ok && y > 0.f) ? acc[mt][nt][r] : 0.f;
                        dg[nt] += dy * xh[nt];
                        db[nt] += dy;
                        gg[nt] = dy * ga[nt];
                        m1 += gg[nt];
                        m2 += gg[nt] * xh[nt];
                    }
                    m1 = row16_sum(m1);
                    m2 = row16_sum(m2);
                    m1 *= inv_c;
                    m2 *= inv_c;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const bool ok = row_ok && (nt * 16 + li) < c_in;
                        out[nt] = ok ? rstd * (gg[nt] - m1 - xh[nt] * m2) : 0.f;
                        dbias[nt] += out[nt];
                    }
                } else {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const bool ok = row_ok && (nt * 16 + li) < c_in;
                        out[nt] = (ok && zv[nt] > 0.f) ? acc[mt][nt][r] : 0.f;
                        dbias[nt] += out[nt];
                    }
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) tr[(grp * 4 + r) * TR_PITCH + nt * 16 + li] = out[nt];
                if (r == 3) {
                    // the 16 rows of this row tile are staged (all four lane groups of THIS wave wrote them: the LDS unit
                    // serves a wave's operations in order, and the reads below are waited for before the next tile's writes)
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int chunk = lane + u * 64, rl = chunk >> 3, c8 = (chunk & 7) * 8;
                        const int orow = m_wave + mt * 16 + rl;
                        float v[8];
                        const f32x4 lo4 = *reinterpret_cast<const f32x4*>(tr + rl * TR_PITCH + c8);
                        const f32x4 hi4 = *reinterpret_cast<const f32x4*>(tr + rl * TR_PITCH + c8 + 4);
                        v[0] = lo4[0]; v[1] = lo4[1]; v[2] = lo4[2]; v[3] = lo4[3];
                        v[4] = hi4[0]; v[5] = hi4[1]; v[6] = hi4[2]; v[7] = hi4[3];
                        if (orow < M) s8_store_group(dz_out + (int64_t)orow * ldc + t.n0 + c8, v);
                    }
                }
            }
        // partial sums: over the 4 row groups of the wave, then over the 4 waves (fixed order)
        float* sp = reinterpret_cast<float*>(smem);  // [4 * KG waves][3][64] (in front of the transposition tiles)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            dg[nt] += __shfl_xor(dg[nt], 16); dg[nt] += __shfl_xor(dg[nt], 32);
            db[nt] += __shfl_xor(db[nt], 16); db[nt] += __shfl_xor(db[nt], 32);
            dbias[nt] += __shfl_xor(dbias[nt], 16); dbias[nt] += __shfl_xor(dbias[nt], 32);
            if (grp == 0) {
                sp[(wave * 3 + 0) * 64 + nt * 16 + li] = dg[nt];
                sp[(wave * 3 + 1) * 64 + nt * 16 + li] = db[nt];
                sp[(wave * 3 + 2) * 64 + nt * 16 + li] = dbias[nt];
            }
        }
        __syncthreads();
        const int tid = threadIdx.x;
        if (tid < 3 * 64) {
            const int which = tid / 64, c = tid % 64;
            float sum = sp[(0 * 3 + which) * 64 + c];
#pragma unroll
            for (int w = 1; w < 4 * KG; ++w) sum += sp[(w * 3 + which) * 64 + c];  // fixed order
            part[((int64_t)t.wg * 3 + which) * 64 + c] = sum;
        }
    }
};

// ---------------------------------------------------------------------------------------------
// Convolution geometry shared by the conv problems
// ---------------------------------------------------------------------------------------------
struct ConvGeom {
    int hin, win, cin_p, hout, wout, cout, cout_p, ksz, stride, pad, npix, K;
    int stride_sh;  // log2(stride): the torso's strides are 4, 2, 1 (dqn.py:55-69), divisions by the stride are shifts
    FastDiv d_npix, d_wout, d_cinp, d_ksz, d_coutp;
};

// Forward convolution + bias + LayerNorm(channels) + ReLU, fused (dqn.py:55-58, 62-65, 69-72).
//   M = output channels (BM >= cout_p, WM = 1), N = output pixels of all images, K = taps*cin_p.
template <int BM_, int PASSES_, bool U8>
struct ConvFwd {
    static constexpr int BM = BM_, BN = 128, WM = 1, WN = 4, PASSES = PASSES_;
    static constexpr bool A_TR = false, B_TR = false;
    static constexpr bool A_S8 = true;  // weights: S8 mirror
    static constexpr bool B_S8 = !U8;   // input activations: S8 (uint8 frames are converted)
    static constexpr int EPI_LDS_BYTES = 0;
    ConvGeom g;
    MatSrc W;            // [cout_p][K] (S8 mirror)
    const float* in;     // fp32 NHWC input (if !U8)
    FrameSrc fs;         // uint8 frames (if U8)
    const float *bias, *gamma, *beta;  // gamma == nullptr: no LayerNorm
    float scale;         // 1/255 folded into the first conv (dqn.py:51)
    float* act;          // [n_pix_total][cout_p]
    float* z;            // pre-LayerNorm output for the first z_pix pixels (needed by the backward)
    int n_pix_total, z_pix;
    struct Tile { int m0, n0, k0, k1; };
    struct ACtx { int row; };
    struct BCtx { int j, iy0, ix0, valid, f0, f1, f2, f3; };
    __device__ __forceinline__ bool tile(int bid, Tile& t) const {
        t.m0 = 0; t.n0 = bid * BN; t.k0 = 0; t.k1 = g.K;
        return t.n0 < n_pix_total;
    }
    __device__ __forceinline__ ACtx a_ctx(const Tile&, int row) const { return ACtx{row}; }
    __device__ __forceinline__ void load_a(const Tile&, const ACtx& c, int k, float (&v)[8]) const { W.load(c.row, k, v); }
    __device__ __forceinline__ BCtx b_ctx(const Tile&, int pix) const {
        BCtx c;
        c.valid = pix < n_pix_total;
        uint32_t j, p, oy, ox;
        g.d_npix.divmod(c.valid ? pix : 0, j, p);
        g.d_wout.divmod(p, oy, ox);
        c.j = (int)j;
        c.iy0 = (int)oy * g.stride - g.pad;
        c.ix0 = (int)ox * g.stride - g.pad;
        c.f0 = c.f1 = c.f2 = c.f3 = -1;
        if constexpr (U8) {  // the id-table lookup is hoisted out of the K loop (it would be a dependent load per chunk)
            c.f0 = fs.frame_id(c.j, 0);
            if (fs.stack > 1) c.f1 = fs.frame_id(c.j, 1);
            if (fs.stack > 2) c.f2 = fs.frame_id(c.j, 2);
            if (fs.stack > 3) c.f3 = fs.frame_id(c.j, 3);
        }
        return c;
    }
    __device__ __forceinline__ void load_b(const Tile&, const BCtx& c, int k, float (&v)[8]) const {
        const bool kok = c.valid && (k < g.K);
        if constexpr (U8) {
            const int plane = k >> 6;
            int id = plane == 0 ? c.f0 : plane == 1 ? c.f1 : plane == 2 ? c.f2 : c.f3;
            if (plane > 3) id = fs.frame_id(c.j, kok ? plane : 0);  // stacks deeper than 4: wave-uniform slow path
            fs.patch8_id(kok ? id : -1, c.iy0 + ((k >> 3) & 7), c.ix0, v);
        } else {
            uint32_t tap, ci, ky, kx;
            g.d_cinp.divmod(kok ? k : 0, tap, ci);
            g.d_ksz.divmod(tap, ky, kx);
            const int iy = c.iy0 + (int)ky, ix = c.ix0 + (int)kx;
            const bool ok = kok && iy >= 0 && iy < g.hin && ix >= 0 && ix < g.win;
            load8_aligned(ok ? in + ((((int64_t)c.j * g.hin + iy) * g.win + ix) * g.cin_p + ci) : zero_chunk(), v);
        }
    }
    template <int MT, int NT>
    __device__ __forceinline__ void epilogue(const Tile&, f32x4 (&acc)[MT][NT], int m_wave, int n_wave, int lane, char* smem) const {
        const int grp = lane >> 4;
        float bi[MT][4], ga[MT][4], be[MT][4];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int ch = m_wave + mt * 16 + grp * 4 + r;
                bool ok = ch < g.cout;
                bi[mt][r] = ok ? bias[ch] : 0.f;
                ga[mt][r] = (ok && gamma) ? gamma[ch] : 1.f;
                be[mt][r] = (ok && gamma) ? beta[ch] : 0.f;
            }
        const float inv_c = 1.0f / (float)g.cout;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int pix = n_wave + nt * 16 + (lane & 15);
            float zv[MT][4];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int ch = m_wave + mt * 16 + grp * 4 + r;
                    float zz = ch < g.cout ? acc[mt][nt][r] * scale + bi[mt][r] : 0.f;
                    zv[mt][r] = zz;
                    s1 += zz;
                    s2 += zz * zz;
                }
            float mean = 0.f, rstd = 1.f;
            if (gamma != nullptr) {  // wave-uniform
                s1 += __shfl_xor(s1, 16); s1 += __shfl_xor(s1, 32);
                s2 += __shfl_xor(s2, 16); s2 += __shfl_xor(s2, 32);
                mean = s1 * inv_c;
                float var = fmaxf(s2 * inv_c - mean * mean, 0.f);
                rstd = rsqrtf(var + 1e-6f);
            }
            if (pix < n_pix_total) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    int ch0 = m_wave + mt * 16 + grp * 4;
                    if (ch0 >= g.cout_p) continue;
                    float4 a, zq;
                    float* ap = &a.x;
                    float* zp = &zq.x;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float y = gamma != nullptr ? (zv[mt][r] - mean) * (rstd * ga[mt][r]) + be[mt][r] : zv[mt][r];
                        ap[r] = (ch0 + r < g.cout) ? fmaxf(y, 0.f) : 0.f;
                        zp[r] = zv[mt][r];
                    }
                    if (act != nullptr)  // (wave-uniform; the impala torso takes some outputs as fp32 `z` only)
                        s8_store_quad_paired(act + (int64_t)pix * g.cout_p, ch0, a.x, a.y, a.z, a.w);
                    if (pix < z_pix) *reinterpret_cast<float4*>(z + (int64_t)pix * g.cout_p + ch0) = zq;
                }
            }
        }
    }
};

// Data gradient of a strided SAME convolution, gather form, one stride-parity class per tile:
//   da[j, iy, ix, ci] = sum_{ky,kx,co} dz[j, (iy+pad-ky)/s, (ix+pad-kx)/s, co] * W[co][ky][kx][ci]
// Only taps with ky = (iy+pad) mod s (+ multiples of s) hit an output pixel, so input pixels are
// grouped by (iy mod s, ix mod s): every class is a dense GEMM with K = (ksz/s)^2 * cout_p.
//   M = input channels (A = weights, TR image: rows of W are output channels = contraction),
//   N = input pixels of the class, B = im2col(dz) ROW image.
template <int BM_, int PASSES_>
struct ConvDgrad {
    static constexpr int BM = BM_, BN = 128, WM = 1, WN = 4, PASSES = PASSES_;
    static constexpr bool A_TR = true, B_TR = false;
    static constexpr bool A_S8 = true;  // weights: S8 mirror
    static constexpr bool B_S8 = true;  // dz: S8
    static constexpr int EPI_LDS_BYTES = 0;
    ConvGeom g;
    const float* W;   // [cout_p][taps][cin_p] (S8 mirror)
    const float* dz;  // [n_img][hout][wout][cout_p]
    float* da;        // [n_img][hin][win][cin_p]
    int n_img, T, Kc; // T = ksz/stride taps per dim per class, Kc = T*T*cout_p
    int tile_start[17];  // prefix of tiles per class (stride^2 <= 16 classes: the 8x8 / 4 first convolution's data gradient exists
                         // only under BatchNorm, whose input site has parameters of its own)
    int n_classes;
    FastDiv cls_d_hw[16], cls_d_w[16];  // per class: divide by Ha*Wb and by Wb
    struct Tile { int m0, n0, k0, k1, cy, cx, py, px, Ha, Wb, Nc; FastDiv d_hw, d_w; };
    struct ACtx { int ci0; };
    struct BCtx { int j, oyb, oxb, valid; };
    __device__ __forceinline__ bool tile(int bid, Tile& t) const {
        int cls = 0;
        while (cls + 1 < n_classes && bid >= tile_start[cls + 1]) ++cls;
        if (bid >= tile_start[n_classes]) return false;
        t.cy = cls / g.stride; t.cx = cls % g.stride;
        t.Ha = (g.hin - t.cy + g.stride - 1) / g.stride;
        t.Wb = (g.win - t.cx + g.stride - 1) / g.stride;
        t.Nc = n_img * t.Ha * t.Wb;
        t.py = (t.cy + g.pad) % g.stride; t.px = (t.cx + g.pad) % g.stride;
        t.m0 = 0; t.n0 = (bid - tile_start[cls]) * BN; t.k0 = 0; t.k1 = Kc;
        t.d_hw = cls_d_hw[cls];
        t.d_w = cls_d_w[cls];
        return true;
    }
    __device__ __forceinline__ ACtx a_ctx(const Tile&, int ci0) const { return ACtx{ci0}; }
    __device__ __forceinline__ void load_a(const Tile& t, const ACtx& c, int k, float (&v)[8]) const {
        const bool ok = (k < Kc) && (c.ci0 < g.cin_p);
        uint32_t jt, co;
        g.d_coutp.divmod(ok ? k : 0, jt, co);
        int jy = (int)jt / T, jx = (int)jt % T;
        int ky = t.py + g.stride * jy, kx = t.px + g.stride * jx;
        load8_aligned(ok ? W + ((int64_t)co * g.K + (ky * g.ksz + kx) * g.cin_p + c.ci0) : zero_chunk(), v);
    }
    __device__ __forceinline__ void decode(const Tile& t, int q, int& j, int& iy, int& ix) const {
        uint32_t jj, rem, a, b;
        t.d_hw.divmod(q, jj, rem);
        t.d_w.divmod(rem, a, b);
        j = (int)jj; iy = t.cy + g.stride * (int)a; ix = t.cx + g.stride * (int)b;
    }
    __device__ __forceinline__ BCtx b_ctx(const Tile& t, int q) const {
        BCtx c;
        c.valid = q < t.Nc;
        int iy, ix;
        decode(t, c.valid ? q : 0, c.j, iy, ix);
        c.oyb = (iy + g.pad - t.py) / g.stride;
        c.oxb = (ix + g.pad - t.px) / g.stride;
        return c;
    }
    __device__ __forceinline__ void load_b(const Tile&, const BCtx& c, int k, float (&v)[8]) const {
        const bool kok = c.valid && (k < Kc);
        uint32_t jt, co;
        g.d_coutp.divmod(kok ? k : 0, jt, co);
        int jy = (int)jt / T, jx = (int)jt % T;
        int oy = c.oyb - jy, ox = c.oxb - jx;
        const bool ok = kok && oy >= 0 && oy < g.hout && ox >= 0 && ox < g.wout;
        load8_aligned(ok ? dz + ((((int64_t)c.j * g.hout + oy) * g.wout + ox) * g.cout_p + co) : zero_chunk(), v);
    }
    template <int MT, int NT>
    __device__ __forceinline__ void epilogue(const Tile& t, f32x4 (&acc)[MT][NT], int m_wave, int n_wave, int lane, char* smem) const {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            int q = n_wave + nt * 16 + (lane & 15);
            if (q >= t.Nc) continue;
            int j, iy, ix;
            decode(t, q, j, iy, ix);
            float* dst = da + (((int64_t)j * g.hin + iy) * g.win + ix) * g.cin_p;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                int ch0 = m_wave + mt * 16 + (lane >> 4) * 4;
                if (ch0 >= g.cin_p) continue;
                *reinterpret_cast<float4*>(dst + ch0) = float4{acc[mt][nt][0], acc[mt][nt][1], acc[mt][nt][2], acc[mt][nt][3]};
            }
        }
    }
};

// Weight gradient of a convolution: dW[co][k'] = sum_pix dz[pix][co] * im2col(in)[pix][k'].
// The contraction index (pixels) is the slow one of BOTH operands in memory -> both TR images.
//   M = cout, N = K' = taps*cin_p, K = online pixels, split over workgroups into fp32 slabs.
template <int PASSES_, bool U8>
struct ConvWgrad {
    static constexpr int BM = 64, BN = 64, WM = 2, WN = 2, PASSES = PASSES_;
    static constexpr bool A_TR = true, B_TR = true;
    static constexpr bool A_S8 = true;  // dz: S8
    static constexpr bool B_S8 = !U8;  // input activations: S8
    static constexpr int EPI_LDS_BYTES = 0;
    ConvGeom g;
    MatSrc DZ;        // [n_pix][cout_p]: outer = pixels (K), inner = cout_p
    const float* in;  // fp32 NHWC input (if !U8)
    FrameSrc fs;
    float* slabs;     // [splits][cout_p][K']
    float scale;      // 1/255 of the first conv (the network input is frames/255, dqn.py:51)
    int n_pix, tiles_n, splits, steps_per_split;
    struct Tile { int m0, n0, k0, k1, split; };
    struct ACtx { int co0; };
    struct BCtx { int c_or_ci, ky, kx, valid; };
    __device__ __forceinline__ bool tile(int bid, Tile& t) const {
        int tn = bid % tiles_n;
        t.split = bid / tiles_n;
        if (t.split >= splits) return false;
        t.m0 = 0; t.n0 = tn * BN;
        t.k0 = t.split * steps_per_split * GEMM_BK;
        t.k1 = min(n_pix, t.k0 + steps_per_split * GEMM_BK);
        return true;
    }
    __device__ __forceinline__ ACtx a_ctx(const Tile&, int co0) const { return ACtx{co0}; }
    __device__ __forceinline__ void load_a(const Tile&, const ACtx& c, int pix, float (&v)[8]) const { DZ.load(pix, c.co0, v); }
    __device__ __forceinline__ BCtx b_ctx(const Tile&, int kp) const {
        BCtx c;
        c.valid = kp < g.K;
        if constexpr (U8) {
            c.c_or_ci = kp >> 6; c.ky = (kp >> 3) & 7; c.kx = 0;
        } else {
            uint32_t tap, ci, ky, kx;
            g.d_cinp.divmod(c.valid ? kp : 0, tap, ci);
            g.d_ksz.divmod(tap, ky, kx);
            c.c_or_ci = (int)ci; c.ky = (int)ky; c.kx = (int)kx;
        }
        return c;
    }
    __device__ __forceinline__ void load_b(const Tile&, const BCtx& c, int pix, float (&v)[8]) const {
        const bool pok = c.valid && (pix < n_pix);
        uint32_t j, p, oy, ox;
        g.d_npix.divmod(pok ? pix : 0, j, p);
        g.d_wout.divmod(p, oy, ox);
        int iy = (int)oy * g.stride - g.pad + c.ky;
        int ix = (int)ox * g.stride - g.pad + c.kx;
        if constexpr (U8) {
            int id = fs.frame_id((int)j, c.c_or_ci);
            fs.patch8_id(pok ? id : -1, iy, ix, v);
        } else {
            const bool ok = pok && iy >= 0 && iy < g.hin && ix >= 0 && ix < g.win;
            load8_aligned(ok ? in + ((((int64_t)j * g.hin + iy) * g.win + ix) * g.cin_p + c.c_or_ci) : zero_chunk(), v);
        }
    }
    template <int MT, int NT>
    __device__ __forceinline__ void epilogue(const Tile& t, f32x4 (&acc)[MT][NT], int m_wave, int n_wave, int lane, char* smem) const {
        float* c = slabs + (int64_t)t.split * g.cout_p * g.K;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                int col = n_wave + nt * 16 + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int row = m_wave + mt * 16 + (lane >> 4) * 4 + r;
                    if (row < g.cout_p && col < g.K) c[(int64_t)row * g.K + col] = acc[mt][nt][r] * scale;
                }
            }
    }
};

}  // namespace isdqn
