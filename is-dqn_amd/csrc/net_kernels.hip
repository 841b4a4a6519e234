// iS-DQN Q-network on gfx950: forward over the 1+K shared heads, iterated Bellman target, squared TD
// loss, backward and Adam.  Reference: slimdqn/networks/architectures/dqn.py:47-103 and
// slimdqn/networks/isdqn.py:82-135.  Contractions run on the MFMA tile engine (gemm_core.h,
// net_problems.h); the row-wise kernels below are the HBM-bound remainder.
#include <stdarg.h>
#include <stdlib.h>

#include <algorithm>
#include <type_traits>
#include <vector>
#include <atomic>
#include <functional>

#include "conv_img.h"
#include "conv_u8_pair.h"
#include "conv_s8_pair.h"
#include "net_plan.h"
#include "net_problems.h"

namespace isdqn {

// b^t for the Adam bias corrections 1 - b^t (optax.adam, isdqn.py:46): exponentiation by squaring, at most 62 double
// multiplies (relative error ~1e-15).  The library pow() is thousands of instructions on ONE lane -- 10-20 us -- and
// the kernel that owns the step counter lasts as long as that lane.
__device__ inline double pow_int(double b, int t) {
    double r = 1.0, x = b;
    for (unsigned u = (unsigned)t; u != 0; u >>= 1) {
        if (u & 1u) r *= x;
        x *= x;
    }
    return r;
}

// =============================================================================================
// Row-wise kernels
// =============================================================================================

// Dense layer tail (dqn.py:96-99): sum the split-K slabs, add the bias, LayerNorm over the row,
// ReLU.  One workgroup per row; the row is staged in LDS between the two passes.
__global__ __launch_bounds__(256) void dense_post_kernel(const float* __restrict__ slabs, int n_slabs,
                                                         int64_t slab_stride, int64_t row_pitch, int rows, int F, int Fp,
                                                         const float* __restrict__ bias,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, int has_relu,
                                                         float* __restrict__ act, float* __restrict__ z, int z_rows, int act_s8) {
    extern __shared__ float s_row[];  // [3][Fp]: the summed row, gamma, beta
    __shared__ float s_red[2][4];
    const int row = blockIdx.x, tid = threadIdx.x;
    float* s_g = s_row + Fp;
    float* s_b = s_row + 2 * Fp;
    float s1 = 0.f, s2 = 0.f;
    // two adjacent columns per thread (Fp is a multiple of 8: 8-byte loads), 16 slabs in flight, fixed summation order.
    // bias / gamma / beta of the two columns are requested FIRST and travel under the slab loads: a workgroup lives
    // for a few microseconds, and three dependent round trips behind the sums were a third of that.
    for (int c = 2 * tid; c < Fp; c += 512) {
        float par[3][2];
#pragma unroll
        for (int w = 0; w < 3; ++w) {
            const float* src = w == 0 ? bias : w == 1 ? gamma : beta;
#pragma unroll
            for (int e = 0; e < 2; ++e)
            {
                ISDQN_BOUNDS_CHECK((src != nullptr && c + e < F) ? src + c + e : zero_chunk(), 4, 28);
                par[w][e] = *(const ISDQN_GLOBAL float*)((src != nullptr && c + e < F) ? src + c + e : zero_chunk());
            }
        }
        float v0 = 0.f, v1 = 0.f;
        const float* p = slabs + (int64_t)row * row_pitch + c;
        int s = 0;
        for (; s + 16 <= n_slabs; s += 16) {
            float2 t[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) t[u] = *reinterpret_cast<const float2*>(p + (int64_t)(s + u) * slab_stride);
#pragma unroll
            for (int u = 0; u < 16; ++u) { v0 += t[u].x; v1 += t[u].y; }
        }
        if (s < n_slabs) {
            float2 t[16];
#pragma unroll
            for (int u = 0; u < 16; ++u)  // tail: clamped to the last slab, masked in the sum
                t[u] = *reinterpret_cast<const float2*>(p + (int64_t)min(s + u, n_slabs - 1) * slab_stride);
#pragma unroll
            for (int u = 0; u < 16; ++u) { v0 += (s + u < n_slabs) ? t[u].x : 0.f; v1 += (s + u < n_slabs) ? t[u].y : 0.f; }
        }
        v0 = c < F ? v0 + par[0][0] : 0.f;
        v1 = c + 1 < F ? v1 + par[0][1] : 0.f;
        s_row[c] = v0;
        s_row[c + 1] = v1;
        s_g[c] = par[1][0]; s_g[c + 1] = par[1][1];
        s_b[c] = par[2][0]; s_b[c + 1] = par[2][1];
        s1 += v0 + v1;
        s2 += v0 * v0 + v1 * v1;
    }
    float mean = 0.f, rstd = 1.f;
    if (gamma != nullptr) {
        for (int off = 32; off > 0; off >>= 1) {
            s1 += __shfl_xor(s1, off);
            s2 += __shfl_xor(s2, off);
        }
        if ((tid & 63) == 0) {
            s_red[0][tid >> 6] = s1;
            s_red[1][tid >> 6] = s2;
        }
        __syncthreads();
        s1 = s_red[0][0] + s_red[0][1] + s_red[0][2] + s_red[0][3];
        s2 = s_red[1][0] + s_red[1][1] + s_red[1][2] + s_red[1][3];
        mean = s1 / (float)F;
        float var = fmaxf(s2 / (float)F - mean * mean, 0.f);
        rstd = rsqrtf(var + 1e-6f);
    }
    for (int c0 = 2 * tid; c0 < Fp; c0 += 512) {
        float y2[2], v2[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int c = c0 + e;
            const float v = s_row[c];  // (row, gamma, beta: written by this same thread)
            float y = v;
            if (gamma != nullptr && c < F) y = (v - mean) * (rstd * s_g[c]) + s_b[c];
            if (has_relu) y = fmaxf(y, 0.f);
            if (c >= F) y = 0.f;
            y2[e] = y; v2[e] = v;
        }
        if (act_s8) s8_store_pair(act + (int64_t)row * Fp, c0, y2[0], y2[1]);  // hidden activations: S8; the head's q stays fp32
        else *reinterpret_cast<float2*>(act + (int64_t)row * Fp + c0) = make_float2(y2[0], y2[1]);
        if (row < z_rows) *reinterpret_cast<float2*>(z + (int64_t)row * Fp + c0) = make_float2(v2[0], v2[1]);
    }
}

// LayerNorm + ReLU backward over the last axis (rows x C, C <= 4*TPR), TPR lanes per row, float4 per lane.
//   xhat = (z-mean)*rstd ; y = xhat*gamma+beta ; dy = da*[y>0] ; dgamma += dy*xhat ; dbeta += dy
//   g = dy*gamma ; dz = rstd*(g - mean(g) - xhat*mean(g*xhat)) ; dbias += dz
// Per-workgroup partial sums go to part[block][3][Cp] (deterministic; reduced by the Adam kernel).
template <int TPR>
__global__ __launch_bounds__(256) void ln_bwd_small_kernel(const float* __restrict__ da, const float* __restrict__ z,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, int rows, int C, int Cp,
                                                           float* __restrict__ dz, float* __restrict__ part) {
    constexpr int RPB = 256 / TPR;
    __shared__ float s_part[RPB][3][4 * TPR];
    const int tid = threadIdx.x, grp = tid / TPR, sub = tid % TPR;
    const int ch0 = sub * 4;
    const bool lane_on = ch0 < Cp;
    float ga[4], be[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        bool ok = lane_on && (ch0 + r) < C;
        ga[r] = (ok && gamma) ? gamma[ch0 + r] : 1.f;
        be[r] = (ok && gamma) ? beta[ch0 + r] : 0.f;
    }
    float dg[4] = {0, 0, 0, 0}, db[4] = {0, 0, 0, 0}, dbias[4] = {0, 0, 0, 0};
    const float inv_c = 1.f / (float)C;
    for (int row0 = blockIdx.x * RPB; row0 < rows; row0 += gridDim.x * RPB) {
        const int row = row0 + grp;
        const bool on = lane_on && row < rows;
        float zv[4] = {0, 0, 0, 0}, dv[4] = {0, 0, 0, 0};
        if (on) {
            float4 a = *reinterpret_cast<const float4*>(z + (int64_t)row * Cp + ch0);
            float4 b = *reinterpret_cast<const float4*>(da + (int64_t)row * Cp + ch0);
            zv[0] = a.x; zv[1] = a.y; zv[2] = a.z; zv[3] = a.w;
            dv[0] = b.x; dv[1] = b.y; dv[2] = b.z; dv[3] = b.w;
        }
        float out[4];
        if (gamma != nullptr) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                bool ok = (ch0 + r) < C;
                s1 += ok ? zv[r] : 0.f;
                s2 += ok ? zv[r] * zv[r] : 0.f;
            }
#pragma unroll
            for (int off = TPR / 2; off > 0; off >>= 1) {
                s1 += __shfl_xor(s1, off);
                s2 += __shfl_xor(s2, off);
            }
            float mean = s1 * inv_c;
            float rstd = rsqrtf(fmaxf(s2 * inv_c - mean * mean, 0.f) + 1e-6f);
            float xh[4], gg[4], m1 = 0.f, m2 = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                bool ok = on && (ch0 + r) < C;
                xh[r] = (zv[r] - mean) * rstd;
                float y = xh[r] * ga[r] + be[r];
                float dy = (ok && y > 0.f) ? dv[r] : 0.f;
                dg[r] += dy * xh[r];
                db[r] += dy;
                gg[r] = dy * ga[r];
                m1 += gg[r];
                m2 += gg[r] * xh[r];
            }
#pragma unroll
            for (int off = TPR / 2; off > 0; off >>= 1) {
                m1 += __shfl_xor(m1, off);
                m2 += __shfl_xor(m2, off);
            }
            m1 *= inv_c;
            m2 *= inv_c;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                bool ok = on && (ch0 + r) < C;
                out[r] = ok ? rstd * (gg[r] - m1 - xh[r] * m2) : 0.f;
                dbias[r] += out[r];
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                bool ok = on && (ch0 + r) < C;
                out[r] = (ok && zv[r] > 0.f) ? dv[r] : 0.f;
                dbias[r] += out[r];
            }
        }
        if (on) s8_store_quad(dz + (int64_t)row * Cp, ch0, out[0], out[1], out[2], out[3]);  // dz: S8
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        s_part[grp][0][ch0 + r] = dg[r];
        s_part[grp][1][ch0 + r] = db[r];
        s_part[grp][2][ch0 + r] = dbias[r];
    }
    __syncthreads();
    for (int i = tid; i < 3 * Cp; i += 256) {
        int which = i / Cp, c = i % Cp;
        float s = 0.f;
        for (int g2 = 0; g2 < RPB; ++g2) s += s_part[g2][which][c];
        part[((int64_t)blockIdx.x * 3 + which) * Cp + c] = s;
    }
}

// Same, wide rows (dense layers): one workgroup per row at a time, columns strided over the threads.
constexpr int LN_WIDE_MAX_COLS = 16;  // per thread: widths up to 4096
__global__ __launch_bounds__(256) void ln_bwd_wide_kernel(const float* __restrict__ da, const float* __restrict__ z,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, int rows, int C, int Cp,
                                                          float* __restrict__ dz, float* __restrict__ part) {
    __shared__ float s_red[4][4];
    const int tid = threadIdx.x;
    float dg[LN_WIDE_MAX_COLS], db[LN_WIDE_MAX_COLS], dbias[LN_WIDE_MAX_COLS];
#pragma unroll
    for (int i = 0; i < LN_WIDE_MAX_COLS; ++i) dg[i] = db[i] = dbias[i] = 0.f;
    const float inv_c = 1.f / (float)C;
    for (int row = blockIdx.x; row < rows; row += gridDim.x) {
        const float* zr = z + (int64_t)row * Cp;
        const float* dr = da + (int64_t)row * Cp;
        float* outr = dz + (int64_t)row * Cp;
        if (gamma != nullptr) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < LN_WIDE_MAX_COLS; ++i) {
                int c = tid + i * 256;
                if (c < C) {
                    float v = zr[c];
                    s1 += v;
                    s2 += v * v;
                }
            }
            for (int off = 32; off > 0; off >>= 1) {
                s1 += __shfl_xor(s1, off);
                s2 += __shfl_xor(s2, off);
            }
            __syncthreads();
            if ((tid & 63) == 0) { s_red[0][tid >> 6] = s1; s_red[1][tid >> 6] = s2; }
            __syncthreads();
            s1 = s_red[0][0] + s_red[0][1] + s_red[0][2] + s_red[0][3];
            s2 = s_red[1][0] + s_red[1][1] + s_red[1][2] + s_red[1][3];
            float mean = s1 * inv_c;
            float rstd = rsqrtf(fmaxf(s2 * inv_c - mean * mean, 0.f) + 1e-6f);
            float m1 = 0.f, m2 = 0.f;
#pragma unroll
            for (int i = 0; i < LN_WIDE_MAX_COLS; ++i) {
                int c = tid + i * 256;
                if (c < C) {
                    float xh = (zr[c] - mean) * rstd;
                    float y = xh * gamma[c] + beta[c];
                    float dy = y > 0.f ? dr[c] : 0.f;
                    dg[i] += dy * xh;
                    db[i] += dy;
                    float g2 = dy * gamma[c];
                    m1 += g2;
                    m2 += g2 * xh;
                }
            }
            for (int off = 32; off > 0; off >>= 1) {
                m1 += __shfl_xor(m1, off);
                m2 += __shfl_xor(m2, off);
            }
            if ((tid & 63) == 0) { s_red[2][tid >> 6] = m1; s_red[3][tid >> 6] = m2; }
            __syncthreads();
            m1 = (s_red[2][0] + s_red[2][1] + s_red[2][2] + s_red[2][3]) * inv_c;
            m2 = (s_red[3][0] + s_red[3][1] + s_red[3][2] + s_red[3][3]) * inv_c;
#pragma unroll
            for (int i = 0; i < LN_WIDE_MAX_COLS; ++i) {
                int c = tid + i * 256;
                if (c < Cp) {
                    float o = 0.f;
                    if (c < C) {
                        float xh = (zr[c] - mean) * rstd;
                        float y = xh * gamma[c] + beta[c];
                        float dy = y > 0.f ? dr[c] : 0.f;
                        o = rstd * (dy * gamma[c] - m1 - xh * m2);
                    }
                    dbias[i] += o;
                    s8_store_elem(outr, c, o);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < LN_WIDE_MAX_COLS; ++i) {
                int c = tid + i * 256;
                if (c < Cp) {
                    float o = (c < C && zr[c] > 0.f) ? dr[c] : 0.f;
                    dbias[i] += o;
                    s8_store_elem(outr, c, o);
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < LN_WIDE_MAX_COLS; ++i) {
        int c = tid + i * 256;
        if (c < Cp) {
            float* p = part + (int64_t)blockIdx.x * 3 * Cp;
            p[c] = dg[i];
            p[Cp + c] = db[i];
            p[2 * Cp + c] = dbias[i];
        }
    }
}

// Iterated Bellman target + squared TD loss (isdqn.py:92-109).
//   q_k(s,a) from head 1+k of the online rows, target_k = r + (1-terminal)*gamma^n*max_a' Q_k(s',a')
//   from head k of the next-state rows (same parameters, stop-gradient), td = (q - target)^2.
// One workgroup per 64 transitions, wave w takes heads k = w, w+4, ...  Emits dL/dq rows (dout),
// q_values/targets/priorities and per-workgroup partials of the per-head loss sums and of the head
// bias gradient (column sums of dout); loss_finalize_kernel reduces them in a fixed order.
// Per-element loss and its derivative w.r.t. q for d = q - target: squared error (the reference, isdqn.py:102) or, with
// huber_delta > 0, the Huber loss (0.5 d^2 inside, delta (|d| - delta / 2) outside; derivative clip(d, -delta, delta)).
__device__ __forceinline__ float td_loss(float d, float huber_delta) {
    const float a = fabsf(d);
    return huber_delta > 0.f ? (a <= huber_delta ? 0.5f * d * d : huber_delta * (a - 0.5f * huber_delta)) : d * d;
}
__device__ __forceinline__ float td_dloss(float d, float huber_delta) {
    return huber_delta > 0.f ? fminf(fmaxf(d, -huber_delta), huber_delta) : 2.f * d;
}
constexpr int TD_ROWS = 64;
__global__ __launch_bounds__(256) void td_kernel(const float* __restrict__ q, int B, int K, int oh, int th, int A, int nha_p,
                                                 const int* __restrict__ action, const float* __restrict__ reward,
                                                 const uint8_t* __restrict__ terminal, float gamma_n, float huber_delta,
                                                 float* __restrict__ dout, float* __restrict__ q_values,
                                                 float* __restrict__ targets, double* __restrict__ priorities,
                                                 float* __restrict__ loss_part, float* __restrict__ dbh_part) {
    extern __shared__ float s_d[];  // [TD_ROWS][K] : 2*(q-target)/B ; then [TD_ROWS][K] td
    __shared__ int s_action[TD_ROWS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b0 = blockIdx.x * TD_ROWS;
    const int b = b0 + lane;
    const bool on = b < B;
    float* s_td = s_d + TD_ROWS * K;
    const float inv_b = 1.f / (float)B;
    if (dout != nullptr) {
        const int rows = min(TD_ROWS, B - b0);
        for (int i = tid; i < rows * nha_p; i += 256) dout[(int64_t)b0 * nha_p + i] = 0.f;
    }
    int a = 0;
    float r = 0.f, nt = 0.f;
    if (on) {
        a = action[b];
        r = reward[b];
        nt = 1.f - (float)terminal[b];
    }
    if (wave == 0) s_action[lane] = on ? a : -1;
    for (int k = wave; k < K; k += 4) {
        float d = 0.f, td = 0.f;
        if (on) {
            float qv = q[(int64_t)b * nha_p + (oh + k) * A + a];
            const float* nq = q + (int64_t)(B + b) * nha_p + (th + k) * A;
            float mx = nq[0];
            for (int j = 1; j < A; ++j) mx = fmaxf(mx, nq[j]);
            float tg = r + nt * gamma_n * mx;
            d = qv - tg;
            td = td_loss(d, huber_delta);
            if (q_values) q_values[(int64_t)b * K + k] = qv;
            if (targets) targets[(int64_t)b * K + k] = tg;
        }
        s_d[lane * K + k] = td_dloss(d, huber_delta) * inv_b;
        s_td[lane * K + k] = td;
        float sum = td;
        for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
        if (lane == 0) loss_part[(int64_t)blockIdx.x * K + k] = sum;
    }
    __syncthreads();  // dout zero-fill (this workgroup's rows) and s_d / s_td complete
    if (dout != nullptr) {
        for (int i = tid; i < TD_ROWS * K; i += 256) {
            int bl = i / K, k = i - bl * K;
            if (b0 + bl < B) dout[(int64_t)(b0 + bl) * nha_p + (oh + k) * A + s_action[bl]] = s_d[i];
        }
        // column sums over this workgroup's rows: column (oh+k)*A + a' collects rows whose action is a'
        // (round 3: the regressed heads start at `oh`, which is 0 for the single-head baselines -- the test `head >= 1` left the
        // head-bias gradient of DQN / of TF-DQN off the head chain at zero)
        for (int c = tid; c < nha_p; c += 256) {
            float sum = 0.f;
            int head = c / A, aa = c - head * A;
            if (head >= oh && head < oh + K)
                for (int bl = 0; bl < TD_ROWS; ++bl) sum += (s_action[bl] == aa) ? s_d[bl * K + head - oh] : 0.f;
            dbh_part[(int64_t)blockIdx.x * nha_p + c] = sum;
        }
    }
    if (priorities != nullptr && tid < TD_ROWS && b0 + tid < B) {
        float sum = 0.f;
        for (int k = 0; k < K; ++k) sum += s_td[tid * K + k];
        priorities[b0 + tid] = sqrt((double)(sum / (float)K) + 1e-10);
    }
}

// ---------------------------------------------------------------------------------------------
// Head chain (learn path).  Everything between the last hidden activation and the gradient w.r.t. that
// layer's pre-activation is per-transition work on a few hundred floats, so one kernel does it for S
// transitions per workgroup instead of five latency-bound launches:
//   q = a W_head^T + b           (dqn.py:100-103; online rows 0..S-1 and next-state rows S..2S-1 form one
//                                  16-row MFMA tile, the waves split the contraction and reduce through LDS)
//   iterated Bellman target, squared TD loss, dL/dq   (isdqn.py:92-109, as td_kernel)
//   da = dL/dq W_head            (dL/dq has one non-zero per head and transition: K fp32 row-AXPYs, exact)
//   dz = backward of a = relu(LN(z)) over the hidden row (as ln_bwd_wide_kernel), partial (dgamma, dbeta,
//        dbias) sums per workgroup to part[wg][3][Fp]
// Workgroup 0 also advances the Adam step counter (it used to ride on loss_finalize_kernel, which now runs
// off the critical path).
// ---------------------------------------------------------------------------------------------
struct HeadChainParams {
    // hidden dense layer, finished inside this kernel for the rows a workgroup owns (dense_post_kernel's work)
    const float* slabs;     // [2B][n_slabs][Fp] split-K partial products of the hidden layer (row-interleaved)
    int n_slabs;
    int64_t slab_stride, row_pitch;  // Fp, n_slabs * Fp
    const float* hbias;     // [F] hidden bias
    float* act;             // [2B][Fp] post-activation rows (rows [0,B): states, [B,2B): next states), written here
    float* z;               // [B][Fp]  pre-LayerNorm rows of the states, written here
    const float* W;     // [O][Fp]
    const float* bias;  // [O]
    const float* gamma; // hidden LayerNorm scale / bias, or null
    const float* beta;
    int B, S, F, Fp, O, Op, K, A, oh;  // oh: online head k + oh is regressed on head k (1: iS-DQN, 0: single head)
    const int* action;
    const float* reward;
    const uint8_t* terminal;
    float gamma_n, huber_delta;
    float* dout;        // [B][Op]  dL/dq
    float* dz;          // [B][Fp]
    float* part;        // [n_wg][3][Fp]
    float* q_values;    // [B][K]
    float* targets;     // [B][K]
    double* priorities; // [B] or null
    float* loss_part;   // [n_wg][K]
    float* dbh_part;    // [n_wg][Op]
    int* adam_count;
    float b1, b2;
    float* adam_consts;
    long long* stamps;  // profiling only (isdqn_debug_set_stamps "head_chain"): [workgroup][8] phase boundaries
};

// Transitions per workgroup (template parameter SMAX = 1, 2 or 4; HC_MAX_S in net_plan.h is the largest).  The kernel
// is instruction-issue bound (a few thousand instructions executed once per wave), and the target / data-gradient /
// LayerNorm phases scale with S, so few transitions on many workgroups win -- until every one of too many workgroups
// streams the whole head matrix from L2: S = 1 at B = 256 (20 us; S = 2: 23 us; S = 4 on 256 threads: 22 us before the
// dense_post fusion), S = 4 at B = 1024 (learn_or_loss picks about 256 workgroups).
constexpr int HC_MAX_COLS = 4;  // hidden width up to HC_THREADS * 4 (columns per thread: template parameter COLS)
// Eight waves: the kernel executes a few thousand instructions ONCE per wave, so it is bound by instruction issue and
// its dependency stalls; two waves per SIMD interleave, and each owns half the K-steps / columns.
constexpr int HC_THREADS = 512, HC_WAVES = HC_THREADS / 64;
constexpr int HC_KU = 16 / HC_WAVES;  // K-steps per wave and work item (an item covers 16 K-steps)

__host__ __device__ inline int head_chain_pitch(int Fp) { return (Fp + 31) / 32 * 32 + 8; }
static inline int head_chain_lds_bytes(int Fp, int Op, int K, int passes, int S) {
    const int PA = head_chain_pitch(Fp);
    return (passes >= 2 ? 2 : 1) * 16 * PA * 2 + (HC_WAVES * 16 * Op + 16 * Op + S * Op + Op + 4 + 2 * S * Fp) * 4 + S * K * 8 +
           HC_WAVES * 2 * S * 2 * 4 + 64;
}

template <int PASSES, int COLS, int SMAX>
__global__ __launch_bounds__(HC_THREADS) void head_chain_kernel(const HeadChainParams p) {
    ISDQN_EMPTY_KERNEL_RETURN
    extern __shared__ __attribute__((aligned(16))) char hc_smem[];
    const int PA = head_chain_pitch(p.Fp);
    constexpr int A_PLANES = PASSES >= 2 ? 2 : 1;
    __bf16* a_hi = reinterpret_cast<__bf16*>(hc_smem);
    __bf16* a_lo = a_hi + 16 * PA;
    float* qpart = reinterpret_cast<float*>(a_hi + A_PLANES * 16 * PA);  // [HC_WAVES][16][Op]
    float* s_q = qpart + HC_WAVES * 16 * p.Op;                                  // [16][Op]
    float* s_dq = s_q + 16 * p.Op;                                       // [SMAX][Op]
    float* s_d = s_dq + SMAX * p.Op;                                 // [SMAX][K]
    float* s_td = s_d + SMAX * p.K;
    float* s_red = s_td + SMAX * p.K;                                // [HC_WAVES][2 * SMAX][2]
    float* s_bias = s_red + HC_WAVES * 2 * SMAX * 2;                 // [Op]
    float* s_pre = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(s_bias + p.Op) + 15) & ~(uintptr_t)15);  // [2 * SMAX][Fp] hidden pre-activations, 16-B aligned

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#if defined(ISDQN_DEV)
#define HC_STAMP(i)                                                                                     \
    if (p.stamps != nullptr && threadIdx.x == 0) {                                                     \
        p.stamps[(int64_t)blockIdx.x * 8 + (i)] = (long long)__builtin_amdgcn_s_memtime();              \
        if ((i) == 0) p.stamps[(int64_t)blockIdx.x * 8 + 7] = (long long)__builtin_amdgcn_s_memrealtime(); \
    }
#else
#define HC_STAMP(i)
#endif
    HC_STAMP(0);
    const int S = p.S, b0 = (int)blockIdx.x * S;
    const int Fp = p.Fp, Op = p.Op, K = p.K, A = p.A;

    if (blockIdx.x == 0 && tid == 0) {
        int t = *p.adam_count + 1;
        *p.adam_count = t;
        p.adam_consts[0] = (float)(1.0 - pow_int((double)p.b1, t));
        p.adam_consts[1] = (float)(1.0 - pow_int((double)p.b2, t));
    }

    // Global loads are requested a phase before they are needed, and nothing touches a loaded register where it is
    // requested (the data were last written from other XCDs: a first touch costs ~4000 cycles).
    // ---- operands of the later phases (raw: converted where they are used) ----
    int act[SMAX];  // action of the S transitions (selects the head rows the data gradient reads)
#pragma unroll
    for (int s = 0; s < SMAX; ++s) {
        ISDQN_BOUNDS_CHECK((s < S && b0 + s < p.B) ? (const void*)(p.action + b0 + s) : zero_chunk(), 4, 20);
        act[s] = *(const ISDQN_GLOBAL int*)((s < S && b0 + s < p.B) ? (const void*)(p.action + b0 + s) : zero_chunk());
    }
    float td_r;          // reward, terminal flag of this thread's (transition, head) pair
    uint8_t td_term;
    {
        const int s = tid / K;
        const bool ok = tid < S * K && b0 + s < p.B;
        ISDQN_BOUNDS_CHECK(ok ? (const void*)(p.reward + b0 + s) : zero_chunk(), 4, 21);
        ISDQN_BOUNDS_CHECK(ok ? (const void*)(p.terminal + b0 + s) : zero_chunk(), 1, 22);
        td_r = *(const ISDQN_GLOBAL float*)(ok ? (const void*)(p.reward + b0 + s) : zero_chunk());
        td_term = *(const ISDQN_GLOBAL uint8_t*)(ok ? (const void*)(p.terminal + b0 + s) : zero_chunk());
    }
    ISDQN_BOUNDS_CHECK(tid < p.O ? (const void*)(p.bias + tid) : zero_chunk(), 4, 23);
    const float bias_v = *(const ISDQN_GLOBAL float*)(tid < p.O ? (const void*)(p.bias + tid) : zero_chunk());
    float ga[COLS], be[COLS];
#pragma unroll
    for (int j = 0; j < COLS; ++j) {
        const int c = tid + j * HC_THREADS;
        const bool ok = p.gamma != nullptr && c < p.F;
        ISDQN_BOUNDS_CHECK(ok ? (const void*)(p.gamma + c) : zero_chunk(), 4, 24);
        ISDQN_BOUNDS_CHECK(ok ? (const void*)(p.beta + c) : zero_chunk(), 4, 24);
        ga[j] = *(const ISDQN_GLOBAL float*)(ok ? (const void*)(p.gamma + c) : zero_chunk());
        be[j] = *(const ISDQN_GLOBAL float*)(ok ? (const void*)(p.beta + c) : zero_chunk());
    }

    // ---- the hidden dense layer is finished HERE for the 2S rows of this workgroup (dqn.py:96-99; what
    //      dense_post_kernel does in the forward-only path): split-K slabs summed in slab order, + bias, LayerNorm, ReLU.
    //      The rows go to HBM once (act: the head's weight gradient reads the state rows; z: debug / tests), to LDS as
    //      bf16 hi/lo for the head GEMM, and the pre-LayerNorm values stay in registers for the backward below.
    //      Tile rows 2S..15 are not staged: MFMA output rows depend on their own A row only, and nothing reads the q
    //      rows of those tile rows ----
    constexpr int R2 = 2 * SMAX;
    constexpr int TPR = HC_THREADS / R2;  // threads per tile row (two waves): 16-byte loads, one row per thread
    constexpr int SB = 32;                // slabs per batch = loads in flight per thread
    static_assert(TPR % 64 == 0, "a wave must not straddle two rows");
    float m1 = 0.f, m2 = 0.f;  // LayerNorm sums of this thread's row (over its columns)
    {
        const int r = tid / TPR, q = tid - r * TPR;
        const int smp = r < S ? r : r - S;
        const bool row_ok = r < 2 * S && b0 + smp < p.B;
        const int64_t grow = r < S ? (int64_t)(b0 + smp) : (int64_t)p.B + b0 + smp;
        for (int c0 = q * 4; c0 < Fp; c0 += TPR * 4) {  // (one pass up to 512 columns)
            float hb4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                ISDQN_BOUNDS_CHECK(c0 + e < p.F ? (const void*)(p.hbias + c0 + e) : zero_chunk(), 4, 25);
                hb4[e] = *(const ISDQN_GLOBAL float*)(c0 + e < p.F ? (const void*)(p.hbias + c0 + e) : zero_chunk());
            }
            const ISDQN_GLOBAL f32x4* base = (const ISDQN_GLOBAL f32x4*)(row_ok ? (const void*)(p.slabs + grow * p.row_pitch + c0) : zero_chunk());
            const int64_t strd = row_ok ? p.slab_stride / 4 : 0;  // (Fp is a multiple of 8)
            f32x4 sum = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int s0 = 0; s0 < p.n_slabs; s0 += SB) {
                f32x4 t[SB];
#pragma unroll
                for (int u = 0; u < SB; ++u) {
                    ISDQN_BOUNDS_CHECK((const void*)(base + min(s0 + u, p.n_slabs - 1) * strd), 16, 26);
                    t[u] = base[min(s0 + u, p.n_slabs - 1) * strd];  // tail: clamped, masked in the sum
                }
#pragma unroll
                for (int u = 0; u < SB; ++u)
                    if (s0 + u < p.n_slabs) sum += t[u];  // (uniform condition; slab order)
            }
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = c0 + e < p.F ? sum[e] + hb4[e] : 0.f;
                m1 += v[e];
                m2 += v[e] * v[e];
            }
            *reinterpret_cast<float4*>(s_pre + r * Fp + c0) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }

    HC_STAMP(6);  // slab sums of this thread's row done
    // ---- q = a W^T work items, see below; the first batch of W fragments travels under the LayerNorm ----
    const int nkt = (PA - 8) / 32, NT = (Op + 15) / 16;
    const int nks = (nkt + 15) / 16, n_items = ((NT + 1) / 2) * nks;
    const int kg8 = (lane >> 4) * 8;
    const int arow = (lane & 15) * PA + kg8;
    auto issue_w = [&](float (&v)[2][HC_KU][8], int item) {
        const int pair = item / nks, ksb = wave + 16 * (item - pair * nks);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int u = 0; u < HC_KU; ++u) {
                const int o = (pair * 2 + t) * 16 + (lane & 15), ks = ksb + HC_WAVES * u;
                const bool ok = item < n_items && o < p.O && ks < nkt && ks * 32 + kg8 < Fp;
                load8_aligned(ok ? p.W + (int64_t)o * Fp + ks * 32 + kg8 : zero_chunk(), v[t][u]);
            }
    };
    float wv0[2][HC_KU][8], wv1[2][HC_KU][8];
    issue_w(wv0, 0);

    float zv[SMAX][COLS];  // pre-LayerNorm values of the state rows (zero outside the row / the transition range)
    {
        for (int off = 32; off > 0; off >>= 1) {
            m1 += __shfl_xor(m1, off);
            m2 += __shfl_xor(m2, off);
        }
        if (lane == 0) {
            s_red[wave * 2] = m1;
            s_red[wave * 2 + 1] = m2;
        }
        __syncthreads();  // row sums and the pre-activation rows (s_pre) are visible
        constexpr int WPR = TPR / 64;  // waves per row
#pragma unroll
        for (int r = 0; r < R2; ++r) {
            float a1 = s_red[r * WPR * 2], a2 = s_red[r * WPR * 2 + 1];
#pragma unroll
            for (int w = 1; w < WPR; ++w) {  // fixed order
                a1 += s_red[(r * WPR + w) * 2];
                a2 += s_red[(r * WPR + w) * 2 + 1];
            }
            const float mean = a1 / (float)p.F;
            const float rstd = rsqrtf(fmaxf(a2 / (float)p.F - mean * mean, 0.f) + 1e-6f);
            const int smp = r < S ? r : r - S;
            const bool row_ok = r < 2 * S && b0 + smp < p.B;
            const int64_t grow = r < S ? (int64_t)(b0 + smp) : (int64_t)p.B + b0 + smp;
#pragma unroll
            for (int j = 0; j < COLS; ++j) {
                const int c = tid + j * HC_THREADS;
                const float v = c < Fp ? s_pre[r * Fp + c] : 0.f;
                float y = v;
                if (p.gamma != nullptr && c < p.F) y = (v - mean) * (rstd * ga[j]) + be[j];
                y = fmaxf(y, 0.f);  // (the head chain requires a ReLU hidden layer)
                if (c >= p.F) y = 0.f;
                if (row_ok && c < Fp) {
                    s8_store_elem(p.act + grow * Fp, c, y);  // hidden activations: S8 (read back by the head weight gradient)
                    if (r < S) p.z[grow * Fp + c] = v;
                }
                if (r < 2 * S && c < PA - 8) {
                    const float ys = row_ok ? y : 0.f;
                    const __bf16 h = (__bf16)ys;
                    a_hi[r * PA + c] = h;
                    if constexpr (PASSES >= 2) a_lo[r * PA + c] = (__bf16)(ys - (float)h);
                }
                if (r < SMAX) zv[r < SMAX ? r : 0][j] = (row_ok && r < S && c < p.F) ? v : 0.f;
            }
        }
    }
    for (int i = tid; i < SMAX * Op; i += HC_THREADS) s_dq[i] = 0.f;
    if (tid < Op) s_bias[tid] = bias_v;
    for (int o = tid + HC_THREADS; o < Op; o += HC_THREADS) s_bias[o] = o < p.O ? p.bias[o] : 0.f;  // (more outputs than threads)
    __syncthreads();  // hidden rows staged
    HC_STAMP(1);  // hidden rows staged

    // ---- q = a W^T : wave w takes K-steps w, w+16, ... of every pair of 16-column tiles.  A work item is (tile pair,
    //      K-step group): its W fragments (two tiles x four K-steps) are one batch of loads, and the batch of the NEXT
    //      item is in flight while this one is multiplied ----
    {
        f32x4 acc[2];
        auto run_item = [&](float (&v)[2][HC_KU][8], int item) {
            const int pair = item / nks, ksi = item - pair * nks, ksb = wave + 16 * ksi;
            if (ksi == 0) {
                acc[0] = acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < HC_KU; ++u) {
                const int ks = min(ksb + HC_WAVES * u, nkt - 1);  // past the end: B is zero, A only has to be finite
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(a_hi + arow + ks * 32);
                bf16x8 al;
                if constexpr (PASSES >= 2) al = *reinterpret_cast<const bf16x8*>(a_lo + arow + ks * 32);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    bf16x8 bh, bl;
                    if constexpr (PASSES >= 2) {
                        split8(v[t][u], bh, bl);
                        if constexpr (PASSES >= 3)
                            mfma_acc(acc[t], ah, bl);
                        mfma_acc(acc[t], al, bh);
                    } else {
                        round8(v[t][u], bh);
                    }
                    mfma_acc(acc[t], ah, bh);
                }
            }
            const bool last = ksi + 1 == nks;
            if (last) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int o = (pair * 2 + t) * 16 + (lane & 15);
                    if (o < Op) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) qpart[(wave * 16 + (lane >> 4) * 4 + r) * Op + o] = acc[t][r];
                    }
                }
            }
        };
        for (int item = 0; item < n_items; item += 2) {
            issue_w(wv1, item + 1);  // (past the last item: zero block)
            run_item(wv0, item);
            issue_w(wv0, item + 2);
            if (item + 1 < n_items) run_item(wv1, item + 1);
        }
    }

    // ---- data-gradient rows: head row (1 + k) * A + action of every (transition, head), four heads per batch; the
    //      first batch is requested here and travels under the target / TD phase ----
    auto issue_rows = [&](float (&w)[SMAX][4][COLS], int k0) {
#pragma unroll
        for (int s = 0; s < SMAX; ++s)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool on = s < S && k0 + u < K;
                const float* wr = p.W + (int64_t)(on ? (p.oh + k0 + u) * A + act[s] : 0) * Fp;
#pragma unroll
                for (int j = 0; j < COLS; ++j) {
                    const int c = tid + j * HC_THREADS;
                    ISDQN_BOUNDS_CHECK(wr + (c < p.F ? c : 0), 4, 27);
                    w[s][u][j] = *(const ISDQN_GLOBAL float*)(wr + (c < p.F ? c : 0));
                }
            }
    };
    HC_STAMP(2);  // head GEMM done (this wave)
    float wr0[SMAX][4][COLS], wr1[SMAX][4][COLS];
    issue_rows(wr0, 0);
    __syncthreads();
    for (int i = tid; i < 2 * S * Op; i += HC_THREADS) {  // (tile rows 2S..15 are nobody's)
        const int o = i % Op;
        float v = qpart[i];
#pragma unroll
        for (int w = 1; w < HC_WAVES; ++w) v += qpart[w * 16 * Op + i];  // fixed order
        s_q[i] = o < p.O ? v + s_bias[o] : 0.f;
    }
    __syncthreads();

    // ---- iterated Bellman target, TD, dL/dq (isdqn.py:92-109) ----
    const float inv_b = 1.f / (float)p.B;
    if (tid < S * K) {
        const int s = tid / K, k = tid - s * K;
        const int b = b0 + s;
        float d = 0.f, td = 0.f;
        int a = 0;
        if (b < p.B) {
            a = act[0];
#pragma unroll
            for (int i = 1; i < SMAX; ++i) a = s == i ? act[i] : a;
            const float qv = s_q[s * Op + (p.oh + k) * A + a];
            const float* nq = s_q + (S + s) * Op + k * A;
            float mx = nq[0];
            for (int j = 1; j < A; ++j) mx = fmaxf(mx, nq[j]);
            const float tg = td_r + (1.f - (float)td_term) * p.gamma_n * mx;
            d = qv - tg;
            td = td_loss(d, p.huber_delta);
            if (p.q_values) p.q_values[(int64_t)b * K + k] = qv;
            if (p.targets) p.targets[(int64_t)b * K + k] = tg;
        }
        const float dd = td_dloss(d, p.huber_delta) * inv_b;
        s_d[tid] = dd;
        s_td[tid] = td;
        s_dq[s * Op + (p.oh + k) * A + a] = dd;
    }
    __syncthreads();
    for (int i = tid; i < S * Op; i += HC_THREADS) {
        const int s = i / Op;
        if (b0 + s < p.B) p.dout[(int64_t)b0 * Op + i] = s_dq[i];
    }
    for (int c = tid; c < Op; c += HC_THREADS) {
        float sum = 0.f;
        for (int s = 0; s < S; ++s) sum += s_dq[s * Op + c];
        p.dbh_part[(int64_t)blockIdx.x * Op + c] = sum;
    }
    if (tid < K) {
        float sum = 0.f;
        for (int s = 0; s < S; ++s) sum += s_td[s * K + tid];
        p.loss_part[(int64_t)blockIdx.x * K + tid] = sum;
    }
    if (p.priorities != nullptr && tid < S && b0 + tid < p.B) {
        float sum = 0.f;
        for (int k = 0; k < K; ++k) sum += s_td[tid * K + k];
        p.priorities[b0 + tid] = sqrt((double)(sum / (float)K) + 1e-10);
    }

    HC_STAMP(3);  // targets / TD / small stores issued
    // ---- da = dL/dq W (K row-AXPYs per transition, heads in ascending order), then the LayerNorm/ReLU backward of
    //      the hidden row ----
    float da[SMAX][COLS];
#pragma unroll
    for (int s = 0; s < SMAX; ++s)
#pragma unroll
        for (int j = 0; j < COLS; ++j) da[s][j] = 0.f;
    auto axpy_rows = [&](const float (&w)[SMAX][4][COLS], int k0) {
#pragma unroll
        for (int s = 0; s < SMAX; ++s)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool on = s < S && k0 + u < K;
                const float dd = on ? s_d[s * K + k0 + u] : 0.f;
#pragma unroll
                for (int j = 0; j < COLS; ++j) da[s][j] += dd * w[s][u][j];
            }
    };
    for (int k0 = 0; k0 < K; k0 += 8) {  // 4 heads x S transitions x COLS columns of W rows per batch, two batches in flight
        issue_rows(wr1, k0 + 4);
        axpy_rows(wr0, k0);
        issue_rows(wr0, k0 + 8);
        if (k0 + 4 < K) axpy_rows(wr1, k0 + 4);
    }
    HC_STAMP(4);  // data-gradient rows accumulated
    float dg[COLS], db[COLS], dbias[COLS];
#pragma unroll
    for (int j = 0; j < COLS; ++j) dg[j] = db[j] = dbias[j] = 0.f;
    if (p.gamma != nullptr) {
        const float inv_c = 1.f / (float)p.F;
        float r1[SMAX], r2[SMAX];
        auto block_sum2 = [&]() {  // r1[s], r2[s] summed over the workgroup
#pragma unroll
            for (int s = 0; s < SMAX; ++s)
                for (int off = 32; off > 0; off >>= 1) {
                    r1[s] += __shfl_xor(r1[s], off);
                    r2[s] += __shfl_xor(r2[s], off);
                }
            __syncthreads();
            if (lane == 0) {
#pragma unroll
                for (int s = 0; s < SMAX; ++s) {
                    s_red[(wave * SMAX + s) * 2] = r1[s];
                    s_red[(wave * SMAX + s) * 2 + 1] = r2[s];
                }
            }
            __syncthreads();
#pragma unroll
            for (int s = 0; s < SMAX; ++s) {
                r1[s] = s_red[s * 2];
                r2[s] = s_red[s * 2 + 1];
#pragma unroll
                for (int w = 1; w < HC_WAVES; ++w) {  // fixed order
                    r1[s] += s_red[(w * SMAX + s) * 2];
                    r2[s] += s_red[(w * SMAX + s) * 2 + 1];
                }
            }
        };
#pragma unroll
        for (int s = 0; s < SMAX; ++s) {
            r1[s] = r2[s] = 0.f;
#pragma unroll
            for (int j = 0; j < COLS; ++j) {  // zv is zero outside the row / the transition range
                r1[s] += zv[s][j];
                r2[s] += zv[s][j] * zv[s][j];
            }
        }
        block_sum2();
        float mean[SMAX], rstd[SMAX];
#pragma unroll
        for (int s = 0; s < SMAX; ++s) {
            mean[s] = r1[s] * inv_c;
            rstd[s] = rsqrtf(fmaxf(r2[s] * inv_c - mean[s] * mean[s], 0.f) + 1e-6f);
            r1[s] = r2[s] = 0.f;
#pragma unroll
            for (int j = 0; j < COLS; ++j) {
                const int c = tid + j * HC_THREADS;
                const bool ok = c < p.F && s < S && b0 + s < p.B;
                const float xh = (zv[s][j] - mean[s]) * rstd[s];
                const float y = xh * ga[j] + be[j];
                const float dy = (ok && y > 0.f) ? da[s][j] : 0.f;
                dg[j] += dy * xh;
                db[j] += dy;
                const float g2 = dy * ga[j];
                da[s][j] = g2;   // keep g
                zv[s][j] = ok ? xh : 0.f;  // keep xhat
                r1[s] += g2;
                r2[s] += g2 * zv[s][j];
            }
        }
        block_sum2();
#pragma unroll
        for (int s = 0; s < SMAX; ++s) {
            const float m1 = r1[s] * inv_c, m2 = r2[s] * inv_c;
            if (s < S && b0 + s < p.B) {
#pragma unroll
                for (int j = 0; j < COLS; ++j) {
                    const int c = tid + j * HC_THREADS;
                    if (c < Fp) {
                        const float o = c < p.F ? rstd[s] * (da[s][j] - m1 - zv[s][j] * m2) : 0.f;
                        dbias[j] += o;
                        s8_store_elem(p.dz + (int64_t)(b0 + s) * Fp, c, o);  // dz: S8
                    }
                }
            }
        }
    } else {
#pragma unroll
        for (int s = 0; s < SMAX; ++s) {
            if (s < S && b0 + s < p.B) {
#pragma unroll
                for (int j = 0; j < COLS; ++j) {
                    const int c = tid + j * HC_THREADS;
                    if (c < Fp) {
                        const float o = (c < p.F && zv[s][j] > 0.f) ? da[s][j] : 0.f;
                        dbias[j] += o;
                        s8_store_elem(p.dz + (int64_t)(b0 + s) * Fp, c, o);  // dz: S8
                    }
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < COLS; ++j) {
        const int c = tid + j * HC_THREADS;
        if (c < Fp) {
            float* pp = p.part + (int64_t)blockIdx.x * 3 * Fp;
            pp[c] = dg[j];
            pp[Fp + c] = db[j];
            pp[2 * Fp + c] = dbias[j];
        }
    }
    HC_STAMP(5);  // LayerNorm backward done, stores issued
#undef HC_STAMP
}

// losses[k] = mean_b td (isdqn.py:103), optional running sum (update_online_params' cumulated_losses,
// isdqn.py:62, kept on the device), head-bias gradient, Adam step counter and bias corrections.
__global__ __launch_bounds__(256) void loss_finalize_kernel(const float* __restrict__ loss_part,
                                                            const float* __restrict__ dbh_part, int n_blk, int B, int K,
                                                            int nha_p, float* __restrict__ losses,
                                                            float* __restrict__ loss_accum, float* __restrict__ dbh,
                                                            int* adam_count, float b1, float b2,
                                                            float* __restrict__ adam_consts) {
    ISDQN_EMPTY_KERNEL_RETURN
    const int tid = threadIdx.x, sub = tid & 15, grp = tid >> 4;
    // One workgroup per 16 outputs (blocks [0, ceil(K/16)): losses, the rest: head-bias columns).  16 lanes share one
    // output: strided partial sums (eight loads in flight, added in index order), then a fixed shuffle tree.
    const int nkb = (K + 15) / 16;
    const bool is_loss = (int)blockIdx.x < nkb;
    const int col = (is_loss ? (int)blockIdx.x : (int)blockIdx.x - nkb) * 16 + grp;
    const int ncol = is_loss ? K : nha_p;
    const float* src = is_loss ? loss_part : dbh_part;
    if (!is_loss && dbh == nullptr) return;
    float s = 0.f;
    if (col < ncol)
        for (int i0 = sub; i0 < n_blk; i0 += 8 * 16) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * 16;
                v[u] = i < n_blk ? src[(int64_t)i * ncol + col] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
    for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (col < ncol && sub == 0) {
        if (is_loss) {
            s /= (float)B;
            losses[col] = s;
            if (loss_accum != nullptr) loss_accum[col] += s;
        } else {
            dbh[col] = s;
        }
    }
    if (blockIdx.x != 0) return;
    if (adam_count != nullptr && tid == 0) {
        int t = *adam_count + 1;
        *adam_count = t;
        adam_consts[0] = (float)(1.0 - pow_int((double)b1, t));
        adam_consts[1] = (float)(1.0 - pow_int((double)b2, t));
    }
}

// Adam (optax.adam, isdqn.py:46, 85-86) over the flat parameter buffer; the gradient of every
// tensor is the sum of its slabs (split-K partials / LayerNorm-backward partials), so gradients
// are reduced here and never stored in reduced form.
struct AdamEntry {
    int64_t p_off, size, slab_stride;
    const float* g;
    int n_slabs, block_start;
};
constexpr int ADAM_MAX_ENTRIES = 4 * MAX_LAYERS;
struct AdamTable {
    AdamEntry e[ADAM_MAX_ENTRIES];
    int n, total_blocks;
};
__global__ __launch_bounds__(256) void adam_kernel(const AdamTable tab, float* __restrict__ p, float* __restrict__ m,
                                                   float* __restrict__ v, const float* __restrict__ consts, float lr,
                                                   float b1, float b2, float eps, float* __restrict__ grad_out,
                                                   float* __restrict__ mirror, int update) {
    ISDQN_EMPTY_KERNEL_RETURN
    // A workgroup covers 16 float4 positions; 16 "slab lanes" per position split the slab reduction (up to a
    // few hundred split-K / per-image-group slabs for the conv kernels) and combine through LDS in a fixed
    // order, so the reduction is deterministic and never a long serial chain of dependent loads.
    __shared__ float4 s_g[16][16];
    int e = 0;
    while (e + 1 < tab.n && (int)blockIdx.x >= tab.e[e + 1].block_start) ++e;
    const AdamEntry en = tab.e[e];
    const float c1 = consts[0], c2 = consts[1];  // 1 - b1^t, 1 - b2^t (loss_finalize_kernel)
    const int pos = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int64_t i = ((int64_t)(blockIdx.x - en.block_start) * 16 + pos) * 4;
    const bool on = i < en.size;
    float4 g = float4{0.f, 0.f, 0.f, 0.f};
    // the updating lanes request their moments and parameters now: they arrive under the slab reduction
    const int64_t o = en.p_off + i;
    float4 pm = g, pv = g, pp = g;
    if (on && sl == 0) {
        pm = *reinterpret_cast<const float4*>(m + o);
        pv = *reinterpret_cast<const float4*>(v + o);
        pp = *reinterpret_cast<const float4*>(p + o);
    }
    if (on) {
        int s = sl;
        // (eight slabs in flight per thread first: the first layer's weight gradient leaves one slab per image at B = 256 -- 16 dependent
        // loads per thread in batches of four were four round trips at the tail of the side queue)
#if !defined(ISDQN_ADAM_4_IN_FLIGHT)
        for (; s + 7 * 16 < en.n_slabs; s += 8 * 16) {
            float4 h[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) h[u] = *reinterpret_cast<const float4*>(en.g + (int64_t)(s + u * 16) * en.slab_stride + i);
#pragma unroll
            for (int u = 0; u < 8; ++u) { g.x += h[u].x; g.y += h[u].y; g.z += h[u].z; g.w += h[u].w; }
        }
#endif
        for (; s + 3 * 16 < en.n_slabs; s += 4 * 16) {
            float4 h[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) h[u] = *reinterpret_cast<const float4*>(en.g + (int64_t)(s + u * 16) * en.slab_stride + i);
#pragma unroll
            for (int u = 0; u < 4; ++u) { g.x += h[u].x; g.y += h[u].y; g.z += h[u].z; g.w += h[u].w; }
        }
        for (; s < en.n_slabs; s += 16) {
            float4 h = *reinterpret_cast<const float4*>(en.g + (int64_t)s * en.slab_stride + i);
            g.x += h.x; g.y += h.y; g.z += h.z; g.w += h.w;
        }
    }
    s_g[sl][pos] = g;
    __syncthreads();
    if (sl != 0 || !on) return;
    g = s_g[0][pos];
#pragma unroll
    for (int k = 1; k < 16; ++k) {
        const float4 h = s_g[k][pos];
        g.x += h.x; g.y += h.y; g.z += h.z; g.w += h.w;
    }
    if (grad_out != nullptr) *reinterpret_cast<float4*>(grad_out + o) = g;
    if (!update) return;  // gradient only (isdqn_net_grad_on_batch)
    float* gp = &g.x; float* mp = &pm.x; float* vp = &pv.x; float* xp = &pp.x;
    const float inv_c1 = 1.f / c1, inv_c2 = 1.f / c2;
#pragma unroll
    for (int r = 0; r < 4; ++r) xp[r] = adam_element(mp[r], vp[r], xp[r], gp[r], b1, b2, lr, eps, inv_c1, inv_c2);
    *reinterpret_cast<float4*>(m + o) = pm;
    *reinterpret_cast<float4*>(v + o) = pv;
    *reinterpret_cast<float4*>(p + o) = pp;
    s8_store_quad(mirror, (int)o, pp.x, pp.y, pp.z, pp.w);  // S8 mirror of the updated parameters (read by the next step's MFMA stages)
}

// shift_params (isdqn.py:111-125): rows [0, nha-A) <- rows [A, nha) of the last Dense ([out][in] layout).
__global__ __launch_bounds__(256) void shift_kernel(float* __restrict__ w, float* __restrict__ bias, int nha, int A,
                                                    int in_p) {
    int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f < in_p)
        for (int r = 0; r + A < nha; ++r) w[(int64_t)r * in_p + f] = w[(int64_t)(r + A) * in_p + f];
    if (f == in_p)
        for (int r = 0; r + A < nha; ++r) bias[r] = bias[r + A];
}

// best_action (isdqn.py:127-135): first argmax over the actions of head 1+idx.
__global__ void argmax_kernel(const float* __restrict__ q, int A, int head, int* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const float* r = q + head * A;
        int best = 0;
        for (int a = 1; a < A; ++a)
            if (r[a] > r[best]) best = a;
        *out = best;
    }
}

// AnalysisNet (slimdqn/utils/analysis_architecture.py:46-122): per-neuron sums over the rows of one hidden layer's post-ReLU
// activations (S8 storage, [n_rows][pitch], `cpp` padded channels per pixel of which `c` are real), optionally also the
// activations themselves as fp32 [n_rows][feat_ld] in the reference's (unpadded) feature order.  A workgroup owns 32 groups
// of 8 elements; 8 row lanes stride over the rows and combine through LDS in a fixed order (deterministic).
__global__ __launch_bounds__(256) void act_rowsum_kernel(const float* __restrict__ act, int n_rows, int pitch, int cpp, int c,
                                                         float* __restrict__ scores, float* __restrict__ feat, int feat_ld) {
    __shared__ float s_sum[8][32][8];
    const int gl = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int e0 = ((int)blockIdx.x * 32 + gl) * 8;
    const bool on = e0 < pitch;
    const int pix = on ? e0 / cpp : 0, ch0 = on ? e0 % cpp : 0;
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    if (on)
        for (int r = rl; r < n_rows; r += 8) {
            const float* src = act + (int64_t)r * pitch + e0;
            const f32x4 h = *reinterpret_cast<const f32x4*>(src), l = *reinterpret_cast<const f32x4*>(src + 4);
            const bf16x8 hi = __builtin_bit_cast(bf16x8, h), lo = __builtin_bit_cast(bf16x8, l);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float v = (float)hi[i] + (float)lo[i];
                acc[i] += v;
                if (feat != nullptr && ch0 + i < c) feat[(int64_t)r * feat_ld + pix * c + ch0 + i] = v;
            }
        }
#pragma unroll
    for (int i = 0; i < 8; ++i) s_sum[rl][gl][i] = acc[i];
    __syncthreads();
    if (rl == 0 && on) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) t += s_sum[k][gl][i];
            if (ch0 + i < c) scores[pix * c + ch0 + i] = t;
        }
    }
}

// batched form: row i -> first argmax of head oh + idx[i] of its own q row
__global__ __launch_bounds__(64) void argmax_rows_kernel(const float* __restrict__ q, int n_rows, int nha_p, int A, int oh, int K,
                                                         const int* __restrict__ idx, int* __restrict__ out) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n_rows) return;
    int head = idx[i];
    head = oh + (head < 0 ? 0 : head >= K ? K - 1 : head);
    const float* r = q + (int64_t)i * nha_p + head * A;
    int best = 0;
    for (int a = 1; a < A; ++a)
        if (r[a] > r[best]) best = a;
    out[i] = best;
}

// =============================================================================================
// Host orchestration
// =============================================================================================
// profiling hook (not in the public header): phase stamps of the image-resident forward kernel of one layer
#if defined(ISDQN_DEV)
static long long* g_stamps = nullptr;
static int g_stamp_layer = -1;
static char g_stamp_name[32] = "";
extern "C" int isdqn_debug_set_stamps(void* buf, const char* layer_name) {
    g_stamps = (long long*)buf;
    g_stamp_layer = buf ? 0 : -1;
    snprintf(g_stamp_name, sizeof(g_stamp_name), "%s", layer_name ? layer_name : "");
    return ISDQN_OK;
}
static long long* stamps_for(const char* kernel_tag) {
    return (g_stamp_layer >= 0 && strcmp(kernel_tag, g_stamp_name) == 0) ? g_stamps : nullptr;
}
#else
static long long* stamps_for(const char*) { return nullptr; }
#endif

// fp32 master parameters -> S8 mirror (gemm_core.h): every 32-byte group of 8 values becomes 8 hi + 8 lo bf16.  The
// MFMA consumers of the weights stage the mirror by copy.  Run at the head of every entry point that takes `params`:
// the master is caller-owned memory (imports, head shifts, target copies happen outside this library).
__global__ __launch_bounds__(256) void split_params_kernel(const float* __restrict__ p, float* __restrict__ mirror, int64_t n_groups) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= n_groups) return;
    float v[8];
    load8_aligned(p + g * 8, v);
    s8_store_group(mirror + g * 8, v);
}
static int refresh_mirror(const Plan& P, const float* params, float* ws, hipStream_t st) {
    const int64_t n_groups = P.n_params / 8;  // every tensor size is a multiple of 8 (padded widths)
    hipLaunchKernelGGL(split_params_kernel, dim3((unsigned)((n_groups + 255) / 256)), dim3(256), 0, st, params, ws + P.wsplit_off, n_groups);
    ISDQN_HIP_CHECK(hipGetLastError());
    return ISDQN_OK;
}

static ConvGeom conv_geom(const Layer& l) {
    ConvGeom g;
    g.hin = l.hin; g.win = l.win; g.cin_p = l.cin_p; g.hout = l.hout; g.wout = l.wout;
    g.cout = l.cout; g.cout_p = l.cout_p; g.ksz = l.ksz; g.stride = l.stride; g.pad = l.pad;
    g.npix = l.npix; g.K = l.K;
    g.stride_sh = l.stride == 4 ? 2 : l.stride == 2 ? 1 : 0;
    g.d_npix = FastDiv(l.npix); g.d_wout = FastDiv(l.wout); g.d_cinp = FastDiv(l.cin_p);
    g.d_ksz = FastDiv(l.ksz); g.d_coutp = FastDiv(l.cout_p);
    return g;
}

// Storage of the pre-activation gradients dz of hidden layers (bit 0 of the S8 operand masks below): S8, written so
// by every producer (head chain, LayerNorm-backward epilogues and kernels); the head's dL/dq stays fp32
constexpr int DZ_S8 = 1;

struct NetInput {
    const uint8_t* frames; int64_t frame_stride; const int* frame_ids; int paired_B;  // cnn
    const float* obs; const float* obs2; int obs_split;                             // fc
    int id_pitch = 0, id_off = 0;                                                   // cnn, unpaired: see FrameSrc
};

template <int BM, int PASSES, bool U8>
static int launch_conv_fwd(const Layer& l, const float* params, const float* wmir, const NetInput& in, const float* act_in, int n_img,
                           int z_img, float* act, float* z, hipStream_t st) {
    ConvFwd<BM, PASSES, U8> p;
    p.g = conv_geom(l);
    p.W = MatSrc{wmir + l.w_off, l.K, l.cout_p, l.K, 1};
    p.in = act_in;
    p.fs = FrameSrc{in.frames, in.frame_stride, in.frame_ids, l.cin, in.paired_B, l.hin, l.win, in.id_pitch, in.id_off};
    p.bias = params + l.b_off;
    p.gamma = l.has_ln ? params + l.g_off : nullptr;
    p.beta = l.has_ln ? params + l.be_off : nullptr;
    p.scale = U8 ? (1.0f / 255.0f) : 1.0f;
    p.act = act; p.z = z;
    p.n_pix_total = n_img * l.npix;
    p.z_pix = z_img * l.npix;
    return launch_gemm(p, ceil_div(p.n_pix_total, 128), st);
}

// image-resident forward convolution (conv_img.h) when the input tile fits in LDS; generic engine otherwise
static int conv_fwd_img(const Layer& l, bool x3, const float* params, const float* wmir, const NetInput& in, const float* act_in, int n_img,
                        int z_img, float* act, float* z, hipStream_t st, bool* done) {
    *done = false;
    // The image-resident kernels store `act` unconditionally (only the `z` store is guarded, by z_img): a caller without an activation
    // output -- the impala Stacks pass act == nullptr for the convolutions whose output is only needed pre-activation -- must take the
    // generic engine.  (Round 3's uncommitted experiment that routed the impala convolutions here faulted on exactly that store:
    // "Memory access fault ... on address 0x9000" = null + the first image tile's offset, gpurun_out/imp1.log)
    if (act == nullptr || (z == nullptr && z_img > 0)) return ISDQN_OK;
    ConvImgParams ip;
    ip.g = conv_geom(l);
    // (two channel tiles: the kernel then alternates two accumulator sets, see conv_fwd_img_kernel)
    const int mt = l.cout_p <= 32 ? 2 : 4;
    const int passes = x3 ? (l.is_u8 ? 2 : 3) : 1;
    // pixel pitch: +8 / +16 elements so that the 16 pixels of an MFMA column tile do not share LDS banks
    // (128-byte pixel rows put them 4-5 deep on the same banks; measured model in DESIGN.md)
    ip.PP = l.is_u8 ? 0 : l.cin_p + (l.cin_p % 64 == 0 ? 16 : 8);
    int lds = conv_img_geometry(ip.g, l.is_u8, l.cin, passes >= 3 ? 2 : 1, mt, passes >= 2 ? 2 : 1, ip.R, ip.Wp,
                                ip.plane_elems, ip.PP);
    constexpr int LDS_LIMIT = 158 * 1024;  // of the 160 KB, minus the kernels' static arrays
    if (lds > LDS_LIMIT || (l.is_u8 && (l.win < 8 || l.cin > 4))) return ISDQN_OK;  // (frame ids of a stack live in 4 registers)
    if (!l.is_u8 && l.cin_p < 16) return ISDQN_OK;  // (8-channel S8 inputs -- the BatchNorm networks' first convolution -- take the generic engine)
    // A row pitch of wout (mod 8) pixels keeps the bank pattern of a fragment's pixel columns going across the end of an
    // image row (conv_img.h, PixelOrder: the short last strip of every row is where the conflicts are left); taken when it
    // costs no workgroup per CU (11-pixel rows of the stride-2 layer: 24 -> 27 columns, conflict factor 1.75 -> 1.0).
    if (!l.is_u8 && l.npix <= 128) {
        const int want = ip.Wp + (((l.wout - ip.Wp) % 8) + 8) % 8;
        const int planes = passes >= 3 ? 2 : 1;
        const int lds2 = lds + planes * ip.R * (want - ip.Wp) * ip.PP * 2;
        const int stage2 = (passes >= 2 ? 2 : 1) * (mt * 16) * 48 * 2 * 2;  // the second K group's stages (see kg2 below)
        const int old_total = lds > 80 * 1024 ? lds + stage2 : lds, new_total = lds2 > 80 * 1024 ? lds2 + stage2 : lds2;
        if (want != ip.Wp && new_total <= LDS_LIMIT && (old_total <= 80 * 1024) == (new_total <= 80 * 1024)) {
            ip.Wp = want;
            ip.plane_elems = ip.R * want * ip.PP;
            lds = lds2;
        }
    }
    ip.W = MatSrc{wmir + l.w_off, l.K, l.cout_p, l.K, 1};
    ip.in = act_in;
    ip.fs = FrameSrc{in.frames, in.frame_stride, in.frame_ids, l.cin, in.paired_B, l.hin, l.win, in.id_pitch, in.id_off};
    ip.bias = params + l.b_off;
    ip.gamma = l.has_ln ? params + l.g_off : nullptr;
    ip.beta = l.has_ln ? params + l.be_off : nullptr;
    ip.scale = l.is_u8 ? (1.0f / 255.0f) : 1.0f;
    ip.act = act; ip.z = z; ip.n_img = n_img; ip.z_img = z_img;
    ip.tiles_per_img = ceil_div(l.npix, 128);
    ip.d_chunk = FastDiv((uint32_t)(l.is_u8 ? ip.Wp / 8 : l.cin_p / 8));
    ip.d_Wp = FastDiv((uint32_t)ip.Wp);
    ip.d_R = FastDiv((uint32_t)ip.R);
    ip.order = PixelOrder(l.hout, l.wout, !l.is_u8 && ip.tiles_per_img == 1);
    ip.stamps = stamps_for(l.name);
    ip.ablate = 0;
#if defined(ISDQN_DEV)
    if (const char* e = getenv("ISDQN_ABLATE")) ip.ablate = atoi(e);
#endif
    *done = true;
    if (l.is_u8) {
#if !defined(ISDQN_NO_U8_PAIR)
        // two pixel tiles per workgroup, the second one's frame rows prefetched into registers (conv_u8_pair.h): the tile image
        // must be one batch of 2 x 256 positions, the frame ids four at most
        // (four stacked frames: K = 256, eight K steps as straight-line code)
        if (mt == 2 && ip.tiles_per_img % 2 == 0 && ip.R * (ip.Wp / 8) <= U8P_PB * GEMM_THREADS && l.cin == 4 && l.K == 256 && ip.ablate == 0)
            return passes == 2 ? launch_conv_fwd_u8_pair<2, 8>(ip, st) : launch_conv_fwd_u8_pair<1, 8>(ip, st);
#endif
        if (passes == 2) return mt == 2 ? launch_conv_fwd_img<2, 2, true>(ip, st) : launch_conv_fwd_img<4, 2, true>(ip, st);
        return mt == 2 ? launch_conv_fwd_img<2, 1, true>(ip, st) : launch_conv_fwd_img<4, 1, true>(ip, st);
    }
    // An image that leaves room for one workgroup per CU only (more than half of the 160 KB) runs with two K groups of
    // four waves, if the second pair of weight stages still fits and the accumulator exchange fits the image area.
    const int stage_bytes = (passes >= 2 ? 2 : 1) * (mt * 16) * 48 * 2 * 2;  // two stages of one K group
    static const bool no_kg = ISDQN_DEV_ENV("ISDQN_NO_KGROUPS");
    const bool kg2 = !no_kg && mt == 4 && lds > 80 * 1024 && lds + stage_bytes <= LDS_LIMIT &&
                     mt * 16 * 128 * 4 <= lds - stage_bytes && l.K >= 8 * GEMM_BK;
#if !defined(ISDQN_NO_U8_PAIR)
    // two images per workgroup, the second one prefetched into registers (conv_s8_pair.h): the one-workgroup-per-CU layer with 16 K steps
    if (passes == 3 && mt == 4 && kg2 && ip.tiles_per_img == 1 && l.K == 512 && n_img % 2 == 0 && ip.ablate == 0 && ip.stamps == nullptr &&
        ip.R * ip.Wp * (l.cin_p / 8) <= S8P_BATCH * GEMM_THREADS * 2)
        return launch_conv_fwd_s8_pair<8>(ip, st);
#endif
    if (passes == 3) return mt == 2 ? launch_conv_fwd_img<2, 3, false>(ip, st)
                          : kg2 ? launch_conv_fwd_img<4, 3, false, 2>(ip, st) : launch_conv_fwd_img<4, 3, false>(ip, st);
    return mt == 2 ? launch_conv_fwd_img<2, 1, false>(ip, st)
         : kg2 ? launch_conv_fwd_img<4, 1, false, 2>(ip, st) : launch_conv_fwd_img<4, 1, false>(ip, st);
}

static int conv_fwd(const Layer& l, bool x3, const float* params, const float* wmir, const NetInput& in, const float* act_in, int n_img,
                    int z_img, float* act, float* z, hipStream_t st) {
    bool done = false;
    int rc = conv_fwd_img(l, x3, params, wmir, in, act_in, n_img, z_img, act, z, st, &done);
    if (rc || done) return rc;
    const bool small = l.cout_p <= 32;
    if (l.is_u8) {
        if (x3) return small ? launch_conv_fwd<32, 2, true>(l, params, wmir, in, act_in, n_img, z_img, act, z, st)
                             : launch_conv_fwd<64, 2, true>(l, params, wmir, in, act_in, n_img, z_img, act, z, st);
        return small ? launch_conv_fwd<32, 1, true>(l, params, wmir, in, act_in, n_img, z_img, act, z, st)
                     : launch_conv_fwd<64, 1, true>(l, params, wmir, in, act_in, n_img, z_img, act, z, st);
    }
    if (x3) return small ? launch_conv_fwd<32, 3, false>(l, params, wmir, in, act_in, n_img, z_img, act, z, st)
                         : launch_conv_fwd<64, 3, false>(l, params, wmir, in, act_in, n_img, z_img, act, z, st);
    return small ? launch_conv_fwd<32, 1, false>(l, params, wmir, in, act_in, n_img, z_img, act, z, st)
                 : launch_conv_fwd<64, 1, false>(l, params, wmir, in, act_in, n_img, z_img, act, z, st);
}

// C[split][M][N] = A . B^T over row-major sources.  a_tr/b_tr: the operand is stored [K][rows].
template <int BM, int BN, int WM, int WN, bool ATR, bool BTR, int PASSES, bool AL, bool A2PART, bool ADAM = false, int S8M = 0>
static int launch_plain(const MatSrc& A, const float* A2, int a_split, const MatSrc& B, float* C, int ldc, int M,
                        int N, int K, int splits, int64_t slab_stride, hipStream_t st,
                        const AdamFuse* adam = nullptr) {
    PlainGemm<BM, BN, WM, WN, ATR, BTR, PASSES, AL, A2PART, ADAM, S8M> p;
    if (adam) p.adam = *adam;
    p.A = A; p.B = B; p.A2 = A2; p.a_split = a_split; p.C = C; p.ldc = ldc; p.M = M; p.N = N; p.K = K;
    p.tiles_m = ceil_div(M, BM); p.tiles_n = ceil_div(N, BN);
    int ksteps = ceil_div(K, GEMM_BK);
    if (splits < 1) splits = 1;
    if (splits > ksteps) splits = ksteps;
    p.steps_per_split = ceil_div(ksteps, splits);
    p.splits = ceil_div(ksteps, p.steps_per_split);
    p.slab_stride = slab_stride;
    return launch_gemm(p, p.tiles_m * p.tiles_n * p.splits, st);
}
// number of slabs launch_plain will actually write for (K, splits)
static int effective_splits(int K, int splits) {
    int ksteps = ceil_div(K, GEMM_BK);
    if (splits < 1) splits = 1;
    if (splits > ksteps) splits = ksteps;
    int sps = ceil_div(ksteps, splits);
    return ceil_div(ksteps, sps);
}

template <bool ATR, bool BTR, bool AL = true, bool A2PART = false, int S8M = 0>
static int plain_big(bool x3, const MatSrc& A, const float* A2, int a_split, const MatSrc& B, float* C, int ldc,
                     int M, int N, int K, int splits, int64_t slab_stride, hipStream_t st) {
    if (x3)
        return launch_plain<128, 128, 2, 2, ATR, BTR, 3, AL, A2PART, false, S8M>(A, A2, a_split, B, C, ldc, M, N, K, splits,
                                                                                 slab_stride, st);
    return launch_plain<128, 128, 2, 2, ATR, BTR, 1, AL, A2PART, false, S8M>(A, A2, a_split, B, C, ldc, M, N, K, splits,
                                                                             slab_stride, st);
}

// 128 x 64 tiles: twice the workgroups of plain_big for GEMMs whose 128 x 128 grid cannot fill 256 CUs
template <bool ATR, bool BTR, int S8M = 0>
static int plain_narrow(bool x3, const MatSrc& A, const MatSrc& B, float* C, int ldc, int M, int N, int K, int splits,
                        int64_t slab_stride, hipStream_t st) {
    if (x3) return launch_plain<128, 64, 2, 2, ATR, BTR, 3, true, false, false, S8M>(A, nullptr, 0, B, C, ldc, M, N, K, splits, slab_stride, st);
    return launch_plain<128, 64, 2, 2, ATR, BTR, 1, true, false, false, S8M>(A, nullptr, 0, B, C, ldc, M, N, K, splits, slab_stride, st);
}

// `skip_post`: leave the split-K slabs as they are (the head chain kernel finishes the layer for its own rows)
static int dense_fwd(const Layer& l, bool x3, const float* params, const float* wmir, const NetInput& in, const float* act_in, int rows,
                     int z_rows, float* slab, float* act, float* z, hipStream_t st, bool skip_post = false) {
    MatSrc A, B;
    const float* A2 = nullptr;
    int a_split = 0;
    if (l.in_unpadded_ld) {  // fc first layer: caller's [rows][obs] matrices
        A = MatSrc{in.obs, l.in_unpadded_ld, rows, l.in_f, 0};
        if (in.obs2) { A2 = in.obs2; a_split = in.obs_split; }
    } else {
        A = MatSrc{act_in, l.in_p, rows, l.in_p, 1};
    }
    B = MatSrc{wmir + l.w_off, l.in_p, l.out_f, l.in_p, 1};  // weights: S8 mirror (rows padded to in_p with zeros)
    // split-K partial products, row-interleaved: slab s of row m at slab[(m * ns + s) * out_p].  The ns partial rows of
    // one output row are one contiguous block (64 KB at the headline size) for whoever sums them; in [slab][row]
    // order they sit 1 MB apart, 32 pages per reader.
    const int ns = effective_splits(l.in_unpadded_ld ? l.in_f : l.K, l.fwd_splits);
    const int64_t slab_stride = l.out_p;
    const int ldc = ns * l.out_p;
    int rc;
    if (l.in_unpadded_ld) {
        // caller-provided observations: no alignment promise; MatSrc bounds use the true widths
        // (the observation operand bounds its own chunks to in_f; the mirror rows are read in whole aligned groups)
        rc = A2 ? plain_big<false, false, false, true, 2>(x3, A, A2, a_split, B, slab, ldc, rows, l.out_f, l.in_f,
                                                          l.fwd_splits, slab_stride, st)
                : plain_big<false, false, false, false, 2>(x3, A, nullptr, 0, B, slab, ldc, rows, l.out_f, l.in_f,
                                                           l.fwd_splits, slab_stride, st);
    } else if (l.fwd_narrow) {  // few row tiles: 128 x 64 tiles reach the workgroup count with half the split-K slabs
        rc = plain_narrow<false, false, 3>(x3, A, B, slab, ldc, rows, l.out_f, l.K, l.fwd_splits, slab_stride, st);
    } else {
        rc = plain_big<false, false, true, false, 3>(x3, A, nullptr, 0, B, slab, ldc, rows, l.out_f, l.K, l.fwd_splits, slab_stride,
                                                     st);  // activations and weights both S8
    }
    if (rc || skip_post) return rc;
    hipLaunchKernelGGL(dense_post_kernel, dim3(rows), dim3(256), 3 * l.out_p * sizeof(float), st, slab, ns, slab_stride,
                       (int64_t)ldc, rows, l.out_f, l.out_p, params + l.b_off, l.has_ln ? params + l.g_off : nullptr,
                       l.has_ln ? params + l.be_off : nullptr, l.has_relu, act, z, z_rows, l.is_head ? 0 : 1);
    ISDQN_HIP_CHECK(hipGetLastError());
    return ISDQN_OK;
}

static int impala_forward(const Plan& P, bool x3, const float* params, const float* wmir, const NetInput& in, int n_img, int z_img,
                          float* ws, hipStream_t st, int bn_mode = 0);  // impala.h

static int bn_forward(const Plan& P, bool x3, const float* params, const NetInput& in, int n_img, int z_img, float* ws, float* q_out, bool running,
                      hipStream_t st);  // batchnorm.h

// forward over n_img images; hidden activations -> ws act regions, head output -> q_out [n_img][nha_p]
static int net_forward(const Plan& P, bool x3, const float* params, const NetInput& in, int n_img, int z_img,
                       float* ws, float* q_out, hipStream_t st, int n_run = -1, int skip_post_layer = -1) {
    if (P.bn) {  // BatchNorm networks outside a learn step: the running averages (isdqn.py:130, use_running_average=True)
        ISDQN_REQUIRE(n_run < 0 && skip_post_layer < 0, ISDQN_ERR_UNSUPPORTED, "partial forward passes are not built for BatchNorm networks");
        return bn_forward(P, x3, params, in, n_img, z_img, ws, q_out, true, st);
    }
    const float* prev = nullptr;
    const float* wmir = ws + P.wsplit_off;  // refreshed by the caller (refresh_mirror) from `params`
    if (n_run < 0) n_run = P.n_layers;
    for (int i = 0; i < n_run; ++i) {
        const Layer& l = P.L[i];
        float* act = l.is_head ? q_out : ws + l.act_off;
        float* z = l.is_head ? nullptr : ws + l.z_off;
        int rc;
        if (l.kind == 2)
            rc = impala_forward(P, x3, params, wmir, in, n_img, z_img, ws, st);
        else if (l.kind == 0)
            rc = conv_fwd(l, x3, params, wmir, in, prev, n_img, z_img, act, z, st);
        else
            rc = dense_fwd(l, x3, params, wmir, in, prev, n_img, l.is_head ? 0 : z_img, ws + P.slab_off, act, z, st,
                           i == skip_post_layer);
        if (rc) return rc;
        prev = act;
    }
    return ISDQN_OK;
}

static int ln_bwd(const Layer& l, const float* params, const float* da, const float* z, int rows, float* dz,
                  float* part, int* n_blocks_out, hipStream_t st) {
    const float* gamma = l.has_ln ? params + l.g_off : nullptr;
    const float* beta = l.has_ln ? params + l.be_off : nullptr;
    const int C = l.out_f, Cp = l.out_p;
    int nb;
#define LAUNCH_SMALL(TPR)                                                                                          \
    do {                                                                                                           \
        nb = ceil_div(rows, 256 / TPR);                                                                            \
        if (nb > LN_MAX_BLOCKS) nb = LN_MAX_BLOCKS;                                                                \
        hipLaunchKernelGGL(ln_bwd_small_kernel<TPR>, dim3(nb), dim3(256), 0, st, da, z, gamma, beta, rows, C, Cp,  \
                           dz, part);                                                                              \
    } while (0)
    if (Cp <= 32) LAUNCH_SMALL(8);
    else if (Cp <= 64) LAUNCH_SMALL(16);
    else if (Cp <= 128) LAUNCH_SMALL(32);
    else if (Cp <= 256) LAUNCH_SMALL(64);
    else {
        ISDQN_REQUIRE(Cp <= 256 * LN_WIDE_MAX_COLS, ISDQN_ERR_UNSUPPORTED, "hidden dense width above 4096");
        nb = rows < LN_MAX_BLOCKS ? rows : LN_MAX_BLOCKS;
        hipLaunchKernelGGL(ln_bwd_wide_kernel, dim3(nb), dim3(256), 0, st, da, z, gamma, beta, rows, C, Cp, dz, part);
    }
#undef LAUNCH_SMALL
    ISDQN_HIP_CHECK(hipGetLastError());
    *n_blocks_out = nb;
    return ISDQN_OK;
}

template <int BM, int PASSES>
static int launch_conv_dgrad(const Layer& l, const float* wmir, const float* dz, float* da, int n_img,
                             hipStream_t st) {
    ConvDgrad<BM, PASSES> p;
    p.g = conv_geom(l);
    p.W = wmir + l.w_off;
    p.dz = dz; p.da = da; p.n_img = n_img;
    p.T = l.ksz / l.stride;
    p.Kc = p.T * p.T * l.cout_p;
    p.n_classes = l.stride * l.stride;
    int acc = 0;
    for (int c = 0; c < p.n_classes; ++c) {
        int cy = c / l.stride, cx = c % l.stride;
        int Ha = (l.hin - cy + l.stride - 1) / l.stride, Wb = (l.win - cx + l.stride - 1) / l.stride;
        p.tile_start[c] = acc;
        p.cls_d_hw[c] = FastDiv((uint32_t)(Ha * Wb));
        p.cls_d_w[c] = FastDiv((uint32_t)Wb);
        acc += ceil_div(n_img * Ha * Wb, 128);
    }
    p.tile_start[p.n_classes] = acc;
    return launch_gemm(p, acc, st);
}

template <int PASSES, bool U8>
static int launch_conv_wgrad(const Layer& l, const NetInput& in, const float* act_in, const float* dz, float* slabs,
                             int n_img, hipStream_t st) {
    ConvWgrad<PASSES, U8> p;
    p.g = conv_geom(l);
    p.n_pix = n_img * l.npix;
    p.DZ = MatSrc{dz, l.cout_p, p.n_pix, l.cout_p, 1};
    p.in = act_in;
    p.fs = FrameSrc{in.frames, in.frame_stride, in.frame_ids, l.cin, in.paired_B, l.hin, l.win, in.id_pitch, in.id_off};
    p.slabs = slabs;
    p.scale = U8 ? (1.0f / 255.0f) : 1.0f;
    p.tiles_n = ceil_div(l.K, 64);
    int ksteps = ceil_div(p.n_pix, GEMM_BK);
    p.steps_per_split = ceil_div(ksteps, l.gw_slabs);
    p.splits = ceil_div(ksteps, p.steps_per_split);
    return launch_gemm(p, p.tiles_n * p.splits, st);
}
// image-resident weight gradient (conv_img.h); *slabs_out = 0 when the layer does not qualify
static int conv_wgrad_img(const Layer& l, bool x3, const NetInput& in, const float* act_in, const float* dz, float* slabs,
                          int n_img, hipStream_t st, int* slabs_out) {
    *slabs_out = 0;
    if (!l.wgi_ntw || n_img != 0 && ceil_div(n_img, l.wgi_G) > l.gw_slabs) return ISDQN_OK;
    ConvWgradImgParams wp;
    wp.g = conv_geom(l);
    const int passes = x3 ? (l.is_u8 ? 2 : 3) : 1;
    const int mt = l.cout_p <= 32 ? 2 : 4;
    int lds_unused_R, Wp, plane;
    conv_img_geometry(wp.g, l.is_u8, l.cin, 1, mt, 1, lds_unused_R, Wp, plane);
    wp.R = l.stride * (l.hout - 1) + l.ksz;  // the whole image
    wp.Wp = Wp;
    // pixel pitches: an odd multiple of 32 bytes between consecutive contraction pixels (dz rows: PA; input: stride * PPin)
    wp.PPin = l.cin_p;
    for (int pad = 0; pad <= 24; pad += 8)
        if ((l.stride * (l.cin_p + pad)) % 32 == 16) { wp.PPin = l.cin_p + pad; break; }
    wp.in_plane = l.is_u8 ? l.cin * wp.R * Wp : wp.R * Wp * wp.PPin;
    wp.PA = l.cout_p + (l.cout_p % 32 == 0 ? 16 : 8);
    wp.npix_pad = round_up(l.npix, 32);
    wp.dz_plane = wp.npix_pad * wp.PA;
    const int lds = ((passes >= 2 ? 2 : 1) * wp.dz_plane + (passes >= 3 ? 2 : 1) * wp.in_plane) * 2;
    if (lds > 150 * 1024 || (l.is_u8 && (l.win < 8 || l.cin > 4))) return ISDQN_OK;
    wp.d_chunk = FastDiv((uint32_t)(l.is_u8 ? Wp / 8 : l.cin_p / 8));
    wp.d_Wp = FastDiv((uint32_t)Wp);
    wp.d_R = FastDiv((uint32_t)wp.R);
    wp.d_dzchunk = FastDiv((uint32_t)(l.cout_p / 8));
    wp.d_npixpad = FastDiv((uint32_t)wp.npix_pad);
    wp.dz = dz; wp.in = act_in;
    wp.fs = FrameSrc{in.frames, in.frame_stride, in.frame_ids, l.cin, in.paired_B, l.hin, l.win, in.id_pitch, in.id_off};
    wp.slabs = slabs;
    wp.scale = l.is_u8 ? (1.0f / 255.0f) : 1.0f;
    wp.n_img = n_img; wp.G = l.wgi_G;
    wp.NC = 64 * l.wgi_ntw;
    wp.n_col_groups = l.K / wp.NC;
    wp.d_ncg = FastDiv((uint32_t)wp.n_col_groups);
    {
        char tag[32];
        snprintf(tag, sizeof(tag), "wgrad:%s", l.name);
        wp.stamps = stamps_for(tag);
    }
    const int groups = ceil_div(n_img, l.wgi_G);
    *slabs_out = groups;
#define WGI(MT_, NTW_, P_, U_) return launch_conv_wgrad_img<MT_, NTW_, P_, U_>(wp, groups, st)
#define WGI8(MT_, NTW_, P_, U_) return launch_conv_wgrad_img<MT_, NTW_, P_, U_, 8>(wp, groups, st)
    if (l.is_u8) {
        if (mt == 2) { if (l.wgi_ntw == 4) { if (passes == 2) WGI(2, 4, 2, true); else WGI(2, 4, 1, true); }
                       if (l.wgi_ntw == 2) { if (passes == 2) WGI(2, 2, 2, true); else WGI(2, 2, 1, true); } }
        else         { if (l.wgi_ntw == 4) { if (passes == 2) WGI(4, 4, 2, true); else WGI(4, 4, 1, true); }
                       if (l.wgi_ntw == 2) { if (passes == 2) WGI(4, 2, 2, true); else WGI(4, 2, 1, true); } }
    } else {
        if (mt == 2) { if (l.wgi_ntw == 4) { if (passes == 3) WGI(2, 4, 3, false); else WGI(2, 4, 1, false); }
                       if (l.wgi_ntw == 3) { if (passes == 3) WGI(2, 3, 3, false); else WGI(2, 3, 1, false); }
                       if (l.wgi_ntw == 2) { if (passes == 3) WGI(2, 2, 3, false); else WGI(2, 2, 1, false); } }
        else         { // full-width tiles: 32 column tiles on eight waves (25.7 -> 21 us); the 36 tiles of a 3x3x64 layer stay on four
                       // (six or eight waves re-read the dz fragments too often: 33.5 -> 37 us)
                       static const bool w4 = ISDQN_DEV_ENV("ISDQN_WGRAD_4WAVES");
                       if (l.wgi_ntw == 8 && !w4) { if (passes == 3) WGI8(4, 4, 3, false); else WGI8(4, 4, 1, false); }
                       if (l.wgi_ntw == 9) { if (passes == 3) WGI(4, 9, 3, false); else WGI(4, 9, 1, false); }
                       if (l.wgi_ntw == 8) { if (passes == 3) WGI(4, 8, 3, false); else WGI(4, 8, 1, false); }
                       if (l.wgi_ntw == 4) { if (passes == 3) WGI(4, 4, 3, false); else WGI(4, 4, 1, false); }
                       if (l.wgi_ntw == 3) { if (passes == 3) WGI(4, 3, 3, false); else WGI(4, 3, 1, false); }
                       if (l.wgi_ntw == 2) { if (passes == 3) WGI(4, 2, 3, false); else WGI(4, 2, 1, false); } }
    }
#undef WGI
#undef WGI8
    *slabs_out = 0;  // (u8 with NTW 3 is not instantiated)
    return ISDQN_OK;
}

// image-resident data gradient of conv layer `l` fused with the LayerNorm/ReLU backward of layer `below`:
// writes dz of `below` and the reduced (dgamma, dbeta, dbias) row of `below`.  *done = false: not applicable.
static void add_reduce_job(ReduceJobs& jobs, const float* part, int n_rows, int width, float* out) {
    const int i = jobs.n++;
    jobs.part[i] = part; jobs.out[i] = out; jobs.n_rows[i] = n_rows; jobs.width[i] = width;
    jobs.block_start[i + 1] = jobs.block_start[i] + ceil_div(width, 8);
}

static int conv_dgrad_img(const Layer& l, const Layer& below, bool x3, const float* params, const float* wmir, const float* dz, float* ws,
                          int n_img, hipStream_t st, bool* done, ReduceJobs* jobs) {
    *done = false;
    if (!l.dgi_tiles || below.part_rows < n_img * l.dgi_tiles) return ISDQN_OK;
    ConvDgradImgParams dp;
    dp.g = conv_geom(l);
    dp.W = wmir + l.w_off;
    dp.dz = dz;
    dp.z_in = ws + below.z_off;
    dp.gamma = below.has_ln ? params + below.g_off : nullptr;
    dp.beta = below.has_ln ? params + below.be_off : nullptr;
    dp.c_in = below.out_f;
    dp.dz_in = ws + below.dz_off;
    dp.part = ws + below.part_off;
    dp.n_img = n_img;
    dp.T = l.ksz / l.stride;
    dp.Kc = dp.T * dp.T * l.cout_p;
    // top / left zero border of the dz image: the taps of a class reach dz rows (iy + pad - py) / stride - jy, jy < T, i.e. down
    // to pad / stride - (T - 1).  (Round 1 took T - 1 rows whatever the padding; for the 3x3 / 1 layer that is one row and one
    // column too many -- a 14 x 14 instead of a 13 x 13 image, 83 KB of LDS instead of 75: ONE workgroup per CU instead of two.)
#if defined(ISDQN_DGRAD_WIDE_BORDER)
    dp.bt = dp.T - 1;
#else
    dp.bt = dp.T - 1 - l.pad / l.stride > 0 ? dp.T - 1 - l.pad / l.stride : 0;
#endif
    const int need_h = (l.hin - 1 + l.pad) / l.stride + 1, need_w = (l.win - 1 + l.pad) / l.stride + 1;
    dp.Hd = dp.bt + (l.hout > need_h ? l.hout : need_h);
    dp.Wd = dp.bt + (l.wout > need_w ? l.wout : need_w);
    // pixel pitch with (pitch bytes / 16) = 2 (mod 4): conflict-free ds_read_b128 of 16 consecutive pixels (conv_img.h)
    dp.PPd = l.cout_p + ((16 - l.cout_p % 32) + 32) % 32;
    dp.dz_plane = dp.Hd * dp.Wd * dp.PPd;
    dp.d_chunk = FastDiv((uint32_t)(l.cout_p / 8));
    dp.d_Wd = FastDiv((uint32_t)dp.Wd);
    dp.d_T = FastDiv((uint32_t)dp.T);
    {
        char tag[32];
        snprintf(tag, sizeof(tag), "dgrad:%s", l.name);
        dp.stamps = stamps_for(tag);
    }
    dp.n_classes = l.stride * l.stride;
    int acc = 0;
    for (int c = 0; c < dp.n_classes; ++c) {
        int cy = c / l.stride, cx = c % l.stride;
        int Ha = (l.hin - cy + l.stride - 1) / l.stride, Wb = (l.win - cx + l.stride - 1) / l.stride;
        dp.cls_tile_start[c] = acc;
        dp.cls_order[c] = PixelOrder(Ha, Wb, Ha * Wb <= 128);  // (a class of several tiles keeps row-major tiles)
        acc += ceil_div(Ha * Wb, l.dgi_tile_pix);
    }
    dp.cls_tile_start[dp.n_classes] = acc;
    dp.tiles_per_img = acc;
    const int mt = l.cin_p <= 32 ? 2 : 4;
    const int passes = x3 ? 3 : 1;
    const int lds = (2 * (passes >= 2 ? 2 : 1) * 32 * (mt * 16 + 16) + (passes >= 3 ? 2 : 1) * dp.dz_plane) * 2;  // TileGeom<.., true>
    if (lds > 150 * 1024) return ISDQN_OK;
    int rc;
    ISDQN_REQUIRE(acc == l.dgi_tiles, ISDQN_ERR_ARG, "data-gradient tiling differs from the plan's");
#if ISDQN_DGRAD_TILE64_BELOW > 0
    if (l.dgi_tile_pix == 64) {
        if (passes == 3) rc = mt == 2 ? launch_conv_dgrad_img<2, 3, 1>(dp, st) : launch_conv_dgrad_img<4, 3, 1>(dp, st);
        else rc = mt == 2 ? launch_conv_dgrad_img<2, 1, 1>(dp, st) : launch_conv_dgrad_img<4, 1, 1>(dp, st);
    } else
#endif
    if (passes == 3) rc = mt == 2 ? launch_conv_dgrad_img<2, 3>(dp, st) : launch_conv_dgrad_img<4, 3>(dp, st);
    else rc = mt == 2 ? launch_conv_dgrad_img<2, 1>(dp, st) : launch_conv_dgrad_img<4, 1>(dp, st);
    if (rc) return rc;
    add_reduce_job(*jobs, ws + below.part_off, n_img * dp.tiles_per_img, 3 * below.out_p, ws + below.red_off);
    *done = true;
    return ISDQN_OK;
}

static int conv_wgrad_slabs(const Layer& l, int n_img) {
    int ksteps = ceil_div(n_img * l.npix, GEMM_BK);
    int sps = ceil_div(ksteps, l.gw_slabs);
    return ceil_div(ksteps, sps);
}

// (batchnorm.h, used by the impala torso's BatchNorm sites)
static inline const BnSite* bn_site_of(const Plan& P, int layer);
static int bn_site_forward(const BnSite& b, const float* params, float* ws, int rows, bool running, hipStream_t st);
static int bn_site_backward(const BnSite& b, const float* params, float* ws, float* dy, int rows, bool apply, hipStream_t st);
// Which heads a loss regresses (default: the plan's iterated pairs, online head oh + k on target head k, k < K).  The analysis
// agents evaluate single-pair losses on the multi-head network: online head 1 on target head 1 (analysisdqn.py:156-183).
struct HeadSel {
    int on0, tg0, K;
};

#include "impala.h"
#include "batchnorm.h"

}  // namespace isdqn

using namespace isdqn;

#if defined(ISDQN_BOUNDS)
// Bounds-checked development build (gemm_core.h: ISDQN_BOUNDS_CHECK): register the byte extents of every tensor the caller hands to
// the library -- [lo[i], hi[i]) -- and clear the record; read back the first load that fell outside all of them (synchronises).
extern "C" int isdqn_debug_bounds_set(const uint64_t* lo, const uint64_t* hi, int32_t n) {
    ISDQN_REQUIRE(n >= 0 && n <= 64, ISDQN_ERR_ARG, "at most 64 regions");
    isdqn::BoundsTable t;
    memset(&t, 0, sizeof(t));
    t.n = n;
    for (int i = 0; i < n; ++i) { t.lo[i] = lo[i]; t.hi[i] = hi[i]; }
    ISDQN_HIP_CHECK(hipDeviceSynchronize());
    ISDQN_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(isdqn::isdqn_bounds), &t, sizeof(t)));
    return ISDQN_OK;
}
extern "C" int isdqn_debug_bounds_get(int32_t* bad, int32_t* site, uint64_t* addr) {
    isdqn::BoundsTable t;
    ISDQN_HIP_CHECK(hipDeviceSynchronize());
    ISDQN_HIP_CHECK(hipMemcpyFromSymbol(&t, HIP_SYMBOL(isdqn::isdqn_bounds), sizeof(t)));
    *bad = t.bad; *site = t.bad_site; *addr = t.bad_addr;
    return ISDQN_OK;
}
#endif

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" int isdqn_net_param_layout(const isdqn_net_config* cfg, int64_t* n_param_floats, isdqn_tensor_info* infos,
                                      int32_t max_infos, int32_t* n_infos) {
    Plan P;
    int rc = build_plan(cfg, P);
    if (rc) return rc;
    if (n_param_floats) *n_param_floats = P.n_params;
    int n = 0;
    auto add = [&](const char* mod, const char* leaf, int64_t off, int64_t size, int kind, int layer, int ndim,
                   int s0, int s1, int s2, int s3, int d0, int d1, int d2, int d3 = 0) {
        if (infos && n < max_infos) {
            isdqn_tensor_info& t = infos[n];
            memset(&t, 0, sizeof(t));
            snprintf(t.name, sizeof(t.name), "%s/%s", mod, leaf);
            t.offset = off; t.size = size; t.kind = kind; t.layer = layer; t.ndim = ndim;
            t.flax_shape[0] = s0; t.flax_shape[1] = s1; t.flax_shape[2] = s2; t.flax_shape[3] = s3;
            t.dims[0] = d0; t.dims[1] = d1; t.dims[2] = d2; t.dims[3] = d3;
        }
        ++n;
    };
    for (int i = 0; i < P.n_layers; ++i) {
        const Layer& l = P.L[i];
        if (l.kind == 2) {  // impala torso: Stack_s / {Conv_k, LayerNorm_b} (Flax nests the Stack's modules: "Stack_0/Conv_1/kernel")
            for (int s = 0; s < IMP_STACKS; ++s) {
                const ImpalaStack& S = P.imp[s];
                char mod[40];
                for (int k = 0; k < IMP_CONVS; ++k) {
                    const Layer& c = S.conv[k];
                    if (k >= 1 && (k & 1) && S.ln_g[(k - 1) / 2] >= 0) {  // LayerNorm_b sits in front of Conv_{1+2b}
                        const int b = (k - 1) / 2;
                        snprintf(mod, sizeof(mod), "Stack_%d/LayerNorm_%d", s, b);
                        add(mod, "scale", S.ln_g[b], S.C_p, 3, i, 1, S.C, 0, 0, 0, S.C_p, 0, 0);
                        add(mod, "bias", S.ln_b[b], S.C_p, 4, i, 1, S.C, 0, 0, 0, S.C_p, 0, 0);
                    }
                    snprintf(mod, sizeof(mod), "Stack_%d/%s", s, c.name);
                    add(mod, "kernel", c.w_off, c.w_size, 0, i, 4, 3, 3, c.cin, c.cout, c.cout_p, c.taps, c.cin_p);
                    add(mod, "bias", c.b_off, c.out_p, 2, i, 1, c.cout, 0, 0, 0, c.out_p, 0, 0);
                }
            }
            if (l.has_ln) {
                add(l.ln_name, "scale", l.g_off, l.out_p, 3, i, 1, l.out_f, 0, 0, 0, l.out_p, 0, 0);
                add(l.ln_name, "bias", l.be_off, l.out_p, 4, i, 1, l.out_f, 0, 0, 0, l.out_p, 0, 0);
            }
            continue;
        }
        if (l.kind == 0) {
            add(l.name, "kernel", l.w_off, l.w_size, 0, i, 4, l.ksz, l.ksz, l.cin, l.cout, l.cout_p, l.is_u8 ? l.cin : l.taps,
                l.is_u8 ? 64 : l.cin_p);
        } else {
            add(l.name, "kernel", l.w_off, l.w_size, 1, i, 2, l.in_f, l.out_f, 0, 0, l.out_p, l.in_p, 0);
        }
        add(l.name, "bias", l.b_off, l.out_p, 2, i, 1, l.out_f, 0, 0, 0, l.out_p, 0, 0);
        if (l.has_ln) {
            add(l.ln_name, "scale", l.g_off, l.out_p, 3, i, 1, l.out_f, 0, 0, 0, l.out_p, 0, 0);
            add(l.ln_name, "bias", l.be_off, l.out_p, 4, i, 1, l.out_f, 0, 0, 0, l.out_p, 0, 0);
        }
    }
    for (int s = 0; s < P.n_bn; ++s) {  // BatchNorm_s: scale / bias ("params"), mean / var ("batch_stats")
        const BnSite& b = P.bns[s];
        int H, W;
        if (b.layer == -1) { H = P.L[0].hin; W = P.L[0].win; }
        else if (b.layer <= -2) { H = P.imp[(-2 - b.layer) / 2].Hp; W = P.imp[(-2 - b.layer) / 2].Wp; }
        else { H = P.L[b.layer].hout; W = P.L[b.layer].wout; }
        const int64_t offs[4] = {b.scale_off, b.bias_off, b.mean_off, b.var_off};
        static const char* const leaves[4] = {"scale", "bias", "mean", "var"};
        for (int k = 0; k < 4; ++k) {
            if (b.spatial) add(b.name, leaves[k], offs[k], b.G_p, 5 + k, b.layer, 2, H, W, 0, 0, b.G, b.P, b.C, b.Cp);
            else add(b.name, leaves[k], offs[k], b.G_p, 5 + k, b.layer, 1, b.P * b.C, 0, 0, 0, b.G, b.P, b.C, b.Cp);
        }
    }
    if (n_infos) *n_infos = n;
    return ISDQN_OK;
}

extern "C" int isdqn_net_workspace_bytes(const isdqn_net_config* cfg, int64_t* bytes) {
    Plan P;
    int rc = build_plan(cfg, P);
    if (rc) return rc;
    ISDQN_REQUIRE(bytes != nullptr, ISDQN_ERR_ARG, "null pointer");
    *bytes = P.ws_bytes;
    return ISDQN_OK;
}

extern "C" int isdqn_net_workspace_region(const isdqn_net_config* cfg, const char* name, int64_t* offset_bytes,
                                          int64_t* size_bytes) {
    Plan P;
    int rc = build_plan(cfg, P);
    if (rc) return rc;
    ISDQN_REQUIRE(name != nullptr, ISDQN_ERR_ARG, "null name");
    for (auto& r : P.regions)
        if (r.first == name) {
            if (offset_bytes) *offset_bytes = r.second.first;
            if (size_bytes) *size_bytes = r.second.second;
            return ISDQN_OK;
        }
    set_last_error("unknown workspace region '%s'", name);
    return ISDQN_ERR_ARG;
}

static int check_input(const isdqn_net_config* cfg, const uint8_t* frames, int64_t frame_stride,
                       const int32_t* frame_ids, const float* obs) {
    if (cfg->arch != ISDQN_ARCH_FC) {
        ISDQN_REQUIRE(frames && frame_ids, ISDQN_ERR_ARG, "cnn / impala need frames and frame_ids");
        ISDQN_REQUIRE(frame_stride >= (int64_t)cfg->obs_h * cfg->obs_w, ISDQN_ERR_SHAPE, "frame_stride < h*w");
    } else {
        ISDQN_REQUIRE(obs != nullptr, ISDQN_ERR_ARG, "fc needs obs");
    }
    return ISDQN_OK;
}

extern "C" int isdqn_net_forward(const isdqn_net_config* cfg, const float* params, const uint8_t* frames,
                                 int64_t frame_stride, const int32_t* frame_ids, const float* obs, int32_t n_rows,
                                 float* q_out, void* workspace, void* stream) {
    int rc;
    const Plan* Pp = cached_plan(cfg, &rc);
    if (!Pp) return rc;
    const Plan& P = *Pp;
    ISDQN_REQUIRE(params && q_out && workspace, ISDQN_ERR_ARG, "null pointer");
    ISDQN_REQUIRE(n_rows >= 1 && n_rows <= P.N2, ISDQN_ERR_SHAPE, "n_rows must be in [1, 2*batch_size]");
    rc = check_input(cfg, frames, frame_stride, frame_ids, obs);
    if (rc) return rc;
    NetInput in{frames, frame_stride, frame_ids, 0, obs, nullptr, 0};
    float* ws = (float*)workspace;
    hipStream_t st = (hipStream_t)stream;
    rc = refresh_mirror(P, params, ws, st);
    if (rc) return rc;
    rc = net_forward(P, cfg->precision == ISDQN_PRECISION_BF16X3, params, in, n_rows, 0, ws, ws + P.q_off, st);
    if (rc) return rc;
    // q_out is the unpadded (n_rows, nha) view
    ISDQN_HIP_CHECK(hipMemcpy2DAsync(q_out, (size_t)P.nha * 4, ws + P.q_off, (size_t)P.nha_p * 4, (size_t)P.nha * 4,
                                     n_rows, hipMemcpyDeviceToDevice, st));
    return ISDQN_OK;
}

// Side stream for the weight gradients.  The backward's critical path is dgrad -> LayerNorm-backward -> dgrad ...;
// every weight gradient only needs its layer's dz and is needed again by Adam at the very end, so they run on a
// second HIP stream and share the CUs with the data-gradient chain (both sides are partly latency bound and
// their workgroups co-reside).  One stream + event pool per device, created on first use, never destroyed.
struct SideStream {
    hipStream_t stream = nullptr;
    hipEvent_t ev[2 * MAX_LAYERS + 4];
    int n_ev = 0;
    std::atomic<unsigned> next{0};  // several agents / host threads of one process may share a device's pool
    std::atomic<int> state{0};      // 0 = not created, 1 = being created, 2 = ready, -1 = unavailable
};
static SideStream* side_stream() {
    static SideStream per_dev[ISDQN_MAX_DEVICES];
    if (ISDQN_DEV_ENV("ISDQN_SINGLE_STREAM")) return nullptr;
    SideStream& s = per_dev[current_device_slot()];
    int st = s.state.load(std::memory_order_acquire);
    if (st == 2) return &s;
    if (st == -1) return nullptr;
    int expected = 0;
    if (s.state.compare_exchange_strong(expected, 1)) {
        bool ok = hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) == hipSuccess;
        s.n_ev = 2 * MAX_LAYERS + 4;
        for (int i = 0; ok && i < s.n_ev; ++i) ok = hipEventCreateWithFlags(&s.ev[i], hipEventDisableTiming) == hipSuccess;
        s.state.store(ok ? 2 : -1, std::memory_order_release);
        return ok ? &s : nullptr;
    }
    while ((st = s.state.load(std::memory_order_acquire)) == 1) {}  // another thread is creating it
    return st == 2 ? &s : nullptr;
}
// make `waiter` wait for everything enqueued so far on `signaller`
static int chain(SideStream* ss, hipStream_t signaller, hipStream_t waiter) {
    hipEvent_t e = ss->ev[ss->next.fetch_add(1, std::memory_order_relaxed) % (unsigned)ss->n_ev];
    ISDQN_HIP_CHECK(hipEventRecord(e, signaller));
    ISDQN_HIP_CHECK(hipStreamWaitEvent(waiter, e, 0));
    return ISDQN_OK;
}

static int learn_or_loss(const isdqn_net_config* cfg, float* params, float* adam_m, float* adam_v, int32_t* adam_count,
                         const isdqn_batch* batch, float* losses, float* loss_accum, float* q_values, float* targets,
                         double* priorities, void* workspace, void* stream, bool learn, float* grad_out,
                         const float* target_params = nullptr, const HeadSel* sel = nullptr, bool update = true) {
    int rc;
    const Plan* Pp = cached_plan(cfg, &rc);
    if (!Pp) return rc;
    const Plan& P = *Pp;
    ISDQN_REQUIRE(params && batch && losses && workspace, ISDQN_ERR_ARG, "null pointer");
    ISDQN_REQUIRE(batch->B == P.B, ISDQN_ERR_SHAPE, "batch->B != cfg->batch_size");
    ISDQN_REQUIRE(batch->action && batch->reward && batch->terminal, ISDQN_ERR_ARG, "null batch field");
    rc = check_input(cfg, batch->frames, batch->frame_stride, batch->frame_ids, batch->state);
    if (rc) return rc;
    if (cfg->arch == ISDQN_ARCH_FC) ISDQN_REQUIRE(batch->next_state != nullptr, ISDQN_ERR_ARG, "fc needs next_state");
    if (learn && update) ISDQN_REQUIRE(adam_m && adam_v && adam_count, ISDQN_ERR_ARG, "null optimizer state");
    if (learn && !update) {
        ISDQN_REQUIRE(grad_out != nullptr, ISDQN_ERR_ARG, "a gradient-only pass needs grad_out");
        // the optimizer kernels request p / m / v before they know whether they update: a gradient-only pass has no moments, so
        // those (unused) loads read the parameters instead of a null pointer
        adam_m = params;
        adam_v = params;
    }
    if (P.bn) {  // BatchNorm: the layer-by-layer form over all 2B rows (batchnorm.h)
        ISDQN_REQUIRE(target_params == nullptr || (learn && !update), ISDQN_ERR_UNSUPPORTED,
                      "BatchNorm networks: separate target parameters only in gradient-only passes (the reference's DQN cannot run with "
                      "batch_norm, dqn.py:86)");
        return bn_learn_or_loss(cfg, P, params, adam_m, adam_v, adam_count, batch, losses, loss_accum, q_values, targets, priorities,
                                (float*)workspace, (hipStream_t)stream, learn, grad_out, update, target_params, sel);
    }
    const bool x3 = cfg->precision == ISDQN_PRECISION_BF16X3;
    const int B = P.B, K = sel ? sel->K : P.K;
    const int on0 = sel ? sel->on0 : P.oh, tg0 = sel ? sel->tg0 : 0;
    if (sel) ISDQN_REQUIRE(sel->K >= 1 && on0 >= 0 && tg0 >= 0 && on0 + K <= P.n_heads && tg0 + K <= P.n_heads && K <= P.K, ISDQN_ERR_ARG,
                           "head selection outside the network's heads");
    float* ws = (float*)workspace;
    hipStream_t st = (hipStream_t)stream;
    NetInput in{batch->frames, batch->frame_stride, batch->frame_ids, B, batch->state, batch->next_state, B};

    // ---- head chain eligibility (learn path): last hidden layer dense + ReLU, widths within the kernel's limits ----
    const Layer& hid = P.L[P.n_layers >= 2 ? P.n_layers - 2 : 0];
    const Layer& head = P.L[P.n_layers - 1];
    int hc_S = 0, hc_wg = 0;
    static const bool hc_disabled = ISDQN_DEV_ENV("ISDQN_NO_HEAD_CHAIN");
    if (learn && update && sel == nullptr && !hc_disabled && target_params == nullptr && P.n_layers >= 2 && hid.kind == 1 && !hid.is_head && hid.has_relu &&
        hid.out_p <= HC_THREADS * HC_MAX_COLS && hid.out_p % 8 == 0) {
        // transitions per workgroup: the per-transition phases scale with S (the kernel is instruction-issue bound) while
        // every workgroup streams the whole head matrix from L2, so S follows the batch: about 256 workgroups
        const int S = B >= 1024 ? 4 : B >= 512 ? 2 : 1;
        const int n_wg = ceil_div(B, S);
        if (n_wg <= hid.part_rows && S * K <= HC_THREADS &&
            head_chain_lds_bytes(hid.out_p, P.nha_p, K, x3 ? 3 : 1, S) <= 150 * 1024) {
            hc_S = S;
            hc_wg = n_wg;
        }
    }
    SideStream* ss = learn ? side_stream() : nullptr;
    hipStream_t wst = ss ? ss->stream : st;  // stream of the weight gradients
    const float* wmir = ws + P.wsplit_off;

    if (target_params != nullptr) {
        // DQN (dqn.py:74-88): the next states go through the TARGET parameters, the states through the online ones: two
        // forwards of B images each over the same workspace, q rows [B, 2B) first, then rows [0, B) (+ z of every layer)
        const int stack = cfg->arch != ISDQN_ARCH_FC ? cfg->obs_c : 0;
        NetInput nx{batch->frames, batch->frame_stride, batch->frame_ids, 0, batch->next_state, nullptr, 0, 2 * stack, stack};
        rc = refresh_mirror(P, target_params, ws, st);
        if (rc) return rc;
        rc = net_forward(P, x3, target_params, nx, B, 0, ws, ws + P.q_off + (int64_t)B * P.nha_p, st);
        if (rc) return rc;
        rc = refresh_mirror(P, params, ws, st);
        if (rc) return rc;
        NetInput on{batch->frames, batch->frame_stride, batch->frame_ids, 0, batch->state, nullptr, 0, 2 * stack, 0};
        rc = net_forward(P, x3, params, on, B, B, ws, ws + P.q_off, st);
        if (rc) return rc;
    } else {
        // The optimizer writes the updated parameters in both forms, so a learn step leaves the mirror current; a caller
        // that chains learn steps on one workspace with nothing else writing `params` in between says so
        // (ISDQN_BATCH_MIRROR_CURRENT: the captured multi-step graphs) and the refresh is skipped.
        if (!(batch->flags & ISDQN_BATCH_MIRROR_CURRENT)) {
            rc = refresh_mirror(P, params, ws, st);
            if (rc) return rc;
        }
        // ---- forward on concat(state, next_state) (isdqn.py:95) ----
        rc = net_forward(P, x3, params, in, P.N2, B, ws, ws + P.q_off, st, hc_S ? P.n_layers - 1 : -1,
                         hc_S ? P.n_layers - 2 : -1);
        if (rc) return rc;
    }

    // ---- targets, loss, dL/dq ----
    float* qv = q_values ? q_values : ws + P.qv_off;
    float* tg = targets ? targets : ws + P.tg_off;
    const int n_blk = hc_S ? hc_wg : ceil_div(B, TD_ROWS);
    float* loss_part = ws + P.lpart_off;
    float* dbh_part = loss_part + (int64_t)n_blk * K;
    float* adam_consts = ws + P.adam_tab_off;
    if (hc_S) {
        HeadChainParams hp;
        hp.act = ws + hid.act_off; hp.z = ws + hid.z_off;
        hp.slabs = ws + P.slab_off;
        hp.n_slabs = effective_splits(hid.in_unpadded_ld ? hid.in_f : hid.K, hid.fwd_splits);
        hp.slab_stride = hid.out_p;
        hp.row_pitch = (int64_t)hp.n_slabs * hid.out_p;
        hp.hbias = params + hid.b_off;
        hp.W = params + head.w_off; hp.bias = params + head.b_off;
        hp.gamma = hid.has_ln ? params + hid.g_off : nullptr;
        hp.beta = hid.has_ln ? params + hid.be_off : nullptr;
        hp.B = B; hp.S = hc_S; hp.F = hid.out_f; hp.Fp = hid.out_p; hp.O = P.nha; hp.Op = P.nha_p; hp.K = K; hp.oh = P.oh;
        hp.A = P.n_actions;
        hp.action = batch->action; hp.reward = batch->reward; hp.terminal = batch->terminal;
        hp.gamma_n = cfg->gamma_n; hp.huber_delta = cfg->huber_delta;
        hp.dout = ws + P.dout_off; hp.dz = ws + hid.dz_off; hp.part = ws + hid.part_off;
        hp.q_values = qv; hp.targets = tg; hp.priorities = priorities;
        hp.loss_part = loss_part; hp.dbh_part = dbh_part;
        hp.adam_count = adam_count; hp.b1 = cfg->adam_b1; hp.b2 = cfg->adam_b2; hp.adam_consts = adam_consts;
        hp.stamps = stamps_for("head_chain");
        const int lds = head_chain_lds_bytes(hid.out_p, P.nha_p, K, x3 ? 3 : 1, hc_S);
        const int cols = ceil_div(hid.out_p, HC_THREADS) <= 1 ? 1 : ceil_div(hid.out_p, HC_THREADS) <= 2 ? 2 : 4;
        auto launch_hc = [&](auto kern, int slot) -> int {
            static LdsConfigured configured[18];
            if (int rc2 = ensure_dynamic_lds(kern, lds, configured[slot])) return rc2;
            ISDQN_REPORT_OCCUPANCY(kern, HC_THREADS, lds, hc_wg);
            hipLaunchKernelGGL(kern, dim3(hc_wg), dim3(HC_THREADS), lds, st, hp);
            ISDQN_HIP_CHECK(hipGetLastError());
            return ISDQN_OK;
        };
        auto by_cols = [&](auto passes_c, auto s_c, int base) -> int {
            constexpr int PS = decltype(passes_c)::value, SS = decltype(s_c)::value;
            return cols == 1 ? launch_hc(&head_chain_kernel<PS, 1, SS>, base) : cols == 2 ? launch_hc(&head_chain_kernel<PS, 2, SS>, base + 1)
                                                                            : launch_hc(&head_chain_kernel<PS, 4, SS>, base + 2);
        };
        auto by_s = [&](auto passes_c, int base) -> int {
            return hc_S == 1 ? by_cols(passes_c, std::integral_constant<int, 1>{}, base)
                 : hc_S == 2 ? by_cols(passes_c, std::integral_constant<int, 2>{}, base + 3)
                             : by_cols(passes_c, std::integral_constant<int, 4>{}, base + 6);
        };
        rc = x3 ? by_s(std::integral_constant<int, 3>{}, 0) : by_s(std::integral_constant<int, 1>{}, 9);
        if (rc) return rc;
        // The loss / head-bias reductions and the head's weight gradient leave the critical path: with a weight-
        // gradient stream they are enqueued there behind the first fork the backward pass makes anyway (every fork
        // costs the main stream a dependency bubble, so none is spent on these two small kernels alone).
        if (!ss) {
            hipLaunchKernelGGL(loss_finalize_kernel, dim3(ceil_div(K, 16) + ceil_div(P.nha_p, 16)), dim3(256), 0, st, loss_part, dbh_part, n_blk, B, K, P.nha_p,
                               losses, loss_accum, ws + P.dbh_off, (int*)nullptr, cfg->adam_b1, cfg->adam_b2, adam_consts);
            ISDQN_HIP_CHECK(hipGetLastError());
        }
    } else {
        hipLaunchKernelGGL(td_kernel, dim3(n_blk), dim3(256), 2 * TD_ROWS * K * sizeof(float), st, ws + P.q_off, B, K,
                           on0, tg0, P.n_actions, P.nha_p, batch->action, batch->reward, batch->terminal, cfg->gamma_n, cfg->huber_delta,
                           learn ? ws + P.dout_off : nullptr, qv, tg, priorities, loss_part, dbh_part);
        ISDQN_HIP_CHECK(hipGetLastError());
        hipLaunchKernelGGL(loss_finalize_kernel, dim3(ceil_div(K, 16) + ceil_div(P.nha_p, 16)), dim3(256), 0, st, loss_part, dbh_part, n_blk, B, K, P.nha_p,
                           losses, loss_accum, learn ? ws + P.dbh_off : nullptr, (learn && update) ? adam_count : nullptr,
                           cfg->adam_b1, cfg->adam_b2, adam_consts);
        ISDQN_HIP_CHECK(hipGetLastError());
    }
    // q_values / targets / priorities / per-head loss partials are final here.  The caller's event (batch->priorities_ready) is
    // recorded at the first point that forks the caller's stream anyway (every fork costs it a dependency bubble), else now.
    bool prio_pending = batch->priorities_ready != nullptr;
    auto signal_priorities = [&]() -> int {
        if (prio_pending) {
            prio_pending = false;
            ISDQN_HIP_CHECK(hipEventRecord((hipEvent_t)batch->priorities_ready, st));
        }
        return ISDQN_OK;
    };
    if (!learn || !ss) {
        rc = signal_priorities();
        if (rc) return rc;
    }
    if (!learn) return ISDQN_OK;

    // ---- backward (online rows only: the next-state half has a zero cotangent, isdqn.py:99) ----
    // Two optimizer tables: tensors whose gradient is produced on the weight-gradient stream are updated there, the others on
    // the caller's stream, so the step ends with two short Adam launches side by side instead of join -> reduce -> one Adam.
    AdamTable tabs[2];  // [0] caller's stream, [1] weight-gradient stream
    tabs[0].n = tabs[1].n = 0;
    tabs[0].total_blocks = tabs[1].total_blocks = 0;
    auto add_entry_on = [&](int which, int64_t p_off, int64_t size, const float* g, int n_slabs, int64_t stride) {
        AdamTable& tab = tabs[which];
        AdamEntry& e = tab.e[tab.n++];
        e.p_off = p_off; e.size = size; e.g = g; e.n_slabs = n_slabs; e.slab_stride = stride;
        e.block_start = tab.total_blocks;
        tab.total_blocks += (int)((size + 63) / 64);
    };
    auto add_entry = [&](int64_t p_off, int64_t size, const float* g, int n_slabs, int64_t stride) {
        add_entry_on(0, p_off, size, g, n_slabs, stride);
    };
    // With a weight-gradient stream and at least three layers, the last two weight gradients swap streams: layer 1's follows
    // its data gradient on the caller's stream, layer 0's (which only needs that data gradient's output) runs beside it.
    const bool tail_swap = ss && P.n_layers >= 3 && !P.L[1].is_head;
    bool forked = false, layer0_chained = false;
    const float* dz_cur = ws + P.dout_off;  // gradient w.r.t. the current layer's pre-activation output
    int dz_ld = P.nha_p;
    bool dz_fused = false;  // dz of layer i was already produced by the fused data gradient of layer i+1
    ReduceJobs red_jobs;    // partial-row reductions left by the fused data gradients (one launch before Adam)
    red_jobs.n = 0;
    red_jobs.block_start[0] = 0;
    bool head_deferred = hc_S && ss;  // loss_finalize + head weight gradient still to be enqueued on the side stream
    auto run_head_deferred = [&](hipStream_t s2) -> int {
        if (!head_deferred) return ISDQN_OK;
        head_deferred = false;
        hipLaunchKernelGGL(loss_finalize_kernel, dim3(ceil_div(K, 16) + ceil_div(P.nha_p, 16)), dim3(256), 0, s2, loss_part, dbh_part, n_blk, B, K, P.nha_p, losses,
                           loss_accum, ws + P.dbh_off, (int*)nullptr, cfg->adam_b1, cfg->adam_b2, adam_consts);
        ISDQN_HIP_CHECK(hipGetLastError());
        MatSrc A{ws + P.dout_off, P.nha_p, B, head.out_p, 1};
        MatSrc Bm{ws + hid.act_off, head.in_p, B, head.in_p, 1};
        return plain_big<true, true, true, false, 2>(x3, A, nullptr, 0, Bm, ws + head.gw_off, head.in_p, head.out_p, head.in_p, B,
                                                     head.gw_slabs, head.w_size, s2);  // dL/dq fp32, hidden activations S8
    };
    std::function<int()> deferred;  // side-stream launches of the layer above, held back until this layer's data gradient is enqueued
    // Which queue the replayed graph gives a kernel: a node's FIRST-created successor stays on its queue, the others move to
    // another one and start ~12 us late.  Round 4's order (c2 +1.2 %, c3 +7.5 %; -DISDQN_FORK_LATE_MAX_B=0 keeps round 3's):
    //   * the data-gradient chain is created first behind every fork, so it stays on the capturing queue all the way;
    //   * the two small kernels that only need the head chain's outputs (loss sums, head weight gradient) become successors of the
    //     HEAD CHAIN on the side stream -- created behind the dense data gradient, which therefore stays the head chain's first
    //     successor -- and run under the dense data gradient; the new queue's late start is hidden there, and it is what lets the
    //     first convolution data gradient become resident before the fused-Adam GEMM's 968 workgroups ask for the CUs.
    // Large batches keep round 3's order: at B = 1024 every kernel is several rounds of workgroups, which queue gets the CUs first
    // no longer matters and the new order measured 0.6 % slower (profiles/round4/ab_fork_c5.txt; ab_thresholds.txt: +2.0 % at
    // B = 32, +1.7 % at 128, +1.5 % at 512, +2.0 % at 768).
    hipEvent_t head_event = nullptr;
#if !defined(ISDQN_FORK_LATE_MAX_B)
#define ISDQN_FORK_LATE_MAX_B 768
#endif
    const bool fork_late = ss != nullptr && P.L[0].kind != 2 && B <= ISDQN_FORK_LATE_MAX_B;
    if (fork_late && head_deferred) {
        head_event = ss->ev[ss->next.fetch_add(1, std::memory_order_relaxed) % (unsigned)ss->n_ev];
        ISDQN_HIP_CHECK(hipEventRecord(head_event, st));
    }
    for (int i = P.n_layers - 1; i >= 0; --i) {
        const Layer& l = P.L[i];
        const float* act_in = i > 0 ? ws + P.L[i - 1].act_off : nullptr;
        if (!l.is_head) {
            dz_cur = ws + l.dz_off;
            dz_ld = l.out_p;
            if (hc_S && i == P.n_layers - 2) {  // dz and the partial sums came from the head chain
                if (l.has_ln) {
                    add_entry(l.g_off, l.out_p, ws + l.part_off, hc_wg, 3 * (int64_t)l.out_p);
                    add_entry(l.be_off, l.out_p, ws + l.part_off + l.out_p, hc_wg, 3 * (int64_t)l.out_p);
                }
                add_entry(l.b_off, l.out_p, ws + l.part_off + 2 * l.out_p, hc_wg, 3 * (int64_t)l.out_p);
            } else if (dz_fused) {
                if (l.has_ln) {
                    add_entry(l.g_off, l.out_p, ws + l.red_off, 1, 0);
                    add_entry(l.be_off, l.out_p, ws + l.red_off + l.out_p, 1, 0);
                }
                if (l.b_off >= 0) add_entry(l.b_off, l.out_p, ws + l.red_off + 2 * l.out_p, 1, 0);  // (the impala torso has no bias of its own)
            } else {
                // da (w.r.t. this layer's activation) was left in ws+da_off by layer i+1's data-gradient
                int rows = l.kind != 1 ? B * l.npix : B;
                int nb = 0;
                rc = ln_bwd(l, params, ws + P.da_off, ws + l.z_off, rows, ws + l.dz_off, ws + l.part_off, &nb, st);
                if (rc) return rc;
                if (l.has_ln) {
                    add_entry(l.g_off, l.out_p, ws + l.part_off, nb, 3 * (int64_t)l.out_p);
                    add_entry(l.be_off, l.out_p, ws + l.part_off + l.out_p, nb, 3 * (int64_t)l.out_p);
                }
                if (l.b_off >= 0) add_entry(l.b_off, l.out_p, ws + l.part_off + 2 * l.out_p, nb, 3 * (int64_t)l.out_p);
            }
        } else {
            add_entry_on(hc_S && ss ? 1 : 0, l.b_off, l.out_p, ws + P.dbh_off, 1, 0);  // loss_finalize_kernel's stream
        }
        if (l.kind == 2) {  // the impala torso: its own backward (generic engine + row-wise kernels) and optimizer launches, on the caller's stream
            rc = impala_backward(P, cfg, x3, params, adam_m, adam_v, wmir, ws, B, grad_out, update, st);
            if (rc) return rc;
            continue;
        }
        // Weight gradients of the middle layers go to the side stream.  Every fork costs the main stream an event
        // record (a ~6 us bubble), so the head's tiny weight gradient and the first layer's (nothing is left to
        // overlap with) stay on the main stream, and a layer whose Adam is fused forks once, after its data gradient.
        const bool head_chained = l.is_head && hc_S;  // the fork happened right after the head chain
        const bool wg_on_side = ss && !l.is_head && (tail_swap ? i != 1 : i > 0);
        hipStream_t lws = (wg_on_side || head_chained) ? wst : st;
        const bool fork_after_dgrad = wg_on_side && l.kind == 1 && i > 0;
        if (wg_on_side && !fork_after_dgrad && !(i == 0 && layer0_chained)) {  // dz of this layer is final on the main stream
            rc = chain(ss, st, lws);
            if (rc) return rc;
            rc = signal_priorities();
            if (rc) return rc;
            forked = true;
            if (!deferred) {  // (held-back side work of the layer above enqueues the two head kernels itself, behind this layer's data gradient)
                rc = run_head_deferred(lws);
                if (rc) return rc;
            }
        }
        // data gradient for the layer below first: it reads this layer's weights, which the fused-Adam
        // weight-gradient epilogue below updates in place
        dz_fused = false;
        if (i > 0) {
            if (l.kind == 0) {
                rc = conv_dgrad_img(l, P.L[i - 1], x3, params, wmir, dz_cur, ws, B, st, &dz_fused, &red_jobs);
                if (rc) return rc;
            }
            if (l.kind == 1 && P.L[i - 1].kind != 1 && P.L[i - 1].cout_p == 64 && !l.in_unpadded_ld &&
                P.L[i - 1].part_rows >= ceil_div(B, 128) * P.L[i - 1].npix) {
                // first dense layer over a 64-channel conv output: data gradient + LN/ReLU backward in one kernel
                const Layer& below = P.L[i - 1];
                auto launch = [&](auto prob) {
                    prob.A = MatSrc{dz_cur, dz_ld, B, l.out_p, 1};
                    prob.B = MatSrc{wmir + l.w_off, l.in_p, l.out_f, l.in_p, 1};
                    prob.z = ws + below.z_off;
                    prob.gamma = below.has_ln ? params + below.g_off : nullptr;
                    prob.beta = below.has_ln ? params + below.be_off : nullptr;
                    prob.dz_out = ws + below.dz_off;
                    prob.part = ws + below.part_off;
                    prob.ldc = l.in_p; prob.M = B; prob.N = l.in_p; prob.K = l.out_p; prob.c_in = below.out_f;
                    prob.tiles_m = ceil_div(B, decltype(prob)::BM); prob.tiles_n = l.in_p / 64;
                    return launch_gemm(prob, prob.tiles_m * prob.tiles_n, st);
                };
                // 128-row tiles (8 accumulators per wave).  64-row tiles fill the chip better (196 workgroups, -3 us)
                // but are a four-accumulator kernel, and those are not run-to-run stable on gfx950
                // at most one workgroup per CU (242 at the headline size): two K groups of four waves (gemm_core.h)
                static const bool no_kg = ISDQN_DEV_ENV("ISDQN_NO_KGROUPS");
                const bool kg2 = !no_kg && ceil_div(B, 128) * (l.in_p / 64) <= 256 && l.out_p % 64 == 0;
                // ISDQN_DGRAD64 (development): 64-row tiles, one K group
#if defined(ISDQN_DGRAD64)
                (void)kg2;
                rc = x3 ? launch(DenseDgradLN<3, 64>{}) : launch(DenseDgradLN<1, 64>{});
                constexpr int DG_BM = 64;
#else
                if (kg2) rc = x3 ? launch(DenseDgradLN<3, 128, 2>{}) : launch(DenseDgradLN<1, 128, 2>{});
                else rc = x3 ? launch(DenseDgradLN<3, 128>{}) : launch(DenseDgradLN<1, 128>{});
                constexpr int DG_BM = 128;
#endif
                if (rc) return rc;
                add_reduce_job(red_jobs, ws + below.part_off, ceil_div(B, DG_BM) * (l.in_p / 64), 3 * below.out_p,
                               ws + below.red_off);
                dz_fused = true;
            }
            if (dz_fused || head_chained) {
            } else if (l.kind == 0) {
                const bool small = l.cin_p <= 32;
                if (x3) rc = small ? launch_conv_dgrad<32, 3>(l, wmir, dz_cur, ws + P.da_off, B, st)
                                   : launch_conv_dgrad<64, 3>(l, wmir, dz_cur, ws + P.da_off, B, st);
                else rc = small ? launch_conv_dgrad<32, 1>(l, wmir, dz_cur, ws + P.da_off, B, st)
                                : launch_conv_dgrad<64, 1>(l, wmir, dz_cur, ws + P.da_off, B, st);
            } else {
                // da[b][in_p] = sum_o dz[b][o] * W[o][in_p]
                MatSrc A{dz_cur, dz_ld, B, l.out_p, 1};
                MatSrc Bm{wmir + l.w_off, l.in_p, l.out_f, l.in_p, 1};
                const bool narrow = ceil_div(B, 128) * ceil_div(l.in_p, 128) < 200;
                if (l.is_head)  // dL/dq: fp32
                    rc = narrow ? plain_narrow<false, true, 2>(x3, A, Bm, ws + P.da_off, l.in_p, B, l.in_p, l.out_p, 1, 0, st)
                                : plain_big<false, true, true, false, 2>(x3, A, nullptr, 0, Bm, ws + P.da_off, l.in_p, B, l.in_p, l.out_p, 1, 0, st);
                else
                    rc = narrow ? plain_narrow<false, true, DZ_S8 | 2>(x3, A, Bm, ws + P.da_off, l.in_p, B, l.in_p, l.out_p, 1, 0, st)
                                : plain_big<false, true, true, false, DZ_S8 | 2>(x3, A, nullptr, 0, Bm, ws + P.da_off, l.in_p, B, l.in_p, l.out_p, 1, 0, st);
            }
            if (rc) return rc;
        }
        if (deferred) {  // the side-stream work of the layer above, now that this layer's data gradient is enqueued
            auto f = std::move(deferred);
            deferred = nullptr;
            rc = f();
            if (rc) return rc;
        }
        // weight gradient -> slabs (or straight into Adam when one workgroup holds the whole contraction)
        bool defer_side = false;
        hipEvent_t defer_event = nullptr;
        if (fork_after_dgrad) {  // dz is final AND the data gradient (which reads W) is enqueued: an in-place
                                 // fused-Adam update on the side stream cannot overtake it
            if (fork_late && i > 0) {
                // Graph replay keeps the FIRST successor of a node on its queue and moves the others to another queue, where they
                // start ~12 us late.  The successor that matters is the next data gradient: record the fork here,
                // but create the side stream's kernels (loss_finalize, head weight gradient, this layer's weight gradient) only
                // after that data gradient has been enqueued on the caller's stream.
                if (head_event != nullptr && head_deferred) {  // (the dense data gradient is enqueued: the head chain keeps it as first successor)
                    ISDQN_HIP_CHECK(hipStreamWaitEvent(lws, head_event, 0));
                    rc = run_head_deferred(lws);
                    if (rc) return rc;
                }
                defer_event = ss->ev[ss->next.fetch_add(1, std::memory_order_relaxed) % (unsigned)ss->n_ev];
                ISDQN_HIP_CHECK(hipEventRecord(defer_event, st));
                defer_side = true;
                rc = signal_priorities();
                if (rc) return rc;
                forked = true;
            } else {
                rc = chain(ss, st, lws);
                if (rc) return rc;
                rc = signal_priorities();
                if (rc) return rc;
                forked = true;
                rc = run_head_deferred(lws);
                if (rc) return rc;
            }
        }
        if (tail_swap && i == 1 && dz_fused) {  // layer 0's dz is final: its weight gradient may start beside this layer's
            rc = chain(ss, st, wst);
            if (rc) return rc;
            forked = layer0_chained = true;
            rc = run_head_deferred(wst);
            if (rc) return rc;
        }
        // (the weight-gradient part of this layer as a closure: with ISDQN_FORK_LATE a layer that forks the side stream behind its
        // own data gradient enqueues it only after the NEXT layer's data gradient is on the caller's stream -- see `deferred`)
        const Layer* lp = &l;
        auto wgrad_part = [&, lp, lws, head_chained, dz_cur, dz_ld, act_in]() -> int {
            const Layer& l = *lp;
            int rc = ISDQN_OK;
            int w_slabs;
            bool fused_adam = false;
            if (head_chained && ss) {  // enqueued by run_head_deferred()
                add_entry_on(1, l.w_off, l.w_size, ws + l.gw_off, effective_splits(B, l.gw_slabs), l.w_size);
                return ISDQN_OK;
            }
            if (l.kind == 0) {
                int img_slabs = 0;
                rc = conv_wgrad_img(l, x3, in, act_in, dz_cur, ws + l.gw_off, B, lws, &img_slabs);
                if (rc) return rc;
                if (img_slabs) {
                    w_slabs = img_slabs;
                } else {
                if (l.is_u8) rc = x3 ? launch_conv_wgrad<2, true>(l, in, act_in, dz_cur, ws + l.gw_off, B, lws)
                                     : launch_conv_wgrad<1, true>(l, in, act_in, dz_cur, ws + l.gw_off, B, lws);
                else rc = x3 ? launch_conv_wgrad<3, false>(l, in, act_in, dz_cur, ws + l.gw_off, B, lws)
                             : launch_conv_wgrad<1, false>(l, in, act_in, dz_cur, ws + l.gw_off, B, lws);
                w_slabs = conv_wgrad_slabs(l, B);
                }
            } else {
                // dW[out][in_p] = sum_b dz[b][out] * a[b][in_p] : both operands stored [K = b][rows]
                MatSrc A{dz_cur, dz_ld, B, l.out_p, 1};
                MatSrc Bm = l.in_unpadded_ld ? MatSrc{in.obs, l.in_unpadded_ld, B, l.in_f, 0}
                                             : MatSrc{act_in, l.in_p, B, l.in_p, 1};
                w_slabs = effective_splits(B, l.gw_slabs);
                if (w_slabs == 1 && !l.in_unpadded_ld) {
                    fused_adam = true;  // (a gradient-only pass runs the same kernel with the stores of p / m / v switched off)

                    AdamFuse af{params + l.w_off, adam_m + l.w_off, adam_v + l.w_off, ws + P.adam_tab_off,
                                cfg->learning_rate, cfg->adam_b1, cfg->adam_b2, cfg->adam_eps,
                                grad_out ? grad_out + l.w_off : nullptr, ws + P.wsplit_off + l.w_off, update ? 1 : 0};
                    // 64x64 tiles: the contraction is only B deep, the kernel lives off streaming p/m/v through the Adam
                    // epilogue, and 128x128 tiles would leave 100 workgroups for 256 CUs
                    // (operands: dz of a hidden layer -- or the fp32 dL/dq of the head -- and the S8 activations below it)
                    if (l.is_head)
                        rc = x3 ? launch_plain<64, 64, 2, 2, true, true, 3, true, false, true, 2>(A, nullptr, 0, Bm, nullptr, l.in_p,
                                                                                                 l.out_p, l.in_p, B, 1, 0, lws, &af)
                                : launch_plain<64, 64, 2, 2, true, true, 1, true, false, true, 2>(A, nullptr, 0, Bm, nullptr, l.in_p,
                                                                                                 l.out_p, l.in_p, B, 1, 0, lws, &af);
                    else
                        rc = x3 ? launch_plain<64, 64, 2, 2, true, true, 3, true, false, true, DZ_S8 | 2>(A, nullptr, 0, Bm, nullptr, l.in_p,
                                                                                                         l.out_p, l.in_p, B, 1, 0, lws, &af)
                                : launch_plain<64, 64, 2, 2, true, true, 1, true, false, true, DZ_S8 | 2>(A, nullptr, 0, Bm, nullptr, l.in_p,
                                                                                                         l.out_p, l.in_p, B, 1, 0, lws, &af);
                } else if (l.in_unpadded_ld) {  // fc first layer: caller's fp32 observations
                    rc = plain_big<true, true, false, false, DZ_S8>(x3, A, nullptr, 0, Bm, ws + l.gw_off, l.in_p, l.out_p, l.in_p, B,
                                                                    l.gw_slabs, l.w_size, lws);
                } else if (l.is_head) {         // dL/dq is fp32, the hidden activations are S8
                    rc = plain_big<true, true, true, false, 2>(x3, A, nullptr, 0, Bm, ws + l.gw_off, l.in_p, l.out_p, l.in_p, B,
                                                               l.gw_slabs, l.w_size, lws);
                } else {
                    rc = plain_big<true, true, true, false, DZ_S8 | 2>(x3, A, nullptr, 0, Bm, ws + l.gw_off, l.in_p, l.out_p, l.in_p, B,
                                                                       l.gw_slabs, l.w_size, lws);
                }
            }
            if (rc) return rc;
            if (!fused_adam) add_entry_on(lws == wst && ss ? 1 : 0, l.w_off, l.w_size, ws + l.gw_off, w_slabs, l.w_size);
            return ISDQN_OK;
        };
        if (defer_side) {
            hipEvent_t e = defer_event;
            deferred = [&, e, lws, wgrad_part]() -> int {
                ISDQN_HIP_CHECK(hipStreamWaitEvent(lws, e, 0));
                if (int r = run_head_deferred(lws)) return r;
                return wgrad_part();
            };
        } else {
            rc = wgrad_part();
            if (rc) return rc;
        }
    }
    if (deferred) {
        auto f = std::move(deferred);
        deferred = nullptr;
        rc = f();
        if (rc) return rc;
    }
    if (head_deferred) {  // no layer forked: keep the two kernels in line
        rc = run_head_deferred(st);
        if (rc) return rc;
    }
    rc = signal_priorities();
    if (rc) return rc;
    auto run_adam = [&](const AdamTable& tab, hipStream_t s2) -> int {
        if (tab.n == 0) return ISDQN_OK;
        hipLaunchKernelGGL(adam_kernel, dim3(tab.total_blocks), dim3(256), 0, s2, tab, params, adam_m, adam_v, ws + P.adam_tab_off,
                           cfg->learning_rate, cfg->adam_b1, cfg->adam_b2, cfg->adam_eps, grad_out, ws + P.wsplit_off, update ? 1 : 0);
        ISDQN_HIP_CHECK(hipGetLastError());
        return ISDQN_OK;
    };
    // The weight-gradient stream may update its own tensors only when every reader of the weights is behind it: with the
    // tail swap its last kernel (layer 0's weight gradient) waited for the last data gradient of the caller's stream.
    const bool adam_on_side = ss && forked && tail_swap;
    if (adam_on_side) {
        rc = run_adam(tabs[1], wst);
        if (rc) return rc;
    } else if (ss) {  // all weight gradients done before Adam
        rc = chain(ss, wst, st);
        if (rc) return rc;
    }
    if (red_jobs.n > 0) {
        hipLaunchKernelGGL(reduce_rows_kernel, dim3(red_jobs.block_start[red_jobs.n]), dim3(256), 0, st, red_jobs);
        ISDQN_HIP_CHECK(hipGetLastError());
    }
    rc = run_adam(tabs[0], st);
    if (rc) return rc;
    if (adam_on_side) {  // the call is complete on the caller's stream (and the next call may touch the workspace)
        rc = chain(ss, wst, st);
        if (rc) return rc;
    } else {
        rc = run_adam(tabs[1], st);
        if (rc) return rc;
    }
    return ISDQN_OK;
}

extern "C" int isdqn_net_learn_on_batch(const isdqn_net_config* cfg, float* params, float* adam_m, float* adam_v,
                                        int32_t* adam_count, const isdqn_batch* batch, float* losses,
                                        float* losses_accum, float* q_values, float* targets, double* priorities,
                                        void* workspace, void* stream) {
    return learn_or_loss(cfg, params, adam_m, adam_v, adam_count, batch, losses, losses_accum, q_values, targets,
                         priorities, workspace, stream, true, nullptr);
}

// Test hook (not in the public header): learn_on_batch that also writes the reduced gradient (internal layout).
extern "C" int isdqn_net_learn_on_batch_debug(const isdqn_net_config* cfg, float* params, float* adam_m, float* adam_v,
                                              int32_t* adam_count, const isdqn_batch* batch, float* losses,
                                              float* losses_accum, float* q_values, float* targets, double* priorities,
                                              void* workspace, void* stream, float* grad_out) {
    return learn_or_loss(cfg, params, adam_m, adam_v, adam_count, batch, losses, losses_accum, q_values, targets,
                         priorities, workspace, stream, true, grad_out);
}

extern "C" int isdqn_net_loss_on_batch(const isdqn_net_config* cfg, const float* params, const isdqn_batch* batch,
                                       float* losses, float* q_values, float* targets, void* workspace, void* stream) {
    return learn_or_loss(cfg, const_cast<float*>(params), nullptr, nullptr, nullptr, batch, losses, nullptr, q_values,
                         targets, nullptr, workspace, stream, false, nullptr);
}

// Gradient of a TD loss without an update (the diagnostics of AnalysisDQN, analysisdqn.py:156-219): `n_pairs` > 0 regresses
// online heads online_head + k on target heads target_head + k (k < n_pairs) instead of the plan's iterated pairs;
// `target_params` != NULL takes the next states through those parameters (the target-based loss).
extern "C" int isdqn_net_grad_on_batch(const isdqn_net_config* cfg, const float* params, const float* target_params,
                                       const isdqn_batch* batch, int32_t online_head, int32_t target_head, int32_t n_pairs,
                                       float* grad_out, float* losses, float* q_values, float* targets, void* workspace,
                                       void* stream) {
    ISDQN_REQUIRE(grad_out != nullptr, ISDQN_ERR_ARG, "null grad_out");
    HeadSel sel{online_head, target_head, n_pairs};
    return learn_or_loss(cfg, const_cast<float*>(params), nullptr, nullptr, nullptr, batch, losses, nullptr, q_values, targets, nullptr,
                         workspace, stream, true, grad_out, target_params, n_pairs > 0 ? &sel : nullptr, false);
}

extern "C" int isdqn_net_refresh_mirror(const isdqn_net_config* cfg, const float* params, void* workspace, void* stream) {
    int rc;
    const Plan* Pp = cached_plan(cfg, &rc);
    if (!Pp) return rc;
    ISDQN_REQUIRE(params != nullptr && workspace != nullptr, ISDQN_ERR_ARG, "null pointer");
    return refresh_mirror(*Pp, params, (float*)workspace, (hipStream_t)stream);
}

extern "C" int isdqn_net_bn_commit_running(const isdqn_net_config* cfg, float* params, const void* workspace, void* stream) {
    int rc;
    const Plan* Pp = cached_plan(cfg, &rc);
    if (!Pp) return rc;
    ISDQN_REQUIRE(params != nullptr && workspace != nullptr, ISDQN_ERR_ARG, "null pointer");
    ISDQN_REQUIRE(Pp->bn, ISDQN_ERR_ARG, "not a BatchNorm network");
    return bn_commit_running(*Pp, params, (const float*)workspace, (hipStream_t)stream);
}

// DQN.learn_on_batch / loss_on_batch (dqn.py:59-83): separate target parameters for the next states.
extern "C" int isdqn_net_learn_on_batch_target(const isdqn_net_config* cfg, float* params, const float* target_params,
                                               float* adam_m, float* adam_v, int32_t* adam_count, const isdqn_batch* batch,
                                               float* losses, float* losses_accum, float* q_values, float* targets,
                                               double* priorities, void* workspace, void* stream) {
    ISDQN_REQUIRE(target_params != nullptr, ISDQN_ERR_ARG, "null target_params");
    return learn_or_loss(cfg, params, adam_m, adam_v, adam_count, batch, losses, losses_accum, q_values, targets,
                         priorities, workspace, stream, true, nullptr, target_params);
}

extern "C" int isdqn_net_loss_on_batch_target(const isdqn_net_config* cfg, const float* params, const float* target_params,
                                              const isdqn_batch* batch, float* losses, float* q_values, float* targets,
                                              void* workspace, void* stream) {
    ISDQN_REQUIRE(target_params != nullptr, ISDQN_ERR_ARG, "null target_params");
    return learn_or_loss(cfg, const_cast<float*>(params), nullptr, nullptr, nullptr, batch, losses, nullptr, q_values,
                         targets, nullptr, workspace, stream, false, nullptr, target_params);
}

extern "C" int isdqn_net_shift_params(const isdqn_net_config* cfg, float* params, void* stream) {
    Plan P;
    int rc = build_plan(cfg, P);
    if (rc) return rc;
    ISDQN_REQUIRE(params != nullptr, ISDQN_ERR_ARG, "null pointer");
    const Layer& l = P.L[P.n_layers - 1];
    hipLaunchKernelGGL(shift_kernel, dim3(ceil_div(l.in_p + 1, 256)), dim3(256), 0, (hipStream_t)stream,
                       params + l.w_off, params + l.b_off, P.nha, P.n_actions, l.in_p);
    ISDQN_HIP_CHECK(hipGetLastError());
    return ISDQN_OK;
}

extern "C" int isdqn_net_best_action(const isdqn_net_config* cfg, const float* params, const uint8_t* frames,
                                     int64_t frame_stride, const int32_t* frame_ids, const float* obs,
                                     int32_t idx_network, int32_t* out_action, void* workspace, void* stream) {
    int rc;
    const Plan* Pp = cached_plan(cfg, &rc);
    if (!Pp) return rc;
    const Plan& P = *Pp;
    ISDQN_REQUIRE(params && out_action && workspace, ISDQN_ERR_ARG, "null pointer");
    ISDQN_REQUIRE(idx_network >= 0 && idx_network < P.K, ISDQN_ERR_ARG, "idx_network out of range");
    rc = check_input(cfg, frames, frame_stride, frame_ids, obs);
    if (rc) return rc;
    NetInput in{frames, frame_stride, frame_ids, 0, obs, nullptr, 0};
    float* ws = (float*)workspace;
    hipStream_t st = (hipStream_t)stream;
    rc = refresh_mirror(P, params, ws, st);
    if (rc) return rc;
    rc = net_forward(P, cfg->precision == ISDQN_PRECISION_BF16X3, params, in, 1, 0, ws, ws + P.q_off, st);
    if (rc) return rc;
    hipLaunchKernelGGL(argmax_kernel, dim3(1), dim3(64), 0, st, ws + P.q_off, P.n_actions, P.oh + idx_network, out_action);
    ISDQN_HIP_CHECK(hipGetLastError());
    return ISDQN_OK;
}

// best_action for n observations in one forward (vectorised environments: n host envs, one launch chain), with the
// "weight mirror is current" promise of ISDQN_BATCH_MIRROR_CURRENT as a flag.
extern "C" int isdqn_net_best_actions(const isdqn_net_config* cfg, const float* params, const uint8_t* frames,
                                      int64_t frame_stride, const int32_t* frame_ids, const float* obs, int32_t n_rows,
                                      const int32_t* idx_networks, int32_t* out_actions, int32_t flags, void* workspace,
                                      void* stream) {
    int rc;
    const Plan* Pp = cached_plan(cfg, &rc);
    if (!Pp) return rc;
    const Plan& P = *Pp;
    ISDQN_REQUIRE(params && idx_networks && out_actions && workspace, ISDQN_ERR_ARG, "null pointer");
    ISDQN_REQUIRE(n_rows >= 1 && n_rows <= P.N2, ISDQN_ERR_SHAPE, "n_rows must be in [1, 2 * batch_size] (workspace rows)");
    rc = check_input(cfg, frames, frame_stride, frame_ids, obs);
    if (rc) return rc;
    NetInput in{frames, frame_stride, frame_ids, 0, obs, nullptr, 0};
    float* ws = (float*)workspace;
    hipStream_t st = (hipStream_t)stream;
    if (!(flags & ISDQN_BATCH_MIRROR_CURRENT)) {
        rc = refresh_mirror(P, params, ws, st);
        if (rc) return rc;
    }
    rc = net_forward(P, cfg->precision == ISDQN_PRECISION_BF16X3, params, in, n_rows, 0, ws, ws + P.q_off, st);
    if (rc) return rc;
    hipLaunchKernelGGL(argmax_rows_kernel, dim3((n_rows + 63) / 64), dim3(64), 0, st, ws + P.q_off, n_rows, P.nha_p, P.n_actions,
                       P.oh, P.K, idx_networks, out_actions);
    ISDQN_HIP_CHECK(hipGetLastError());
    return ISDQN_OK;
}

// eval_srank_and_dead_neurons' device part (experiments/base/srank_and_dead_neurons.py:8-22): the torso on n_rows observations.
extern "C" int isdqn_net_analysis_layout(const isdqn_net_config* cfg, int32_t* n_hidden, int64_t* sizes, int32_t max_sizes) {
    int rc;
    const Plan* Pp = cached_plan(cfg, &rc);
    if (!Pp) return rc;
    const Plan& P = *Pp;
    ISDQN_REQUIRE(n_hidden != nullptr, ISDQN_ERR_ARG, "null pointer");
    int n = 0;
    auto put = [&](int64_t v) {
        if (sizes != nullptr && n < max_sizes) sizes[n] = v;
        ++n;
    };
    if (P.L[0].kind == 2)  // impala: the two ReLU outputs of each residual block of each Stack (analysis_architecture.py:27-40)
        for (int s = 0; s < IMP_STACKS; ++s)
            for (int k = 0; k < 4; ++k) put((int64_t)P.imp[s].Hp * P.imp[s].Wp * P.imp[s].C);
    for (int i = 0; i < P.n_layers - 1; ++i) {
        const Layer& l = P.L[i];
        put(l.kind != 1 ? (int64_t)l.npix * l.cout : (int64_t)l.out_f);
    }
    *n_hidden = n;
    return ISDQN_OK;
}

extern "C" int isdqn_net_analysis(const isdqn_net_config* cfg, const float* params, const uint8_t* frames, int64_t frame_stride,
                                  const int32_t* frame_ids, const float* obs, int32_t n_rows, float* features_out,
                                  float* scores_out, void* workspace, void* stream) {
    int rc;
    const Plan* Pp = cached_plan(cfg, &rc);
    if (!Pp) return rc;
    const Plan& P = *Pp;
    ISDQN_REQUIRE(params && features_out && scores_out && workspace, ISDQN_ERR_ARG, "null pointer");
    ISDQN_REQUIRE(n_rows >= 1 && n_rows <= P.N2, ISDQN_ERR_SHAPE, "n_rows must be in [1, 2 * batch_size] (workspace rows)");
    ISDQN_REQUIRE(P.n_layers >= 2, ISDQN_ERR_SHAPE, "no hidden layer");
    rc = check_input(cfg, frames, frame_stride, frame_ids, obs);
    if (rc) return rc;
    NetInput in{frames, frame_stride, frame_ids, 0, obs, nullptr, 0};
    float* ws = (float*)workspace;
    hipStream_t st = (hipStream_t)stream;
    const bool x3 = cfg->precision == ISDQN_PRECISION_BF16X3;
    rc = refresh_mirror(P, params, ws, st);
    if (rc) return rc;
    // BatchNorm networks: as the reference applies AnalysisNet (srank_and_dead_neurons.py:17: mutable batch_stats, use_running_average
    // left False) -- the batch statistics of the analysed rows; the whole network runs (its head output is not used)
    if (P.bn) rc = bn_forward(P, x3, params, in, n_rows, 0, ws, ws + P.q_off, /*running=*/false, st);
    else rc = net_forward(P, x3, params, in, n_rows, 0, ws, ws + P.q_off, st, P.n_layers - 1);
    if (rc) return rc;
    int64_t off = 0;
    auto rowsum = [&](const float* act_s8, int elems_p, int cpp, int c, int64_t width, float* feat) -> int {
        hipLaunchKernelGGL(act_rowsum_kernel, dim3(ceil_div(elems_p, 256)), dim3(256), 0, st, act_s8, n_rows, elems_p, cpp, c, scores_out + off, feat,
                           feat ? (int)width : 0);
        ISDQN_HIP_CHECK(hipGetLastError());
        off += width;
        return ISDQN_OK;
    };
    if (P.L[0].kind == 2)
        for (int s = 0; s < IMP_STACKS; ++s) {
            const ImpalaStack& S = P.imp[s];
            const int elems_p = S.Hp * S.Wp * S.C_p;
            for (int b = 0; b < 2; ++b) {  // relu([LN](r)) in front of the block's BatchNorm, relu(conv) behind its first convolution
                if ((rc = rowsum(ws + S.a1_off[b], elems_p, S.C_p, S.C, (int64_t)S.Hp * S.Wp * S.C, nullptr))) return rc;
                if ((rc = rowsum(ws + S.a2_off[b], elems_p, S.C_p, S.C, (int64_t)S.Hp * S.Wp * S.C, nullptr))) return rc;
            }
        }
    for (int i = 0; i < P.n_layers - 1; ++i) {
        const Layer& l = P.L[i];
        const int cpp = l.kind != 1 ? l.cout_p : l.out_p, c = l.kind != 1 ? l.cout : l.out_f;
        const bool last = i == P.n_layers - 2;
        const int64_t width = l.kind != 1 ? (int64_t)l.npix * l.cout : (int64_t)l.out_f;
        // the sums are those of the ReLU output (in front of a BatchNorm); the feature matrix leaves the network behind the last
        // BatchNorm (analysis_architecture.py:115-122): a second pass over that site's output, sums discarded into the head's q rows
        if ((rc = rowsum(ws + l.act_off, l.out_elems_p, cpp, c, width, (last && !P.bn) ? features_out : nullptr))) return rc;
        if (last && P.bn) {
            const BnSite* b = bn_site_of(P, i);
            hipLaunchKernelGGL(act_rowsum_kernel, dim3(ceil_div(l.out_elems_p, 256)), dim3(256), 0, st, (const float*)(ws + b->out_off), n_rows,
                               l.out_elems_p, cpp, c, ws + P.slab_off, features_out, (int)width);
            ISDQN_HIP_CHECK(hipGetLastError());
        }
    }
    return ISDQN_OK;
}

extern "C" int isdqn_selftest_gemm(const float* A, const float* B, float* C, int32_t M, int32_t N, int32_t K,
                                   int32_t a_tr, int32_t b_tr, int32_t precision, int32_t split_k, void* stream) {
    ISDQN_REQUIRE(A && B && C, ISDQN_ERR_ARG, "null pointer");
    ISDQN_REQUIRE(M > 0 && N > 0 && K > 0, ISDQN_ERR_SHAPE, "bad shape");
    const bool x3 = precision == ISDQN_PRECISION_BF16X3;
    hipStream_t st = (hipStream_t)stream;
    // ROW operand: [rows][K] ; TR operand: [K][rows].  No alignment promises (self-test shapes are arbitrary).
    MatSrc a = a_tr ? MatSrc{A, M, K, M, 0} : MatSrc{A, K, M, K, 0};
    MatSrc b = b_tr ? MatSrc{B, N, K, N, 0} : MatSrc{B, K, N, K, 0};
    const int64_t slab = (int64_t)M * N;
    if (!a_tr && !b_tr) return plain_big<false, false, false>(x3, a, nullptr, 0, b, C, N, M, N, K, split_k, slab, st);
    if (!a_tr && b_tr) return plain_big<false, true, false>(x3, a, nullptr, 0, b, C, N, M, N, K, split_k, slab, st);
    if (a_tr && !b_tr) return plain_big<true, false, false>(x3, a, nullptr, 0, b, C, N, M, N, K, split_k, slab, st);
    return plain_big<true, true, false>(x3, a, nullptr, 0, b, C, N, M, N, K, split_k, slab, st);
}
