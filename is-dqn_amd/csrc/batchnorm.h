// BatchNorm variants of the cnn / fc networks (slimdqn/networks/architectures/dqn.py:52-53, 59-60, 66-67, 73-74, 100-101) and
// their learn step (slimdqn/networks/isdqn.py:82-103 with `batch_norm=True`, tfdqn.py:58-80).
//
//   flax.linen.BatchNorm (0.10.2 defaults: momentum 0.99, epsilon 1e-5, fast variance var = max(0, E[x^2] - E[x]^2), scale and bias):
//     y = (x - mean) * (rsqrt(var + eps) * scale) + bias
//   * `BatchNorm(use_running_average, axis=(1, 2))` on an image tensor (N, H, W, C): the FEATURE axes are (H, W), so there is one
//     (mean, var, scale, bias) per pixel position and the statistics run over the batch AND the channels ("spatial" sites);
//   * `BatchNorm(use_running_average)` on a 2-D tensor (N, F): one per feature, statistics over the batch ("feature" sites).
//   Training (learn_on_batch / loss_on_batch, isdqn.py:95: apply_fn with mutable batch_stats): batch statistics of the 2B rows of
//   concat(state, next_state); the running averages move by ra = 0.99 ra + 0.01 batch (learn_on_batch keeps them, :87-88).
//   Acting (best_action, isdqn.py:130: use_running_average=True): the running averages.
//
// The statistics couple the two halves of the batch: the next-state rows carry no cotangent of their own (stop_gradient, :99) but
// they receive one through mean / var, so the backward runs over all 2B rows (Plan::Bb = N2) -- which is why this path shares none of
// the headline step's fusions (head chain, data gradients with the LayerNorm backward in their epilogue, B-row buffers).  It is the
// plain layer-by-layer form on ONE stream: forward kernels of the engine (conv_fwd / dense_fwd with their LayerNorm + ReLU
// epilogues), the generic data / weight gradient problems, ln_bwd, td_kernel, adam_kernel, and the BatchNorm kernels below between
// them.  Not on the headline path and not tuned; deterministic (fixed reduction orders, no atomics).
//
// Included by net_kernels.hip inside namespace isdqn, after impala.h (imp_frames_kernel).
#pragma once

constexpr float BN_EPS = 1e-5f, BN_MOMENTUM = 0.99f;

__device__ __forceinline__ void s8_load_group_f32(const float* group, float (&v)[8]) {
    float raw[8];
    load8_aligned(group, raw);
    bf16x8 hi, lo;
    s8_unpack(raw, hi, lo);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)hi[i] + (float)lo[i];
}

// ---- statistics, spatial sites: one wave per pixel position p; lanes walk the rows, fixed shuffle tree ----------------------------
// DY == false: sum / sum of squares of x -> mean[p], var[p].   DY == true: s1[p] = sum dy, s2[p] = sum dy * xhat.
template <bool DY>
__global__ __launch_bounds__(256) void bn_stats_spatial_kernel(const float* __restrict__ x_s8, const float* __restrict__ dy, int N, int P,
                                                               int C, int Cp, const float* __restrict__ mean_in, const float* __restrict__ var_in,
                                                               float* __restrict__ out1, float* __restrict__ out2) {
    const int lane = threadIdx.x & 63, p = (int)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= P) return;  // (whole waves leave: no barrier below)
    float mu = 0.f, rstd = 0.f;
    if (DY) {
        mu = mean_in[p];
        rstd = rsqrtf(var_in[p] + BN_EPS);
    }
    float a = 0.f, b = 0.f;
    for (int n = lane; n < N; n += 64) {
        const int64_t base = ((int64_t)n * P + p) * Cp;
        for (int c8 = 0; c8 < Cp; c8 += 8) {
            float v[8];
            s8_load_group_f32(x_s8 + base + c8, v);
            if (DY) {
                const float4 d0 = *reinterpret_cast<const float4*>(dy + base + c8), d1 = *reinterpret_cast<const float4*>(dy + base + c8 + 4);
                const float d[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (c8 + i < C) {
                        a += d[i];
                        b += d[i] * ((v[i] - mu) * rstd);
                    }
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (c8 + i < C) {
                        a += v[i];
                        b += v[i] * v[i];
                    }
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        a += __shfl_xor(a, off);
        b += __shfl_xor(b, off);
    }
    if (lane == 0) {
        if (DY) {
            out1[p] = a;
            out2[p] = b;
        } else {
            const float inv = 1.f / ((float)N * (float)C);
            const float m = a * inv;
            out1[p] = m;
            out2[p] = fmaxf(b * inv - m * m, 0.f);
        }
    }
}

// ---- statistics, feature sites: 32 groups of 8 columns x 8 row slices per workgroup, slices combined in slice order ---------------
template <bool DY>
__global__ __launch_bounds__(256) void bn_stats_feature_kernel(const float* __restrict__ x_s8, const float* __restrict__ dy, int N, int width,
                                                               const float* __restrict__ mean_in, const float* __restrict__ var_in,
                                                               float* __restrict__ out1, float* __restrict__ out2) {
    __shared__ float s_a[8][32][8], s_b[8][32][8];
    const int fg = threadIdx.x & 31, ns = threadIdx.x >> 5;
    const int col0 = ((int)blockIdx.x * 32 + fg) * 8;
    const bool on = col0 < width;
    float a[8] = {0, 0, 0, 0, 0, 0, 0, 0}, b[8] = {0, 0, 0, 0, 0, 0, 0, 0}, mu[8], rstd[8];
    if (DY && on) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            mu[i] = mean_in[col0 + i];
            rstd[i] = rsqrtf(var_in[col0 + i] + BN_EPS);
        }
    }
    if (on)
        for (int n = ns; n < N; n += 8) {
            const int64_t base = (int64_t)n * width + col0;
            float v[8];
            s8_load_group_f32(x_s8 + base, v);
            if (DY) {
                const float4 d0 = *reinterpret_cast<const float4*>(dy + base), d1 = *reinterpret_cast<const float4*>(dy + base + 4);
                const float d[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    a[i] += d[i];
                    b[i] += d[i] * ((v[i] - mu[i]) * rstd[i]);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    a[i] += v[i];
                    b[i] += v[i] * v[i];
                }
            }
        }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        s_a[ns][fg][i] = a[i];
        s_b[ns][fg][i] = b[i];
    }
    __syncthreads();
    // 256 threads = 32 groups x 8 columns: each sums the 8 slices of one column in slice order
    const int g2 = threadIdx.x >> 3, i2 = threadIdx.x & 7;
    const int col = ((int)blockIdx.x * 32 + g2) * 8 + i2;
    if (col >= width) return;
    float sa = 0.f, sb = 0.f;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        sa += s_a[s][g2][i2];
        sb += s_b[s][g2][i2];
    }
    if (DY) {
        out1[col] = sa;
        out2[col] = sb;
    } else {
        const float inv = 1.f / (float)N;
        const float m = sa * inv;
        out1[col] = m;
        out2[col] = fmaxf(sb * inv - m * m, 0.f);
    }
}

// ---- y = (x - mean) * (rsqrt(var + eps) * scale) + bias, S8 in, S8 out; one thread per group of 8 columns -------------------------
template <bool SPATIAL>
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x_s8, int64_t n_groups, int P, int C, int Cp,
                                                       const float* __restrict__ mean, const float* __restrict__ var,
                                                       const float* __restrict__ scale, const float* __restrict__ bias, float* __restrict__ out_s8) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= n_groups) return;
    const int64_t e0 = g * 8;
    const int width = P * Cp;
    const int col0 = (int)(e0 % width);  // p * Cp + c8
    const int c8 = col0 % Cp;
    float v[8], y[8];
    s8_load_group_f32(x_s8 + e0, v);
    if (SPATIAL) {
        const int p = col0 / Cp;
        const float mu = mean[p], mul = rsqrtf(var[p] + BN_EPS) * scale[p], be = bias[p];
#pragma unroll
        for (int i = 0; i < 8; ++i) y[i] = c8 + i < C ? (v[i] - mu) * mul + be : 0.f;
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int col = col0 + i;
            y[i] = c8 + i < C ? (v[i] - mean[col]) * (rsqrtf(var[col] + BN_EPS) * scale[col]) + bias[col] : 0.f;
        }
    }
    s8_store_group(out_s8 + e0, y);
}

// ---- backward: dx = scale * rstd * (dy - s1 / M - xhat * s2 / M), in place on the fp32 gradient rows ------------------------------
template <bool SPATIAL>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(float* __restrict__ dy, const float* __restrict__ x_s8, int64_t n_groups, int P, int C,
                                                           int Cp, const float* __restrict__ mean, const float* __restrict__ var,
                                                           const float* __restrict__ scale, const float* __restrict__ s1,
                                                           const float* __restrict__ s2, float inv_m) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= n_groups) return;
    const int64_t e0 = g * 8;
    const int width = P * Cp;
    const int col0 = (int)(e0 % width);
    const int c8 = col0 % Cp;
    float v[8], o[8];
    s8_load_group_f32(x_s8 + e0, v);
    const float4 d0 = *reinterpret_cast<const float4*>(dy + e0), d1 = *reinterpret_cast<const float4*>(dy + e0 + 4);
    const float d[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int gi = SPATIAL ? col0 / Cp : col0 + i;
        const float rstd = rsqrtf(var[gi] + BN_EPS);
        const float xh = (v[i] - mean[gi]) * rstd;
        o[i] = c8 + i < C ? scale[gi] * rstd * (d[i] - s1[gi] * inv_m - xh * (s2[gi] * inv_m)) : 0.f;
    }
    *reinterpret_cast<float4*>(dy + e0) = float4{o[0], o[1], o[2], o[3]};
    *reinterpret_cast<float4*>(dy + e0 + 4) = float4{o[4], o[5], o[6], o[7]};
}

// ra = momentum * ra + (1 - momentum) * batch  (flax BatchNorm, not while initialising)
__global__ __launch_bounds__(256) void bn_running_kernel(float* __restrict__ ra_mean, float* __restrict__ ra_var, const float* __restrict__ bmean,
                                                         const float* __restrict__ bvar, int G) {
    const int i = (int)blockIdx.x * 256 + threadIdx.x;
    if (i >= G) return;
    ra_mean[i] = BN_MOMENTUM * ra_mean[i] + (1.f - BN_MOMENTUM) * bmean[i];
    ra_var[i] = BN_MOMENTUM * ra_var[i] + (1.f - BN_MOMENTUM) * bvar[i];
}

// rows [r0, r1) of a [rows][ld] fp32 matrix <- 0
__global__ __launch_bounds__(256) void bn_zero_kernel(float* __restrict__ p, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = 0.f;
}

// -------------------------------------------------------------------------------------------------------------------------------------
static inline bool cfg_is_cnn(const Plan& P) { return P.L[0].kind == 0; }
static inline const BnSite* bn_site_of(const Plan& P, int layer) {
    for (int s = 0; s < P.n_bn; ++s)
        if (P.bns[s].layer == layer) return &P.bns[s];
    return nullptr;
}

// forward of one site on `rows` rows: batch statistics (into the workspace) or the running averages of `params`
static int bn_site_forward(const BnSite& b, const float* params, float* ws, int rows, bool running, hipStream_t st) {
    const float* x = ws + b.in_off;
    const int width = b.P * b.Cp;
    const float *mean = params + b.mean_off, *var = params + b.var_off;
    if (!running) {
        if (b.spatial)
            hipLaunchKernelGGL(bn_stats_spatial_kernel<false>, dim3(ceil_div(b.P, 4)), dim3(256), 0, st, x, (const float*)nullptr, rows, b.P, b.C, b.Cp,
                               (const float*)nullptr, (const float*)nullptr, ws + b.bmean_off, ws + b.bvar_off);
        else
            hipLaunchKernelGGL(bn_stats_feature_kernel<false>, dim3(ceil_div(width, 256)), dim3(256), 0, st, x, (const float*)nullptr, rows, width,
                               (const float*)nullptr, (const float*)nullptr, ws + b.bmean_off, ws + b.bvar_off);
        ISDQN_HIP_CHECK(hipGetLastError());
        mean = ws + b.bmean_off;
        var = ws + b.bvar_off;
    }
    const int64_t n_groups = (int64_t)rows * width / 8;
    const unsigned nb = (unsigned)((n_groups + 255) / 256);
    if (b.spatial)
        hipLaunchKernelGGL(bn_apply_kernel<true>, dim3(nb), dim3(256), 0, st, x, n_groups, b.P, b.C, b.Cp, mean, var, params + b.scale_off,
                           params + b.bias_off, ws + b.out_off);
    else
        hipLaunchKernelGGL(bn_apply_kernel<false>, dim3(nb), dim3(256), 0, st, x, n_groups, b.P, b.C, b.Cp, mean, var, params + b.scale_off,
                           params + b.bias_off, ws + b.out_off);
    ISDQN_HIP_CHECK(hipGetLastError());
    return ISDQN_OK;
}

// backward of one site on `rows` rows: `dy` (gradient w.r.t. the site's output, fp32 [rows][P * Cp]) becomes the gradient w.r.t. its
// input in place (apply == false: only the sums, i.e. the scale / bias gradients -- the input site has nothing below it)
static int bn_site_backward(const BnSite& b, const float* params, float* ws, float* dy, int rows, bool apply, hipStream_t st) {
    const float* x = ws + b.in_off;
    const int width = b.P * b.Cp;
    if (b.spatial)
        hipLaunchKernelGGL(bn_stats_spatial_kernel<true>, dim3(ceil_div(b.P, 4)), dim3(256), 0, st, x, (const float*)dy, rows, b.P, b.C, b.Cp,
                           (const float*)(ws + b.bmean_off), (const float*)(ws + b.bvar_off), ws + b.s1_off, ws + b.s2_off);
    else
        hipLaunchKernelGGL(bn_stats_feature_kernel<true>, dim3(ceil_div(width, 256)), dim3(256), 0, st, x, (const float*)dy, rows, width,
                           (const float*)(ws + b.bmean_off), (const float*)(ws + b.bvar_off), ws + b.s1_off, ws + b.s2_off);
    ISDQN_HIP_CHECK(hipGetLastError());
    if (!apply) return ISDQN_OK;
    const int64_t n_groups = (int64_t)rows * width / 8;
    const unsigned nb = (unsigned)((n_groups + 255) / 256);
    const float inv_m = 1.f / (b.spatial ? (float)rows * (float)b.C : (float)rows);
    if (b.spatial)
        hipLaunchKernelGGL(bn_bwd_apply_kernel<true>, dim3(nb), dim3(256), 0, st, dy, x, n_groups, b.P, b.C, b.Cp, (const float*)(ws + b.bmean_off),
                           (const float*)(ws + b.bvar_off), params + b.scale_off, (const float*)(ws + b.s1_off), (const float*)(ws + b.s2_off), inv_m);
    else
        hipLaunchKernelGGL(bn_bwd_apply_kernel<false>, dim3(nb), dim3(256), 0, st, dy, x, n_groups, b.P, b.C, b.Cp, (const float*)(ws + b.bmean_off),
                           (const float*)(ws + b.bvar_off), params + b.scale_off, (const float*)(ws + b.s1_off), (const float*)(ws + b.s2_off), inv_m);
    ISDQN_HIP_CHECK(hipGetLastError());
    return ISDQN_OK;
}

// DQNNet.__call__ with batch_norm=True on n_img rows (dqn.py:47-103).  `running`: use_running_average (acting); otherwise the batch
// statistics of these rows, left in the workspace for the backward.  z of the first z_img rows of every hidden layer is saved.
static int bn_forward(const Plan& P, bool x3, const float* params, const NetInput& in, int n_img, int z_img, float* ws, float* q_out, bool running,
                      hipStream_t st) {
    const float* wmir = ws + P.wsplit_off;
    const float* prev = nullptr;
    if (const BnSite* b0 = cfg_is_cnn(P) ? bn_site_of(P, -1) : nullptr) {  // cnn: BatchNorm(x / 255) (dqn.py:51-53)
        const Layer& l0 = P.L[0];
        const int hw = l0.hin * l0.win;
        FrameSrc fs{in.frames, in.frame_stride, in.frame_ids, l0.cin, in.paired_B, l0.hin, l0.win, in.id_pitch, in.id_off};
        hipLaunchKernelGGL(imp_frames_kernel, dim3((unsigned)(((int64_t)n_img * hw + 255) / 256)), dim3(256), 0, st, fs, n_img, hw, ws + P.x0_off);
        ISDQN_HIP_CHECK(hipGetLastError());
        if (int rc = bn_site_forward(*b0, params, ws, n_img, running, st)) return rc;
        prev = ws + b0->out_off;
    }
    for (int i = 0; i < P.n_layers; ++i) {
        const Layer& l = P.L[i];
        float* act = l.is_head ? q_out : ws + l.act_off;
        float* z = l.is_head ? nullptr : ws + l.z_off;
        int rc;
        if (l.kind == 2)  // impala torso: its own sites inside (impala.h)
            rc = impala_forward(P, x3, params, wmir, in, n_img, z_img, ws, st, running ? 2 : 1);
        else if (l.kind == 0)
            rc = conv_fwd(l, x3, params, wmir, in, prev, n_img, z_img, act, z, st);
        else
            rc = dense_fwd(l, x3, params, wmir, in, prev, n_img, l.is_head ? 0 : z_img, ws + P.slab_off, act, z, st);
        if (rc) return rc;
        prev = act;
        if (!l.is_head) {
            const BnSite* b = bn_site_of(P, i);
            ISDQN_REQUIRE(b != nullptr, ISDQN_ERR_ARG, "BatchNorm plan without a site behind a hidden layer");
            if ((rc = bn_site_forward(*b, params, ws, n_img, running, st))) return rc;
            prev = ws + b->out_off;
        }
    }
    return ISDQN_OK;
}

// learn_on_batch / loss_on_batch with BatchNorm (isdqn.py:82-103, tfdqn.py:58-80): see the header of this file
// `sel`: the heads a gradient-only pass regresses (analysisdqn.py:162-183).  `target_params` (gradient-only passes: compute_loss_tb,
// analysisdqn.py:165-173): the next states go through them in a forward of their own B rows -- on THEIR batch statistics -- and the
// states through `params` on the statistics of the B state rows; the backward then runs over those B rows only.
static int bn_learn_or_loss(const isdqn_net_config* cfg, const Plan& P, float* params, float* adam_m, float* adam_v, int32_t* adam_count,
                            const isdqn_batch* batch, float* losses, float* loss_accum, float* q_values, float* targets, double* priorities,
                            float* ws, hipStream_t st, bool learn, float* grad_out, bool update, const float* target_params, const HeadSel* sel) {
    const bool x3 = cfg->precision == ISDQN_PRECISION_BF16X3;
    const int B = P.B, N2 = P.N2, K = sel ? sel->K : P.K;
    const int on0 = sel ? sel->on0 : P.oh, tg0 = sel ? sel->tg0 : 0;
    if (sel) ISDQN_REQUIRE(sel->K >= 1 && on0 >= 0 && tg0 >= 0 && on0 + K <= P.n_heads && tg0 + K <= P.n_heads && K <= P.K, ISDQN_ERR_ARG,
                           "head selection outside the network's heads");
    int rc;
    NetInput in{batch->frames, batch->frame_stride, batch->frame_ids, B, batch->state, batch->next_state, B};
    if (target_params != nullptr) {
        const int stack = cfg->arch != ISDQN_ARCH_FC ? cfg->obs_c : 0;
        NetInput nx{batch->frames, batch->frame_stride, batch->frame_ids, 0, batch->next_state, nullptr, 0, 2 * stack, stack};
        rc = refresh_mirror(P, target_params, ws, st);
        if (rc) return rc;
        rc = bn_forward(P, x3, target_params, nx, B, 0, ws, ws + P.q_off + (int64_t)B * P.nha_p, /*running=*/false, st);
        if (rc) return rc;
        in = NetInput{batch->frames, batch->frame_stride, batch->frame_ids, 0, batch->state, nullptr, 0, 2 * stack, 0};
    } else if (cfg->arch == ISDQN_ARCH_FC) {
        // the 2B observation rows as ONE matrix: the first layer's weight gradient contracts over all of them
        const int64_t n = (int64_t)B * P.L[0].in_f;
        float* cat = ws + P.x0_off;
        ISDQN_HIP_CHECK(hipMemcpyAsync(cat, batch->state, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
        ISDQN_HIP_CHECK(hipMemcpyAsync(cat + n, batch->next_state, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
        in.obs = cat; in.obs2 = nullptr; in.obs_split = 0;
    }
    if (target_params != nullptr || !(batch->flags & ISDQN_BATCH_MIRROR_CURRENT)) {
        rc = refresh_mirror(P, params, ws, st);
        if (rc) return rc;
    }
    const float* wmir = ws + P.wsplit_off;
    const int Bb = target_params ? B : N2;  // rows of the forward the gradient flows through
    rc = bn_forward(P, x3, params, in, Bb, learn ? Bb : 0, ws, ws + P.q_off, /*running=*/false, st);
    if (rc) return rc;

    // ---- targets, loss, dL/dq (rows [0, B); the next-state rows of dL/dq are zero: stop_gradient, isdqn.py:99) ----
    float* qv = q_values ? q_values : ws + P.qv_off;
    float* tg = targets ? targets : ws + P.tg_off;
    const int n_blk = ceil_div(B, TD_ROWS);
    float* loss_part = ws + P.lpart_off;
    float* dbh_part = loss_part + (int64_t)n_blk * K;
    hipLaunchKernelGGL(td_kernel, dim3(n_blk), dim3(256), 2 * TD_ROWS * K * sizeof(float), st, ws + P.q_off, B, K, on0, tg0, P.n_actions, P.nha_p,
                       batch->action, batch->reward, batch->terminal, cfg->gamma_n, cfg->huber_delta, learn ? ws + P.dout_off : nullptr, qv, tg, priorities,
                       loss_part, dbh_part);
    ISDQN_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(ceil_div(K, 16) + ceil_div(P.nha_p, 16)), dim3(256), 0, st, loss_part, dbh_part, n_blk, B, K, P.nha_p,
                       losses, loss_accum, learn ? ws + P.dbh_off : nullptr, (learn && update) ? adam_count : nullptr, cfg->adam_b1, cfg->adam_b2,
                       ws + P.adam_tab_off);
    ISDQN_HIP_CHECK(hipGetLastError());
    if (batch->priorities_ready != nullptr) ISDQN_HIP_CHECK(hipEventRecord((hipEvent_t)batch->priorities_ready, st));
    if (!learn) return ISDQN_OK;
    if (Bb > B) {
        const int64_t n = (int64_t)(N2 - B) * P.nha_p;
        hipLaunchKernelGGL(bn_zero_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ws + P.dout_off + (int64_t)B * P.nha_p, n);
        ISDQN_HIP_CHECK(hipGetLastError());
    }

    // ---- backward over all N2 rows ----
    std::vector<AdamEntry> entries;
    auto entry = [&](int64_t p_off, int64_t size, const float* g, int n_slabs, int64_t stride) {
        AdamEntry e;
        e.p_off = p_off; e.size = size; e.g = g; e.n_slabs = n_slabs; e.slab_stride = stride; e.block_start = 0;
        entries.push_back(e);
    };
    float* da = ws + P.da_off;
    const float* dz_cur = ws + P.dout_off;
    int dz_ld = P.nha_p;
    for (int i = P.n_layers - 1; i >= 0; --i) {
        const Layer& l = P.L[i];
        const BnSite* below = (i > 0 || l.kind == 0) ? bn_site_of(P, i - 1) : nullptr;  // the site this layer reads (i == 0: the cnn's input site)
        const float* act_in = below ? ws + below->out_off : nullptr;
        if (!l.is_head) {
            // `da` holds the gradient w.r.t. this layer's BatchNorm output (left by layer i + 1's data gradient)
            const BnSite* site = bn_site_of(P, i);
            rc = bn_site_backward(*site, params, ws, da, Bb, true, st);
            if (rc) return rc;
            entry(site->scale_off, site->G_p, ws + site->s2_off, 1, 0);
            entry(site->bias_off, site->G_p, ws + site->s1_off, 1, 0);
            const int rows = l.kind != 1 ? Bb * l.npix : Bb;
            int nb = 0;
            rc = ln_bwd(l, params, da, ws + l.z_off, rows, ws + l.dz_off, ws + l.part_off, &nb, st);
            if (rc) return rc;
            if (l.has_ln) {
                entry(l.g_off, l.out_p, ws + l.part_off, nb, 3 * (int64_t)l.out_p);
                entry(l.be_off, l.out_p, ws + l.part_off + l.out_p, nb, 3 * (int64_t)l.out_p);
            }
            if (l.b_off >= 0) entry(l.b_off, l.out_p, ws + l.part_off + 2 * l.out_p, nb, 3 * (int64_t)l.out_p);  // (the impala torso has no bias of its own)
            dz_cur = ws + l.dz_off;
            dz_ld = l.out_p;
            if (l.kind == 2) {  // the impala torso: its own backward and optimizer launches over the 2B rows
                rc = impala_backward(P, cfg, x3, params, adam_m, adam_v, wmir, ws, Bb, grad_out, update, st, true);
                if (rc) return rc;
                continue;
            }
        } else {
            entry(l.b_off, l.out_p, ws + P.dbh_off, 1, 0);
        }
        // data gradient w.r.t. this layer's input (the BatchNorm output below); the first convolution's feeds the input site
        if (i > 0 || (l.kind == 0 && below != nullptr)) {
            if (l.kind == 0) {
                const bool small = l.cin_p <= 32;
                if (x3) rc = small ? launch_conv_dgrad<32, 3>(l, wmir, dz_cur, da, Bb, st) : launch_conv_dgrad<64, 3>(l, wmir, dz_cur, da, Bb, st);
                else rc = small ? launch_conv_dgrad<32, 1>(l, wmir, dz_cur, da, Bb, st) : launch_conv_dgrad<64, 1>(l, wmir, dz_cur, da, Bb, st);
            } else {
                MatSrc A{dz_cur, dz_ld, Bb, l.out_p, 1};
                MatSrc Bm{wmir + l.w_off, l.in_p, l.out_f, l.in_p, 1};
                if (l.is_head) rc = plain_big<false, true, true, false, 2>(x3, A, nullptr, 0, Bm, da, l.in_p, Bb, l.in_p, l.out_p, 1, 0, st);
                else rc = plain_big<false, true, true, false, DZ_S8 | 2>(x3, A, nullptr, 0, Bm, da, l.in_p, Bb, l.in_p, l.out_p, 1, 0, st);
            }
            if (rc) return rc;
        }
        // weight gradient -> slabs, summed inside adam_kernel
        int w_slabs;
        if (l.kind == 0) {
            int img_slabs = 0;
            rc = conv_wgrad_img(l, x3, in, act_in, dz_cur, ws + l.gw_off, Bb, st, &img_slabs);
            if (rc) return rc;
            if (img_slabs) {
                w_slabs = img_slabs;
            } else {
                rc = x3 ? launch_conv_wgrad<3, false>(l, in, act_in, dz_cur, ws + l.gw_off, Bb, st)
                        : launch_conv_wgrad<1, false>(l, in, act_in, dz_cur, ws + l.gw_off, Bb, st);
                w_slabs = conv_wgrad_slabs(l, Bb);
            }
        } else {
            MatSrc A{dz_cur, dz_ld, Bb, l.out_p, 1};
            MatSrc Bm = l.in_unpadded_ld ? MatSrc{in.obs, l.in_unpadded_ld, Bb, l.in_f, 0} : MatSrc{act_in, l.in_p, Bb, l.in_p, 1};
            w_slabs = effective_splits(Bb, l.gw_slabs);
            if (l.in_unpadded_ld)
                rc = plain_big<true, true, false, false, DZ_S8>(x3, A, nullptr, 0, Bm, ws + l.gw_off, l.in_p, l.out_p, l.in_p, Bb, l.gw_slabs, l.w_size, st);
            else if (l.is_head)
                rc = plain_big<true, true, true, false, 2>(x3, A, nullptr, 0, Bm, ws + l.gw_off, l.in_p, l.out_p, l.in_p, Bb, l.gw_slabs, l.w_size, st);
            else
                rc = plain_big<true, true, true, false, DZ_S8 | 2>(x3, A, nullptr, 0, Bm, ws + l.gw_off, l.in_p, l.out_p, l.in_p, Bb, l.gw_slabs, l.w_size, st);
        }
        if (rc) return rc;
        entry(l.w_off, l.w_size, ws + l.gw_off, w_slabs, l.w_size);
        if (i == 0 && below != nullptr) {  // scale / bias of the input site (dqn.py:52-53): sums only, nothing lies below it
            rc = bn_site_backward(*below, params, ws, da, Bb, false, st);
            if (rc) return rc;
            entry(below->scale_off, below->G_p, ws + below->s2_off, 1, 0);
            entry(below->bias_off, below->G_p, ws + below->s1_off, 1, 0);
        }
    }
    for (size_t e0 = 0; e0 < entries.size(); e0 += ADAM_MAX_ENTRIES) {
        AdamTable tab;
        tab.n = 0;
        tab.total_blocks = 0;
        for (size_t e = e0; e < entries.size() && e < e0 + ADAM_MAX_ENTRIES; ++e) {
            AdamEntry& t = tab.e[tab.n++];
            t = entries[e];
            t.block_start = tab.total_blocks;
            tab.total_blocks += (int)((t.size + 63) / 64);
        }
        hipLaunchKernelGGL(adam_kernel, dim3(tab.total_blocks), dim3(256), 0, st, tab, params, adam_m, adam_v, ws + P.adam_tab_off, cfg->learning_rate,
                           cfg->adam_b1, cfg->adam_b2, cfg->adam_eps, grad_out, ws + P.wsplit_off, update ? 1 : 0);
        ISDQN_HIP_CHECK(hipGetLastError());
    }
    if (update)  // params["batch_stats"] = batch_stats (isdqn.py:87-88): the running averages the forward's flax modules produced
        for (int s = 0; s < P.n_bn; ++s) {
            const BnSite& b = P.bns[s];
            hipLaunchKernelGGL(bn_running_kernel, dim3(ceil_div(b.G, 256)), dim3(256), 0, st, params + b.mean_off, params + b.var_off,
                               (const float*)(ws + b.bmean_off), (const float*)(ws + b.bvar_off), b.G);
            ISDQN_HIP_CHECK(hipGetLastError());
        }
    return ISDQN_OK;
}

// params["batch_stats"] = the batch_stats collection the LAST training-mode forward in this workspace returned (flax: mutable=["batch_stats"]):
// the analysis agents keep the collection of a forward that is not the learn step's own (analysisdqn.py:121-131, analysistfdqn.py:85-95).
static int bn_commit_running(const Plan& P, float* params, const float* ws, hipStream_t st) {
    for (int s = 0; s < P.n_bn; ++s) {
        const BnSite& b = P.bns[s];
        hipLaunchKernelGGL(bn_running_kernel, dim3(ceil_div(b.G, 256)), dim3(256), 0, st, params + b.mean_off, params + b.var_off, ws + b.bmean_off,
                           ws + b.bvar_off, b.G);
        ISDQN_HIP_CHECK(hipGetLastError());
    }
    return ISDQN_OK;
}
