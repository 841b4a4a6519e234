// Impala torso (slimdqn/networks/architectures/dqn.py:7-36 `Stack`, :75-88) on the generic MFMA engine.
//
//   Stack(x):  x = Conv3x3(x) ; x = max_pool(x, 3x3, stride 2, SAME)
//              twice:  r = x ; [x = LayerNorm(x)] ; x = relu(x) ; x = Conv3x3(x) ; x = relu(x) ; x = Conv3x3(x) ; x = x + r
//   torso:     x / 255 -> Stack_0 -> Stack_1 -> Stack_2 -> [LayerNorm] -> relu -> flatten
//
// Not on the headline path (no BASELINE config uses it; the reference's timing sweep launch_time.sh:13-27 does): the
// convolutions run as ConvFwd / ConvDgrad / ConvWgrad problems of the tile engine (S8 operands, split-bf16 passes, fp32
// accumulation -- the arithmetic of the cnn torso), everything between them as plain row-wise kernels on fp32 tensors
// [image][y][x][c padded to 8].  The residual stream stays fp32; S8 copies are written where an MFMA problem consumes a tensor.
// The torso is one pseudo-layer of the plan (net_plan.h): the dense tail, the head chain and the fused dense data gradient see
// the geometry of a conv layer's output.  BatchNorm (dqn.py:29-30, 78-79): the sites of batchnorm.h between the kernels below
// (`bn_mode`: 1 batch statistics, 2 running averages), the backward then over all 2B rows.
//
// Included by net_kernels.hip inside namespace isdqn, after the conv launchers and the Adam kernel.
#pragma once

// uint8 frames / 255 (dqn.py:77) -> S8 [n][H][W][8] (channels = stacked frames, padded to 8), through the frame-id table
__global__ __launch_bounds__(256) void imp_frames_kernel(const FrameSrc fs, int n_img, int hw, float* __restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)n_img * hw) return;
    const int j = (int)(idx / hw), pix = (int)(idx - (int64_t)j * hw);
    float v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        v[c] = 0.f;
        if (c < fs.stack) {
            const int id = fs.frame_id(j, c);
            if (id >= 0) v[c] = (float)fs.frames[(int64_t)id * fs.stride + pix] / 255.0f;
        }
    }
    s8_store_group(out + idx * 8, v);
}

// max_pool 3x3, stride 2, SAME (-inf padding, pad_lo = total / 2): P[n][Hp][Wp][Cp], arg = window position (ky * 3 + kx) of the
// first maximum.  One thread per (output pixel, 4 channels).
__global__ __launch_bounds__(256) void imp_pool_fwd_kernel(const float* __restrict__ z, int n_img, int H, int W, int Hp, int Wp, int Cp,
                                                           int pad, float* __restrict__ out, uint8_t* __restrict__ arg) {
    const int cq = Cp >> 2;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)n_img * Hp * Wp * cq) return;
    const int c4 = (int)(idx % cq) * 4;
    const int64_t opix = idx / cq;
    const int ox = (int)(opix % Wp), oy = (int)((opix / Wp) % Hp), j = (int)(opix / ((int64_t)Wp * Hp));
    float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int where[4] = {0, 0, 0, 0};
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = oy * 2 - pad + ky;
        if (iy < 0 || iy >= H) continue;
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = ox * 2 - pad + kx;
            if (ix < 0 || ix >= W) continue;
            const float4 v = *reinterpret_cast<const float4*>(z + (((int64_t)j * H + iy) * W + ix) * Cp + c4);
            const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (vv[e] > best[e]) { best[e] = vv[e]; where[e] = ky * 3 + kx; }
        }
    }
    *reinterpret_cast<float4*>(out + opix * Cp + c4) = float4{best[0], best[1], best[2], best[3]};
    *reinterpret_cast<uchar4*>(arg + opix * Cp + c4) = uchar4{(uint8_t)where[0], (uint8_t)where[1], (uint8_t)where[2], (uint8_t)where[3]};
}

// its backward, gather form (deterministic): dz[iy][ix] = sum over the <= 4 windows that contain the pixel and chose it
__global__ __launch_bounds__(256) void imp_pool_bwd_kernel(const float* __restrict__ dp, const uint8_t* __restrict__ arg, int n_img, int H, int W,
                                                           int Hp, int Wp, int Cp, int pad, float* __restrict__ dz) {
    const int cq = Cp >> 2;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)n_img * H * W * cq) return;
    const int c4 = (int)(idx % cq) * 4;
    const int64_t ipix = idx / cq;
    const int ix = (int)(ipix % W), iy = (int)((ipix / W) % H), j = (int)(ipix / ((int64_t)W * H));
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int ky = 0; ky < 3; ++ky) {
        const int ty = iy + pad - ky;  // = 2 * oy
        if (ty < 0 || (ty & 1) || (ty >> 1) >= Hp) continue;
        for (int kx = 0; kx < 3; ++kx) {
            const int tx = ix + pad - kx;
            if (tx < 0 || (tx & 1) || (tx >> 1) >= Wp) continue;
            const int64_t opix = ((int64_t)j * Hp + (ty >> 1)) * Wp + (tx >> 1);
            const uchar4 a = *reinterpret_cast<const uchar4*>(arg + opix * Cp + c4);
            const float4 g = *reinterpret_cast<const float4*>(dp + opix * Cp + c4);
            const int me = ky * 3 + kx;
            acc[0] += a.x == me ? g.x : 0.f;
            acc[1] += a.y == me ? g.y : 0.f;
            acc[2] += a.z == me ? g.z : 0.f;
            acc[3] += a.w == me ? g.w : 0.f;
        }
    }
    *reinterpret_cast<float4*>(dz + ipix * Cp + c4) = float4{acc[0], acc[1], acc[2], acc[3]};
}

// Row-wise kernels on [rows][Cp] (Cp <= 64): 16 lanes per row, 4 channels per lane, 16 rows per workgroup, grid-stride over rows.
constexpr int IMP_TPR = 16, IMP_RPB = 256 / IMP_TPR;

// a = relu([LayerNorm](x))  (dqn.py:26-28, 83-85), S8 output for the convolution / dense layer that consumes it
__global__ __launch_bounds__(256) void imp_lnrelu_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, int64_t rows, int C, int Cp,
                                                             float* __restrict__ out_s8) {
    const int tid = threadIdx.x, grp = tid / IMP_TPR, sub = tid % IMP_TPR, ch0 = sub * 4;
    const bool lane_on = ch0 < Cp;
    float ga[4], be[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const bool ok = lane_on && ch0 + r < C && gamma != nullptr;
        ga[r] = ok ? gamma[ch0 + r] : 1.f;
        be[r] = ok ? beta[ch0 + r] : 0.f;
    }
    const float inv_c = 1.f / (float)C;
    for (int64_t row0 = (int64_t)blockIdx.x * IMP_RPB; row0 < rows; row0 += (int64_t)gridDim.x * IMP_RPB) {
        const int64_t row = row0 + grp;
        const bool on = lane_on && row < rows;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (on) {
            const float4 a = *reinterpret_cast<const float4*>(x + row * Cp + ch0);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
        }
        float mean = 0.f, rstd = 1.f;
        if (gamma != nullptr) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool ok = ch0 + r < C;
                s1 += ok ? v[r] : 0.f;
                s2 += ok ? v[r] * v[r] : 0.f;
            }
#pragma unroll
            for (int off = IMP_TPR / 2; off > 0; off >>= 1) {
                s1 += __shfl_xor(s1, off);
                s2 += __shfl_xor(s2, off);
            }
            mean = s1 * inv_c;
            rstd = rsqrtf(fmaxf(s2 * inv_c - mean * mean, 0.f) + 1e-6f);
        }
        float y[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float t = gamma != nullptr ? (v[r] - mean) * (rstd * ga[r]) + be[r] : v[r];
            y[r] = (ch0 + r < C) ? fmaxf(t, 0.f) : 0.f;
        }
        if (on) s8_store_quad(out_s8 + row * Cp, ch0, y[0], y[1], y[2], y[3]);
    }
}

// backward of the same: dx_out = dx_base + d relu([LN](x)) / dx applied to da; per-workgroup partials of (dgamma, dbeta) to
// part[block][2][Cp].  dx_base may alias dx_out (the residual stream gradient is updated in place).
__global__ __launch_bounds__(256) void imp_lnrelu_bwd_kernel(const float* __restrict__ da, const float* __restrict__ x,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta, int64_t rows,
                                                             int C, int Cp, const float* dx_base, float* dx_out, float* __restrict__ part) {
    __shared__ float s_part[IMP_RPB][2][4 * IMP_TPR];
    const int tid = threadIdx.x, grp = tid / IMP_TPR, sub = tid % IMP_TPR, ch0 = sub * 4;
    const bool lane_on = ch0 < Cp;
    float ga[4], be[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const bool ok = lane_on && ch0 + r < C && gamma != nullptr;
        ga[r] = ok ? gamma[ch0 + r] : 1.f;
        be[r] = ok ? beta[ch0 + r] : 0.f;
    }
    float dg[4] = {0, 0, 0, 0}, db[4] = {0, 0, 0, 0};
    const float inv_c = 1.f / (float)C;
    for (int64_t row0 = (int64_t)blockIdx.x * IMP_RPB; row0 < rows; row0 += (int64_t)gridDim.x * IMP_RPB) {
        const int64_t row = row0 + grp;
        const bool on = lane_on && row < rows;
        float zv[4] = {0, 0, 0, 0}, dv[4] = {0, 0, 0, 0}, base[4] = {0, 0, 0, 0};
        if (on) {
            const float4 a = *reinterpret_cast<const float4*>(x + row * Cp + ch0);
            const float4 b = *reinterpret_cast<const float4*>(da + row * Cp + ch0);
            const float4 c = *reinterpret_cast<const float4*>(dx_base + row * Cp + ch0);
            zv[0] = a.x; zv[1] = a.y; zv[2] = a.z; zv[3] = a.w;
            dv[0] = b.x; dv[1] = b.y; dv[2] = b.z; dv[3] = b.w;
            base[0] = c.x; base[1] = c.y; base[2] = c.z; base[3] = c.w;
        }
        float out[4];
        if (gamma != nullptr) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool ok = ch0 + r < C;
                s1 += ok ? zv[r] : 0.f;
                s2 += ok ? zv[r] * zv[r] : 0.f;
            }
#pragma unroll
            for (int off = IMP_TPR / 2; off > 0; off >>= 1) {
                s1 += __shfl_xor(s1, off);
                s2 += __shfl_xor(s2, off);
            }
            const float mean = s1 * inv_c;
            const float rstd = rsqrtf(fmaxf(s2 * inv_c - mean * mean, 0.f) + 1e-6f);
            float xh[4], gg[4], m1 = 0.f, m2 = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool ok = on && ch0 + r < C;
                xh[r] = (zv[r] - mean) * rstd;
                const float y = xh[r] * ga[r] + be[r];
                const float dy = (ok && y > 0.f) ? dv[r] : 0.f;
                dg[r] += dy * xh[r];
                db[r] += dy;
                gg[r] = dy * ga[r];
                m1 += gg[r];
                m2 += gg[r] * xh[r];
            }
#pragma unroll
            for (int off = IMP_TPR / 2; off > 0; off >>= 1) {
                m1 += __shfl_xor(m1, off);
                m2 += __shfl_xor(m2, off);
            }
            m1 *= inv_c;
            m2 *= inv_c;
#pragma unroll
            for (int r = 0; r < 4; ++r) out[r] = (on && ch0 + r < C) ? rstd * (gg[r] - m1 - xh[r] * m2) : 0.f;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) out[r] = (on && ch0 + r < C && zv[r] > 0.f) ? dv[r] : 0.f;
        }
        if (on) *reinterpret_cast<float4*>(dx_out + row * Cp + ch0) = float4{base[0] + out[0], base[1] + out[1], base[2] + out[2], base[3] + out[3]};
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        s_part[grp][0][ch0 + r] = dg[r];
        s_part[grp][1][ch0 + r] = db[r];
    }
    __syncthreads();
    for (int i = tid; i < 2 * Cp; i += 256) {
        const int which = i / Cp, c = i % Cp;
        float s = 0.f;
        for (int g2 = 0; g2 < IMP_RPB; ++g2) s += s_part[g2][which][c];
        part[((int64_t)blockIdx.x * 2 + which) * Cp + c] = s;
    }
}

// out = a + b (the residual connection, dqn.py:34), optionally also as S8 (the next Stack's first convolution reads it)
__global__ __launch_bounds__(256) void imp_add_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t n_groups,
                                                      float* __restrict__ out, float* __restrict__ out_s8) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= n_groups) return;
    float va[8], vb[8], vo[8];
    load8_aligned(a + g * 8, va);
    load8_aligned(b + g * 8, vb);
#pragma unroll
    for (int i = 0; i < 8; ++i) vo[i] = va[i] + vb[i];
    *reinterpret_cast<float4*>(out + g * 8) = float4{vo[0], vo[1], vo[2], vo[3]};
    *reinterpret_cast<float4*>(out + g * 8 + 4) = float4{vo[4], vo[5], vo[6], vo[7]};
    if (out_s8 != nullptr) s8_store_group(out_s8 + g * 8, vo);
}

// dz (S8) = d * [mask > 0] for the data- / weight-gradient problems, and per-workgroup column sums (the bias gradient of the
// convolution whose output gradient this is) to part[block][Cp].  `mask_s8`: the S8 post-ReLU activation (null: no ReLU).
__global__ __launch_bounds__(256) void imp_to_s8_kernel(const float* __restrict__ d, const float* __restrict__ mask_s8, int64_t rows, int Cp,
                                                        float* __restrict__ out_s8, float* __restrict__ part) {
    __shared__ float s_part[IMP_RPB][4 * IMP_TPR];
    const int tid = threadIdx.x, grp = tid / IMP_TPR, sub = tid % IMP_TPR, ch0 = sub * 4;
    const bool lane_on = ch0 < Cp;
    float cs[4] = {0, 0, 0, 0};
    for (int64_t row0 = (int64_t)blockIdx.x * IMP_RPB; row0 < rows; row0 += (int64_t)gridDim.x * IMP_RPB) {
        const int64_t row = row0 + grp;
        if (!(lane_on && row < rows)) continue;
        const float4 a = *reinterpret_cast<const float4*>(d + row * Cp + ch0);
        float v[4] = {a.x, a.y, a.z, a.w};
        if (mask_s8 != nullptr) {
            // hi halves of the four activations: bf16 keeps the sign and the exponent range of fp32, so hi > 0 <=> relu input > 0
            typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
            const bf16x4 h = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const char*>(mask_s8 + row * Cp + (ch0 & ~7)) + (ch0 & 4) * 2);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (float)h[r] > 0.f ? v[r] : 0.f;
        }
        s8_store_quad(out_s8 + row * Cp, ch0, v[0], v[1], v[2], v[3]);
#pragma unroll
        for (int r = 0; r < 4; ++r) cs[r] += v[r];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) s_part[grp][ch0 + r] = cs[r];
    __syncthreads();
    for (int c = tid; c < Cp; c += 256) {
        float s = 0.f;
        for (int g2 = 0; g2 < IMP_RPB; ++g2) s += s_part[g2][c];
        part[(int64_t)blockIdx.x * Cp + c] = s;
    }
}

__global__ __launch_bounds__(256) void imp_s8_to_f32_kernel(const float* __restrict__ in_s8, int64_t n_groups, float* __restrict__ out) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= n_groups) return;
    float raw[8];
    load8_aligned(in_s8 + g * 8, raw);
    bf16x8 hi, lo;
    s8_unpack(raw, hi, lo);
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)hi[i] + (float)lo[i];
    *reinterpret_cast<float4*>(out + g * 8) = float4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<float4*>(out + g * 8 + 4) = float4{v[4], v[5], v[6], v[7]};
}

// ---------------------------------------------------------------------------------------------------------------------------------
static inline int imp_blocks(int64_t rows) {
    const int64_t nb = (rows + IMP_RPB - 1) / IMP_RPB;
    return (int)(nb < LN_MAX_BLOCKS ? nb : LN_MAX_BLOCKS);
}

static int imp_conv_fwd(const Layer& c, bool x3, const float* params, const float* wmir, const float* x_s8, int n_img, float* act, float* z,
                        hipStream_t st) {
    const NetInput none{nullptr, 0, nullptr, 0, nullptr, nullptr, 0};
    const int z_img = z != nullptr ? n_img : 0;
    const bool small = c.cout_p <= 32;
    if (x3) return small ? launch_conv_fwd<32, 3, false>(c, params, wmir, none, x_s8, n_img, z_img, act, z, st)
                         : launch_conv_fwd<64, 3, false>(c, params, wmir, none, x_s8, n_img, z_img, act, z, st);
    return small ? launch_conv_fwd<32, 1, false>(c, params, wmir, none, x_s8, n_img, z_img, act, z, st)
                 : launch_conv_fwd<64, 1, false>(c, params, wmir, none, x_s8, n_img, z_img, act, z, st);
}

// frames -> relu(LN(residual stream behind Stack_2)) as the pseudo-layer's S8 activation (+ its fp32 pre-LayerNorm rows for the backward)
static int impala_forward(const Plan& P, bool x3, const float* params, const float* wmir, const NetInput& in, int n_img, int z_img,
                          float* ws, hipStream_t st, int bn_mode) {
    const Layer& L0 = P.L[0];
    for (int s = 0; s < IMP_STACKS; ++s) {
        const ImpalaStack& S = P.imp[s];
        const int64_t big = (int64_t)n_img * S.H * S.W, small = (int64_t)n_img * S.Hp * S.Wp;
        const float* x_in = ws + S.xin_off;
        if (s == 0) {
            FrameSrc fs{in.frames, in.frame_stride, in.frame_ids, S.cin, in.paired_B, S.H, S.W, in.id_pitch, in.id_off};
            hipLaunchKernelGGL(imp_frames_kernel, dim3((unsigned)((big + 255) / 256)), dim3(256), 0, st, fs, n_img, S.H * S.W,
                               ws + (bn_mode ? P.x0_off : S.xin_off));
            ISDQN_HIP_CHECK(hipGetLastError());
            if (bn_mode) {  // BatchNorm(x / 255) (dqn.py:78-79)
                const BnSite* b0 = bn_site_of(P, -1);
                if (int rc0 = bn_site_forward(*b0, params, ws, n_img, bn_mode == 2, st)) return rc0;
                x_in = ws + b0->out_off;
            }
        }
        int rc = imp_conv_fwd(S.conv[0], x3, params, wmir, x_in, n_img, nullptr, ws + S.z0_off, st);
        if (rc) return rc;
        const int cq = S.C_p / 4;
        hipLaunchKernelGGL(imp_pool_fwd_kernel, dim3((unsigned)((small * cq + 255) / 256)), dim3(256), 0, st, ws + S.z0_off, n_img, S.H, S.W, S.Hp,
                           S.Wp, S.C_p, S.pool_pad, ws + S.r_off[0], reinterpret_cast<uint8_t*>(ws + S.arg_off));
        ISDQN_HIP_CHECK(hipGetLastError());
        for (int b = 0; b < 2; ++b) {
            const float* g = S.ln_g[b] >= 0 ? params + S.ln_g[b] : nullptr;
            const float* be = S.ln_b[b] >= 0 ? params + S.ln_b[b] : nullptr;
            hipLaunchKernelGGL(imp_lnrelu_fwd_kernel, dim3(imp_blocks(small)), dim3(256), 0, st, ws + S.r_off[b], g, be, small, S.C, S.C_p,
                               ws + S.a1_off[b]);
            ISDQN_HIP_CHECK(hipGetLastError());
            const float* a1 = ws + S.a1_off[b];
            if (bn_mode) {  // dqn.py:29-30
                const BnSite* bs = bn_site_of(P, -2 - (2 * s + b));
                if ((rc = bn_site_forward(*bs, params, ws, n_img, bn_mode == 2, st))) return rc;
                a1 = ws + bs->out_off;
            }
            rc = imp_conv_fwd(S.conv[1 + 2 * b], x3, params, wmir, a1, n_img, ws + S.a2_off[b], nullptr, st);  // act = relu(conv)
            if (rc) return rc;
            rc = imp_conv_fwd(S.conv[2 + 2 * b], x3, params, wmir, ws + S.a2_off[b], n_img, nullptr, ws + S.zt_off, st);
            if (rc) return rc;
            float* next_s8 = (b == 1 && s + 1 < IMP_STACKS) ? ws + P.imp[s + 1].xin_off : nullptr;
            const int64_t n_groups = small * S.C_p / 8;
            hipLaunchKernelGGL(imp_add_kernel, dim3((unsigned)((n_groups + 255) / 256)), dim3(256), 0, st, ws + S.zt_off, ws + S.r_off[b], n_groups,
                               ws + S.r_off[b + 1], next_s8);
            ISDQN_HIP_CHECK(hipGetLastError());
        }
    }
    const ImpalaStack& T = P.imp[IMP_STACKS - 1];
    const int64_t rows = (int64_t)n_img * T.Hp * T.Wp;
    hipLaunchKernelGGL(imp_lnrelu_fwd_kernel, dim3(imp_blocks(rows)), dim3(256), 0, st, ws + T.r_off[2], L0.has_ln ? params + L0.g_off : nullptr,
                       L0.has_ln ? params + L0.be_off : nullptr, rows, T.C, T.C_p, ws + L0.act_off);
    ISDQN_HIP_CHECK(hipGetLastError());
    if (z_img > 0)
        ISDQN_HIP_CHECK(hipMemcpyAsync(ws + L0.z_off, ws + T.r_off[2], (size_t)z_img * L0.out_elems_p * 4, hipMemcpyDeviceToDevice, st));
    return ISDQN_OK;
}

// backward through the torso for the first B images, from dz of the pseudo-layer (S8, gradient w.r.t. the residual stream behind
// Stack_2); every inner tensor's gradient goes straight into its own optimizer launch (slab sums inside adam_kernel)
// (`bn`: BatchNorm networks -- B is then the 2B rows of concat(state, next_state), and the first Stack's convolution gets a data
// gradient too: the input site has a scale and a bias)
static int impala_backward(const Plan& P, const isdqn_net_config* cfg, bool x3, float* params, float* adam_m, float* adam_v, const float* wmir,
                           float* ws, int B, float* grad_out, bool update, hipStream_t st, bool bn = false) {
    const Layer& L0 = P.L[0];
    std::vector<AdamEntry> entries;
    auto entry = [&](int64_t p_off, int64_t size, const float* g, int n_slabs, int64_t stride) {
        AdamEntry e;
        e.p_off = p_off; e.size = size; e.g = g; e.n_slabs = n_slabs; e.slab_stride = stride; e.block_start = 0;
        entries.push_back(e);
    };
    const NetInput none{nullptr, 0, nullptr, 0, nullptr, nullptr, 0};
    auto wgrad = [&](const Layer& c, const float* x_s8, const float* dz_s8) -> int {
        int rc = x3 ? launch_conv_wgrad<3, false>(c, none, x_s8, dz_s8, ws + c.gw_off, B, st)
                    : launch_conv_wgrad<1, false>(c, none, x_s8, dz_s8, ws + c.gw_off, B, st);
        entry(c.w_off, c.w_size, ws + c.gw_off, conv_wgrad_slabs(c, B), c.w_size);
        return rc;
    };
    auto dgrad = [&](const Layer& c, const float* dz_s8, float* da) -> int {
        const bool small = c.cin_p <= 32;
        if (x3) return small ? launch_conv_dgrad<32, 3>(c, wmir, dz_s8, da, B, st) : launch_conv_dgrad<64, 3>(c, wmir, dz_s8, da, B, st);
        return small ? launch_conv_dgrad<32, 1>(c, wmir, dz_s8, da, B, st) : launch_conv_dgrad<64, 1>(c, wmir, dz_s8, da, B, st);
    };
    auto to_s8 = [&](const float* d, const float* mask, int64_t rows, int Cp, float* out, float* part, int64_t b_off) -> int {
        const int nb = imp_blocks(rows);
        hipLaunchKernelGGL(imp_to_s8_kernel, dim3(nb), dim3(256), 0, st, d, mask, rows, Cp, out, part);
        ISDQN_HIP_CHECK(hipGetLastError());
        entry(b_off, Cp, part, nb, Cp);
        return ISDQN_OK;
    };

    // dz of the pseudo-layer -> fp32 gradient of Stack_2's residual stream
    const ImpalaStack& T = P.imp[IMP_STACKS - 1];
    {
        const int64_t n_groups = (int64_t)B * L0.out_elems_p / 8;
        hipLaunchKernelGGL(imp_s8_to_f32_kernel, dim3((unsigned)((n_groups + 255) / 256)), dim3(256), 0, st, ws + L0.dz_off, n_groups, ws + T.dr_off);
        ISDQN_HIP_CHECK(hipGetLastError());
    }
    float* dr = ws + T.dr_off;
    for (int s = IMP_STACKS - 1; s >= 0; --s) {
        const ImpalaStack& S = P.imp[s];
        const int64_t small = (int64_t)B * S.Hp * S.Wp, big = (int64_t)B * S.H * S.W;
        float* da = ws + S.da_off;
        float* dzs = ws + S.dzs_off;
        int rc;
        for (int b = 1; b >= 0; --b) {
            const Layer &c1 = S.conv[1 + 2 * b], &c2 = S.conv[2 + 2 * b];
            // stream = conv2(relu(conv1(a1))) + r : the gradient of conv2's output is the stream gradient itself
            rc = to_s8(dr, nullptr, small, S.C_p, dzs, ws + S.bpart_off[2 + 2 * b], c2.b_off);
            if (rc) return rc;
            rc = wgrad(c2, ws + S.a2_off[b], dzs);
            if (rc) return rc;
            rc = dgrad(c2, dzs, da);
            if (rc) return rc;
            rc = to_s8(da, ws + S.a2_off[b], small, S.C_p, dzs, ws + S.bpart_off[1 + 2 * b], c1.b_off);  // through relu(conv1)
            if (rc) return rc;
            const BnSite* bs = bn ? bn_site_of(P, -2 - (2 * s + b)) : nullptr;
            rc = wgrad(c1, bs ? ws + bs->out_off : ws + S.a1_off[b], dzs);
            if (rc) return rc;
            rc = dgrad(c1, dzs, da);
            if (rc) return rc;
            if (bs) {  // through the block's BatchNorm: da becomes the gradient w.r.t. relu([LN](r)) in place
                rc = bn_site_backward(*bs, params, ws, da, B, true, st);
                if (rc) return rc;
                entry(bs->scale_off, bs->G_p, ws + bs->s2_off, 1, 0);
                entry(bs->bias_off, bs->G_p, ws + bs->s1_off, 1, 0);
            }
            const float* g = S.ln_g[b] >= 0 ? params + S.ln_g[b] : nullptr;
            const float* be = S.ln_b[b] >= 0 ? params + S.ln_b[b] : nullptr;
            const int nb = imp_blocks(small);
            hipLaunchKernelGGL(imp_lnrelu_bwd_kernel, dim3(nb), dim3(256), 0, st, da, ws + S.r_off[b], g, be, small, S.C, S.C_p, dr, dr,
                               ws + S.lnpart_off[b]);
            ISDQN_HIP_CHECK(hipGetLastError());
            if (g != nullptr) {
                entry(S.ln_g[b], S.C_p, ws + S.lnpart_off[b], nb, 2 * (int64_t)S.C_p);
                entry(S.ln_b[b], S.C_p, ws + S.lnpart_off[b] + S.C_p, nb, 2 * (int64_t)S.C_p);
            }
        }
        const int cq = S.C_p / 4;
        hipLaunchKernelGGL(imp_pool_bwd_kernel, dim3((unsigned)((big * cq + 255) / 256)), dim3(256), 0, st, dr, reinterpret_cast<const uint8_t*>(ws + S.arg_off),
                           B, S.H, S.W, S.Hp, S.Wp, S.C_p, S.pool_pad, ws + S.dz0_off);
        ISDQN_HIP_CHECK(hipGetLastError());
        rc = to_s8(ws + S.dz0_off, nullptr, big, S.C_p, dzs, ws + S.bpart_off[0], S.conv[0].b_off);
        if (rc) return rc;
        const BnSite* b0 = (bn && s == 0) ? bn_site_of(P, -1) : nullptr;
        rc = wgrad(S.conv[0], b0 ? ws + b0->out_off : ws + S.xin_off, dzs);
        if (rc) return rc;
        if (s > 0 || b0) {
            rc = dgrad(S.conv[0], dzs, da);  // [B][H][W][cin_p] = the gradient of the previous Stack's residual stream
            if (rc) return rc;
            dr = da;
        }
        if (b0) {  // scale / bias of the input site (sums only)
            rc = bn_site_backward(*b0, params, ws, da, B, false, st);
            if (rc) return rc;
            entry(b0->scale_off, b0->G_p, ws + b0->s2_off, 1, 0);
            entry(b0->bias_off, b0->G_p, ws + b0->s1_off, 1, 0);
        }
    }
    // the optimizer over the torso's tensors (slab / partial sums inside the kernel), a table at a time
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
    return ISDQN_OK;
}
