// Host-side plan of the Q-network: layer geometry, internal parameter layout and workspace map.
// Mirrors slimdqn/networks/architectures/dqn.py:47-103 (cnn / fc branches) and the
// (1+K)*A head view of slimdqn/networks/isdqn.py:34-41.
#pragma once
#include <cstdlib>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "common.h"

namespace isdqn {

constexpr int MAX_LAYERS = 12;
constexpr int LN_MAX_BLOCKS = 256;  // partial-sum slabs of the LayerNorm backward
constexpr int HC_MAX_S = 4;         // head chain kernel: most transitions per workgroup (1, 2 or 4 by batch size; online + next rows share one 16-row MFMA tile)

struct Layer {
    int kind;  // 0 conv, 1 dense
    int has_ln, has_relu, is_head;
    // conv geometry (SAME padding, dqn.py:55-69; kernel/stride fixed by the reference: 8/4, 4/2, 3/1)
    int hin, win, cin, cin_p, hout, wout, cout, cout_p, ksz, stride, pad, taps, npix;
    int is_u8;  // first conv: reads uint8 frames through the id table, K order = (plane, ky, kx)
    // dense
    int in_f, in_p;  // true / internal input width (in_p: padded to 8; after a conv torso: npix * cout_p)
    int in_unpadded_ld;  // fc first layer: the caller's obs matrix has ld = in_f
    int out_f, out_p;    // true / padded output width (conv: cout / cout_p)
    int K;               // forward contraction length (internal)
    int in_elems_p, out_elems_p;  // per-image element counts (internal)
    // parameters (float offsets into the flat buffer); -1 if absent
    int64_t w_off, b_off, g_off, be_off, w_size;
    // workspace (float offsets)
    int64_t act_off, z_off, dz_off, gw_off, part_off;
    int gw_slabs;   // split-K slabs of the weight gradient
    int fwd_splits; // dense: split-K factor of the forward GEMM
    bool fwd_narrow = false;  // dense forward on 128 x 64 tiles (S8 operands only)
    // conv: image-resident weight gradient (conv_img.h): 0 = use the generic engine
    int wgi_ntw, wgi_G, wgi_groups;
    // conv (layer >= 1): image-resident data gradient with the LayerNorm backward of the layer below fused in
    int dgi_tiles;       // workgroups per image (0 = generic engine + separate ln_bwd)
    int dgi_tile_pix;    // pixels of a class per workgroup: 128, or 64 when 128 would leave fewer than two workgroups per CU
    int64_t red_off;     // hidden layers: reduced (dgamma, dbeta, dbias) row [3][out_p]
    int part_rows;       // rows of the partial-sum region
    char name[16];
    char ln_name[16];
};

// Impala torso (architectures/dqn.py:7-36, 75-88): three Stacks.  The torso is ONE pseudo-layer (kind 2) of the layer list -- its
// output (the residual stream behind Stack_2, LayerNorm + ReLU + flatten) has the geometry of a conv layer's output, so the dense
// tail, the head chain and the fused dense data gradient treat it exactly like the cnn torso's last convolution; what is inside
// runs on the generic MFMA engine (ConvFwd / ConvDgrad / ConvWgrad problems) and a few row-wise kernels (impala.h).
constexpr int IMP_STACKS = 3, IMP_CONVS = 5;
struct ImpalaStack {
    int H, W, Hp, Wp, pool_pad;     // the first conv runs at H x W, everything behind the max-pool at Hp x Wp
    int cin, cin_p, C, C_p;
    Layer conv[IMP_CONVS];           // Conv_0 (cin -> C at H x W), Conv_1 .. Conv_4 (C -> C at Hp x Wp): geometry, parameter and slab offsets
    int64_t ln_g[2], ln_b[2];        // LayerNorm_0 / LayerNorm_1 of the two residual blocks (-1: no LayerNorm)
    // workspace (float offsets; rows of N2 images in the forward, B in the backward)
    int64_t xin_off, z0_off, arg_off, r_off[3], a1_off[2], a2_off[2], zt_off;
    int64_t dr_off, da_off, dzs_off, dz0_off, bpart_off[IMP_CONVS], lnpart_off[2];
};

// One flax.linen.BatchNorm call site (architectures/dqn.py:52-53, 59-60, 66-67, 73-74, 100-101; csrc/batchnorm.h).  The tensor it
// normalises is [rows][P][Cp] in the internal layout.  `spatial` (BatchNorm(axis=(1, 2)) on an image tensor): one statistic per pixel
// position p over the batch AND the C channels; otherwise (2-D input) one per internal column p * Cp + c over the batch.
struct BnSite {
    int layer;    // hidden layer whose post-ReLU activation it normalises (-1: the network input x / 255; -2 - (2 s + b): residual
                  // block b of impala Stack s, behind its ReLU -- dqn.py:29-30)
    int spatial;
    int P, C, Cp; // positions, channels, padded channels
    int G, G_p;   // statistic groups (spatial: P; feature: P * Cp internal columns), padded to 8
    int64_t scale_off, bias_off, mean_off, var_off;  // parameter buffer (floats): scale / bias are optimised, mean / var are the running averages
    int64_t in_off, out_off, bmean_off, bvar_off, s1_off, s2_off;  // workspace (floats): S8 input / output rows, batch statistics, backward sums
    char name[32];  // Flax module path ("BatchNorm_3", "Stack_1/BatchNorm_0")
};

struct Plan {
    ImpalaStack imp[IMP_STACKS];
    int n_layers;
    Layer L[MAX_LAYERS];
    int B, N2;
    int bn;   // cfg->batch_norm
    int Bb;   // rows the backward runs over: B (the next-state half has a zero cotangent, isdqn.py:99) -- N2 with BatchNorm, whose
              // batch statistics carry the gradient into the next-state rows
    int n_bn;
    BnSite bns[MAX_LAYERS + 1 + 2 * IMP_STACKS];
    int64_t x0_off;  // BatchNorm, cnn: the network input frames / 255 as S8 [N2][h * w][8]; fc: concat(state, next_state) fp32 [N2][obs]
    int n_heads, n_actions, nha, nha_p;
    // K regressed heads; head k + oh (online rows) is regressed on head k (next-state rows).  iS-DQN: n_heads = 1 + K,
    // oh = 1 (isdqn.py:96-98).  A single head (n_heads = 1) is TF-DQN: K = 1, oh = 0 -- the head is regressed on its own
    // stop-gradient target (tfdqn.py:68-80).
    int K, oh;
    int64_t n_params;
    // workspace (float offsets unless noted)
    int64_t q_off, dout_off, da_off, slab_off, qv_off, tg_off, dbh_off, adam_tab_off, lpart_off;
    int64_t wsplit_off;  // S8 mirror of the parameter buffer (same offsets as the fp32 master; weights only are read from it)
    int64_t slab_floats, da_floats;
    int64_t ws_bytes;
    std::vector<std::pair<std::string, std::pair<int64_t, int64_t>>> regions;  // name -> (byte offset, byte size)
};

static inline void same_padding(int size, int k, int s, int& out, int& lo) {
    out = (size + s - 1) / s;
    int total = (out - 1) * s + k - size;
    if (total < 0) total = 0;
    lo = total / 2;
}

static inline int build_plan_uncached(const isdqn_net_config* cfg, Plan& P) {
    ISDQN_REQUIRE(cfg != nullptr, ISDQN_ERR_ARG, "null config");
    ISDQN_REQUIRE(cfg->arch == ISDQN_ARCH_CNN || cfg->arch == ISDQN_ARCH_FC || cfg->arch == ISDQN_ARCH_IMPALA, ISDQN_ERR_UNSUPPORTED,
                  "architecture_type must be cnn, impala or fc");
    ISDQN_REQUIRE(cfg->n_features >= (cfg->arch == ISDQN_ARCH_FC ? 0 : 3) && cfg->n_features <= ISDQN_MAX_FEATURES,
                  ISDQN_ERR_ARG, "bad n_features");
    ISDQN_REQUIRE(cfg->n_actions >= 1 && cfg->n_heads >= 1, ISDQN_ERR_ARG, "need n_actions >= 1 and n_heads >= 1");
    ISDQN_REQUIRE(cfg->batch_size >= 1 && cfg->batch_size <= 4096, ISDQN_ERR_ARG, "batch_size must be in [1, 4096]");
    ISDQN_REQUIRE(cfg->precision == ISDQN_PRECISION_BF16X3 || cfg->precision == ISDQN_PRECISION_BF16, ISDQN_ERR_ARG,
                  "bad precision");
    ISDQN_REQUIRE(cfg->huber_delta >= 0.f, ISDQN_ERR_ARG, "huber_delta must be >= 0 (0 = squared error)");
    ISDQN_REQUIRE(cfg->batch_norm == 0 || cfg->batch_norm == 1, ISDQN_ERR_ARG, "batch_norm must be 0 or 1");
    P.regions.clear();
    P.B = cfg->batch_size;
    P.N2 = 2 * P.B;
    P.bn = cfg->batch_norm;
    P.Bb = P.bn ? P.N2 : P.B;
    P.n_bn = 0;
    P.x0_off = -1;
    P.n_heads = cfg->n_heads;
    P.K = cfg->n_heads >= 2 ? cfg->n_heads - 1 : 1;
    P.oh = cfg->n_heads >= 2 ? 1 : 0;
    P.n_actions = cfg->n_actions;
    P.nha = cfg->n_heads * cfg->n_actions;
    P.nha_p = round_up(P.nha, 8);
    int nl = 0;
    int n_conv = 0, n_dense = 0, n_ln = 0;
    int64_t poff = 0;
    int in_elems_p = 0, in_f = 0, in_p = 0;
    static const int KS[3] = {8, 4, 3}, ST[3] = {4, 2, 1};

    auto finish_params = [&](Layer& l) {
        l.w_off = poff;
        poff += l.w_size;
        l.b_off = poff;
        poff += l.out_p;
        if (l.has_ln) {
            l.g_off = poff;
            poff += l.out_p;
            l.be_off = poff;
            poff += l.out_p;
            snprintf(l.ln_name, sizeof(l.ln_name), "LayerNorm_%d", n_ln++);
        } else {
            l.g_off = l.be_off = -1;
            l.ln_name[0] = 0;
        }
    };

    // BatchNorm_i in call order (Flax auto-names per class): scale and bias join the optimised parameters here, the running
    // averages are placed behind all of them (below)
    int n_bn_top = 0;  // (Flax counts BatchNorm modules per parent: the Stacks' own ones are "Stack_s/BatchNorm_b")
    auto add_bn = [&](int layer, int spatial, int Pn, int C, int Cp, const char* name = nullptr) {
        BnSite& b = P.bns[P.n_bn];
        memset(&b, 0, sizeof(b));
        b.layer = layer; b.spatial = spatial; b.P = Pn; b.C = C; b.Cp = Cp;
        b.G = spatial ? Pn : Pn * Cp;
        b.G_p = round_up(b.G, 8);
        b.scale_off = poff; poff += b.G_p;
        b.bias_off = poff; poff += b.G_p;
        if (name) snprintf(b.name, sizeof(b.name), "%s", name);
        else snprintf(b.name, sizeof(b.name), "BatchNorm_%d", n_bn_top++);
        ++P.n_bn;
    };

    if (cfg->arch == ISDQN_ARCH_CNN) {
        ISDQN_REQUIRE(cfg->obs_h >= 8 && cfg->obs_w >= 8 && cfg->obs_c >= 1 && cfg->obs_c <= 16, ISDQN_ERR_ARG,
                      "bad observation shape");
        int h = cfg->obs_h, w = cfg->obs_w, c = cfg->obs_c, c_p = cfg->obs_c;
        if (P.bn) {  // dqn.py:52-53: the first convolution reads BatchNorm(x / 255): S8 rows of 8-padded channels, generic K order
            ISDQN_REQUIRE(cfg->obs_c <= 8, ISDQN_ERR_UNSUPPORTED, "BatchNorm: at most 8 stacked frames");
            c_p = 8;
            add_bn(-1, 1, h * w, c, c_p);
        }
        for (int i = 0; i < 3; ++i) {
            Layer& l = P.L[nl++];
            memset(&l, 0, sizeof(l));
            l.kind = 0;
            l.has_ln = cfg->layer_norm ? 1 : 0;
            l.has_relu = 1;
            l.is_u8 = (i == 0) && !P.bn;
            l.hin = h; l.win = w; l.cin = c; l.cin_p = c_p;
            l.ksz = KS[i]; l.stride = ST[i]; l.taps = KS[i] * KS[i];
            int pad_w;
            same_padding(h, l.ksz, l.stride, l.hout, l.pad);
            same_padding(w, l.ksz, l.stride, l.wout, pad_w);
            ISDQN_REQUIRE(pad_w == l.pad, ISDQN_ERR_UNSUPPORTED, "non-square padding is not supported");
            l.cout = cfg->features[i];
            ISDQN_REQUIRE(l.cout >= 1 && l.cout <= 64, ISDQN_ERR_UNSUPPORTED,
                          "conv widths above 64 channels are not supported by the fused LayerNorm epilogue");
            l.cout_p = round_up(l.cout, 8);
            l.npix = l.hout * l.wout;
            l.out_f = l.cout; l.out_p = l.cout_p;
            l.K = l.is_u8 ? l.cin * 64 : l.taps * l.cin_p;
            l.in_elems_p = l.hin * l.win * l.cin_p;
            l.out_elems_p = l.npix * l.cout_p;
            l.w_size = (int64_t)l.cout_p * l.K;
            snprintf(l.name, sizeof(l.name), "Conv_%d", n_conv++);
            finish_params(l);
            // dqn.py:59-60, 66-67: BatchNorm(axis=(1, 2)) behind the first two ReLUs; :72-74: behind the flatten (a 2-D tensor: per feature)
            if (P.bn) add_bn(i, i < 2 ? 1 : 0, l.npix, l.cout, l.cout_p);
            h = l.hout; w = l.wout; c = l.cout; c_p = l.cout_p;
        }
        in_elems_p = h * w * c_p;
        in_f = h * w * c;
        in_p = in_elems_p;
    } else if (cfg->arch == ISDQN_ARCH_IMPALA) {
        ISDQN_REQUIRE(cfg->obs_h >= 8 && cfg->obs_w >= 8 && cfg->obs_c >= 1 && cfg->obs_c <= 8, ISDQN_ERR_ARG, "bad observation shape");
        int h = cfg->obs_h, w = cfg->obs_w, c = cfg->obs_c, c_p = 8;
        if (P.bn) add_bn(-1, 1, h * w, c, c_p);  // dqn.py:78-79
        auto conv3 = [&](Layer& l, int hh, int ww, int cin, int cin_p, int cout, const char* nm) {
            memset(&l, 0, sizeof(l));
            l.kind = 0; l.has_relu = 0; l.is_u8 = 0;
            l.hin = hh; l.win = ww; l.cin = cin; l.cin_p = cin_p; l.ksz = 3; l.stride = 1; l.taps = 9;
            int pad_w;
            same_padding(hh, 3, 1, l.hout, l.pad);
            same_padding(ww, 3, 1, l.wout, pad_w);
            l.cout = cout; l.cout_p = round_up(cout, 8);
            l.npix = l.hout * l.wout;
            l.out_f = l.cout; l.out_p = l.cout_p;
            l.K = l.taps * l.cin_p;
            l.in_elems_p = hh * ww * cin_p;
            l.out_elems_p = l.npix * l.cout_p;
            l.w_size = (int64_t)l.cout_p * l.K;
            snprintf(l.name, sizeof(l.name), "%s", nm);
            l.w_off = poff; poff += l.w_size;
            l.b_off = poff; poff += l.out_p;
            l.g_off = l.be_off = -1;
        };
        for (int s = 0; s < IMP_STACKS; ++s) {
            ImpalaStack& S = P.imp[s];
            memset(&S, 0, sizeof(S));
            S.H = h; S.W = w; S.cin = c; S.cin_p = c_p;
            S.C = cfg->features[s];
            ISDQN_REQUIRE(S.C >= 1 && S.C <= 64, ISDQN_ERR_UNSUPPORTED, "impala stack widths above 64 channels are not supported");
            S.C_p = round_up(S.C, 8);
            int pw;
            same_padding(h, 3, 2, S.Hp, S.pool_pad);
            same_padding(w, 3, 2, S.Wp, pw);
            ISDQN_REQUIRE(pw == S.pool_pad, ISDQN_ERR_UNSUPPORTED, "non-square observations are not supported by the impala torso");
            // parameters in Flax's order inside a Stack: Conv_0, then per block LayerNorm_b, Conv_{1+2b}, Conv_{2+2b} (dqn.py:17-34)
            char nm[32];
            snprintf(nm, sizeof(nm), "Conv_0");
            conv3(S.conv[0], h, w, c, c_p, S.C, nm);
            for (int b = 0; b < 2; ++b) {
                if (cfg->layer_norm) {
                    S.ln_g[b] = poff; poff += S.C_p;
                    S.ln_b[b] = poff; poff += S.C_p;
                } else {
                    S.ln_g[b] = S.ln_b[b] = -1;
                }
                if (P.bn) {  // dqn.py:29-30: BatchNorm(axis=(1, 2)) behind the block's ReLU
                    snprintf(nm, sizeof(nm), "Stack_%d/BatchNorm_%d", s, b);
                    add_bn(-2 - (2 * s + b), 1, S.Hp * S.Wp, S.C, S.C_p, nm);
                }
                for (int k = 1 + 2 * b; k <= 2 + 2 * b; ++k) {
                    snprintf(nm, sizeof(nm), "Conv_%d", k);
                    conv3(S.conv[k], S.Hp, S.Wp, S.C, S.C_p, S.C, nm);
                }
            }
            h = S.Hp; w = S.Wp; c = S.C; c_p = S.C_p;
        }
        // the torso as one pseudo-layer: output geometry of a conv layer, the LayerNorm behind the last Stack as its LayerNorm
        Layer& l = P.L[nl++];
        memset(&l, 0, sizeof(l));
        l.kind = 2;
        l.has_ln = cfg->layer_norm ? 1 : 0;
        l.has_relu = 1;
        l.hin = cfg->obs_h; l.win = cfg->obs_w; l.cin = cfg->obs_c; l.cin_p = 8;
        l.hout = h; l.wout = w; l.cout = c; l.cout_p = c_p; l.npix = h * w;
        l.out_f = c; l.out_p = c_p;
        l.in_elems_p = cfg->obs_h * cfg->obs_w * 8;
        l.out_elems_p = l.npix * c_p;
        l.w_size = 0; l.w_off = -1; l.b_off = -1;
        snprintf(l.name, sizeof(l.name), "Impala");
        if (l.has_ln) {
            l.g_off = poff; poff += l.out_p;
            l.be_off = poff; poff += l.out_p;
            snprintf(l.ln_name, sizeof(l.ln_name), "LayerNorm_%d", n_ln++);
        } else {
            l.g_off = l.be_off = -1;
        }
        if (P.bn) add_bn(nl - 1, 0, l.npix, l.cout, l.cout_p);  // dqn.py:86-88: per feature behind the flatten
        in_elems_p = h * w * c_p;
        in_f = h * w * c;
        in_p = in_elems_p;
    } else {
        ISDQN_REQUIRE(cfg->obs_c >= 1, ISDQN_ERR_ARG, "bad observation dim");
        in_f = cfg->obs_c;
        in_p = round_up(in_f, 8);
        in_elems_p = in_p;
    }
    const int first_dense = cfg->arch == ISDQN_ARCH_FC ? 0 : 3;
    for (int i = first_dense; i <= cfg->n_features; ++i) {
        ISDQN_REQUIRE(nl < MAX_LAYERS, ISDQN_ERR_ARG, "too many layers");
        Layer& l = P.L[nl++];
        memset(&l, 0, sizeof(l));
        l.kind = 1;
        l.is_head = (i == cfg->n_features);
        l.has_ln = (!l.is_head && cfg->layer_norm) ? 1 : 0;
        l.has_relu = l.is_head ? 0 : 1;
        l.in_f = in_f; l.in_p = in_p;
        l.in_unpadded_ld = (cfg->arch == ISDQN_ARCH_FC && i == 0) ? in_f : 0;
        l.out_f = l.is_head ? P.nha : cfg->features[i];
        ISDQN_REQUIRE(l.out_f >= 1 && l.out_f <= 8192, ISDQN_ERR_ARG, "bad dense width");
        l.out_p = round_up(l.out_f, 8);
        l.K = l.in_p;
        l.in_elems_p = in_elems_p;
        l.out_elems_p = l.out_p;
        l.w_size = (int64_t)l.out_p * l.in_p;
        snprintf(l.name, sizeof(l.name), "Dense_%d", n_dense++);
        finish_params(l);
        if (P.bn && !l.is_head) add_bn(nl - 1, 0, 1, l.out_f, l.out_p);  // dqn.py:100-101
        in_f = l.out_f; in_p = l.out_p; in_elems_p = l.out_p;
    }
    P.n_layers = nl;
    for (int s = 0; s < P.n_bn; ++s) {  // batch_stats (running mean / var): behind every optimised tensor
        P.bns[s].mean_off = poff; poff += P.bns[s].G_p;
        P.bns[s].var_off = poff; poff += P.bns[s].G_p;
    }
    P.n_params = poff;

    // ---- workspace ----
    int64_t off = 0;  // floats
    auto region = [&](const std::string& name, int64_t floats) {
        int64_t o = off;
        floats = (floats + 63) / 64 * 64;  // 256-B granules
        P.regions.push_back({name, {o * 4, floats * 4}});
        off += floats;
        return o;
    };
    P.da_floats = 0;
    P.slab_floats = 0;
    for (int i = 0; i < nl; ++i) {
        Layer& l = P.L[i];
        l.act_off = l.z_off = l.dz_off = l.part_off = -1;
        l.dgi_tiles = 0;
        if (l.kind == 0 && i > 0 && l.ksz % l.stride == 0 && l.cout_p <= 64 && l.cin_p <= 64) {
            auto tiles_of = [&](int tile_pix) {
                int tiles = 0;
                for (int c = 0; c < l.stride * l.stride; ++c) {
                    int cy = c / l.stride, cx = c % l.stride;
                    int Ha = (l.hin - cy + l.stride - 1) / l.stride, Wb = (l.win - cx + l.stride - 1) / l.stride;
                    tiles += ceil_div(Ha * Wb, tile_pix);
                }
                return tiles;
            };
            // One four-wave workgroup per CU cannot hide its own latencies (round 4, stamps_dgrad.txt: a K step of 384 MFMA
            // cycles takes 1 080), and half tiles make the conv2 data gradient of c2 7 us shorter (26.2 -> 18.9) -- but they fill
            // the image and read the weight fragments twice, and in the step that CU time is taken from the kernels of the other
            // queue (ab_tile64_c2.txt: the step is 3.7 % SLOWER at B = 256; at B = 32, where the chip is
            // mostly empty either way, +0.2 %: noise).  Off in the product; -DISDQN_DGRAD_TILE64_BELOW=<workgroups> builds them.
#if !defined(ISDQN_DGRAD_TILE64_BELOW)
#define ISDQN_DGRAD_TILE64_BELOW 0
#endif
            l.dgi_tile_pix = P.Bb * tiles_of(128) < ISDQN_DGRAD_TILE64_BELOW ? 64 : 128;
            l.dgi_tiles = tiles_of(l.dgi_tile_pix);
        }
        if (!l.is_head) {
            l.act_off = region(std::string("act/") + l.name, (int64_t)P.N2 * l.out_elems_p);
            l.z_off = region(std::string("z/") + l.name, (int64_t)P.Bb * l.out_elems_p);
            l.dz_off = region(std::string("dz/") + l.name, (int64_t)P.Bb * l.out_elems_p);
            l.part_rows = LN_MAX_BLOCKS;
            if (l.kind == 1 && P.B > l.part_rows) l.part_rows = P.B;  // head chain: one row per workgroup, as few as one transition each
            l.part_off = -2;  // sized below, once the consumer layer's tiling is known
            l.red_off = region(std::string("red/") + l.name, 3 * (int64_t)l.out_p);
        }
        // (BatchNorm: the first convolution's data gradient is needed too -- the input BatchNorm's scale / bias gradient)
        if ((i > 0 || (P.bn && l.kind == 0)) && (int64_t)P.Bb * l.in_elems_p > P.da_floats) P.da_floats = (int64_t)P.Bb * l.in_elems_p;
        // weight-gradient slabs: split the contraction (online pixels / batch rows) over workgroups
        if (l.kind == 0) {
            int tiles_n = ceil_div(l.K, 64);
            int ksteps = ceil_div(P.Bb * l.npix, 32);
            int s = 256 / tiles_n;
            if (s < 1) s = 1;
            if (s > ksteps) s = ksteps;
            l.gw_slabs = s;
            // image-resident variant: column groups of 64*NTW k' columns x groups of G images ~ 256 workgroups
            l.wgi_ntw = 0;
            if (l.K % 64 == 0 && (l.is_u8 || l.cin_p % 16 == 0)) {
                int ntw = (l.K % 256 == 0) ? 4 : (l.K % 192 == 0) ? 3 : (l.K % 128 == 0) ? 2 : 0;
                if (l.K == 512) ntw = 2;
                // 64 output channels and K <= 576: one workgroup owns every k' column (8 or 9 column tiles per
                // wave), so each image pair is staged into LDS exactly once instead of once per column group
                const bool wide = l.cout_p == 64 && !l.is_u8 && (l.K == 512 || l.K == 576);
                if (wide) ntw = l.K / 64;
                if (ntw) {
                    int ncg = l.K / (64 * ntw);
                    int G = (P.Bb * ncg + 255) / 256;
                    // two images per workgroup at the headline batch (measured best, 1: 2681, 2: 2806 steps/s: 128 workgroups share
                    // the CUs with the data-gradient chain); larger batches keep at least one workgroup per CU instead of 128
                    // workgroups of 8 images (c5: conv2 / conv1 weight gradients 74 / 66 us on half the chip)
                    if (wide) G = (P.Bb + 127) / 128;
#if !defined(ISDQN_WGRAD_G128)
                    if (wide && G > 2) G = std::max(2, (P.Bb + 255) / 256);
#endif
                    if (G < 1) G = 1;
                    l.wgi_ntw = ntw;
                    l.wgi_G = G;
                    l.wgi_groups = ceil_div(P.Bb, G);
                    if (l.wgi_groups > l.gw_slabs) l.gw_slabs = l.wgi_groups;
                }
            }
        } else if (l.kind == 1) {
            int tiles = ceil_div(l.out_p, 128) * ceil_div(l.in_p, 128);
            int ksteps = ceil_div(P.Bb, 32);
            int s = tiles >= 128 ? 1 : 256 / tiles;
            if (s > ksteps) s = ksteps;
            if (s < 1) s = 1;
            l.gw_slabs = s;
            // forward split-K slabs
            // ~512 workgroups (two per CU hide each other's operand latency).  A GEMM of few 128 x 128 tiles gets there with
            // 128 x 64 tiles and half the split-K slabs: same time at the headline size, 32 MB less slab traffic per step.
            int ft = ceil_div(P.N2, 128) * ceil_div(l.out_p, 128);
            l.fwd_narrow = ft <= 64 && !l.in_unpadded_ld && l.out_p % 64 == 0;
            if (l.fwd_narrow) ft = ceil_div(P.N2, 128) * ceil_div(l.out_p, 64);
            int fs = ft >= 256 ? 1 : 512 / ft;
            int fk = ceil_div(l.K, 32);
#if defined(ISDQN_DEV)
            if (const char* e = getenv("ISDQN_FWD_SPLITS")) fs = atoi(e);  // development: split-K factor of the forward GEMM
#endif
            if (fs > fk) fs = fk;
            if (fs < 1) fs = 1;
            l.fwd_splits = fs;
            int64_t need = (int64_t)fs * P.N2 * l.out_p;
            if (need > P.slab_floats) P.slab_floats = need;
        }
        l.gw_off = region(std::string("gw/") + l.name, (int64_t)l.gw_slabs * l.w_size);
    }
    for (int i = 0; i < nl; ++i) {
        Layer& l = P.L[i];
        if (l.is_head) continue;
        if (i + 1 < nl && P.L[i + 1].dgi_tiles > 0 && P.B * P.L[i + 1].dgi_tiles > l.part_rows)
            l.part_rows = P.B * P.L[i + 1].dgi_tiles;
        // conv layer under the first dense layer: the fused dense data-gradient + LN backward emits one partial
        // row per (64-sample tile, pixel)
        if (l.kind != 1 && i + 1 < nl && P.L[i + 1].kind == 1 && l.cout_p == 64 && ceil_div(P.B, 64) * l.npix > l.part_rows)
            l.part_rows = ceil_div(P.B, 64) * l.npix;
        l.part_off = region(std::string("part/") + l.name, (int64_t)l.part_rows * 3 * l.out_p);
    }
    if (cfg->arch == ISDQN_ARCH_IMPALA) {
        for (int s = 0; s < IMP_STACKS; ++s) {
            ImpalaStack& S = P.imp[s];
            const std::string pre = "imp/s" + std::to_string(s) + "/";
            const int64_t big = (int64_t)S.H * S.W, small = (int64_t)S.Hp * S.Wp;
            S.xin_off = region(pre + "xin", P.N2 * big * S.cin_p);
            S.z0_off = region(pre + "z0", P.N2 * big * S.C_p);
            S.arg_off = region(pre + "argmax", (P.N2 * small * S.C_p + 3) / 4);
            for (int k = 0; k < 3; ++k) S.r_off[k] = region(pre + "r" + std::to_string(k), P.N2 * small * S.C_p);
            for (int b = 0; b < 2; ++b) {
                S.a1_off[b] = region(pre + "a1_" + std::to_string(b), P.N2 * small * S.C_p);
                S.a2_off[b] = region(pre + "a2_" + std::to_string(b), P.N2 * small * S.C_p);
                S.lnpart_off[b] = region(pre + "lnpart" + std::to_string(b), (int64_t)LN_MAX_BLOCKS * 2 * S.C_p);
            }
            S.zt_off = region(pre + "zt", P.N2 * small * S.C_p);
            S.dr_off = region(pre + "dr", P.Bb * small * S.C_p);
            S.da_off = region(pre + "da", P.Bb * std::max(big * S.cin_p, small * S.C_p));
            S.dzs_off = region(pre + "dzs", P.Bb * big * S.C_p);
            S.dz0_off = region(pre + "dz0", P.Bb * big * S.C_p);
            for (int k = 0; k < IMP_CONVS; ++k) {
                Layer& c = S.conv[k];
                S.bpart_off[k] = region(pre + "bpart" + std::to_string(k), (int64_t)LN_MAX_BLOCKS * S.C_p);
                int tiles_n = ceil_div(c.K, 64);
                int ksteps = ceil_div(P.Bb * c.npix, 32);
                int sp = 256 / tiles_n;
                if (sp < 1) sp = 1;
                if (sp > ksteps) sp = ksteps;
                c.gw_slabs = sp;
                c.gw_off = region(pre + "gw" + std::to_string(k), (int64_t)c.gw_slabs * c.w_size);
            }
        }
    }
    for (int s = 0; s < P.n_bn; ++s) {
        BnSite& b = P.bns[s];
        const std::string pre = std::string("bn/") + b.name + "/";
        const int64_t width = (int64_t)b.P * b.Cp;
        if (b.layer == -1) {
            P.x0_off = region("bn/x0", (int64_t)P.N2 * width);
            b.in_off = P.x0_off;
        } else if (b.layer <= -2) {  // impala block: the S8 output of its [LayerNorm +] ReLU
            const int sb = -2 - b.layer;
            b.in_off = P.imp[sb / 2].a1_off[sb % 2];
        } else {
            b.in_off = P.L[b.layer].act_off;
        }
        b.out_off = region(pre + "out", (int64_t)P.N2 * width);
        b.bmean_off = region(pre + "mean", b.G_p);
        b.bvar_off = region(pre + "var", b.G_p);
        b.s1_off = region(pre + "dbias", b.G_p);
        b.s2_off = region(pre + "dscale", b.G_p);
    }
    if (P.bn && cfg->arch == ISDQN_ARCH_FC)  // concat(state, next_state) as one matrix (the first layer's weight gradient contracts over all 2B rows)
        P.x0_off = region("bn/x0", (int64_t)P.N2 * P.L[0].in_f);
    P.q_off = region("q", (int64_t)P.N2 * P.nha_p);
    P.dout_off = region("dout", (int64_t)P.Bb * P.nha_p);
    P.da_off = region("da", P.da_floats);
    P.slab_off = region("slab", P.slab_floats);
    P.qv_off = region("q_values", (int64_t)P.B * P.K);
    P.tg_off = region("targets", (int64_t)P.B * P.K);
    P.dbh_off = region("dbh", P.nha_p);
    P.adam_tab_off = region("adam_consts", 64);
    P.lpart_off = region("loss_partials", (int64_t)P.B * (P.K + P.nha_p));
    P.wsplit_off = region("wsplit", P.n_params);
    P.ws_bytes = off * 4;
    return ISDQN_OK;
}

// The plan of a configuration never changes: cache the last few (a training process uses one or two
// configurations; building the plan allocates strings/vectors, which showed up in the per-step host time).
static inline const Plan* cached_plan(const isdqn_net_config* cfg, int* rc_out) {
    struct Slot { isdqn_net_config cfg; Plan plan; bool used = false; };
    static thread_local Slot slots[4];
    static thread_local int next = 0;
    *rc_out = ISDQN_OK;
    if (cfg == nullptr) { *rc_out = ISDQN_ERR_ARG; set_last_error("null config"); return nullptr; }
    for (auto& s : slots)
        if (s.used && memcmp(&s.cfg, cfg, sizeof(*cfg)) == 0) return &s.plan;
    Slot& s = slots[next];
    next = (next + 1) % 4;
    s.used = false;
    int rc = build_plan_uncached(cfg, s.plan);
    if (rc) { *rc_out = rc; return nullptr; }
    memcpy(&s.cfg, cfg, sizeof(*cfg));
    s.used = true;
    return &s.plan;
}

static inline int build_plan(const isdqn_net_config* cfg, Plan& P) { return build_plan_uncached(cfg, P); }

}  // namespace isdqn
