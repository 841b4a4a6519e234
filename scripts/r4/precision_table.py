"""Precision plan: what each choice of MFMA pass counts per contraction does to the quantities the north
star puts a bar on -- Bellman targets / per-head losses within 1e-3 on fixed batches -- and to the gradients, on the CPU.

Every contraction of the network (three convolutions as im2col GEMMs, two dense layers) and of its backward (data gradient,
weight gradient) runs through ONE emulated GEMM whose operands are rounded the way an MFMA kernel would see them and whose products
are accumulated exactly (float64):
    f32      no rounding (reference arithmetic)
    bf16x3   a = hi + lo (bf16 each): hi*hi + lo*hi + hi*lo      -- the product default (3 passes; uint8 pixels: 2, exact)
    bf16x2w  activations / gradients hi only, weights hi + lo    -- 2 passes
    bf16x2a  activations / gradients hi + lo, weights hi only    -- 2 passes
    bf16x1   hi*hi                                               -- 1 pass
    fp16x1   fp16(a)*fp16(w)                                     -- 1 pass (would need scaled gradients in the backward)
    fp16x2w  fp16 activations, weights fp16 hi + lo              -- 2 passes
Forward modes are compared on q / targets / losses against float64; backward modes on every leaf's first-step gradient with the
forward held at bf16x3 (max |dg| / max |g| per leaf, worst leaf) and on the parameters after 3 Adam steps.

    python scripts/r4/precision_table.py [B]          (B = 32 by default; full-size network K = 9, A = 9; ~2 minutes)
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import network as net  # noqa: E402

FEATS, OBS, K, A = [32, 64, 64, 512], (84, 84, 4), 9, 9
GAMMA, LR, EPS = 0.99, 6.25e-5, 1.5e-4


def split(x, dt):
    hi = x.to(dt).to(torch.float64)
    lo = (x - hi).to(dt).to(torch.float64)
    return hi, lo


def emm(a, w, mode):
    """a [M, Kc] x w [Kc, N] with operand rounding `mode`, exact (float64) accumulation."""
    if mode == "f32":
        return a @ w
    dt = torch.bfloat16 if mode.startswith("bf16") else torch.float16
    ah, al = split(a, dt)
    wh, wl = split(w, dt)
    kind = mode[4:]
    if kind == "x3":
        return ah @ wh + al @ wh + ah @ wl
    if kind == "x2w":
        return ah @ (wh + wl)
    if kind == "x2a":
        return (ah + al) @ wh
    if kind == "x1":
        return ah @ wh
    raise ValueError(mode)


class EMM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, w, fmode, bmode):
        ctx.save_for_backward(a, w)
        ctx.bmode = bmode
        return emm(a, w, fmode)

    @staticmethod
    def backward(ctx, g):
        a, w = ctx.saved_tensors
        return emm(g, w.t(), ctx.bmode), emm(a.t(), g, ctx.bmode), None, None


def same(size, k, s):
    out = -(-size // s)
    tot = max((out - 1) * s + k - size, 0)
    return tot // 2, tot - tot // 2


def ln(z, p):
    mean = z.mean(-1, keepdim=True)
    var = ((z * z).mean(-1, keepdim=True) - mean * mean).clamp_min(0)
    return (z - mean) * torch.rsqrt(var + 1e-6) * p["scale"] + p["bias"]


def forward(P, x_u8, fmode, bmode, first_exact=True):
    x = x_u8.to(torch.float64) / 255.0
    n = x.shape[0]
    for i, (k, s) in enumerate(((8, 4), (4, 2), (3, 1))):
        lo_h, hi_h = same(x.shape[1], k, s)
        lo_w, hi_w = same(x.shape[2], k, s)
        xp = F.pad(x.permute(0, 3, 1, 2), (lo_w, hi_w, lo_h, hi_h))
        cols = F.unfold(xp, k, stride=s)  # [n, cin*k*k, L], channel-major then (ky, kx)
        L = cols.shape[-1]
        wmat = P[f"Conv_{i}"]["kernel"].permute(2, 0, 1, 3).reshape(-1, P[f"Conv_{i}"]["kernel"].shape[-1])  # (cin, ky, kx) x cout
        # the first layer's pixels are n / 255 with n an integer: the kernel feeds the integers (exact in bf16) and folds 1 / 255
        # into the epilogue, so its activation operand has no `lo` part: any "x3"/"x2w" mode degenerates to weights hi + lo
        fm = fmode
        a = cols.permute(0, 2, 1).reshape(n * L, -1)
        if i == 0 and first_exact and fmode != "f32":
            a = a * 255.0
            z = EMM.apply(a, wmat, fmode, bmode) / 255.0
        else:
            z = EMM.apply(a, wmat, fm, bmode)
        z = z + P[f"Conv_{i}"]["bias"]
        ho = -(-x.shape[1] // s)
        x = torch.relu(ln(z.reshape(n, ho, L // ho, -1), P[f"LayerNorm_{i}"]))
    h = x.reshape(n, -1)
    z = EMM.apply(h, P["Dense_0"]["kernel"], fmode, bmode) + P["Dense_0"]["bias"]
    h = torch.relu(ln(z, P["LayerNorm_3"]))
    q = EMM.apply(h, P["Dense_1"]["kernel"], fmode, bmode) + P["Dense_1"]["bias"]
    return q.reshape(n, 1 + K, A)


def loss_terms(P, batch, fmode, bmode):
    st, nx, act, rew, term = batch
    B = st.shape[0]
    q = forward(P, torch.cat((st, nx)), fmode, bmode)
    qv = q[:B, 1:, :][torch.arange(B), :, act]
    tg = (rew[:, None] + (1 - term)[:, None] * GAMMA * q[B:, :K].max(-1).values).detach()
    td = (qv - tg) ** 2
    return qv, tg, td.mean(0)


def grads(P, batch, fmode, bmode):
    leaves = [(m, n) for m in P for n in P[m]]
    req = {m: {n: P[m][n].detach().clone().requires_grad_(True) for n in P[m]} for m in P}
    _, _, per_head = loss_terms(req, batch, fmode, bmode)
    g = torch.autograd.grad(per_head.sum(), [req[m][n] for m, n in leaves])
    return {mn: t for mn, t in zip(leaves, g)}, per_head.detach()


def adam_steps(P, batch, fmode, bmode, steps=3):
    P = {m: {n: t.clone() for n, t in l.items()} for m, l in P.items()}
    mu = {m: {n: torch.zeros_like(t) for n, t in l.items()} for m, l in P.items()}
    nu = {m: {n: torch.zeros_like(t) for n, t in l.items()} for m, l in P.items()}
    for c in range(1, steps + 1):
        g, _ = grads(P, batch, fmode, bmode)
        for (m, n), gt in g.items():
            gt = gt.to(torch.float32).to(torch.float64)  # the gradient reaches Adam as float32
            mu[m][n] = 0.9 * mu[m][n] + 0.1 * gt
            nu[m][n] = 0.999 * nu[m][n] + 0.001 * gt * gt
            P[m][n] = P[m][n] - LR * (mu[m][n] / (1 - 0.9**c)) / (torch.sqrt(nu[m][n] / (1 - 0.999**c)) + EPS)
    return P


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    rng = np.random.default_rng(0)
    params = net.init_params(0, OBS, FEATS, "cnn", (1 + K) * A, True)
    for m in params:  # off the trivial initialisation of biases / LayerNorm parameters, as in the GPU parity tests
        for n in params[m]:
            if n != "kernel":
                params[m][n] = (params[m][n] + rng.normal(0, 0.1, params[m][n].shape)).astype(np.float32)
    P = {m: {n: torch.tensor(v, dtype=torch.float64) for n, v in l.items()} for m, l in params.items()}
    batch = (torch.tensor(rng.integers(0, 256, (B,) + OBS, dtype=np.uint8)), torch.tensor(rng.integers(0, 256, (B,) + OBS, dtype=np.uint8)),
             torch.tensor(rng.integers(0, A, B)), torch.tensor(rng.choice([-1.0, 0.0, 1.0], B)), torch.tensor((rng.random(B) < 0.2).astype(np.float64)))
    with torch.no_grad():
        q0, t0, l0 = loss_terms(P, batch, "f32", "f32")
    print(f"# full-size network (cnn 32/64/64/512 + LayerNorm, K = {K}, A = {A}), B = {B}, random uint8 frames; float64 reference")
    print(f"# |q| max {q0.abs().max():.3f}  |target| max {t0.abs().max():.3f}  losses {l0.min():.3f} .. {l0.max():.3f}\n")
    print("## forward modes: error of the quantities under the 1e-3 bar\n")
    print("| forward operands | MFMA passes (conv0 / rest) | max abs error q | max abs error targets | max abs error per-head losses | inside 1e-3 |")
    print("|---|---|---|---|---|---|")
    passes = {"bf16x3": "2 / 3", "bf16x2w": "2 / 2", "bf16x2a": "1 / 2", "bf16x1": "1 / 1", "fp16x1": "1 / 1", "fp16x2w": "2 / 2"}
    for mode in ("bf16x3", "bf16x2w", "bf16x2a", "fp16x2w", "fp16x1", "bf16x1"):
        with torch.no_grad():
            q, t, l = loss_terms(P, batch, mode, "f32")
        eq, et, el = float((q - q0).abs().max()), float((t - t0).abs().max()), float((l - l0).abs().max())
        print(f"| {mode} | {passes[mode]} | {eq:.1e} | {et:.1e} | {el:.1e} | {'yes' if max(eq, et, el) < 1e-3 else 'NO'} |")
    print("\n## backward modes (forward held at bf16x3): first-step gradient and 3 Adam steps against the all-f32 backward\n")
    g_ref, _ = grads(P, batch, "bf16x3", "f32")
    P_ref = adam_steps(P, batch, "bf16x3", "f32")
    with torch.no_grad():
        _, t_ref, l_ref = loss_terms(P_ref, batch, "bf16x3", "f32")
    print("| backward operands (dz, weights / activations) | MFMA passes | worst leaf: max abs dg / max abs g | whole gradient: relative L2 error | "
          "max abs parameter difference after 3 Adam steps (lr 6.25e-5) | max abs difference of the step-4 targets / losses |")
    print("|---|---|---|---|---|---|")
    for mode in ("bf16x3", "bf16x2w", "bf16x2a", "bf16x1"):
        g, _ = grads(P, batch, "bf16x3", mode)
        worst = max(float((g[k] - g_ref[k]).abs().max() / g_ref[k].abs().max()) for k in g)
        num = sum(float(((g[k] - g_ref[k]) ** 2).sum()) for k in g) ** 0.5
        den = sum(float((g_ref[k] ** 2).sum()) for k in g) ** 0.5
        P3 = adam_steps(P, batch, "bf16x3", mode)
        dp = max(float((P3[m][n] - P_ref[m][n]).abs().max()) for m in P3 for n in P3[m])
        with torch.no_grad():
            _, t3, l3 = loss_terms(P3, batch, "bf16x3", "f32")
        d4 = max(float((t3 - t_ref).abs().max()), float((l3 - l_ref).abs().max()))
        print(f"| {mode} | {mode[5] if mode != 'bf16x3' else '3 (conv0 weight gradient: 2)'} | {worst:.1e} | {num / den:.1e} | {dp:.1e} | {d4:.1e} |")


if __name__ == "__main__":
    main()
