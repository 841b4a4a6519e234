"""ORACLE (test infrastructure) -- DQNNet forward + iS-DQN loss/grad/Adam/shift, CPU.

PARITY UNPINNED for the floating-point network numerics: the arithmetic of the
reference lives in flax==0.10.2 (``nn.Conv``, ``nn.LayerNorm``, ``nn.Dense``),
jax==0.4.30 (``jax.grad``) and optax==0.2.4 (``adam``) -- pinned in the
reference's setup.cfg:14-32, absent from this image -- and the reference's
tests/test_isdqn.py holds formulas, not numbers.  This file restates the
documented behaviour of those versions at the reference's own call sites:

  * ``DQNNet.__call__`` cnn / fc branches ... slimdqn/networks/architectures/dqn.py:47-74, 89-103
      - ``x / 255.0``; ``nn.Conv`` default padding "SAME" (asymmetric: lo = total//2),
        use_bias, kernel HWIO, activations NHWC;
      - ``nn.LayerNorm`` defaults: normalise + scale/bias over the LAST axis only,
        epsilon 1e-6, fast variance  var = max(0, E[x^2] - E[x]^2),
        y = (x - mean) * (rsqrt(var + eps) * scale) + bias;
      - flatten order (h, w, c); ``nn.Dense`` kernel (in, out);
      - module auto-names Conv_i / LayerNorm_i / Dense_i counted per class.
  * initialisers .............................. dqn.py:49, 90: xavier_uniform
        (limit sqrt(6/(fan_in+fan_out)), conv fans x receptive field) for cnn,
        lecun_normal (truncated normal, std sqrt(1/fan_in)/0.87962566103423978) for fc;
        biases 0, LayerNorm scale 1 / bias 0.  (JAX threefry streams are not
        reproducible offline: draws come from numpy PCG64 instead.)
  * ``nn.BatchNorm`` (batch_norm=True) ......... dqn.py:52-53, 59-60, 66-67, 73-74, 100-101 -- flax 0.10.2 defaults: momentum 0.99,
        epsilon 1e-5, use_fast_variance (var = max(0, E[x^2] - E[x]^2)), scale 1 / bias 0, batch_stats mean 0 / var 1;
        ``axis`` names the FEATURE axes: ``BatchNorm(use_running_average, axis=(1, 2))`` on (N, H, W, C) keeps one statistic and
        one (scale, bias) per pixel position (h, w) and reduces over the batch AND the channels; the default axis=-1 on the
        flattened / dense (N, F) tensors is per feature.  y = (x - mean) * (rsqrt(var + eps) * scale) + bias.  Training-mode
        calls (apply_fn with mutable batch_stats, isdqn.py:40, 95) use the batch statistics and move the running averages
        ra = 0.99 ra + 0.01 batch; best_action (isdqn.py:130) passes use_running_average=True.
  * iSDQN.loss_on_batch / compute_target ....... slimdqn/networks/isdqn.py:92-109
  * iSDQN.learn_on_batch (grad + optax.adam) ... isdqn.py:82-90, 46
        adam: m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; t += 1 ;
        p -= lr * (m/(1-b1^t)) / (sqrt(v/(1-b2^t)) + eps)       (eps_root = 0)
  * iSDQN.shift_params ......................... isdqn.py:111-125 (moments untouched)
  * iSDQN.best_action .......................... isdqn.py:127-135

Gradients come from torch autograd on the restated forward (an implementation
independent of the hand-derived HIP backward); ``forward_numpy`` is a second,
plain-numpy im2col statement of the forward used to cross-check the torch one.
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence

import numpy as np
import torch
import torch.nn.functional as F

LN_EPS = 1e-6
BN_EPS, BN_MOMENTUM = 1e-5, 0.99
ADAM_B1, ADAM_B2 = 0.9, 0.999

# (kernel, stride) of the three cnn torso convolutions (dqn.py:55, 62, 69)
CNN_GEOMETRY = ((8, 4), (4, 2), (3, 1))


def same_padding(size: int, kernel: int, stride: int):
    """JAX/Flax "SAME": out = ceil(in/stride); lo = total//2; hi = total-lo."""
    out = -(-size // stride)
    total = max((out - 1) * stride + kernel - size, 0)
    lo = total // 2
    return out, lo, total - lo


def layer_names(features: Sequence[int], architecture_type: str, layer_norm: bool):
    """Ordered (module_name, kind) list as Flax auto-names them."""
    names = []
    n_conv = n_ln = n_dense = 0
    if architecture_type == "cnn":
        for _ in range(3):
            names.append((f"Conv_{n_conv}", "conv"))
            n_conv += 1
            if layer_norm:
                names.append((f"LayerNorm_{n_ln}", "ln"))
                n_ln += 1
        start = 3
    elif architecture_type == "fc":
        start = 0
    else:
        raise ValueError(architecture_type)
    for _ in range(start, len(features)):
        names.append((f"Dense_{n_dense}", "dense"))
        n_dense += 1
        if layer_norm:
            names.append((f"LayerNorm_{n_ln}", "ln"))
            n_ln += 1
    names.append((f"Dense_{n_dense}", "dense"))
    return names


def init_params(
    seed: int,
    observation_dim: Sequence[int],
    features: Sequence[int],
    architecture_type: str,
    final_feature: int,
    layer_norm: bool,
    batch_norm: bool = False,
) -> Dict[str, Dict[str, np.ndarray]]:
    """Flax-layout parameter pytree (the inner ``params["params"]`` dict), float32.  ``batch_norm``: BatchNorm_i scale / bias
    join it in call order (the running averages are a second collection: ``init_batch_stats``)."""
    rng = np.random.default_rng(seed)
    params: Dict[str, Dict[str, np.ndarray]] = {}
    n_bn = 0

    def add_bn(shape):
        nonlocal n_bn
        if batch_norm:
            params[f"BatchNorm_{n_bn}"] = {"scale": np.ones(shape, np.float32), "bias": np.zeros(shape, np.float32)}
            n_bn += 1


    def xavier(shape, fan_in, fan_out):
        lim = math.sqrt(6.0 / (fan_in + fan_out))
        return rng.uniform(-lim, lim, size=shape).astype(np.float32)

    def lecun(shape, fan_in):
        std = math.sqrt(1.0 / fan_in) / 0.87962566103423978
        out = np.empty(int(np.prod(shape)), dtype=np.float64)
        filled = 0
        while filled < out.size:  # truncated normal on [-2, 2]
            draw = rng.standard_normal(out.size - filled)
            draw = draw[np.abs(draw) <= 2.0]
            out[filled : filled + draw.size] = draw
            filled += draw.size
        return (out.reshape(shape) * std).astype(np.float32)

    n_conv = n_ln = n_dense = 0

    def add_ln(width):
        nonlocal n_ln
        if layer_norm:
            params[f"LayerNorm_{n_ln}"] = {
                "scale": np.ones(width, np.float32),
                "bias": np.zeros(width, np.float32),
            }
            n_ln += 1

    if architecture_type == "cnn":
        h, w, c = observation_dim
        add_bn((h, w))  # dqn.py:52-53: BatchNorm(axis=(1, 2)) on x / 255
        for i, (k, s) in enumerate(CNN_GEOMETRY):
            cout = int(features[i])
            params[f"Conv_{n_conv}"] = {
                "kernel": xavier((k, k, c, cout), k * k * c, k * k * cout),
                "bias": np.zeros(cout, np.float32),
            }
            n_conv += 1
            add_ln(cout)
            h = same_padding(h, k, s)[0]
            w = same_padding(w, k, s)[0]
            c = cout
            add_bn((h, w) if i < 2 else (h * w * c,))  # :59-60, 66-67 on the image tensor; :72-74 behind the flatten
        width = h * w * c
        dense_feats = [int(f) for f in features[3:]]
        init = lambda shape: xavier(shape, shape[0], shape[1])
    elif architecture_type == "impala":
        # dqn.py:75-88 + Stack (dqn.py:7-36): three stacks of conv3x3 -> max_pool 3x3 / 2 SAME -> two residual blocks
        # ([LayerNorm] -> relu -> conv3x3 -> relu -> conv3x3 -> + input); modules are auto-named per class INSIDE each Stack
        # (Conv_0 .. Conv_4, LayerNorm_0 .. LayerNorm_1), the Stacks Stack_0 .. Stack_2, and the LayerNorm behind the last stack
        # is the top level's LayerNorm_0.  The stack's first conv takes xavier_uniform, the block convs Flax's default
        # lecun_normal (dqn.py:17-21 passes kernel_init only there).  Nested modules are flattened to "Stack_s/Conv_k" keys.
        # batch_norm: BatchNorm(axis=(1, 2)) on x / 255 (:78-79, top-level BatchNorm_0), inside every residual block behind the
        # ReLU (:29-30, "Stack_s/BatchNorm_b", one statistic per pooled pixel position) and per feature behind the flatten (:86-88)
        h, w, c = observation_dim
        add_bn((h, w))
        for s_idx in range(3):
            cout = int(features[s_idx])
            params[f"Stack_{s_idx}/Conv_0"] = {"kernel": xavier((3, 3, c, cout), 9 * c, 9 * cout), "bias": np.zeros(cout, np.float32)}
            hp, wp = same_padding(h, 3, 2)[0], same_padding(w, 3, 2)[0]
            for b in range(2):
                if layer_norm:
                    params[f"Stack_{s_idx}/LayerNorm_{b}"] = {"scale": np.ones(cout, np.float32), "bias": np.zeros(cout, np.float32)}
                if batch_norm:
                    params[f"Stack_{s_idx}/BatchNorm_{b}"] = {"scale": np.ones((hp, wp), np.float32), "bias": np.zeros((hp, wp), np.float32)}
                for k in (1 + 2 * b, 2 + 2 * b):
                    params[f"Stack_{s_idx}/Conv_{k}"] = {"kernel": lecun((3, 3, cout, cout), 9 * cout), "bias": np.zeros(cout, np.float32)}
            h, w, c = hp, wp, cout
        add_ln(c)
        add_bn((h * w * c,))
        width = h * w * c
        dense_feats = [int(f) for f in features[3:]]
        init = lambda shape: xavier(shape, shape[0], shape[1])
    elif architecture_type == "fc":
        width = int(np.prod(observation_dim))
        dense_feats = [int(f) for f in features]
        init = lambda shape: lecun(shape, shape[0])
    else:
        raise ValueError(architecture_type)

    for f in dense_feats:
        params[f"Dense_{n_dense}"] = {"kernel": init((width, f)), "bias": np.zeros(f, np.float32)}
        n_dense += 1
        add_ln(f)
        add_bn((f,))  # dqn.py:100-101
        width = f
    params[f"Dense_{n_dense}"] = {
        "kernel": init((width, final_feature)),
        "bias": np.zeros(final_feature, np.float32),
    }
    return params


def init_batch_stats(params) -> Dict[str, Dict[str, np.ndarray]]:
    """Flax's ``batch_stats`` collection at initialisation: mean 0, var 1 with the shape of each BatchNorm_i's scale."""
    return {m: {"mean": np.zeros_like(np.asarray(l["scale"]), dtype=np.float32), "var": np.ones_like(np.asarray(l["scale"]), dtype=np.float32)}
            for m, l in params.items() if m.rsplit("/", 1)[-1].startswith("BatchNorm_")}


# ----------------------------------------------------------------------------- torch forward
def _batch_norm(x: torch.Tensor, p, stats, use_running_average: bool, spatial: bool, new_stats, name: str) -> torch.Tensor:
    """flax.linen.BatchNorm: ``spatial`` = axis=(1, 2) on (N, H, W, C) -- features (H, W), reduction over (N, C); else axis=-1 on
    (N, F).  ``new_stats[name]`` receives the moved running averages of a training-mode call."""
    red = (0, 3) if spatial else (0,)
    if use_running_average:
        mean, var = stats[name]["mean"], stats[name]["var"]
    else:
        mean = x.mean(dim=red)
        var = torch.clamp((x * x).mean(dim=red) - mean * mean, min=0.0)
        if new_stats is not None:
            new_stats[name] = {"mean": (BN_MOMENTUM * stats[name]["mean"] + (1 - BN_MOMENTUM) * mean).detach(),
                               "var": (BN_MOMENTUM * stats[name]["var"] + (1 - BN_MOMENTUM) * var).detach()}
    if spatial:
        mean, var, scale, bias = (t[None, :, :, None] for t in (mean, var, p["scale"], p["bias"]))
    else:
        scale, bias = p["scale"], p["bias"]
    return (x - mean) * (torch.rsqrt(var + BN_EPS) * scale) + bias


def _layer_norm(x: torch.Tensor, scale: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    mean = x.mean(dim=-1, keepdim=True)
    mean2 = (x * x).mean(dim=-1, keepdim=True)
    var = torch.clamp(mean2 - mean * mean, min=0.0)
    mul = torch.rsqrt(var + LN_EPS) * scale
    return (x - mean) * mul + bias


def _conv_same(x_nhwc: torch.Tensor, kernel_hwio: torch.Tensor, bias: torch.Tensor, stride: int) -> torch.Tensor:
    kh, kw = kernel_hwio.shape[0], kernel_hwio.shape[1]
    _, plo_h, phi_h = same_padding(x_nhwc.shape[1], kh, stride)
    _, plo_w, phi_w = same_padding(x_nhwc.shape[2], kw, stride)
    x = x_nhwc.permute(0, 3, 1, 2)
    x = F.pad(x, (plo_w, phi_w, plo_h, phi_h))
    y = F.conv2d(x, kernel_hwio.permute(3, 2, 0, 1), bias=bias, stride=stride)
    return y.permute(0, 2, 3, 1)


def _max_pool_same(x_nhwc: torch.Tensor, window: int = 3, stride: int = 2) -> torch.Tensor:
    """flax.linen.max_pool(x, (3, 3), strides=(2, 2), padding="SAME") (dqn.py:22): -inf padding, lo = total // 2."""
    _, plo_h, phi_h = same_padding(x_nhwc.shape[1], window, stride)
    _, plo_w, phi_w = same_padding(x_nhwc.shape[2], window, stride)
    x = F.pad(x_nhwc.permute(0, 3, 1, 2), (plo_w, phi_w, plo_h, phi_h), value=float("-inf"))
    return F.max_pool2d(x, window, stride).permute(0, 2, 3, 1)


def _impala_stack(params, prefix: str, x: torch.Tensor, layer_norm: bool, capture: dict | None = None, bn=None) -> torch.Tensor:
    """Stack.__call__ (dqn.py:14-36).  ``capture`` receives the residual stream in front of each block; ``bn(x, name)`` applies the
    block's BatchNorm(axis=(1, 2)) behind the ReLU (dqn.py:29-30) when the network has them."""
    p = params[f"{prefix}/Conv_0"]
    x = _conv_same(x, p["kernel"], p["bias"], 1)
    x = _max_pool_same(x)
    for b in range(2):
        block_input = x
        if capture is not None:
            capture[f"{prefix}/r{b}"] = x
        if layer_norm:
            q = params[f"{prefix}/LayerNorm_{b}"]
            x = _layer_norm(x, q["scale"], q["bias"])
        x = torch.relu(x)
        if capture is not None:
            capture[f"{prefix}/a1_{b}"] = x  # (AnalysisNet's Stack records the sums of both ReLU outputs of a block)
        if bn is not None:
            x = bn(x, f"{prefix}/BatchNorm_{b}")
        p = params[f"{prefix}/Conv_{1 + 2 * b}"]
        x = _conv_same(x, p["kernel"], p["bias"], 1)
        x = torch.relu(x)
        if capture is not None:
            capture[f"{prefix}/a2_{b}"] = x
        p = params[f"{prefix}/Conv_{2 + 2 * b}"]
        x = _conv_same(x, p["kernel"], p["bias"], 1)
        x = x + block_input
    return x


def forward(params, x, features, architecture_type: str, layer_norm: bool, capture: dict | None = None, batch_norm: bool = False,
            batch_stats=None, use_running_average: bool = False, new_stats: dict | None = None):
    """DQNNet.__call__ (dqn.py:47-103) for a batch.  ``params``: dict of dicts of torch tensors.
    ``batch_norm``: ``batch_stats`` holds the running averages ({"BatchNorm_i": {"mean", "var"}} of torch tensors);
    training-mode calls (use_running_average=False) normalise with the batch statistics and write the moved averages to ``new_stats``.

    x: (N,84,84,4) raw pixel values (any dtype; converted, then /255) for cnn, (N,obs) for fc.
    Returns (N, final_feature).  ``capture`` (optional dict) receives the post-activation of
    every hidden layer under the module name that produced it.
    """
    dtype = next(iter(next(iter(params.values())).values())).dtype
    n_ln = n_dense = n_bn = 0
    x = x.to(dtype)

    def bn(x, spatial):
        nonlocal n_bn
        if not batch_norm:
            return x
        name = f"BatchNorm_{n_bn}"
        n_bn += 1
        y = _batch_norm(x, params[name], batch_stats, use_running_average, spatial, new_stats, name)
        if capture is not None:
            capture[name] = y
        return y

    if architecture_type == "cnn":
        x = bn(x / 255.0, True)
        for i, (_k, s) in enumerate(CNN_GEOMETRY):
            p = params[f"Conv_{i}"]
            x = _conv_same(x, p["kernel"], p["bias"], s)
            if layer_norm:
                q = params[f"LayerNorm_{n_ln}"]
                x = _layer_norm(x, q["scale"], q["bias"])
                n_ln += 1
            x = torch.relu(x)
            if capture is not None:
                capture[f"Conv_{i}"] = x
            if i < 2:
                x = bn(x, True)
        x = bn(x.reshape(x.shape[0], -1), False)
        start = 3
    elif architecture_type == "impala":  # dqn.py:75-88
        x = bn(x / 255.0, True)
        inner = (lambda t, name: _batch_norm(t, params[name], batch_stats, use_running_average, True, new_stats, name)) if batch_norm else None
        for s_idx in range(3):
            x = _impala_stack(params, f"Stack_{s_idx}", x, layer_norm, capture, bn=inner)
            if capture is not None:
                capture[f"Stack_{s_idx}"] = x
        if layer_norm:
            q = params[f"LayerNorm_{n_ln}"]
            x = _layer_norm(x, q["scale"], q["bias"])
            n_ln += 1
        x = torch.relu(x).reshape(x.shape[0], -1)
        if capture is not None:
            capture["ImpalaOut"] = x
        x = bn(x, False)
        start = 3
    else:
        x = x.reshape(x.shape[0], -1)
        start = 0
    for _ in range(start, len(features)):
        p = params[f"Dense_{n_dense}"]
        x = x @ p["kernel"] + p["bias"]
        if layer_norm:
            q = params[f"LayerNorm_{n_ln}"]
            x = _layer_norm(x, q["scale"], q["bias"])
            n_ln += 1
        x = torch.relu(x)
        if capture is not None:
            capture[f"Dense_{n_dense}"] = x
        x = bn(x, False)
        n_dense += 1
    p = params[f"Dense_{n_dense}"]
    return x @ p["kernel"] + p["bias"]


def to_torch(params_np, dtype=torch.float32, requires_grad=False):
    out = {}
    for mod, leaves in params_np.items():
        out[mod] = {}
        for name, arr in leaves.items():
            t = torch.tensor(np.asarray(arr), dtype=dtype)
            t.requires_grad_(requires_grad)
            out[mod][name] = t
    return out


def to_numpy(params_t):
    return {m: {n: t.detach().cpu().numpy().copy() for n, t in leaves.items()} for m, leaves in params_t.items()}


# ----------------------------------------------------------------------------- numpy cross-check
def forward_numpy(params_np, x, features, architecture_type: str, layer_norm: bool):
    """Plain-numpy float64 statement of the same forward (im2col loops; small inputs only)."""
    x = np.asarray(x, np.float64)
    P = {m: {n: np.asarray(a, np.float64) for n, a in l.items()} for m, l in params_np.items()}

    def ln(z, q):
        mean = z.mean(-1, keepdims=True)
        var = np.maximum((z * z).mean(-1, keepdims=True) - mean * mean, 0.0)
        return (z - mean) * (1.0 / np.sqrt(var + LN_EPS) * q["scale"]) + q["bias"]

    n_ln = n_dense = 0
    if architecture_type == "cnn":
        x = x / 255.0
        for i, (k, s) in enumerate(CNN_GEOMETRY):
            W, b = P[f"Conv_{i}"]["kernel"], P[f"Conv_{i}"]["bias"]
            N, H, Wd, C = x.shape
            oh, plo_h, phi_h = same_padding(H, k, s)
            ow, plo_w, phi_w = same_padding(Wd, k, s)
            xp = np.zeros((N, H + plo_h + phi_h, Wd + plo_w + phi_w, C))
            xp[:, plo_h : plo_h + H, plo_w : plo_w + Wd] = x
            y = np.zeros((N, oh, ow, W.shape[3]))
            for oy in range(oh):
                for ox in range(ow):
                    patch = xp[:, oy * s : oy * s + k, ox * s : ox * s + k, :]
                    y[:, oy, ox, :] = np.tensordot(patch, W, axes=([1, 2, 3], [0, 1, 2])) + b
            if layer_norm:
                y = ln(y, P[f"LayerNorm_{n_ln}"])
                n_ln += 1
            x = np.maximum(y, 0.0)
        x = x.reshape(x.shape[0], -1)
        start = 3
    else:
        x = x.reshape(x.shape[0], -1)
        start = 0
    for _ in range(start, len(features)):
        x = x @ P[f"Dense_{n_dense}"]["kernel"] + P[f"Dense_{n_dense}"]["bias"]
        if layer_norm:
            x = ln(x, P[f"LayerNorm_{n_ln}"])
            n_ln += 1
        x = np.maximum(x, 0.0)
        n_dense += 1
    return x @ P[f"Dense_{n_dense}"]["kernel"] + P[f"Dense_{n_dense}"]["bias"]
