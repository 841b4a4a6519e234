"""ORACLE (test infrastructure) -- the reference's representation diagnostics, CPU restatement.

  * AnalysisNet.apply ......... slimdqn/utils/analysis_architecture.py:9-122 (cnn / impala / fc torsos, optional LayerNorm and BatchNorm): the
                                network without its last Dense; after every ReLU the sum over the batch axis is recorded
  * compute_srank ............. slimdqn/utils/analysis.py:4-8
  * compute_dead_neurons ...... slimdqn/utils/analysis.py:11-17
  * eval_srank_and_dead_neurons experiments/base/srank_and_dead_neurons.py:8-22

PARITY UNPINNED for the torso numerics (oracle/network.py); the two host formulas are plain numpy in the reference too.
"""
from __future__ import annotations

import numpy as np
import torch

from oracle import network as net


def analysis_net(params, states, features, architecture_type: str, layer_norm: bool, batch_norm: bool = False, batch_stats=None):
    """(feature_matrix (N, width), [per-layer activation sums over the batch]) -- float64 torch CPU.

    The recorded layers, in the reference's order: cnn -- the three conv ReLUs, then every hidden Dense ReLU; impala -- per Stack the
    two ReLU outputs of each of its two residual blocks (analysis_architecture.py:27-40), then the flattened torso output, then the
    Dense ReLUs; fc -- the Dense ReLUs.  BatchNorm networks are applied as the reference applies them here (mutable batch_stats,
    use_running_average left False: the batch statistics of the analysed states, srank_and_dead_neurons.py:17); the sums are taken
    in front of each BatchNorm, the returned feature matrix behind the last one (analysis_architecture.py:115-122)."""
    capture: dict = {}
    p64 = {m: {n: t.to(torch.float64) for n, t in l.items()} for m, l in params.items()}
    s64 = None if batch_stats is None else {m: {n: t.to(torch.float64) for n, t in l.items()} for m, l in batch_stats.items()}
    x = torch.as_tensor(np.asarray(states))
    net.forward(p64, x, features, architecture_type, layer_norm, capture=capture, batch_norm=batch_norm, batch_stats=s64, new_stats=None)
    n_dense = len(features) - (0 if architecture_type == "fc" else 3)
    keys = []
    if architecture_type == "cnn":
        keys += [f"Conv_{i}" for i in range(3)]
    elif architecture_type == "impala":
        for s_idx in range(3):
            for b in range(2):
                keys += [f"Stack_{s_idx}/a1_{b}", f"Stack_{s_idx}/a2_{b}"]
        keys.append("ImpalaOut")
    keys += [f"Dense_{i}" for i in range(n_dense)]
    acts = [capture[k] for k in keys]
    scores = [a.sum(dim=0).reshape(-1).numpy() for a in acts]
    feat = acts[-1]
    if batch_norm:  # the features leave the network behind its last BatchNorm
        last_bn = max(int(k.split("_")[1]) for k in capture if k.startswith("BatchNorm_"))
        feat = capture[f"BatchNorm_{last_bn}"]
    return feat.reshape(feat.shape[0], -1).numpy(), scores


def compute_srank(feature_matrix, delta=0.01):
    sv = np.sort(np.linalg.svd(feature_matrix, full_matrices=False, compute_uv=False))[::-1]
    cum = np.cumsum(sv)
    return int(np.searchsorted(cum, (1 - delta) * cum[-1], side="left") + 1)


def compute_dead_neurons(score_neurons, tau=0):
    dead = sum(int(np.count_nonzero(s / (s.mean() + 1e-9) <= tau)) for s in score_neurons)
    return dead / sum(s.size for s in score_neurons)
