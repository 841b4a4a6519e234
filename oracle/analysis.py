"""ORACLE (test infrastructure) -- the reference's representation diagnostics, CPU restatement.

  * AnalysisNet.apply ......... slimdqn/utils/analysis_architecture.py:46-122 (cnn / fc torsos, optional LayerNorm): the
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


def analysis_net(params, states, features, architecture_type: str, layer_norm: bool):
    """(feature_matrix (N, width), [per-layer activation sums over the batch]) -- float64 torch CPU."""
    capture: dict = {}
    p64 = {m: {n: t.to(torch.float64) for n, t in l.items()} for m, l in params.items()}
    x = torch.as_tensor(np.asarray(states))
    net.forward(p64, x, features, architecture_type, layer_norm, capture=capture)
    acts = list(capture.values())  # hidden layers in network order
    scores = [a.sum(dim=0).reshape(-1).numpy() for a in acts]
    return acts[-1].reshape(acts[-1].shape[0], -1).numpy(), scores


def compute_srank(feature_matrix, delta=0.01):
    sv = np.sort(np.linalg.svd(feature_matrix, full_matrices=False, compute_uv=False))[::-1]
    cum = np.cumsum(sv)
    return int(np.searchsorted(cum, (1 - delta) * cum[-1], side="left") + 1)


def compute_dead_neurons(score_neurons, tau=0):
    dead = sum(int(np.count_nonzero(s / (s.mean() + 1e-9) <= tau)) for s in score_neurons)
    return dead / sum(s.size for s in score_neurons)
