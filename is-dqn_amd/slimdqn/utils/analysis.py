"""Representation diagnostics of the reference (slimdqn/utils/analysis.py:4-17): host arithmetic on what
``isdqn_net_analysis`` returns -- the feature matrix of the last hidden layer and the per-neuron activation sums."""
from __future__ import annotations

import numpy as np


def compute_srank(feature_matrix, delta: float = 0.01) -> int:
    """Effective rank: the smallest k whose k largest singular values hold a (1 - delta) share of their sum."""
    s = np.linalg.svd(np.asarray(feature_matrix), compute_uv=False)  # (descending)
    share = np.cumsum(s)
    return int(np.searchsorted(share, (1.0 - delta) * share[-1], side="left")) + 1


def compute_dead_neurons(score_neurons, tau: float = 0.0) -> float:
    """Fraction of neurons whose activation sum, normalised by their layer's mean, is <= tau."""
    dead = total = 0
    for layer in score_neurons:
        layer = np.asarray(layer)
        dead += int(np.count_nonzero(layer / (layer.mean() + 1e-9) <= tau))
        total += layer.size
    return dead / total
