"""``--analysis`` hook of the trainer (experiments/base/srank_and_dead_neurons.py:8-22 of the reference): sample 2048 states,
run the torso on the GPU (isdqn_net_analysis), compute the srank and the dead-neuron fraction on the host."""
from __future__ import annotations

import numpy as np
import torch

from slimdqn._engine import QNetEngine
from slimdqn.utils.analysis import compute_dead_neurons, compute_srank

N_ANALYSIS_SAMPLES = 2048  # srank_and_dead_neurons.py:16
_engines: dict = {}


def _analysis_engine(eng: QNetEngine, n_rows: int) -> QNetEngine:
    """An engine of the same network whose workspace holds n_rows forward rows (rows = 2 * batch_size)."""
    if 2 * eng.batch_size >= n_rows:
        return eng
    key = (id(eng), n_rows)
    if key not in _engines:
        c = eng.cfg
        _engines.clear()  # one at a time: a workspace of this size is hundreds of MB
        _engines[key] = QNetEngine(
            eng.observation_dim, eng.n_actions, eng.n_heads, eng.features, eng.architecture_type, bool(c.layer_norm), (n_rows + 1) // 2,
            gamma_n=float(c.gamma_n), learning_rate=float(c.learning_rate), adam_eps=float(c.adam_eps), precision=eng.precision,
            device=eng.device, batch_norm=eng.batch_norm,
        )
    return _engines[key]


def eval_srank_and_dead_neurons(params, rb, p=None, size: int = N_ANALYSIS_SAMPLES):
    """``params``: the agent's parameter handle (DeviceParams).  Returns {"srank", "dead_neurons"} as the reference does."""
    eng = params._engine
    samples = rb.sample(size=size)
    big = _analysis_engine(eng, size)
    stack = samples.frame_ids.shape[1] // 2
    ids = samples.frame_ids[:, :stack].contiguous()  # the states of the sampled transitions
    feats, scores = big.analysis(frames=samples.frames, frame_stride=samples.frame_stride, frame_ids=ids, n_rows=size, params=params.tensor)
    return {
        "srank": float(compute_srank(feats.cpu().numpy())),
        "dead_neurons": float(compute_dead_neurons([s.cpu().numpy() for s in scores])),
    }
