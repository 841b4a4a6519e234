"""AnalysisDQN on Atari: the reference's entry point (experiments/atari/analysisdqn.py) on the HIP engine -- iS-DQN with target-churn
and gradient-cosine diagnostics (slimdqn/networks/analysisdqn.py).

    python experiments/atari/analysisdqn.py -en L2_K9_LN1_cnn_Asterix -s 1 -f 32 64 64 512 -at cnn -ln -nbi 9 ...

``experiment_name`` must end in ``_<Game>``; outputs go under experiments/atari/exp_output/<name>/analysisdqn/.
"""
import os
import sys

_PKG = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)

import numpy as np

from experiments.base.dqn import train
from experiments.atari.common import make_environment, make_replay, seeds
from experiments.base.utils import prepare_logs
from slimdqn.networks.analysisdqn import AnalysisDQN


def run(argvs=sys.argv[1:], root=None):
    from experiments.base import dist as replicas

    replicas.init_from_env()  # one process per GPU: picks this rank's device before the first GPU call (no-op alone)
    p = prepare_logs("atari", "analysisdqn", argvs, root=root)
    q_seed, train_seed = seeds(p)
    env = make_environment(p)  # (worker processes of a VectorEnv start here, before this process's first GPU call)
    rb = make_replay(p, prioritized=p["prioritized"])
    agent = AnalysisDQN(
        q_seed,
        (env.state_height, env.state_width, env.n_stacked_frames),
        env.n_actions,
        n_bellman_iterations=p["n_bellman_iterations"],
        features=p["features"],
        layer_norm=p["layer_norm"],
        batch_norm=p["batch_norm"],
        architecture_type=p["architecture_type"],
        learning_rate=p["learning_rate"],
        gamma=p["gamma"],
        update_horizon=p["update_horizon"],
        data_to_update=p["data_to_update"],
        target_update_frequency=p["target_update_frequency"],
        adam_eps=1.5e-4,
        batch_size=p["batch_size"],
        precision=p["precision"],
        huber_delta=p["huber_delta"],
    )
    if p["prioritized"]:
        _wire_prioritized(agent, rb)
    try:
        out = train(np.random.default_rng(train_seed), p, agent, env, rb)
    finally:
        if hasattr(env, "close"):
            env.close()
    replicas.finalize()
    return out


def _wire_prioritized(agent, rb):
    """Trainer wiring the reference does not have: new elements enter with the
    largest priority seen so far (Dopamine's convention), sampled elements get sqrt(mean_k td) written back.
    Neither costs a read-back: the maximum is resolved on the device when the staged leaf writes are flushed
    (samplers.py MAX_PRIORITY), the write-back is part of the captured step (networks/isdqn.py)."""
    sampler = rb._sampling_distribution
    plain_add = rb.add
    rb.add = lambda transition, **kw: plain_add(transition, **{"priority": sampler.MAX_PRIORITY, **kw})
    agent.priority_writeback = True


if __name__ == "__main__":
    run()
