"""AnalysisTFDQN on Atari (target-free DQN + target-churn diagnostics): the reference's entry point (experiments/atari/analysistfdqn.py) on the HIP engine.

    python experiments/atari/analysistfdqn.py -en LN1_cnn_Asterix -s 1 -f 32 64 64 512 -at cnn -ln ...

``experiment_name`` must end in ``_<Game>``; outputs go under experiments/atari/exp_output/<name>/analysistfdqn/.
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
from slimdqn.networks.analysistfdqn import AnalysisTFDQN


def run(argvs=sys.argv[1:], root=None):
    from experiments.base import dist as replicas

    replicas.init_from_env()  # one process per GPU: picks this rank's device before the first GPU call (no-op alone)
    p = prepare_logs("atari", "analysistfdqn", argvs, root=root)
    q_seed, train_seed = seeds(p)
    env = make_environment(p)  # (worker processes of a VectorEnv start here, before this process's first GPU call)
    rb = make_replay(p, prioritized=False)
    agent = AnalysisTFDQN(
        q_seed,
        (env.state_height, env.state_width, env.n_stacked_frames),
        env.n_actions,
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
    )
    try:
        out = train(np.random.default_rng(train_seed), p, agent, env, rb)
    finally:
        if hasattr(env, "close"):
            env.close()
    replicas.finalize()
    return out


if __name__ == "__main__":
    run()
