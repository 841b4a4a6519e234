"""DQN on LunarLander (slimdqn/networks/dqn.py on the HIP engine, MLP torso).

    python experiments/lunar_lander/dqn.py -en test -s 1 -f 100 100 -at fc ...

The reference tests call this entry point (tests/test_lunar_lander.py:8-59) but its tree does not ship it; this one follows
experiments/atari/dqn.py with the fc architecture and stack size 1 (experiments/lunar_lander/common.py).
"""
import os
import sys

_PKG = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)

import numpy as np

from experiments.base.dqn import train
from experiments.base.utils import prepare_logs
from experiments.lunar_lander.common import make_environment, make_replay, seeds
from slimdqn.networks.dqn import DQN


def run(argvs=sys.argv[1:], root=None):
    from experiments.base import dist as replicas

    replicas.init_from_env()
    p = prepare_logs("lunar_lander", "dqn", argvs, root=root)
    q_seed, train_seed = seeds(p)
    env = make_environment(p)
    rb = make_replay(p)
    agent = DQN(
        q_seed,
        env.observation_shape,
        env.n_actions,
        features=p["features"],
        layer_norm=p["layer_norm"],
        architecture_type="fc",
        learning_rate=p["learning_rate"],
        gamma=p["gamma"],
        update_horizon=p["update_horizon"],
        data_to_update=p["data_to_update"],
        target_update_frequency=p["target_update_frequency"],
        batch_size=p["batch_size"],
        precision=p["precision"],
    )
    out = train(np.random.default_rng(train_seed), p, agent, env, rb)
    replicas.finalize()
    return out


if __name__ == "__main__":
    run()
