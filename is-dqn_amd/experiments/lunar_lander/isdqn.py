"""iS-DQN on LunarLander (BASELINE configs[0]: K=1, MLP torso [100, 100], uniform replay, batch 32) on the HIP engine.

    python experiments/lunar_lander/isdqn.py -en test -s 1 -f 100 100 -nbi 1 ...

The reference ships the LunarLander wrapper (slimdqn/environments/lunar_lander.py:5-23) and tests that call
``experiments/lunar_lander/*.py`` (tests/test_lunar_lander.py:18) but not the entry points themselves; this one follows
experiments/atari/isdqn.py:15-48 with the fc architecture and stack size 1.  There is no CPU backend: the fc torso runs on the
same HIP kernels as the Atari configurations (tests/test_gpu_network.py: test_fc_architecture_lunar_lander_shape).
``-env synthetic`` replaces gymnasium's LunarLander-v3 (absent from the build image) by a seeded stand-in.
"""
import os
import sys

_PKG = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)

import numpy as np

from experiments.atari.common import seeds
from experiments.base.dqn import train
from experiments.base.utils import prepare_logs
from slimdqn.networks.isdqn import iSDQN
from slimdqn.sample_collection.replay_buffer import ReplayBuffer
from slimdqn.sample_collection.samplers import UniformSamplingDistribution


def run(argvs=sys.argv[1:], root=None):
    from experiments.base import dist as replicas

    replicas.init_from_env()
    p = prepare_logs("lunar_lander", "isdqn", argvs, root=root)
    assert p["architecture_type"] == "fc", "LunarLander observations are vectors: -at fc"
    q_seed, train_seed = seeds(p)
    if p["env_backend"] == "synthetic":
        from slimdqn.environments.lunar_lander import SyntheticLunarLander

        env = SyntheticLunarLander(seed=p["seed"])
    else:
        from slimdqn.environments.lunar_lander import LunarLander

        env = LunarLander()
    rb = ReplayBuffer(
        sampling_distribution=UniformSamplingDistribution(p["seed"]),
        max_capacity=p["replay_buffer_capacity"],
        batch_size=p["batch_size"],
        update_horizon=p["update_horizon"],
        gamma=p["gamma"],
        clipping=None,
        stack_size=1,
        compress=False,
    )
    agent = iSDQN(
        q_seed,
        env.observation_shape,
        env.n_actions,
        n_bellman_iterations=p["n_bellman_iterations"],
        features=p["features"],
        layer_norm=p["layer_norm"],
        batch_norm=p["batch_norm"],
        architecture_type="fc",
        learning_rate=p["learning_rate"],
        gamma=p["gamma"],
        update_horizon=p["update_horizon"],
        data_to_update=p["data_to_update"],
        target_update_frequency=p["target_update_frequency"],
        batch_size=p["batch_size"],
        precision=p["precision"],
        huber_delta=p["huber_delta"],
    )
    out = train(np.random.default_rng(train_seed), p, agent, env, rb)
    replicas.finalize()
    return out


if __name__ == "__main__":
    run()
