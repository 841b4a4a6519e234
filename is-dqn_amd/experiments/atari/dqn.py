"""DQN on Atari: the reference's entry point (experiments/atari/dqn.py:15-48) on the HIP engine (separate target
parameters, refreshed every -tuf steps).

    python experiments/atari/dqn.py -en LN1_cnn_Asterix -s 1 -f 32 64 64 512 -at cnn -ln ...

``experiment_name`` must end in ``_<Game>``; outputs go under experiments/atari/exp_output/<name>/dqn/.
"""
import os
import sys

_PKG = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)

import numpy as np

from experiments.base.dqn import train
from experiments.base.utils import prepare_logs
from slimdqn.networks.dqn import DQN
from slimdqn.sample_collection.replay_buffer import ReplayBuffer
from slimdqn.sample_collection.samplers import UniformSamplingDistribution


def run(argvs=sys.argv[1:], root=None):
    from experiments.base import dist as replicas

    replicas.init_from_env()  # one process per GPU: picks this rank's device before the first GPU call (no-op alone)
    p = prepare_logs("atari", "dqn", argvs, root=root)
    rng = np.random.default_rng(p["seed"])
    q_seed, train_seed = (int(s) for s in rng.integers(0, 2**31 - 1, size=2))

    game = p["experiment_name"].split("_")[-1]
    if p["env_backend"] == "synthetic":
        from slimdqn.environments.synthetic import SyntheticAtariEnv

        make_env = lambda i: SyntheticAtariEnv(game, seed=p["seed"] + 1000 * i)
    else:
        from slimdqn.environments.atari import AtariEnv

        make_env = lambda i: AtariEnv(game)
    if p["n_envs"] > 1:
        from slimdqn.environments.vector import VectorEnv

        env = VectorEnv([make_env(i) for i in range(p["n_envs"])])
    else:
        env = make_env(0)
    sampler = UniformSamplingDistribution(p["seed"])
    rb = ReplayBuffer(
        sampling_distribution=sampler,
        max_capacity=p["replay_buffer_capacity"],
        batch_size=p["batch_size"],
        update_horizon=p["update_horizon"],
        gamma=p["gamma"],
        clipping=lambda x: np.clip(x, -1, 1),
        stack_size=4,
        compress=True,
    )
    agent = DQN(
        q_seed,
        (env.state_height, env.state_width, env.n_stacked_frames),
        env.n_actions,
        features=p["features"],
        layer_norm=p["layer_norm"],
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
    out = train(np.random.default_rng(train_seed), p, agent, env, rb)
    replicas.finalize()
    return out


if __name__ == "__main__":
    run()
