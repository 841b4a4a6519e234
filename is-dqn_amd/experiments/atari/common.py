"""Environment / replay wiring shared by the Atari entry points (reference: experiments/atari/isdqn.py:19-33, dqn.py, tfdqn.py
build the same AtariEnv + ReplayBuffer + UniformSamplingDistribution triple).  Extras of this build: the synthetic
environment (no ROMs in the image), n lockstep environments on host worker processes, the prioritized sampler."""
import numpy as np

from slimdqn.sample_collection.replay_buffer import ReplayBuffer
from slimdqn.sample_collection.samplers import PrioritizedSamplingDistribution, UniformSamplingDistribution


def env_spec(p: dict) -> dict:
    """Picklable description of one environment (environments/vector.py: build_env): module, class, arguments, per-environment seed."""
    game = p["experiment_name"].split("_")[-1]  # experiment names end in _<Game> (experiments/atari/isdqn.py:21)
    if p["env_backend"] == "synthetic":
        return dict(module="slimdqn.environments.synthetic", **{"class": "SyntheticAtariEnv"}, kwargs=dict(name=game), seed_kw="seed",
                    seed0=p["seed"], seed_step=1000)
    return dict(module="slimdqn.environments.atari", **{"class": "AtariEnv"}, kwargs=dict(name=game), seed_kw=None)


def make_environment(p: dict):
    """One environment (the reference's loop), or a VectorEnv of ``-nenvs`` environments -- in this process, or on
    ``-nworkers`` host worker processes (started here: before this process's first GPU call, and never touching a GPU)."""
    from slimdqn.environments.vector import VectorEnv, build_env

    spec = env_spec(p)
    if p["n_envs"] <= 1:
        return build_env(spec, 0)
    if p.get("n_env_workers", 0) > 0:
        return VectorEnv(make_env=spec, n_envs=p["n_envs"], n_workers=p["n_env_workers"], horizon=p["horizon"])
    return VectorEnv([build_env(spec, i) for i in range(p["n_envs"])], horizon=p["horizon"])


def make_replay(p: dict, prioritized: bool = False) -> ReplayBuffer:
    sampler = PrioritizedSamplingDistribution(p["seed"], p["replay_buffer_capacity"]) if prioritized else UniformSamplingDistribution(p["seed"])
    return ReplayBuffer(
        sampling_distribution=sampler,
        max_capacity=p["replay_buffer_capacity"],
        batch_size=p["batch_size"],
        update_horizon=p["update_horizon"],
        gamma=p["gamma"],
        clipping=lambda x: np.clip(x, -1, 1),
        stack_size=4,
        compress=True,
    )


def seeds(p: dict):
    rng = np.random.default_rng(p["seed"])
    return (int(s) for s in rng.integers(0, 2**31 - 1, size=2))  # (network init, training loop)
