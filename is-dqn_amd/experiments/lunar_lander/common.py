"""Environment and replay wiring shared by the LunarLander entry points (experiments/lunar_lander/{isdqn,dqn,tfdqn}.py)."""
from experiments.atari.common import seeds  # noqa: F401  (re-exported: the (network, training) seed pair of an entry point)
from slimdqn.sample_collection.replay_buffer import ReplayBuffer
from slimdqn.sample_collection.samplers import UniformSamplingDistribution


def make_environment(p):
    """gymnasium's LunarLander-v3 behind the reference's wrapper (slimdqn/environments/lunar_lander.py:5-23), or the seeded synthetic
    stand-in (``-env synthetic``: gymnasium is absent from the build image)."""
    assert p["architecture_type"] == "fc", "LunarLander observations are vectors: -at fc"
    if p["env_backend"] == "synthetic":
        from slimdqn.environments.lunar_lander import SyntheticLunarLander

        return SyntheticLunarLander(seed=p["seed"])
    from slimdqn.environments.lunar_lander import LunarLander

    return LunarLander()


def make_replay(p):
    return ReplayBuffer(
        sampling_distribution=UniformSamplingDistribution(p["seed"]),
        max_capacity=p["replay_buffer_capacity"],
        batch_size=p["batch_size"],
        update_horizon=p["update_horizon"],
        gamma=p["gamma"],
        clipping=None,
        stack_size=1,
        compress=False,
    )
