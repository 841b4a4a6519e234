"""Host-parallel vectorised environments (slimdqn/environments/vector.py; no GPU): worker processes against the in-process
twin, step for step -- the reference steps one environment at a time (slimdqn/sample_collection/utils.py:21-43), so the
contract is that every environment of the vector behaves exactly as if it were stepped alone."""
import sys

import numpy as np

SPEC = dict(module="slimdqn.environments.synthetic", **{"class": "SyntheticAtariEnv"}, kwargs=dict(name="X", n_actions=6, episode_length=23),
            seed_kw="seed", seed0=5, seed_step=1000)


def test_workers_equal_the_in_process_twin_and_never_load_torch():
    from slimdqn.environments.synthetic import SyntheticAtariEnv
    from slimdqn.environments.vector import VectorEnv

    n = 7  # uneven deal over 3 workers
    had_torch = "torch" in sys.modules
    v = VectorEnv(make_env=SPEC, n_envs=n, n_workers=3, horizon=17)
    try:
        assert v.n_workers == 3 and len(v) == n and (v.n_actions, v.state_height, v.n_stacked_frames) == (6, 84, 4)
        assert had_torch or "torch" not in sys.modules  # building a worker vector does not pull torch into this process either
        s = VectorEnv([SyntheticAtariEnv("X", 6, seed=5 + 1000 * i, episode_length=23) for i in range(n)], horizon=17)
        s.reset()
        np.testing.assert_array_equal(v.planes, s.planes)
        rng = np.random.default_rng(0)
        ends = absorbs = 0
        for r in range(60):
            a = rng.integers(0, 6, n)
            v.step_async(a)
            s.step_async(a)
            got, want = v.step_wait(), s.step_wait()
            for x, y in zip(got, want):
                np.testing.assert_array_equal(x, y)
            np.testing.assert_array_equal(v.planes, s.planes)
            ends += int(want[3].sum())
            absorbs += int(want[2].sum())
        assert ends >= 3 * n and absorbs == 0  # horizon (17) truncates before the synthetic terminal (23): episode_end without absorbing
        # observation = the newest frame BEFORE the step; state layout (n, h, w, stack)
        before = s.planes[:, -1].copy()
        a = rng.integers(0, 6, n)
        v.step(a)
        obs, _, _, ended = s.step(a)
        np.testing.assert_array_equal(obs.reshape(n, -1), before)
        assert s.states.shape == (n, 84, 84, 4)
        keep = ~ended
        np.testing.assert_array_equal(s.planes[keep, -2], before[keep])  # the stack rolled by one frame
        v.reset()
        s.reset()
        np.testing.assert_array_equal(v.planes, s.planes)
    finally:
        v.close()
    assert v.n_workers == 0


def test_terminal_before_horizon_is_absorbing():
    from slimdqn.environments.synthetic import SyntheticAtariEnv
    from slimdqn.environments.vector import VectorEnv

    s = VectorEnv([SyntheticAtariEnv("X", 4, seed=1, episode_length=5)], horizon=1000)
    s.reset()
    flags = [s.step([0])[2:] for _ in range(10)]
    assert [bool(a[0]) for a, _ in flags] == [False] * 4 + [True] + [False] * 4 + [True]
    assert [bool(e[0]) for _, e in flags] == [bool(a[0]) for a, _ in flags]


def _helper_spec(cls):
    import os

    here = os.path.dirname(os.path.abspath(__file__))
    os.environ["PYTHONPATH"] = here + os.pathsep + os.environ.get("PYTHONPATH", "")  # the workers' import path
    if here not in sys.path:
        sys.path.insert(0, here)  # the parent builds one environment itself (action count, frame geometry)
    return dict(module="helpers.chatty_env", **{"class": cls}, kwargs=dict(name="X", n_actions=6, episode_length=23), seed_kw="seed", seed0=5,
                seed_step=1000)


def test_worker_protocol_survives_an_environment_that_prints_on_stdout():
    """The emulator stack's own stdout (ALE / gymnasium banners, warnings, raw writes to descriptor 1) must not reach the one-byte
    protocol: the worker keeps a private duplicate of the pipe and points descriptor 1 at stderr (environments/_worker.py)."""
    from slimdqn.environments.synthetic import SyntheticAtariEnv
    from slimdqn.environments.vector import VectorEnv

    n = 3
    v = VectorEnv(make_env=_helper_spec("ChattyEnv"), n_envs=n, n_workers=2, horizon=17)
    try:
        s = VectorEnv([SyntheticAtariEnv("X", 6, seed=5 + 1000 * i, episode_length=23) for i in range(n)], horizon=17)
        s.reset()
        rng = np.random.default_rng(1)
        for _ in range(20):
            a = rng.integers(0, 6, n)
            for x, y in zip(v.step(a), s.step(a)):
                np.testing.assert_array_equal(x, y)
        np.testing.assert_array_equal(v.planes, s.planes)
    finally:
        v.close()


def test_a_hung_worker_raises_instead_of_hanging_the_trainer(monkeypatch):
    import pytest

    from slimdqn.environments.vector import VectorEnv

    monkeypatch.setattr(VectorEnv, "WORKER_TIMEOUT_S", 2.0)
    v = VectorEnv(make_env=_helper_spec("HungEnv"), n_envs=2, n_workers=1, horizon=17)
    try:
        with pytest.raises(RuntimeError, match="did not answer"):
            v.step(np.zeros(2, dtype=np.int64))
    finally:
        v.close()
