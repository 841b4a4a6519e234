"""GPU tests of the drop-in agent surface (slimdqn.networks.isdqn.iSDQN on the HIP engine) and of the
trainer loop: cadence of update_online_params / update_target_params (isdqn.py:55-80), parity of a short
training run against the oracle agent fed by the oracle replay on the same seed, reference-layout batches,
and the experiments/atari/isdqn.py entry point end to end on the synthetic environment."""
import json
import os
import pickle

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

FEATS = (8, 12, 16, 24)


def _agents(K=3, A=5, B=8, T=6, utd=2, lr=2e-4, n=1):
    from oracle.isdqn import iSDQN as Oracle
    from slimdqn.networks.isdqn import iSDQN
    from tests.gpu_helpers import perturbed_params

    params = perturbed_params(4, (84, 84, 4), FEATS, "cnn", (1 + K) * A, True)
    hip = iSDQN(0, (84, 84, 4), A, K, list(FEATS), True, False, "cnn", lr, 0.99, n, utd, T, adam_eps=1.5e-4, batch_size=B)
    hip._engine.import_flax(params)
    ora = Oracle(0, (84, 84, 4), A, K, list(FEATS), True, False, "cnn", lr, 0.99, n, utd, T, adam_eps=1.5e-4, params=params)
    return hip, ora


def test_training_run_matches_oracle_agent_and_replay():
    from oracle.replay_buffer import ReplayBuffer as ORB, TransitionElement as OT
    from oracle.samplers import UniformSamplingDistribution as OU
    from slimdqn.sample_collection.replay_buffer import ReplayBuffer, TransitionElement
    from slimdqn.sample_collection.samplers import UniformSamplingDistribution

    K, A, B, T, utd = 3, 5, 8, 6, 2
    hip, ora = _agents(K, A, B, T, utd)
    rb = ReplayBuffer(UniformSamplingDistribution(3), B, 100, update_horizon=1, gamma=0.99)
    orb = ORB(OU(3), B, 100, update_horizon=1, gamma=0.99)
    rng = np.random.default_rng(0)
    logs_h, logs_o = [], []
    for step in range(1, 41):
        obs = rng.integers(0, 256, (84, 84), dtype=np.uint8)
        a, r, term = int(rng.integers(0, A)), float(rng.choice([-1.0, 0.0, 1.0])), bool(rng.random() < 0.05)
        rb.add(TransitionElement(obs, a, r, term, term))
        orb.add(OT(obs, a, r, term, term))
        if step > 12:
            hip.update_online_params(step, rb)
            ora.update_online_params(step, orb)
            uh, lh = hip.update_target_params(step)
            uo, lo = ora.update_target_params(step)
            assert uh == uo
            if uh:
                logs_h.append(lh)
                logs_o.append(lo)
    assert len(logs_h) >= 4
    for lh, lo in zip(logs_h, logs_o):
        assert lh.keys() == lo.keys()
        for k in lh:
            # a dozen Adam steps on 8-sample batches: trajectories of two fp32-class implementations drift apart
            # by a few 1e-3 (Adam normalises near-zero gradients to +-lr); the first log is held to 1e-3
            tol = 1e-3 if lh is logs_h[0] else 5e-3
            assert abs(lh[k] - lo[k]) < tol * max(1.0, abs(lo[k])), (k, lh[k], lo[k])
    got = hip.get_model()["params"]["params"]
    exp = ora.get_model()["params"]["params"]
    for mod in exp:
        for leaf in exp[mod]:
            assert got[mod][leaf].shape == exp[mod][leaf].shape
            assert np.abs(got[mod][leaf] - exp[mod][leaf]).max() < 1e-3, (mod, leaf)


def test_reference_layout_batches_and_functional_signatures():
    """learn_on_batch / loss_on_batch accept ReplayElement-style batches with (B,84,84,4) uint8 stacks and a
    Flax-layout pytree, like the reference's jitted methods (isdqn.py:82-103)."""
    from oracle.replay_buffer import ReplayElement

    K, A, B = 3, 5, 8
    hip, ora = _agents(K, A, B)
    rng = np.random.default_rng(1)
    s = ReplayElement(state=rng.integers(0, 256, (B, 84, 84, 4), dtype=np.uint8), action=rng.integers(0, A, B),
                      reward=rng.normal(size=B), next_state=rng.integers(0, 256, (B, 84, 84, 4), dtype=np.uint8),
                      is_terminal=rng.integers(0, 2, B))
    loss_h, (per_head_h, _) = hip.loss_on_batch(hip.params, s)
    loss_o, (per_head_o, _) = ora.loss_on_batch(ora.params, s)
    assert abs(float(loss_h) - float(loss_o)) < 1e-3 * max(1.0, abs(float(loss_o)))
    np.testing.assert_allclose(per_head_h.cpu().numpy(), per_head_o.detach().numpy(), atol=1e-3)
    # a pytree handed in from outside is honoured
    tree = ora.get_model()["params"]
    loss_t, _ = hip.loss_on_batch(tree, s)
    assert abs(float(loss_t) - float(loss_o)) < 1e-3 * max(1.0, abs(float(loss_o)))
    # ... in every nesting a reference user can hold: the model pickle {"params": variables} (isdqn.py:137-138) after a pickle
    # round trip, the variables dict {"params": tree} (the reference's agent.params), the bare tree
    model = pickle.loads(pickle.dumps(ora.get_model()))
    assert set(model) == {"params"} and set(model["params"]) == {"params"} and "Conv_0" in model["params"]["params"]
    for handed in (model, model["params"], model["params"]["params"]):
        loss_p, _ = hip.loss_on_batch(handed, s)
        assert float(loss_p) == float(loss_t)
    assert set(hip.get_model()) == {"params"} and set(hip.get_model()["params"]) == {"params"}
    p, st, losses = hip.learn_on_batch(hip.params, hip.optimizer_state, s)
    assert p is hip.params and losses.shape == (K,)
    # shift on an external copy leaves the agent's parameters alone (shift_params is functional in the reference)
    before = hip.get_model()["params"]["params"]["Dense_1"]["bias"].copy()
    shifted = hip.shift_params(hip.params.clone())
    np.testing.assert_array_equal(hip.get_model()["params"]["params"]["Dense_1"]["bias"], before)
    np.testing.assert_array_equal(shifted["params"]["Dense_1"]["bias"][:-A], before[A:])
    # compute_target formula (test_isdqn.py:51-63)
    nq = torch.randn(K, A)
    one = ReplayElement(None, 0, 0.7, None, 0)
    tgt = hip.compute_target(one, nq)
    np.testing.assert_allclose(tgt.numpy(), 0.7 + 0.99 * nq.max(-1).values.numpy(), rtol=1e-6)


def test_best_action_follows_head_choice():
    K, A, B = 3, 5, 8
    hip, ora = _agents(K, A, B)
    state = np.random.default_rng(2).integers(0, 256, (84, 84, 4)).astype(np.float32)  # env.state is float32
    q = hip.q_values(hip.params, state)
    for idx in range(K):
        assert hip.best_action(hip.params, state, key=idx) == int(np.argmax(q[1 + idx]))
        assert hip.best_action(hip.params, state, key=idx) == ora.best_action(ora.params, state.astype(np.uint8), idx)
    assert 0 <= hip.best_action(hip.params, state) < A


def test_acting_path_frame_ring_and_weight_mirror_reuse():
    """The acting path keeps the frame stack on the device as a ring (one 7 KB frame uploaded per step) and -- on a TRUSTED engine
    (trust_mirror, set here; the default rebuilds in every call) -- reuses the
    workspace's pre-split weights while nothing has written the parameters: both must be invisible -- every action equals
    the argmax of a fresh full forward, across rolls, a reset, a learn step, a torch-side parameter write and a head shift."""
    from slimdqn._engine import QNetEngine
    from slimdqn.environments.synthetic import SyntheticAtariEnv
    from slimdqn.sample_collection.replay_buffer import ReplayBuffer, TransitionElement
    from slimdqn.sample_collection.samplers import UniformSamplingDistribution

    K, A, B = 3, 5, 8
    hip, _ = _agents(K, A, B)
    eng = hip._engine
    eng.trust_mirror = True
    # the checker: a second engine (own workspace, own weight mirror) run on the agent's parameter buffer
    chk = QNetEngine((84, 84, 4), A, 1 + K, FEATS, "cnn", True, B, gamma_n=0.99, learning_rate=2e-4, adam_eps=1.5e-4)
    env = SyntheticAtariEnv("Synthetic", n_actions=A, seed=3, episode_length=9)
    env.reset()
    rb = ReplayBuffer(UniformSamplingDistribution(3), B, 100, update_horizon=1, gamma=0.99)
    rng = np.random.default_rng(0)
    reused = fulls = 0
    for t in range(40):
        state = env.state
        idx = t % K
        shifts_before = None if hip._ring is None else hip._ring["shifts"]
        was_current = eng._mirror_is_current(None)
        action = hip.best_action(hip.params, state, key=idx)
        reused += was_current
        fulls += hip._ring["shifts"] == 0
        q = chk.forward(n_rows=1, params=eng.params, **_planes(state, eng.device)).cpu().numpy().reshape(1 + K, A)
        assert action == int(np.argmax(q[1 + idx])), t
        obs = env.observation
        reward, terminal = env.step(action)
        rb.add(TransitionElement(obs, action, reward, terminal, terminal))
        if terminal:
            env.reset()
        if t >= 12 and t % 4 == 0:
            hip.update_online_params(t, rb)  # data_to_update = 2: a learn step
        if t == 25:
            eng.params.mul_(1.01)  # a torch-side write: the mirror must be rebuilt
            assert not eng._mirror_is_current(None)
        if t == 30:
            hip.shift_params(hip.params)
            assert not eng._mirror_is_current(None)
    assert reused >= 20 and 3 <= fulls <= 8  # most steps reuse the mirror; full uploads only at episode starts


def test_default_acting_sees_writes_the_version_counter_cannot():
    """Default engines rebuild the weight mirror in every acting call: a write through `.data` (no version bump) between two
    one-frame-shift steps -- the hipGraph acting path -- must change the action exactly as a fresh full forward says."""
    from slimdqn._engine import QNetEngine
    from slimdqn.environments.synthetic import SyntheticAtariEnv

    K, A, B = 3, 5, 8
    hip, _ = _agents(K, A, B)
    eng = hip._engine
    assert not eng.trust_mirror
    chk = QNetEngine((84, 84, 4), A, 1 + K, FEATS, "cnn", True, B, gamma_n=0.99, learning_rate=2e-4, adam_eps=1.5e-4)
    env = SyntheticAtariEnv("Synthetic", n_actions=A, seed=5, episode_length=50)
    env.reset()
    gen = torch.Generator(device="cpu").manual_seed(0)
    for t in range(24):
        state = env.state
        if t >= 4 and t % 2 == 0:  # a new random head matrix behind torch's back
            v = eng.params._version
            noise = torch.randn(eng.params.shape, generator=gen).to(eng.device)
            eng.params.data.add_(0.05 * noise)
            assert eng.params._version == v
        action = hip.best_action(hip.params, state, key=t % K)
        q = chk.forward(n_rows=1, params=eng.params, **_planes(state, eng.device)).cpu().numpy().reshape(1 + K, A)
        assert action == int(np.argmax(q[1 + t % K])), t
        env.step(action)
    assert hip._ring is not None and hip._ring["shifts"] > 10  # the one-frame-shift graph path ran


def _planes(state, device):
    s = np.asarray(state).astype(np.uint8)
    h, w, stack = s.shape
    fr = torch.from_numpy(np.ascontiguousarray(np.moveaxis(s, -1, 0)).reshape(stack, h * w)).to(device)
    return dict(frames=fr, frame_stride=h * w, frame_ids=torch.arange(stack, dtype=torch.int32, device=device))


def test_batched_best_actions_match_single_and_oracle():
    K, A, B = 3, 5, 8
    hip, ora = _agents(K, A, B)
    rng = np.random.default_rng(7)
    n = 2 * B  # the most rows one workspace takes
    states = rng.integers(0, 256, (n, 84, 84, 4), dtype=np.uint8)
    heads = rng.integers(0, K, n)
    got = hip.best_actions(hip.params, states, key=heads)
    assert got.shape == (n,) and got.dtype == np.int32
    for i in range(n):
        assert got[i] == hip.best_action(hip.params, states[i], key=int(heads[i]))
        assert got[i] == ora.best_action(ora.params, states[i], int(heads[i]))
    same_stream = hip.best_actions(hip.params, states, key=np.random.default_rng(1))
    np.testing.assert_array_equal(same_stream, hip.best_actions(hip.params, states, key=np.random.default_rng(1).integers(0, K, n)))
    with pytest.raises(AssertionError):  # ISDQN_ERR_SHAPE: more rows than the workspace holds
        hip.best_actions(hip.params, np.zeros((2 * B + 1, 84, 84, 4), np.uint8), key=0)


def test_vector_collection_round():
    """collect_vector_samples: pipelined rounds (collect the previous round, act, start the next), one batched forward for the
    greedy environments read from the planar host block, every transition into its own n-step stream."""
    from slimdqn.environments.synthetic import SyntheticAtariEnv
    from slimdqn.environments.vector import VectorEnv
    from slimdqn.sample_collection.replay_buffer import ReplayBuffer
    from slimdqn.sample_collection.samplers import UniformSamplingDistribution
    from slimdqn.sample_collection.utils import collect_vector_samples

    K, A, B = 3, 5, 8
    hip, ora = _agents(K, A, B)
    venv = VectorEnv([SyntheticAtariEnv("Synthetic", n_actions=A, seed=10 + i, episode_length=7 + i) for i in range(4)], horizon=1000)
    venv.reset()
    assert venv.states.shape == (4, 84, 84, 4) and venv.states.dtype == np.uint8
    # acting from the planar block == acting from (n, h, w, stack) states == the oracle's argmax
    heads = np.array([0, 2, 1], np.int32)
    rows = np.array([3, 0, 2])
    got = hip.best_actions_planes(hip.params, venv.planes, rows, key=heads)
    np.testing.assert_array_equal(got, hip.best_actions(hip.params, venv.states[rows], key=heads))
    for r, h, a in zip(rows, heads, got):
        assert a == ora.best_action(ora.params, venv.states[r], int(h))
    rb = ReplayBuffer(UniformSamplingDistribution(3), B, 500, update_horizon=3, gamma=0.99)
    rng = np.random.default_rng(0)
    ends, n_steps = 0, 0
    for t in range(31):
        out = collect_vector_samples(rng, venv, hip, rb, {"horizon": 1000}, lambda step: 0.3, n_steps)
        assert len(out) == (0 if t == 0 else 4)  # the first call only starts a round
        n_steps += len(out)
        ends += sum(e for _, e in out)
    assert ends >= 8 and len(rb._trajectories) == 4
    assert rb.add_count > 80
    batch = rb.sample()
    assert batch.state.shape == (B, 84, 84, 4)


@pytest.mark.parametrize("prioritized", [False, True])
def test_entry_point_end_to_end_on_synthetic_env(tmp_path, prioritized):
    from experiments.atari.isdqn import run

    argv = ["-en", "smoke_Synthetic", "-s", "1", "-dw", "-f", "8", "8", "8", "16", "-rbc", "200", "-bs", "8", "-n", "3" if prioritized else "1",
            "-horizon", "50", "-at", "cnn", "-ne", "2", "-ntspe", "60", "-utd", "4", "-nis", "20", "-ed", "100", "-nbi", "2", "-ln",
            "-tuf", "16", "-env", "synthetic"] + (["-per"] if prioritized else [])
    gathered = run(argv, root=str(tmp_path))
    assert len(gathered) == 2 and gathered[0].shape == (1, 4)  # (avg_return, avg_length, n_training_steps, env steps/s) of one replica
    out = tmp_path / "atari" / "exp_output" / "smoke_Synthetic"
    params = json.load(open(out / "parameters.json"))
    assert params["shared_parameters"]["features"] == [8, 8, 8, 16] and params["isdqn"]["n_bellman_iterations"] == 2
    res = json.load(open(out / "isdqn" / "episode_returns_and_lengths" / "1.json"))
    assert len(res["episode_returns"]) == 2
    model = pickle.load(open(out / "isdqn" / "models" / "1", "rb"))["params"]
    assert model["params"]["Conv_0"]["kernel"].shape == (8, 8, 4, 8)
    assert model["params"]["Dense_1"]["kernel"].shape == (16, 3 * 9)
    with pytest.raises(AssertionError):  # same seed again: refused (experiments/base/utils.py:46-51)
        run(argv, root=str(tmp_path))


def test_entry_point_with_vectorised_environments(tmp_path):
    from experiments.atari.isdqn import run

    argv = ["-en", "vec_Synthetic", "-s", "1", "-dw", "-f", "8", "8", "8", "16", "-rbc", "300", "-bs", "8", "-n", "3", "-horizon", "40",
            "-at", "cnn", "-ne", "2", "-ntspe", "90", "-utd", "4", "-nis", "30", "-ed", "100", "-nbi", "2", "-ln", "-tuf", "16",
            "-env", "synthetic", "-nenvs", "3", "-per"]
    gathered = run(argv, root=str(tmp_path))
    assert len(gathered) == 2 and gathered[0].shape == (1, 4)
    res = json.load(open(tmp_path / "atari" / "exp_output" / "vec_Synthetic" / "isdqn" / "episode_returns_and_lengths" / "1.json"))
    assert len(res["episode_returns"]) == 2 and all(len(r) >= 1 for r in res["episode_returns"])
    assert all(l == 40 for epoch in res["episode_lengths"] for l in epoch)  # horizon-truncated episodes, whole ones only


def test_entry_point_with_environment_worker_processes(tmp_path):
    """-nenvs 8 -nworkers 2: the emulators on host worker processes, gradient steps of a round issued as one graph replay."""
    from experiments.atari.isdqn import run

    argv = ["-en", "work_Synthetic", "-s", "2", "-dw", "-f", "8", "8", "8", "16", "-rbc", "400", "-bs", "8", "-n", "1", "-horizon", "25",
            "-at", "cnn", "-ne", "2", "-ntspe", "160", "-utd", "4", "-nis", "40", "-ed", "100", "-nbi", "2", "-ln", "-tuf", "64",
            "-env", "synthetic", "-nenvs", "8", "-nworkers", "2"]
    gathered = run(argv, root=str(tmp_path))
    assert len(gathered) == 2 and gathered[0].shape == (1, 4)
    res = json.load(open(tmp_path / "atari" / "exp_output" / "work_Synthetic" / "isdqn" / "episode_returns_and_lengths" / "2.json"))
    assert len(res["episode_returns"]) == 2 and all(len(r) >= 8 for r in res["episode_returns"])
    assert all(l == 25 for epoch in res["episode_lengths"] for l in epoch)


def test_fc_agent_acting_single_and_batched_match_oracle():
    """BASELINE config 1's shape (LunarLander-style fc [100, 100], K = 1 is the DQN head count; here K = 2 heads + target): the
    acting path of the fc architecture -- single observations (no frame ring) and batched rows -- against the oracle argmax."""
    from oracle.isdqn import iSDQN as Oracle
    from slimdqn.networks.isdqn import iSDQN
    from tests.gpu_helpers import perturbed_params

    K, A, B, feats, obs = 2, 4, 32, (100, 100), (8,)
    params = perturbed_params(9, obs, feats, "fc", (1 + K) * A, True)
    hip = iSDQN(0, obs, A, K, list(feats), True, False, "fc", 1e-3, 0.99, 1, 1, 4, adam_eps=1.5e-4, batch_size=B)
    hip._engine.import_flax(params)
    ora = Oracle(0, obs, A, K, list(feats), True, False, "fc", 1e-3, 0.99, 1, 1, 4, adam_eps=1.5e-4, params=params)
    rng = np.random.default_rng(4)
    states = rng.normal(size=(2 * B,) + obs).astype(np.float32)
    heads = rng.integers(0, K, 2 * B)
    got = hip.best_actions(hip.params, states, key=heads)
    for i in range(2 * B):
        want = ora.best_action(ora.params, states[i], int(heads[i]))
        assert got[i] == want
        if i < 6:
            assert hip.best_action(hip.params, states[i], key=int(heads[i])) == want


def test_lunar_lander_shape_training_run_matches_oracle_agent_and_replay():
    """BASELINE configs[0] end to end on the device: LunarLander-shaped (8,) float32 observations stored in the device replay
    (the bytes of the vectors as 1 x 32 "frames", stack size 1), sampled, and learned on by the fc [100, 100] K = 1 agent --
    against the oracle agent fed by the oracle replay on the same seed (the oracle's (8, 1) stacks squeezed as the
    reference network does, architectures/dqn.py:93)."""
    from oracle.isdqn import iSDQN as Oracle
    from oracle.replay_buffer import ReplayBuffer as ORB, TransitionElement as OT
    from oracle.samplers import UniformSamplingDistribution as OU
    from slimdqn.networks.isdqn import iSDQN
    from slimdqn.sample_collection.replay_buffer import ReplayBuffer, TransitionElement
    from slimdqn.sample_collection.samplers import UniformSamplingDistribution
    from tests.gpu_helpers import perturbed_params

    K, A, B, T, utd, feats, obs_dim = 1, 4, 32, 8, 2, (100, 100), (8,)
    params = perturbed_params(6, obs_dim, feats, "fc", (1 + K) * A, True)
    hip = iSDQN(0, obs_dim, A, K, list(feats), True, False, "fc", 3e-4, 0.99, 1, utd, T, adam_eps=1e-5, batch_size=B)
    hip._engine.import_flax(params)
    ora = Oracle(0, obs_dim, A, K, list(feats), True, False, "fc", 3e-4, 0.99, 1, utd, T, adam_eps=1e-5, params=params)
    rb = ReplayBuffer(UniformSamplingDistribution(3), B, 200, stack_size=1, update_horizon=1, gamma=0.99)
    orb = ORB(OU(3), B, 200, stack_size=1, update_horizon=1, gamma=0.99)
    rng = np.random.default_rng(0)
    logs_h, logs_o = [], []
    for step in range(1, 81):
        obs = rng.normal(size=8).astype(np.float32)
        a, r, term = int(rng.integers(0, A)), float(rng.normal()), bool(rng.random() < 0.05)
        rb.add(TransitionElement(obs, a, r, term, term))
        orb.add(OT(obs, a, r, term, term))
        if step == 40:  # the device batch is the oracle batch: same keys, same float bits
            b, ob = rb.sample(), orb.sample()
            np.testing.assert_array_equal(b.state.cpu().numpy(), np.asarray(ob.state).reshape(B, 8))
            np.testing.assert_array_equal(b.next_state.cpu().numpy(), np.asarray(ob.next_state).reshape(B, 8))
            np.testing.assert_array_equal(b.action.cpu().numpy(), np.asarray(ob.action))
        if step > 40:
            hip.update_online_params(step, rb)
            ora.update_online_params(step, orb)
            uh, lh = hip.update_target_params(step)
            uo, lo = ora.update_target_params(step)
            assert uh == uo
            if uh:
                logs_h.append(lh)
                logs_o.append(lo)
    assert len(logs_h) >= 4
    for lh, lo in zip(logs_h, logs_o):
        for k in lh:
            tol = 1e-3 if lh is logs_h[0] else 5e-3
            assert abs(lh[k] - lo[k]) < tol * max(1.0, abs(lo[k])), (k, lh[k], lo[k])


def test_lunar_lander_entry_point_end_to_end(tmp_path):
    from experiments.lunar_lander.isdqn import run

    argv = ["-en", "test_ll", "-s", "1", "-dw", "-f", "100", "100", "-rbc", "500", "-bs", "32", "-n", "1", "-horizon", "60", "-at", "fc",
            "-ne", "2", "-ntspe", "150", "-utd", "1", "-nis", "50", "-ed", "100", "-nbi", "1", "-ln", "-tuf", "20", "-env", "synthetic"]
    gathered = run(argv, root=str(tmp_path))
    assert len(gathered) == 2 and gathered[0].shape == (1, 4)
    out = tmp_path / "lunar_lander" / "exp_output" / "test_ll"
    params = json.load(open(out / "parameters.json"))
    assert params["shared_parameters"]["features"] == [100, 100] and params["isdqn"]["n_bellman_iterations"] == 1
    model = pickle.load(open(out / "isdqn" / "models" / "1", "rb"))["params"]
    assert model["params"]["Dense_0"]["kernel"].shape == (8, 100) and model["params"]["Dense_2"]["kernel"].shape == (100, 2 * 4)


@pytest.mark.parametrize("algo", ["dqn", "tfdqn", "isdqn"])
def test_reference_atari_test_settings_on_the_synthetic_environment(tmp_path, algo):
    """The reference's tests/test_atari.py:8-61 (and its isdqn / tfdqn siblings in tests/test_launch_job.py): the entry point with
    features 2 3 1 15, batch 3, capacity 100, 10 training steps, an update every 3 steps.  ALE is not installed, so the emulator is the
    seeded synthetic one (-env synthetic); everything behind it -- replay, one- to three-channel conv layers padded to 8, a 15-wide
    dense layer, target updates every 3 steps -- is the device path."""
    import importlib

    run = importlib.import_module(f"experiments.atari.{algo}").run
    argv = ["--experiment_name", f"_test_{algo}_Pong", "--seed", "1", "--disable_wandb", "--features", "2", "3", "1", "15",
            "--replay_buffer_capacity", "100", "--batch_size", "3", "--update_horizon", "1", "--gamma", "0.99", "--learning_rate", "1e-4",
            "--horizon", "10", "--n_epochs", "1", "--n_training_steps_per_epoch", "10", "--data_to_update", "3",
            "--target_update_frequency", "3", "--n_initial_samples", "3", "--epsilon_end", "0.01", "--epsilon_duration", "4",
            "--architecture_type", "cnn", "-env", "synthetic"]
    run(argv, root=str(tmp_path))
    out = tmp_path / "atari" / "exp_output" / f"_test_{algo}_Pong" / algo
    res = json.load(open(out / "episode_returns_and_lengths" / "1.json"))
    assert len(res["episode_returns"]) == 1
    model = pickle.load(open(out / "models" / "1", "rb"))["params"]
    assert model["params"]["Conv_2"]["kernel"].shape == (3, 3, 3, 1) and model["params"]["Dense_0"]["kernel"].shape == (11 * 11 * 1, 15)
    assert all(np.isfinite(v).all() for leaves in model["params"].values() for v in leaves.values())


@pytest.mark.parametrize("algo", ["dqn", "tfdqn", "isdqn"])
def test_reference_lunar_lander_test_settings(tmp_path, algo):
    """The reference's tests/test_lunar_lander.py:8-59: experiments/lunar_lander/<algo>.py with features 25 15, batch 3, capacity 100,
    10 training steps, fc.  (gymnasium is not installed: the seeded synthetic LunarLander stands in for the emulator.)"""
    import importlib

    run = importlib.import_module(f"experiments.lunar_lander.{algo}").run
    argv = ["--experiment_name", f"_test_{algo}", "--seed", "1", "--disable_wandb", "--features", "25", "15", "--replay_buffer_capacity", "100",
            "--batch_size", "3", "--update_horizon", "1", "--gamma", "0.99", "--learning_rate", "1e-4", "--horizon", "10", "--n_epochs", "1",
            "--n_training_steps_per_epoch", "10", "--data_to_update", "3", "--target_update_frequency", "3", "--n_initial_samples", "3",
            "--epsilon_end", "0.01", "--epsilon_duration", "4", "--architecture_type", "fc", "-env", "synthetic"]
    run(argv, root=str(tmp_path))
    out = tmp_path / "lunar_lander" / "exp_output" / f"_test_{algo}" / algo
    assert len(json.load(open(out / "episode_returns_and_lengths" / "1.json"))["episode_returns"]) == 1
    model = pickle.load(open(out / "models" / "1", "rb"))["params"]
    assert model["params"]["Dense_0"]["kernel"].shape == (8, 25) and model["params"]["Dense_1"]["kernel"].shape == (25, 15)
    assert all(np.isfinite(v).all() for leaves in model["params"].values() for v in leaves.values())
