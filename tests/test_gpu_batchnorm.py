"""GPU parity of the BatchNorm variants (slimdqn/networks/architectures/dqn.py:52-53, 59-60, 66-67, 73-74, 100-101; learn step
isdqn.py:82-103 with batch_norm=True) against the CPU oracle (oracle/network.py: parity unpinned, like the rest of the network
numerics -- the reference holds no numbers and flax is not installed), through the C ABI (csrc/batchnorm.h).

What is compared: the acting forward on the running averages; q-values / targets / per-head losses of the training-mode pass on
concat(state, next_state) within 1e-3; every leaf's first-step gradient (BatchNorm scale / bias included: they carry the part of the
gradient that reaches the next-state rows); the running averages a learn step leaves; parameters after Adam; three chained steps."""
import numpy as np
import pytest
import torch

from oracle.replay_buffer import ReplayElement
from tests.gpu_helpers import device_batch, make_frame_batch, make_pair

pytestmark = pytest.mark.gpu

CNN = [
    # obs, feats, K, A, B, layer_norm
    pytest.param(((84, 84, 4), (7, 9, 11, 13), 3, 5, 6, True), id="tiny-ln"),
    pytest.param(((84, 84, 4), (16, 20, 12, 24), 2, 3, 5, False), id="tiny-noln"),
    pytest.param(((84, 84, 4), (32, 64, 64, 512), 9, 9, 8, True), id="headline-arch-B8"),
    pytest.param(((44, 44, 2), (8, 16, 8, 32), 2, 4, 9, True), id="44x44x2-B9-ragged"),
]


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-12)


def _check_step(oracle, eng, batch, ref, n_steps=3, grad_tol=3e-3, later_loss_tol=1e-3, trajectory=True):
    """``trajectory=False`` (the impala torso: max-pool winners and ReLU masks of 2B small images -- a few decisions differ between two
    fp32-class forwards, and Adam moves an entry whose gradient changes sign by a full lr either way): the first step's update is
    checked against Adam applied to the path's OWN gradient instead of against the oracle's parameters, later losses loosely."""
    K = oracle.n_bellman_iterations
    # ---- training-mode pass: q-values, targets, losses (batch statistics of the 2B rows) ----
    o_q, o_t, o_td = oracle.loss_terms(oracle.params, ref)
    o_loss = o_td.mean(0).detach().numpy()
    stats_before = eng.export_batch_stats()
    losses = eng.loss_on_batch(batch).cpu().numpy()
    assert np.abs(eng.q_values.cpu().numpy() - o_q.detach().numpy()).max() < 1e-3
    assert np.abs(eng.targets.cpu().numpy() - o_t.detach().numpy()).max() < 1e-3
    assert np.abs(losses - o_loss).max() < 1e-3 * max(1.0, np.abs(o_loss).max())
    after_loss = eng.export_batch_stats()
    for m in stats_before:  # loss_on_batch leaves the running averages alone
        for n in stats_before[m]:
            np.testing.assert_array_equal(after_loss[m][n], stats_before[m][n])

    grad = torch.zeros_like(eng.params)
    p, st = oracle.params, oracle.optimizer_state
    for step in range(n_steps):
        o_grads, _ = oracle.grads(p, ref)
        p, st, o_losses = oracle.learn_on_batch(p, st, ref)
        p0 = eng.params.clone()
        losses = eng.learn_on_batch(batch, grad_out=grad).cpu().numpy()
        assert np.abs(losses - o_losses).max() < (1e-3 if step == 0 else later_loss_tol) * max(1.0, np.abs(o_losses).max()), f"step {step}"
        if step == 0:
            g = eng.internal_to_flax_grads(grad)
            for mod in o_grads:
                for leaf in o_grads[mod]:
                    e = _rel(g[mod][leaf], o_grads[mod][leaf].numpy())
                    assert e < grad_tol, f"grad {mod}/{leaf}: rel err {e}"
            num = sum(float(np.sum((np.asarray(g[m][n], np.float64) - o_grads[m][n].numpy()) ** 2)) for m in o_grads for n in o_grads[m])
            den = sum(float(np.sum(o_grads[m][n].numpy().astype(np.float64) ** 2)) for m in o_grads for n in o_grads[m])
            assert num <= (3 * grad_tol) ** 2 * den, f"whole gradient, Euclidean: {(num / den) ** 0.5}"
            if not trajectory:  # optax.adam's first step on the path's own gradient: p -= lr * g / (|g| + eps), optimised tensors only
                want = p0 - eng.cfg.learning_rate * grad / (grad.abs() + eng.cfg.adam_eps)
                sel = torch.zeros_like(grad, dtype=torch.bool)
                for info in eng.infos:
                    if info.kind < 7:
                        sel[info.offset : info.offset + info.size] = True
                assert float((eng.params - want)[sel].abs().max()) < 2e-6
            pri = eng.priorities.cpu().numpy()
            exp = np.sqrt(o_td.detach().numpy().mean(1) + 1e-10)
            assert np.abs(pri - exp).max() < 1e-2 * max(1.0, exp.max())
        got_stats = eng.export_batch_stats()
        for m, l in oracle.batch_stats.items():  # ra = 0.99 ra + 0.01 batch (isdqn.py:87-88)
            for n, t in l.items():
                d = np.abs(got_stats[m][n] - t.numpy()).max()
                # (after the first update the two parameter sets differ by Adam's normalised steps: activations, and with
                # them the batch statistics, drift by ~1e-3)
                assert d < (2e-5 if step == 0 else max(2e-4, 0.1 * later_loss_tol)) * max(1.0, float(t.abs().max())), f"step {step}: running {n} of {m}: {d}"
    assert int(eng.adam_count.item()) == n_steps
    if not trajectory:
        return
    got = eng.export_flax()
    for mod in p:
        for leaf in p[mod]:
            d = np.abs(got[mod][leaf] - p[mod][leaf].numpy()).max()
            # three Adam steps of lr = 1e-3: an element whose gradient is ~0 can take its normalised steps with the other sign
            assert d < 3e-3, f"param {mod}/{leaf}: {d}"
    assert int(eng.adam_count.item()) == n_steps


@pytest.mark.parametrize("cfg", CNN)
def test_cnn_batchnorm_matches_the_oracle(cfg):
    obs, feats, K, A, B, ln = cfg
    oracle, eng, _ = make_pair(feats, K, A, B, obs=obs, layer_norm=ln, seed=3, batch_norm=True)
    h, w, stack = obs
    frames, ids, action, reward, terminal, ref = make_frame_batch(B, A, seed=11, h=h, w=w, stack=stack)
    batch = device_batch(eng, frames, ids, action, reward, terminal)

    # ---- acting forward: the running averages (isdqn.py:130) ----
    both = torch.cat((torch.tensor(ref.state), torch.tensor(ref.next_state)))
    q_run = oracle.apply(oracle.params, both, use_running_average=True).detach().numpy()
    flat_ids = np.concatenate([ids[:, :stack], ids[:, stack:]], 0).copy()
    q = eng.forward(frames=batch._keep[0], frame_stride=frames.shape[1], frame_ids=torch.from_numpy(flat_ids).cuda(), n_rows=2 * B)
    q = q.cpu().numpy().reshape(2 * B, 1 + K, A)
    assert np.abs(q - q_run).max() < 1e-3 * max(1.0, np.abs(q_run).max()), f"acting forward max err {np.abs(q - q_run).max()}"
    # one row alone gives the same values as inside the batch: nothing of the acting path depends on the other rows
    q1 = eng.forward(frames=batch._keep[0], frame_stride=frames.shape[1], frame_ids=torch.from_numpy(flat_ids[3:4].copy()).cuda(), n_rows=1)
    assert np.abs(q1.cpu().numpy().reshape(1 + K, A) - q[3]).max() < 1e-5

    _check_step(oracle, eng, batch, ref)


IMPALA = [
    pytest.param(((84, 84, 4), (8, 16, 8, 32), 2, 3, 8, True), id="84x84x4-ln-B8"),
    pytest.param(((36, 36, 2), (16, 8, 16, 24), 3, 4, 9, False), id="36x36x2-noln-B9"),
]


@pytest.mark.parametrize("cfg", IMPALA)
def test_impala_batchnorm_matches_the_oracle(cfg):
    """The impala torso with BatchNorm (dqn.py:29-30, 78-79, 86-88): a site on x / 255, one behind the ReLU of each of the six residual
    blocks ("Stack_s/BatchNorm_b"), one per feature behind the flatten, one behind Dense_0."""
    obs, feats, K, A, B, ln = cfg
    # (float64 oracle: per-feature statistics over 16-18 rows divide by small deviations, fp32 noise on the oracle's side would
    # eat a good part of the 1e-3 bar.  The gradient bound is direct -- no pinned ReLU / max-pool decisions as in
    # tests/test_gpu_impala.py, whose independent-oracle bound is 15 % Euclidean for the same reason -- hence 5e-2 of each leaf's
    # largest entry and 15 % of the whole vector; measured: <= 1.6e-2 per leaf)
    oracle, eng, params = make_pair(feats, K, A, B, arch="impala", obs=obs, layer_norm=ln, seed=6, batch_norm=True, dtype=torch.float64)
    assert "Stack_2/BatchNorm_1" in params and eng.export_batch_stats()["Stack_0/BatchNorm_0"]["var"].shape == params["Stack_0/BatchNorm_0"]["scale"].shape
    h, w, stack = obs
    frames, ids, action, reward, terminal, ref = make_frame_batch(B, A, seed=13, h=h, w=w, stack=stack)
    batch = device_batch(eng, frames, ids, action, reward, terminal)
    both = torch.cat((torch.tensor(ref.state), torch.tensor(ref.next_state)))
    q_run = oracle.apply(oracle.params, both, use_running_average=True).detach().numpy()
    flat_ids = np.concatenate([ids[:, :stack], ids[:, stack:]], 0).copy()
    q = eng.forward(frames=batch._keep[0], frame_stride=frames.shape[1], frame_ids=torch.from_numpy(flat_ids).cuda(), n_rows=2 * B)
    q = q.cpu().numpy().reshape(2 * B, 1 + K, A)
    assert np.abs(q - q_run).max() < 1e-3 * max(1.0, np.abs(q_run).max()), f"acting forward max err {np.abs(q - q_run).max()}"
    _check_step(oracle, eng, batch, ref, n_steps=2, grad_tol=5e-2, later_loss_tol=5e-2, trajectory=False)


FC = [
    pytest.param((8, (100, 100), 1, 4, 32, True), id="lunar-lander-100x100-ln"),
    pytest.param((6, (24, 40, 16), 3, 3, 10, False), id="three-hidden-noln-B10"),
]


@pytest.mark.parametrize("cfg", FC)
def test_fc_batchnorm_matches_the_oracle(cfg):
    d, feats, K, A, B, ln = cfg
    oracle, eng, _ = make_pair(feats, K, A, B, arch="fc", obs=(d,), layer_norm=ln, seed=5, adam_eps=1e-8, batch_norm=True)
    rng = np.random.default_rng(7)
    st, nx = rng.normal(size=(B, d)).astype(np.float32), rng.normal(size=(B, d)).astype(np.float32)
    action = rng.integers(0, A, B).astype(np.int32)
    reward = rng.normal(size=B).astype(np.float32)
    terminal = (rng.random(B) < 0.3).astype(np.uint8)
    ref = ReplayElement(state=st, action=action.astype(np.int64), reward=reward.astype(np.float64), next_state=nx,
                        is_terminal=terminal.astype(np.int64))
    dev = lambda a: torch.from_numpy(a).cuda()
    batch = eng.make_batch(state=dev(st), next_state=dev(nx), action=dev(action), reward=dev(reward), terminal=dev(terminal))
    q_run = oracle.apply(oracle.params, torch.tensor(st), use_running_average=True).detach().numpy()
    q = eng.forward(obs=dev(st), n_rows=B).cpu().numpy().reshape(B, 1 + K, A)
    assert np.abs(q - q_run).max() < 1e-3 * max(1.0, np.abs(q_run).max())
    _check_step(oracle, eng, batch, ref)


def test_batchnorm_learn_steps_are_bitwise_repeatable():
    feats, K, A, B = (16, 32, 32, 64), 3, 4, 16
    outs = []
    for _ in range(2):
        _, eng, _ = make_pair(feats, K, A, B, seed=9, batch_norm=True)
        frames, ids, action, reward, terminal, _ref = make_frame_batch(B, A, seed=4)
        batch = device_batch(eng, frames, ids, action, reward, terminal)
        for _s in range(3):
            eng.learn_on_batch(batch)
        outs.append((eng.params.cpu().numpy().copy(), eng.adam_v.cpu().numpy().copy(), eng.losses.cpu().numpy().copy()))
    for a, b in zip(*outs):
        np.testing.assert_array_equal(a, b)


def test_unbuilt_batchnorm_combinations_raise():
    _, eng, _ = make_pair((7, 9, 11, 13), 2, 3, 4, seed=1, batch_norm=True)
    frames, ids, action, reward, terminal, _ref = make_frame_batch(4, 3, seed=2)
    batch = device_batch(eng, frames, ids, action, reward, terminal)
    with pytest.raises(Exception):  # the DQN form (separate target parameters) does not exist with BatchNorm -- in the reference either
        eng.learn_on_batch_target(batch, eng.params.clone())
    with pytest.raises(Exception):
        eng.loss_on_batch_target(batch, eng.params.clone())
    # (the gradient-only pass takes separate target parameters: the analysis agents, tests/test_gpu_analysis_agents.py)


def test_agent_with_batchnorm_trains_acts_and_exports_like_the_oracle_agent():
    """iSDQN(batch_norm=True) end to end on the device replay (captured one-step graph included): per-step losses against the
    oracle agent fed from the oracle replay with the same seed, the greedy action from the running averages, and the model pickle
    layout {"params", "batch_stats"}."""
    from oracle.isdqn import iSDQN as Oracle
    from oracle.replay_buffer import ReplayBuffer as ORB, TransitionElement as OT
    from oracle.samplers import UniformSamplingDistribution as OU
    from slimdqn.networks.isdqn import iSDQN
    from slimdqn.sample_collection.replay_buffer import ReplayBuffer, TransitionElement
    from slimdqn.sample_collection.samplers import UniformSamplingDistribution

    K, A, B, feats = 2, 4, 8, [8, 16, 16, 32]
    agent = iSDQN(0, (84, 84, 4), A, K, feats, True, True, "cnn", 1e-3, 0.99, 1, 1, 6, adam_eps=1.5e-4, batch_size=B)
    model = agent.get_model()["params"]
    assert set(model) == {"params", "batch_stats"} and model["batch_stats"]["BatchNorm_0"]["var"].shape == (84, 84)
    assert model["params"]["BatchNorm_3"]["scale"].shape == (11 * 11 * 16,) and model["params"]["Conv_0"]["kernel"].shape == (8, 8, 4, 8)
    oracle = Oracle(0, (84, 84, 4), A, K, feats, True, True, "cnn", 1e-3, 0.99, 1, 1, 6, adam_eps=1.5e-4, params=model["params"])
    rb = ReplayBuffer(UniformSamplingDistribution(1), B, 64)
    orb = ORB(OU(1), B, 64)
    rng = np.random.default_rng(0)
    for t in range(30):
        obs = rng.integers(0, 256, (84, 84), dtype=np.uint8)
        a, r, term = int(rng.integers(0, A)), float(rng.choice([-1.0, 0.0, 1.0])), bool(t % 13 == 12)
        rb.add(TransitionElement(obs, a, r, term, term))
        orb.add(OT(obs, a, r, term, term))
    per_step = []
    for step in range(1, 5):
        agent.update_online_params(step, rb)
        oracle.update_online_params(step, orb)
        per_step.append((agent._engine.losses.cpu().numpy().astype(np.float64), oracle.cumulated_losses.copy()))
    # per-step losses: the first step starts from identical parameters (1e-3); afterwards the two trajectories take Adam's
    # normalised steps apart and 16-row batch statistics amplify that (a few per cent after four steps at lr = 1e-3)
    prev = np.zeros(K)
    for i, (got, cum) in enumerate(per_step):
        exp = cum - prev
        prev = cum
        tol = 1e-3 if i == 0 else 5e-2
        assert np.abs(got - exp).max() < tol * max(1.0, np.abs(exp).max()), (i, got, exp)
    acc = agent._engine.losses_accum.cpu().numpy()
    np.testing.assert_allclose(acc, np.sum([g for g, _ in per_step], axis=0), rtol=1e-5)  # the device accumulator (isdqn.py:62)
    stats = agent.get_model()["params"]["batch_stats"]
    for m, l in oracle.batch_stats.items():
        for n, t in l.items():
            assert np.abs(stats[m][n] - t.numpy()).max() < 5e-3 * max(1.0, float(t.abs().max())), (m, n)  # (four drifting steps, see above)
    # acting (isdqn.py:127-135, use_running_average=True) on the ORACLE's trained model, handed over as the reference's pytree
    # {"params", "batch_stats"}: no trajectory drift between the two sides, so the 1e-3 bar applies
    state = rng.integers(0, 256, (84, 84, 4), dtype=np.uint8)
    o_model = oracle.get_model()["params"]
    q_all = oracle.apply(oracle.params, torch.tensor(state)[None], use_running_average=True)[0].detach().numpy()
    q_a = agent.q_values(o_model, state)
    assert np.abs(q_a - q_all).max() < 1e-3 * max(1.0, np.abs(q_all).max())
    for head in range(K):
        top2 = np.sort(q_all[1 + head])[-2:]
        if top2[1] - top2[0] > 2e-3:  # (a near tie may legitimately resolve differently within the parity tolerance)
            assert agent.best_action(o_model, state, key=head) == oracle.best_action(oracle.params, state, head)
    # and on the agent's own parameters the greedy action is the argmax of its own q row
    q_own = agent.q_values(agent.params, state)
    for head in range(K):
        assert agent.best_action(agent.params, state, key=head) == int(np.argmax(q_own[1 + head]))


def test_tfdqn_with_batchnorm_matches_the_oracle():
    """TFDQN(batch_norm=True) (tfdqn.py:56-80): one shared-parameter head regressed on its own stop-gradient target, training-mode
    BatchNorm over concat(state, next_state); a learn step's loss, running averages and parameters against the oracle's."""
    from oracle.dqn import TFDQN as Oracle
    from slimdqn.networks.tfdqn import TFDQN

    A, B, feats = 4, 8, [8, 16, 16, 32]
    agent = TFDQN(0, (84, 84, 4), A, feats, True, True, "cnn", 1e-3, 0.99, 1, 1, 6, adam_eps=1.5e-4, batch_size=B)
    model = agent.get_model()["params"]
    assert set(model) == {"params", "batch_stats"}
    oracle = Oracle(0, (84, 84, 4), A, feats, True, True, "cnn", 1e-3, 0.99, 1, 1, 6, adam_eps=1.5e-4, params=model["params"])
    _frames, _ids, _a, _r, _t, ref = make_frame_batch(B, A, seed=21)
    o_loss = float(oracle.loss_on_batch(oracle.params, ref)[0])
    oracle.params, oracle.optimizer_state, o_loss2 = oracle.learn_on_batch(oracle.params, oracle.optimizer_state, ref)
    _, _, loss = agent.learn_on_batch(agent.params, agent.optimizer_state, ref)
    got = float(loss.cpu().numpy().reshape(-1)[0])
    assert abs(got - o_loss) < 1e-3 * max(1.0, abs(o_loss)) and abs(o_loss - o_loss2) < 1e-12
    after = agent.get_model()["params"]
    for m, l in oracle.batch_stats.items():
        for n, t in l.items():
            assert np.abs(after["batch_stats"][m][n] - t.numpy()).max() < 2e-5 * max(1.0, float(t.abs().max())), (m, n)
    exp = oracle.get_model()["params"]["params"]
    for m in exp:
        for n in exp[m]:
            assert np.abs(after["params"][m][n] - exp[m][n]).max() < 2.001e-3, (m, n)  # one Adam step of lr = 1e-3


@pytest.mark.parametrize("algo, arch", [("isdqn", "cnn"), ("tfdqn", "cnn"), ("isdqn", "impala")])
def test_entry_points_with_the_batch_norm_flag(tmp_path, algo, arch):
    """`-bn` through the reference's entry points (experiments/atari/isdqn.py, tfdqn.py; launch_job/atari/launch.sh BATCH_NORM=1):
    training runs, the vectorised acting path reads the running averages, and the saved model carries both Flax collections."""
    import json
    import pickle

    from experiments.atari import isdqn as e_isdqn, tfdqn as e_tfdqn

    run = {"isdqn": e_isdqn.run, "tfdqn": e_tfdqn.run}[algo]
    name = f"bn{algo}{arch}_Synthetic"
    argv = ["-en", name, "-s", "1", "-dw", "-f", "8", "8", "8", "16", "-rbc", "200", "-bs", "8", "-n", "1", "-horizon", "40", "-at", arch,
            "-ne", "2", "-ntspe", "48", "-utd", "4", "-nis", "16", "-ed", "100", "-ln", "-bn", "-tuf", "16", "-env", "synthetic", "-nenvs", "2"]
    if algo == "isdqn":
        argv += ["-nbi", "2"]
    run(argv, root=str(tmp_path))
    out = tmp_path / "atari" / "exp_output" / name
    assert json.load(open(out / "parameters.json"))[algo]["batch_norm"] is True  # (an agent parameter: parser_argument.py:27-36)
    model = pickle.load(open(out / algo / "models" / "1", "rb"))["params"]
    assert set(model) == {"params", "batch_stats"}
    stats = model["batch_stats"]
    top = stats["BatchNorm_0"]
    assert top["mean"].shape == (84, 84) and float(np.abs(top["mean"]).max()) > 0.0  # moved off Flax's initial (0, 1) by the learn steps
    if arch == "impala":
        assert stats["Stack_2"]["BatchNorm_1"]["var"].shape == (11, 11) and model["params"]["Stack_0"]["BatchNorm_0"]["scale"].shape == (42, 42)
    else:
        assert model["params"]["BatchNorm_3"]["scale"].shape == (11 * 11 * 8,) and stats["BatchNorm_4"]["var"].shape == (16,)
