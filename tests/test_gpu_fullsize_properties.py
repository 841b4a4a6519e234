"""BASELINE.json's full sizes (configs[1]/[2]: B=256, K=9, A=9 and configs[4]: B=1024, K=32, A=4; cnn 32/64/64/512 +
LayerNorm) are too big for the CPU oracle to follow step by step in seconds, so the HIP path is checked there through
properties that do not depend on the size (every test runs at both shapes):

* run-to-run bitwise determinism of a learn step (every reduction in the path has a fixed order);
* the fused learn path (head chain kernel) against the independent forward-only path (head GEMM + post kernels):
  q_values / targets / losses recomputed on the host from `forward()` must agree within the 1e-3 parity bar
  (isdqn.py:92-109: target_k = r + (1 - terminal) * gamma^n * max_a Q_k(s'), head k+1 regressed on head k);
* `loss_on_batch` (no update) reports the same losses as the step that follows it;
* permuting the transitions of the batch permutes q_values / targets / priorities (to 1e-4: the summation order of
  an image depends on its workgroup index) and leaves the losses (a mean over the batch) unchanged;
* priorities = sqrt(mean_k td + 1e-10) (float64), the value the prioritized sampler writes back (isdqn.py:73-80).
"""
import numpy as np
import pytest
import torch

from tests.gpu_helpers import device_batch, make_frame_batch

pytestmark = pytest.mark.gpu

FEATS = (32, 64, 64, 512)
GAMMA_N = 0.99
SHAPES = [pytest.param((256, 9, 9), id="c2-B256-K9-A9"), pytest.param((1024, 32, 4), id="c5-B1024-K32-A4")]


def _engine(shape, seed=0):
    from slimdqn._engine import QNetEngine

    B, K, A = shape
    eng = QNetEngine((84, 84, 4), A, 1 + K, FEATS, "cnn", True, B, gamma_n=GAMMA_N, learning_rate=6.25e-5, adam_eps=1.5e-4)
    eng.init_params(seed)
    return eng


def _host_targets(shape, eng, frames, ids, action, reward, terminal):
    B, K, A = shape
    fr = torch.from_numpy(frames).cuda()
    both = torch.from_numpy(np.concatenate([ids[:, :4], ids[:, 4:]], 0).copy()).cuda()
    q = eng.forward(frames=fr, frame_stride=frames.shape[1], frame_ids=both, n_rows=2 * B).cpu().numpy().astype(np.float64)
    q = q.reshape(2 * B, 1 + K, A)
    q_on, q_next = q[:B], q[B:]
    qv = np.stack([q_on[np.arange(B), 1 + k, action] for k in range(K)], 1)
    tg = reward[:, None] + (1.0 - terminal[:, None]) * GAMMA_N * q_next[:, :K].max(-1)
    return qv, tg


@pytest.mark.parametrize("shape", SHAPES)
def test_learn_step_is_bitwise_deterministic_and_matches_forward_path(shape):
    B, K, A = shape
    frames, ids, action, reward, terminal, _ = make_frame_batch(B, A, seed=5)
    outs = []
    for _ in range(2):
        eng = _engine(shape, seed=1)
        batch = device_batch(eng, frames, ids, action, reward, terminal)
        qv, tg = _host_targets(shape, eng, frames, ids, action, reward.astype(np.float64), terminal.astype(np.float64))
        pre = eng.loss_on_batch(batch).cpu().numpy().copy()
        losses = eng.learn_on_batch(batch).cpu().numpy().copy()
        torch.cuda.synchronize()
        outs.append((losses, eng.q_values.cpu().numpy().copy(), eng.targets.cpu().numpy().copy(),
                     eng.priorities.cpu().numpy().copy(), eng.params.cpu().numpy().copy()))
        # forward-only path vs fused learn path (different kernels, same mathematics): the 1e-3 parity bar
        assert np.abs(outs[-1][1] - qv).max() < 1e-3
        assert np.abs(outs[-1][2] - tg).max() < 1e-3
        td = (qv - tg) ** 2
        assert np.abs(losses - td.mean(0)).max() < 1e-3 * max(1.0, td.mean(0).max())
        # loss_on_batch (no update) saw the same parameters
        assert np.abs(pre - losses).max() < 1e-3 * max(1.0, losses.max())
        # priorities (float64): sqrt(mean_k td + 1e-10) of the step's own q / targets
        own_td = (outs[-1][1].astype(np.float64) - outs[-1][2].astype(np.float64)) ** 2
        np.testing.assert_allclose(outs[-1][3], np.sqrt(own_td.mean(1) + 1e-10), rtol=2e-6, atol=1e-9)
    for a, b in zip(outs[0], outs[1]):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("shape", SHAPES)
def test_several_steps_and_every_intermediate_are_run_to_run_identical(shape):
    """Several back-to-back steps from identical state, three times: parameters, Adam moments and the backward
    intermediates of the last step must be bit-identical.  (This is the test that exposed compiler-renamed dependent
    MFMAs and an LDS hazard in the pipelined K loops: about 1 % of workgroups returned a tile that lacked one MFMA
    pass, invisible to the small-batch oracle comparisons.)"""
    B, K, A = shape
    frames, ids, action, reward, terminal, _ = make_frame_batch(B, A, seed=13)
    names = ["act/Conv_0", "act/Conv_1", "act/Conv_2", "act/Dense_0", "dz/Dense_0", "dz/Conv_2", "dz/Conv_1", "dz/Conv_0",
             "gw/Conv_0", "gw/Conv_1", "gw/Conv_2"]
    ref = None
    for _ in range(3):
        eng = _engine(shape, seed=3)
        batch = device_batch(eng, frames, ids, action, reward, terminal)
        for _ in range(4):
            eng.learn_on_batch(batch)
        torch.cuda.synchronize()
        cur = {n: eng.region(n).clone() for n in names}
        cur["params"], cur["adam_m"], cur["adam_v"] = eng.params.clone(), eng.adam_m.clone(), eng.adam_v.clone()
        if ref is None:
            ref = cur
        else:
            for n, t in cur.items():
                assert torch.equal(t, ref[n]), f"{n}: {(t != ref[n]).sum().item()} elements differ between identical runs"


@pytest.mark.parametrize("shape", SHAPES)
def test_batch_permutation_permutes_rows_and_keeps_losses(shape):
    B, K, A = shape
    frames, ids, action, reward, terminal, _ = make_frame_batch(B, A, seed=9)
    perm = np.random.default_rng(0).permutation(B)
    res = []
    for order in (np.arange(B), perm):
        eng = _engine(shape, seed=2)
        batch = device_batch(eng, frames, ids[order], action[order], reward[order], terminal[order])
        losses = eng.learn_on_batch(batch).cpu().numpy().copy()
        res.append((losses, eng.q_values.cpu().numpy().copy(), eng.targets.cpu().numpy().copy(), eng.priorities.cpu().numpy().copy()))
    (l0, q0, t0, p0), (l1, q1, t1, p1) = res
    # per-transition outputs follow their transition.  Not bit for bit: a workgroup starts its K walk at a slice that
    # depends on its index (L2 channel spreading), so the fp32 summation order of an image depends on its position
    np.testing.assert_allclose(q0[perm], q1, rtol=0, atol=1e-4)
    np.testing.assert_allclose(t0[perm], t1, rtol=0, atol=1e-4)
    np.testing.assert_allclose(p0[perm], p1, rtol=1e-4, atol=1e-6)
    # the loss is a mean over the batch: same up to the summation order
    np.testing.assert_allclose(l0, l1, rtol=1e-5)


FC_PLANS = [pytest.param(((64,), 8, 1, 4, 32), id="fc-2-layer-plan"), pytest.param(((100, 100), 8, 1, 4, 32), id="fc-3-layer-lunar-lander"),
            pytest.param(((300, 600), 8, 3, 5, 1000), id="fc-wide-B1000")]


@pytest.mark.parametrize("plan", FC_PLANS)
def test_all_dense_plans_are_run_to_run_identical_and_consistent_with_the_forward_path(plan):
    """The same properties for all-dense networks (BASELINE configs[0] and the plans around it): the backward's tail takes other
    branches there than in any cnn network (net_kernels.hip: `tail_swap` without a fused data gradient, the unaligned first-layer
    operand on the side stream), so determinism and the learn path against the forward-only path are held on them too."""
    from slimdqn._engine import QNetEngine

    feats, d, K, A, B = plan
    rng = np.random.default_rng(3)
    st, nx = rng.normal(size=(B, d)).astype(np.float32), rng.normal(size=(B, d)).astype(np.float32)
    action = rng.integers(0, A, B).astype(np.int32)
    reward = rng.normal(size=B).astype(np.float32)
    terminal = (rng.random(B) < 0.2).astype(np.uint8)
    dev = lambda a: torch.from_numpy(a).cuda()
    outs = []
    for rep in range(3):
        eng = QNetEngine((d,), A, 1 + K, feats, "fc", True, B, gamma_n=GAMMA_N, learning_rate=1e-3, adam_eps=1e-8)
        eng.init_params(7)
        batch = eng.make_batch(state=dev(st), next_state=dev(nx), action=dev(action), reward=dev(reward), terminal=dev(terminal))
        if rep == 0:  # q / targets / losses of the learn path against a host evaluation of forward() on the same parameters
            q = eng.forward(obs=dev(np.concatenate([st, nx])), n_rows=2 * B).cpu().numpy().astype(np.float64).reshape(2 * B, 1 + K, A)
            qv = q[np.arange(B), 1:, :][np.arange(B), :, action]
            tg = reward[:, None] + (1.0 - terminal[:, None]) * GAMMA_N * q[B:, :K].max(-1)
            pre = eng.loss_on_batch(batch).cpu().numpy()
            np.testing.assert_allclose(eng.q_values.cpu().numpy(), qv, atol=1e-3)
            np.testing.assert_allclose(eng.targets.cpu().numpy(), tg, atol=1e-3)
            np.testing.assert_allclose(pre, ((qv - tg) ** 2).mean(0), rtol=1e-3, atol=1e-3)
        losses = [eng.learn_on_batch(batch).cpu().numpy().copy() for _ in range(4)]
        if rep == 0:
            np.testing.assert_allclose(losses[0], pre, rtol=1e-5, atol=1e-6)  # loss_on_batch == the step that follows it
        outs.append((eng.params.cpu().numpy().copy(), eng.adam_m.cpu().numpy().copy(), eng.adam_v.cpu().numpy().copy(), np.stack(losses),
                     eng.priorities.cpu().numpy().copy()))
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            np.testing.assert_array_equal(a, b)
