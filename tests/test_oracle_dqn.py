"""Properties the reference's tests/test_dqn.py:41-79 and tests/test_tfdqn.py pin for the baselines (target, per-sample
loss, best action as formulas of the network output), restated on the oracle DQN / TFDQN, plus the relations between the
three algorithms that the HIP path exploits: TF-DQN is DQN with target_params == params, and both are the K = 1 iS-DQN
machinery with one head."""
import numpy as np
import torch

from oracle.dqn import DQN, TFDQN
from oracle.replay_buffer import ReplayElement

FEATS = [5, 6, 7, 9]


def _batch(B, A, seed=0):
    rng = np.random.default_rng(seed)
    return ReplayElement(state=rng.integers(0, 256, (B, 84, 84, 4), dtype=np.uint8), action=rng.integers(0, A, B),
                         reward=rng.normal(size=B), next_state=rng.integers(0, 256, (B, 84, 84, 4), dtype=np.uint8),
                         is_terminal=(rng.random(B) < 0.3).astype(np.int64))


def test_dqn_target_loss_and_best_action_formulas():
    A, B = 5, 4
    q = DQN(3, (84, 84, 4), A, FEATS, True, "cnn", 1e-3, 0.94, 1, 1, 1, dtype=torch.float64)
    s = _batch(B, A)
    qv, tg, td = q.loss_terms(q.params, q.target_params, s)
    nq = q.apply(q.target_params, s.next_state)
    assert nq.shape == (B, A)
    exp_t = torch.as_tensor(s.reward) + (1 - torch.as_tensor(s.is_terminal, dtype=torch.float64)) * 0.94 * nq.max(-1).values
    assert torch.allclose(tg, exp_t, rtol=1e-12, atol=0)                       # test_dqn.py:41-51 (two forwards: threaded conv sums)
    pred = q.apply(q.params, s.state)[torch.arange(B), torch.as_tensor(s.action)]
    assert torch.allclose(td, (pred - exp_t) ** 2, rtol=1e-10, atol=1e-14)     # test_dqn.py:53-63
    assert torch.equal(q.loss_on_batch(q.params, q.target_params, s), td.mean())
    for b in range(B):                                                        # test_dqn.py:65-79
        assert q.best_action(q.params, s.state[b]) == int(torch.argmax(q.apply(q.params, s.state[b : b + 1])[0]))


def test_tfdqn_is_dqn_with_the_online_parameters_as_target():
    A, B = 4, 6
    d = DQN(1, (84, 84, 4), A, FEATS, True, "cnn", 1e-3, 0.99, 3, 1, 1, dtype=torch.float64)
    t = TFDQN(1, (84, 84, 4), A, FEATS, True, False, "cnn", 1e-3, 0.99, 3, 1, 1, dtype=torch.float64)
    s = _batch(B, A, seed=2)
    assert torch.allclose(d.loss_on_batch(d.params, d.params, s), t.loss_on_batch(t.params, s)[0], rtol=1e-12)
    gd, _ = d.grads(d.params, d.params, s)
    gt, _ = t.grads(t.params, s)
    for m in gd:
        for n in gd[m]:
            assert torch.allclose(gd[m][n], gt[m][n], rtol=1e-9, atol=1e-12), (m, n)
    # DQN proper: a stale target changes the loss, and update_target_params refreshes it (dqn.py:49-57)
    p1, st, _ = d.learn_on_batch(d.params, d.target_params, d.optimizer_state, s)
    d.params = p1
    assert not torch.allclose(d.loss_on_batch(d.params, d.target_params, s), d.loss_on_batch(d.params, d.params, s))
    updated, logs = d.update_target_params(7)
    assert updated and "loss" in logs
    assert torch.allclose(d.loss_on_batch(d.params, d.target_params, s), d.loss_on_batch(d.params, d.params, s), rtol=1e-12)


def test_adam_step_moves_every_parameter_by_about_lr_on_the_first_step():
    A, B = 3, 4
    t = TFDQN(5, (84, 84, 4), A, FEATS, True, False, "cnn", 1e-3, 0.99, 1, 1, 1, adam_eps=1e-8, dtype=torch.float64)
    s = _batch(B, A, seed=4)
    g, _ = t.grads(t.params, s)
    p1, st, loss = t.learn_on_batch(t.params, t.optimizer_state, s)
    assert st["count"] == 1 and loss > 0
    for m in g:
        for n in g[m]:
            moved = (p1[m][n] - t.params[m][n]).abs()
            nz = g[m][n].abs() > 1e-4
            assert torch.allclose(moved[nz], torch.full_like(moved[nz], 1e-3), rtol=1e-3)


def test_huber_option_of_the_isdqn_oracle_matches_torch_and_reduces_to_l2_scale():
    """The Huber option (north-star wording; the reference trains on the squared error, isdqn.py:102): per-element values equal
    torch's huber_loss, the gradient w.r.t. q is clip(d, -delta, delta) / B, and huber_delta = 0 is the reference's loss."""
    import torch.nn.functional as F

    from oracle.isdqn import iSDQN

    K, A, B = 3, 4, 6
    l2 = iSDQN(2, (84, 84, 4), A, K, FEATS, True, False, "cnn", 1e-3, 0.99, 1, 1, 1, dtype=torch.float64)
    hub = iSDQN(2, (84, 84, 4), A, K, FEATS, True, False, "cnn", 1e-3, 0.99, 1, 1, 1, dtype=torch.float64, huber_delta=0.25)
    s = _batch(B, A, seed=6)
    q, t, td2 = l2.loss_terms(l2.params, s)
    qh, th, tdh = hub.loss_terms(hub.params, s)
    assert torch.allclose(q, qh) and torch.allclose(t, th)
    assert torch.allclose(td2, (q - t) ** 2)
    assert torch.allclose(tdh, F.huber_loss(qh, th, reduction="none", delta=0.25), rtol=1e-12, atol=1e-15)
    assert ((q - t).abs() > 0.25).any() and ((q - t).abs() < 0.25).any()  # both branches are exercised
