"""Oracle network/agent self-checks.  The reference's tests/test_isdqn.py pins formulas only
(both sides call the same Flax network); those properties are restated here on the oracle
(:51-63 target, :65-82 loss, :84-97 best_action, :99-116 shift), plus a cross-check of the
torch forward against an independent numpy im2col statement and of autograd + Adam against
finite differences / a hand-rolled update."""
import numpy as np
import torch

from oracle import network as net
from oracle.isdqn import iSDQN
from oracle.replay_buffer import ReplayElement


def _agent(seed=3, K=3, A=5, feats=(7, 9, 11, 13), dtype=torch.float64, arch="cnn", obs=(84, 84, 4)):
    ag = iSDQN(seed, obs, A, K, list(feats), True, False, arch, 1e-3, 0.94, 1, 1, 1, adam_eps=1e-8, dtype=dtype)
    rng = np.random.default_rng(seed)
    for m in ag.params:  # move LN scale/bias and biases off their trivial init
        for n in ag.params[m]:
            ag.params[m][n] = ag.params[m][n] + torch.tensor(rng.normal(0, 0.05, ag.params[m][n].shape), dtype=dtype)
    return ag


def _batch(B, A, seed=0, obs=(84, 84, 4)):
    rng = np.random.default_rng(seed)
    return ReplayElement(
        state=rng.integers(0, 256, (B,) + obs, dtype=np.uint8),
        action=rng.integers(0, A, B),
        reward=rng.normal(size=B),
        next_state=rng.integers(0, 256, (B,) + obs, dtype=np.uint8),
        is_terminal=rng.integers(0, 2, B),
    )


def test_same_padding_geometry():
    assert net.same_padding(84, 8, 4) == (21, 2, 2)
    assert net.same_padding(21, 4, 2) == (11, 1, 2)
    assert net.same_padding(11, 3, 1) == (11, 1, 1)


def test_torch_forward_matches_numpy_im2col():
    ag = _agent()
    x = np.random.default_rng(1).integers(0, 256, (3, 84, 84, 4), dtype=np.uint8)
    yt = net.forward(ag.params, torch.tensor(x), ag.features, "cnn", True).numpy()
    yn = net.forward_numpy(net.to_numpy(ag.params), x, ag.features, "cnn", True)
    np.testing.assert_allclose(yt, yn, rtol=0, atol=1e-12)


def test_param_count_headline_config():
    p = net.init_params(0, (84, 84, 4), [32, 64, 64, 512], "cnn", 90, True)
    assert sum(a.size for l in p.values() for a in l.values()) == 4_090_938
    assert p["Dense_0"]["kernel"].shape == (7744, 512)
    assert [n for n, _ in net.layer_names([32, 64, 64, 512], "cnn", True)] == [
        "Conv_0", "LayerNorm_0", "Conv_1", "LayerNorm_1", "Conv_2", "LayerNorm_2",
        "Dense_0", "LayerNorm_3", "Dense_1",
    ]
    assert [n for n, _ in net.layer_names([100, 100], "fc", True)] == [
        "Dense_0", "LayerNorm_0", "Dense_1", "LayerNorm_1", "Dense_2"]


def test_compute_target_and_loss_formula():
    ag = _agent()
    s = _batch(6, ag.n_actions)
    q, t, td = ag.loss_terms(ag.params, s)
    all_q = ag.apply(ag.params, torch.cat((torch.tensor(s.state), torch.tensor(s.next_state))))
    B = 6
    for b in range(B):
        for k in range(ag.n_bellman_iterations):
            exp_t = s.reward[b] + (1 - s.is_terminal[b]) * ag.gamma * all_q[B + b, k].max().item()
            assert abs(t[b, k].item() - exp_t) < 1e-12
            assert q[b, k].item() == all_q[b, 1 + k, s.action[b]].item()
    loss, (per_head, _) = ag.loss_on_batch(ag.params, s)
    assert abs(loss.item() - ((q - t) ** 2).mean(0).sum().item()) < 1e-12
    assert per_head.shape == (ag.n_bellman_iterations,)


def test_best_action_and_shift():
    ag = _agent()
    name = f"Dense_{ag.last_idx_mlp}"
    n_out = (1 + ag.n_bellman_iterations) * ag.n_actions
    ag.params[name]["bias"] = torch.arange(n_out, dtype=ag.dtype) / 100
    x = np.random.default_rng(2).integers(0, 256, (84, 84, 4), dtype=np.uint8)
    q = ag.apply(ag.params, torch.tensor(x)[None])[0]
    for idx in range(ag.n_bellman_iterations):
        assert ag.best_action(ag.params, x, idx) == int(q[1 + idx].argmax())
    shifted = ag.shift_params(ag.params)
    q2 = ag.apply(shifted, torch.tensor(x)[None])[0]
    assert torch.linalg.norm(q2[:-1] - q[1:]).item() == 0.0
    assert torch.equal(q2[-1], q[-1])


def test_grad_against_finite_differences_and_adam():
    ag = _agent(K=2, A=3, feats=(5, 6, 7, 8))
    s = _batch(4, 3)
    grads, _ = ag.grads(ag.params, s)
    rng = np.random.default_rng(0)
    for mod in ("Conv_1", "LayerNorm_2", "Dense_0", "Dense_1"):
        for leaf in ag.params[mod]:
            t = ag.params[mod][leaf]
            idx = tuple(int(rng.integers(0, d)) for d in t.shape)
            eps = 1e-6
            old = t[idx].item()
            t[idx] = old + eps
            lp = ag.loss_on_batch(ag.params, s)[0].item()
            t[idx] = old - eps
            lm = ag.loss_on_batch(ag.params, s)[0].item()
            t[idx] = old
            fd = (lp - lm) / (2 * eps)
            assert abs(fd - grads[mod][leaf][idx].item()) < 1e-5 * max(1.0, abs(fd))
    # two Adam steps against a hand-rolled scalar update on one leaf
    p0 = ag.params["Dense_1"]["bias"].clone()
    p1, st1, _ = ag.learn_on_batch(ag.params, ag.optimizer_state, s)
    g1 = grads["Dense_1"]["bias"]
    m = 0.1 * g1
    v = 0.001 * g1 * g1
    exp = p0 - 1e-3 * (m / 0.1) / (torch.sqrt(v / 0.001) + 1e-8)
    torch.testing.assert_close(p1["Dense_1"]["bias"], exp, rtol=1e-12, atol=1e-14)
    assert st1["count"] == 1


def test_fc_architecture_runs():
    ag = _agent(K=1, A=4, feats=(20, 20), arch="fc", obs=(8,))
    rng = np.random.default_rng(0)
    s = ReplayElement(state=rng.normal(size=(5, 8)), action=rng.integers(0, 4, 5), reward=rng.normal(size=5),
                      next_state=rng.normal(size=(5, 8)), is_terminal=rng.integers(0, 2, 5))
    p, st, losses = ag.learn_on_batch(ag.params, ag.optimizer_state, s)
    assert losses.shape == (1,) and np.isfinite(losses).all()


def test_impala_torso_restatement_against_plain_numpy_loops():
    """Stack (architectures/dqn.py:7-36): conv3x3 SAME, max_pool 3x3/2 SAME (-inf padding, pad_lo = total // 2), two residual blocks
    [LayerNorm] -> relu -> conv -> relu -> conv -> + input; torso :75-88.  torch statement vs loops in float64 on a small input."""
    import numpy as np
    import torch

    from oracle import network as net

    feats, obs, final = [3, 4, 2, 5], (13, 13, 2), 6
    params = net.init_params(1, obs, feats, "impala", final, True)
    rng = np.random.default_rng(0)
    for m in params:
        for n in params[m]:
            if n != "kernel":
                params[m][n] = (params[m][n] + rng.normal(0, 0.2, params[m][n].shape)).astype(np.float32)
    x = rng.integers(0, 256, (2,) + obs).astype(np.uint8)
    got = net.forward(net.to_torch(params, torch.float64), torch.tensor(x), feats, "impala", True).numpy()

    P = {m: {n: np.asarray(a, np.float64) for n, a in l.items()} for m, l in params.items()}

    def conv(t, p):
        N, H, W, C = t.shape
        tp = np.zeros((N, H + 2, W + 2, C))
        tp[:, 1:-1, 1:-1] = t
        out = np.zeros((N, H, W, p["kernel"].shape[3]))
        for y in range(H):
            for xx in range(W):
                out[:, y, xx] = np.tensordot(tp[:, y : y + 3, xx : xx + 3], p["kernel"], axes=([1, 2, 3], [0, 1, 2])) + p["bias"]
        return out

    def pool(t):
        N, H, W, C = t.shape
        Ho, Wo = -(-H // 2), -(-W // 2)
        lo_h, lo_w = max((Ho - 1) * 2 + 3 - H, 0) // 2, max((Wo - 1) * 2 + 3 - W, 0) // 2
        out = np.full((N, Ho, Wo, C), -np.inf)
        for oy in range(Ho):
            for ox in range(Wo):
                for ky in range(3):
                    for kx in range(3):
                        iy, ix = 2 * oy - lo_h + ky, 2 * ox - lo_w + kx
                        if 0 <= iy < H and 0 <= ix < W:
                            out[:, oy, ox] = np.maximum(out[:, oy, ox], t[:, iy, ix])
        return out

    def ln(z, q):
        mean = z.mean(-1, keepdims=True)
        var = np.maximum((z * z).mean(-1, keepdims=True) - mean * mean, 0.0)
        return (z - mean) / np.sqrt(var + 1e-6) * q["scale"] + q["bias"]

    t = x.astype(np.float64) / 255.0
    for s in range(3):
        t = pool(conv(t, P[f"Stack_{s}/Conv_0"]))
        for b in range(2):
            r = t
            t = np.maximum(ln(t, P[f"Stack_{s}/LayerNorm_{b}"]), 0.0)
            t = np.maximum(conv(t, P[f"Stack_{s}/Conv_{1 + 2 * b}"]), 0.0)
            t = conv(t, P[f"Stack_{s}/Conv_{2 + 2 * b}"]) + r
    t = np.maximum(ln(t, P["LayerNorm_0"]), 0.0).reshape(2, -1)
    t = np.maximum(ln(t @ P["Dense_0"]["kernel"] + P["Dense_0"]["bias"], P["LayerNorm_1"]), 0.0)
    want = t @ P["Dense_1"]["kernel"] + P["Dense_1"]["bias"]
    np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-10)


def test_per_sample_statement_of_the_loss_agrees_with_the_batched_one():
    """tests/flops_computation/isdqn.py:93-100 states the loss a second time, per sample under vmap: q = heads 1..K of the state at the
    taken action, targets from heads 0..K-1 of the next state (stop-gradient), sum_k of the squared errors, mean over the batch --
    the same number as isdqn.py:92-103's batched td.mean(0).sum().  Both statements, evaluated on the oracle's network."""
    ag = _agent(K=3, A=4)
    batch = _batch(5, 4, seed=2)
    batched, _ = ag.loss_on_batch(ag.params, batch)
    per_sample = []
    for b in range(5):
        q_s = ag.apply(ag.params, torch.as_tensor(batch.state[b])[None])[0]          # (1+K, A)
        q_n = ag.apply(ag.params, torch.as_tensor(batch.next_state[b])[None])[0]
        q = q_s[1:, int(batch.action[b])]
        target = float(batch.reward[b]) + (1 - int(batch.is_terminal[b])) * ag.gamma**ag.update_horizon * q_n[:-1].max(dim=-1).values
        per_sample.append(((q - target.detach()) ** 2).sum())
    assert abs(float(torch.stack(per_sample).mean()) - float(batched)) < 1e-10 * max(1.0, abs(float(batched)))
