"""Oracle self-checks for the BatchNorm variants (slimdqn/networks/architectures/dqn.py:52-53, 59-60, 66-67, 73-74, 100-101;
isdqn.py:87-88, 95, 130).  The reference holds no numbers for them (parity unpinned, like the rest of the network numerics): the
torch restatement is cross-checked against plain numpy loops written from flax.linen.BatchNorm's documented behaviour, and the
properties that define the training-mode step -- the gradient reaches the next-state rows through the batch statistics, the running
averages move with momentum 0.99 and only acting reads them -- are asserted on the oracle agent."""
import numpy as np
import torch

from oracle import network as net
from oracle.isdqn import iSDQN
from oracle.replay_buffer import ReplayElement


def _bn_numpy(x, scale, bias, spatial):
    """flax.linen.BatchNorm, training mode, by explicit loops.  spatial: x (N, H, W, C), one statistic per (h, w) over n and c."""
    x = np.asarray(x, np.float64)
    y = np.empty_like(x)
    if spatial:
        N, H, W, C = x.shape
        mean, var = np.empty((H, W)), np.empty((H, W))
        for h in range(H):
            for w in range(W):
                v = x[:, h, w, :].reshape(-1)
                mean[h, w] = v.sum() / v.size
                var[h, w] = max((v * v).sum() / v.size - mean[h, w] ** 2, 0.0)
                y[:, h, w, :] = (x[:, h, w, :] - mean[h, w]) * (1.0 / np.sqrt(var[h, w] + 1e-5) * scale[h, w]) + bias[h, w]
    else:
        N, F = x.shape
        mean, var = np.empty(F), np.empty(F)
        for f in range(F):
            v = x[:, f]
            mean[f] = v.sum() / N
            var[f] = max((v * v).sum() / N - mean[f] ** 2, 0.0)
            y[:, f] = (v - mean[f]) * (1.0 / np.sqrt(var[f] + 1e-5) * scale[f]) + bias[f]
    return y, mean, var


def test_batch_norm_restatement_against_plain_numpy_loops():
    rng = np.random.default_rng(0)
    for spatial, shape, pshape in ((True, (5, 4, 3, 6), (4, 3)), (False, (7, 10), (10,))):
        x = rng.normal(1.0, 2.0, shape)
        p = {"scale": torch.tensor(rng.normal(1, 0.2, pshape)), "bias": torch.tensor(rng.normal(0, 0.2, pshape))}
        stats = {"bn": {"mean": torch.tensor(rng.normal(0, 1, pshape)), "var": torch.tensor(rng.uniform(0.5, 2, pshape))}}
        new = {}
        y = net._batch_norm(torch.tensor(x), p, stats, False, spatial, new, "bn").numpy()
        y_np, mean, var = _bn_numpy(x, p["scale"].numpy(), p["bias"].numpy(), spatial)
        np.testing.assert_allclose(y, y_np, rtol=1e-10, atol=1e-10)
        np.testing.assert_allclose(new["bn"]["mean"].numpy(), 0.99 * stats["bn"]["mean"].numpy() + 0.01 * mean, rtol=1e-12)
        np.testing.assert_allclose(new["bn"]["var"].numpy(), 0.99 * stats["bn"]["var"].numpy() + 0.01 * var, rtol=1e-12)
        # use_running_average=True: the stored statistics, nothing moves
        y_run = net._batch_norm(torch.tensor(x), p, stats, True, spatial, None, "bn").numpy()
        m, v = stats["bn"]["mean"].numpy(), stats["bn"]["var"].numpy()
        if spatial:
            m, v, sc, bi = (a[None, :, :, None] for a in (m, v, p["scale"].numpy(), p["bias"].numpy()))
        else:
            sc, bi = p["scale"].numpy(), p["bias"].numpy()
        np.testing.assert_allclose(y_run, (x - m) / np.sqrt(v + 1e-5) * sc + bi, rtol=1e-10, atol=1e-10)


def test_module_names_and_shapes_follow_flax_call_order():
    p = net.init_params(0, (20, 20, 4), [8, 8, 8, 16], "cnn", 6, True, batch_norm=True)
    assert p["BatchNorm_0"]["scale"].shape == (20, 20)          # x / 255 (axis=(1, 2): the features are the pixel positions)
    assert p["BatchNorm_1"]["scale"].shape == (5, 5)            # behind Conv_0 (20 -> 5 at stride 4)
    assert p["BatchNorm_2"]["scale"].shape == (3, 3)            # behind Conv_1
    assert p["BatchNorm_3"]["scale"].shape == (3 * 3 * 8,)      # behind the flatten: per feature
    assert p["BatchNorm_4"]["scale"].shape == (16,)             # behind Dense_0
    assert "BatchNorm_5" not in p                               # nothing behind the last Dense (dqn.py:103)
    stats = net.init_batch_stats(p)
    assert set(stats) == {f"BatchNorm_{i}" for i in range(5)}
    assert float(stats["BatchNorm_3"]["var"].min()) == 1.0 and float(np.abs(stats["BatchNorm_3"]["mean"]).max()) == 0.0
    q = net.init_params(0, (6,), [10, 12], "fc", 4, False, batch_norm=True)
    assert [k for k in q if k.startswith("BatchNorm")] == ["BatchNorm_0", "BatchNorm_1"] and q["BatchNorm_1"]["scale"].shape == (12,)
    # the weight draws do not depend on the flag
    np.testing.assert_array_equal(p["Conv_1"]["kernel"], net.init_params(0, (20, 20, 4), [8, 8, 8, 16], "cnn", 6, True)["Conv_1"]["kernel"])


def _agent_and_batch(arch, obs, feats, B=4, K=2, A=3, ln=True):
    ag = iSDQN(5, obs, A, K, feats, ln, True, arch, 1e-3, 0.9, 1, 1, 1, adam_eps=1e-8, dtype=torch.float64)
    rng = np.random.default_rng(1)
    for m in ag.params:
        for n in ag.params[m]:
            ag.params[m][n] = ag.params[m][n] + torch.tensor(rng.normal(0, 0.05, ag.params[m][n].shape))
    if arch == "fc":
        st, nx = rng.normal(size=(B,) + obs), rng.normal(size=(B,) + obs)
    else:
        st, nx = rng.integers(0, 256, (B,) + obs, dtype=np.uint8), rng.integers(0, 256, (B,) + obs, dtype=np.uint8)
    batch = ReplayElement(state=st, action=rng.integers(0, A, B), reward=rng.normal(size=B), next_state=nx, is_terminal=rng.integers(0, 2, B))
    return ag, batch


def test_gradient_reaches_the_next_states_through_the_batch_statistics():
    """isdqn.py:95-99: the targets are stop-gradient, but q of the online rows depends on the next-state rows through mean / var of
    concat(state, next_state) -- d loss / d next_state is not zero with BatchNorm and exactly zero without."""
    for bn in (True, False):
        ag = iSDQN(5, (6,), 3, 2, [10, 12], True, bn, "fc", 1e-3, 0.9, 1, 1, 1, dtype=torch.float64)
        rng = np.random.default_rng(2)
        st = torch.tensor(rng.normal(size=(4, 6)))
        nx = torch.tensor(rng.normal(size=(4, 6)), requires_grad=True)
        all_q = ag.apply(ag.params, torch.cat((st, nx)))
        q = all_q[:4, 1:, 0]
        targets = (0.5 + 0.9 * all_q[4:, :-1].max(dim=-1).values).detach()
        ((q - targets) ** 2).mean(0).sum().backward()
        g = float(nx.grad.abs().max())
        assert (g > 1e-6) if bn else (g == 0.0), (bn, g)


def test_learn_moves_the_running_averages_and_only_acting_reads_them():
    for arch, obs, feats in (("cnn", (20, 20, 4), [8, 8, 8, 16]), ("fc", (6,), [10, 12])):
        ag, batch = _agent_and_batch(arch, obs, feats)
        before = {m: {n: t.clone() for n, t in l.items()} for m, l in ag.batch_stats.items()}
        loss0, _ = ag.loss_on_batch(ag.params, batch)
        # loss_on_batch does not depend on the running averages ...
        ag.batch_stats = {m: {n: t + 0.3 for n, t in l.items()} for m, l in ag.batch_stats.items()}
        loss1, _ = ag.loss_on_batch(ag.params, batch)
        assert float(abs(loss0 - loss1)) == 0.0
        # ... acting does
        s = batch.state[0]
        q_shifted = ag.apply(ag.params, torch.as_tensor(np.asarray(s))[None], use_running_average=True)
        ag.batch_stats = before
        q_run = ag.apply(ag.params, torch.as_tensor(np.asarray(s))[None], use_running_average=True)
        assert float((q_run - q_shifted).abs().max()) > 1e-6
        # learn_on_batch keeps ra = 0.99 ra + 0.01 batch of its own forward
        ag.params, ag.optimizer_state, _ = ag.learn_on_batch(ag.params, ag.optimizer_state, batch)
        name = "BatchNorm_0"
        x = torch.cat((torch.as_tensor(np.asarray(batch.state)), torch.as_tensor(np.asarray(batch.next_state)))).to(torch.float64)
        if arch == "cnn":
            x = x / 255.0
            mean = x.mean(dim=(0, 3))
            np.testing.assert_allclose(ag.batch_stats[name]["mean"].numpy(), 0.01 * mean.numpy(), rtol=1e-10, atol=1e-12)
        else:
            assert float((ag.batch_stats[name]["var"] - 1.0).abs().max()) > 0.0
        assert set(ag.get_model()["params"]) == {"params", "batch_stats"}


def test_grad_against_finite_differences_with_batch_norm():
    """autograd through the batch statistics against central differences of the loss with the targets held at their
    stop-gradient values (isdqn.py:99)."""
    ag, batch = _agent_and_batch("fc", (6,), [10, 12], B=5)
    grads, _ = ag.grads(ag.params, batch)
    targets0 = ag.loss_terms(ag.params, batch)[1]

    def frozen_loss():
        q, _t, _td = ag.loss_terms(ag.params, batch)
        return float(((q - targets0) ** 2).mean(0).sum())

    rng = np.random.default_rng(3)
    for mod, leaf in (("Dense_0", "kernel"), ("BatchNorm_0", "scale"), ("BatchNorm_1", "bias"), ("LayerNorm_0", "scale"), ("Dense_2", "bias")):
        t = ag.params[mod][leaf]
        idx = tuple(int(rng.integers(0, d)) for d in t.shape)
        eps = 1e-6
        old = float(t[idx])
        t[idx] = old + eps
        lp = frozen_loss()
        t[idx] = old - eps
        lm = frozen_loss()
        t[idx] = old
        fd = (lp - lm) / (2 * eps)
        assert abs(fd - float(grads[mod][leaf][idx])) <= 1e-6 * max(1.0, abs(fd)), (mod, leaf, fd, float(grads[mod][leaf][idx]))
