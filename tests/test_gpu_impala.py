"""GPU parity of the Impala torso (reference slimdqn/networks/architectures/dqn.py:7-36 `Stack`, :75-88) against
the oracle restatement (oracle/network.py: _impala_stack): forward, Bellman targets, per-head losses within 1e-3, the first-step
gradient of EVERY leaf (15 convolutions, 6 block LayerNorms, the LayerNorm behind the torso, the dense tail), Adam steps, acting and
the parameter layout round trip with Flax's nested module names."""
import numpy as np
import pytest
import torch

from tests.gpu_helpers import device_batch, make_frame_batch, make_pair

pytestmark = pytest.mark.gpu

SHAPES = [
    # (feats, K, A, B, layer_norm, obs)
    pytest.param(((8, 16, 16, 24), 2, 5, 4, True, (84, 84, 4)), id="tiny-ln-B4"),
    pytest.param(((8, 16, 16, 24), 2, 5, 3, False, (84, 84, 4)), id="tiny-noln-B3"),
    pytest.param(((12, 20, 9, 16), 3, 4, 2, True, (44, 44, 3)), id="odd-widths-44x44x3"),
    pytest.param(((32, 64, 64, 512), 2, 6, 2, True, (84, 84, 4)), id="reference-widths-32-64-64-512-B2"),  # launch_time.sh:1
]


@pytest.mark.parametrize("shape", SHAPES)
def test_impala_forward_loss_gradients_and_adam_match_the_oracle(shape):
    feats, K, A, B, ln, obs = shape
    oracle, eng, params = make_pair(feats, K, A, B, arch="impala", obs=obs, layer_norm=ln, seed=3, lr=1e-3)
    frames, ids, action, reward, terminal, ref = make_frame_batch(B, A, seed=7, h=obs[0], w=obs[1], stack=obs[2])
    batch = device_batch(eng, frames, ids, action, reward, terminal)
    # parameter layout: Flax names, shapes and values survive the round trip
    got = eng.export_flax()
    assert set(got) == set(params)
    for mod in params:
        for leaf in params[mod]:
            np.testing.assert_array_equal(got[mod][leaf], params[mod][leaf])
    # forward of one observation and of the whole batch (loss path): q, targets, losses
    o_q, o_t, o_td = oracle.loss_terms(oracle.params, ref)
    losses = eng.loss_on_batch(batch).cpu().numpy()
    assert np.abs(eng.q_values.cpu().numpy() - o_q.detach().numpy()).max() < 1e-3
    assert np.abs(eng.targets.cpu().numpy() - o_t.detach().numpy()).max() < 1e-3
    assert np.abs(losses - o_td.mean(0).detach().numpy()).max() < 1e-3 * max(1.0, float(o_td.mean(0).max()))
    # gradient of every leaf (gradient-only pass) against a float64 reference that takes every DECISION of the online half -- ReLU
    # masks, max-pool winners -- from the HIP path's own tensors (as tests/gpu_helpers.py: masked_reference_grads does for the cnn
    # torso: a batch holds ~1e5 decisions, a few within the forward's 1e-5 of a tie; one differing decision changes that image's
    # gradient in a whole receptive field).  With the decisions pinned the comparison is arithmetic only: 2e-4 per leaf.
    g = torch.zeros_like(eng.params)
    eng.grad_on_batch(batch, g)
    torch.cuda.synchronize()
    m_grads = _masked_impala_grads(eng, params, feats, K, A, B, ln, ref, eng.targets.cpu().numpy())
    hip_g = eng.internal_to_flax_grads(g)
    for mod in m_grads:
        for leaf in m_grads[mod]:
            a, b = np.asarray(hip_g[mod][leaf], np.float64), m_grads[mod][leaf]
            assert np.linalg.norm(b) > 0, (mod, leaf)
            e = np.linalg.norm(a - b) / np.linalg.norm(b)
            assert e <= 2e-4, (mod, leaf, e)
    # and the independent oracle (its own decisions): the whole gradient, Euclidean -- loose (a few differing decisions in a batch of
    # 2-4 images move small leaves by tens of per cent), but a wrong formula or a missing term is an O(1) error of the whole vector
    o_grads, _ = oracle.grads(oracle.params, ref)
    num = sum(float(np.sum((np.asarray(hip_g[m][n], np.float64) - o_grads[m][n].numpy()) ** 2)) for m in o_grads for n in o_grads[m])
    den = sum(float(np.sum(o_grads[m][n].numpy().astype(np.float64) ** 2)) for m in o_grads for n in o_grads[m])
    assert num <= 0.15**2 * den, (num / den) ** 0.5
    # the update: Adam applied by the learn step == optax.adam (isdqn.py:46, 85-86) applied to the path's own gradient, element by
    # element in the internal layout (first step: m_hat = g, v_hat = g^2, so p -= lr * g / (|g| + eps)); losses of that step == oracle's.
    # (Comparing parameter TRAJECTORIES with the oracle is meaningless at B = 2-4: a few differing decisions flip the sign of small
    # gradients, and Adam moves those entries by a full lr either way.)
    p0 = eng.params.clone()
    _, _, o_losses = oracle.learn_on_batch(oracle.params, oracle.optimizer_state, ref)
    g2 = torch.zeros_like(eng.params)  # the gradient of THIS pass (the learn step takes the head chain, the gradient-only pass does not:
    h_losses = eng.learn_on_batch(batch, grad_out=g2).cpu().numpy()  # a ReLU decision of the hidden layer may differ between the two)
    assert np.abs(h_losses - o_losses).max() < 1e-3 * max(1.0, np.abs(o_losses).max())
    want = p0 - 1e-3 * g2 / (g2.abs() + 1.5e-4)
    assert float((eng.params - want).abs().max()) < 2e-6
    np.testing.assert_allclose(eng.adam_m.cpu().numpy(), (0.1 * g2).cpu().numpy(), rtol=1e-5, atol=1e-10)
    assert float((g2 - g).norm() / g.norm()) < 0.05
    second = eng.learn_on_batch(batch).cpu().numpy()
    assert np.isfinite(second).all() and int(eng.adam_count.item()) == 2
    p = {m: {n: torch.tensor(v) for n, v in l.items()} for m, l in eng.export_flax().items()}  # acting is checked on the updated net
    # acting: argmax of every online head for one observation
    fr = torch.from_numpy(frames).cuda()
    one = torch.from_numpy(ids[:1, : obs[2]].copy()).cuda()
    for idx in range(K):
        a_hip = int(eng.best_action(frames=fr, frame_stride=frames.shape[1], frame_ids=one, idx_network=idx).item())
        assert a_hip == oracle.best_action(p, ref.state[0], idx)


def _s8_positive(region: torch.Tensor, shape):
    """[shape] bool: hi half of an S8 tensor > 0 (S8: every 8 floats hold 8 bf16 hi halves, then 8 lo halves; csrc/gemm_core.h)."""
    n = int(np.prod(shape))
    u16 = region.cpu().numpy()[:n].view(np.uint16).reshape(-1, 16)[:, :8].reshape(shape)
    return ((u16 & 0x7FFF) != 0) & ((u16 & 0x8000) == 0)


def _masked_impala_grads(eng, params, feats, K, A, B, layer_norm, ref, hip_targets):
    """float64 torch gradients of the iS-DQN loss through the impala torso for the B online images, every ReLU mask and max-pool
    winner taken from the HIP path's workspace (first B rows of its forward tensors); targets = the HIP path's own (stop-gradient)."""
    import torch.nn.functional as F

    P = {m: {k: torch.tensor(np.asarray(v), dtype=torch.float64, requires_grad=True) for k, v in d.items()} for m, d in params.items()}
    f64 = lambda name, shape: torch.tensor(eng.region(name).cpu().numpy()[: int(np.prod(shape))].reshape(shape).astype(np.float64))

    def ln(z, name):
        if not layer_norm:
            return z
        mean = z.mean(-1, keepdim=True)
        var = ((z * z).mean(-1, keepdim=True) - mean * mean).clamp_min(0)
        return (z - mean) * torch.rsqrt(var + 1e-6) * P[name]["scale"] + P[name]["bias"]

    def conv(x, name):
        k = P[name]["kernel"]
        y = F.conv2d(F.pad(x.permute(0, 3, 1, 2), (1, 1, 1, 1)), k.permute(3, 2, 0, 1), P[name]["bias"])
        return y.permute(0, 2, 3, 1)

    x = torch.tensor(np.asarray(ref.state), dtype=torch.float64) / 255.0
    n_ln = 0
    for s in range(3):
        C = feats[s]
        Cp = (C + 7) // 8 * 8
        z0 = conv(x, f"Stack_{s}/Conv_0")
        H, W = z0.shape[1], z0.shape[2]
        Hp, Wp = -(-H // 2), -(-W // 2)
        pad = max((Hp - 1) * 2 + 3 - H, 0) // 2
        n2 = eng.batch_size * 2
        arg = eng.region(f"imp/s{s}/argmax").view(torch.uint8).cpu().numpy()[: n2 * Hp * Wp * Cp].reshape(n2, Hp, Wp, Cp)[:B, :, :, :C].astype(np.int64)
        oy, ox = np.meshgrid(np.arange(Hp), np.arange(Wp), indexing="ij")
        iy = torch.tensor(2 * oy[None, :, :, None] - pad + arg // 3)
        ix = torch.tensor(2 * ox[None, :, :, None] - pad + arg % 3)
        bi = torch.arange(B)[:, None, None, None].expand_as(iy)
        ci = torch.arange(C)[None, None, None, :].expand_as(iy)
        x = z0[bi, iy, ix, ci]  # the HIP path's max-pool winners
        for b in range(2):
            r = x
            r_hip = f64(f"imp/s{s}/r{b}", (n2, Hp, Wp, Cp))[:B, :, :, :C]
            with torch.no_grad():
                keep = (ln(r_hip, f"Stack_{s}/LayerNorm_{b}") > 0).to(torch.float64)
            x = ln(r, f"Stack_{s}/LayerNorm_{b}") * keep
            keep2 = torch.tensor(_s8_positive(eng.region(f"imp/s{s}/a2_{b}"), (n2, Hp, Wp, Cp))[:B, :, :, :C].astype(np.float64))
            x = conv(x, f"Stack_{s}/Conv_{1 + 2 * b}") * keep2
            x = conv(x, f"Stack_{s}/Conv_{2 + 2 * b}") + r
    C, Cp = feats[2], (feats[2] + 7) // 8 * 8
    r_hip = f64("z/Impala", (B, x.shape[1], x.shape[2], Cp))[:, :, :, :C]
    name = f"LayerNorm_{n_ln}"
    with torch.no_grad():
        keep = (ln(r_hip, name) > 0).to(torch.float64)
    h = (ln(x, name) * keep).reshape(B, -1)
    n_ln += 1 if layer_norm else 0
    n_dense = 0
    for width in feats[3:]:
        z = h @ P[f"Dense_{n_dense}"]["kernel"] + P[f"Dense_{n_dense}"]["bias"]
        wp = (width + 7) // 8 * 8
        zh = f64(f"z/Dense_{n_dense}", (B, wp))[:, :width]
        name = f"LayerNorm_{n_ln}"
        with torch.no_grad():
            keep = (ln(zh, name) > 0).to(torch.float64)
        h = ln(z, name) * keep
        n_ln += 1 if layer_norm else 0
        n_dense += 1
    q = (h @ P[f"Dense_{n_dense}"]["kernel"] + P[f"Dense_{n_dense}"]["bias"]).reshape(B, 1 + K, A)
    act = torch.tensor(np.asarray(ref.action), dtype=torch.long)
    qv = q[:, 1:, :][torch.arange(B), :, act]
    ((qv - torch.tensor(hip_targets.astype(np.float64))) ** 2).mean(0).sum().backward()
    return {m: {k: v.grad.numpy() for k, v in d.items()} for m, d in P.items()}


def test_impala_agent_surface_and_nested_model_export():
    from slimdqn.networks.isdqn import iSDQN

    agent = iSDQN(0, (84, 84, 4), 4, 2, [8, 8, 8, 16], True, False, "impala", 1e-3, 0.99, 1, 1, 4, batch_size=4)
    model = agent.get_model()["params"]["params"]
    # Flax nests the Stack's modules (dqn.py:7-36): params["Stack_1"]["Conv_3"]["kernel"]
    assert model["Stack_1"]["Conv_3"]["kernel"].shape == (3, 3, 8, 8) and model["Stack_0"]["Conv_0"]["kernel"].shape == (3, 3, 4, 8)
    assert model["Stack_2"]["LayerNorm_1"]["scale"].shape == (8,) and model["LayerNorm_0"]["scale"].shape == (8,)
    assert model["Dense_0"]["kernel"].shape == (11 * 11 * 8, 16) and model["Dense_1"]["kernel"].shape == (16, 3 * 4)
    state = np.random.default_rng(0).integers(0, 256, (84, 84, 4)).astype(np.float32)
    assert 0 <= agent.best_action(agent.params, state, key=1) < 4


def test_entry_point_with_the_impala_torso(tmp_path):
    """experiments/atari/isdqn.py -at impala (launch_job/atari/launch_time.sh:13 times cnn AND impala): trainer end to end."""
    import json
    import pickle

    from experiments.atari.isdqn import run

    argv = ["-en", "imp_Synthetic", "-s", "1", "-dw", "-f", "8", "8", "8", "16", "-rbc", "200", "-bs", "4", "-n", "1", "-horizon", "30", "-at", "impala",
            "-ne", "1", "-ntspe", "48", "-utd", "4", "-nis", "16", "-ed", "100", "-nbi", "2", "-ln", "-tuf", "16", "-env", "synthetic"]
    gathered = run(argv, root=str(tmp_path))
    assert len(gathered) == 1
    out = tmp_path / "atari" / "exp_output" / "imp_Synthetic"
    assert json.load(open(out / "parameters.json"))["shared_parameters"]["architecture_type"] == "impala"
    model = pickle.load(open(out / "isdqn" / "models" / "1", "rb"))["params"]
    assert model["params"]["Stack_2"]["Conv_4"]["kernel"].shape == (3, 3, 8, 8)
