"""GPU parity of the Q-network path against the CPU oracle (fp32/fp64 torch restatement of the
reference, PARITY UNPINNED for Flax/Optax numerics -- see oracle/network.py), through the C ABI.

Tolerances: BASELINE.md section 4 / north_star: Bellman targets and per-head losses within 1e-3
of the fp32 restatement for the default (split-bf16) precision.  The single-pass bf16 mode is
checked against a correspondingly looser bound (bf16 has 8 significant bits)."""
import numpy as np
import pytest
import torch

from tests.gpu_helpers import device_batch, make_frame_batch, make_pair, masked_reference_grads

pytestmark = pytest.mark.gpu

TOL = {"bf16x3": dict(q=1e-3, loss=1e-3, grad=3e-3, param=2e-5), "bf16": dict(q=8e-2, loss=5e-2, grad=2.5e-1, param=1e-3)}

CONFIGS = [
    # feats, K, A, B, layer_norm
    pytest.param(((7, 9, 11, 13), 3, 5, 6, True), id="tiny-ln"),
    pytest.param(((16, 20, 5, 24), 2, 3, 5, False), id="tiny-noln"),
    pytest.param(((32, 64, 64, 512), 9, 9, 8, True), id="headline-arch-B8"),
    # odd batch: ragged last tiles of every image / row tiling (the head chain runs S = 1 transition per workgroup here)
    pytest.param(((32, 64, 64, 512), 9, 9, 7, True), id="headline-arch-B7-ragged"),
    # configs[4] head shape (K=32, A=4: 132 outputs -> 9 column tiles, padded to 136)
    pytest.param(((32, 64, 64, 512), 32, 4, 12, True), id="c5-head-B12"),
    # 18 actions (full Atari set), B crossing no 64-row tile boundary evenly
    pytest.param(((32, 64, 64, 512), 4, 18, 33, True), id="a18-B33"),
    pytest.param(((32, 64, 64, 512), 3, 6, 10, False), id="headline-arch-noln"),
    # the widths of the reference's own smoke test (tests/test_atari.py:25-29: features 2 3 1 15, batch 3): one- to three-channel
    # conv layers inside 8-channel padding, LayerNorm over a single channel
    pytest.param(((2, 3, 1, 15), 2, 3, 3, True), id="reference-smoke-widths-2-3-1-15-B3"),
    pytest.param(((2, 3, 1, 15), 2, 3, 3, False), id="reference-smoke-widths-noln"),
]

# Kernel instantiations that only large batches select (BASELINE configs[4]: B = 1024, K = 32, A = 4): the head chain
# runs S = 2 transitions per workgroup from B = 512 and S = 4 from B = 1024 (net_kernels.hip: `hc_S`), the dense
# forward / data-gradient / weight-gradient tilings change their split factors, and the image-resident weight
# gradients group more images per workgroup.  Tiny features keep the CPU oracle at seconds; one case runs the
# headline widths so that the 64-channel kernels (DenseDgradLN, conv <4, ...>) see B >= 512 against the oracle too.
LARGE_BATCH = [
    pytest.param(((8, 8, 8, 16), 3, 4, 512, True), id="tiny-B512-S2"),
    pytest.param(((8, 8, 8, 16), 32, 4, 1024, True), id="tiny-c5head-B1024-S4"),
    pytest.param(((8, 8, 8, 16), 9, 9, 515, True), id="tiny-B515-ragged-S2"),
    pytest.param(((8, 8, 8, 16), 2, 3, 1030, False), id="tiny-noln-B1030-ragged-S4"),
    pytest.param(((32, 64, 64, 512), 3, 4, 512, True), id="headline-arch-B512-S2"),
    # two images per weight-gradient workgroup with a one-image last group: the register prefetch of the next image (conv_img.h:
    # request_image_s8 / commit_image_s8) has nothing to request there; the backward forks in round 4's order (B <= 768)
    pytest.param(((32, 64, 64, 512), 3, 4, 131, True), id="headline-arch-B131-ragged-wgrad-groups"),
    # BASELINE configs[4] at its full size (the torch oracle needs ~10 s for it on the GPU box's host cores)
    pytest.param(((32, 64, 64, 512), 32, 4, 1024, True), id="c5-full-size-B1024-K32-A4"),
]


def _z_width(eng, name):
    """per-image element count of the z region of a layer (internal layout: channels padded to 8)"""
    info = {"Conv_0": 0, "Conv_1": 1, "Conv_2": 2}
    if name in info:
        c = eng.features[info[name]]
        hw = eng.observation_dim[0]
        for k, s in ((8, 4), (4, 2), (3, 1))[: info[name] + 1]:
            hw = -(-hw // s)
        return hw * hw * ((c + 7) // 8 * 8), c
    c = eng.features[3 + int(name.split("_")[1])]
    return (c + 7) // 8 * 8, c


def _flat_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-12)


# other observation geometries: (obs (h, w, stack), feats, K, A, B)
GEOMETRIES = [
    # a 6-frame stack: the image-resident uint8 kernels hold at most 4 frame ids, so the first layer takes the generic engine
    pytest.param(((44, 44, 6), (7, 9, 11, 13), 2, 3, 5), id="44x44x6-generic-first-layer"),
    # non-square frames, 2-frame stack (image-resident, padded width not a multiple of the stride)
    pytest.param(((52, 60, 2), (16, 20, 12, 24), 3, 4, 6), id="52x60x2"),
    pytest.param(((84, 84, 1), (32, 64, 64, 64), 2, 6, 4), id="84x84x1"),
]


@pytest.mark.parametrize("geo", GEOMETRIES)
def test_other_observation_geometries(geo):
    """Forward, loss terms, first-step gradients and one Adam step against the oracle for observation shapes other
    than the Atari 84x84x4 (dqn.py:49-72 is geometry-agnostic: SAME padding, any stack depth)."""
    obs, feats, K, A, B = geo
    tol = TOL["bf16x3"]
    h, w, stack = obs
    oracle, eng, params = make_pair(feats, K, A, B, obs=obs, seed=5)
    frames, ids, action, reward, terminal, ref = make_frame_batch(B, A, seed=17, h=h, w=w, stack=stack)
    batch = device_batch(eng, frames, ids, action, reward, terminal)
    all_q = oracle.apply(oracle.params, torch.cat((torch.tensor(ref.state), torch.tensor(ref.next_state)))).detach().numpy()
    flat_ids = np.concatenate([ids[:, :stack], ids[:, stack:]], 0).copy()
    q = eng.forward(frames=batch._keep[0], frame_stride=frames.shape[1], frame_ids=torch.from_numpy(flat_ids).cuda(), n_rows=2 * B)
    q = q.cpu().numpy().reshape(2 * B, 1 + K, A)
    assert np.abs(q - all_q).max() < tol["q"], f"forward max err {np.abs(q - all_q).max()}"
    o_q, o_t, o_td = oracle.loss_terms(oracle.params, ref)
    o_grads, _ = oracle.grads(oracle.params, ref)
    _, _, o_losses = oracle.learn_on_batch(oracle.params, oracle.optimizer_state, ref)
    grad = torch.zeros_like(eng.params)
    losses = eng.learn_on_batch(batch, grad_out=grad).cpu().numpy()
    assert np.abs(eng.targets.cpu().numpy() - o_t.detach().numpy()).max() < tol["q"]
    assert np.abs(eng.q_values.cpu().numpy() - o_q.detach().numpy()).max() < tol["q"]
    assert np.abs(losses - o_losses).max() < tol["loss"] * max(1.0, np.abs(o_losses).max())
    g = eng.internal_to_flax_grads(grad)
    for mod in o_grads:
        for leaf in o_grads[mod]:
            e = _flat_err(g[mod][leaf], o_grads[mod][leaf].numpy())
            assert e < tol["grad"], f"grad {mod}/{leaf}: rel err {e}"


@pytest.mark.parametrize("cfg", LARGE_BATCH)
def test_large_batch_kernel_variants(cfg):
    """Forward, targets / losses, first-step gradients of every leaf and one Adam step against the oracle at the batch
    sizes that switch the kernels to their B >= 512 / B >= 1024 instantiations (isdqn.py:92-109)."""
    feats, K, A, B, ln = cfg
    tol = TOL["bf16x3"]
    oracle, eng, params = make_pair(feats, K, A, B, layer_norm=ln, seed=7)
    frames, ids, action, reward, terminal, ref = make_frame_batch(B, A, seed=23, n_frames=B + 64)
    batch = device_batch(eng, frames, ids, action, reward, terminal)
    with torch.no_grad():
        all_q = oracle.apply(oracle.params, torch.cat((torch.tensor(ref.state), torch.tensor(ref.next_state)))).numpy()
    flat_ids = np.concatenate([ids[:, :4], ids[:, 4:]], 0).copy()
    q = eng.forward(frames=batch._keep[0], frame_stride=frames.shape[1], frame_ids=torch.from_numpy(flat_ids).cuda(), n_rows=2 * B)
    q = q.cpu().numpy().reshape(2 * B, 1 + K, A)
    assert np.abs(q - all_q).max() < tol["q"], f"forward max err {np.abs(q - all_q).max()}"
    o_q, o_t, o_td = oracle.loss_terms(oracle.params, ref)
    o_grads, _ = oracle.grads(oracle.params, ref)
    p, st, o_losses = oracle.learn_on_batch(oracle.params, oracle.optimizer_state, ref)
    grad = torch.zeros_like(eng.params)
    losses = eng.learn_on_batch(batch, grad_out=grad).cpu().numpy()
    assert np.abs(eng.targets.cpu().numpy() - o_t.detach().numpy()).max() < tol["q"]
    assert np.abs(eng.q_values.cpu().numpy() - o_q.detach().numpy()).max() < tol["q"]
    assert np.abs(losses - o_losses).max() < tol["loss"] * max(1.0, np.abs(o_losses).max())
    pri = eng.priorities.cpu().numpy()
    exp = np.sqrt(o_td.detach().numpy().mean(1) + 1e-10)
    assert np.abs(pri - exp).max() < 10 * tol["q"] * max(1.0, exp.max())
    # Gradients against the float64 reference whose ReLU decisions are pinned to the HIP path's own pre-activations
    # (gpu_helpers.masked_reference_grads: a batch this large always holds a few ReLU inputs within the forward's 1e-5
    # of zero, and one differing decision moves the leaves by O(1/B) -- as much as a missing image would).  What is left
    # is arithmetic: 1e-4 of the leaf's maximum, 30 x tighter than the small-batch bound.
    names = ["Conv_0", "Conv_1", "Conv_2"] + [f"Dense_{i}" for i in range(len(feats) - 3)]
    z_hip = {}
    for n in names:
        width, c = _z_width(eng, n)  # internal layout [B][pixels][channels padded to 8] -> true channels
        z = eng.region("z/" + n).cpu().numpy()[: B * width].reshape(B, -1, (c + 7) // 8 * 8)
        z_hip[n] = z[:, :, :c].reshape(B, -1)
    m_grads = masked_reference_grads(params, feats, K, A, ref, z_hip, layer_norm=ln, gamma_n=0.99)
    g = eng.internal_to_flax_grads(grad)
    for mod in m_grads:
        for leaf in m_grads[mod]:
            e = _flat_err(g[mod][leaf], m_grads[mod][leaf])
            assert e < 1e-4, f"grad {mod}/{leaf}: rel err {e} against the mask-pinned reference"
    # and the independent oracle (own ReLU decisions): Euclidean error, insensitive to isolated flips
    for mod in o_grads:
        for leaf in o_grads[mod]:
            a, b = np.asarray(g[mod][leaf], np.float64), o_grads[mod][leaf].numpy().astype(np.float64)
            assert np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-12) < 10 * tol["grad"], f"grad {mod}/{leaf} vs oracle"
    # Adam (optax.adam, isdqn.py:46, 85-86): the first step moves every parameter by lr * g / (|g| + eps), a sign-like
    # function of g -- so the update is checked against Adam applied to the gradient the HIP path itself reported
    # (arithmetic of the update, 1e-6), and against the oracle's parameters within one full step (2 lr: an entry whose
    # gradient is at rounding level may legitimately move the other way).
    got = eng.export_flax()
    lr, eps = 1e-3, 1.5e-4
    for mod in p:
        for leaf in p[mod]:
            gh = np.asarray(g[mod][leaf], np.float64)
            exp = np.asarray(params[mod][leaf], np.float64) - lr * gh / (np.abs(gh) + eps)
            d = np.abs(got[mod][leaf] - exp).max()
            assert d < 2e-6, f"param {mod}/{leaf}: {d} off Adam(own gradient)"
            assert np.abs(got[mod][leaf] - p[mod][leaf].numpy()).max() < 2.001 * lr, f"param {mod}/{leaf} vs oracle"
    assert int(eng.adam_count.item()) == 1


@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
@pytest.mark.parametrize("cfg", CONFIGS)
def test_forward_loss_grad_adam(cfg, precision):
    feats, K, A, B, ln = cfg
    tol = TOL[precision]
    oracle, eng, params = make_pair(feats, K, A, B, layer_norm=ln, precision=precision, seed=3)
    frames, ids, action, reward, terminal, ref = make_frame_batch(B, A, seed=11)
    batch = device_batch(eng, frames, ids, action, reward, terminal)

    # ---- forward over concat(state, next_state) ----
    all_q = oracle.apply(oracle.params, torch.cat((torch.tensor(ref.state), torch.tensor(ref.next_state)))).detach().numpy()
    flat_ids = np.concatenate([ids[:, :4], ids[:, 4:]], 0).copy()
    q = eng.forward(frames=batch._keep[0], frame_stride=frames.shape[1], frame_ids=torch.from_numpy(flat_ids).cuda(), n_rows=2 * B)
    q = q.cpu().numpy().reshape(2 * B, 1 + K, A)
    assert np.abs(q - all_q).max() < tol["q"], f"forward max err {np.abs(q - all_q).max()}"

    # ---- loss / targets (isdqn.py:92-109) ----
    o_q, o_t, o_td = oracle.loss_terms(oracle.params, ref)
    o_loss = o_td.mean(0).detach().numpy()
    losses = eng.loss_on_batch(batch).cpu().numpy()
    assert np.abs(eng.targets.cpu().numpy() - o_t.detach().numpy()).max() < tol["q"]
    assert np.abs(eng.q_values.cpu().numpy() - o_q.detach().numpy()).max() < tol["q"]
    assert np.abs(losses - o_loss).max() < tol["loss"] * max(1.0, np.abs(o_loss).max())

    # ---- gradient step: gradients (debug hook), Adam, three steps ----
    grad = torch.zeros_like(eng.params)
    p, st = oracle.params, oracle.optimizer_state
    for step in range(3):
        o_grads, _ = oracle.grads(p, ref)
        p, st, o_losses = oracle.learn_on_batch(p, st, ref)
        losses = eng.learn_on_batch(batch, grad_out=grad).cpu().numpy()
        # single-pass bf16 gradients are 10-25 % off, so its trajectories drift after the first update
        loss_tol = tol["loss"] if (precision == "bf16x3" or step == 0) else 0.3
        assert np.abs(losses - o_losses).max() < loss_tol * max(1.0, np.abs(o_losses).max()), f"step {step}"
        if step == 0:
            g = eng.internal_to_flax_grads(grad)
            for mod in o_grads:
                for leaf in o_grads[mod]:
                    e = _flat_err(g[mod][leaf], o_grads[mod][leaf].numpy())
                    assert e < tol["grad"], f"grad {mod}/{leaf}: rel err {e}"
            pri = eng.priorities.cpu().numpy()
            exp = np.sqrt(o_td.detach().numpy().mean(1) + 1e-10)
            assert np.abs(pri - exp).max() < 10 * tol["q"] * max(1.0, exp.max())
    got = eng.export_flax()
    for mod in p:
        for leaf in p[mod]:
            d = np.abs(got[mod][leaf] - p[mod][leaf].numpy()).max()
            # Adam normalises the step: an element whose gradient is ~0 can flip sign of its update, so the
            # bound is a few learning rates, tightened by the gradient accuracy
            assert d < 3 * 1e-3 * 3 * max(tol["grad"] * 50, 0.02) + tol["param"], f"param {mod}/{leaf}: {d}"
    assert int(eng.adam_count.item()) == 3


@pytest.mark.parametrize("delta", [0.25, 1.0])
def test_huber_loss_option_matches_the_oracle(delta):
    """cfg.huber_delta > 0 (include/isdqn_hip.h): Huber loss instead of the reference's squared TD error -- losses, priorities
    and the first-step gradients against the oracle with the same option, through both learn paths (head chain at B = 8,
    td_kernel in loss_on_batch)."""
    from oracle.isdqn import iSDQN as OracleAgent
    from slimdqn._engine import QNetEngine
    from tests.gpu_helpers import perturbed_params

    feats, K, A, B = (32, 64, 64, 512), 3, 5, 8
    params = perturbed_params(3, (84, 84, 4), feats, "cnn", (1 + K) * A, True)
    oracle = OracleAgent(3, (84, 84, 4), A, K, list(feats), True, False, "cnn", 1e-3, 0.99, 1, 1, 1, adam_eps=1.5e-4, params=params, huber_delta=delta)
    eng = QNetEngine((84, 84, 4), A, 1 + K, feats, "cnn", True, B, gamma_n=0.99, learning_rate=1e-3, adam_eps=1.5e-4, huber_delta=delta)
    eng.import_flax(params)
    frames, ids, action, reward, terminal, ref = make_frame_batch(B, A, seed=11)
    reward = (3.0 * reward).astype(np.float32)  # TD errors on both sides of delta
    ref = ref.__class__(state=ref.state, action=ref.action, reward=reward.astype(np.float64), next_state=ref.next_state, is_terminal=ref.is_terminal)
    batch = device_batch(eng, frames, ids, action, reward, terminal)
    o_q, o_t, o_td = oracle.loss_terms(oracle.params, ref)
    d = (o_q - o_t).abs()
    assert (d > delta).any() and (d < delta).any()
    pre = eng.loss_on_batch(batch).cpu().numpy().copy()
    assert np.abs(pre - o_td.mean(0).detach().numpy()).max() < 1e-3 * max(1.0, float(o_td.mean(0).max()))
    o_grads, _ = oracle.grads(oracle.params, ref)
    grad = torch.zeros_like(eng.params)
    losses = eng.learn_on_batch(batch, grad_out=grad).cpu().numpy()
    assert np.abs(losses - o_td.mean(0).detach().numpy()).max() < 1e-3 * max(1.0, float(o_td.mean(0).max()))
    exp = np.sqrt(o_td.detach().numpy().mean(1) + 1e-10)
    assert np.abs(eng.priorities.cpu().numpy() - exp).max() < 1e-2 * max(1.0, exp.max())
    # gradients against the float64 reference with the HIP path's own ReLU decisions (see test_large_batch_kernel_variants)
    z_hip = {}
    for n in ("Conv_0", "Conv_1", "Conv_2", "Dense_0"):
        width, c = _z_width(eng, n)
        z_hip[n] = eng.region("z/" + n).cpu().numpy()[: B * width].reshape(B, -1, (c + 7) // 8 * 8)[:, :, :c].reshape(B, -1)
    m_grads = masked_reference_grads(params, feats, K, A, ref, z_hip, layer_norm=True, gamma_n=0.99, huber_delta=delta)
    g = eng.internal_to_flax_grads(grad)
    for mod in m_grads:
        for leaf in m_grads[mod]:
            e = _flat_err(g[mod][leaf], m_grads[mod][leaf])
            assert e < 1e-4, f"grad {mod}/{leaf}: rel err {e} against the mask-pinned reference"
    for mod in o_grads:  # and the independent oracle, Euclidean (a ReLU decision may differ: 1/B of a leaf)
        for leaf in o_grads[mod]:
            a, b = np.asarray(g[mod][leaf], np.float64), o_grads[mod][leaf].numpy().astype(np.float64)
            assert np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-12) < 10 * TOL["bf16x3"]["grad"], f"grad {mod}/{leaf} vs oracle"


def test_shift_and_best_action():
    feats, K, A, B = (7, 9, 11, 13), 4, 6, 4
    oracle, eng, params = make_pair(feats, K, A, B, seed=5)
    frames, ids, action, reward, terminal, ref = make_frame_batch(B, A, seed=2)
    fr = torch.from_numpy(frames).cuda()
    one = torch.from_numpy(ids[:1, :4].copy()).cuda()
    q_before = eng.forward(frames=fr, frame_stride=frames.shape[1], frame_ids=one, n_rows=1).cpu().numpy().reshape(1 + K, A)
    for idx in range(K):
        a = int(eng.best_action(frames=fr, frame_stride=frames.shape[1], frame_ids=one, idx_network=idx).item())
        assert a == int(np.argmax(q_before[1 + idx]))
        assert a == oracle.best_action(oracle.params, ref.state[0], idx)
    eng.shift_params()
    q_after = eng.forward(frames=fr, frame_stride=frames.shape[1], frame_ids=one, n_rows=1).cpu().numpy().reshape(1 + K, A)
    # test_isdqn.py:99-116: the target heads equal the former online heads exactly; last head unchanged
    assert np.linalg.norm(q_after[:-1] - q_before[1:]) == 0
    np.testing.assert_array_equal(q_after[-1], q_before[-1])
    shifted = oracle.shift_params(oracle.params)
    got = eng.export_flax()
    np.testing.assert_array_equal(got["Dense_1"]["kernel"], shifted["Dense_1"]["kernel"].numpy())
    np.testing.assert_array_equal(got["Dense_1"]["bias"], shifted["Dense_1"]["bias"].numpy())


def test_param_layout_round_trip():
    feats, K, A, B = (7, 9, 11, 13), 2, 3, 2
    oracle, eng, params = make_pair(feats, K, A, B, seed=1)
    got = eng.export_flax()
    for mod in params:
        for leaf in params[mod]:
            np.testing.assert_array_equal(got[mod][leaf], params[mod][leaf])


FC_SHAPES = [
    # (feats, K, A, B), adam_eps
    pytest.param(((100, 100), 1, 4, 32, 1e-8), id="lunar-lander-100x100"),   # BASELINE config 1
    pytest.param(((64,), 2, 3, 9, 1e-8), id="one-hidden-layer-on-raw-observations"),
    # hidden width above 512: two column passes and two K-step groups in the head chain kernel; ragged batch.
    # (Atari epsilon: with 1e-8 any of the 180k parameters whose gradient is at rounding-noise level moves by a full
    # +-lr per step, and the comparison would measure that amplification, not the kernels)
    pytest.param(((300, 600), 3, 5, 17, 1.5e-4), id="wide-hidden-600"),
]


@pytest.mark.parametrize("precision", ["bf16x3"])
@pytest.mark.parametrize("shape", FC_SHAPES)
def test_fc_architecture_lunar_lander_shape(shape, precision):
    """fc torso (BASELINE config 1 shape: fc [100,100], K=1, batch 32, 8-dim observations, 4 actions) and variants."""
    from slimdqn._engine import QNetEngine
    from oracle.isdqn import iSDQN as OracleAgent
    from oracle.replay_buffer import ReplayElement
    from tests.gpu_helpers import perturbed_params

    (feats, K, A, B, eps), obs = shape, (8,)
    params = perturbed_params(0, obs, feats, "fc", (1 + K) * A, True)
    oracle = OracleAgent(0, obs, A, K, list(feats), True, False, "fc", 3e-4, 0.99, 1, 1, 1, adam_eps=eps, params=params)
    eng = QNetEngine(obs, A, 1 + K, feats, "fc", True, B, gamma_n=0.99, learning_rate=3e-4, adam_eps=eps, precision=precision)
    eng.import_flax(params)
    rng = np.random.default_rng(0)
    state = rng.normal(size=(B, 8)).astype(np.float32)
    nxt = rng.normal(size=(B, 8)).astype(np.float32)
    action = rng.integers(0, A, B).astype(np.int32)
    reward = rng.normal(size=B).astype(np.float32)
    term = (rng.random(B) < 0.2).astype(np.uint8)
    ref = ReplayElement(state=state, action=action.astype(np.int64), reward=reward.astype(np.float64), next_state=nxt, is_terminal=term.astype(np.int64))
    d = lambda a: torch.from_numpy(a).cuda()
    batch = eng.make_batch(state=d(state), next_state=d(nxt), action=d(action), reward=d(reward), terminal=d(term))
    p, st = oracle.params, oracle.optimizer_state
    for step in range(2):
        p, st, o_losses = oracle.learn_on_batch(p, st, ref)
        losses = eng.learn_on_batch(batch).cpu().numpy()
        assert np.abs(losses - o_losses).max() < 1e-3 * max(1.0, np.abs(o_losses).max())
    got = eng.export_flax()
    for mod in p:
        for leaf in p[mod]:
            assert np.abs(got[mod][leaf] - p[mod][leaf].numpy()).max() < 3e-4, (mod, leaf)


def test_first_layer_pair_kernel_is_bit_identical_to_the_one_tile_kernel():
    """csrc/conv_u8_pair.h (first layer: two pixel tiles per workgroup, the second one's frame rows prefetched into registers) and
    csrc/conv_s8_pair.h (second layer: two images per workgroup, the second one prefetched) against the one-tile / one-image kernels
    they replace at the Nature geometry (-DISDQN_NO_U8_PAIR build of the same sources): parameters, Adam moments, losses,
    q-values, targets and priorities after three learn steps, the loss-only pass and a one-row forward of eight cnn shapes (c2 and c5
    at full size among them) hash to the same bits (scripts/r2/bits.py)."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "is-dqn_amd"))
    import build

    old = build.build(verbose=False, variant="nopair", defines=("ISDQN_NO_U8_PAIR",))
    outs = []
    for lib in (old, build.LIB):
        r = subprocess.run([sys.executable, os.path.join(root, "scripts", "r2", "bits.py")], env=dict(os.environ, ISDQN_HIP_LIB=lib),
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([l for l in r.stdout.splitlines() if "params" in l])
    assert len(outs[0]) >= 8 and outs[0] == outs[1]
