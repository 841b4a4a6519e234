"""GPU parity of the representation diagnostics (--analysis): isdqn_net_analysis against the oracle's AnalysisNet restatement
(slimdqn/utils/analysis_architecture.py:46-122), and the two host formulas against the oracle's (utils/analysis.py:4-17)."""
import json

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("feats,arch,obs,n_rows,B", [
    ((8, 12, 16, 24), "cnn", (84, 84, 4), 37, 32),
    ((32, 64, 64, 512), "cnn", (84, 84, 4), 64, 32),
    ((40, 24), "fc", (11,), 50, 32),
])
def test_analysis_net_matches_oracle(feats, arch, obs, n_rows, B):
    from oracle import analysis as oa
    from oracle import network as onet
    from slimdqn._engine import QNetEngine
    from tests.gpu_helpers import perturbed_params

    A, K = 5, 3
    params = perturbed_params(11, obs, feats, arch, (1 + K) * A, True)
    eng = QNetEngine(obs, A, 1 + K, feats, arch, True, B, gamma_n=0.99, learning_rate=1e-3, adam_eps=1.5e-4)
    eng.import_flax(params)
    rng = np.random.default_rng(3)
    if arch == "cnn":
        states = rng.integers(0, 256, (n_rows,) + obs, dtype=np.uint8)
        planes = torch.from_numpy(np.ascontiguousarray(np.moveaxis(states, -1, 1)).reshape(n_rows * obs[2], obs[0] * obs[1])).cuda()
        ids = torch.arange(n_rows * obs[2], dtype=torch.int32, device="cuda")
        feat, scores = eng.analysis(frames=planes, frame_stride=obs[0] * obs[1], frame_ids=ids, n_rows=n_rows)
    else:
        states = rng.normal(size=(n_rows,) + obs).astype(np.float32)
        feat, scores = eng.analysis(obs=torch.from_numpy(states).cuda(), n_rows=n_rows)
    o_feat, o_scores = oa.analysis_net(onet.to_torch(params, torch.float64), states, list(feats), arch, True)
    assert feat.shape == o_feat.shape and len(scores) == len(o_scores)
    np.testing.assert_allclose(feat.cpu().numpy(), o_feat, atol=1e-3, rtol=0)  # the 1e-3 bar of the forward
    for s, o in zip(scores, o_scores):
        assert s.numel() == o.size
        np.testing.assert_allclose(s.cpu().numpy(), o, atol=1e-3 * n_rows, rtol=0)
    # the host formulas on the device results: same srank (an integer) unless a singular value sits on the threshold
    from slimdqn.utils.analysis import compute_dead_neurons, compute_srank

    assert abs(compute_srank(feat.cpu().numpy()) - oa.compute_srank(o_feat)) <= 1
    assert compute_dead_neurons([s.cpu().numpy() for s in scores]) == pytest.approx(oa.compute_dead_neurons(o_scores), abs=2e-3)


def test_analysis_flag_of_the_trainer(tmp_path):
    from experiments.atari.isdqn import run

    argv = ["-en", "ana_Synthetic", "-s", "1", "-dw", "-f", "8", "8", "8", "16", "-rbc", "300", "-bs", "8", "-horizon", "40", "-at", "cnn",
            "-ne", "1", "-ntspe", "80", "-utd", "4", "-nis", "30", "-ed", "100", "-nbi", "2", "-ln", "-tuf", "16", "-env", "synthetic", "-a"]
    run(argv, root=str(tmp_path))
    logs = json.load(open(tmp_path / "atari" / "exp_output" / "ana_Synthetic" / "isdqn" / "analysis" / "1.json"))
    assert set(logs) == {"srank", "dead_neurons"} and len(logs["srank"]) == len(logs["dead_neurons"]) >= 2
    assert all(1 <= s <= 16 for s in logs["srank"]) and all(0.0 <= d <= 1.0 for d in logs["dead_neurons"])


@pytest.mark.parametrize("feats,arch,obs,n_rows,B,ln,bn", [
    ((8, 16, 8, 24), "impala", (84, 84, 4), 21, 16, True, False),     # 12 block activations + torso output + Dense_0
    ((8, 12, 16, 24), "cnn", (84, 84, 4), 30, 16, True, True),        # BatchNorm: batch statistics of the analysed rows
    ((40, 24), "fc", (11,), 50, 32, False, True),
    ((8, 16, 8, 24), "impala", (36, 36, 2), 18, 16, True, True),
])
def test_analysis_net_of_impala_and_batchnorm_networks_matches_oracle(feats, arch, obs, n_rows, B, ln, bn):
    """AnalysisNet (analysis_architecture.py:9-122) for the impala torso -- the two ReLU outputs of each residual block of each Stack,
    then the flattened torso output, then the Dense layers -- and for BatchNorm networks, which the reference applies here with a
    mutable batch_stats collection and use_running_average left False (srank_and_dead_neurons.py:17): batch statistics of the analysed
    rows, sums in front of each BatchNorm, the feature matrix behind the last one."""
    from oracle import analysis as oa
    from oracle import network as onet
    from slimdqn._engine import QNetEngine
    from tests.gpu_helpers import perturbed_params

    A, K = 5, 3
    params = perturbed_params(11, obs, feats, arch, (1 + K) * A, ln, batch_norm=bn)
    eng = QNetEngine(obs, A, 1 + K, feats, arch, ln, B, gamma_n=0.99, learning_rate=1e-3, adam_eps=1.5e-4, batch_norm=bn)
    stats = None
    if bn:  # non-trivial running averages: they must NOT enter (training-mode statistics)
        rng0 = np.random.default_rng(5)
        stats = {m: {"mean": rng0.normal(0, 0.5, l["mean"].shape).astype(np.float32), "var": rng0.uniform(0.5, 2.0, l["var"].shape).astype(np.float32)}
                 for m, l in onet.init_batch_stats(params).items()}
    eng.import_flax(params, batch_stats=stats)
    rng = np.random.default_rng(3)
    if arch == "fc":
        states = rng.normal(size=(n_rows,) + obs).astype(np.float32)
        feat, scores = eng.analysis(obs=torch.from_numpy(states).cuda(), n_rows=n_rows)
    else:
        states = rng.integers(0, 256, (n_rows,) + obs, dtype=np.uint8)
        planes = torch.from_numpy(np.ascontiguousarray(np.moveaxis(states, -1, 1)).reshape(n_rows * obs[2], obs[0] * obs[1])).cuda()
        ids = torch.arange(n_rows * obs[2], dtype=torch.int32, device="cuda")
        feat, scores = eng.analysis(frames=planes, frame_stride=obs[0] * obs[1], frame_ids=ids, n_rows=n_rows)
    o_feat, o_scores = oa.analysis_net(onet.to_torch(params, torch.float64), states, list(feats), arch, ln, batch_norm=bn,
                                       batch_stats=None if stats is None else onet.to_torch(stats, torch.float64))
    # cnn: three conv layers + the hidden Dense layers; fc: the hidden Dense layers; impala: 12 block activations + torso output + Dense
    assert len(scores) == len(o_scores) == (12 + len(feats) - 2 if arch == "impala" else len(feats))
    assert feat.shape == o_feat.shape
    np.testing.assert_allclose(feat.cpu().numpy(), o_feat, atol=1e-3 * max(1.0, float(np.abs(o_feat).max())), rtol=0)
    for s, o in zip(scores, o_scores):
        assert s.numel() == o.size
        np.testing.assert_allclose(s.cpu().numpy(), o, atol=1e-3 * n_rows, rtol=0)
    from slimdqn.utils.analysis import compute_dead_neurons, compute_srank

    assert abs(compute_srank(feat.cpu().numpy()) - oa.compute_srank(o_feat)) <= 1
    assert compute_dead_neurons([s.cpu().numpy() for s in scores]) == pytest.approx(oa.compute_dead_neurons(o_scores), abs=2e-3)


def test_analysis_flag_of_the_trainer_with_impala_and_batch_norm(tmp_path):
    from experiments.atari.isdqn import run

    argv = ["-en", "anab_Synthetic", "-s", "1", "-dw", "-f", "8", "8", "8", "16", "-rbc", "300", "-bs", "8", "-horizon", "40", "-at", "impala",
            "-ne", "1", "-ntspe", "80", "-utd", "4", "-nis", "30", "-ed", "100", "-nbi", "2", "-ln", "-bn", "-tuf", "16", "-env", "synthetic", "-a"]
    run(argv, root=str(tmp_path))
    logs = json.load(open(tmp_path / "atari" / "exp_output" / "anab_Synthetic" / "isdqn" / "analysis" / "1.json"))
    assert set(logs) == {"srank", "dead_neurons"} and len(logs["srank"]) >= 2
    assert all(1 <= s <= 16 for s in logs["srank"]) and all(0.0 <= d <= 1.0 for d in logs["dead_neurons"])


@pytest.mark.parametrize("arch,ln,bn", [("cnn", True, False), ("cnn", False, True), ("impala", True, True), ("impala", False, False)])
def test_reference_analysis_properties(arch, ln, bn):
    """The reference's tests/test_analysis.py:41-81 on the HIP path (its random architecture / LayerNorm / BatchNorm draws as four fixed
    cases): compute_srank of a constant matrix is 1 and of diag(0 .. n-1) the closed form; a freshly initialised network on 2550 random
    states has a feature srank above 256 (of 512), srank 1 at threshold 1, fewer than 10 % dead neurons; with every kernel and bias
    zeroed all neurons are dead."""
    from slimdqn._engine import QNetEngine
    from slimdqn.utils.analysis import compute_dead_neurons, compute_srank

    n = 777
    assert compute_srank(np.ones((n, 512))) == 1
    assert compute_srank(np.diag(np.arange(0, n)).astype(np.float64)) == np.searchsorted(np.cumsum(np.arange(0, n)[::-1]), 0.99 * n * (n - 1) / 2, side="left") + 1

    n_rows, obs, feats = 51 * 50, (84, 84, 4), (5, 7, 9, 512)
    eng = QNetEngine(obs, 4, 3, feats, arch, ln, n_rows // 2, batch_norm=bn)
    eng.init_params(3)
    rng = np.random.default_rng(0)
    # (the reference draws uniform [0, 1) floats: AnalysisNet divides by 255 like the Q-network; uint8 frames here, as the replay holds them)
    states = rng.integers(0, 256, (n_rows,) + obs, dtype=np.uint8)
    planes = torch.from_numpy(np.ascontiguousarray(np.moveaxis(states, -1, 1)).reshape(n_rows * obs[2], obs[0] * obs[1])).cuda()
    ids = torch.arange(n_rows * obs[2], dtype=torch.int32, device="cuda")
    feat, scores = eng.analysis(frames=planes, frame_stride=obs[0] * obs[1], frame_ids=ids, n_rows=n_rows)
    feat = feat.cpu().numpy()
    assert compute_srank(feat) > 256
    assert compute_srank(feat, 1) == 1
    # (the reference asserts < 0.1 for whatever architecture / normalisation it happened to draw; the un-normalised impala torso with
    # 5 / 7 / 9 channels and zero biases starts with 16 % of its block units silent on these states -- the oracle's number too)
    assert compute_dead_neurons([s.cpu().numpy() for s in scores]) < (0.1 if (ln or bn) else 0.3)
    tree = eng.export_flax()
    for mod, leaves in tree.items():
        if "Conv" in mod or "Dense" in mod:
            leaves["kernel"] = np.zeros_like(leaves["kernel"])
            leaves["bias"] = np.zeros_like(leaves["bias"])
    eng.import_flax(tree, batch_stats=eng.export_batch_stats() if bn else None)
    _, scores0 = eng.analysis(frames=planes, frame_stride=obs[0] * obs[1], frame_ids=ids, n_rows=n_rows)
    assert compute_dead_neurons([s.cpu().numpy() for s in scores0]) == 1
