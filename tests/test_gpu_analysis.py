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
