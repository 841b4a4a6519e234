"""tests/golden/network_K{1,9}.npz and replay.npz (fixture groups 2 and 3) against the oracle as it is NOW:
the generator's functions are run again and every array compared with the committed file, so a change of the restatement that moves
a number -- intended or not -- fails here first, on the CPU (the GPU twin, tests/test_gpu_golden.py, holds the HIP path to the same
files).  These fixtures do not pin the oracle to the reference (it has no numbers to pin to: oracle/make_golden_network.py)."""
import hashlib
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _compare(name, got, want):
    assert set(got) == set(want.files), f"{name}: arrays differ {set(got) ^ set(want.files)}"
    for k in want.files:
        a, b = np.asarray(got[k]), want[k]
        assert a.shape == b.shape and a.dtype == b.dtype, (name, k, a.shape, b.shape, a.dtype, b.dtype)
        if a.dtype.kind in "iub":
            np.testing.assert_array_equal(a, b, err_msg=f"{name}:{k}")
        elif k.startswith("f64/") and a.dtype == np.float64:
            np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-12, err_msg=f"{name}:{k}")
        else:  # float32 torch-CPU arithmetic: the same container gives the same bits; another BLAS / thread count a few ulps
            np.testing.assert_allclose(a, b, rtol=2e-5, atol=2e-6, err_msg=f"{name}:{k}")


@pytest.mark.parametrize("case", ["K9", "K1"])
def test_network_fixture_is_what_the_oracle_computes(case):
    from oracle.make_golden_network import network_case

    _compare(case, network_case(case), np.load(os.path.join(GOLD, f"network_{case}.npz")))


def test_initial_parameters_are_the_seeded_init():
    from oracle import network as net
    from oracle.make_golden_network import CASES, FEATS, OBS

    for case, c in CASES.items():
        want = np.load(os.path.join(GOLD, f"network_{case}.npz"))
        params = net.init_params(c["seed"], OBS, FEATS, "cnn", (1 + c["K"]) * c["A"], True)
        for m, leaves in params.items():
            for n, v in leaves.items():
                sha = hashlib.sha256(np.ascontiguousarray(v, np.float32).tobytes()).digest()
                assert sha == want[f"init_sha256/{m}/{n}"].tobytes(), (case, m, n)


def test_replay_fixture_is_what_the_oracle_computes_and_what_the_reference_tests_say():
    from oracle.make_golden_network import replay_cases

    want = np.load(os.path.join(GOLD, "replay.npz"))
    _compare("replay", replay_cases(), want)
    # the reference's own known answers (tests/test_replay_buffer.py:49-105, 107-133)
    np.testing.assert_array_equal(want["fifo/keys"], np.arange(5, 15))
    np.testing.assert_array_equal(want["nstep/rewards"], np.full(8, 10.0))
    np.testing.assert_array_equal(want["stack/first_state_fill"], [0, 0, 0, 1])
    # tests/test_samplers.py:10-35: a zero-priority key is never drawn, keys updated to zero disappear, a removed key too
    assert 3 not in want["prio/sample_a"] and not set(want["prio/sample_b"]) & {1, 2, 3} and not set(want["prio/sample_c"]) & {0, 1, 2, 3}
