"""CPU-side check of the drop-in boundary: the C-ABI library builds, loads and exports every symbol
that include/isdqn_hip.h declares; host-only entry points answer without a GPU."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import importlib.util

    spec = importlib.util.spec_from_file_location("isdqn_build", os.path.join(ROOT, "is-dqn_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.build(verbose=False)
    from slimdqn import _hip

    return _hip.lib()


def test_every_declared_symbol_is_exported(lib):
    header = open(os.path.join(ROOT, "include", "isdqn_hip.h")).read()
    declared = set(re.findall(r"\b(isdqn_[a-z_0-9]+)\s*\(", header))
    declared -= {"isdqn_net_config", "isdqn_tensor_info", "isdqn_batch"}
    assert len(declared) >= 18
    from slimdqn import _hip

    assert declared == set(_hip.PUBLIC_SYMBOLS)
    for name in declared:
        assert getattr(lib, name) is not None


def test_host_only_entry_points(lib):
    from slimdqn import _hip

    assert b"gfx950" in lib.isdqn_version()
    depth, first, n = ctypes.c_int32(), ctypes.c_int64(), ctypes.c_int64()
    assert lib.isdqn_tree_layout(1_000_000, ctypes.byref(depth), ctypes.byref(first), ctypes.byref(n)) == 0
    assert (depth.value, first.value, n.value) == (21, 1_048_575, 2_097_151)
    assert lib.isdqn_tree_layout(0, ctypes.byref(depth), ctypes.byref(first), ctypes.byref(n)) == _hip.ERR_CAPACITY
    with pytest.raises(AssertionError):
        _hip.check(_hip.ERR_CAPACITY)
    with pytest.raises(ValueError):
        _hip.check(_hip.ERR_RANGE)
    cfg = _hip.NetConfig()
    cfg.arch, cfg.obs_h, cfg.obs_w, cfg.obs_c = _hip.ARCH_CNN, 84, 84, 4
    cfg.n_features = 4
    for i, f in enumerate((32, 64, 64, 512)):
        cfg.features[i] = f
    cfg.n_actions, cfg.n_heads, cfg.layer_norm, cfg.batch_size = 9, 10, 1, 256
    nparam, cnt = ctypes.c_int64(), ctypes.c_int32()
    assert lib.isdqn_net_param_layout(ctypes.byref(cfg), ctypes.byref(nparam), None, 0, ctypes.byref(cnt)) == 0
    # 4,090,938 reference parameters + 6 padding floats (head 90 -> 96 rows: 6*512 weights + 6 biases)
    assert nparam.value == 4_090_938 + 6 * 512 + 6
    assert cnt.value == 18
    ws = ctypes.c_int64()
    assert lib.isdqn_net_workspace_bytes(ctypes.byref(cfg), ctypes.byref(ws)) == 0
    assert 0 < ws.value < 2 << 30
    cfg.arch = 7
    assert lib.isdqn_net_workspace_bytes(ctypes.byref(cfg), ctypes.byref(ws)) == _hip.ERR_UNSUPPORTED


def test_fastdiv_matches_integer_division():
    """The device index math divides by runtime constants with a multiply-high; restated here."""
    def fastdiv(d):
        if d == 1:
            return lambda n: n
        lg = 0
        while (1 << lg) < d:
            lg += 1
        p = 31 + lg
        m = ((1 << p) + d - 1) // d
        assert m < (1 << 32)
        return lambda n: ((n * m) >> 32) >> (p - 32)

    import numpy as np

    rng = np.random.default_rng(0)
    for d in [1, 2, 3, 4, 5, 7, 8, 9, 11, 21, 32, 40, 64, 72, 100, 121, 441, 7744, 30976, 112896]:
        f = fastdiv(d)
        ns = np.concatenate([np.arange(0, 5000), rng.integers(0, 2**31 - 1, 20000), np.arange(2**31 - 2000, 2**31 - 1)])
        for n in ns.tolist():
            assert f(n) == n // d, (n, d)
