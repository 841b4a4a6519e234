"""GPU parity: HIP sum tree vs the golden vectors produced by the REFERENCE class, bit for bit,
through the C ABI (slimdqn.sample_collection.sum_tree.SumTree wraps isdqn_tree_*)."""
import hashlib

import numpy as np
import pytest

from tests.sumtree_cases import all_cases, replay

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", list(all_cases()), ids=lambda c: c[0])
def test_hip_tree_matches_reference_golden(case, golden_sum_tree):
    from slimdqn.sample_collection.sum_tree import SumTree

    name, capacity, ops = case
    g = golden_sum_tree
    tree = SumTree(capacity)
    results = replay(tree, ops)
    nodes = tree._nodes
    assert tree._depth == int(g[f"{name}/depth"])
    assert tree._first_leaf_offset == int(g[f"{name}/first_leaf_offset"])
    assert nodes.size == int(g[f"{name}/n_nodes"])
    if f"{name}/nodes" in g:
        np.testing.assert_array_equal(nodes, g[f"{name}/nodes"])
    else:
        np.testing.assert_array_equal(nodes[:1023], g[f"{name}/nodes_top"])
    assert hashlib.sha256(nodes.tobytes()).digest() == g[f"{name}/nodes_sha256"].tobytes()
    assert tree.root == float(g[f"{name}/root"])
    assert tree.max_recorded_priority == float(g[f"{name}/max_recorded_priority"])
    assert len(results) == int(g[f"{name}/n_queries"])
    for i, r in enumerate(results):
        np.testing.assert_array_equal(r, g[f"{name}/query{i}"])


def test_hip_tree_matches_oracle_on_fresh_random_ops():
    from oracle.sum_tree import SumTree as Oracle
    from slimdqn.sample_collection.sum_tree import SumTree

    rng = np.random.default_rng(99)
    for capacity in (37, 5000, 300_000):
        o, h = Oracle(capacity), SumTree(capacity)
        for _ in range(8):
            n = int(rng.integers(1, 700))
            idx = rng.integers(0, capacity, n).astype(np.int32)
            val = rng.uniform(0, 5, n)
            o.set(idx, val)
            h.set(idx, val)
            t = rng.uniform(0, o.root, 333)
            np.testing.assert_array_equal(o.query(t), h.query(t))
        np.testing.assert_array_equal(o._nodes, h._nodes)


def test_hip_tree_reference_error_conventions():
    from slimdqn.sample_collection.sum_tree import SumTree

    with pytest.raises(AssertionError):
        SumTree(capacity=-1)
    t = SumTree(100)
    with pytest.raises(AssertionError):
        t.set(0, -1)
    with pytest.raises(ValueError):
        t.query(1.0)  # empty tree
    t.set(5, 1.0)
    assert t.query(0.99) == 5
    with pytest.raises(ValueError):
        t.query(1.0)  # target == root
    # device-side latch: negative value leaves the tree untouched
    import torch

    before = t._nodes.copy()
    t.set_device(torch.tensor([1, 2], dtype=torch.int32, device="cuda"), torch.tensor([1.0, -2.0], dtype=torch.float64, device="cuda"))
    with pytest.raises(AssertionError):
        t.check_status()
    np.testing.assert_array_equal(before, t._nodes)


def test_unit_target_query_matches_numpy_uniform():
    """query(unit draws) == reference sampling: rng.uniform(0, root, n) -> query (samplers.py:110-111)."""
    import torch
    from oracle.sum_tree import SumTree as Oracle
    from slimdqn.sample_collection.sum_tree import SumTree

    rng = np.random.default_rng(5)
    cap = 100_000
    o, h = Oracle(cap), SumTree(cap)
    for s in range(0, cap, 4096):
        n = min(4096, cap - s)
        idx = np.arange(s, s + n, dtype=np.int32)
        val = rng.uniform(0.1, 2.0, n)
        o.set(idx, val)
        h.set(idx, val)
    ref = o.query(np.random.default_rng(0).uniform(0.0, o.root, size=256))
    u = torch.from_numpy(np.random.default_rng(0).random(256)).cuda()
    got = h.query_device(u, unit=True).cpu().numpy()
    h.check_status()
    np.testing.assert_array_equal(ref, got)
