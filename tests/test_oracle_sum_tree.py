"""Pins the oracle sum-tree: (1) golden vectors produced by the REFERENCE class
(oracle/make_golden.py), bit for bit; (2) the known answers of the reference's
tests/test_sum_tree.py restated one to one."""
import hashlib

import numpy as np
import pytest

from oracle.sum_tree import SumTree, tree_layout
from tests.sumtree_cases import all_cases, replay


@pytest.mark.parametrize("case", list(all_cases()), ids=lambda c: c[0])
def test_oracle_matches_reference_golden(case, golden_sum_tree):
    name, capacity, ops = case
    g = golden_sum_tree
    tree = SumTree(capacity)
    results = replay(tree, ops)
    assert tree._depth == int(g[f"{name}/depth"])
    assert tree._first_leaf_offset == int(g[f"{name}/first_leaf_offset"])
    assert tree._nodes.size == int(g[f"{name}/n_nodes"])
    assert hashlib.sha256(tree._nodes.tobytes()).digest() == g[f"{name}/nodes_sha256"].tobytes()
    assert float(tree.root) == float(g[f"{name}/root"])
    assert float(tree.max_recorded_priority) == float(g[f"{name}/max_recorded_priority"])
    assert len(results) == int(g[f"{name}/n_queries"])
    for i, r in enumerate(results):
        np.testing.assert_array_equal(r, g[f"{name}/query{i}"])


# ---- reference tests/test_sum_tree.py, restated -------------------------------------------
def test_negative_capacity_raises():  # :16-18
    with pytest.raises(AssertionError):
        SumTree(capacity=-1)


def test_negative_value_raises():  # :20-22
    with pytest.raises(AssertionError):
        SumTree(100).set(0, -1)


def test_set_small_capacity():  # :24-27
    t = SumTree(1)
    t.set(0, 1.5)
    assert t.root == 1.5


def test_set_and_get_value():  # :29-37
    t = SumTree(100)
    t.set(0, 1.0)
    assert t.get(0) == 1.0
    leaf = t._first_leaf_offset
    while leaf > 0:
        leaf //= 2
        assert t._nodes[leaf] == 1.0


def test_set_vectorized_and_duplicates():  # :39-55
    t = SumTree(100)
    t.set(np.array([1, 2], np.int32), np.array([3.0, 4.0], np.float32))
    assert (t.get(1), t.get(2), t.root) == (3.0, 4.0, 7.0)
    t = SumTree(100)
    t.set(np.array([1, 1, 1, 2, 2], np.int32), np.array([3.0, 3.0, 3.0, 4.0, 4.0], np.float32))
    assert (t.get(1), t.get(2), t.root) == (3.0, 4.0, 7.0)


def test_capacity_and_empty_query():  # :57-62
    t = SumTree(100)
    assert t._nodes.size >= 100
    with pytest.raises(ValueError):
        t.query(1.0)


def test_query_known_answers():  # :64-128
    t = SumTree(100)
    t.set(5, 1.0)
    assert t.query(0.99) == 5
    t = SumTree(4)
    t.set(np.arange(4, dtype=np.int32), np.array([0.5, 1.0, 0.5, 0.5], np.float32))
    assert (t.root, t._depth, t._nodes.size) == (2.5, 3, 7)
    np.testing.assert_array_equal(t.query(np.array([1.5, 1.0])), np.array([2, 1], np.int32))
    t.set(0, 0.25)
    assert t.root == 2.25
    assert (t.query(0.249), t.query(0.5), t.query(1.25)) == (0, 1, 2)
    t = SumTree(8)
    t.set(np.arange(8, dtype=np.int32), np.ones(8, np.float32))
    assert (t.root, t._depth, t._nodes.size) == (8.0, 4, 15)
    np.testing.assert_array_equal(t.query(np.arange(8, dtype=np.int32)), np.arange(8, dtype=np.int32))


def test_max_recorded_priority():  # :130-136
    t = SumTree(100)
    t.set(0, 0)
    assert t.max_recorded_priority == 1
    for i in range(1, 32):
        t.set(i, i)
        assert t.max_recorded_priority == i


def test_layout_1e6():
    assert tree_layout(1_000_000) == (21, 1_048_575, 2_097_151)


def test_uniform_targets_are_root_times_unit_draw():
    """numpy's Generator.uniform(0, root, n) is 0.0 + root*next_double: the device path
    relies on this to pre-draw root-independent unit doubles (samplers.py:110)."""
    for root in (1.0, 2.25, 1234.56789, 1e6 / 3):
        a = np.random.default_rng(7).uniform(0.0, root, size=1000)
        b = 0.0 + root * np.random.default_rng(7).random(1000)
        np.testing.assert_array_equal(a, b)
