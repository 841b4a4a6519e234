"""Scripted and seeded sum-tree operation sequences shared by oracle/make_golden.py
(which replays them through the REFERENCE class to produce tests/golden/sum_tree.npz)
and by the parity tests (which replay them through the oracle and the HIP path).

An op is ("set", indices int32[n], values float64|float32[n]) or ("query", targets float64[n]).
Scalar calls of the reference tests are written as 1-element arrays (the reference
promotes scalars to 1-element arrays itself, sum_tree.py:26-29, :69-70).
"""
import numpy as np


def _i(*a):
    return np.asarray(a, dtype=np.int32)


def _f(*a):
    return np.asarray(a, dtype=np.float64)


def _f32(*a):
    return np.asarray(a, dtype=np.float32)


def scripted_cases():
    """The known-answer cases of the reference's tests/test_sum_tree.py (:24-136)."""
    yield "small_capacity", 1, [("set", _i(0), _f(1.5))]
    yield "set_get", 100, [("set", _i(0), _f(1.0))]
    yield "set_vectorized", 100, [("set", _i(1, 2), _f32(3.0, 4.0))]
    yield "set_duplicates", 100, [("set", _i(1, 1, 1, 2, 2), _f32(3.0, 3.0, 3.0, 4.0, 4.0))]
    yield "query_value", 100, [("set", _i(5), _f(1.0)), ("query", _f(0.99))]
    yield "query_vectorized", 4, [
        ("set", _i(0, 1, 2, 3), _f32(0.5, 1.0, 0.5, 0.5)),
        ("query", _f(1.5, 1.0)),
    ]
    yield "update_sum", 4, [
        ("set", _i(0, 1, 2, 3), _f32(0.5, 1.0, 0.5, 0.5)),
        ("set", _i(0), _f(0.25)),
        ("query", _f(0.249)),
        ("query", _f(0.5)),
        ("query", _f(1.25)),
    ]
    yield "large_tree", 8, [
        ("set", np.arange(8, dtype=np.int32), np.ones(8, dtype=np.float32)),
        ("query", np.arange(8, dtype=np.float64)),
    ]
    ops = [("set", _i(0), _f(0.0))]
    for i in range(1, 32):
        ops.append(("set", _i(i), _f(float(i))))
    yield "max_recorded", 100, ops


def seeded_cases():
    """Random fills / duplicate-heavy updates / swap-remove patterns / queries."""
    for capacity, batch, rounds in [
        (1, 1, 3),
        (4, 8, 6),
        (8, 16, 6),
        (100, 64, 10),
        (1000, 256, 10),
        (65536, 256, 12),
        (1_000_000, 256, 12),
        (1_000_000, 1024, 6),
    ]:
        rng = np.random.default_rng(1234 + capacity + batch)
        ops = []
        # dense initial fill in chunks (exercises long ascending runs sharing ancestors)
        fill = min(capacity, 20000)
        start = 0
        while start < fill:
            n = min(4096, fill - start)
            ops.append(("set", np.arange(start, start + n, dtype=np.int32), rng.uniform(0.1, 2.0, n)))
            start += n
        for r in range(rounds):
            idx = rng.integers(0, min(capacity, fill), size=batch).astype(np.int32)
            if r % 3 == 1:  # force many duplicates
                idx[: batch // 2] = idx[0]
            vals = rng.uniform(0.0, 3.0, size=batch)
            if r % 4 == 2:
                vals[rng.integers(0, batch, size=max(1, batch // 8))] = 0.0
            if r % 2 == 0:
                vals = vals.astype(np.float32)
            ops.append(("set", idx, vals))
            # the sampler's swap-remove pattern (samplers.py:98-101): two leaves, one zeroed
            a, b = rng.integers(0, min(capacity, fill), size=2)
            if a != b:
                ops.append(("swap_remove", _i(a, b), None))
            ops.append(("query_u", rng.random(batch), None))  # targets = 0.0 + root * u
        yield f"seeded_c{capacity}_b{batch}", capacity, ops


def all_cases():
    yield from scripted_cases()
    yield from seeded_cases()


def replay(tree, ops):
    """Run ops through any object with the reference SumTree surface; return query results."""
    results = []
    for op in ops:
        kind = op[0]
        if kind == "set":
            tree.set(op[1], op[2])
        elif kind == "swap_remove":
            a, b = int(op[1][0]), int(op[1][1])
            tree.set(np.asarray([a, b], dtype=np.int32), np.asarray([tree.get(b), 0.0]))
        elif kind == "query":
            results.append(np.asarray(tree.query(op[1])).astype(np.int32).reshape(-1))
        elif kind == "query_u":
            root = float(tree.root)
            if root > 0.0:
                results.append(np.asarray(tree.query(0.0 + root * op[1])).astype(np.int32).reshape(-1))
            else:
                results.append(np.zeros(0, np.int32))
        else:
            raise ValueError(kind)
    return results
