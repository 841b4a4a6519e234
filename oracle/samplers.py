"""ORACLE (test infrastructure) -- sampling distributions, CPU restatement.

Follows reference ``slimdqn/sample_collection/samplers.py``:
  * UniformSamplingDistribution ...... :13-49  (PCG64 ``default_rng(seed)``,
    key<->dense-index maps with swap-with-last removal, ``integers(len,size)``)
  * PrioritizedSamplingDistribution .. :52-116 (``priority**exponent`` with 0
    kept at 0, vector update, removal moves the last leaf's priority into the
    vacated slot, ``uniform(0, root, size)`` -> ``SumTree.query``)

The reference's latent ``root == 0`` bug (samplers.py:106-108 calls ``.keys`` on
an ndarray) is reproduced as the same AttributeError.

Pinned by the known answers of the reference's tests/test_samplers.py and
tests/test_replay_buffer.py (seed-0 key stream).
"""
from __future__ import annotations

import numpy as np

from oracle.sum_tree import SumTree


class UniformSamplingDistribution:
    def __init__(self, seed: int) -> None:
        self._rng_key = np.random.default_rng(seed)
        self._key_to_index = {}
        self._index_to_key = []

    def add(self, key) -> None:
        self._key_to_index[key] = len(self._index_to_key)
        self._index_to_key.append(key)

    def remove(self, key) -> None:
        assert key in self._key_to_index, ValueError(f"Key {key} not found.")
        hole = self._key_to_index.pop(key)
        last_key = self._index_to_key.pop()
        if last_key != key:
            self._index_to_key[hole] = last_key
            self._key_to_index[last_key] = hole

    def sample(self, size: int):
        assert self._index_to_key, ValueError("No keys to sample from.")
        indices = self._rng_key.integers(len(self._index_to_key), size=size)
        return np.asarray([self._index_to_key[i] for i in indices], dtype=np.int32)


class PrioritizedSamplingDistribution(UniformSamplingDistribution):
    def __init__(self, seed: int, max_capacity: int, priority_exponent: float = 1.0) -> None:
        self._max_capacity = max_capacity
        self._priority_exponent = priority_exponent
        self._sum_tree = SumTree(max_capacity)
        super().__init__(seed=seed)

    def _transform(self, priority):
        return 0.0 if priority == 0.0 else priority**self._priority_exponent

    def add(self, key, priority) -> None:
        super().add(key)
        if priority is None:
            priority = 0.0
        self._sum_tree.set(self._key_to_index[key], self._transform(priority))

    def update(self, keys, priorities) -> None:
        if not isinstance(keys, np.ndarray):
            keys = np.asarray([keys], dtype=np.int32)
        priorities = np.where(priorities == 0.0, 0.0, priorities**self._priority_exponent)
        idx = np.asarray([self._key_to_index[k] for k in keys.tolist()], dtype=np.int32)
        self._sum_tree.set(idx, priorities)

    def remove(self, key) -> None:
        index = self._key_to_index[key]
        last_index = len(self._index_to_key) - 1
        if index == last_index:
            self._sum_tree.set(index, 0.0)
        else:
            self._sum_tree.set(
                np.asarray([index, last_index], dtype=np.int32),
                np.asarray([self._sum_tree.get(last_index), 0.0]),
            )
        super().remove(key)

    def sample(self, size: int):
        if self._sum_tree.root == 0.0:
            # reference samplers.py:106-108: ndarray has no ``.keys`` -> AttributeError
            return super().sample(size).keys
        targets = self._rng_key.uniform(0.0, self._sum_tree.root, size=size)
        indices = self._sum_tree.query(targets)
        return np.asarray([self._index_to_key[i] for i in indices], dtype=np.int32)
