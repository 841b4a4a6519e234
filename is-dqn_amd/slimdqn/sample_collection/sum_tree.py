"""Device-resident float64 sum tree: same surface as the reference's numpy SumTree
(slimdqn/sample_collection/sum_tree.py:8-102), arithmetic in HIP (csrc/tree_kernels.hip),
bit-exact with the reference for set/query.

Attributes the reference tests read (tests/test_sum_tree.py:34, 82-83) are kept: ``_depth``,
``_first_leaf_offset``, ``_nodes`` (a host copy, synchronising), ``max_recorded_priority``.

Two calling styles:
  * the reference's synchronous API (numpy in, numpy/scalars out, Python exceptions);
  * ``*_device`` methods taking/returning torch device tensors and never synchronising -- what
    the fused sampling -> update -> priority-writeback path uses.  Violations are latched in a
    device status word; ``check_status()`` raises them the way the reference would.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from slimdqn import _hip


class SumTree:
    def __init__(self, capacity: int, device: str | None = None) -> None:
        assert capacity > 0, "Capacity to sum tree must be positive."
        _hip.require_gpu()
        self._lib = _hip.lib()
        depth, first, n_nodes = ctypes.c_int32(), ctypes.c_int64(), ctypes.c_int64()
        _hip.check(self._lib.isdqn_tree_layout(int(capacity), ctypes.byref(depth), ctypes.byref(first), ctypes.byref(n_nodes)))
        self._capacity = capacity
        self._depth = int(depth.value)
        self._first_leaf_offset = int(first.value)
        self.device = _hip.resolve_device(device)
        _hip.bind_device(self.device)
        self._nodes_dev = torch.zeros(int(n_nodes.value), dtype=torch.float64, device=self.device)
        self._max_dev = torch.ones(1, dtype=torch.float64, device=self.device)
        self._status = torch.zeros(1, dtype=torch.int32, device=self.device)

    # ---------------------------------------------------------------- device API (no synchronisation)
    def set_device(self, indices: torch.Tensor, values: torch.Tensor) -> None:
        """indices int32 [n], values float64 [n] on the device, n <= 4096."""
        assert indices.shape == values.shape, "Indices and values must have the same shape."
        assert indices.dtype == torch.int32 and values.dtype == torch.float64
        n = int(indices.numel())
        _hip.check(
            self._lib.isdqn_tree_set(
                _hip.ptr(self._nodes_dev), self._depth, _hip.ptr(indices), _hip.ptr(values), n,
                _hip.ptr(self._max_dev), _hip.ptr(self._status), _hip.stream_ptr(self.device),
            ),
            "isdqn_tree_set",
        )

    def query_device(self, targets: torch.Tensor, out: torch.Tensor | None = None, unit: bool = False) -> torch.Tensor:
        """targets float64 [n] on the device (``unit``: draws in [0,1) scaled by the root on the device)."""
        assert targets.dtype == torch.float64
        n = int(targets.numel())
        if out is None:
            out = torch.empty(n, dtype=torch.int32, device=self.device)
        _hip.check(
            self._lib.isdqn_tree_query(
                _hip.ptr(self._nodes_dev), self._depth, _hip.ptr(targets), n, 1 if unit else 0, _hip.ptr(out),
                _hip.ptr(self._status), _hip.stream_ptr(self.device),
            ),
            "isdqn_tree_query",
        )
        return out

    def swap_remove_device(self, index: int, last_index: int) -> None:
        """samplers.py:89-103 on the device: leaf[index] <- leaf[last_index]; leaf[last_index] <- 0."""
        _hip.check(
            self._lib.isdqn_tree_swap_remove(
                _hip.ptr(self._nodes_dev), self._depth, int(index), int(last_index), _hip.ptr(self._status), _hip.stream_ptr(self.device)
            ),
            "isdqn_tree_swap_remove",
        )

    def check_status(self) -> None:
        """Raise what the reference would have raised for latched violations (synchronises)."""
        s = int(self._status.item())
        if s:
            self._status.zero_()
        if s & _hip.STATUS_NEGATIVE_VALUE:
            raise AssertionError("Values must be positive.")
        if s & (_hip.STATUS_TARGET_RANGE | _hip.STATUS_EMPTY_TREE):
            raise ValueError(f"Targets must be in the interval [0.0, {self.root}).")

    # ---------------------------------------------------------------- reference API (synchronous)
    def set(self, indices, values) -> None:
        if isinstance(indices, (int, np.integer)):
            indices = np.asarray([indices], np.int32)
        if isinstance(values, (int, float, np.floating)):
            values = np.asarray([values], np.float64)
        indices = np.asarray(indices)
        values = np.asarray(values)
        assert indices.shape == values.shape, "Indices and values must have the same shape."
        assert (values >= 0.0).all(), "Values must be positive."
        idx = indices.reshape(-1).astype(np.int32)
        val = values.reshape(-1).astype(np.float64)  # float32 -> float64 is exact (sum_tree.py:34 promotes too)
        if idx.size > _hip.TREE_MAX_BATCH:
            # The kernel takes <= 4096 pairs.  De-duplicate on the host exactly as np.unique does (first
            # occurrence per leaf, leaves ascending) and feed ascending chunks: every ancestor then receives
            # the same deltas in the same order as in the single reference call (bit-identical).
            uniq, first = np.unique(idx, return_index=True)
            idx, val_u = uniq.astype(np.int32), val[first]
            self._max_dev.clamp_(min=float(val.max()))
            val = val_u
        for s in range(0, idx.size, _hip.TREE_MAX_BATCH):
            di = torch.from_numpy(idx[s : s + _hip.TREE_MAX_BATCH]).to(self.device)
            dv = torch.from_numpy(val[s : s + _hip.TREE_MAX_BATCH]).to(self.device)
            self.set_device(di, dv)
        self.check_status()

    def get(self, index):
        if isinstance(index, (int, np.integer)):
            return float(self._nodes_dev[self._first_leaf_offset + int(index)].item())
        idx = torch.as_tensor(np.asarray(index), device=self.device).long() + self._first_leaf_offset
        return self._nodes_dev[idx].cpu().numpy()

    @property
    def root(self) -> float:
        return float(self._nodes_dev[0].item())

    @property
    def max_recorded_priority(self) -> float:
        return float(self._max_dev.item())

    @max_recorded_priority.setter
    def max_recorded_priority(self, v: float) -> None:
        self._max_dev.fill_(float(v))

    @property
    def _nodes(self) -> np.ndarray:
        return self._nodes_dev.cpu().numpy()

    def query(self, targets):
        scalar = isinstance(targets, (int, float))
        if scalar:
            targets = np.asarray([targets], np.float64)
        t = torch.from_numpy(np.asarray(targets).astype(np.float64).reshape(-1)).to(self.device)
        out = self.query_device(t)
        self.check_status()
        res = out.cpu().numpy().reshape(np.asarray(targets).shape)
        return res
