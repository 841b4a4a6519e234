"""Sampling distributions with the reference's surface (slimdqn/sample_collection/samplers.py).

Host side (python, like the reference): the PCG64 generator and the key <-> dense-index maps with
swap-with-last removal (samplers.py:13-49).  Device side: the float64 sum tree of the prioritized
distribution (csrc/tree_kernels.hip) and an index -> element-slot table so that a sampled batch
never leaves the GPU:

    sample_device(size)  ->  int32 device tensor of dense indices   (no synchronisation)
    sample(size)         ->  int32 numpy keys, exactly the reference's return value

Both consume the generator identically (``integers(len, size)`` / ``uniform(0, root, size)`` which is
``root * random(size)`` bit for bit), so a run can mix them freely and stay on the reference's stream.
"""
from __future__ import annotations

import numpy as np
import torch

from slimdqn import _hip
from slimdqn.sample_collection import sum_tree


class _Uploader:
    """Small host -> device uploads (the drawn indices / unit draws of one batch, ~1-2 KB) through a ring of
    pinned buffers on a side stream, so that the copy engine moves them while the previous steps still compute;
    the consuming stream only waits on an event.  (A pageable copy costs a ~25 us blit kernel in the step.)"""

    SLOTS = 4        # pinning host memory costs ~30 ms per buffer: a few slots, allocated once, reused forever
    SLOT_BYTES = 1 << 18

    def __init__(self, device):
        self.device = device
        self.stream = torch.cuda.Stream(device)
        self.pinned = [None] * self.SLOTS
        self.events = [None] * self.SLOTS
        self.count = 0

    def upload(self, host: np.ndarray) -> torch.Tensor:
        host = np.ascontiguousarray(host)
        k = self.count % self.SLOTS
        self.count += 1
        if self.events[k] is not None:
            self.events[k].synchronize()  # the slot's previous copy has long finished; never races the host write
        n = host.nbytes
        if self.pinned[k] is None or self.pinned[k].numel() < n:
            self.pinned[k] = torch.empty(max(n, self.SLOT_BYTES), dtype=torch.uint8).pin_memory()
        src = torch.from_numpy(host)
        view = self.pinned[k][:n].view(src.dtype).view(src.shape)
        view.copy_(src)
        consumer = torch.cuda.current_stream(self.device)
        with torch.cuda.stream(self.stream):
            dev = view.to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self.events[k] = ev
        consumer.wait_event(ev)
        dev.record_stream(consumer)
        return dev


class UniformSamplingDistribution:
    """samplers.py:13-49."""

    def __init__(self, seed: int, device: str | None = None) -> None:
        self._rng_key = np.random.default_rng(seed)
        self._key_to_index = {}
        self._index_to_key = []
        self.device = _hip.resolve_device(device)
        self._uploader = None
        self._pf = None        # prefetched draws: dict(host, dev, next, n, size, length, state)
        self._last_len = -1

    # -- bookkeeping -----------------------------------------------------------------------------
    def add(self, key) -> None:
        self._key_to_index[key] = len(self._index_to_key)
        self._index_to_key.append(key)

    def add_bulk(self, keys, priorities=None) -> None:
        """Vectorised ``add`` for consecutive fresh keys (synthetic prefill); same resulting maps."""
        start = len(self._index_to_key)
        keys = [int(k) for k in keys]
        self._index_to_key.extend(keys)
        self._key_to_index.update({k: start + i for i, k in enumerate(keys)})

    def remove(self, key) -> None:
        assert key in self._key_to_index, ValueError(f"Key {key} not found.")
        hole = self._key_to_index.pop(key)
        last_key = self._index_to_key.pop()
        if last_key != key:
            self._index_to_key[hole] = last_key
            self._key_to_index[last_key] = hole

    def __len__(self) -> int:
        return len(self._index_to_key)

    # -- sampling ----------------------------------------------------------------------------------
    def _draw_indices(self, size: int) -> np.ndarray:
        assert self._index_to_key, ValueError("No keys to sample from.")
        return self._rng_key.integers(len(self._index_to_key), size=size)

    # Draws are produced PREFETCH batches at a time once the number of keys has stopped changing (a full
    # FIFO buffer): n consecutive ``integers(len, size)`` calls are exactly what n reference ``sample`` calls
    # consume, so one upload serves n steps.  If the length changes while pre-drawn batches are unused, the
    # generator is rewound to the state before the block and advanced by the batches actually consumed --
    # the stream position is always the reference's.
    PREFETCH = 64

    def _next_row(self, size: int):
        length = len(self._index_to_key)
        assert length, ValueError("No keys to sample from.")
        pf = self._pf
        if pf is not None and (pf["size"] != size or pf["length"] != length or pf["next"] >= pf["n"]):
            if pf["next"] < pf["n"]:
                self._rng_key.bit_generator.state = pf["state"]
                for _ in range(pf["next"]):
                    self._rng_key.integers(pf["length"], size=pf["size"])
            pf = self._pf = None
        if pf is None:
            n = self.PREFETCH if self._last_len == length else 1
            self._last_len = length
            state = self._rng_key.bit_generator.state
            host = np.stack([self._rng_key.integers(length, size=size) for _ in range(n)]).astype(np.int32)
            pf = self._pf = dict(host=host, dev=None, next=0, n=n, size=size, length=length, state=state)
        k = pf["next"]
        pf["next"] = k + 1
        return pf, k

    def sample(self, size: int):
        pf, k = self._next_row(size)
        i2k = self._index_to_key
        return np.fromiter((i2k[i] for i in pf["host"][k]), dtype=np.int32, count=size)

    def _to_device(self, host: np.ndarray, dtype) -> torch.Tensor:
        if self.device.type != "cuda":
            return torch.from_numpy(np.ascontiguousarray(host)).to(dtype)
        if self._uploader is None:
            self._uploader = _Uploader(self.device)
        t = self._uploader.upload(host)
        return t if t.dtype == dtype else t.to(dtype)

    def sample_device(self, size: int) -> torch.Tensor:
        """Dense indices on the device (int32); the draw is the reference's ``integers(len, size)``."""
        pf, k = self._next_row(size)
        if pf["dev"] is None:
            pf["dev"] = self._to_device(pf["host"], torch.int32)
        return pf["dev"][k]

    def draw_rows_device(self, n: int, size: int) -> torch.Tensor:
        """The next ``n`` batches of dense indices as one [n, size] device tensor (graph-replayed steps);
        exactly the draws n ``sample_device(size)`` calls would make."""
        rows = [self.sample_device(size) for _ in range(n)]
        first = rows[0]
        # rows of one prefetched block are consecutive views of the same tensor: hand out the slab without a copy
        base = first.untyped_storage().data_ptr()
        if all(r.untyped_storage().data_ptr() == base and r.storage_offset() == first.storage_offset() + i * size
               for i, r in enumerate(rows)):
            return torch.as_strided(first, (n, size), (size, 1), first.storage_offset())
        return torch.stack(rows)

    def keys_of(self, indices: np.ndarray) -> np.ndarray:
        i2k = self._index_to_key
        return np.fromiter((i2k[i] for i in indices), dtype=np.int32, count=len(indices))


class PrioritizedSamplingDistribution(UniformSamplingDistribution):
    """samplers.py:52-116 with the sum tree resident on the GPU."""

    def __init__(self, seed: int, max_capacity: int, priority_exponent: float = 1.0, device: str | None = None) -> None:
        self._max_capacity = max_capacity
        self._priority_exponent = priority_exponent
        # The buffer briefly holds max_capacity + 1 keys (add, then evict: replay_buffer.py:190-196), so leaf
        # `max_capacity` is written.  The reference's tree has that leaf unless max_capacity is a power of two
        # (then sum_tree.py:33 raises IndexError at the first eviction).  One spare leaf in that case only: the
        # extra level has a zero right subtree, so every query descends exactly as in the smaller tree.
        leaves = self._max_capacity + 1 if (self._max_capacity & (self._max_capacity - 1)) == 0 else self._max_capacity
        self._tree = sum_tree.SumTree(leaves, device=device)
        super().__init__(seed=seed, device=self._tree.device)
        # leaf writes of add() are staged on the host and reach the tree as ONE set per flush (next sample / remove /
        # update, or MAX_PENDING adds): an env step then costs no kernel launch and no host<->device traffic here
        self._pend_idx, self._pend_val = [], []

    MAX_PENDING = 1024
    MAX_PRIORITY = "max"  # add(key, MAX_PRIORITY): the tree's max_recorded_priority at the time the add reaches the tree

    @property
    def _sum_tree(self):
        """The device tree with every staged add applied (what the reference's attribute of this name holds)."""
        self.flush()
        return self._tree

    def flush(self) -> None:
        if not self._pend_idx:
            return
        idx = np.asarray(self._pend_idx, dtype=np.int32)
        val = np.asarray(self._pend_val, dtype=np.float64)
        self._pend_idx, self._pend_val = [], []
        d_idx, d_val = self._to_device(idx, torch.int32), self._to_device(val, torch.float64)
        if np.isnan(val).any():  # MAX_PRIORITY entries: resolved on the device, no read-back
            d_val = torch.where(torch.isnan(d_val), self._tree._max_dev, d_val)
        self._tree.set_device(d_idx, d_val)

    def _transform(self, priority):
        return 0.0 if priority == 0.0 else priority**self._priority_exponent

    def add(self, key, priority) -> None:
        super().add(key)
        if isinstance(priority, str) and priority == self.MAX_PRIORITY:
            value = float("nan")  # (the recorded maximum is already in the leaves' transformed domain)
        else:
            if priority is None:
                priority = 0.0
            value = float(self._transform(priority))
            assert value >= 0.0, "Values must be positive."
        # Staged: consecutive adds append at consecutive dense indices (distinct leaves), so one set over the staged
        # block leaves every node with the bits of the one-leaf sets in the same order (sum_tree.py:20-47).
        self._pend_idx.append(self._key_to_index[key])
        self._pend_val.append(value)
        if len(self._pend_idx) >= self.MAX_PENDING:
            self.flush()

    def add_bulk(self, keys, priorities=None) -> None:
        start = len(self._index_to_key)
        super().add_bulk(keys)
        pr = np.zeros(len(keys)) if priorities is None else np.asarray(priorities, np.float64)
        pr = np.where(pr == 0.0, 0.0, pr**self._priority_exponent)
        self.flush()
        self._tree.set(np.arange(start, start + len(keys), dtype=np.int32), pr)

    def update(self, keys, priorities) -> None:
        if not isinstance(keys, np.ndarray):
            keys = np.asarray([keys], dtype=np.int32)
        priorities = np.where(priorities == 0.0, 0.0, priorities**self._priority_exponent)
        k2i = self._key_to_index
        self.flush()
        self._tree.set(np.fromiter((k2i[k] for k in keys.tolist()), dtype=np.int32), priorities)

    def update_device(self, indices: torch.Tensor, priorities: torch.Tensor) -> None:
        """TD-error writeback without leaving the GPU: ``indices`` are the dense indices returned by
        ``sample_device`` (the leaves of the tree), ``priorities`` float64.  exponent 1.0 keeps the
        values bit-identical to what ``update`` would write; other exponents are applied here with
        torch's float64 pow (not part of the bit-exact contract, like the reference's own libm pow)."""
        if self._priority_exponent != 1.0:
            priorities = torch.where(priorities == 0.0, priorities, priorities**self._priority_exponent)
        self.flush()
        self._tree.set_device(indices, priorities)

    def remove(self, key) -> None:
        index = self._key_to_index[key]
        last_index = len(self._index_to_key) - 1
        self.flush()
        self._tree.swap_remove_device(index, last_index)  # samplers.py:92-102 on the device
        super().remove(key)

    def sample(self, size: int):
        if self._sum_tree.root == 0.0:
            # reference samplers.py:106-108 calls ``.keys`` on an ndarray here -> AttributeError (kept)
            return super().sample(size).keys
        indices = self.sample_device(size).cpu().numpy()
        self._sum_tree.check_status()
        return self.keys_of(indices)

    def sample_device(self, size: int) -> torch.Tensor:
        # uniform(0, root, size) == 0.0 + root * random(size): the unit draws depend on neither the root nor the
        # number of keys, so PREFETCH batches are drawn and uploaded at once; the root is applied on the device
        pu = getattr(self, "_pu", None)
        if pu is None or pu["size"] != size or pu["next"] >= self.PREFETCH:
            if pu is not None and pu["next"] < self.PREFETCH:  # batch size changed mid-block: rewind, replay consumed
                self._rng_key.bit_generator.state = pu["state"]
                for _ in range(pu["next"]):
                    self._rng_key.random(pu["size"])
            state = self._rng_key.bit_generator.state
            host = np.stack([self._rng_key.random(size) for _ in range(self.PREFETCH)])
            pu = self._pu = dict(dev=self._to_device(host, torch.float64), next=0, size=size, state=state)
        k = pu["next"]
        pu["next"] = k + 1
        self.flush()
        return self._tree.query_device(pu["dev"][k], unit=True)

    def _next_units(self, size: int) -> torch.Tensor:
        pu = getattr(self, "_pu", None)
        if pu is None or pu["size"] != size or pu["next"] >= self.PREFETCH:
            if pu is not None and pu["next"] < self.PREFETCH:
                self._rng_key.bit_generator.state = pu["state"]
                for _ in range(pu["next"]):
                    self._rng_key.random(pu["size"])
            state = self._rng_key.bit_generator.state
            host = np.stack([self._rng_key.random(size) for _ in range(self.PREFETCH)])
            pu = self._pu = dict(dev=self._to_device(host, torch.float64), next=0, size=size, state=state)
        k = pu["next"]
        pu["next"] = k + 1
        return pu["dev"][k]

    def draw_rows_device(self, n: int, size: int) -> torch.Tensor:
        """The next ``n`` batches of UNIT draws ([n, size] float64): the graph applies the root on the device."""
        rows = [self._next_units(size) for _ in range(n)]
        first = rows[0]
        base = first.untyped_storage().data_ptr()
        if all(r.untyped_storage().data_ptr() == base and r.storage_offset() == first.storage_offset() + i * size
               for i, r in enumerate(rows)):
            return torch.as_strided(first, (n, size), (size, 1), first.storage_offset())
        return torch.stack(rows)
