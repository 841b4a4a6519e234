"""Engine plumbing shared by the agents (iSDQN, DQN, TFDQN): one QNetEngine per batch size, parameter handles, batch
conversion, single-observation forward.  No arithmetic of the hot path lives here."""
from __future__ import annotations

import numpy as np
import torch

from slimdqn import _hip
from slimdqn._engine import QNetEngine


class DeviceParams:
    """Handle on the flat device parameter buffer; ``handle["params"]`` gives the Flax-layout pytree (a copy)."""

    def __init__(self, engine: QNetEngine, tensor: torch.Tensor):
        self._engine = engine
        self.tensor = tensor

    @staticmethod
    def _nest(flat):
        out = {}
        for mod, leaves in flat.items():
            node = out
            for part in mod.split("/"):
                node = node.setdefault(part, {})
            node.update(leaves)
        return out

    def to_flax(self):
        """The reference's ``params["params"]`` pytree; nested modules (the impala Stacks) as nested dicts like Flax's."""
        return self._nest(self._engine.export_flax(self.tensor))

    def batch_stats(self):
        """Flax's second collection of a BatchNorm network (isdqn.py:87-88): {"BatchNorm_i": {"mean", "var"}} (the impala Stacks'
        own modules nested: {"Stack_0": {"BatchNorm_1": ...}})."""
        return self._nest(self._engine.export_batch_stats(self.tensor))

    def __getitem__(self, key):
        if key == "batch_stats" and self._engine.batch_norm:
            return self.batch_stats()
        if key != "params":
            raise KeyError(key)
        return self.to_flax()

    def clone(self) -> "DeviceParams":
        return DeviceParams(self._engine, self.tensor.clone())

    copy = clone  # the reference's ``self.params.copy()`` (dqn.py:34, 50)


def _is_one_frame_shift(old: np.ndarray, new: np.ndarray) -> bool:
    """new[..., :-1] == old[..., 1:] for (h, w, stack) uint8 stacks.  stack == 4: one pixel's frames are one little-endian
    uint32, so the test is a shift and a mask over contiguous words (a few us; the strided comparison costs ~70 us)."""
    if old.shape[-1] == 4 and old.flags.c_contiguous and new.flags.c_contiguous:
        o, n = old.reshape(-1).view("<u4"), new.reshape(-1).view("<u4")
        return bool(np.array_equal(n & np.uint32(0x00FFFFFF), o >> np.uint32(8)))
    return bool(np.array_equal(new[..., :-1], old[..., 1:]))


class EngineAgent:
    """Common state: ``n_heads`` network heads of ``n_actions`` outputs each on the HIP engine."""

    def _init_engine_agent(self, key, observation_dim, n_actions, n_heads, features, layer_norm, architecture_type,
                           learning_rate, gamma, update_horizon, adam_eps, batch_size, precision, device, huber_delta=0.0, batch_norm=False):
        self.n_actions = n_actions
        self.batch_norm = bool(batch_norm)
        self._n_heads = int(n_heads)
        self.features = [int(f) for f in features]
        self.architecture_type = architecture_type
        self.layer_norm = bool(layer_norm)
        self.observation_dim = tuple(int(d) for d in np.atleast_1d(observation_dim))
        self.learning_rate = learning_rate
        self.adam_eps = adam_eps
        self.gamma = gamma
        self.update_horizon = update_horizon
        self.precision = precision
        self.huber_delta = float(huber_delta)
        self.device = device
        self._seed = int(key) if not isinstance(key, torch.Generator) else int(key.initial_seed())
        self._engine = None
        self._graphed = None
        self._trust_mirror = None  # None: the engine's default (_engine.py: rebuild in every call); see the property below
        self._ring = None  # device copy of the acting path's frame stack (see _obs_to_device)
        self._make_engine(batch_size, init=True)

    # ------------------------------------------------------------------ engine management
    def _make_engine(self, batch_size: int, init: bool = False) -> None:
        old = self._engine
        eng = QNetEngine(
            self.observation_dim, self.n_actions, self._n_heads, self.features, self.architecture_type,
            self.layer_norm, batch_size, gamma_n=self.gamma**self.update_horizon, learning_rate=self.learning_rate,
            adam_eps=self.adam_eps, precision=self.precision, device=self.device, huber_delta=self.huber_delta, batch_norm=self.batch_norm,
        )
        if init:
            eng.init_params(self._seed)
        else:  # a batch of another size arrived: same parameter layout, new workspace
            eng.params.copy_(old.params)
            eng.adam_m.copy_(old.adam_m)
            eng.adam_v.copy_(old.adam_v)
            eng.adam_count.copy_(old.adam_count)
            eng.losses_accum.copy_(old.losses_accum)
        self._drop_graph()  # captured against the old engine's buffers
        self._engine = eng
        if self._trust_mirror is not None:
            eng.trust_mirror = self._trust_mirror
        self._ring = None
        self.params = DeviceParams(eng, eng.params)
        self.optimizer_state = {"count": eng.adam_count, "mu": eng.adam_m, "nu": eng.adam_v}
        self._engine_changed(old)

    @property
    def trust_mirror(self) -> bool:
        """Whether the engine may skip the weight-mirror rebuild while its bookkeeping says nothing wrote the parameters (an owner's
        declaration that every write goes through torch or the agent: _engine.py).  Setting it re-captures the update graph."""
        return bool(self._engine.trust_mirror)

    @trust_mirror.setter
    def trust_mirror(self, value: bool) -> None:
        value = bool(value)
        if value != bool(self._engine.trust_mirror):
            self._drop_graph()  # (the captured replay has the rebuild node or not)
            self._ring = None   # (and so have the acting graphs)
        self._trust_mirror = value
        self._engine.trust_mirror = value

    def _engine_changed(self, old) -> None:
        """Hook: a new engine replaced ``old`` (subclasses re-home extra device state)."""

    def _drop_graph(self) -> None:
        """Destroy the live captured update, if any (one executable graph per agent: _graph.py, "hipGraphLaunch fault")."""
        g, self._graphed = self._graphed, None
        if g is not None:
            g.destroy()

    def _graphed_update(self, replay_buffer, learn=None, key=None, steps: int = 1):
        """The captured sample -> learn -> [write-back] update of ``steps`` consecutive steps for this (replay, engine) pair, or None
        when the replay is not the device replay of this GPU (reference-layout buffers take the eager path).  ``learn`` / ``key``:
        another learn call than the engine's learn_on_batch and what it is bound to (DQN: the target parameters' buffer).  The agent
        owns ONE executable graph at a time: a request that does not match the live one destroys it first, then captures."""
        if not getattr(self, "use_graph", True) or not hasattr(replay_buffer, "_d_elem_frames") or getattr(replay_buffer, "_lib", None) is None:
            return None
        if replay_buffer.add_count == 0:
            return None
        if self.architecture_type == "fc" and not (replay_buffer._obs_float and replay_buffer._stack_size == 1):
            return None  # (the captured fc step reads float32 vector observations with stack_size 1: _graph.py)
        eng = self._engine_for(replay_buffer._batch_size)
        prioritized = hasattr(replay_buffer._sampling_distribution, "_tree")
        writeback = bool(getattr(self, "priority_writeback", False))
        g = self._graphed
        if (g is None or g.rb is not replay_buffer or g.eng is not eng or g.writeback != (writeback and prioritized)
                or getattr(g, "key", None) != key or g.S != steps):
            from slimdqn._graph import GraphedUpdate

            self._drop_graph()
            g = self._graphed = GraphedUpdate(replay_buffer, eng, prioritized, steps_per_graph=steps, writeback=writeback, learn=learn)
            g.key = key
            self._captures = getattr(self, "_captures", 0) + 1
        return g

    def _engine_for(self, batch_size: int) -> QNetEngine:
        if self._engine.batch_size != batch_size:
            self._make_engine(batch_size)
        return self._engine

    def _bind(self, params):
        """Accept the agent's own handle, another DeviceParams, or a Flax-layout pytree."""
        if params is None or params is self.params:
            return None
        if isinstance(params, DeviceParams):
            return params.tensor
        def flatten(tree):  # Flax nesting -> the engine's flattened "Stack_0/Conv_1" keys
            if tree is None or not any(k.startswith("Stack_") and "/" not in k for k in tree):
                return tree
            return {**{f"{k}/{m}": v for k, sub in tree.items() if k.startswith("Stack_") for m, v in sub.items()},
                    **{k: v for k, v in tree.items() if not k.startswith("Stack_")}}

        # a model pickle `{"params": variables}` (get_model), the variables dict `{"params": tree[, "batch_stats": tree]}`
        # (the reference's agent.params), or the bare tree
        while isinstance(params.get("params"), dict) and isinstance(params["params"].get("params"), dict):
            params = params["params"]
        tree = flatten(params["params"] if "params" in params else params)
        t = torch.empty_like(self._engine.params)
        self._engine.import_flax(tree, target=t, batch_stats=flatten(params.get("batch_stats")) if "params" in params else None)
        return t

    # ------------------------------------------------------------------ batches
    def _c_batch(self, eng: QNetEngine, samples):
        if hasattr(samples, "frame_ids") and self.architecture_type == "fc":
            # device replay of vector observations (LunarLander): the stored float32 bytes come back as (B, d) rows
            return eng.make_batch(state=samples.state, next_state=samples.next_state, action=samples.action, reward=samples.reward,
                                  terminal=samples.is_terminal)
        if hasattr(samples, "frame_ids"):  # DeviceBatch from the device replay
            return eng.make_batch(
                frames=samples.frames, frame_stride=samples.frame_stride, frame_ids=samples.frame_ids,
                action=samples.action, reward=samples.reward, terminal=samples.is_terminal,
            )
        dev = eng.device
        as_t = lambda x, dt: torch.as_tensor(np.asarray(x) if not torch.is_tensor(x) else x).to(dev).to(dt).contiguous()
        action = as_t(samples.action, torch.int32)
        reward = as_t(samples.reward, torch.float32)
        terminal = as_t(samples.is_terminal, torch.uint8)
        if self.architecture_type == "fc":
            st = as_t(samples.state, torch.float32).reshape(len(action), -1)
            nx = as_t(samples.next_state, torch.float32).reshape(len(action), -1)
            return eng.make_batch(state=st, next_state=nx, action=action, reward=reward, terminal=terminal)
        st = as_t(samples.state, torch.uint8)
        nx = as_t(samples.next_state, torch.uint8)
        B, h, w, stack = st.shape
        planes = torch.empty(2 * B * stack, h * w, dtype=torch.uint8, device=dev)
        ids = torch.empty(B, 2 * stack, dtype=torch.int32, device=dev)
        _hip.check(
            eng.lib.isdqn_replay_deinterleave(_hip.ptr(st), _hip.ptr(nx), h, w, stack, B, _hip.ptr(planes), _hip.ptr(ids), _hip.stream_ptr(dev))
        )
        return eng.make_batch(frames=planes, frame_stride=h * w, frame_ids=ids, action=action, reward=reward, terminal=terminal)

    @staticmethod
    def _batch_len(samples) -> int:
        return int(samples.action.shape[0])

    def _obs_to_device(self, state):
        eng = self._engine
        if self.architecture_type == "fc":
            obs = torch.as_tensor(np.asarray(state, dtype=np.float32)).reshape(1, -1).to(eng.device)
            return dict(obs=obs)
        s = np.asarray(state)
        s = s.astype(np.uint8) if s.dtype != np.uint8 else s.copy()
        h, w, stack = s.shape
        ring = self._ring
        if ring is None or ring["last"].shape != s.shape:
            rot = (torch.arange(stack)[None, :] + torch.arange(stack)[:, None]) % stack
            ring = self._ring = dict(
                planes=torch.empty(stack, h * w, dtype=torch.uint8, device=eng.device),
                rot=rot.to(torch.int32).to(eng.device),  # row r: plane slots oldest..newest after r one-frame shifts
                shifts=0,
                last=None,
            )
        # Consecutive acting states differ by one frame (atari.py:71-78 rolls the stack): the device keeps the stack as a
        # ring of planes and only the newest 7 KB frame crosses PCIe; anything else (reset, a foreign state) is a full upload.
        if ring["last"] is not None and _is_one_frame_shift(ring["last"], s):
            slot = ring["shifts"] % stack  # the oldest plane's slot
            ring["planes"][slot].copy_(torch.from_numpy(np.ascontiguousarray(s[..., -1]).reshape(-1)))
            ring["shifts"] += 1
        else:
            ring["planes"].copy_(torch.from_numpy(np.ascontiguousarray(np.moveaxis(s, -1, 0)).reshape(stack, h * w)))
            ring["shifts"] = 0
        ring["last"] = s
        return dict(frames=ring["planes"], frame_stride=h * w, frame_ids=ring["rot"][ring["shifts"] % stack])

    def _states_to_device(self, states):
        """n observations (n, h, w, stack) or (n, d) -> planes / rows for one batched forward."""
        eng = self._engine
        if self.architecture_type == "fc":
            s = np.asarray(states, dtype=np.float32)
            return dict(obs=torch.from_numpy(s.reshape(s.shape[0], -1)).to(eng.device))
        s = np.asarray(states)
        s = s.astype(np.uint8) if s.dtype != np.uint8 else s
        n, h, w, stack = s.shape
        planes = torch.from_numpy(np.ascontiguousarray(np.moveaxis(s, -1, 1)).reshape(n * stack, h * w)).to(eng.device)
        ids = torch.arange(n * stack, dtype=torch.int32, device=eng.device)
        return dict(frames=planes, frame_stride=h * w, frame_ids=ids)

    # ------------------------------------------------------------------ one environment step as one graph launch
    def _act_graph_step(self, s: np.ndarray, idx_network: int):
        """The common acting step -- the new state is the previous one shifted by a frame -- as ONE hipGraph launch: newest
        frame and head index up (one pinned 7 KB copy), frame into the ring slot of the oldest plane, the forward with the
        plane ids of this rotation, the action back into pinned memory.  One graph per (ring slot, weight mirror trusted or
        rebuilt), captured at first use; eager launches cost ~45 us of launch calls for the nine kernels of this forward."""
        eng, ring = self._engine, self._ring
        h, w, stack = s.shape
        hw = h * w
        g = ring.get("graph")
        if g is None:
            g = ring["graph"] = dict(
                pin_in=torch.empty(hw + 16, dtype=torch.uint8).pin_memory(),
                dev_in=torch.empty(hw + 16, dtype=torch.uint8, device=eng.device),
                out=torch.zeros(1, dtype=torch.int32, device=eng.device),
                pin_out=torch.zeros(1, dtype=torch.int32).pin_memory(),
                graphs={},
            )
            g["np_in"] = g["pin_in"].numpy()
        slot = ring["shifts"] % stack
        g["np_in"][:hw] = s[..., -1].reshape(-1)
        g["np_in"][hw : hw + 4].view(np.int32)[0] = idx_network
        current = eng._mirror_is_current(None)
        key = (slot, current)
        graph = g["graphs"].get(key)
        if graph is None:
            ids = ring["rot"][(ring["shifts"] + 1) % stack]
            idx_dev = g["dev_in"][hw : hw + 4].view(torch.int32)

            def enqueue():
                g["dev_in"].copy_(g["pin_in"], non_blocking=True)
                ring["planes"][slot].copy_(g["dev_in"][:hw])
                eng.best_actions(frames=ring["planes"], frame_stride=hw, frame_ids=ids, idx_networks=idx_dev, out=g["out"],
                                 mirror_current=current)
                g["pin_out"].copy_(g["out"], non_blocking=True)

            torch.cuda.synchronize(eng.device)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                enqueue()
            g["graphs"][key] = graph
        graph.replay()
        eng._mirror_holds(None)
        ring["shifts"] += 1
        ring["last"] = s
        torch.cuda.current_stream(eng.device).synchronize()
        return int(g["pin_out"][0])

    def _best_action(self, params, state, idx_network: int) -> int:
        """One observation: newest frame up, one forward (weight mirror reused when nothing wrote the parameters since the
        last learn step), one 4-byte read back."""
        eng = self._engine
        if self._bind(params) is None and self.architecture_type != "fc" and getattr(self, "act_graph", True):
            ring = self._ring
            s = np.asarray(state)
            s = s.astype(np.uint8) if s.dtype != np.uint8 else s.copy()
            if ring is not None and ring["last"] is not None and ring["last"].shape == s.shape and _is_one_frame_shift(ring["last"], s):
                return self._act_graph_step(s, int(idx_network))
            state = s
        heads = getattr(self, "_head_ids", None)
        if heads is None or heads.device != eng.device:
            heads = self._head_ids = torch.arange(max(self._n_heads, 1), dtype=torch.int32, device=eng.device)
        out = eng.best_actions(idx_networks=heads[idx_network : idx_network + 1], params=self._bind(params), **self._obs_to_device(state))
        return int(out.item())

    def _best_actions(self, params, states, idx_networks) -> np.ndarray:
        """Greedy actions of n observations in one forward and one device->host copy."""
        eng = self._engine
        idx = torch.as_tensor(np.array(idx_networks, dtype=np.int32)).to(eng.device)
        out = eng.best_actions(idx_networks=idx, params=self._bind(params), **self._states_to_device(states))
        return out.cpu().numpy()

    def _best_actions_planes(self, params, planes: np.ndarray, rows: np.ndarray, idx_networks, wait: bool = True):
        """Greedy actions of the environments ``rows`` of a VectorEnv: ``planes`` is its host block uint8 [n][stack][h*w]
        (planar stacks, oldest .. newest; pinned when the runtime registered the mapping).  ONE host-to-device copy of the
        block, one small copy of (plane ids, head indices), one forward over len(rows) observations, one read-back."""
        eng = self._engine
        n, stack, hw = planes.shape
        v = getattr(self, "_vec", None)
        if v is None or v["dev"].shape != (n * stack, hw) or v["dev"].device != eng.device:
            v = self._vec = dict(
                dev=torch.empty(n * stack, hw, dtype=torch.uint8, device=eng.device),
                small=torch.empty(n * (stack + 1), dtype=torch.int32, device=eng.device),
                host_small=torch.empty(n * (stack + 1), dtype=torch.int32).pin_memory(),
                out=torch.empty(n, dtype=torch.int32, device=eng.device),
                host_out=torch.empty(n, dtype=torch.int32).pin_memory(),
                ar=np.arange(stack, dtype=np.int32),
            )
        m = len(rows)
        v["dev"].copy_(torch.from_numpy(planes.reshape(n * stack, hw)), non_blocking=True)
        hs = v["host_small"].numpy()
        hs[: m * stack] = (np.asarray(rows, np.int32)[:, None] * stack + v["ar"][None, :]).reshape(-1)
        hs[n * stack : n * stack + m] = np.asarray(idx_networks, np.int32)
        v["small"].copy_(v["host_small"], non_blocking=True)
        eng.best_actions(frames=v["dev"], frame_stride=hw, frame_ids=v["small"][: m * stack], idx_networks=v["small"][n * stack : n * stack + m],
                         params=self._bind(params), out=v["out"][:m])
        v["host_out"][:m].copy_(v["out"][:m], non_blocking=True)
        ev = v.setdefault("event", torch.cuda.Event())
        ev.record(torch.cuda.current_stream(eng.device))

        def result() -> np.ndarray:
            ev.synchronize()  # (this forward only: work enqueued behind it -- the round's gradient steps -- keeps running)
            return v["host_out"][:m].numpy().astype(np.int64)

        return result() if wait else result

    # generic forms (one head, per-step replays); iSDQN overrides both
    def best_actions_planes(self, params, planes, rows, key=None, wait: bool = True):
        return self._best_actions_planes(params, planes, rows, np.zeros(len(rows), dtype=np.int32), wait=wait)

    def learn_steps(self, n_steps: int, replay_buffer) -> None:
        for _ in range(n_steps):
            self.update_online_params(0, replay_buffer)

    def _q_row(self, params, state) -> torch.Tensor:
        """network.apply on one observation: device row of n_heads * n_actions values."""
        return self._engine.forward(n_rows=1, params=self._bind(params), **self._obs_to_device(state))

    def get_model(self):
        """The reference's `{"params": self.params}` (isdqn.py:137-138), where `self.params` is the full Flax variables dict that
        `network.init` returned: `{"params": tree}` plus, for a BatchNorm network, the second collection `"batch_stats"` beside it
        (isdqn.py:87-88).  The pickle the trainer writes is therefore `{"params": {"params": tree[, "batch_stats": tree]}}`, and
        `network.apply(model["params"], ...)` works on it where JAX exists.  `_bind` loads either nesting back."""
        variables = {"params": self.params.to_flax()}
        if self.batch_norm:
            variables["batch_stats"] = self.params.batch_stats()
        return {"params": variables}
