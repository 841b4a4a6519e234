"""Host-side driver of the HIP Q-network: configuration, parameter import/export between the
reference's Flax pytree layout and the library's internal layout, workspace ownership, and thin
wrappers over the C-ABI entry points.  Pure plumbing: no arithmetic of the hot path lives here.
"""
from __future__ import annotations

import ctypes
import os
import math
from typing import Dict, Sequence

import numpy as np
import torch

from slimdqn import _hip


def _round_up(a, b):
    return (a + b - 1) // b * b


class QNetEngine:
    """One Q-network (slimdqn/networks/architectures/dqn.py DQNNet + isdqn.py head view) on one GPU."""

    def __init__(
        self,
        observation_dim: Sequence[int],
        n_actions: int,
        n_heads: int,
        features: Sequence[int],
        architecture_type: str,
        layer_norm: bool,
        batch_size: int,
        gamma_n: float = 0.99,
        learning_rate: float = 1e-3,
        adam_eps: float = 1e-8,
        precision: str = "bf16x3",
        device: str | None = None,
        huber_delta: float = 0.0,
        batch_norm: bool = False,
    ):
        _hip.require_gpu()
        self.lib = _hip.lib()
        self.device = _hip.resolve_device(device)
        _hip.bind_device(self.device)
        cfg = _hip.NetConfig()
        if architecture_type == "cnn":
            cfg.arch = _hip.ARCH_CNN
            cfg.obs_h, cfg.obs_w, cfg.obs_c = (int(d) for d in observation_dim)
        elif architecture_type == "impala":  # architectures/dqn.py:7-36, 75-88: three Stacks on the generic engine (csrc/impala.h)
            cfg.arch = _hip.ARCH_IMPALA
            cfg.obs_h, cfg.obs_w, cfg.obs_c = (int(d) for d in observation_dim)
        elif architecture_type == "fc":
            cfg.arch = _hip.ARCH_FC
            cfg.obs_h = cfg.obs_w = 1
            cfg.obs_c = int(np.prod(observation_dim))
        else:
            raise NotImplementedError(f"architecture_type={architecture_type!r}: 'cnn', 'impala' or 'fc'")
        feats = [int(f) for f in features]
        if len(feats) > _hip.MAX_FEATURES:
            raise ValueError("too many features")
        cfg.n_features = len(feats)
        for i, f in enumerate(feats):
            cfg.features[i] = f
        cfg.n_actions = int(n_actions)
        cfg.n_heads = int(n_heads)
        cfg.layer_norm = 1 if layer_norm else 0
        cfg.batch_size = int(batch_size)
        cfg.precision = {"bf16x3": _hip.PRECISION_BF16X3, "bf16": _hip.PRECISION_BF16}[precision]
        cfg.gamma_n = float(gamma_n)
        cfg.learning_rate = float(learning_rate)
        cfg.adam_b1, cfg.adam_b2 = 0.9, 0.999
        cfg.adam_eps = float(adam_eps)
        cfg.huber_delta = float(huber_delta)  # 0: the reference's squared TD error
        cfg.batch_norm = 1 if batch_norm else 0  # architectures/dqn.py:52-53, 59-60, 66-67, 73-74, 100-101 (csrc/batchnorm.h)
        self.batch_norm = bool(batch_norm)
        self.cfg = cfg
        self.features = feats
        self.architecture_type = architecture_type
        self.observation_dim = tuple(int(d) for d in observation_dim)
        self.n_actions, self.n_heads = int(n_actions), int(n_heads)
        self.batch_size = int(batch_size)
        self.precision = precision

        n = ctypes.c_int64()
        cnt = ctypes.c_int32()
        _hip.check(self.lib.isdqn_net_param_layout(ctypes.byref(cfg), ctypes.byref(n), None, 0, ctypes.byref(cnt)))
        infos = (_hip.TensorInfo * cnt.value)()
        _hip.check(self.lib.isdqn_net_param_layout(ctypes.byref(cfg), ctypes.byref(n), infos, cnt.value, ctypes.byref(cnt)))
        self.n_param_floats = int(n.value)
        self.infos = list(infos)
        wb = ctypes.c_int64()
        _hip.check(self.lib.isdqn_net_workspace_bytes(ctypes.byref(cfg), ctypes.byref(wb)))
        self.workspace_bytes = int(wb.value)
        with torch.cuda.device(self.device):
            # zero-initialised once: padded rows/columns of gradient slabs are never written and must read as 0
            self.workspace = torch.zeros(self.workspace_bytes // 4, dtype=torch.float32, device=self.device)
            self.params = torch.zeros(self.n_param_floats, dtype=torch.float32, device=self.device)
            self.adam_m = torch.zeros_like(self.params)
            self.adam_v = torch.zeros_like(self.params)
            self.adam_count = torch.zeros(1, dtype=torch.int32, device=self.device)
            K = self.n_regressed = self.n_heads - 1 if self.n_heads >= 2 else 1  # a single head (DQN / TF-DQN) regresses itself
            self.losses = torch.zeros(K, dtype=torch.float32, device=self.device)
            self.losses_accum = torch.zeros(K, dtype=torch.float32, device=self.device)
            self.q_values = torch.zeros(batch_size, K, dtype=torch.float32, device=self.device)
            self.targets = torch.zeros(batch_size, K, dtype=torch.float32, device=self.device)
            self.priorities = torch.zeros(batch_size, dtype=torch.float64, device=self.device)
            self.action_out = torch.zeros(1, dtype=torch.int32, device=self.device)

    # ------------------------------------------------------------------ parameter layout
    def _first_dense_after_conv(self, info) -> bool:
        # (the first Dense behind the torso: layer 3 of the cnn plan, layer 1 behind the impala pseudo-layer)
        return info.kind == 1 and ((self.architecture_type == "cnn" and info.layer == 3) or (self.architecture_type == "impala" and info.layer == 1))

    def _to_internal(self, info, arr: np.ndarray) -> np.ndarray:
        arr = np.asarray(arr, dtype=np.float32)
        d = list(info.dims)
        if info.kind == 0:
            k1, k2, cin, cout = arr.shape
            if self.architecture_type == "cnn" and info.layer == 0 and not self.batch_norm:
                out = np.zeros((d[0], cin, k1 * k2), np.float32)  # [out][plane][ky*8+kx]
                out[:cout] = arr.transpose(3, 2, 0, 1).reshape(cout, cin, k1 * k2)
            else:
                out = np.zeros((d[0], d[1], d[2]), np.float32)  # [out][tap][in_p]
                out[:cout, :, :cin] = arr.transpose(3, 0, 1, 2).reshape(cout, k1 * k2, cin)
            return out.reshape(-1)
        if info.kind == 1:
            in_f, out_f = arr.shape
            out = np.zeros((d[0], d[1]), np.float32)  # [out_p][in_p]
            if self._first_dense_after_conv(info):
                c = self.features[2]
                c_p = _round_up(c, 8)
                npix = in_f // c
                tmp = np.zeros((out_f, npix, c_p), np.float32)
                tmp[:, :, :c] = arr.T.reshape(out_f, npix, c)
                out[:out_f] = tmp.reshape(out_f, npix * c_p)
            else:
                out[:out_f, :in_f] = arr.T
            return out.reshape(-1)
        if info.kind >= 5:  # BatchNorm_i scale / bias / mean / var: dims = [groups, P, C, C padded]
            out = np.zeros(info.size, np.float32)
            if info.ndim == 2:  # spatial site: one value per pixel position, flax shape (H, W)
                out[: d[0]] = arr.reshape(-1)
            else:  # feature site: flax column p * C + c -> internal column p * Cp + c
                out[: d[0]].reshape(d[1], d[3])[:, : d[2]] = arr.reshape(d[1], d[2])
            return out
        out = np.zeros(d[0], np.float32)
        out[: arr.shape[0]] = arr
        return out

    def _from_internal(self, info, flat: np.ndarray) -> np.ndarray:
        d = list(info.dims)
        shape = tuple(info.flax_shape[: info.ndim])
        if info.kind == 0:
            k1, k2, cin, cout = shape
            if self.architecture_type == "cnn" and info.layer == 0 and not self.batch_norm:
                w = flat.reshape(d[0], cin, k1, k2)[:cout]
                return np.ascontiguousarray(w.transpose(2, 3, 1, 0))
            w = flat.reshape(d[0], k1, k2, d[2])[:cout, :, :, :cin]
            return np.ascontiguousarray(w.transpose(1, 2, 3, 0))
        if info.kind == 1:
            in_f, out_f = shape
            w = flat.reshape(d[0], d[1])[:out_f]
            if self._first_dense_after_conv(info):
                c = self.features[2]
                c_p = _round_up(c, 8)
                npix = in_f // c
                w = w.reshape(out_f, npix, c_p)[:, :, :c].reshape(out_f, in_f)
            else:
                w = w[:, :in_f]
            return np.ascontiguousarray(w.T)
        if info.kind >= 5:
            if info.ndim == 2:
                return flat[: d[0]].reshape(shape).copy()
            return np.ascontiguousarray(flat[: d[0]].reshape(d[1], d[3])[:, : d[2]]).reshape(shape)
        return flat[: shape[0]].copy()

    def import_flax(self, params: Dict[str, Dict[str, np.ndarray]], target: torch.Tensor | None = None, batch_stats=None) -> None:
        """Load a reference-layout pytree ({"Conv_0": {"kernel": HWIO, "bias"}, "LayerNorm_0": ...}).  BatchNorm networks: the
        running averages come from ``batch_stats`` ({"BatchNorm_0": {"mean", "var"}}, Flax's second collection); without it
        they keep Flax's initial values (mean 0, var 1)."""
        flat = np.zeros(self.n_param_floats, np.float32)
        for info in self.infos:
            mod, leaf = info.name.decode().rsplit("/", 1)  # (impala: "Stack_0/Conv_1" / "kernel")
            if info.kind >= 7:
                shape = tuple(info.flax_shape[: info.ndim])
                src = batch_stats[mod][leaf] if batch_stats is not None else (np.zeros(shape, np.float32) if info.kind == 7 else np.ones(shape, np.float32))
                v = self._to_internal(info, src)
            else:
                v = self._to_internal(info, params[mod][leaf])
            assert v.size == info.size, (info.name, v.size, info.size)
            flat[info.offset : info.offset + info.size] = v
        (self.params if target is None else target).copy_(torch.from_numpy(flat))

    def export_flax(self, source: torch.Tensor | None = None) -> Dict[str, Dict[str, np.ndarray]]:
        flat = (self.params if source is None else source).detach().cpu().numpy()
        out: Dict[str, Dict[str, np.ndarray]] = {}
        for info in self.infos:
            if info.kind >= 7:  # (running averages: export_batch_stats)
                continue
            mod, leaf = info.name.decode().rsplit("/", 1)  # (impala: "Stack_0/Conv_1" / "kernel")
            out.setdefault(mod, {})[leaf] = self._from_internal(info, flat[info.offset : info.offset + info.size])
        return out

    def export_batch_stats(self, source: torch.Tensor | None = None) -> Dict[str, Dict[str, np.ndarray]]:
        """Flax's ``batch_stats`` collection of a BatchNorm network: {"BatchNorm_i": {"mean", "var"}} (empty without BatchNorm)."""
        flat = (self.params if source is None else source).detach().cpu().numpy()
        out: Dict[str, Dict[str, np.ndarray]] = {}
        for info in self.infos:
            if info.kind >= 7:
                mod, leaf = info.name.decode().rsplit("/", 1)
                out.setdefault(mod, {})[leaf] = self._from_internal(info, flat[info.offset : info.offset + info.size])
        return out

    def init_params(self, seed: int) -> None:
        """Flax defaults (dqn.py:49, 90): xavier_uniform for cnn, lecun_normal for fc; biases 0, LN scale 1.
        Draws come from numpy PCG64 (JAX threefry streams are not reproducible offline)."""
        rng = np.random.default_rng(seed)
        params: Dict[str, Dict[str, np.ndarray]] = {}
        for info in self.infos:
            mod, leaf = info.name.decode().rsplit("/", 1)  # (impala: "Stack_0/Conv_1" / "kernel")
            shape = tuple(info.flax_shape[: info.ndim])
            if info.kind in (0, 1):
                if info.kind == 0:
                    rf = shape[0] * shape[1]
                    fan_in, fan_out = rf * shape[2], rf * shape[3]
                else:
                    fan_in, fan_out = shape
                xavier = self.architecture_type == "cnn" or (self.architecture_type == "impala" and (info.kind == 1 or mod.endswith("/Conv_0")))
                if xavier:  # (impala: kernel_init is passed to each Stack's first conv and to the Dense layers only, dqn.py:17-21, 96)
                    lim = math.sqrt(6.0 / (fan_in + fan_out))
                    w = rng.uniform(-lim, lim, size=shape)
                else:
                    std = math.sqrt(1.0 / fan_in) / 0.87962566103423978
                    w = np.empty(int(np.prod(shape)))
                    filled = 0
                    while filled < w.size:
                        draw = rng.standard_normal(w.size - filled)
                        draw = draw[np.abs(draw) <= 2.0]
                        w[filled : filled + draw.size] = draw
                        filled += draw.size
                    w = w.reshape(shape) * std
                params.setdefault(mod, {})[leaf] = w.astype(np.float32)
            elif info.kind in (3, 5):  # LayerNorm / BatchNorm scale
                params.setdefault(mod, {})[leaf] = np.ones(shape, np.float32)
            elif info.kind >= 7:  # running averages: import_flax's defaults (mean 0, var 1)
                continue
            else:
                params.setdefault(mod, {})[leaf] = np.zeros(shape, np.float32)
        self.import_flax(params)

    # ------------------------------------------------------------------ workspace regions (tests / debugging)
    def region(self, name: str) -> torch.Tensor:
        off, size = ctypes.c_int64(), ctypes.c_int64()
        _hip.check(self.lib.isdqn_net_workspace_region(ctypes.byref(self.cfg), name.encode(), ctypes.byref(off), ctypes.byref(size)))
        return self.workspace[off.value // 4 : (off.value + size.value) // 4]

    # ------------------------------------------------------------------ C-ABI calls
    def make_batch(self, *, frames=None, frame_stride=0, frame_ids=None, state=None, next_state=None, action=None, reward=None, terminal=None,
                   mirror_current: bool = False, priorities_ready: torch.cuda.Event | None = None) -> _hip.Batch:
        """``mirror_current``: promise that the previous call on this engine was learn_on_batch and nothing wrote the
        parameters since (include/isdqn_hip.h, ISDQN_BATCH_MIRROR_CURRENT) -- only the captured multi-step graphs do.
        ``priorities_ready``: an event (already recorded once, so that its handle exists) the learn call records on the
        caller's stream as soon as q_values / targets / priorities are final."""
        b = _hip.Batch()
        b.flags = _hip.BATCH_MIRROR_CURRENT if mirror_current else 0
        b.priorities_ready = None if priorities_ready is None else int(priorities_ready.cuda_event)
        b.B = self.batch_size
        b.frames = _hip.ptr(frames)
        b.frame_stride = int(frame_stride)
        b.frame_ids = _hip.ptr(frame_ids)
        b.state = _hip.ptr(state)
        b.next_state = _hip.ptr(next_state)
        b.action = _hip.ptr(action)
        b.reward = _hip.ptr(reward)
        b.terminal = _hip.ptr(terminal)
        # keep the tensors alive for the duration of the asynchronous call
        b._keep = (frames, frame_ids, state, next_state, action, reward, terminal, priorities_ready)
        return b

    def forward(self, *, frames=None, frame_stride=0, frame_ids=None, obs=None, n_rows: int, params=None) -> torch.Tensor:
        q = torch.empty(n_rows, self.n_heads * self.n_actions, dtype=torch.float32, device=self.device)
        p = self.params if params is None else params
        _hip.check(
            self.lib.isdqn_net_forward(
                ctypes.byref(self.cfg), _hip.ptr(p), _hip.ptr(frames), int(frame_stride), _hip.ptr(frame_ids),
                _hip.ptr(obs), int(n_rows), _hip.ptr(q), _hip.ptr(self.workspace), _hip.stream_ptr(self.device),
            ),
            "isdqn_net_forward",
        )
        self._mirror_holds(params)
        return q

    def learn_on_batch(self, batch: _hip.Batch, grad_out: torch.Tensor | None = None) -> torch.Tensor:
        """One gradient step in place; returns the device tensor of per-head losses (no sync)."""
        args = [
            ctypes.byref(self.cfg), _hip.ptr(self.params), _hip.ptr(self.adam_m), _hip.ptr(self.adam_v),
            _hip.ptr(self.adam_count), ctypes.byref(batch), _hip.ptr(self.losses), _hip.ptr(self.losses_accum), _hip.ptr(self.q_values),
            _hip.ptr(self.targets), _hip.ptr(self.priorities), _hip.ptr(self.workspace), _hip.stream_ptr(self.device),
        ]
        if grad_out is None:
            rc = self.lib.isdqn_net_learn_on_batch(*args)
        else:
            rc = self.lib.isdqn_net_learn_on_batch_debug(*args, _hip.ptr(grad_out))
        _hip.check(rc, "isdqn_net_learn_on_batch")
        self._mirror_made_current()  # Adam wrote the updated weights in both forms
        return self.losses

    def grad_on_batch(self, batch: _hip.Batch, grad_out: torch.Tensor, target_params: torch.Tensor | None = None, online_head: int = 0,
                      target_head: int = 0, n_pairs: int = 0, params=None) -> torch.Tensor:
        """Gradient of a TD loss into ``grad_out`` (internal layout), nothing updated (analysisdqn.py:156-219): the
        configuration's own loss (n_pairs = 0) or online heads online_head + k on target heads target_head + k; next states
        through ``target_params`` when given.  Returns the device tensor of the loss per pair (the first n_pairs / K entries)."""
        p = self.params if params is None else params
        _hip.check(
            self.lib.isdqn_net_grad_on_batch(
                ctypes.byref(self.cfg), _hip.ptr(p), _hip.ptr(target_params), ctypes.byref(batch), int(online_head), int(target_head),
                int(n_pairs), _hip.ptr(grad_out), _hip.ptr(self.losses), _hip.ptr(self.q_values), _hip.ptr(self.targets),
                _hip.ptr(self.workspace), _hip.stream_ptr(self.device),
            ),
            "isdqn_net_grad_on_batch",
        )
        self._mirror_holds(params)
        return self.losses

    def learn_on_batch_target(self, batch: _hip.Batch, target_params: torch.Tensor) -> torch.Tensor:
        """DQN step (dqn.py:59-72): next states through `target_params`, in-place update of the online parameters."""
        _hip.check(
            self.lib.isdqn_net_learn_on_batch_target(
                ctypes.byref(self.cfg), _hip.ptr(self.params), _hip.ptr(target_params), _hip.ptr(self.adam_m), _hip.ptr(self.adam_v),
                _hip.ptr(self.adam_count), ctypes.byref(batch), _hip.ptr(self.losses), _hip.ptr(self.losses_accum),
                _hip.ptr(self.q_values), _hip.ptr(self.targets), _hip.ptr(self.priorities), _hip.ptr(self.workspace),
                _hip.stream_ptr(self.device),
            ),
            "isdqn_net_learn_on_batch_target",
        )
        self._mirror_made_current()
        return self.losses

    def loss_on_batch_target(self, batch: _hip.Batch, target_params: torch.Tensor, params=None) -> torch.Tensor:
        p = self.params if params is None else params
        _hip.check(
            self.lib.isdqn_net_loss_on_batch_target(
                ctypes.byref(self.cfg), _hip.ptr(p), _hip.ptr(target_params), ctypes.byref(batch), _hip.ptr(self.losses),
                _hip.ptr(self.q_values), _hip.ptr(self.targets), _hip.ptr(self.workspace), _hip.stream_ptr(self.device),
            ),
            "isdqn_net_loss_on_batch_target",
        )
        self._mirror_holds(params)
        return self.losses

    def loss_on_batch(self, batch: _hip.Batch, params=None) -> torch.Tensor:
        p = self.params if params is None else params
        _hip.check(
            self.lib.isdqn_net_loss_on_batch(
                ctypes.byref(self.cfg), _hip.ptr(p), ctypes.byref(batch), _hip.ptr(self.losses),
                _hip.ptr(self.q_values), _hip.ptr(self.targets), _hip.ptr(self.workspace), _hip.stream_ptr(self.device),
            ),
            "isdqn_net_loss_on_batch",
        )
        self._mirror_holds(params)
        return self.losses

    def commit_batch_stats(self) -> None:
        """params["batch_stats"] <- the collection the last training-mode forward in this workspace returned (the analysis agents keep
        the evaluation batch's: analysisdqn.py:121-131)."""
        _hip.check(self.lib.isdqn_net_bn_commit_running(ctypes.byref(self.cfg), _hip.ptr(self.params), _hip.ptr(self.workspace),
                                                        _hip.stream_ptr(self.device)), "isdqn_net_bn_commit_running")

    def batch_stats_slice(self) -> slice:
        """Where the running averages (tensor kinds 7, 8) sit in the flat parameter buffer: one contiguous block behind the optimised ones."""
        stats = [i for i in self.infos if i.kind in (7, 8)]
        lo, hi = min(i.offset for i in stats), max(i.offset + i.size for i in stats)
        assert all(i.offset + i.size <= lo or i.offset >= hi for i in self.infos if i.kind not in (7, 8)), "running averages are not contiguous"
        return slice(lo, hi)

    def shift_params(self, params=None) -> None:
        p = self.params if params is None else params
        _hip.check(self.lib.isdqn_net_shift_params(ctypes.byref(self.cfg), _hip.ptr(p), _hip.stream_ptr(self.device)))
        self._mirror_version = None  # the head rows moved in the master only

    def best_action(self, *, frames=None, frame_stride=0, frame_ids=None, obs=None, idx_network: int, params=None) -> torch.Tensor:
        p = self.params if params is None else params
        _hip.check(
            self.lib.isdqn_net_best_action(
                ctypes.byref(self.cfg), _hip.ptr(p), _hip.ptr(frames), int(frame_stride), _hip.ptr(frame_ids),
                _hip.ptr(obs), int(idx_network), _hip.ptr(self.action_out), _hip.ptr(self.workspace), _hip.stream_ptr(self.device),
            ),
            "isdqn_net_best_action",
        )
        self._mirror_holds(params)
        return self.action_out

    def analysis(self, *, frames=None, frame_stride=0, frame_ids=None, obs=None, n_rows: int, params=None):
        """AnalysisNet.apply (utils/analysis_architecture.py:9-122) on n_rows observations: (features [n_rows, width of the last
        hidden layer], [per-layer sums over the rows of the post-ReLU activations, in the reference's shapes flattened]).  impala:
        the two ReLU outputs of every residual block come first; BatchNorm networks run on the batch statistics of these rows."""
        n_hidden = ctypes.c_int32()
        sizes = (ctypes.c_int64 * 32)()  # (impala: 12 block activations + the torso output + the hidden Dense layers)
        _hip.check(self.lib.isdqn_net_analysis_layout(ctypes.byref(self.cfg), ctypes.byref(n_hidden), sizes, 32))
        sizes = [int(sizes[i]) for i in range(n_hidden.value)]
        feats = torch.empty(n_rows, sizes[-1], dtype=torch.float32, device=self.device)
        scores = torch.empty(sum(sizes), dtype=torch.float32, device=self.device)
        p = self.params if params is None else params
        _hip.check(
            self.lib.isdqn_net_analysis(
                ctypes.byref(self.cfg), _hip.ptr(p), _hip.ptr(frames), int(frame_stride), _hip.ptr(frame_ids), _hip.ptr(obs),
                int(n_rows), _hip.ptr(feats), _hip.ptr(scores), _hip.ptr(self.workspace), _hip.stream_ptr(self.device),
            ),
            "isdqn_net_analysis",
        )
        self._mirror_holds(params)
        return feats, list(torch.split(scores, sizes))

    # ------------------------------------------------------------------ weight-mirror bookkeeping
    # The library keeps a pre-split mirror of the weights in the workspace and rebuilds it at the head of every call unless
    # told that it is current.  DEFAULT: never told -- every entry point outside a captured replay rebuilds (one 8 us launch),
    # and every captured replay rebuilds at its head (slimdqn/_graph.py), so a writer this class cannot see (a raw pointer, a
    # DLPack consumer, `params.data.copy_()`: none of them moves torch's version counter) is still acted on.
    # `trust_mirror = True` is an explicit declaration by the owner of the engine that every parameter write goes through
    # torch operations on `self.params` or through this class: then the rebuild is skipped while (a) the call reads the
    # engine's own buffer, (b) the last call left the mirror equal to it, (c) no torch operation has written the tensor since
    # (version counter) and (d) tensor object and storage are the ones the engine allocated.
    trust_mirror = os.environ.get("ISDQN_TRUST_MIRROR", "0") == "1"

    def _mirror_is_current(self, params) -> bool:
        return bool(self.trust_mirror and params is None and getattr(self, "_mirror_version", None) == self.params._version
                    and self.params.data_ptr() == getattr(self, "_mirror_ptr", None))

    def rebuild_mirror(self) -> None:
        """Rebuild the weight mirror from the engine's parameters (one launch on the current stream), unconditionally."""
        _hip.check(self.lib.isdqn_net_refresh_mirror(ctypes.byref(self.cfg), _hip.ptr(self.params), _hip.ptr(self.workspace),
                                                     _hip.stream_ptr(self.device)), "isdqn_net_refresh_mirror")
        self._mirror_made_current()

    def refresh_mirror(self) -> None:
        """Rebuild the weight mirror from the engine's parameters now (one launch on the current stream) unless it is current."""
        if self._mirror_is_current(None):
            return
        self.rebuild_mirror()

    def invalidate_mirror(self) -> None:
        """Force the next call to rebuild the weight mirror from ``params`` (for writers that bypass torch's version counter)."""
        self._mirror_version = None

    def _mirror_made_current(self) -> None:
        self._mirror_version = self.params._version
        self._mirror_ptr = self.params.data_ptr()

    def _mirror_holds(self, params) -> None:
        """The call just enqueued rebuilt the mirror from `params` (None = the engine's own buffer)."""
        self._mirror_version = self.params._version if params is None else None
        self._mirror_ptr = self.params.data_ptr()

    def best_actions(self, *, frames=None, frame_stride=0, frame_ids=None, obs=None, idx_networks: torch.Tensor, params=None,
                     out: torch.Tensor | None = None, mirror_current: bool | None = None) -> torch.Tensor:
        """Greedy actions of n observations in one forward (isdqn.py:127-135 per row): int32 device tensor [n].
        ``mirror_current``: None = decided by the engine's bookkeeping; a captured graph fixes it at capture time."""
        n = int(idx_networks.numel())
        if out is None:
            out = torch.empty(n, dtype=torch.int32, device=self.device)
        p = self.params if params is None else params
        if mirror_current is None:
            mirror_current = self._mirror_is_current(params)
        flags = _hip.BATCH_MIRROR_CURRENT if mirror_current else 0
        _hip.check(
            self.lib.isdqn_net_best_actions(
                ctypes.byref(self.cfg), _hip.ptr(p), _hip.ptr(frames), int(frame_stride), _hip.ptr(frame_ids), _hip.ptr(obs), n,
                _hip.ptr(idx_networks), _hip.ptr(out), flags, _hip.ptr(self.workspace), _hip.stream_ptr(self.device),
            ),
            "isdqn_net_best_actions",
        )
        self._mirror_holds(params)
        return out

    def internal_to_flax_grads(self, grad_flat: torch.Tensor) -> Dict[str, Dict[str, np.ndarray]]:
        return self.export_flax(grad_flat)
