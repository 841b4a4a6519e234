"""AnalysisDQN with the reference's surface (slimdqn/networks/analysisdqn.py:14-236) on the HIP engine: iS-DQN training plus,
at every gradient step, the diagnostics the reference logs --

  * target churn on the training batch and on a second ("eval") batch: mean_b |target before the update - target after it| per
    head (analysisdqn.py:123-160);
  * cosine similarity between the iS-DQN gradient and a target-based gradient, and between a target-free gradient and the
    target-based one (analysisdqn.py:162-219).  The three gradients come from the library's gradient-only pass
    (isdqn_net_grad_on_batch): the iS loss is the step's own gradient, the two single-pair losses regress online head 1 on
    target head 1 -- of a parameter copy taken at the last target update (tb) or of the same parameters (tf).

The step itself is the iS-DQN step of iSDQN (same kernels, same bits); the diagnostics cost three more forward/backward passes
and four forwards, as in the reference."""
from __future__ import annotations

import numpy as np
import torch

from slimdqn.networks._agent import DeviceParams
from slimdqn.networks.isdqn import iSDQN


class AnalysisDQN(iSDQN):
    def __init__(self, *args, **kwargs):
        kwargs["use_graph"] = False  # every update reads diagnostics back: nothing to capture
        super().__init__(*args, **kwargs)
        self.target_params = self.params.clone()  # analysisdqn.py:49
        K = self.n_bellman_iterations
        self.cumulated_target_churns_train = np.zeros(K)
        self.cumulated_target_churns_eval = np.zeros(K)
        self.cumulated_cosine_sim_is_to_tb = 0.0
        self.cumulated_cosine_sim_tf_to_tb = 0.0
        self._feature_mask = None

    def _engine_changed(self, old) -> None:
        self._feature_mask = None
        if old is not None and getattr(self, "target_params", None) is not None:
            self.target_params = DeviceParams(self._engine, self.target_params.tensor.clone())

    # ------------------------------------------------------------------ analysisdqn.py:63-121
    def update_online_params(self, step: int, replay_buffer):
        if step % self.data_to_update == 0:
            batch_samples = replay_buffer.sample()
            batch_samples_eval = replay_buffer.sample()
            (self.params, self.optimizer_state, losses, churn_train, churn_eval, cos_is_tb, cos_tf_tb) = self.learn_on_batch(
                self.params, self.target_params, self.optimizer_state, batch_samples, batch_samples_eval)
            # (the step accumulates `losses` on the device as iSDQN does; the diagnostics are host sums like the reference's)
            self.cumulated_target_churns_train += churn_train
            self.cumulated_target_churns_eval += churn_eval
            self.cumulated_cosine_sim_is_to_tb += cos_is_tb
            self.cumulated_cosine_sim_tf_to_tb += cos_tf_tb

    def update_target_params(self, step: int):
        if step % self.target_update_frequency == 0:
            self.target_params = self.params.clone()  # before the window shift (analysisdqn.py:89-90)
            norm = self.target_update_frequency / self.data_to_update
            updated, logs = super().update_target_params(step)  # shift, "loss", "networks/k_loss"
            logs["analysis/target_churns_train"] = self.cumulated_target_churns_train[0] / norm
            logs["analysis/target_churns_eval"] = self.cumulated_target_churns_eval[0] / norm
            logs["analysis/cosine_sim_iS_to_TB"] = self.cumulated_cosine_sim_is_to_tb / norm
            logs["analysis/cosine_sim_TF_to_TB"] = self.cumulated_cosine_sim_tf_to_tb / norm
            for k in range(min(self.n_bellman_iterations, 5)):
                logs[f"networks/{k}_target_churns_train"] = self.cumulated_target_churns_train[k] / norm
                logs[f"networks/{k}_target_churns_eval"] = self.cumulated_target_churns_eval[k] / norm
            self.cumulated_target_churns_train = np.zeros_like(self.cumulated_target_churns_train)
            self.cumulated_target_churns_eval = np.zeros_like(self.cumulated_target_churns_eval)
            self.cumulated_cosine_sim_is_to_tb = 0.0
            self.cumulated_cosine_sim_tf_to_tb = 0.0
            return updated, logs
        return False, {}

    # ------------------------------------------------------------------ analysisdqn.py:123-219
    def _mask(self) -> torch.Tensor:
        """1 on the entries extract_feature_gradients keeps (analysisdqn.py:196-212): kernels and biases of every Conv / Dense
        (leaves under a "*Norm*" module dropped), the last Dense restricted to head 1 (outputs A .. 2A); 0 elsewhere, padding
        included (padded entries have zero gradients anyway).  Dot products and norms do not depend on the order of entries."""
        if self._feature_mask is None:
            eng, A = self._engine, self.n_actions
            m = torch.zeros(eng.n_param_floats, dtype=torch.float32)
            last = max(i.layer for i in eng.infos)
            for info in eng.infos:
                if info.kind not in (0, 1, 2):
                    continue  # LayerNorm scale / bias
                if info.layer == last:
                    if info.kind == 1:
                        in_p = info.dims[1]
                        m[info.offset + A * in_p : info.offset + 2 * A * in_p] = 1.0
                    elif info.kind == 2:
                        m[info.offset + A : info.offset + 2 * A] = 1.0
                else:
                    m[info.offset : info.offset + info.size] = 1.0
            self._feature_mask = m.to(eng.device)
        return self._feature_mask

    def three_gradients(self, params, params_target, batch_samples):
        """(grad_is, grad_tf, grad_tb) as flat device tensors in the engine's internal layout -- nothing is updated."""
        eng = self._engine_for(self._batch_len(batch_samples))
        cb = self._c_batch(eng, batch_samples)
        p = self._bind(params)
        tp = self._bind(params_target)
        tp = eng.params if tp is None else tp
        g_is, g_tf, g_tb = (torch.zeros_like(eng.params) for _ in range(3))
        eng.grad_on_batch(cb, g_tb, target_params=tp, online_head=1, target_head=1, n_pairs=1, params=p)
        eng.grad_on_batch(cb, g_tf, online_head=1, target_head=1, n_pairs=1, params=p)
        eng.grad_on_batch(cb, g_is, params=p)
        return g_is, g_tf, g_tb

    def _cosine(self, a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
        m = self._mask()
        a, b = a * m, b * m
        return torch.dot(a, b) / (torch.linalg.norm(a) * torch.linalg.norm(b) + 1e-9)

    def learn_on_batch(self, params, params_target, optimizer_state, batch_samples, batch_samples_eval):
        eng = self._engine_for(self._batch_len(batch_samples))
        bound = self._bind(params)
        if bound is not None:
            eng.params.copy_(bound)
        cb, cb_eval = self._c_batch(eng, batch_samples), self._c_batch(eng, batch_samples_eval)
        tp = self._bind(params_target)
        tp = eng.params if tp is None else tp
        g_tf, g_tb, g_is = (torch.empty_like(eng.params) for _ in range(3))
        eng.grad_on_batch(cb, g_tb, target_params=tp, online_head=1, target_head=1, n_pairs=1)
        eng.grad_on_batch(cb, g_tf, online_head=1, target_head=1, n_pairs=1)
        eng.loss_on_batch(cb_eval)
        eval_pre = eng.targets.clone()
        if self.batch_norm:
            # the collection the reference stores with the updated parameters is the one its LAST forward before the update returned:
            # the evaluation batch's (analysisdqn.py:121 overwrites `batch_stats`, :130-131 stores it) -- not the training batch's
            stats = eng.batch_stats_slice()
            eng.commit_batch_stats()
            eval_stats = eng.params[stats].clone()
        losses = eng.learn_on_batch(cb, grad_out=g_is).clone()  # the iS-DQN step; its gradient is the third one
        if self.batch_norm:
            eng.params[stats] = eval_stats
        train_pre = eng.targets.clone()
        eng.loss_on_batch(cb)
        churn_train = (train_pre - eng.targets).abs().mean(dim=0)
        eng.loss_on_batch(cb_eval)
        churn_eval = (eval_pre - eng.targets).abs().mean(dim=0)
        cos = torch.stack((self._cosine(g_is, g_tb), self._cosine(g_tf, g_tb)))
        host = torch.cat((churn_train, churn_eval, cos)).cpu().numpy().astype(np.float64)  # one read-back
        K = self.n_bellman_iterations
        return (self.params, self.optimizer_state, losses, host[:K], host[K : 2 * K], float(host[2 * K]), float(host[2 * K + 1]))
