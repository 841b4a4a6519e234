"""AnalysisTFDQN with the reference's surface (slimdqn/networks/analysistfdqn.py:14-144) on the HIP engine: target-free DQN
plus the target churn of every update on the training batch and on a second ("eval") batch -- mean_b |target before the update
- target after it| (analysistfdqn.py:81-118)."""
from __future__ import annotations

from slimdqn.networks.tfdqn import TFDQN


class AnalysisTFDQN(TFDQN):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.cumulated_target_churn_train = 0.0
        self.cumulated_target_churn_eval = 0.0

    def update_online_params(self, step: int, replay_buffer):
        if step % self.data_to_update == 0:
            batch_samples = replay_buffer.sample()
            batch_samples_eval = replay_buffer.sample()
            self.params, self.optimizer_state, _, churn_train, churn_eval = self.learn_on_batch(
                self.params, self.optimizer_state, batch_samples, batch_samples_eval)
            self.cumulated_target_churn_train += churn_train
            self.cumulated_target_churn_eval += churn_eval

    def update_target_params(self, step: int):
        if step % self.target_update_frequency == 0:
            norm = self.target_update_frequency / self.data_to_update
            updated, logs = super().update_target_params(step)
            logs["analysis/target_churn_train"] = self.cumulated_target_churn_train / norm
            logs["analysis/target_churn_eval"] = self.cumulated_target_churn_eval / norm
            self.cumulated_target_churn_train = 0.0
            self.cumulated_target_churn_eval = 0.0
            return updated, logs
        return False, {}

    def learn_on_batch(self, params, optimizer_state, batch_samples, batch_samples_eval=None):
        if batch_samples_eval is None:
            return super().learn_on_batch(params, optimizer_state, batch_samples)
        eng = self._engine_for(self._batch_len(batch_samples))
        bound = self._bind(params)
        if bound is not None:
            eng.params.copy_(bound)
        cb, cb_eval = self._c_batch(eng, batch_samples), self._c_batch(eng, batch_samples_eval)
        eng.loss_on_batch(cb_eval)
        eval_pre = eng.targets.clone()
        if self.batch_norm:  # the stored collection is the evaluation forward's (analysistfdqn.py:85-95), as in AnalysisDQN
            stats = eng.batch_stats_slice()
            eng.commit_batch_stats()
            eval_stats = eng.params[stats].clone()
        loss = eng.learn_on_batch(cb)[0].clone()
        if self.batch_norm:
            eng.params[stats] = eval_stats
        train_pre = eng.targets.clone()
        eng.loss_on_batch(cb)
        churn_train = (train_pre - eng.targets).abs().mean()
        eng.loss_on_batch(cb_eval)
        churn_eval = (eval_pre - eng.targets).abs().mean()
        return self.params, self.optimizer_state, loss, float(churn_train), float(churn_eval)
