"""The training loop with the reference's cadence (experiments/base/dqn.py:13-85), JAX-free.

Per environment step: epsilon-greedy action -> env.step -> replay add; once more than ``n_initial_samples``
steps were collected, ``update_online_params`` (a gradient step every ``data_to_update`` steps) and
``update_target_params`` (head shift + loss logs every ``target_update_frequency`` steps).  An epoch ends
after ``n_training_steps_per_epoch`` steps at the next episode boundary; per epoch the returns are saved and
the model is kept when the epoch's average return is the best so far.

One process drives one GPU; with torch.distributed initialised (one seed/game per rank, launched by
experiments/launch.py) the only communication is an all_gather of a few per-epoch scalars.
"""
import os
import time

import numpy as np

from experiments.base import dist as replicas
from experiments.base.utils import save_data
from slimdqn.sample_collection.utils import collect_single_sample, collect_vector_samples, linear_schedule

EPOCH_FIELDS = ("avg_return", "avg_length_episode", "n_training_steps", "env_steps_per_s")


def _gather_epoch_metrics(metrics: np.ndarray):
    """all_gather of the per-epoch metric vector over RCCL (xGMI); identity when not distributed."""
    return replicas.gather_metrics(metrics)


def train(key, p: dict, agent, env, rb):
    """``key``: a numpy Generator (the reference threads a jax PRNGKey through the loop, dqn.py:34)."""
    rng = key if isinstance(key, np.random.Generator) else np.random.default_rng(key)
    # This loop owns the agent: every parameter write it causes goes through the agent's own methods (learn steps, head shifts,
    # imports), so it may declare the weight mirror trusted between calls -- the acting forward of every environment step then skips
    # an 8 us rebuild (one-environment loop: 918 -> 979 gradient steps/s).  Anyone who writes the parameters behind the agent's back
    # keeps the default (slimdqn/_engine.py: trust_mirror).
    if hasattr(agent, "trust_mirror"):
        agent.trust_mirror = True
    epsilon_schedule = linear_schedule(1.0, p["epsilon_end"], p["epsilon_duration"])
    n_training_steps = 0
    env.reset()
    returns, lengths = [[0]], [[0]]
    best_avg_return = -float("inf")
    gathered = []

    vector = hasattr(env, "envs")  # VectorEnv: n environments per round, same per-step cadence of the updates
    run_return, run_length = ([0.0] * len(env), [0] * len(env)) if vector else (None, None)

    analysis_logs = {"srank": [], "dead_neurons": []}

    def after_target_update(step, logs):
        if p.get("analysis"):  # dqn.py:54-58 of the reference
            from experiments.base.srank_and_dead_neurons import eval_srank_and_dead_neurons

            at_update = eval_srank_and_dead_neurons(agent.params, rb, p)
            logs.update(at_update)
            for metric in analysis_logs:
                analysis_logs[metric].append(at_update[metric])
        p["wandb"].log({"n_training_steps": step, **logs})

    def after_step():
        if n_training_steps > p["n_initial_samples"]:
            agent.update_online_params(n_training_steps, rb)
            updated, logs = agent.update_target_params(n_training_steps)
            if updated:
                after_target_update(n_training_steps, logs)

    def after_round(first_step, count):
        """The cadence of `after_step` for the steps first_step+1 .. first_step+count of one vector round, with the gradient
        steps between two target updates issued together (agent.learn_steps: one graph replay)."""
        owed = 0
        for step in range(first_step + 1, first_step + count + 1):
            if step <= p["n_initial_samples"]:
                continue
            if step % agent.data_to_update == 0:
                owed += 1
            if step % agent.target_update_frequency == 0:
                agent.learn_steps(owed, rb)
                owed = 0
                updated, logs = agent.update_target_params(step)
                if updated:
                    after_target_update(step, logs)
        agent.learn_steps(owed, rb)

    for idx_epoch in range(p["n_epochs"]):
        steps_in_epoch, has_reset = 0, False
        t_epoch = time.perf_counter()
        if vector:
            returns[idx_epoch], lengths[idx_epoch] = [], []  # finished episodes of this epoch, any environment
        while steps_in_epoch < p["n_training_steps_per_epoch"] or not has_reset:
            if vector:
                # (collects the round started by the previous call, starts the next one, returns: the gradient steps below
                # are enqueued while the emulators run)
                first_step = n_training_steps
                round_results = collect_vector_samples(rng, env, agent, rb, p, epsilon_schedule, n_training_steps,
                                                       between=lambda res: after_round(first_step, len(res)))
                for i, (reward, ended) in enumerate(round_results):
                    run_return[i] += reward
                    run_length[i] += 1
                    if ended:
                        returns[idx_epoch].append(run_return[i])
                        lengths[idx_epoch].append(run_length[i])
                        run_return[i], run_length[i] = 0.0, 0
                        has_reset = True
                steps_in_epoch += len(round_results)
                n_training_steps += len(round_results)
                continue
            reward, has_reset = collect_single_sample(rng, env, agent, rb, p, epsilon_schedule, n_training_steps)
            steps_in_epoch += 1
            n_training_steps += 1
            returns[idx_epoch][-1] += reward
            lengths[idx_epoch][-1] += 1
            if has_reset and steps_in_epoch < p["n_training_steps_per_epoch"]:
                returns[idx_epoch].append(0)
                lengths[idx_epoch].append(0)
            after_step()

        avg_return = float(np.mean(returns[idx_epoch]))
        avg_length = float(np.mean(lengths[idx_epoch]))
        print(f"\nEpoch {idx_epoch}: Return {avg_return} averaged on {len(lengths[idx_epoch])} episodes.\n", flush=True)
        p["wandb"].log({"epoch": idx_epoch, "n_training_steps": n_training_steps, "avg_return": avg_return,
                        "avg_length_episode": avg_length})
        rate = steps_in_epoch / max(time.perf_counter() - t_epoch, 1e-9)
        gathered.append(_gather_epoch_metrics(np.asarray([avg_return, avg_length, n_training_steps, rate], np.float32)))

        model = None
        if avg_return > best_avg_return:
            best_avg_return = avg_return
            model = agent.get_model()
        if idx_epoch < p["n_epochs"] - 1:
            returns.append([0])
            lengths.append([0])
        save_data(p, returns, lengths, model, analysis_logs)
        if os.environ.get("WORLD_SIZE", "1") != "1":  # rank 0: every replica's per-epoch metrics in one file
            replicas.write_gathered(os.path.join(os.path.dirname(os.path.dirname(p["save_path"])), "gathered_metrics.json"),
                                    gathered, EPOCH_FIELDS)
    return gathered
