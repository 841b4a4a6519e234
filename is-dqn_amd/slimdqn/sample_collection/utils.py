"""Acting path glue with the reference's names (slimdqn/sample_collection/utils.py:11-43): epsilon-greedy
``select_action`` and ``collect_single_sample``.  JAX-free: the exploration stream is a numpy Generator
(threefry streams are not reproducible offline, so action streams are not a parity target -- SURVEY 8b)."""
from __future__ import annotations

import numpy as np

from slimdqn.sample_collection.replay_buffer import ReplayBuffer, TransitionElement


def linear_schedule(init_value: float, end_value: float, transition_steps: float):
    """optax.linear_schedule(init, end, steps) as used by experiments/base/dqn.py:20."""

    def schedule(count):
        frac = min(max(count / transition_steps, 0.0), 1.0) if transition_steps > 0 else 1.0
        return init_value + frac * (end_value - init_value)

    return schedule


def select_action(best_action_fn, params, state, key: np.random.Generator, n_actions, epsilon_fn, n_training_steps):
    """utils.py:11-18.  The reference evaluates both branches under jnp.where; only the taken one runs here."""
    if key.random() <= epsilon_fn(n_training_steps):
        return int(key.integers(0, n_actions))
    return int(best_action_fn(params, state, key=key))


def collect_single_sample(key, env, agent, rb: ReplayBuffer, p, epsilon_schedule, n_training_steps: int):
    """utils.py:21-43."""
    action = select_action(agent.best_action, agent.params, env.state, key, env.n_actions, epsilon_schedule, n_training_steps)
    obs = env.observation
    reward, absorbing = env.step(action)
    episode_end = absorbing or env.n_steps >= p["horizon"]
    rb.add(
        TransitionElement(
            observation=obs,
            action=action,
            reward=reward if rb._clipping is None else rb._clipping(reward),
            is_terminal=absorbing,
            episode_end=episode_end,
        )
    )
    if episode_end:
        env.reset()
    return reward, episode_end


def collect_vector_samples(key, venv, agent, rb: ReplayBuffer, p, epsilon_schedule, n_training_steps: int, between=None):
    """One round of ``collect_single_sample`` (utils.py:21-43) over the n environments of a VectorEnv, pipelined:

      1. the environment steps started by the PREVIOUS call are collected (``step_wait``) and their n transitions enter the
         replay buffer, each on its own n-step accumulator (``stream=i``);
      2. the next actions are chosen from the environments' current frame stacks -- epsilon draws per environment as in
         ``select_action`` (environment i of the round uses the schedule at step n_training_steps + results + i, which is what
         n consecutive ``collect_single_sample`` calls would use), ONE batched forward and one read-back for the greedy ones;
      3. ``between(results)`` (the trainer: the cadence of the collected round -- its gradient steps as one graph replay) runs
         while the acting forward is in flight, BEFORE its actions are waited for: the GPU always has the next replay queued
         behind the forward instead of idling through the host's flush / draw / launch work;
      4. ``step_async`` starts the next round and the call returns: the emulators run under the update kernels (with worker
         processes: in parallel on the host cores).

    Returns [(reward, episode_end)] of the collected round in environment order -- empty on the very first call.
    Image observations only (the stacks travel as uint8 planes); an fc agent keeps ``collect_single_sample``."""
    assert getattr(agent, "architecture_type", "cnn") != "fc", "vectorised acting moves uint8 frame stacks: fc observations would be truncated"
    n = len(venv)
    out = []
    if not getattr(venv, "_pin_tried", False):  # the planar block goes up once per round: from pinned memory where the runtime allows
        venv._pin_tried = True
        venv.pin()
    if venv._pending:
        obs, reward, absorbing, episode_end = venv.step_wait()
        actions = venv._last_actions
        clip = rb._clipping
        for i in range(n):
            r = float(reward[i])
            rb.add(TransitionElement(observation=obs[i], action=int(actions[i]), reward=r if clip is None else clip(r),
                                     is_terminal=bool(absorbing[i]), episode_end=bool(episode_end[i])), stream=i)
            out.append((r, bool(episode_end[i])))
    first = n_training_steps + len(out)
    eps = np.fromiter((epsilon_schedule(first + i) for i in range(n)), dtype=np.float64, count=n)
    explore = key.random(n) <= eps
    actions = np.empty(n, dtype=np.int64)
    n_explore = int(explore.sum())
    if n_explore:
        actions[explore] = key.integers(0, venv.n_actions, size=n_explore)
    pending = None
    if n_explore < n:
        greedy = np.flatnonzero(~explore)
        pending = agent.best_actions_planes(agent.params, venv.planes, greedy, key=key, wait=False)
    if between is not None:
        between(out)
    if pending is not None:
        actions[greedy] = pending()
    venv._last_actions = actions
    venv.step_async(actions)
    return out
