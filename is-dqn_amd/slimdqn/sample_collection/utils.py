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
