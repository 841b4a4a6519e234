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


def collect_vector_samples(key, venv, agent, rb: ReplayBuffer, p, epsilon_schedule, n_training_steps: int):
    """One round of ``collect_single_sample`` over the n environments of a VectorEnv: the epsilon draws per environment as in
    ``select_action``, ONE batched forward for the greedy ones, then every environment steps and adds to its own stream
    of the replay buffer.  Returns [(reward, episode_end)] in environment order."""
    n = len(venv)
    actions = np.empty(n, dtype=np.int64)
    greedy = []
    for i in range(n):
        if key.random() <= epsilon_schedule(n_training_steps + i):
            actions[i] = key.integers(0, venv.n_actions)
        else:
            greedy.append(i)
    if greedy:
        states = np.stack([np.asarray(venv.envs[i].state).astype(np.uint8, copy=False) for i in greedy])
        actions[greedy] = agent.best_actions(agent.params, states, key=key)
    out = []
    for i, env in enumerate(venv.envs):
        obs = env.observation
        reward, absorbing = env.step(int(actions[i]))
        episode_end = absorbing or env.n_steps >= p["horizon"]
        rb.add(
            TransitionElement(
                observation=obs,
                action=int(actions[i]),
                reward=reward if rb._clipping is None else rb._clipping(reward),
                is_terminal=absorbing,
                episode_end=episode_end,
            ),
            stream=i,
        )
        if episode_end:
            env.reset()
        out.append((reward, episode_end))
    return out
