"""Pins the oracle samplers + replay buffer with the known answers of the reference's
tests/test_samplers.py (:10-35) and tests/test_replay_buffer.py (:49-203), restated."""
import numpy as np

from oracle.replay_buffer import ReplayBuffer, ReplayElement, TransitionElement
from oracle.samplers import PrioritizedSamplingDistribution, UniformSamplingDistribution

OBS = (84, 84)
STACK = 4
BATCH = 32


def test_prioritized_sampler_sequence():  # test_samplers.py:18-31
    s = PrioritizedSamplingDistribution(seed=0, max_capacity=10)
    for key, prio in zip([0, 1, 2, 3, 4], [1.0, 2.0, 3.0, 4.0, 0.0]):
        s.add(key, priority=prio)
    assert (s.sample(5) < 4).all()
    s.update(keys=np.array([2, 3]), priorities=np.array([0.0, 0.0]))
    assert (s.sample(5) < 2).all()
    s.remove(0)
    np.testing.assert_array_equal(s.sample(5), 1)


def test_element_pack_unpack():  # test_replay_buffer.py:20-47
    e = ReplayElement(
        state=np.zeros(OBS + (STACK,), np.uint8), action=1, reward=1.0,
        next_state=np.ones(OBS + (STACK,), np.uint8), is_terminal=False,
    )
    u = e.pack().unpack()
    assert (u.action, u.reward, u.is_terminal) == (1, 1.0, False)
    np.testing.assert_array_equal(u.state, e.state)
    np.testing.assert_array_equal(u.next_state, e.next_state)


def _rb(capacity, n=1, gamma=1.0, stack=STACK):
    return ReplayBuffer(
        sampling_distribution=UniformSamplingDistribution(seed=0), batch_size=BATCH, max_capacity=capacity,
        stack_size=stack, update_horizon=n, gamma=gamma, compress=False,
    )


def test_add_up_to_capacity():  # :49-85
    rb = _rb(10)
    tr = []
    for i in range(16):
        tr.append(TransitionElement(np.full(OBS, i), i, i, False, False))
        rb.add(tr[-1])
    assert list(rb._memory.keys()) == list(range(5, 15))
    for i in range(5, 15):
        np.testing.assert_array_equal(
            rb._memory[i].state, np.array([t.observation for t in tr[i - STACK + 1 : i + 1]]).transpose(1, 2, 0)
        )
        np.testing.assert_array_equal(
            rb._memory[i].next_state, np.array([t.observation for t in tr[i - STACK + 2 : i + 2]]).transpose(1, 2, 0)
        )
        assert rb._memory[i].action == tr[i].action
        assert rb._memory[i].reward == tr[i].reward
        assert rb._memory[i].is_terminal == 0


def test_n_step_rewards():  # :87-105
    rb = _rb(10, n=5)
    for i in range(50):
        rb.add(TransitionElement(np.full(OBS, i), 0, 2.0, False))
    for _ in range(100):
        np.testing.assert_array_equal(rb.sample().reward, np.ones(BATCH) * 10.0)


def test_get_stack():  # :107-133
    rb = _rb(50)
    for i in range(11):
        rb.add(TransitionElement(np.full(OBS, i), 0, 0, False))
    for k in rb._memory:
        assert rb._memory[k].state.shape == OBS + (4,)
    np.testing.assert_array_equal(rb._memory[0].state[:, :, :3], np.zeros(OBS + (3,)))
    st = rb._memory[STACK - 1].state
    for i in range(STACK):
        np.testing.assert_array_equal(st[:, :, i], np.full(OBS, i))


def test_key_mappings_for_sampling():  # :135-203
    capacity = 10
    rb = _rb(capacity, gamma=0.99, stack=1)
    sampler = rb._sampling_distribution
    for i in range(capacity + 1):
        rb.add(TransitionElement(np.full(OBS, i), i, i, False, False))
    for i in range(capacity):
        assert sampler._key_to_index[i] == i and sampler._index_to_key[i] == i
    nk = capacity
    rb.add(TransitionElement(np.full(OBS, nk + 1), nk + 1, nk + 1, False, False))
    assert 0 not in sampler._key_to_index
    assert sampler._index_to_key[0] != 0
    assert sampler._index_to_key[sampler._key_to_index[nk]] == nk
    idx = np.random.default_rng(seed=0).integers(len(sampler._index_to_key), size=BATCH)
    keys = [sampler._index_to_key[i] for i in idx]
    s = rb.sample()
    for i, key in enumerate(keys):
        np.testing.assert_array_equal(s.state[i], np.full(OBS, key)[..., None])
        np.testing.assert_array_equal(s.next_state[i], np.full(OBS, key + 1)[..., None])
        assert (s.action[i], s.reward[i], s.is_terminal[i]) == (key, key, 0)


def test_terminal_flush_and_short_episode():
    """replay_buffer.py:159-177 -- elements emitted around a terminal transition."""
    rb = _rb(100, n=3, gamma=0.5)
    # long episode: 8 transitions, terminal at the last
    for i in range(8):
        rb.add(TransitionElement(np.full((2, 2), i + 1, np.uint8), i, 1.0, i == 7, False))
    els = list(rb._memory.values())
    # non-terminal elements for state_last = 0..4, then terminal flush for 5, 6, 7
    assert [int(e.action) for e in els] == [0, 1, 2, 3, 4, 5, 6, 7]
    assert [bool(e.is_terminal) for e in els] == [False] * 5 + [True] * 3
    assert [e.reward for e in els] == [1.75] * 5 + [1.75, 1.5, 1.0]
    # last element: state = frames 5..8, next_state = frames beyond the episode -> zeros after shift
    np.testing.assert_array_equal(els[7].state[0, 0], [5, 6, 7, 8])
    np.testing.assert_array_equal(els[7].next_state[0, 0], [8, 0, 0, 0])
    np.testing.assert_array_equal(els[0].state[0, 0], [0, 0, 0, 1])
    np.testing.assert_array_equal(els[0].next_state[0, 0], [1, 2, 3, 4])
    # short episode (terminal before stack+n observations)
    rb2 = _rb(100, n=3, gamma=0.5)
    for i in range(3):
        rb2.add(TransitionElement(np.full((2, 2), i + 1, np.uint8), i, 1.0, i == 2, False))
    els = list(rb2._memory.values())
    assert [int(e.action) for e in els] == [0, 1, 2]
    assert [bool(e.is_terminal) for e in els] == [True, True, True]
    assert [e.reward for e in els] == [1.75, 1.5, 1.0]
    # truncation without terminal drops the tail
    rb3 = _rb(100, n=2)
    for i in range(5):
        rb3.add(TransitionElement(np.full((2, 2), i, np.uint8), i, 1.0, False, i == 4))
    assert [int(e.action) for e in rb3._memory.values()] == [0, 1, 2]
    assert len(rb3._trajectory) == 0
