"""GPU drop-in tests of the device replay + samplers: the known answers of the reference's
tests/test_replay_buffer.py (:49-203) and tests/test_samplers.py (:18-31) restated on the HIP-backed
classes, plus batch-for-batch equality with the oracle replay on the same seed."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

OBS = (84, 84)
STACK = 4
BATCH = 32


def _rb(capacity, n=1, gamma=1.0, stack=STACK, seed=0, prioritized=False):
    from slimdqn.sample_collection.replay_buffer import ReplayBuffer
    from slimdqn.sample_collection.samplers import PrioritizedSamplingDistribution, UniformSamplingDistribution

    sampler = PrioritizedSamplingDistribution(seed, capacity) if prioritized else UniformSamplingDistribution(seed)
    return ReplayBuffer(sampling_distribution=sampler, batch_size=BATCH, max_capacity=capacity, stack_size=stack,
                        update_horizon=n, gamma=gamma, compress=False)


def _t(*a):
    from slimdqn.sample_collection.replay_buffer import TransitionElement

    return TransitionElement(*a)


def test_add_up_to_capacity():  # test_replay_buffer.py:49-85
    rb = _rb(10)
    tr = []
    for i in range(16):
        tr.append(_t(np.full(OBS, i, np.uint8), i, i, False, False))
        rb.add(tr[-1])
    assert len(rb._memory) == 10
    assert list(rb._memory.keys()) == list(range(5, 15))
    for i in range(5, 15):
        e = rb._memory[i]
        np.testing.assert_array_equal(e.state, np.array([t.observation for t in tr[i - STACK + 1 : i + 1]]).transpose(1, 2, 0))
        np.testing.assert_array_equal(e.next_state, np.array([t.observation for t in tr[i - STACK + 2 : i + 2]]).transpose(1, 2, 0))
        assert (e.action, e.reward, int(e.is_terminal)) == (tr[i].action, tr[i].reward, 0)


def test_n_step_rewards():  # :87-105
    rb = _rb(10, n=5)
    for i in range(50):
        rb.add(_t(np.full(OBS, i, np.uint8), 0, 2.0, False))
    for _ in range(20):
        np.testing.assert_array_equal(rb.sample().reward.cpu().numpy(), np.ones(BATCH, np.float32) * 10.0)


def test_get_stack():  # :107-133
    rb = _rb(50)
    for i in range(11):
        rb.add(_t(np.full(OBS, i, np.uint8), 0, 0, False))
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
        rb.add(_t(np.full(OBS, i, np.uint8), i, i, False, False))
    for i in range(capacity):
        assert sampler._key_to_index[i] == i and sampler._index_to_key[i] == i
    nk = capacity
    rb.add(_t(np.full(OBS, nk + 1, np.uint8), nk + 1, nk + 1, False, False))
    assert 0 not in sampler._key_to_index
    assert sampler._index_to_key[0] != 0
    assert sampler._index_to_key[sampler._key_to_index[nk]] == nk
    idx = np.random.default_rng(seed=0).integers(len(sampler._index_to_key), size=BATCH)
    keys = [sampler._index_to_key[i] for i in idx]
    s = rb.sample()
    state, nxt = s.state.cpu().numpy(), s.next_state.cpu().numpy()
    for i, key in enumerate(keys):
        np.testing.assert_array_equal(state[i], np.full(OBS, key)[..., None])
        np.testing.assert_array_equal(nxt[i], np.full(OBS, key + 1)[..., None])
        assert (int(s.action[i]), float(s.reward[i]), int(s.is_terminal[i])) == (key, key, 0)


def test_prioritized_sampler_sequence():  # test_samplers.py:18-31
    from slimdqn.sample_collection.samplers import PrioritizedSamplingDistribution

    s = PrioritizedSamplingDistribution(seed=0, max_capacity=10)
    for key, prio in zip([0, 1, 2, 3, 4], [1.0, 2.0, 3.0, 4.0, 0.0]):
        s.add(key, priority=prio)
    assert (s.sample(5) < 4).all()
    s.update(keys=np.array([2, 3]), priorities=np.array([0.0, 0.0]))
    assert (s.sample(5) < 2).all()
    s.remove(0)
    np.testing.assert_array_equal(s.sample(5), 1)


@pytest.mark.parametrize("prioritized", [False, True])
def test_sampled_batches_equal_oracle_batches(prioritized):
    """Same seed, same transition stream -> the device replay returns exactly the oracle's batches
    (keys through the PCG64 stream / the sum tree, stacks, actions, n-step rewards, terminals)."""
    from oracle.replay_buffer import ReplayBuffer as ORB, TransitionElement as OT
    from oracle.samplers import PrioritizedSamplingDistribution as OP, UniformSamplingDistribution as OU

    capacity, n = 60, 3  # not a power of two: the reference's tree has no leaf `capacity` then (see samplers.py)
    rb = _rb(capacity, n=n, gamma=0.99, seed=7, prioritized=prioritized)
    orb = ORB(OP(7, capacity) if prioritized else OU(7), BATCH, capacity, stack_size=STACK, update_horizon=n, gamma=0.99)
    rng = np.random.default_rng(1)
    for t in range(300):
        obs = rng.integers(0, 256, OBS, dtype=np.uint8)
        a, r = int(rng.integers(0, 9)), float(rng.choice([-1.0, 0.0, 1.0]))
        term = bool(rng.random() < 0.03)
        end = term or bool(rng.random() < 0.02)
        kw = {"priority": float(rng.uniform(0.1, 2.0))} if prioritized else {}
        rb.add(_t(obs, a, r, term, end), **kw)
        orb.add(OT(obs, a, r, term, end), **kw)
        if t > 20 and t % 25 == 0:
            got, exp = rb.sample(), orb.sample()
            np.testing.assert_array_equal(got.state.cpu().numpy(), exp.state)
            np.testing.assert_array_equal(got.next_state.cpu().numpy(), exp.next_state)
            np.testing.assert_array_equal(got.action.cpu().numpy(), exp.action)
            np.testing.assert_array_equal(got.reward.cpu().numpy(), exp.reward.astype(np.float32))
            np.testing.assert_array_equal(got.is_terminal.cpu().numpy().astype(bool), exp.is_terminal.astype(bool))
            if prioritized:  # writeback of new priorities on both sides, through update(keys) and update_device
                keys = rb._sampling_distribution.keys_of(got.indices.cpu().numpy())
                pr = rng.uniform(0.0, 3.0, BATCH)
                import torch

                rb.update_device(got, torch.from_numpy(pr).cuda())
                orb.update(keys, priorities=pr)
    if prioritized:
        np.testing.assert_array_equal(rb._sampling_distribution._sum_tree._nodes, orb._sampling_distribution._sum_tree._nodes)


def test_power_of_two_capacity_survives_the_first_eviction():
    """The reference's tree has no leaf `capacity` when capacity is a power of two (IndexError at the first
    eviction, sum_tree.py:33); the device sampler adds one spare leaf and samples identically otherwise."""
    rb = _rb(8, prioritized=True, stack=1)
    for i in range(20):
        rb.add(_t(np.full(OBS, i, np.uint8), i, 0.0, False, False), priority=1.0 + i)
    keys = rb._sampling_distribution.sample(64)
    assert keys.min() >= 20 - 1 - 8 and keys.max() <= 18
