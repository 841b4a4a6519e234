"""CPU tests of the HOST logic of the device replay (is-dqn_amd/slimdqn/sample_collection/replay_buffer.py):
the n-step accumulator working on frame slots, reference counting of single frames, FIFO eviction and the
sampler index -> element-slot table.  Storage is put on the CPU for these tests (no kernels run; sampling
needs the GPU and is covered by the -m gpu tests).  The checker is the oracle replay buffer."""
import numpy as np
import pytest

from oracle.replay_buffer import ReplayBuffer as OracleRB, TransitionElement as OT
from oracle.samplers import UniformSamplingDistribution as OracleUniform


def _pair(capacity, n, stack, gamma=0.9):
    from slimdqn.sample_collection.replay_buffer import ReplayBuffer
    from slimdqn.sample_collection.samplers import UniformSamplingDistribution

    rb = ReplayBuffer(UniformSamplingDistribution(0, device="cpu"), 8, capacity, stack_size=stack, update_horizon=n,
                      gamma=gamma, device="cpu")
    orb = OracleRB(OracleUniform(0), 8, capacity, stack_size=stack, update_horizon=n, gamma=gamma)
    return rb, orb


@pytest.mark.parametrize("capacity,n,stack", [(7, 1, 4), (25, 3, 4), (13, 5, 2), (40, 2, 1)])
def test_random_stream_matches_oracle(capacity, n, stack):
    from slimdqn.sample_collection.replay_buffer import TransitionElement

    rb, orb = _pair(capacity, n, stack)
    rng = np.random.default_rng(capacity * 100 + n)
    for t in range(400):
        obs = rng.integers(0, 256, (6, 5), dtype=np.uint8)
        action, reward = int(rng.integers(0, 5)), float(rng.normal())
        terminal = bool(rng.random() < 0.08)
        end = terminal or bool(rng.random() < 0.05)
        rb.add(TransitionElement(obs, action, reward, terminal, end))
        orb.add(OT(obs, action, reward, terminal, end))
        assert rb.add_count == orb.add_count
        if t % 37 == 0 or t == 399:
            assert list(rb._memory.keys()) == list(orb._memory.keys())
            for key in orb._memory:
                a, b = rb._memory[key], orb._memory[key]
                np.testing.assert_array_equal(a.state, b.state)
                np.testing.assert_array_equal(a.next_state, b.next_state)
                assert (a.action, a.reward, bool(a.is_terminal)) == (b.action, b.reward, bool(b.is_terminal))
            # sampler maps and the device-side index -> slot table
            s, os_ = rb._sampling_distribution, orb._sampling_distribution
            assert s._index_to_key == os_._index_to_key and s._key_to_index == os_._key_to_index
            rb._flush()
            table = rb._d_index_to_slot.numpy()[: len(s._index_to_key)]
            np.testing.assert_array_equal(table, np.asarray(s._index_to_key) % capacity)
            # reference counting: every frame of a live element or of the trajectory is held, nothing else
            held = np.zeros_like(rb._refcount)
            for key in rb._memory.keys():
                for f in rb._h_elem_frames[key % capacity]:
                    if f >= 0:
                        held[f] += 1
            for entry in rb._trajectory:
                held[entry[0]] += 1
            np.testing.assert_array_equal(held, rb._refcount)
            assert all(rb._refcount[f] == 0 for f in rb._free)
            assert len(set(rb._free)) == len(rb._free)


def test_frame_store_stays_bounded_and_grows_when_needed():
    from slimdqn.sample_collection.replay_buffer import TransitionElement

    rb, _ = _pair(50, 3, 4)
    rng = np.random.default_rng(0)
    for t in range(3000):  # many truncated episodes: frames without elements must be recycled
        rb.add(TransitionElement(rng.integers(0, 256, (4, 4), dtype=np.uint8), 0, 0.0, False, t % 5 == 4))
    assert rb._next_fresh <= rb._n_frame_slots
    assert (rb._refcount > 0).sum() <= 50 * 8 + 7


def test_sampling_without_gpu_fails_loudly():
    from slimdqn.sample_collection.replay_buffer import TransitionElement

    rb, _ = _pair(10, 1, 4)
    for t in range(6):
        rb.add(TransitionElement(np.zeros((4, 4), np.uint8), 0, 0.0, False, False))
    with pytest.raises(RuntimeError):
        rb.sample()


def test_uniform_sampler_host_stream_matches_oracle():
    from slimdqn.sample_collection.samplers import UniformSamplingDistribution

    a, b = UniformSamplingDistribution(3, device="cpu"), OracleUniform(3)
    for k in range(20):
        a.add(k)
        b.add(k)
    for k in (0, 7, 19, 3):
        a.remove(k)
        b.remove(k)
    np.testing.assert_array_equal(a.sample(64), b.sample(64))
    assert a._index_to_key == b._index_to_key
    with pytest.raises(AssertionError):
        a.remove(0)


def test_prefetched_draws_stay_on_the_reference_stream():
    """sample_device pre-draws 64 batches once the key count is stable; when the count changes with unused
    batches left, the generator is rewound and re-advanced -- every batch must equal the oracle's (= the
    reference's one ``integers(len, size)`` call per sample)."""
    from slimdqn.sample_collection.samplers import UniformSamplingDistribution

    a, b = UniformSamplingDistribution(11, device="cpu"), OracleUniform(11)
    for k in range(50):
        a.add(k)
        b.add(k)
    nxt = 50
    rng = np.random.default_rng(0)
    for step in range(400):
        if step % 2 == 0:
            got = np.asarray([a._index_to_key[i] for i in a.sample_device(16).numpy()], dtype=np.int32)
        else:
            got = a.sample(16)
        np.testing.assert_array_equal(got, b.sample(16))
        r = rng.random()
        if r < 0.05:  # grow
            a.add(nxt)
            b.add(nxt)
            nxt += 1
        elif r < 0.10:  # FIFO-style replace: length unchanged
            a.add(nxt)
            b.add(nxt)
            victim = a._index_to_key[0]
            a.remove(victim)
            b.remove(victim)
            nxt += 1
        elif r < 0.12:  # different batch size once in a while
            np.testing.assert_array_equal(a.sample(5), b.sample(5))
    assert a._pf is not None and a._pf["n"] in (1, 64)


def test_streams_keep_their_own_trajectories():
    """Vectorised collection (`add(..., stream=i)`): the elements of each stream are exactly those a buffer fed by that stream
    alone produces (frame stacks and n-step returns never mix environments), interleaved in arrival order."""
    from slimdqn.sample_collection.replay_buffer import ReplayBuffer, TransitionElement
    from slimdqn.sample_collection.samplers import UniformSamplingDistribution

    n_streams, n, stack = 3, 3, 4
    rb = ReplayBuffer(UniformSamplingDistribution(0, device="cpu"), 8, 10_000, stack_size=stack, update_horizon=n, gamma=0.9, device="cpu")
    alone = [OracleRB(OracleUniform(0), 8, 10_000, stack_size=stack, update_horizon=n, gamma=0.9) for _ in range(n_streams)]
    rng = np.random.default_rng(5)
    origin = []  # element key of the shared buffer -> (stream, key in that stream's own buffer)
    for t in range(200):
        for i in range(n_streams):
            obs = rng.integers(0, 256, (6, 5), dtype=np.uint8)
            action, reward = int(rng.integers(0, 5)), float(rng.normal())
            terminal = bool(rng.random() < 0.08)
            end = terminal or bool(rng.random() < 0.05)
            before = alone[i].add_count
            rb.add(TransitionElement(obs, action, reward, terminal, end), stream=i)
            alone[i].add(OT(obs, action, reward, terminal, end))
            origin += [(i, k) for k in range(before, alone[i].add_count)]
    assert rb.add_count == len(origin) == sum(o.add_count for o in alone)
    for key, (i, k) in enumerate(origin):
        a, b = rb._memory[key], alone[i]._memory[k]
        np.testing.assert_array_equal(a.state, b.state)
        np.testing.assert_array_equal(a.next_state, b.next_state)
        assert (a.action, a.reward, bool(a.is_terminal)) == (b.action, b.reward, bool(b.is_terminal))
    held = np.zeros_like(rb._refcount)
    for key in rb._memory.keys():
        for f in rb._h_elem_frames[key % 10_000]:
            if f >= 0:
                held[f] += 1
    for traj in rb._trajectories.values():
        for entry in traj:
            held[entry[0]] += 1
    np.testing.assert_array_equal(held, rb._refcount)
