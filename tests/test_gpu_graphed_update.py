"""The hipGraph-replayed training step (slimdqn/_graph.py: what bench.py times) against the eager loop
`rb.sample() -> agent.learn_on_batch() -> [rb.update priorities]` (replay_buffer.py:sample, isdqn.py:49-66,
samplers.py:update): same sampler seed, so the same index draws; parameters, Adam moments and the sum tree must
come out bit-identical after the same number of steps."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _replica(workload, seed=3, capacity=4096):
    from bench import Replica

    return Replica(workload, capacity, "bf16x3", seed, "cuda:0")


@pytest.mark.parametrize("workload", ["c2", "c3"])  # uniform (one gather per replay), prioritized (one per step)
def test_graph_replay_equals_eager_steps(workload):
    S, n_replays = 4, 3
    eager, graphed = _replica(workload), _replica(workload)
    assert torch.equal(eager.eng.params, graphed.eng.params)
    graphed.enable_graph(S)
    for _ in range(S * n_replays):
        eager.step()
    for _ in range(n_replays):
        graphed.graphed.run()
    torch.cuda.synchronize()
    for name in ("params", "adam_m", "adam_v", "adam_count", "losses_accum"):
        a, b = getattr(eager.eng, name), getattr(graphed.eng, name)
        assert torch.equal(a, b), f"{name}: {(a != b).sum().item()} elements differ between eager and graph replay"
    if eager.w["prioritized"]:
        ta = eager.rb._sampling_distribution._sum_tree._nodes_dev
        tb = graphed.rb._sampling_distribution._sum_tree._nodes_dev
        assert torch.equal(ta, tb), "sum tree differs between eager and graph replay"
    # and the step really trained
    fresh = _replica(workload)
    assert not torch.equal(fresh.eng.params, graphed.eng.params)


def test_trusted_graph_only_while_the_mirror_is_current():
    """The opt-in trusted mode (engine.trust_mirror = True: the owner declares that every parameter write goes through torch or the
    engine): every captured step takes the weight mirror as it is and GraphedUpdate.run() rebuilds it in front of the replay only when
    the engine's bookkeeping (_engine.py: _mirror_is_current) says something wrote the parameters since the last replay.  A torch-side
    write and a head shift between replays must both be seen by the next replay."""
    S = 3
    eager, graphed = _replica("c2"), _replica("c2")
    graphed.eng.trust_mirror = True
    graphed.enable_graph(S)
    g = graphed.graphed
    used = []
    for replay in range(5):
        if replay == 2:  # a writer the version counter sees
            for r in (eager, graphed):
                r.eng.params.mul_(1.0009765625)
        if replay == 4:  # a writer inside the library
            for r in (eager, graphed):
                r.eng.shift_params()
        used.append(graphed.eng._mirror_is_current(None))
        for _ in range(S):
            eager.step()
        g.run()
    torch.cuda.synchronize()
    assert used == [False, True, False, True, False]
    for name in ("params", "adam_m", "adam_v", "adam_count", "losses_accum"):
        a, b = getattr(eager.eng, name), getattr(graphed.eng, name)
        assert torch.equal(a, b), f"{name}: {(a != b).sum().item()} elements differ between eager and graph replay"


def test_default_replay_sees_writes_the_version_counter_cannot():
    """Default (engine.trust_mirror False): a replay rebuilds the weight mirror at its head, so parameters written behind torch's
    back -- through `.data` (its own version counter) and through a raw device pointer (hipMemcpy) -- are the ones the next replay
    trains on: bit-identical with the eager loop, which rebuilds in every call."""
    import ctypes
    import os

    hip_rt = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))  # the runtime torch itself loaded
    S = 2
    eager, graphed = _replica("c2"), _replica("c2")
    assert not graphed.eng.trust_mirror
    graphed.enable_graph(S)
    g = graphed.graphed
    for replay in range(4):
        if replay == 1:
            for r in (eager, graphed):
                v = r.eng.params._version
                r.eng.params.data.mul_(1.0009765625)  # `.data` shares the storage, not the version counter
                assert r.eng.params._version == v
        if replay == 2:
            for r in (eager, graphed):
                v = r.eng.params._version
                host = (r.eng.params[:4096] * 0.5).cpu().contiguous()
                torch.cuda.synchronize()
                rc = hip_rt.hipMemcpy(ctypes.c_void_p(r.eng.params.data_ptr()), ctypes.c_void_p(host.data_ptr()),
                                      ctypes.c_size_t(host.numel() * 4), 1)  # hipMemcpyHostToDevice through the raw pointer
                assert rc == 0 and r.eng.params._version == v
        for _ in range(S):
            eager.step()
        g.run()
    torch.cuda.synchronize()
    for name in ("params", "adam_m", "adam_v", "adam_count", "losses_accum"):
        a, b = getattr(eager.eng, name), getattr(graphed.eng, name)
        assert torch.equal(a, b), f"{name}: {(a != b).sum().item()} elements differ between eager and graph replay"


@pytest.mark.parametrize("prioritized", [False, True])
def test_agent_update_online_params_graphed_equals_eager(prioritized):
    """The drop-in trainer gets the captured step: `iSDQN.update_online_params` on a device replay replays a one-step
    hipGraph (isdqn.py:55-62).  Two agents, one with use_graph=False, are fed the same environment stream through a
    buffer that grows, fills up and evicts (capacity 48 < 90 adds); with the prioritized sampler new elements enter at
    the recorded maximum (staged leaf writes, resolved on the device) and TD errors are written back.  Parameters, Adam
    state, accumulated losses and the sum tree must be bit-identical after every update."""
    from slimdqn.networks.isdqn import iSDQN
    from slimdqn.sample_collection.replay_buffer import ReplayBuffer, TransitionElement
    from slimdqn.sample_collection.samplers import PrioritizedSamplingDistribution, UniformSamplingDistribution

    K, A, B, C = 3, 5, 8, 48

    def make(use_graph):
        agent = iSDQN(0, (84, 84, 4), A, K, [8, 12, 16, 24], True, False, "cnn", 2e-4, 0.99, 3, 2, 6, adam_eps=1.5e-4,
                      batch_size=B, use_graph=use_graph)
        sampler = PrioritizedSamplingDistribution(5, C) if prioritized else UniformSamplingDistribution(5)
        rb = ReplayBuffer(sampler, B, C, update_horizon=3, gamma=0.99)
        if prioritized:
            agent.priority_writeback = True
        return agent, rb

    (eager, rb_e), (graphed, rb_g) = make(False), make(True)
    assert torch.equal(eager._engine.params, graphed._engine.params)
    rng = np.random.default_rng(0)
    n_updates = 0
    for step in range(1, 91):
        obs = rng.integers(0, 256, (84, 84), dtype=np.uint8)
        a, r, term = int(rng.integers(0, A)), float(rng.choice([-1.0, 0.0, 1.0])), bool(rng.random() < 0.08)
        for rb in (rb_e, rb_g):
            kw = dict(priority=rb._sampling_distribution.MAX_PRIORITY) if prioritized else {}
            rb.add(TransitionElement(obs, a, r, term, term), **kw)
        if step > 14:
            for agent, rb in ((eager, rb_e), (graphed, rb_g)):
                agent.update_online_params(step, rb)
                agent.update_target_params(step)
            if step % 2 == 0:
                n_updates += 1
                for name in ("params", "adam_m", "adam_v", "adam_count", "losses_accum"):
                    x, y = getattr(eager._engine, name), getattr(graphed._engine, name)
                    assert torch.equal(x, y), f"step {step}: {name} differs between the eager and the captured step"
                if prioritized:
                    ta, tb = rb_e._sampling_distribution._sum_tree, rb_g._sampling_distribution._sum_tree
                    assert torch.equal(ta._nodes_dev, tb._nodes_dev) and torch.equal(ta._max_dev, tb._max_dev)
    assert n_updates >= 30 and graphed._graphed is not None and eager._graphed is None
    if prioritized:
        rb_g._sampling_distribution._sum_tree.check_status()
        assert rb_g._sampling_distribution._sum_tree.max_recorded_priority >= 1.0


@pytest.mark.parametrize("arch, batch_norm", [("cnn", True), ("impala", False), ("impala", True), ("fc", False), ("fc", True)])
def test_captured_step_equals_eager_on_the_other_learn_paths(arch, batch_norm):
    """Steps 2..S of a captured replay take the weight mirror as the previous step's optimizer left it (slimdqn/_graph.py), and a
    trusted engine (trust_mirror, set here) replays WITHOUT rebuilding whenever its bookkeeping says the previous call left it current,
    so the optimizer launches of the BatchNorm path,
    of the impala torso and of the all-dense plan have to leave the mirror equal to the parameters they wrote; the fc step also
    materialises its float32 observation rows inside the graph.  Two agents, one with use_graph=False, same stream: bit-identical after
    every update."""
    from slimdqn.networks.isdqn import iSDQN
    from slimdqn.sample_collection.replay_buffer import ReplayBuffer, TransitionElement
    from slimdqn.sample_collection.samplers import UniformSamplingDistribution

    K, A, B, C = 2, 4, 8, 40
    obs, feats = ((8,), [24, 40]) if arch == "fc" else ((36, 36, 2), [8, 16, 8, 24])

    def make(use_graph):
        agent = iSDQN(0, obs, A, K, feats, True, batch_norm, arch, 2e-4, 0.99, 1, 1, 5, adam_eps=1.5e-4, batch_size=B, use_graph=use_graph)
        agent._engine.trust_mirror = True
        rb = ReplayBuffer(UniformSamplingDistribution(5), B, C, stack_size=(1 if arch == "fc" else obs[2]), update_horizon=1, gamma=0.99)
        return agent, rb

    (eager, rb_e), (graphed, rb_g) = make(False), make(True)
    assert torch.equal(eager._engine.params, graphed._engine.params)
    rng = np.random.default_rng(0)
    for step in range(1, 41):
        o = rng.normal(size=obs).astype(np.float32) if arch == "fc" else rng.integers(0, 256, obs[:2], dtype=np.uint8)
        a, r, term = int(rng.integers(0, A)), float(rng.choice([-1.0, 0.0, 1.0])), bool(rng.random() < 0.08)
        for rb in (rb_e, rb_g):
            rb.add(TransitionElement(o, a, r, term, term))
        if step > 14:
            for agent, rb in ((eager, rb_e), (graphed, rb_g)):
                agent.update_online_params(step, rb)
                agent.update_target_params(step)  # (every 5th step: the head shift invalidates the mirror)
            for name in ("params", "adam_m", "adam_v", "adam_count", "losses_accum"):
                x, y = getattr(eager._engine, name), getattr(graphed._engine, name)
                assert torch.equal(x, y), f"step {step}: {name} differs between the eager and the captured step"
    assert eager._graphed is None and graphed._graphed is not None


def test_priorities_ready_event_orders_a_second_stream():
    """isdqn_batch.priorities_ready (include/isdqn_hip.h): the learn call records the caller's event once q_values / targets /
    priorities are final.  A second stream that only waits for that event must read the same priorities as a full
    synchronisation gives, while the call's own tail (backward, Adam) is still free to run."""
    rep = _replica("c3")
    eng, rb = rep.eng, rep.rb
    batch = rb.sample()
    ev = torch.cuda.Event()
    side = torch.cuda.Stream()
    ev.record(side)  # creates the handle
    c = eng.make_batch(frames=batch.frames, frame_stride=batch.frame_stride, frame_ids=batch.frame_ids, action=batch.action,
                       reward=batch.reward, terminal=batch.is_terminal, priorities_ready=ev)
    eng.priorities.fill_(-1.0)
    torch.cuda.synchronize()
    eng.learn_on_batch(c)
    side.wait_event(ev)
    with torch.cuda.stream(side):
        early = eng.priorities.clone()
        q_early = eng.q_values.clone()
    torch.cuda.synchronize()
    assert torch.equal(early, eng.priorities) and torch.equal(q_early, eng.q_values)
    assert bool((early >= 0).all()) and early.dtype == torch.float64
