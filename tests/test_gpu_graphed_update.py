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
