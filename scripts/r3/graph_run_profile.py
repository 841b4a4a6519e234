"""Host time of the parts of GraphedUpdate.run() (one-step graph) against the eager update, cnn agent."""
import sys, time
sys.path.insert(0, "is-dqn_amd")
import numpy as np, torch
from slimdqn.networks.isdqn import iSDQN
from slimdqn.sample_collection.replay_buffer import ReplayBuffer, TransitionElement
from slimdqn.sample_collection.samplers import UniformSamplingDistribution

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
agent = iSDQN(0, (84, 84, 4), 9, 9, [32, 64, 64, 512], True, False, "cnn", 6.25e-5, 0.99, 1, 1, 10**9, adam_eps=1.5e-4, batch_size=B, use_graph=True)
rb = ReplayBuffer(UniformSamplingDistribution(1), B, 4096)
rng = np.random.default_rng(0)
for t in range(600):
    rb.add(TransitionElement(rng.integers(0, 256, (84, 84), dtype=np.uint8), int(rng.integers(0, 9)), 0.0, t % 97 == 96, t % 97 == 96))
for _ in range(20):
    agent.update_online_params(1, rb)
g = agent._graphed
torch.cuda.synchronize()
N = 300
acc = dict(flush=0.0, draw=0.0, copy=0.0, refresh=0.0, replay=0.0, sync=0.0)
for _ in range(N):
    t0 = time.perf_counter(); rb._flush()
    t1 = time.perf_counter(); rows = rb._sampling_distribution.draw_rows_device(g.S, g.B)
    t2 = time.perf_counter(); g.block.copy_(rows, non_blocking=True)
    t3 = time.perf_counter(); g.eng.refresh_mirror()
    t4 = time.perf_counter(); g.graph.replay(); g.eng._mirror_made_current()
    t5 = time.perf_counter(); torch.cuda.synchronize()
    t6 = time.perf_counter()
    for k, d in zip(acc, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5)):
        acc[k] += d
print("B", B, {k: round(v / N * 1e6, 1) for k, v in acc.items()}, "us per call (with a sync after every replay)")
# back to back, no sync
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(N):
    g.run()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("back to back: issue %.1f us/step, total %.1f us/step" % ((t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6))
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for _ in range(N):
    g.graph.replay()
ev1.record(); torch.cuda.synchronize()
print("replay only, device time %.1f us/step" % (ev0.elapsed_time(ev1) / N * 1e3))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(N):
    agent.update_online_params(1, rb)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("agent.update_online_params: issue %.1f us/step, total %.1f us/step" % ((t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(100):
    agent.update_online_params(1, rb)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
