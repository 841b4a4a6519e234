"""GPU-side time of EAGERLY enqueued steps when the host is not the bottleneck: the stream is held back by a long sleep kernel while
N steps are enqueued, then released.  Compare with the replayed graph (bench.py: ~247 us / step)."""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "is-dqn_amd")
import torch
from bench import Replica

rep = Replica("c2", 200_000, "bf16x3", 0, "cuda:0")
for _ in range(300):
    rep.step()
torch.cuda.synchronize()
N = 120
batches = []
for _ in range(N):  # (sample first: the gather is not what is measured)
    b = rep.rb.sample()
    batches.append(rep.eng.make_batch(frames=b.frames, frame_stride=b.frame_stride, frame_ids=b.frame_ids, action=b.action, reward=b.reward,
                                      terminal=b.is_terminal, mirror_current=True))
rep.eng.refresh_mirror()
torch.cuda.synchronize()
for trial in range(3):
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(int(2.1e9 * 0.08))  # ~80 ms at 2.1 GHz
    t0 = time.perf_counter()
    ev0.record()
    for cb in batches:
        rep.eng.learn_on_batch(cb)
    ev1.record()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print("trial %d: host enqueue %.1f us/step, GPU %.1f us/step" % (trial, (t1 - t0) / N * 1e6, ev0.elapsed_time(ev1) / N * 1e3), flush=True)
