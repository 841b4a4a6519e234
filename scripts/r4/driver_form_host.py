"""The driver's form (20 timed steps = 4 replays of 5 behind 2000 untimed steps): where do the timed region's 5.4 - 5.6 ms go when the
steady state needs 4.86?  Host time of each of the four launches, event time, wall time; variants of what the host does in front."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "is-dqn_amd"))
import torch, bench

cap = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
rep = bench.Replica("c2", cap, "bf16x3", 0, "cuda:0")
S = 5
rep.enable_graph(S)
one = rep.graphed.run


def region(tag, before=None, untimed=400):
    for _ in range(untimed):
        one()
    torch.cuda.synchronize()
    if before is not None:
        before()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    ts = []
    for i in range(4):
        t = time.perf_counter()
        one()
        ts.append((time.perf_counter() - t) * 1e6)
    ev1.record()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f"{tag:34s} wall {el * 1e3:.3f} ms ({20 / el:.0f} steps/s)  events {ev0.elapsed_time(ev1):.3f} ms  issue {t_issue * 1e3:.3f} ms  "
          f"per launch us {' '.join('%.0f' % x for x in ts)}", flush=True)


def spin(ms):
    t = time.perf_counter()
    while time.perf_counter() - t < ms * 1e-3:
        pass


for r in range(3):
    region("as bench.py")
    region("1 ms host spin in front", lambda: spin(1.0))
    region("20 ms host spin in front", lambda: spin(20.0))
    region("one extra replay + sync in front", lambda: (one(), torch.cuda.synchronize()))
    region("short untimed phase (8 replays)", untimed=8)
