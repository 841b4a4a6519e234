"""Driver's form (20 timed steps behind >= 2000 untimed ones): steps per replay x mirror ownership x one drained replay in front."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "is-dqn_amd"))
import torch, bench


def region(one, S, drained):
    for _ in range(2000 // S):
        one()
    torch.cuda.synchronize()
    if drained:
        one()
        torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for i in range(20 // S):
        one()
    ev1.record()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    return 20 / el, ev0.elapsed_time(ev1)


for trust in (False, True):
    for S in (5, 10, 20, 4, 2):
        rep = bench.Replica("c2", 1_000_000, "bf16x3", 0, "cuda:0", trust_mirror=trust)
        rep.enable_graph(S)
        one = rep.graphed.run
        out = []
        for drained in (False, True, False, True, False, True):
            v, e = region(one, S, drained)
            out.append(f"{'drained' if drained else 'plain'} {v:.0f}")
        print(f"trust_mirror={trust} S={S:2d}: " + "  ".join(out), flush=True)
        rep.graphed.destroy()
        del rep, one
        torch.cuda.empty_cache()
