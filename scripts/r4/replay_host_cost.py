"""Host cost of one graph replay of S steps (the driver's 20-step form exposes the first replay's launch): draw + stage + hipGraphLaunch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "is-dqn_amd"))
import torch, bench

for S in (int(a) for a in (sys.argv[1:] or ["5"])):
    rep = bench.Replica("c2", 200000, "bf16x3", 0, "cuda:0")
    rep.enable_graph(S)
    g = rep.graphed
    for _ in range(200 // S + 2):
        g.run()
    torch.cuda.synchronize()
    sampler = rep.rb._sampling_distribution

    def timed(fn, n=20, idle=True):
        ts = []
        for _ in range(n):
            if idle:
                torch.cuda.synchronize()
            t = time.perf_counter()
            fn()
            ts.append((time.perf_counter() - t) * 1e6)
        torch.cuda.synchronize()
        ts.sort()
        return ts[len(ts) // 2]

    rows = sampler.draw_rows_device(S, g.B)
    print(f"S={S}: draw_rows_device {timed(lambda: sampler.draw_rows_device(S, g.B)):.0f} us  block.copy_ {timed(lambda: g.block.copy_(rows, non_blocking=True)):.0f} us  "
          f"graph.replay (idle GPU) {timed(g.graph.replay):.0f} us  graph.replay (busy GPU) {timed(g.graph.replay, idle=False):.0f} us  run() idle {timed(g.run):.0f} us", flush=True)
    # first-replay exposure: sync, then time until the replay has finished on the device
    def to_done():
        g.run(); torch.cuda.synchronize()
    print(f"      run() + synchronize from an idle GPU {timed(to_done):.0f} us  (device time of a replay ~ {S * 0.238 * 1e3:.0f} us)", flush=True)
    g.destroy()
    del rep
