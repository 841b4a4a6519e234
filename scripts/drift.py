"""Per-100-step device time over a long run: does the step time drift (clocks) or is it flat?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "is-dqn_amd"))
import torch, bench
rep = bench.Replica("c2", 1000000, sys.argv[1] if len(sys.argv) > 1 else "bf16x3", 0, "cuda:0")
for _ in range(100): rep.step()
torch.cuda.synchronize()
evs = [torch.cuda.Event(enable_timing=True) for _ in range(31)]
evs[0].record()
for b in range(30):
    for _ in range(100): rep.step()
    evs[b + 1].record()
torch.cuda.synchronize()
print("us/step per block:", [round(evs[i].elapsed_time(evs[i + 1]) * 10, 1) for i in range(30)])
