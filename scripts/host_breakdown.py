"""Where does the host spend its time per step?  (python sampling / ctypes call / C launches)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "is-dqn_amd"))
import torch, bench
rep = bench.Replica("c2", 200000, "bf16x3", 0, "cuda:0")
for _ in range(50): rep.step()
torch.cuda.synchronize()
N = 300
def timeit(fn):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(N): fn()
    dt = time.perf_counter() - t; torch.cuda.synchronize(); return dt / N * 1e6
batch = rep.rb.sample()
cb = rep.eng.make_batch(frames=batch.frames, frame_stride=batch.frame_stride, frame_ids=batch.frame_ids, action=batch.action, reward=batch.reward, terminal=batch.is_terminal)
print("sample()            us", timeit(lambda: rep.rb.sample()))
print("make_batch          us", timeit(lambda: rep.eng.make_batch(frames=batch.frames, frame_stride=batch.frame_stride, frame_ids=batch.frame_ids, action=batch.action, reward=batch.reward, terminal=batch.is_terminal)))
print("learn_on_batch      us", timeit(lambda: rep.eng.learn_on_batch(cb)))
print("full step           us", timeit(rep.step))
