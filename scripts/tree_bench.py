import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "is-dqn_amd"))
import numpy as np, torch
from slimdqn.sample_collection.sum_tree import SumTree
t = SumTree(1_000_000)
rng = np.random.default_rng(0)
for s in range(0, 1_000_000, 4096):
    n = min(4096, 1_000_000 - s)
    t.set_device(torch.arange(s, s + n, dtype=torch.int32, device="cuda"), torch.from_numpy(rng.uniform(0.1, 2, n)).cuda())
torch.cuda.synchronize()
idx = torch.from_numpy(rng.integers(0, 1_000_000, 256).astype(np.int32)).cuda()
val = torch.from_numpy(rng.uniform(0.1, 2, 256)).cuda()
u = torch.from_numpy(rng.random(256)).cuda()
out = torch.empty(256, dtype=torch.int32, device="cuda")
def timeit(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
print("tree_set   B=256 us", timeit(lambda: t.set_device(idx, val)))
print("tree_query B=256 us", timeit(lambda: t.query_device(u, out=out, unit=True)))
