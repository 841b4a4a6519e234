"""Hunt for rare run-to-run differences: many single learn steps from identical state; for every repeat that differs
from the first, report which regions differ (listed in dataflow order, so the first one names the kernel)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "is-dqn_amd"))
import numpy as np, torch
from tests.gpu_helpers import make_frame_batch, device_batch
from slimdqn._engine import QNetEngine
B, K, A = int(os.environ.get("B", "256")), int(os.environ.get("K", "9")), int(os.environ.get("A", "9"))
frames, ids, action, reward, terminal, _ = make_frame_batch(B, A, seed=5)
names = ["act/Conv_0", "z/Conv_0", "act/Conv_1", "z/Conv_1", "act/Conv_2", "z/Conv_2", "act/Dense_0", "z/Dense_0", "dout", "dz/Dense_0", "part/Dense_0",
         "dz/Conv_2", "part/Conv_2", "dz/Conv_1", "part/Conv_1", "dz/Conv_0", "part/Conv_0", "gw/Dense_1", "gw/Conv_2", "gw/Conv_1", "gw/Conv_0", "red/Conv_2", "red/Conv_1", "red/Conv_0"]
R = int(os.environ.get("REPS", "80"))
eng = QNetEngine((84, 84, 4), A, 1 + K, (32, 64, 64, 512), 'cnn', True, B, gamma_n=0.99, learning_rate=6.25e-5, adam_eps=1.5e-4)
b = device_batch(eng, frames, ids, action, reward, terminal)
p0 = None
ref = None; events = 0
for r in range(R):
    eng.init_params(1)
    eng.adam_m.zero_(); eng.adam_v.zero_(); eng.adam_count.zero_()
    eng.learn_on_batch(b); torch.cuda.synchronize()
    cur = {}
    for n in names:
        try: cur[n] = eng.region(n).clone()
        except Exception: pass
    cur["q_values"] = eng.q_values.clone(); cur["params"] = eng.params.clone(); cur["adam_m"] = eng.adam_m.clone()
    if ref is None: ref = cur; continue
    diff = [(n, int((cur[n] != ref[n]).sum().item())) for n in cur if not torch.equal(cur[n], ref[n])]
    if diff:
        events += 1
        print(f"repeat {r}: " + ", ".join(f"{n}:{c}" for n, c in diff))
print(f"{events} differing repeats out of {R-1}")
