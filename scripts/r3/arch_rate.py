"""Learn-step rate of the torsos the reference's timing sweep runs (launch_job/atari/launch_time.sh:13-27: cnn and impala, features
32 64 64 512, LayerNorm, K in 1 4 9 49 at batch 32) -- eager launches of isdqn_net_learn_on_batch on a fixed synthetic batch, HIP events.
usage: python scripts/r3/arch_rate.py [--batch 32] [--K 9] [--bn]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "is-dqn_amd"))
import numpy as np
import torch

from slimdqn._engine import QNetEngine


def rate(arch, B, K, bn, steps=200, warm=50):
    A = 9
    eng = QNetEngine((84, 84, 4), A, 1 + K, (32, 64, 64, 512), arch, True, B, gamma_n=0.99, learning_rate=6.25e-5, adam_eps=1.5e-4, batch_norm=bn)
    eng.init_params(0)
    rng = np.random.default_rng(0)
    frames = torch.from_numpy(rng.integers(0, 256, (3 * B + 8, 84 * 84), dtype=np.uint8)).cuda()
    ids = torch.from_numpy(rng.integers(0, 3 * B + 8, (B, 8)).astype(np.int32)).cuda()
    batch = eng.make_batch(frames=frames, frame_stride=84 * 84, frame_ids=ids, action=torch.from_numpy(rng.integers(0, A, B).astype(np.int32)).cuda(),
                           reward=torch.from_numpy(rng.normal(size=B).astype(np.float32)).cuda(),
                           terminal=torch.from_numpy((rng.random(B) < 0.01).astype(np.uint8)).cuda())
    for _ in range(warm):
        eng.learn_on_batch(batch)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        eng.learn_on_batch(batch)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    return {"arch": arch, "batch": B, "K": K, "batch_norm": bn, "ms_per_step": round(ms, 4), "gradient_steps_per_s": round(1e3 / ms, 1),
            "launch": "eager, fixed batch", "finite": bool(torch.isfinite(eng.losses).all().item())}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, nargs="+", default=[32, 256])
    ap.add_argument("--K", type=int, nargs="+", default=[9])
    ap.add_argument("--arch", nargs="+", default=["cnn", "impala"])
    ap.add_argument("--bn", type=int, nargs="+", default=[0, 1])
    ap.add_argument("--steps", type=int, default=200)
    args = ap.parse_args()
    for B in args.batch:
        for K in args.K:
            for arch in args.arch:
                for bn in args.bn:
                    print(json.dumps(rate(arch, B, K, bool(bn), steps=args.steps, warm=max(5, args.steps // 4))), flush=True)
