"""End-to-end rate of the drop-in trainer on the synthetic environment at the headline network (B=256, K=9, -utd 4):
environment steps/s and gradient steps/s for the reference's single-environment loop and for vectorised environments
in this process / on host worker processes.  Writes gpurun_out/trainer_rate.json.

    python scripts/r3/trainer_rate.py [--steps 12000]
"""
import argparse
import json
import os
import shutil
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "is-dqn_amd"))


def one(n_envs, n_workers, steps, tag):
    from experiments.atari.isdqn import run

    root = f"/tmp/trainer_rate_{tag}"
    shutil.rmtree(root, ignore_errors=True)
    nis = 2000
    argv = (f"-en rate_Synthetic -s 1 -dw -f 32 64 64 512 -at cnn -ln -nbi 9 -rbc 50000 -bs 256 -utd 4 -nis {nis} -ed 4000 -tuf 8000 "
            f"-horizon 1000 -ne 1 -ntspe {steps} -env synthetic -nenvs {n_envs} -nworkers {n_workers}").split()
    # gradient steps are counted where they are issued (every path ends in GraphedUpdate.run: S steps per replay)
    import torch
    from slimdqn import _graph

    stats = dict(steps=0, t_first=None, t_last=None, steps_at_first=0)
    plain_run = _graph.GraphedUpdate.run

    def counted(self):
        plain_run(self)
        now = time.perf_counter()
        stats["steps"] += self.S
        if stats["t_first"] is None and stats["steps"] >= 200:  # (past capture and warm-up)
            stats["t_first"], stats["steps_at_first"] = now, stats["steps"]
        stats["t_last"] = now

    _graph.GraphedUpdate.run = counted
    t0 = time.perf_counter()
    try:
        gathered = run(argv, root=root)
        torch.cuda.synchronize()
    finally:
        _graph.GraphedUpdate.run = plain_run
    wall = time.perf_counter() - t0
    n_steps, rate = float(gathered[0][0][2]), float(gathered[0][0][3])
    grad_rate = (stats["steps"] - stats["steps_at_first"]) / max(stats["t_last"] - stats["t_first"], 1e-9)
    return dict(n_envs=n_envs, n_workers=n_workers, env_steps=n_steps, epoch_env_steps_per_s=rate, gradient_steps=stats["steps"],
                gradient_steps_per_s=grad_rate, env_steps_per_s_while_learning=4 * grad_rate, wall_s=wall)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=12000)
    ap.add_argument("--configs", default="1:0,8:0,32:0,32:8,64:8,64:16,128:16")
    args = ap.parse_args()
    out = []
    for c in args.configs.split(","):
        n, w = (int(x) for x in c.split(":"))
        r = one(n, w, args.steps if n == 1 else args.steps * 4, f"{n}_{w}")
        print(json.dumps(r), flush=True)
        out.append(r)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump({"what": "synthetic-environment trainer, headline network, -utd 4: epoch-level rates (epoch = all steps incl. the 2000 initial samples)",
               "os_cpu_count": os.cpu_count(), "runs": out}, open(os.path.join(ROOT, "gpurun_out", "trainer_rate.json"), "w"), indent=1)
