"""Launcher: one training process per GPU, each an independent (game, seed) replica.

    python experiments/launch.py --gpus 8 --games Asterix Breakout ... --first_seed 1 --n_seeds 1 -- \
        -en L2_K9_LN1_cnn -f 32 64 64 512 -at cnn -ln -nbi 9 -bs 256 ...

The MI355X-native counterpart of the reference's seed fan-out (launch_job/atari/normal/train.sh:12-16: N background
`python3 experiments/atari/isdqn.py --seed $seed` processes sharing one GPU): here every child gets its OWN GPU
(HIP_VISIBLE_DEVICES=<i>, so it sees exactly one device), RANK / WORLD_SIZE / MASTER_* for the per-epoch metric
all_gather, and `--experiment_name <name>_<Game> --seed <seed>` from `rank_assignment`.  The launcher itself never touches
a GPU (it must not: children are started with a narrowed device list).  Exit status: the first non-zero child status.

`--module` runs another entry point per rank with the same environment (tests: a CPU-only worker with --backend gloo).
"""
import argparse
import os
import socket
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_PKG = os.path.dirname(_HERE)
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)

from experiments.base.dist import rank_assignment  # noqa: E402  (no torch import)


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def child_env(rank: int, world: int, port: int, backend: str, gpu, base=None) -> dict:
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0" if gpu is not None else str(rank), MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(port), ISDQN_DIST_BACKEND=backend)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL fails without it on this driver
    if gpu is not None:
        env["HIP_VISIBLE_DEVICES"] = str(gpu)
        env["ISDQN_DEVICE_INDEX"] = "0"
    return env


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--gpus", type=int, default=1, help="processes to start = GPUs used (GPU i -> rank i)")
    ap.add_argument("--games", nargs="+", default=["Asterix"])
    ap.add_argument("--first_seed", type=int, default=None)
    ap.add_argument("--n_seeds", type=int, default=None, help="seeds per game (default 1)")
    ap.add_argument("--last_seed", type=int, default=None, help="the reference launchers' spelling: n_seeds = last_seed - first_seed + 1")
    ap.add_argument("--algo", default="isdqn")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--no-gpu-pinning", action="store_true", help="do not set HIP_VISIBLE_DEVICES (CPU-only / gloo runs)")
    ap.add_argument("--module", default=None, help="python file to run per rank instead of experiments/atari/<algo>.py")
    ap.add_argument("--log_dir", default=None, help="per-rank stdout/stderr files (default: inherit)")
    ap.add_argument("rest", nargs=argparse.REMAINDER, help="-- followed by the trainer's own flags (-en without the game suffix)")
    args = ap.parse_args(argv)
    rest = args.rest[1:] if args.rest[:1] == ["--"] else args.rest
    # The checks of the reference's shell launchers (launch_job/*/normal/local_*.sh via parse_arguments.sh; tests/test_launch_job.py:4-37),
    # before any process is started: an experiment name, a first seed, a non-empty seed range.
    def refuse(msg):
        print(f"[launch] {msg}", file=sys.stderr, flush=True)
        return 2

    if args.module is None and not any(f in rest for f in ("-en", "--experiment_name")):
        return refuse("the experiment name is not specified (-- -en <name> ...)")
    if args.first_seed is None:
        return refuse("the first seed is not specified (--first_seed)")
    if args.last_seed is not None and args.n_seeds is not None and args.n_seeds != args.last_seed - args.first_seed + 1:
        return refuse("--n_seeds and --last_seed disagree")
    if args.last_seed is not None:
        if args.last_seed < args.first_seed:
            return refuse("the last seed should be greater than or equal to the first seed")
        args.n_seeds = args.last_seed - args.first_seed + 1
    if args.n_seeds is None:
        args.n_seeds = 1
    if args.n_seeds < 1 or args.gpus < 1:
        return refuse("--n_seeds and --gpus must be at least 1")
    port = free_port()
    entry = args.module or os.path.join(_HERE, "atari", f"{args.algo}.py")
    procs = []
    for rank in range(args.gpus):
        game, seed = rank_assignment(rank, args.games, args.first_seed, args.n_seeds)
        cmd = [sys.executable, entry]
        if args.module is None:
            child_args = list(rest)
            for flag in ("-en", "--experiment_name"):
                if flag in child_args:
                    i = child_args.index(flag)
                    child_args[i + 1] = f"{child_args[i + 1]}_{game}"
            cmd += child_args + ["--seed", str(seed)]
        else:
            cmd += list(rest)
        env = child_env(rank, args.gpus, port, args.backend, None if args.no_gpu_pinning else rank)
        env["ISDQN_GAME"], env["ISDQN_SEED"] = game, str(seed)
        out = None
        if args.log_dir:
            os.makedirs(args.log_dir, exist_ok=True)
            out = open(os.path.join(args.log_dir, f"rank_{rank}_{game}_{seed}.out"), "w")
        procs.append((subprocess.Popen(cmd, env=env, stdout=out, stderr=subprocess.STDOUT if out else None), out))
    status = 0
    for p, out in procs:
        rc = p.wait()
        if out:
            out.close()
        if rc and not status:
            status = rc
    return status


if __name__ == "__main__":
    sys.exit(main())
