"""One process per GPU: rank bookkeeping for independent (game, seed) replicas.

The reference fans seeds out as independent OS processes (launch_job/atari/normal/train.sh:12-16,
local_isdqn.sh:18-23); nothing is exchanged during training.  Here every rank drives one GPU, and the only traffic is
one small all_gather of per-epoch scalars (RCCL over xGMI with backend "nccl"; "gloo" on CPU-only boxes and in the tests),
after which rank 0 writes the aggregate.  `init_from_env` is what a rank calls first -- before its first GPU call.
"""
import json
import os

import numpy as np


def world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def rank_assignment(rank: int, games, first_seed: int, n_seeds: int):
    """rank -> (game, seed): ranks walk the games first (8 games x 1 seed on 8 GPUs = BASELINE configs[3]), then the seeds."""
    games = list(games)
    assert games and n_seeds >= 1
    return games[rank % len(games)], first_seed + (rank // len(games)) % n_seeds


def init_from_env(backend: str = None, force: bool = False):
    """Join the job the launcher (experiments/launch.py or torchrun) described in the environment.  Selects this rank's
    GPU before anything touches it; returns (world_size, rank).  A single-process run is a no-op unless `force` (or
    ISDQN_DIST_FORCE=1) asks for a one-rank process group: that runs every line of the RCCL branch -- communicator set-up,
    all_gather, all_reduce, barrier -- on a single GPU (the smoke test of the N > 1 path that needs no multi-GPU node)."""
    ws, rank, local_rank = world()
    force = force or os.environ.get("ISDQN_DIST_FORCE") == "1"
    if ws <= 1 and not force:
        return 1, 0
    if ws <= 1:
        ws, rank = 1, 0
        if "MASTER_PORT" not in os.environ:  # a forced one-rank group: any free port (two such runs on one host must not collide)
            import socket

            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sock.getsockname()[1])
    import torch
    import torch.distributed as dist

    if dist.is_initialized():
        return ws, rank
    backend = backend or os.environ.get("ISDQN_DIST_BACKEND", "nccl")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    kw = {}
    if backend == "nccl":
        n = torch.cuda.device_count()  # (does not initialise the GPU)
        idx = local_rank if local_rank < n else 0  # launcher narrowed HIP_VISIBLE_DEVICES to one GPU -> cuda:0
        torch.cuda.set_device(idx)
        kw["device_id"] = torch.device("cuda", idx)
    dist.init_process_group(backend, rank=rank, world_size=ws, **kw)
    return ws, rank


def gather_metrics(metrics: np.ndarray) -> np.ndarray:
    """[world][len(metrics)] float32: all_gather of one small vector; identity ([1][n]) when not distributed."""
    try:
        import torch
        import torch.distributed as dist
    except ImportError:  # pragma: no cover
        return np.asarray(metrics, np.float32)[None]
    if not (dist.is_available() and dist.is_initialized()):
        return np.asarray(metrics, np.float32)[None]
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    mine = torch.tensor(np.asarray(metrics, np.float32), device=dev)
    out = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return torch.stack(out).cpu().numpy()


def write_gathered(path: str, per_epoch, fields, assignment=None) -> None:
    """Rank 0: the gathered per-epoch metrics of every replica as JSON ({"fields", "ranks", "epochs": [[[...]]]})."""
    _, rank, _ = world()
    if rank != 0:
        return
    os.makedirs(os.path.dirname(path), exist_ok=True)
    json.dump({"fields": list(fields), "ranks": assignment, "epochs": [np.asarray(g).tolist() for g in per_epoch]},
              open(path, "w"), indent=1)


def finalize() -> None:
    try:
        import torch.distributed as dist
    except ImportError:  # pragma: no cover
        return
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
