"""world_size-2 CPU (gloo) test of the only cross-rank traffic of the path: the max-over-ranks timing of
bench.py and the per-epoch metric all_gather of the trainer (independent replicas, no data-path
collective).  Launched as two real processes with a 127.0.0.1 rendezvous."""
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent(
    """
    import os, sys
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "is-dqn_amd"))
    import numpy as np, torch, torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from experiments.base.dqn import _gather_epoch_metrics
    # each rank is an independent replica that took a different time
    elapsed = [0.50, 0.80][rank]
    m = bench.max_over_ranks(elapsed, "cpu")
    assert m == 0.80, m
    assert abs(bench.aggregate_value(world, 100, m) - 250.0) < 1e-9
    g = _gather_epoch_metrics(np.asarray([10.0 + rank, 100.0 * (rank + 1), 7.0], np.float32))
    assert g.shape == (2, 3), g.shape
    assert g[:, 0].tolist() == [10.0, 11.0] and g[:, 1].tolist() == [100.0, 200.0], g
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
    """
)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_two_replicas_reduce_timing_and_gather_metrics(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {rank} failed:\\n{out}"
        assert f"rank {rank} ok" in out
