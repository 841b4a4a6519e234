"""One-rank RCCL smoke test: every line of the `nccl` branch of the multi-GPU path (experiments/base/dist.py, bench.py's
max-over-ranks reduction) runs on the real device without a multi-GPU node.  The child is a fresh process that joins the
group BEFORE its first GPU call, as a launcher-started rank does (reference fan-out: launch_job/atari/normal/train.sh:12-16).
The 2/4/8-GPU curve itself stays unmeasured here."""
import json
import os
import socket
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent(
    """
    import os, sys
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "is-dqn_amd"))
    import numpy as np
    from experiments.base import dist as replicas
    world, rank = replicas.init_from_env(backend="nccl", force=True)   # before any GPU call of this process
    assert (world, rank) == (1, 0)
    import torch
    import torch.distributed as dist
    assert dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() == 1
    import bench
    from experiments.base.dqn import _gather_epoch_metrics, EPOCH_FIELDS
    m = bench.max_over_ranks(0.25, "cuda:0")                            # all_reduce(MAX) over RCCL
    assert m == 0.25, m
    g = _gather_epoch_metrics(np.asarray([10.0, 100.0, 7.0, 1000.0], np.float32))   # all_gather over RCCL
    assert g.shape == (1, 4) and g[0].tolist() == [10.0, 100.0, 7.0, 1000.0], g
    replicas.write_gathered({out!r}, [g], EPOCH_FIELDS, assignment=[["Asterix", 1]])
    replicas.finalize()                                                 # barrier + destroy_process_group
    assert not dist.is_initialized()
    print("rccl world-1 ok")
    """
)


@pytest.mark.gpu
def test_one_rank_rccl_group_runs_the_metric_collectives(tmp_path):
    out = tmp_path / "gathered_metrics.json"
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, out=str(out)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "rccl world-1 ok" in r.stdout, f"{r.stdout}\n{r.stderr}"
    d = json.load(open(out))
    assert d["ranks"] == [["Asterix", 1]] and len(d["epochs"]) == 1 and d["epochs"][0] == [[10.0, 100.0, 7.0, 1000.0]]
