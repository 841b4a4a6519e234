"""world_size-2 CPU (gloo) test of the only cross-rank traffic of the path, started THROUGH the launcher
(experiments/launch.py: one process per rank, RANK / WORLD_SIZE / MASTER_* in the environment, 127.0.0.1 rendezvous):
the max-over-ranks timing of bench.py, the per-epoch metric all_gather of the trainer and rank 0's gathered JSON
(independent replicas, no data-path collective; reference fan-out: launch_job/atari/normal/train.sh:12-16)."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "is-dqn_amd")

WORKER = textwrap.dedent(
    """
    import os, sys
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "is-dqn_amd"))
    import numpy as np
    from experiments.base import dist as replicas
    world, rank = replicas.init_from_env()          # backend from ISDQN_DIST_BACKEND (the launcher's --backend gloo)
    assert world == 2 and rank == int(os.environ["RANK"])
    # the launcher's rank -> (game, seed) assignment: two games, one seed
    assert (os.environ["ISDQN_GAME"], os.environ["ISDQN_SEED"]) == (["Asterix", "Breakout"][rank], "7")
    import bench
    from experiments.base.dqn import _gather_epoch_metrics, EPOCH_FIELDS
    # each rank is an independent replica that took a different time
    elapsed = [0.50, 0.80][rank]
    m = bench.max_over_ranks(elapsed, "cpu")
    assert m == 0.80, m
    assert abs(bench.aggregate_value(world, 100, m) - 250.0) < 1e-9
    g = _gather_epoch_metrics(np.asarray([10.0 + rank, 100.0 * (rank + 1), 7.0, 1000.0], np.float32))
    assert g.shape == (2, 4), g.shape
    assert g[:, 0].tolist() == [10.0, 11.0] and g[:, 1].tolist() == [100.0, 200.0], g
    replicas.write_gathered({out!r}, [g], EPOCH_FIELDS, assignment=[["Asterix", 7], ["Breakout", 7]])
    replicas.finalize()
    print("rank", rank, "ok")
    """
)


def test_rank_assignment_walks_games_then_seeds():
    sys.path.insert(0, PKG)
    from experiments.base.dist import rank_assignment

    games = ["Asterix", "Breakout", "Pong", "Seaquest", "Qbert", "SpaceInvaders", "BeamRider", "Enduro"]
    # BASELINE configs[3]: 8 games x 1 seed, one per GPU
    assert [rank_assignment(r, games, 1, 1) for r in range(8)] == [(g, 1) for g in games]
    # 2 games x 2 seeds on 4 GPUs
    assert [rank_assignment(r, games[:2], 5, 2) for r in range(4)] == [("Asterix", 5), ("Breakout", 5), ("Asterix", 6), ("Breakout", 6)]


def test_two_replicas_through_the_launcher(tmp_path):
    out = tmp_path / "gathered.json"
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, out=str(out)))
    logs = tmp_path / "logs"
    cmd = [sys.executable, os.path.join(PKG, "experiments", "launch.py"), "--gpus", "2", "--games", "Asterix", "Breakout",
           "--first_seed", "7", "--backend", "gloo", "--no-gpu-pinning", "--module", str(script), "--log_dir", str(logs)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    texts = {f: open(os.path.join(logs, f)).read() for f in sorted(os.listdir(logs))} if logs.exists() else {}
    assert r.returncode == 0, f"launcher failed: {r.stdout}\n{r.stderr}\n{texts}"
    assert len(texts) == 2 and all(f"rank {i} ok" in t for i, t in enumerate(texts.values())), texts
    d = json.load(open(out))
    assert d["fields"][:3] == ["avg_return", "avg_length_episode", "n_training_steps"]
    assert d["ranks"] == [["Asterix", 7], ["Breakout", 7]] and np_shape(d["epochs"]) == (1, 2, 4)


def np_shape(x):
    import numpy as np

    return np.asarray(x).shape


def test_bench_refuses_a_gpu_count_that_differs_from_the_launched_world(tmp_path):
    """`bench.py --gpus 8` under a 1-rank environment must not print n_gpus = 1 (round-1 defect): it exits with status 2
    before touching a GPU."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "8", "--warmup", "8"],
                       capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 2 and "WORLD_SIZE=1" in r.stderr and not r.stdout.strip()


def test_bench_steps_per_graph_divides_the_timed_steps():
    sys.path.insert(0, ROOT)
    import bench

    # the timed region is exactly --steps: whole replays; the untimed phase (warm-up + declared settle steps) is rounded UP to replays
    assert bench.steps_per_graph(20, 32) == 5        # the driver's --steps 20 --warmup 5: four replays of 5 timed steps
    assert bench.steps_per_graph(20, 4) == 4
    assert bench.steps_per_graph(4000, 32) == 32
    assert bench.steps_per_graph(7, 8) == 1 and bench.steps_per_graph(24, 8) == 6 and bench.steps_per_graph(3, 8) == 1
    assert bench.steps_per_graph(20, 32, min_replays=2) == 10 and bench.steps_per_graph(20, 32, min_replays=1) == 20


def test_launcher_refuses_incomplete_arguments_before_starting_anything():
    """The reference's tests/test_launch_job.py:4-37 on experiments/launch.py: no experiment name, no first seed, a last seed below the
    first one -- every case ends with a non-zero status (and no child process: the refusals come before the first Popen)."""
    from experiments import launch

    assert launch.main([]) > 0                                                                    # no experiment name
    assert launch.main(["--", "-en", "_test_launch_local"]) > 0                                   # no first seed
    assert launch.main(["--first_seed", "10", "--last_seed", "1", "--", "-en", "_test_launch_local"]) > 0  # empty seed range
    assert launch.main(["--first_seed", "1", "--last_seed", "3", "--n_seeds", "2", "--", "-en", "_test_launch_local"]) > 0
