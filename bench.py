"""bench.py -- gradient-steps/sec of the iS-DQN replay-sample -> Bellman-update step on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|c5] [--precision bf16x3|bf16]

One "step" = one pass of the hot path over one batch: draw the batch from the device-resident replay
(uniform PCG64 draw, or float64 sum-tree inverse-CDF query for c3), gather the element rows, run
iSDQN.learn_on_batch (forward on 2B frame stacks read through the frame-id table, iterated Bellman
targets over the K heads, squared TD loss, backward, Adam) and, for c3, write sqrt(mean_k td) back into the
sum tree.  All inputs are resident in HBM when the timed region starts.

Workloads (BASELINE.json configs):
  c2 (default, configs[1]): Asterix-shaped A=9, K=9, B=256, cnn 32/64/64/512 + LayerNorm, uniform replay, n=1
  c3 (configs[2]):          same + prioritized replay (sum tree) + n=3 + priority writeback
  c5 (configs[4]):          Breakout-shaped A=4, K=32, B=1024

N > 1: one process per GPU, independent replicas (own seed, own replay, own parameters) -- the path shards
embarrassingly, no data-path collective; RCCL carries only the timing reduction.  Started as
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (the driver does that) the ranks come from
the environment; started as plain `python bench.py --gpus N` the parent spawns exactly that torchrun command as a child
BEFORE touching any GPU and relays its output.  value = N * K / max-over-ranks(time).  ONE JSON line on rank 0.

The timed region is EXACTLY --steps steps (whole hipGraph replays: the steps captured per graph divide --steps and leave
at least four replays in it) behind an untimed phase of whole replays -- at least --warmup steps plus the settle steps
reported as `settle_steps`, which take a fresh process to its steady state; a separate pass brackets single graph
replays with HIP events for median / p10 / p90.
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "is-dqn_amd"))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_MEASURED_GBS = 6290.0  # same guide ("~6.3 TB/s achievable"; SURVEY.md 8(d): 6.29 TB/s measured copy)
MFMA_PEAK_FLOPS = 2.5e15  # dense bf16 MFMA peak

WORKLOADS = {
    "c2": dict(n_actions=9, K=9, B=256, prioritized=False, n=1, desc="Asterix-shaped iS-DQN K=9 A=9 B=256 cnn32/64/64/512+LN, uniform device replay, n=1"),
    "c3": dict(n_actions=9, K=9, B=256, prioritized=True, n=3, desc="Asterix-shaped iS-DQN K=9 A=9 B=256 cnn32/64/64/512+LN, prioritized (sum-tree) device replay + writeback, n=3"),
    "c5": dict(n_actions=4, K=32, B=1024, prioritized=False, n=1, desc="Breakout-shaped iS-DQN K=32 A=4 B=1024 cnn32/64/64/512+LN, uniform device replay, n=1"),
}
FEATURES = (32, 64, 64, 512)


def algorithmic_bytes_per_step(B, K, A, prioritized):
    """uint8 frames of state+next_state read once, pre-LN activations of the B online samples
    saved and re-read once in bf16, 28 bytes per parameter (fp32 weights read in fwd and bwd, Adam reads m,v and
    writes p,m,v), plus the sum-tree traffic for c3."""
    P = 4_044_768 + (512 + 1) * (1 + K) * A
    frames = 2 * B * 28_224
    acts = B * 30_112 * 2 * 2
    tree = B * 21 * 24 if prioritized else 0
    return frames + acts + 28 * P + tree


def algorithmic_flops_per_step(B, K, A):
    mac = 16_003_072 + 512 * (1 + K) * A
    mac_conv0 = 441 * 256 * 32
    return 2 * mac * 2 * B + 2 * (2 * mac - mac_conv0) * B


def mfma_floor_seconds(B, K, A, precision):
    """The matrix-pipe floor of one step AT THE CHOSEN PRECISION: every contraction's algorithmic flops x the bf16 MFMA passes it is
    computed in, at the dense peak.  bf16x3 = hi*hi + lo*hi + hi*lo (3 passes); the first convolution's uint8 pixels are exact in
    bf16, so it takes 2 (forward, on 2B images) and its weight gradient 2 (no data gradient for the input layer); bf16 = 1 pass.
    DESIGN.md section 4: at bf16x3 this floor (c2: 35 us) is already ABOVE the 28.5 us that "70 % of the HBM roofline" allows."""
    mac = 16_003_072 + 512 * (1 + K) * A
    mac_conv0 = 441 * 256 * 32
    p, p0 = (3, 2) if precision == "bf16x3" else (1, 1)
    flop_passes = 6 * mac_conv0 * B * p0 + 8 * (mac - mac_conv0) * B * p  # fwd on 2B + wgrad on B | fwd on 2B + dgrad + wgrad on B
    return flop_passes / MFMA_PEAK_FLOPS


def measured_traffic(workload, precision):
    """HBM bytes per step from the PMC counters (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of THIS
    script's graphed workload, step kernels only, gfx950 correction applied: scripts/pmc_traffic.py; committed as JSON under
    profiles/).  Counters cannot be read from inside this process, so the line carries the newest committed measurement
    for the same workload and precision together with where it came from -- or null."""
    import glob
    import re

    def version(path):  # (round, v) as numbers: "v10" is newer than "v9"
        m = re.search(r"round(\d+).*_v(\d+)\.json$", path.replace(os.sep, "/"))
        return (int(m.group(1)), int(m.group(2))) if m else (0, 0)

    best, best_path = None, None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "round*", f"{workload}_hbm_traffic_*.json")), key=version):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        if d.get("workload") == workload and d.get("precision") == precision:
            best, best_path = d, f
    if best is None:
        return None, None
    src = {"file": os.path.relpath(best_path, ROOT), "build": best.get("build"), "collected_with": best.get("command"),
           "note": "not measured by this run"}
    return float(best["hbm_bytes_per_step_corrected"]), src


class Replica:
    """One independent (seed) replica of the training state on one GPU."""

    def __init__(self, workload, capacity, precision, seed, device, trust_mirror=False):
        import torch
        from slimdqn._engine import QNetEngine
        from slimdqn.sample_collection.replay_buffer import ReplayBuffer
        from slimdqn.sample_collection.samplers import PrioritizedSamplingDistribution, UniformSamplingDistribution

        w = WORKLOADS[workload]
        self.w = w
        self.device = device
        if w["prioritized"]:
            sampler = PrioritizedSamplingDistribution(seed, capacity, device=device)
        else:
            sampler = UniformSamplingDistribution(seed, device=device)
        self.rb = ReplayBuffer(sampler, w["B"], capacity, stack_size=4, update_horizon=w["n"], gamma=0.99, device=device)
        import numpy as np

        pri = np.random.default_rng(seed).uniform(0.1, 2.0, capacity) if w["prioritized"] else None
        self.rb.prefill_synthetic(capacity, (84, 84), w["n_actions"], seed=seed, p_terminal=0.005, priorities=pri)
        self.eng = QNetEngine((84, 84, 4), w["n_actions"], 1 + w["K"], FEATURES, "cnn", True, w["B"],
                              gamma_n=0.99 ** w["n"], learning_rate=6.25e-5, adam_eps=1.5e-4, precision=precision,
                              device=device)
        self.eng.init_params(seed)
        # `trust_mirror`: main() owns its engines the way the trainer owns its agent (experiments/base/dqn.py: train sets
        # agent.trust_mirror) -- every parameter write of the run goes through the engine, so a replay does not re-derive the bf16
        # weight mirror at its head (8 us per replay).  --rebuild-mirror, and every other user of this class (the GPU tests), keep the
        # engine's default, which assumes nothing about its caller.
        self.eng.trust_mirror = trust_mirror
        torch.cuda.synchronize()
        self.graphed = None

    def enable_graph(self, steps_per_graph):
        from slimdqn._graph import GraphedUpdate

        self.graphed = GraphedUpdate(self.rb, self.eng, self.w["prioritized"], steps_per_graph)

    def step(self):
        batch = self.rb.sample()
        cb = self.eng.make_batch(frames=batch.frames, frame_stride=batch.frame_stride, frame_ids=batch.frame_ids,
                                 action=batch.action, reward=batch.reward, terminal=batch.is_terminal)
        self.eng.learn_on_batch(cb)
        if self.w["prioritized"]:
            self.rb.update_device(batch, self.eng.priorities)
        return batch


def cpu_baseline(workload, seconds_budget=30.0, capacity=1_000_000, min_steps=50):
    """The oracle (torch-CPU fp32 restatement of the reference path: numpy PCG64 sampler (+ float64 sum tree), uint8
    stack gather out of ONE `capacity`-element replay, forward / backward / Adam) timed on this box's host cores on a
    BOUNDED sample of the same workload (>= 50 timed steps):
      1. thread sweep 1, 8, 16, 32, 64 (up to the cores this process may use), a few steps each, stopping at the first
         slowdown -- oversubscribed torch threads are pathological (round 2 timed 256 threads: 0.03 steps/s);
      2. `min_steps` timed steps at the best setting  -> `value`, `cores` = the threads used;
      3. `min_steps` timed steps with one thread        -> `one_thread_value`."""
    import numpy as np
    import torch
    from oracle.isdqn import iSDQN as Oracle
    from oracle.replay_buffer import ReplayBuffer as ORB, ReplayElement
    from oracle.samplers import PrioritizedSamplingDistribution as OP, UniformSamplingDistribution as OU

    w = WORKLOADS[workload]
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    cap = int(capacity)
    rng = np.random.default_rng(0)
    sampler = OP(0, cap) if w["prioritized"] else OU(0)
    rb = ORB(sampler, w["B"], cap, stack_size=4, update_horizon=w["n"], gamma=0.99)
    # one long synthetic stream like prefill_synthetic: element i = frames i..i+3 / i+n..i+n+3.  The stacks are VIEWS of the
    # frame store (7 GB at 1e6 elements; the reference would hold 56 GB of copies), copied at sampling time like the reference.
    # Full-range uint64 draws viewed as bytes: i.i.d. uniform pixels at memory speed.
    n_bytes = (cap + 8) * 84 * 84
    frames = np.empty(n_bytes, np.uint8)
    words = frames[: n_bytes // 8 * 8].view(np.uint64)
    for s0 in range(0, words.size, 1 << 24):
        e0 = min(words.size, s0 + (1 << 24))
        words[s0:e0] = rng.integers(0, np.iinfo(np.uint64).max, e0 - s0, dtype=np.uint64, endpoint=True)
    frames[n_bytes // 8 * 8:] = 0
    frames = frames.reshape(cap + 8, 84, 84)
    actions = rng.integers(0, w["n_actions"], cap)
    rewards = rng.choice([-1.0, 0.0, 1.0], size=cap, p=[0.05, 0.9, 0.05])
    mem = rb._memory
    n = w["n"]
    for i in range(cap):
        mem[i] = ReplayElement(np.moveaxis(frames[i : i + 4], 0, -1), int(actions[i]), float(rewards[i]),
                               np.moveaxis(frames[i + n : i + n + 4], 0, -1), False)
    sampler._index_to_key = list(range(cap))
    sampler._key_to_index = {i: i for i in range(cap)}
    if w["prioritized"]:
        sampler._sum_tree.set(np.arange(cap, dtype=np.int32), rng.uniform(0.1, 2.0, cap))
    rb.add_count = cap
    agent = Oracle(0, (84, 84, 4), w["n_actions"], w["K"], list(FEATURES), True, False, "cnn", 6.25e-5, 0.99, w["n"], 1, 8000, adam_eps=1.5e-4)

    def one():
        batch = rb.sample()
        agent.params, agent.optimizer_state, _ = agent.learn_on_batch(agent.params, agent.optimizer_state, batch)
        # (the oracle trainer has no priority write-back wiring either: reference SURVEY 8a P2)

    def timed(threads, n_steps, budget):
        torch.set_num_threads(threads)
        one()  # warm-up at this thread count
        t0 = time.perf_counter()
        k = 0
        while k < n_steps:
            one()
            k += 1
            if time.perf_counter() - t0 > budget:
                break
        return k / (time.perf_counter() - t0), k

    sweep = {}
    best_threads, best_rate = 1, 0.0
    for threads in [t for t in (1, 8, 16, 32, 64) if t <= usable] or [1]:
        rate, _ = timed(threads, 3, seconds_budget / 10)
        sweep[str(threads)] = round(rate, 3)
        if rate <= best_rate:
            break  # first slowdown: more threads only oversubscribe
        best_threads, best_rate = threads, rate
    all_rate, all_n = timed(best_threads, min_steps, seconds_budget)
    if best_threads == 1:
        one_rate, one_n = all_rate, all_n
    else:
        one_rate, one_n = timed(1, min_steps, seconds_budget)
    if one_rate > all_rate:  # (a box whose cores are busy elsewhere: the single thread is the better baseline)
        all_rate, all_n, best_threads = one_rate, one_n, 1
    return {"value": all_rate, "unit": "gradient-steps/s", "cores": best_threads, "kind": "port",
            "one_thread_value": one_rate, "os_cpu_count": os.cpu_count(), "usable_cores": usable, "thread_sweep": sweep,
            "sample": f"{all_n} timed steps with {best_threads} torch threads (best of the sweep {sweep}) and {one_n} timed steps with 1 thread of "
                      f"the oracle (torch-CPU fp32 restatement of the reference path) at B={w['B']}, K={w['K']}, one replay of {cap} elements "
                      f"(os.cpu_count() = {os.cpu_count()}, usable = {usable})"}


def max_over_ranks(seconds, device):
    """MAX of a scalar over all ranks (identity when torch.distributed is not initialised)."""
    import torch
    import torch.distributed as dist

    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def aggregate_value(world, steps, elapsed_max):
    """Whole-job throughput of `world` independent replicas that each ran `steps` steps."""
    return world * steps / elapsed_max


def spawn_ranks(n_gpus: int) -> int:
    """`python bench.py --gpus N` outside torchrun: start the N ranks as a child torchrun job.  Runs BEFORE this process
    has imported torch or touched a GPU (the parent never does), and hands the child's exit status back."""
    import socket
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def partitioned_streams(R: int, device):
    """R HIP streams whose kernels run on disjoint R-ths of the GPU's compute units (hipExtStreamCreateWithCUMask through the runtime
    torch itself loaded; bit b of the mask is CU b // n_xcd of XCD b % n_xcd on MI355X -- profiles/round4/cu_mask_probe.txt -- so a
    contiguous range of bits is the same share of every XCD, which the runtime requires: a mask that leaves an XCD empty is ignored)."""
    import ctypes

    import torch

    rt = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    n_cu = torch.cuda.get_device_properties(device).multi_processor_count
    assert n_cu % R == 0, f"{n_cu} compute units do not split into {R} equal shares"
    words = (n_cu + 31) // 32
    out = []
    for r in range(R):
        mask = (ctypes.c_uint32 * words)()
        for b in range(r * n_cu // R, (r + 1) * n_cu // R):
            mask[b // 32] |= 1 << (b % 32)
        h = ctypes.c_void_p()
        rc = rt.hipExtStreamCreateWithCUMask(ctypes.byref(h), ctypes.c_uint32(words), mask)
        assert rc == 0 and h.value, f"hipExtStreamCreateWithCUMask failed ({rc})"
        out.append(torch.cuda.ExternalStream(h.value, device=device))
    return out


def steps_per_graph(steps: int, limit: int, min_replays: int = 4) -> int:
    """Largest S <= limit dividing the timed step count (a graph replays S steps at a time: the timed region is EXACTLY `steps`) that
    leaves at least four replays in it -- launching a replay overlaps the previous one's execution; ONE 20-step replay measured 3 460 -
    3 830 steps/s where four 5-step replays give 3 790 - 3 980.  The untimed phase in front runs whole replays too: at least the
    requested warm-up, the surplus is reported with `settle_steps`."""
    limit = max(1, min(limit, steps // max(1, min_replays)))
    return max(d for d in range(1, limit + 1) if steps % d == 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: ~1.2 s of timed work (300-step runs scatter by +-15 %) behind ~1.2 s of warm-up
    ap.add_argument("--steps", type=int, default=4000)
    ap.add_argument("--warmup", type=int, default=4000)
    ap.add_argument("--settle", type=int, default=-1, help="untimed steps in front of the warm-up (default: so that warm-up + settle >= 4000, one second)")
    ap.add_argument("--workload", default="c2", choices=list(WORKLOADS))
    ap.add_argument("--K", type=int, default=0, help="override the number of Bellman iterations of the workload (the reference's "
                    "timing sweep launch_job/atari/launch_time.sh:13-27 runs K in 1, 4, 9, 49)")
    ap.add_argument("--B", type=int, default=0, help="override the batch size of the workload (measurements only: the metric is quoted on the workload's own)")
    ap.add_argument("--precision", default="bf16x3", choices=["bf16x3", "bf16"])
    ap.add_argument("--capacity", type=int, default=1_000_000)
    ap.add_argument("--graph", type=int, default=32, help="most steps captured per hipGraph (0 = eager launches)")
    ap.add_argument("--rebuild-mirror", action="store_true", help="rebuild the bf16 weight mirror at the head of every replay (the engine's "
                    "default for callers it knows nothing about) instead of declaring, as the trainer does, that all parameter writes are the engine's own")
    ap.add_argument("--min-replays", type=int, default=4, help="fewest graph replays the timed region is cut into")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=30.0, help="wall-clock cap of EACH timed leg of the CPU baseline (50 steps each otherwise)")
    ap.add_argument("--cpu-capacity", type=int, default=1_000_000)
    ap.add_argument("--replay-stats", type=int, default=200, help="graph replays timed one by one after the run (0 = skip)")
    ap.add_argument("--replicas-per-gpu", type=int, default=1, help="independent replicas sharing each GPU on their own streams, as the "
                    "reference shares one GPU between seeds (launch_job/atari/normal/train.sh:9-16); the headline number is 1")
    ap.add_argument("--cu-partition", action="store_true", help="with --replicas-per-gpu R: every replica's stream gets its own 1/R of the "
                    "compute units (hipExtStreamCreateWithCUMask: CUs [r*n/R, (r+1)*n/R) of the mask's order, i.e. the same share of every "
                    "XCD), so the replicas run side by side instead of time-slicing the whole chip")
    args = ap.parse_args()
    assert args.steps >= 1 and args.warmup >= 0 and args.gpus >= 1

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"[bench] --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks: refusing to report a wrong n_gpus",
              file=sys.stderr, flush=True)
        sys.exit(2)

    import torch

    dist = None
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    device = f"cuda:{local_rank}"
    if args.K > 0:
        WORKLOADS[args.workload] = dict(WORKLOADS[args.workload], K=args.K, desc=WORKLOADS[args.workload]["desc"].replace(
            f"K={WORKLOADS[args.workload]['K']}", f"K={args.K}"))
    if args.B > 0:
        WORKLOADS[args.workload] = dict(WORKLOADS[args.workload], B=args.B, desc=WORKLOADS[args.workload]["desc"].replace(
            f"B={WORKLOADS[args.workload]['B']}", f"B={args.B}"))
    w = WORKLOADS[args.workload]

    R = max(1, args.replicas_per_gpu)
    reps = [Replica(args.workload, args.capacity, args.precision, seed=rank * R + r, device=device, trust_mirror=not args.rebuild_mirror)
            for r in range(R)]
    rep = reps[0]
    S = 1
    if args.graph > 0:
        S = steps_per_graph(args.steps, args.graph, args.min_replays)
        try:
            for x in reps:
                x.enable_graph(S)
        except Exception as e:  # capture is an optimisation: fall back to eager launches, loudly
            print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr, flush=True)
            for x in reps:
                x.graphed = None
            S = 1
    if R == 1:
        one = (lambda: rep.graphed.run()) if rep.graphed is not None else rep.step
    else:  # every replica on its own stream: their kernels interleave on the GPU (a "step" below is one step of EVERY replica)
        streams = partitioned_streams(R, device) if args.cu_partition else [torch.cuda.Stream(device) for _ in reps]

        def one():
            for x, st in zip(reps, streams):
                with torch.cuda.stream(st):
                    x.graphed.run() if x.graphed is not None else x.step()
    settle = args.settle if args.settle >= 0 else max(0, 4000 - args.warmup)
    untimed = (settle + args.warmup + S - 1) // S * S  # whole replays: >= settle + warm-up
    for _ in range(untimed // S):
        one()
    torch.cuda.synchronize()
    # One more untimed replay, drained on its own: the first launch behind hundreds of back-to-back replays pays for the runtime
    # reclaiming their submission resources (410 - 510 us on the host against 250 for the following ones; 290 behind a drained
    # replay) -- inside a 20-step timed region that is 4 % and most of its scatter (profiles/round4/driver_form_host.txt, _host2.txt).
    one()
    untimed += S
    settle = untimed - args.warmup
    torch.cuda.synchronize()
    # Nothing slow between that synchronisation and the timed region: 20 ms of idle GPU cost the next 5 ms 10 % of their clock (a
    # gc.collect() here measured 3 650 steps/s against 4 150; driver_form_host.txt "20 ms host spin", driver_form_variants.txt).
    gc.disable()  # (no collector pause inside a 5 ms region; re-enabled behind it)
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()

    # HIP events on the stream the kernels are launched on (torch's current stream is handed to the C ABI)
    # (an event record costs the stream a ~6 us bubble on this stack: two events bracket the region, not each step)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for i in range(args.steps // S):
        one()
    ev1.record()
    t_issue = time.perf_counter() - t0  # host time to enqueue the region (== elapsed when the host is the bottleneck)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    gc.enable()
    rep.rb._sampling_distribution._sum_tree.check_status() if w["prioritized"] else None

    elapsed_max = max_over_ranks(elapsed, device)
    per_rank_ms = [elapsed / args.steps * 1e3]
    if dist is not None:  # every rank's own ms_per_step: a straggler (a slow GPU, a busy host core) is visible in the line
        t_all = [torch.zeros(1, dtype=torch.float64, device=device) for _ in range(world)]
        dist.all_gather(t_all, torch.tensor([elapsed / args.steps * 1e3], dtype=torch.float64, device=device))
        per_rank_ms = [float(t.item()) for t in t_all]

    # distribution over single replays (own pass, outside the timed region: every replay bracketed by its own events)
    stats = None
    if rank == 0 and args.replay_stats > 0:
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.replay_stats)]
        for a, b in evs:
            a.record()
            one()
            b.record()
        torch.cuda.synchronize()
        per_step = sorted(a.elapsed_time(b) / S for a, b in evs)
        q = lambda f: per_step[min(len(per_step) - 1, int(f * len(per_step)))]
        stats = {"replays": len(per_step), "steps_per_replay": S, "ms_per_step_median": q(0.5), "ms_per_step_p10": q(0.1),
                 "ms_per_step_p90": q(0.9), "note": "each replay bracketed by its own HIP events (adds ~6 us per replay); not the timed region"}

    if rank == 0:
        # (several replicas per GPU run on their own streams: the two events bracket only the default stream, so the
        # device time per replica step is the synchronised wall time of the region)
        dev_ms_avg = ev0.elapsed_time(ev1) / args.steps if R == 1 else elapsed * 1e3 / (args.steps * R)
        bytes_step = algorithmic_bytes_per_step(w["B"], w["K"], w["n_actions"], w["prioritized"])
        flops_step = algorithmic_flops_per_step(w["B"], w["K"], w["n_actions"])
        achieved = bytes_step / (dev_ms_avg * 1e-3) / 1e9
        traffic, traffic_src = measured_traffic(args.workload, args.precision)
        floor_s = mfma_floor_seconds(w["B"], w["K"], w["n_actions"], args.precision)
        hbm_s = bytes_step / (HBM_PEAK_GBS * 1e9)
        out = {
            "metric": f"gradient-steps/sec (batch={w['B']}, K={w['K']}, 84x84x4)",
            "value": aggregate_value(world * R, args.steps, elapsed_max),
            "unit": "gradient-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "settle_steps": settle,
            "ms_per_step": elapsed_max / args.steps * 1e3,
            "ms_per_step_per_rank": per_rank_ms,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16x3 (split-bf16 MFMA operands hi+lo, fp32 accumulate)" if args.precision == "bf16x3" else "bf16 (single pass, fp32 accumulate)",
            "data": "synthetic",
            "config": {"workload": w["desc"], "replay_capacity": args.capacity, "precision": args.precision,
                       "launch": f"hipGraph x{S} steps" if rep.graphed is not None else "eager",
                       "weight_mirror": "rebuilt at the head of every replay" if args.rebuild_mirror else "owned by the caller (trainer's setting)", "replicas": world * R, "replicas_per_gpu": R, "cu_partition": bool(args.cu_partition and R > 1), "parallelism": f"independent-seed replicas x{world * R}"},
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic, "traffic_source": traffic_src,
                "kernel": "replay-sample -> Bellman-update step (all launches of one step; HIP-event time per step)",
                "algorithmic_bytes_per_step": bytes_step, "device_ms_per_step_avg": dev_ms_avg,
                "host_issue_ms_per_step": t_issue / args.steps * 1e3,
                "mfma_util_vs_2.5PF": flops_step / (dev_ms_avg * 1e-3) / MFMA_PEAK_FLOPS,
                # the same achieved rate against the copy bandwidth this part really delivers (SURVEY 8d asks for both)
                "frac_vs_measured_bw": achieved / HBM_MEASURED_GBS, "measured_bw_GBs": HBM_MEASURED_GBS,
                # what the chosen arithmetic allows at best: the step cannot take less than its MFMA passes at the dense peak, so
                # `frac` is capped at hbm_time / max(hbm_time, mfma_floor) whatever the kernels do
                "mfma_floor_us_at_precision": floor_s * 1e6, "hbm_time_us_at_peak": hbm_s * 1e6,
                "max_frac_at_mfma_floor": hbm_s / max(hbm_s, floor_s),
                "frac_of_attainable": (achieved / HBM_PEAK_GBS) / (hbm_s / max(hbm_s, floor_s)),
            },
        }
        if stats is not None:
            out["replay_stats"] = stats
        if world == 1 and not args.no_cpu_baseline:
            del rep  # (frees nothing the baseline needs; keeps host memory for the 7 GB oracle frame store)
            out["cpu_baseline"] = cpu_baseline(args.workload, args.cpu_seconds, args.cpu_capacity)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
