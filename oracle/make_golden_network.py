"""Generate tests/golden/network_K{1,9}.npz and tests/golden/replay.npz from the ORACLE (fixture groups 2, 3).

What these fixtures are and are not.  The reference holds no numbers for the network path (tests/test_isdqn.py:51-116 compares the
agent with an inline restatement that calls the same Flax network), and jax / flax / optax cannot run here, so nothing can PIN the
network oracle: parity stays "unpinned" (oracle/network.py).  The fixtures FREEZE the restatement instead: the oracle and the HIP
kernels are both checked against committed numbers, so the two cannot drift together unnoticed
(tests/test_golden_fixtures.py regenerates them on the CPU; tests/test_gpu_golden.py holds the HIP path to them).

Group 2 -- one fixed batch through the full-size network (cnn 32/64/64/512 + LayerNorm on 84x84x4, B = 4; K = 9 with A = 9 and
K = 1 with A = 6): uint8 inputs, `all_q` (isdqn.py:95), q / targets / per-head losses (isdqn.py:97-103) in float32 and float64,
gradients of Dense_1 and LayerNorm_3 (float64), parameters after 1 and after 3 Adam steps on that batch (isdqn.py:82-90; small
tensors whole, the three big kernels as strided samples and float64 sums), the head matrix after `shift_params` (isdqn.py:111-125).
The initial parameters are not stored (16 MB): they are `oracle.network.init_params(seed)` -- numpy PCG64, the same everywhere --
and the fixture keeps the SHA-256 of every leaf.

Group 3 -- the replay / sampler known answers of the reference's tests as data: FIFO keys 5..14 and frame contents
(tests/test_replay_buffer.py:49-85), n = 5 / gamma = 1 / r = 2 returns (:87-105), leading zero padding and frame order (:107-133),
the seed-0 batch of 32 keys (:135-203), and the prioritized sampler sequence of tests/test_samplers.py:10-35.

Usage:  python oracle/make_golden_network.py        (CPU, about a minute)
"""
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import network as net  # noqa: E402
from oracle.isdqn import iSDQN  # noqa: E402
from oracle.replay_buffer import ReplayBuffer, ReplayElement, TransitionElement  # noqa: E402
from oracle.samplers import PrioritizedSamplingDistribution, UniformSamplingDistribution  # noqa: E402

FEATS, OBS, B = [32, 64, 64, 512], (84, 84, 4), 4
LR, ADAM_EPS, GAMMA = 6.25e-5, 1.5e-4, 0.99  # launch_job/atari/launch.sh:1-3, experiments/atari/isdqn.py:46
CASES = {"K9": dict(K=9, A=9, seed=0), "K1": dict(K=1, A=6, seed=1)}
SMALL = ("Conv_0", "Conv_1", "Conv_2", "Dense_0")  # their kernels are sampled, everything else is stored whole
STRIDE = 1009


def batch_for(case):
    rng = np.random.default_rng(100 + case["seed"])
    return ReplayElement(
        state=rng.integers(0, 256, (B,) + OBS, dtype=np.uint8), action=rng.integers(0, case["A"], B).astype(np.int64),
        reward=rng.choice([-1.0, 0.0, 1.0], B).astype(np.float64), next_state=rng.integers(0, 256, (B,) + OBS, dtype=np.uint8),
        is_terminal=np.array([0, 0, 1, 0], np.int64))


def leaf_sha(params):
    return {f"{m}/{n}": hashlib.sha256(np.ascontiguousarray(v, np.float32).tobytes()).hexdigest() for m, l in params.items() for n, v in l.items()}


def store_params(out, tag, params_np):
    """Small tensors whole; the big kernels as every STRIDE-th element (flat, Flax layout) plus their float64 sum and sum of squares."""
    for m, leaves in params_np.items():
        for n, v in leaves.items():
            v = np.asarray(v)
            if m in SMALL and n == "kernel":
                out[f"{tag}/{m}/{n}/sample"] = v.reshape(-1)[::STRIDE].astype(np.float32)
                out[f"{tag}/{m}/{n}/sum"] = np.float64(v.astype(np.float64).sum())
                out[f"{tag}/{m}/{n}/sumsq"] = np.float64((v.astype(np.float64) ** 2).sum())
            else:
                out[f"{tag}/{m}/{n}"] = v.astype(np.float32)


def network_case(name):
    """Every array of one fixture file (a dict name -> numpy array), computed from scratch on the CPU."""
    case = CASES[name]
    K, A = case["K"], case["A"]
    params = net.init_params(case["seed"], OBS, FEATS, "cnn", (1 + K) * A, True)
    batch = batch_for(case)
    out = {"K": np.int64(K), "A": np.int64(A), "seed": np.int64(case["seed"]), "B": np.int64(B),
           "lr": np.float64(LR), "adam_eps": np.float64(ADAM_EPS), "gamma": np.float64(GAMMA),
           "state": batch.state, "next_state": batch.next_state, "action": batch.action, "reward": batch.reward,
           "is_terminal": batch.is_terminal}
    for k, h in leaf_sha(params).items():
        out[f"init_sha256/{k}"] = np.frombuffer(bytes.fromhex(h), dtype=np.uint8)
    for tag, dtype in (("f32", torch.float32), ("f64", torch.float64)):
        ag = iSDQN(case["seed"], OBS, A, K, FEATS, True, False, "cnn", LR, GAMMA, 1, 1, 1, adam_eps=ADAM_EPS, dtype=dtype, params=params)
        with torch.no_grad():
            all_q = ag.apply(ag.params, torch.cat((torch.as_tensor(batch.state), torch.as_tensor(batch.next_state))))
            q, tg, td = ag.loss_terms(ag.params, batch)
        out[f"{tag}/all_q"] = all_q.numpy()
        out[f"{tag}/q_values"], out[f"{tag}/targets"] = q.numpy(), tg.numpy()
        out[f"{tag}/losses"] = td.mean(0).numpy()
        full = tag == "f64"  # gradients and parameter trajectories: from the float64 run only (stored as float32 where they are weights)
        if full:
            grads, _ = ag.grads(ag.params, batch)
            for m in ("Dense_1", "LayerNorm_3"):
                for n, g in grads[m].items():
                    out[f"{tag}/grad/{m}/{n}"] = g.numpy()
        p, st = ag.params, ag.optimizer_state
        for step in (1, 2, 3):
            p, st, losses = ag.learn_on_batch(p, st, batch)
            out[f"{tag}/losses_step{step}"] = np.asarray(losses)
            if full and step in (1, 3):
                store_params(out, f"{tag}/after{step}", net.to_numpy(p))
        if full:
            shifted = net.to_numpy(ag.shift_params(p))
            out[f"{tag}/shifted/Dense_1/kernel"] = shifted["Dense_1"]["kernel"].astype(np.float32)
            out[f"{tag}/shifted/Dense_1/bias"] = shifted["Dense_1"]["bias"].astype(np.float32)
    return out


def replay_cases():
    """tests/test_replay_buffer.py / test_samplers.py known answers, produced by the oracle replay (the reference's own numbers are
    asserted in tests/test_oracle_replay.py; here they become data the HIP replay is held to as well)."""
    out = {}
    # FIFO eviction, capacity 10, 15 + stack adds of frames filled with their index (test_replay_buffer.py:49-85)
    rb = ReplayBuffer(UniformSamplingDistribution(0), 2, 10, stack_size=4, update_horizon=1, gamma=0.99, compress=False)
    for i in range(16):
        rb.add(TransitionElement(np.full((84, 84), i, np.uint8), i % 3, float(i), False, False))
    keys = sorted(rb._memory.keys())
    out["fifo/keys"] = np.asarray(keys, np.int64)
    out["fifo/add_count"] = np.int64(rb.add_count)
    out["fifo/state_fill"] = np.asarray([[int(rb._memory[k].state[0, 0, c]) for c in range(4)] for k in keys], np.int64)
    out["fifo/next_state_fill"] = np.asarray([[int(rb._memory[k].next_state[0, 0, c]) for c in range(4)] for k in keys], np.int64)
    out["fifo/action"] = np.asarray([int(rb._memory[k].action) for k in keys], np.int64)
    out["fifo/reward"] = np.asarray([float(rb._memory[k].reward) for k in keys], np.float64)
    # n-step return: n = 5, gamma = 1, reward 2 (test_replay_buffer.py:87-105)
    rb = ReplayBuffer(UniformSamplingDistribution(0), 8, 100, stack_size=4, update_horizon=5, gamma=1.0, compress=False)
    for i in range(50):
        rb.add(TransitionElement(np.full((84, 84), i, np.uint8), 0, 2.0, False, False))
    out["nstep/rewards"] = np.asarray(rb.sample().reward, np.float64)
    # leading zero padding and frame order at an episode start (test_replay_buffer.py:107-133)
    rb = ReplayBuffer(UniformSamplingDistribution(0), 1, 100, stack_size=4, update_horizon=1, gamma=0.99, compress=False)
    for i in range(1, 4):
        rb.add(TransitionElement(np.full((84, 84), i, np.uint8), 0, 0.0, False, False))
    first = rb._memory[min(rb._memory.keys())]
    out["stack/first_state_fill"] = np.asarray([int(first.state[0, 0, c]) for c in range(4)], np.int64)
    out["stack/first_next_state_fill"] = np.asarray([int(first.next_state[0, 0, c]) for c in range(4)], np.int64)
    # the seed-0 batch of 32 keys over a buffer that evicted (test_replay_buffer.py:135-203)
    rb = ReplayBuffer(UniformSamplingDistribution(0), 32, 20, stack_size=4, update_horizon=1, gamma=0.99, compress=False)
    for i in range(40):
        rb.add(TransitionElement(np.full((84, 84), i % 251, np.uint8), i % 4, float(i % 3) - 1.0, i % 11 == 10, i % 11 == 10))
    batch = rb.sample()
    out["seed0/keys_in_memory"] = np.asarray(sorted(rb._memory.keys()), np.int64)
    out["seed0/state_fill"] = np.asarray(batch.state[:, 0, 0, :], np.int64)
    out["seed0/next_state_fill"] = np.asarray(batch.next_state[:, 0, 0, :], np.int64)
    out["seed0/action"], out["seed0/reward"] = np.asarray(batch.action, np.int64), np.asarray(batch.reward, np.float64)
    out["seed0/is_terminal"] = np.asarray(batch.is_terminal, np.int64)
    # prioritized sampler sequence (test_samplers.py:10-35): capacity 10, seed 0; a zero-priority key, an update to zero, a remove
    sp = PrioritizedSamplingDistribution(0, 10)
    for key in range(6):
        sp.add(key, priority=0.0 if key == 3 else 1.0 + 0.5 * key)
    out["prio/sample_a"] = np.asarray(sp.sample(64), np.int64)
    sp.update(np.asarray([1, 2]), priorities=np.asarray([0.0, 0.0]))
    out["prio/sample_b"] = np.asarray(sp.sample(64), np.int64)
    sp.remove(0)
    out["prio/sample_c"] = np.asarray(sp.sample(64), np.int64)
    out["prio/root"] = np.float64(sp._sum_tree.root)
    out["prio/nodes"] = np.asarray(sp._sum_tree._nodes, np.float64)
    return out


def main():
    gold = os.path.join(ROOT, "tests", "golden")
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    for name in CASES:
        path = os.path.join(gold, f"network_{name}.npz")
        out = network_case(name)
        np.savez_compressed(path, **out)
        print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path)} bytes")
    path = os.path.join(gold, "replay.npz")
    out = replay_cases()
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path)} bytes")


if __name__ == "__main__":
    main()
