"""n host environments stepped in lockstep, optionally by worker processes on the host cores.

Not in the reference, whose loop drives ONE ALE instance one action at a time (slimdqn/environments/atari.py:58-78,
slimdqn/sample_collection/utils.py:21-43); the north star keeps the emulators on the host cores and asks for them to feed the
device replay, and once the update step is fast the emulators are the wall-clock bottleneck (4 environment steps per gradient
step).  Two forms behind one surface:

* ``VectorEnv(envs)``: the environments live in this process and are stepped one after another (tests, n = 1..2);
* ``VectorEnv(make_env=spec, n_envs=n, n_workers=w)``: ``w`` worker processes own ``n / w`` environments each and step them
  in parallel.  A worker is a fresh interpreter (``python -m slimdqn.environments._worker``) that imports numpy and the
  environment module only -- it never touches the GPU, and neither torch nor this process's ``__main__`` are loaded into
  it -- and exchanges data with this process through ONE shared-memory block:

      planes   uint8 [n][stack][h*w]   every environment's frame stack, planar, oldest .. newest -- exactly the layout the
                                       batched acting forward reads (``frames[slot][h*w]`` + an id table), so the whole block
                                       goes to the device in ONE host-to-device copy per round (pinned when the HIP runtime
                                       agrees to register the mapping)
      obs      uint8 [n][h*w]          the frame each environment showed BEFORE the step (TransitionElement.observation)
      reward   f64 [n], absorbing / episode_end uint8 [n], actions int64 [n]

  and one byte per round over a pipe in each direction (go / done).  ``step_async(actions)`` returns at once, ``step_wait()``
  when every worker has finished: the trainer enqueues the gradient steps of a round in between, so the emulators run under
  the update kernels instead of beside an idle GPU.

Episode ends are handled where the environment lives (``episode_end = absorbing or n_steps >= horizon`` and the reset, as
``collect_single_sample`` does: utils.py:28-42).
"""
from __future__ import annotations

import json
import os
import select
import subprocess
import sys
from multiprocessing import shared_memory

import numpy as np


def shared_layout(n: int, stack: int, hw: int):
    """Byte offsets of the sections of the shared block (every section 64-byte aligned) and its total size."""
    off, out = 0, {}
    for name, nbytes in (("planes", n * stack * hw), ("obs", n * hw), ("reward", 8 * n), ("absorbing", n), ("episode_end", n),
                         ("actions", 8 * n)):
        out[name] = off
        off += (nbytes + 63) // 64 * 64
    out["total"] = off
    return out


def map_shared(buf, n: int, stack: int, hw: int):
    lay = shared_layout(n, stack, hw)
    view = lambda name, dtype, shape: np.ndarray(shape, dtype=dtype, buffer=buf, offset=lay[name])
    return dict(
        planes=view("planes", np.uint8, (n, stack, hw)), obs=view("obs", np.uint8, (n, hw)), reward=view("reward", np.float64, (n,)),
        absorbing=view("absorbing", np.uint8, (n,)), episode_end=view("episode_end", np.uint8, (n,)), actions=view("actions", np.int64, (n,)),
    )


def write_planes(dst: np.ndarray, env) -> None:
    """dst [stack][h*w] <- the environment's (h, w, stack) state, planar, oldest .. newest."""
    s = env.state_ if hasattr(env, "state_") else env.state
    s = np.asarray(s)
    stack = dst.shape[0]
    dst[:] = np.moveaxis(s, -1, 0).reshape(stack, -1)  # (uint8 stacks copy as they are; a float32 `state` is cast)


def step_one(env, action: int, horizon: int, i: int, sh) -> None:
    """One environment step into the shared arrays (row i): what collect_single_sample does around env.step (utils.py:28-42)."""
    sh["obs"][i] = np.asarray(env.observation).reshape(-1)
    reward, absorbing = env.step(int(action))
    episode_end = bool(absorbing) or env.n_steps >= horizon
    sh["reward"][i] = reward
    sh["absorbing"][i] = 1 if absorbing else 0
    sh["episode_end"][i] = 1 if episode_end else 0
    if episode_end:
        env.reset()
    write_planes(sh["planes"][i], env)


def build_env(spec: dict, index: int):
    """spec: {"module", "class", "kwargs", "seed_kw" (name of the per-environment seed argument or None), "seed0", "seed_step"}."""
    import importlib

    cls = getattr(importlib.import_module(spec["module"]), spec["class"])
    kw = dict(spec.get("kwargs", {}))
    if spec.get("seed_kw"):
        kw[spec["seed_kw"]] = int(spec.get("seed0", 0)) + int(spec.get("seed_step", 1)) * index
    return cls(*spec.get("args", []), **kw)


class VectorEnv:
    def __init__(self, envs=None, *, make_env: dict | None = None, n_envs: int | None = None, n_workers: int = 0,
                 horizon: int = 1 << 30):
        self.horizon = int(horizon)
        self._workers = []
        self._shm = None
        self._pending = False
        self._pinned = False
        if envs is not None:
            self.envs = list(envs)
            assert len(self.envs) >= 1
            self.n = len(self.envs)
            e = self.envs[0]
            self.n_actions = e.n_actions
            self.state_height, self.state_width, self.n_stacked_frames = e.state_height, e.state_width, e.n_stacked_frames
            hw = self.state_height * self.state_width
            self._buf = bytearray(shared_layout(self.n, self.n_stacked_frames, hw)["total"])
            self._sh = map_shared(self._buf, self.n, self.n_stacked_frames, hw)
            return
        assert make_env is not None and n_envs is not None and n_envs >= 1
        self.envs = None
        self.n = int(n_envs)
        probe = build_env(make_env, 0)  # (geometry only; the workers build their own instances)
        self.n_actions = probe.n_actions
        self.state_height, self.state_width, self.n_stacked_frames = probe.state_height, probe.state_width, probe.n_stacked_frames
        del probe
        hw = self.state_height * self.state_width
        n_workers = max(1, min(int(n_workers), self.n))
        self._shm = shared_memory.SharedMemory(create=True, size=shared_layout(self.n, self.n_stacked_frames, hw)["total"])
        self._sh = map_shared(self._shm.buf, self.n, self.n_stacked_frames, hw)
        bounds = [round(w * self.n / n_workers) for w in range(n_workers + 1)]
        pkg = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        env = dict(os.environ, PYTHONPATH=pkg + os.pathsep + os.environ.get("PYTHONPATH", ""),
                   OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
        env["HIP_VISIBLE_DEVICES"] = env["CUDA_VISIBLE_DEVICES"] = ""  # belt and braces: a worker has no business on a GPU
        for w in range(n_workers):
            cfg = dict(shm=self._shm.name, n=self.n, stack=self.n_stacked_frames, hw=hw, lo=bounds[w], hi=bounds[w + 1],
                       horizon=self.horizon, spec=make_env)
            p = subprocess.Popen([sys.executable, "-m", "slimdqn.environments._worker", json.dumps(cfg)], stdin=subprocess.PIPE,
                                 stdout=subprocess.PIPE, env=env, bufsize=0)
            self._workers.append(p)
        for p in self._workers:  # every worker built its environments and mapped the block
            self._expect(p, b"R")

    # ------------------------------------------------------------------ plumbing
    WORKER_TIMEOUT_S = float(os.environ.get("ISDQN_WORKER_TIMEOUT_S", "300"))  # one command of one worker (a hung emulator must not hang the trainer)

    @classmethod
    def _expect(cls, p, what: bytes) -> None:
        ready, _, _ = select.select([p.stdout], [], [], cls.WORKER_TIMEOUT_S)
        if not ready:
            p.kill()
            raise RuntimeError(f"environment worker (pid {p.pid}) did not answer {what!r} within {cls.WORKER_TIMEOUT_S:.0f} s: killed "
                               f"(its stderr is this process's stderr)")
        got = p.stdout.read(1)
        if got != what:
            try:
                rc = p.wait(timeout=5)
            except subprocess.TimeoutExpired:
                rc = None
            raise RuntimeError(f"environment worker (pid {p.pid}) died or answered {got!r} instead of {what!r} (exit code {rc}; "
                               f"its stderr is this process's stderr)")

    def _command(self, c: bytes) -> None:
        for p in self._workers:
            p.stdin.write(c)
        for p in self._workers:
            self._expect(p, c)

    def __len__(self) -> int:
        return self.n

    @property
    def n_workers(self) -> int:
        return len(self._workers)

    def close(self) -> None:
        for p in self._workers:
            try:
                p.stdin.write(b"Q")
                p.stdin.close()
            except (BrokenPipeError, OSError, ValueError):
                pass
        for p in self._workers:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()  # (this exact child)
        self._workers = []
        if self._shm is not None:
            self.unpin()
            self._sh = None
            try:
                self._shm.close()
                self._shm.unlink()
            except (BufferError, FileNotFoundError):
                pass
            self._shm = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ environment surface
    def reset(self) -> None:
        assert not self._pending, "step_wait() first"
        if self.envs is not None:
            for i, e in enumerate(self.envs):
                e.reset()
                write_planes(self._sh["planes"][i], e)
        else:
            self._command(b"Z")

    def step_async(self, actions) -> None:
        assert not self._pending, "one step in flight at a time"
        self._sh["actions"][:] = np.asarray(actions, dtype=np.int64)
        self._pending = True
        if self.envs is None:
            for p in self._workers:
                p.stdin.write(b"S")

    def step_wait(self):
        """(observation before the step uint8 [n][h][w], reward f64 [n], absorbing bool [n], episode_end bool [n]); the
        arrays are views of the shared block, valid until the next step_async."""
        assert self._pending, "no step in flight"
        if self.envs is not None:
            for i, e in enumerate(self.envs):
                step_one(e, int(self._sh["actions"][i]), self.horizon, i, self._sh)
        else:
            for p in self._workers:
                self._expect(p, b"S")
        self._pending = False
        sh = self._sh
        return (sh["obs"].reshape(self.n, self.state_height, self.state_width), sh["reward"], sh["absorbing"].astype(bool),
                sh["episode_end"].astype(bool))

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    @property
    def planes(self) -> np.ndarray:
        """uint8 [n][stack][h*w]: every environment's current frame stack, planar, oldest .. newest (a view of the block)."""
        return self._sh["planes"]

    @property
    def states(self) -> np.ndarray:
        """(n, h, w, stack) uint8: the frame stacks every environment would hand to ``best_action`` (a copy)."""
        return np.ascontiguousarray(np.moveaxis(self.planes.reshape(self.n, self.n_stacked_frames, self.state_height, self.state_width), 1, -1))

    # ------------------------------------------------------------------ pinned mapping (device side of acting)
    def pin(self) -> bool:
        """Register the block with the HIP runtime so that the per-round upload of ``planes`` is one DMA from pinned memory.
        Returns False (and changes nothing) where registration is refused; the upload is then an ordinary pageable copy."""
        if self._pinned or self._shm is None:
            return self._pinned
        import torch

        addr = int(self._sh["planes"].ctypes.data)  # (the planes section starts the block)
        rc = torch.cuda.cudart().cudaHostRegister(addr, self._shm.size, 0)
        self._pinned = int(rc) == 0
        self._pin_addr = addr
        return self._pinned

    def unpin(self) -> None:
        if self._pinned:
            import torch

            try:
                torch.cuda.cudart().cudaHostUnregister(self._pin_addr)
            except Exception:
                pass
            self._pinned = False
