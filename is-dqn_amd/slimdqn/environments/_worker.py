"""Environment worker process of VectorEnv (vector.py): owns environments [lo, hi) of a shared-memory block and steps them on
command.  Started as ``python -m slimdqn.environments._worker '<json config>'``; imports numpy and the environment module only
(never torch, never the GPU).  Protocol on stdin and on a PRIVATE duplicate of the stdout pipe (file descriptor 1 itself is pointed at
stderr before the environment module is imported, so nothing the emulator stack prints -- ALE / gymnasium banners, warnings -- can land
in the byte stream), one byte each way per command:
    S  step every owned environment with ``actions[i]`` (utils.py:28-42 around env.step, reset at episode end)
    Z  reset every owned environment
    Q  (or end of file) leave
and ``R`` once after start-up."""
import json
import os
import sys
from multiprocessing import shared_memory


def main() -> int:
    cfg = json.loads(sys.argv[1])
    # the protocol keeps its own descriptor of the pipe to the parent; descriptor 1 and sys.stdout go to stderr from here on
    out = os.fdopen(os.dup(1), "wb", buffering=0)
    sys.stdout.flush()
    os.dup2(2, 1)
    sys.stdout = sys.stderr
    from slimdqn.environments.vector import build_env, map_shared, step_one, write_planes

    shm = shared_memory.SharedMemory(name=cfg["shm"])
    try:
        # the parent owns the block: keep this process's resource tracker from unlinking it when the worker leaves
        from multiprocessing import resource_tracker

        resource_tracker.unregister(shm._name, "shared_memory")
    except Exception:
        pass
    sh = map_shared(shm.buf, cfg["n"], cfg["stack"], cfg["hw"])
    lo, hi, horizon = cfg["lo"], cfg["hi"], cfg["horizon"]
    envs = [build_env(cfg["spec"], i) for i in range(lo, hi)]
    for k, e in enumerate(envs):
        e.reset()
        write_planes(sh["planes"][lo + k], e)
    inp = sys.stdin.buffer
    out.write(b"R")
    while True:
        c = inp.read(1)
        if c == b"S":
            actions = sh["actions"]
            for k, e in enumerate(envs):
                step_one(e, int(actions[lo + k]), horizon, lo + k, sh)
        elif c == b"Z":
            for k, e in enumerate(envs):
                e.reset()
                write_planes(sh["planes"][lo + k], e)
        else:  # b"Q" or the parent went away
            break
        out.write(c)
    del sh
    try:
        shm.close()
    except BufferError:
        pass
    return 0


if __name__ == "__main__":
    sys.exit(main())
