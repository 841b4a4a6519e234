"""Build libisdqn_hip.so (the C-ABI HIP library) in-tree for gfx950.

    python is-dqn_amd/build.py [--force]

hipcc cross-compiles without a GPU; the .so is git-ignored but travels with gpurun snapshots.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libisdqn_hip.so")
SOURCES = ["api.hip", "tree_kernels.hip", "replay_kernels.hip", "net_kernels.hip"]
HEADERS = sorted(f for f in os.listdir(os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")) if f.endswith(".h")) + [
    os.path.join("..", "..", "include", "isdqn_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-fno-gpu-rdc", "-ffp-contract=off"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    procs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            cmd = [hipcc] + FLAGS + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    failed = [src for src, p in procs if p.wait() != 0]
    if failed:
        raise RuntimeError(f"hipcc failed for {failed}")
    if force or procs or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
