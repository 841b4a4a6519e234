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
# -fno-slp-vectorize: hipcc's SLP vectoriser turns scalar fp32 epilogue math into packed v_pk_*_f32 sequences, and one of
# them (LayerNorm backward of the 64-row DenseDgradLN tile) returned run-to-run different partial sums on gfx950 --
# the one reproducible instability behind round 1's "unstable tiles" (scripts/isa_lint.py rule R3).
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-fno-gpu-rdc", "-ffp-contract=off", "-fno-slp-vectorize"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, variant=None, defines=(), extra_flags=()):
    """`variant` / `defines`: development builds (A/B timing, hazard experiments) go to lib/libisdqn_hip_<variant>.so
    with their own object directory; the product build is the one without arguments."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(HERE, "build" if not variant else "build_" + variant)
    lib = LIB if not variant else os.path.join(LIBDIR, "libisdqn_hip_%s.so" % variant)
    flags = FLAGS + ["-D" + d for d in defines] + list(extra_flags)
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    procs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            cmd = [hipcc] + flags + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    failed = [src for src, p in procs if p.wait() != 0]
    if failed:
        raise RuntimeError(f"hipcc failed for {failed}")
    if force or procs or _stale(lib, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return lib


def emit_asm(force=False, verbose=False):
    """Device-side assembly of every source under the product flags (build/<name>.s), for scripts/isa_lint.py."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, h) for h in HEADERS]
    outs, procs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".s"))
        outs.append(o)
        if force or _stale(o, [s] + headers + [os.path.abspath(__file__)]):
            cmd = [hipcc] + [f for f in FLAGS if f != "-fPIC"] + ["-S", "--cuda-device-only", "-w", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    failed = [src for src, p in procs if p.wait() != 0]
    if failed:
        raise RuntimeError(f"hipcc -S failed for {failed}")
    return outs


if __name__ == "__main__":
    _variant = None
    _defs = []
    for _i, _a in enumerate(sys.argv):
        if _a == "--variant":
            _variant = sys.argv[_i + 1]
        if _a.startswith("-D"):
            _defs.append(_a[2:])
    _extra = [_a for _a in sys.argv[1:] if _a.startswith("-f") or _a.startswith("-m")]
    print(build(force="--force" in sys.argv, variant=_variant, defines=_defs, extra_flags=_extra))
    if "--emit-asm" in sys.argv:
        print("\n".join(emit_asm(force="--force" in sys.argv, verbose=True)))
