"""Run the learn / loss / forward / acting / gradient-only entry points of a bounds-checked build (-DISDQN_BOUNDS: every operand load
of the tile engine, the image kernels, the frame-id table and the head chain is checked on the device against the byte extents
registered here -- the EXACT extents of every tensor handed to the library) over the shapes the GPU suite uses, and print one JSON
line per case: {"case", "bad", "site", "addr"}.  Started by tests/test_gpu_bounds.py with ISDQN_HIP_LIB pointing at that build."""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "is-dqn_amd"))
import numpy as np
import torch

from slimdqn import _hip
from slimdqn._engine import QNetEngine

lib = _hip.lib()
lib.isdqn_debug_bounds_set.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32]
lib.isdqn_debug_bounds_get.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]


_first = [None]  # first violation of the running case (a re-registration clears the device record)


def register(tensors, fresh=True):
    if fresh:
        _first[0] = None
    else:
        r = result()
        if r[0] and _first[0] is None:
            _first[0] = r
    ts = [t for t in tensors if t is not None]
    lo = (ctypes.c_uint64 * len(ts))(*[t.data_ptr() for t in ts])
    hi = (ctypes.c_uint64 * len(ts))(*[t.data_ptr() + t.numel() * t.element_size() for t in ts])
    _hip.check(lib.isdqn_debug_bounds_set(lo, hi, len(ts)))


def result():
    bad, site, addr = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_uint64()
    _hip.check(lib.isdqn_debug_bounds_get(ctypes.byref(bad), ctypes.byref(site), ctypes.byref(addr)))
    return int(bad.value), int(site.value), int(addr.value)


def engine_tensors(eng):
    return [eng.params, eng.adam_m, eng.adam_v, eng.workspace, eng.losses, eng.losses_accum, eng.q_values, eng.targets, eng.priorities,
            eng.adam_count, eng.action_out]


def report(case, omit=None):
    bad, site, addr = _first[0] if _first[0] is not None else result()
    print(json.dumps({"case": case, "bad": bad, "site": site, "addr": hex(addr), "omitted": omit}), flush=True)


def run_case(name, arch, obs, feats, K, A, B, ln=True, bn=False, target=False, grad=True, omit=None):
    n_heads = 1 + K if K > 0 else 1
    eng = QNetEngine(obs, A, n_heads, feats, arch, ln, B, batch_norm=bn)
    eng.init_params(0)
    rng = np.random.default_rng(0)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    action = dev(rng.integers(0, A, B).astype(np.int32))
    reward = dev(rng.normal(size=B).astype(np.float32))
    terminal = dev((rng.random(B) < 0.3).astype(np.uint8))
    extra = []
    if arch == "fc":
        d = int(np.prod(obs))
        st, nx = dev(rng.normal(size=(B, d)).astype(np.float32)), dev(rng.normal(size=(B, d)).astype(np.float32))
        batch = eng.make_batch(state=st, next_state=nx, action=action, reward=reward, terminal=terminal)
        fwd = dict(obs=torch.cat((st, nx)).contiguous(), n_rows=2 * B)
        one = dict(obs=st[:1].contiguous())
        extra += [st, nx, fwd["obs"], one["obs"]]
    else:
        h, w, stack = obs
        n_frames = 3 * B + 8
        frames = dev(rng.integers(0, 256, (n_frames, h * w), dtype=np.uint8))
        ids_np = rng.integers(0, n_frames, (B, 2 * stack)).astype(np.int32)
        ids_np[rng.random(ids_np.shape) < 0.1] = -1
        ids = dev(ids_np)
        flat = dev(np.concatenate([ids_np[:, :stack], ids_np[:, stack:]], 0))
        batch = eng.make_batch(frames=frames, frame_stride=h * w, frame_ids=ids, action=action, reward=reward, terminal=terminal)
        fwd = dict(frames=frames, frame_stride=h * w, frame_ids=flat, n_rows=2 * B)
        one = dict(frames=frames, frame_stride=h * w, frame_ids=flat[:1].contiguous())
        extra += [frames, ids, flat, one["frame_ids"]]
    heads = torch.arange(min(5, 2 * B), dtype=torch.int32, device="cuda") % max(K, 1)
    grad_out = torch.zeros_like(eng.params)
    tparams = eng.params.clone()
    small = [action, reward, terminal]
    tensors = engine_tensors(eng) + extra + [heads, grad_out, tparams] + [t for t in small if omit is None or t is not {"action": action, "reward": reward, "terminal": terminal}[omit]]
    torch.cuda.synchronize()
    register(tensors)
    q = eng.forward(**fwd)
    eng.loss_on_batch(batch)
    eng.learn_on_batch(batch)
    eng.learn_on_batch(batch, grad_out=grad_out)
    register(tensors + [q], fresh=False)  # (outputs allocated meanwhile are written, not read; the table is re-read by every kernel)
    eng.best_action(idx_network=0, **one)
    nrow = int(heads.numel())
    if arch == "fc":
        eng.best_actions(obs=fwd["obs"][:nrow].contiguous() if False else fwd["obs"], idx_networks=heads)
    else:
        eng.best_actions(frames=fwd["frames"], frame_stride=fwd["frame_stride"], frame_ids=fwd["frame_ids"], idx_networks=heads)
    if grad:  # (BatchNorm networks too: the analysis agents' gradient-only passes, with and without separate target parameters)
        eng.grad_on_batch(batch, grad_out)
        if K >= 1 and n_heads >= 2:
            eng.grad_on_batch(batch, grad_out, target_params=tparams, online_head=1, target_head=1, n_pairs=1)
    if target and not bn:
        eng.learn_on_batch_target(batch, tparams)
        eng.loss_on_batch_target(batch, tparams)
    report(name, omit)
    del eng
    torch.cuda.empty_cache()


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    HL = (32, 64, 64, 512)
    cases = [
        dict(name="cnn-headline-B8", arch="cnn", obs=(84, 84, 4), feats=HL, K=9, A=9, B=8),
        dict(name="cnn-headline-B7-ragged", arch="cnn", obs=(84, 84, 4), feats=HL, K=9, A=9, B=7),
        dict(name="cnn-tiny-noln-B5", arch="cnn", obs=(84, 84, 4), feats=(16, 20, 5, 24), K=2, A=3, B=5, ln=False),
        dict(name="cnn-a18-B33", arch="cnn", obs=(84, 84, 4), feats=HL, K=4, A=18, B=33),
        dict(name="cnn-44x44x6-generic-first-layer", arch="cnn", obs=(44, 44, 6), feats=(7, 9, 11, 13), K=2, A=3, B=5),
        dict(name="cnn-52x60x2", arch="cnn", obs=(52, 60, 2), feats=(16, 20, 12, 24), K=3, A=4, B=6),
        dict(name="cnn-one-head-dqn-B8", arch="cnn", obs=(84, 84, 4), feats=HL, K=0, A=6, B=8, target=True),
        dict(name="cnn-tiny-B515-ragged", arch="cnn", obs=(84, 84, 4), feats=(8, 8, 8, 16), K=9, A=9, B=515, grad=False),
        dict(name="cnn-c2-full-B256", arch="cnn", obs=(84, 84, 4), feats=HL, K=9, A=9, B=256, grad=False),
        dict(name="cnn-c5-full-B1024", arch="cnn", obs=(84, 84, 4), feats=HL, K=32, A=4, B=1024, grad=False),
        dict(name="fc-lunar-lander-100x100-B32", arch="fc", obs=(8,), feats=(100, 100), K=1, A=4, B=32, target=False),
        dict(name="fc-three-hidden-B10", arch="fc", obs=(6,), feats=(24, 40, 16), K=3, A=3, B=10, ln=False),
        dict(name="fc-one-head-dqn-B32", arch="fc", obs=(8,), feats=(100, 100), K=0, A=4, B=32, target=True),
        dict(name="impala-tiny-B3", arch="impala", obs=(84, 84, 4), feats=(8, 16, 16, 24), K=2, A=5, B=3),
        dict(name="impala-44x44x3-B2", arch="impala", obs=(44, 44, 3), feats=(12, 20, 9, 16), K=3, A=4, B=2),
        dict(name="bn-cnn-B6", arch="cnn", obs=(84, 84, 4), feats=(7, 9, 11, 13), K=3, A=5, B=6, bn=True),
        dict(name="bn-cnn-headline-B8", arch="cnn", obs=(84, 84, 4), feats=HL, K=9, A=9, B=8, bn=True),
        dict(name="bn-fc-B10", arch="fc", obs=(6,), feats=(24, 40, 16), K=3, A=3, B=10, ln=False, bn=True),
        dict(name="bn-impala-B4", arch="impala", obs=(36, 36, 2), feats=(16, 8, 16, 24), K=3, A=4, B=4, bn=True),
        # negative control: the checker must notice a tensor that was not registered
        dict(name="control-action-not-registered", arch="cnn", obs=(84, 84, 4), feats=(7, 9, 11, 13), K=3, A=5, B=6, omit="action"),
    ]
    for c in cases:
        if which != "all" and which not in c["name"]:
            continue
        run_case(**c)
