"""ctypes binding of libisdqn_hip.so (C ABI: include/isdqn_hip.h).

The product path has no CPU fallback: if the library is missing this module raises at first
use.  Device memory and streams come from PyTorch-ROCm (plumbing only); every hot-path
computation goes through the entry points bound here.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char, c_char_p, c_double, c_float, c_int32, c_int64, c_uint8, c_uint32, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# ISDQN_HIP_LIB: development override (A/B timing of two builds inside one gpurun call); still a HIP build, never a fallback
LIB_PATH = os.environ.get("ISDQN_HIP_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libisdqn_hip.so")

OK = 0
ERR_CAPACITY, ERR_NEGATIVE, ERR_SHAPE, ERR_EMPTY, ERR_RANGE, ERR_UNSUPPORTED, ERR_HIP, ERR_ARG = range(-1, -9, -1)
STATUS_NEGATIVE_VALUE, STATUS_TARGET_RANGE, STATUS_EMPTY_TREE = 1, 2, 4
BATCH_MIRROR_CURRENT = 1  # include/isdqn_hip.h: ISDQN_BATCH_MIRROR_CURRENT
TREE_MAX_BATCH = 4096
ARCH_CNN, ARCH_FC, ARCH_IMPALA = 0, 1, 2
PRECISION_BF16X3, PRECISION_BF16 = 0, 1
MAX_FEATURES = 8


class NetConfig(ctypes.Structure):
    _fields_ = [
        ("arch", c_int32),
        ("obs_h", c_int32),
        ("obs_w", c_int32),
        ("obs_c", c_int32),
        ("n_features", c_int32),
        ("features", c_int32 * MAX_FEATURES),
        ("n_actions", c_int32),
        ("n_heads", c_int32),
        ("layer_norm", c_int32),
        ("batch_size", c_int32),
        ("precision", c_int32),
        ("gamma_n", c_float),
        ("learning_rate", c_float),
        ("adam_b1", c_float),
        ("adam_b2", c_float),
        ("adam_eps", c_float),
        ("huber_delta", c_float),
        ("batch_norm", c_int32),
    ]


class TensorInfo(ctypes.Structure):
    _fields_ = [
        ("name", c_char * 48),
        ("offset", c_int64),
        ("size", c_int64),
        ("kind", c_int32),
        ("layer", c_int32),
        ("ndim", c_int32),
        ("flax_shape", c_int32 * 4),
        ("dims", c_int32 * 4),
    ]


class StagedUpdates(ctypes.Structure):  # isdqn_staged_updates
    _fields_ = [
        ("n_frames", c_int32), ("frame_bytes", c_int32), ("off_frame_slots", c_int64), ("off_frame_data", c_int64),
        ("n_rows", c_int32), ("stack2", c_int32), ("off_rows", c_int64), ("off_row_frames", c_int64), ("off_row_action", c_int64),
        ("off_row_reward", c_int64), ("off_row_terminal", c_int64),
        ("n_index", c_int32), ("reserved", c_int32), ("off_index_rows", c_int64), ("off_index_vals", c_int64),
    ]


class Batch(ctypes.Structure):
    _fields_ = [
        ("B", c_int32),
        ("frames", c_void_p),
        ("frame_stride", c_int64),
        ("frame_ids", c_void_p),
        ("state", c_void_p),
        ("next_state", c_void_p),
        ("action", c_void_p),
        ("reward", c_void_p),
        ("terminal", c_void_p),
        ("flags", c_int32),
        ("priorities_ready", c_void_p),
    ]


_SIGNATURES = {
    "isdqn_version": (c_char_p, []),
    "isdqn_last_error": (c_char_p, []),
    "isdqn_tree_layout": (c_int32, [c_int64, POINTER(c_int32), POINTER(c_int64), POINTER(c_int64)]),
    "isdqn_tree_set": (c_int32, [c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p]),
    "isdqn_tree_swap_remove": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "isdqn_tree_query": (c_int32, [c_void_p, c_int32, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "isdqn_replay_apply_staged": (
        c_int32,
        [c_void_p, POINTER(StagedUpdates), c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    ),
    "isdqn_replay_gather_rows": (
        c_int32,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    ),
    "isdqn_replay_materialize": (
        c_int32,
        [c_void_p, c_int64, c_int32, c_int32, c_int32, c_void_p, c_int32, c_void_p, c_void_p, c_void_p],
    ),
    "isdqn_replay_deinterleave": (
        c_int32,
        [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p],
    ),
    "isdqn_net_param_layout": (
        c_int32,
        [POINTER(NetConfig), POINTER(c_int64), POINTER(TensorInfo), c_int32, POINTER(c_int32)],
    ),
    "isdqn_net_workspace_bytes": (c_int32, [POINTER(NetConfig), POINTER(c_int64)]),
    "isdqn_net_workspace_region": (c_int32, [POINTER(NetConfig), c_char_p, POINTER(c_int64), POINTER(c_int64)]),
    "isdqn_net_forward": (
        c_int32,
        [POINTER(NetConfig), c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p],
    ),
    "isdqn_net_learn_on_batch": (
        c_int32,
        [POINTER(NetConfig), c_void_p, c_void_p, c_void_p, c_void_p, POINTER(Batch), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    ),
    "isdqn_net_learn_on_batch_debug": (
        c_int32,
        [POINTER(NetConfig), c_void_p, c_void_p, c_void_p, c_void_p, POINTER(Batch), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    ),
    "isdqn_net_loss_on_batch": (
        c_int32,
        [POINTER(NetConfig), c_void_p, POINTER(Batch), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    ),
    "isdqn_net_learn_on_batch_target": (
        c_int32,
        [POINTER(NetConfig), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, POINTER(Batch), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    ),
    "isdqn_net_loss_on_batch_target": (
        c_int32,
        [POINTER(NetConfig), c_void_p, c_void_p, POINTER(Batch), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    ),
    "isdqn_net_grad_on_batch": (
        c_int32,
        [POINTER(NetConfig), c_void_p, c_void_p, POINTER(Batch), c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    ),
    "isdqn_net_refresh_mirror": (c_int32, [POINTER(NetConfig), c_void_p, c_void_p, c_void_p]),
    "isdqn_net_bn_commit_running": (c_int32, [POINTER(NetConfig), c_void_p, c_void_p, c_void_p]),
    "isdqn_net_shift_params": (c_int32, [POINTER(NetConfig), c_void_p, c_void_p]),
    "isdqn_net_best_action": (
        c_int32,
        [POINTER(NetConfig), c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p],
    ),
    "isdqn_net_best_actions": (
        c_int32,
        [POINTER(NetConfig), c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_void_p, c_void_p],
    ),
    "isdqn_net_analysis_layout": (c_int32, [POINTER(NetConfig), POINTER(c_int32), POINTER(c_int64), c_int32]),
    "isdqn_net_analysis": (
        c_int32,
        [POINTER(NetConfig), c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p],
    ),
    "isdqn_selftest_gemm": (
        c_int32,
        [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p],
    ),
}

# entry points declared by include/isdqn_hip.h (the debug hook is exported but not part of the header)
PUBLIC_SYMBOLS = [s for s in _SIGNATURES if s != "isdqn_net_learn_on_batch_debug"]

_lib = None


def lib() -> ctypes.CDLL:
    """The loaded C-ABI library; fails loudly when the HIP extension has not been built."""
    global _lib
    if _lib is None:
        # PyTorch-ROCm bundles its own libamdhip64; import it first so that this library binds to the SAME HIP
        # runtime (streams and device pointers are shared with torch) instead of loading a second one.
        import torch  # noqa: F401

        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build the HIP extension first (python is-dqn_amd/build.py). "
                "There is no CPU fallback for the iS-DQN hot path."
            )
        handle = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = handle
    return _lib


def last_error() -> str:
    return lib().isdqn_last_error().decode()


def check(rc: int, what: str = "") -> None:
    """Map C-ABI status codes onto the reference's exception types (error conventions)."""
    if rc == OK:
        return
    msg = f"{what}: {last_error()}" if what else last_error()
    if rc in (ERR_CAPACITY, ERR_NEGATIVE, ERR_SHAPE, ERR_EMPTY):
        raise AssertionError(msg)
    if rc == ERR_RANGE:
        raise ValueError(msg)
    if rc == ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise RuntimeError(f"isdqn_hip error {rc}: {msg}")


def ptr(t) -> int:
    """Device pointer of a torch tensor (None -> NULL)."""
    return 0 if t is None else t.data_ptr()


def default_device():
    """The GPU this process drives.  One process per GPU: under torchrun / experiments/launch.py the rank's
    LOCAL_RANK selects it (a launcher that sets HIP_VISIBLE_DEVICES per child leaves exactly one visible device, cuda:0);
    a plain single-process run gets cuda:0."""
    import torch

    idx = int(os.environ.get("ISDQN_DEVICE_INDEX", os.environ.get("LOCAL_RANK", "0")))
    n = torch.cuda.device_count()
    if n and idx >= n:
        idx = 0  # the launcher narrowed HIP_VISIBLE_DEVICES to this rank's GPU
    return torch.device("cuda", idx)


def resolve_device(device):
    import torch

    dev = default_device() if device is None else torch.device(device)
    if dev.type == "cuda" and dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device() if torch.cuda.is_available() else 0)
    return dev


_CUDA_OK = None


def _cuda_ok() -> bool:
    """torch.cuda.is_available(), asked once: it is a driver call (~80 us), and the per-call device binding below sits on the
    acting / update path."""
    global _CUDA_OK
    if _CUDA_OK is None:
        import torch

        _CUDA_OK = bool(torch.cuda.is_available())
    return _CUDA_OK


def bind_device(device) -> None:
    """Make `device` the calling thread's current HIP device: the library launches on the current device (per-device
    side stream, function attributes), so an object on cuda:k must only ever be driven with cuda:k current."""
    import torch

    if device.type == "cuda" and _cuda_ok() and torch.cuda.current_device() != device.index:
        torch.cuda.set_device(device)


def stream_ptr(device=None) -> int:
    """HIP stream the C-ABI call is enqueued on: torch's current stream OF THE OBJECT'S DEVICE.  Fails loudly when the
    thread's current device is another one (kernels would be launched on device A against pointers of device B)."""
    import torch

    if device is not None and device.type == "cuda" and device.index is not None and device.index != torch.cuda.current_device():
        raise RuntimeError(
            f"isdqn_hip: object lives on {device} but the calling thread's current device is cuda:{torch.cuda.current_device()}; "
            f"one process drives one GPU -- call torch.cuda.set_device({device.index}) (experiments/launch.py does)")
    return torch.cuda.current_stream(device).cuda_stream


def require_gpu():
    import torch

    if not torch.cuda.is_available():
        raise RuntimeError("the iS-DQN HIP path needs a ROCm GPU (torch.cuda.is_available() is False)")
    lib()
