"""GPU check of the byte-moving replay kernels (row gather, stack materialise, de-interleave): bit-exact."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_gather_materialize_deinterleave_round_trip():
    from slimdqn import _hip

    lib = _hip.lib()
    rng = np.random.default_rng(0)
    C, B, stack, h, w = 500, 37, 4, 84, 84
    n_frames = 600
    frames = rng.integers(0, 256, (n_frames, h * w), dtype=np.uint8)
    ef = rng.integers(-1, n_frames, (C, 2 * stack)).astype(np.int32)
    ea = rng.integers(0, 18, C).astype(np.int32)
    er = rng.normal(size=C).astype(np.float32)
    et = rng.integers(0, 2, C).astype(np.uint8)
    slots = rng.integers(0, C, B).astype(np.int32)
    d = lambda a: torch.from_numpy(a).cuda()
    fr, efd, ead, erd, etd, sl = d(frames), d(ef), d(ea), d(er), d(et), d(slots)
    ids = torch.empty(B, 2 * stack, dtype=torch.int32, device="cuda")
    act = torch.empty(B, dtype=torch.int32, device="cuda")
    rew = torch.empty(B, dtype=torch.float32, device="cuda")
    ter = torch.empty(B, dtype=torch.uint8, device="cuda")
    _hip.check(lib.isdqn_replay_gather_rows(_hip.ptr(efd), _hip.ptr(ead), _hip.ptr(erd), _hip.ptr(etd), stack, 0, _hip.ptr(sl), B,
                                            _hip.ptr(ids), _hip.ptr(act), _hip.ptr(rew), _hip.ptr(ter), _hip.stream_ptr()))
    np.testing.assert_array_equal(ids.cpu().numpy(), ef[slots])
    np.testing.assert_array_equal(act.cpu().numpy(), ea[slots])
    np.testing.assert_array_equal(rew.cpu().numpy(), er[slots])
    np.testing.assert_array_equal(ter.cpu().numpy(), et[slots])

    st = torch.empty(B, h, w, stack, dtype=torch.uint8, device="cuda")
    nx = torch.empty_like(st)
    _hip.check(lib.isdqn_replay_materialize(_hip.ptr(fr), h * w, h, w, stack, _hip.ptr(ids), B, _hip.ptr(st), _hip.ptr(nx), _hip.stream_ptr()))
    exp = np.zeros((B, 2, h, w, stack), np.uint8)
    for b in range(B):
        for c in range(2 * stack):
            if ef[slots[b], c] >= 0:
                exp[b, c // stack, :, :, c % stack] = frames[ef[slots[b], c]].reshape(h, w)
    np.testing.assert_array_equal(st.cpu().numpy(), exp[:, 0])
    np.testing.assert_array_equal(nx.cpu().numpy(), exp[:, 1])

    planes = torch.empty(2 * B * stack, h * w, dtype=torch.uint8, device="cuda")
    ids2 = torch.empty(B, 2 * stack, dtype=torch.int32, device="cuda")
    _hip.check(lib.isdqn_replay_deinterleave(_hip.ptr(st), _hip.ptr(nx), h, w, stack, B, _hip.ptr(planes), _hip.ptr(ids2), _hip.stream_ptr()))
    st2 = torch.empty_like(st)
    nx2 = torch.empty_like(st)
    _hip.check(lib.isdqn_replay_materialize(_hip.ptr(planes), h * w, h, w, stack, _hip.ptr(ids2), B, _hip.ptr(st2), _hip.ptr(nx2), _hip.stream_ptr()))
    np.testing.assert_array_equal(st2.cpu().numpy(), exp[:, 0])
    np.testing.assert_array_equal(nx2.cpu().numpy(), exp[:, 1])
