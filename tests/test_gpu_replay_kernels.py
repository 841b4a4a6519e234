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


def _pack_staged(sections, align):
    """Lay the named byte sections out one after another, each start rounded up to `align`; returns (bytes, offsets)."""
    buf, offs = bytearray(), {}
    for name, arr in sections:
        while len(buf) % align:
            buf.append(0xEE)
        offs[name] = len(buf)
        buf += np.ascontiguousarray(arr).tobytes()
    return np.frombuffer(bytes(buf), np.uint8).copy(), offs


@pytest.mark.parametrize("frame_bytes,align,shift", [(84 * 84, 16, 0), (84 * 84, 4, 0), (84 * 84 + 3, 16, 0), (52 * 60, 4, 4), (17, 4, 0)],
                         ids=["16B-aligned", "4B-aligned-sections", "odd-frame-size", "shifted-base", "tiny-frames"])
def test_apply_staged_equals_plain_indexed_writes(frame_bytes, align, shift):
    """isdqn_replay_apply_staged (ReplayBuffer.add's device side, replay_buffer.py:185-196) against numpy indexed writes:
    frames, element rows and index -> slot pairs of ONE flush, with wrap-around (evicting) slots, frame sizes / section
    offsets / bases that are not multiples of 16 (the uint4 fast path must fall back to bytes), and untouched neighbours."""
    from slimdqn import _hip

    lib = _hip.lib()
    rng = np.random.default_rng(frame_bytes + align + shift)
    C, n_slots, stack2 = 64, 80, 8
    stride = frame_bytes if frame_bytes % 16 else frame_bytes + 16 * (align == 16)  # (a padded stride too)
    frames0 = rng.integers(0, 256, (n_slots, stride), dtype=np.uint8)
    ef0 = rng.integers(-1, n_slots, (C, stack2)).astype(np.int32)
    ea0, er0 = rng.integers(0, 18, C).astype(np.int32), rng.normal(size=C).astype(np.float32)
    et0, i2s0 = rng.integers(0, 2, C).astype(np.uint8), rng.permutation(C + 1).astype(np.int32)
    # one flush: 7 new frames (two of them re-using evicted slots at the wrap-around), 5 rows, 6 index pairs
    f_slots = np.array([78, 79, 0, 1, 40, 41, 2], np.int32)
    f_data = rng.integers(0, 256, (len(f_slots), frame_bytes), dtype=np.uint8)
    rows = np.array([62, 63, 0, 1, 30], np.int32)
    r_frames = rng.integers(-1, n_slots, (len(rows), stack2)).astype(np.int32)
    r_action, r_reward = rng.integers(0, 18, len(rows)).astype(np.int32), rng.normal(size=len(rows)).astype(np.float32)
    r_term = rng.integers(0, 2, len(rows)).astype(np.uint8)
    i_rows, i_vals = np.array([0, 5, 64, 63, 17, 1], np.int32), rng.integers(0, C, 6).astype(np.int32)
    staged, o = _pack_staged([("slots", f_slots), ("data", f_data), ("rows", rows), ("rf", r_frames), ("ra", r_action), ("rr", r_reward),
                              ("rt", r_term), ("ir", i_rows), ("iv", i_vals)], align)
    u = _hip.StagedUpdates(len(f_slots), frame_bytes, o["slots"], o["data"], len(rows), stack2, o["rows"], o["rf"], o["ra"], o["rr"], o["rt"],
                           len(i_rows), 0, o["ir"], o["iv"])
    d = lambda a: torch.from_numpy(a).cuda()
    st_full = torch.zeros(shift + staged.size, dtype=torch.uint8, device="cuda")
    st_full[shift:] = d(staged)
    st = st_full[shift:]
    fr, ef, ea, er, et, i2s = d(frames0.copy()), d(ef0.copy()), d(ea0.copy()), d(er0.copy()), d(et0.copy()), d(i2s0.copy())
    import ctypes
    _hip.check(lib.isdqn_replay_apply_staged(_hip.ptr(st), ctypes.byref(u), _hip.ptr(fr), stride, _hip.ptr(ef), _hip.ptr(ea), _hip.ptr(er),
                                             _hip.ptr(et), _hip.ptr(i2s), _hip.stream_ptr()))
    frames0[f_slots, :frame_bytes] = f_data
    ef0[rows], ea0[rows], er0[rows], et0[rows] = r_frames, r_action, r_reward, r_term
    i2s0[i_rows] = i_vals
    np.testing.assert_array_equal(fr.cpu().numpy(), frames0)   # (bytes between frame_bytes and the stride untouched too)
    np.testing.assert_array_equal(ef.cpu().numpy(), ef0)
    np.testing.assert_array_equal(ea.cpu().numpy(), ea0)
    np.testing.assert_array_equal(er.cpu().numpy(), er0)
    np.testing.assert_array_equal(et.cpu().numpy(), et0)
    np.testing.assert_array_equal(i2s.cpu().numpy(), i2s0)


def test_apply_staged_rejects_misaligned_int_sections():
    from slimdqn import _hip
    import ctypes

    lib = _hip.lib()
    st = torch.zeros(256, dtype=torch.uint8, device="cuda")
    buf = torch.zeros(64, dtype=torch.int32, device="cuda")
    u = _hip.StagedUpdates(0, 0, 0, 0, 1, 8, 2, 8, 48, 52, 56, 0, 0, 60, 64)   # off_rows = 2
    rc = lib.isdqn_replay_apply_staged(_hip.ptr(st), ctypes.byref(u), 0, 0, _hip.ptr(buf), _hip.ptr(buf), _hip.ptr(buf), _hip.ptr(buf), 0,
                                       _hip.stream_ptr())
    assert rc != 0
