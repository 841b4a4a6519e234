"""GPU check of the MFMA tile engine (csrc/gemm_core.h) through isdqn_selftest_gemm: every operand
image combination (ROW = ds_read_b128 fragments, TR = ds_read_b64_tr_b16 transposing reads), both
precisions, ragged shapes and split-K, against a float64 torch matmul."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(128, 128, 64), (256, 384, 512), (100, 90, 72), (33, 257, 1000), (512, 96, 520), (16, 16, 8)]


@pytest.mark.parametrize("precision,tol", [(0, 2e-5), (1, 2e-2)])
@pytest.mark.parametrize("a_tr", [0, 1])
@pytest.mark.parametrize("b_tr", [0, 1])
def test_selftest_gemm(precision, tol, a_tr, b_tr):
    from slimdqn import _hip

    lib = _hip.lib()
    g = torch.Generator(device="cpu").manual_seed(1234 + 10 * a_tr + b_tr)
    for (M, N, K) in SHAPES:
        for split in (1, 3):
            A = torch.randn(M, K, generator=g, dtype=torch.float32)
            B = torch.randn(N, K, generator=g, dtype=torch.float32)  # asymmetric on purpose
            ref = (A.double() @ B.double().T)
            a_dev = (A.T.contiguous() if a_tr else A).cuda()
            b_dev = (B.T.contiguous() if b_tr else B).cuda()
            C = torch.zeros(split, M, N, dtype=torch.float32, device="cuda")
            _hip.check(lib.isdqn_selftest_gemm(_hip.ptr(a_dev), _hip.ptr(b_dev), _hip.ptr(C), M, N, K, a_tr, b_tr, precision, split, _hip.stream_ptr()))
            got = C.sum(0).double().cpu()
            scale = (A.abs().double() @ B.abs().double().T)
            err = ((got - ref).abs() / scale.clamp_min(1e-6)).max().item()
            assert err < tol, f"M{M} N{N} K{K} a_tr{a_tr} b_tr{b_tr} split{split} precision{precision}: rel err {err}"
