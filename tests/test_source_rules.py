"""Static guards for rules of the HIP sources that cannot be checked by a small numerical test on the CPU.

Round 2: the MFMAs go through the compiler BUILTIN (hipcc then pads every MFMA dependency
itself), no MFMA may hide in inline asm, and the library is built without the SLP vectoriser, whose shuffled packed-fp32
sequences were the one reproducible source of run-to-run different results on gfx950.  `scripts/isa_lint.py` checks
all of that on the assembly hipcc emits under the product flags; the dynamic check is
tests/test_gpu_fullsize_properties.py on the GPU."""
import importlib.util
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "is-dqn_amd", "csrc")


def _sources():
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".h", ".hip")):
            yield f, open(os.path.join(CSRC, f)).read()


def _code(text):
    """source without // comments"""
    return "\n".join(line.split("//")[0] for line in text.split("\n"))


def _build_module():
    spec = importlib.util.spec_from_file_location("isdqn_build", os.path.join(ROOT, "is-dqn_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_every_mfma_goes_through_the_builtin_helper():
    uses = 0
    for name, text in _sources():
        code = _code(text)
        assert not re.search(r"asm[^;]*v_mfma", code, re.S), f"{name}: MFMA in inline asm (hipcc cannot pad its hazards)"
        if name != "gemm_core.h":
            assert "__builtin_amdgcn_mfma" not in code, f"{name}: use mfma_acc() from gemm_core.h"
        uses += len(re.findall(r"\bmfma_acc\(", code))
    assert uses >= 10  # gemm engine, three conv kernels, head chain


def test_product_flags_disable_the_slp_vectoriser():
    assert "-fno-slp-vectorize" in _build_module().FLAGS
    assert "-ffp-contract=off" in _build_module().FLAGS


def test_isa_lint_is_clean_on_the_product_assembly():
    """hipcc -S under the product flags (cross-compiles without a GPU, cached by mtime), then scripts/isa_lint.py:
    no asm MFMA, the gfx950 MFMA wait-state table holds (back-edges included), no shuffled packed fp32."""
    if not os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")):
        pytest.skip("hipcc not available")
    files = _build_module().emit_asm()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "isa_lint.py")] + files, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-4000:]
    assert re.search(r"net_kernels\.s: \d+ functions, [1-9]\d* MFMA instructions", r.stdout), r.stdout[:500]


def test_isa_lint_rules_fire_on_violations(tmp_path):
    """The lint itself: an asm MFMA, a too-short MFMA -> VALU distance (also across a back-edge) and a shuffled packed
    add are each reported; the padded form is not."""
    bad = tmp_path / "bad.s"
    bad.write_text(
        "k_bad:\n"
        "\tv_add_f32_e32 v1, v2, v3\n"
        "\tv_mfma_f32_16x16x32_bf16 a[0:3], v[0:3], v[4:7], a[0:3]\n"       # v1 written one state earlier: R2
        "\ts_nop 2\n"
        "\tv_accvgpr_read_b32 v9, a0\n"                                      # 3 < 7 states: R2
        "\t;;#ASMSTART\n\tv_mfma_f32_16x16x32_bf16 a[4:7], v[0:3], v[4:7], a[4:7]\n\t;;#ASMEND\n"   # R1
        "\tv_pk_add_f32 v[10:11], v[12:13], v[10:11] op_sel:[0,1] op_sel_hi:[1,0]\n"              # R3
        "\ts_endpgm\n"
        "k_loop:\n"
        ".LBB1_1:\n"
        "\tv_accvgpr_read_b32 v9, a0\n"
        "\ts_nop 7\n"
        "\tv_mfma_f32_16x16x32_bf16 a[0:3], v[0:3], v[4:7], a[0:3]\n"
        "\ts_cbranch_scc1 .LBB1_1\n"                                          # back-edge: 1 state to the read
        "\ts_endpgm\n")
    good = tmp_path / "good.s"
    good.write_text(
        "k_good:\n"
        "\tv_add_f32_e32 v1, v2, v3\n\ts_nop 1\n"
        "\tv_mfma_f32_16x16x32_bf16 a[0:3], v[0:3], v[4:7], a[0:3]\n"
        "\tv_mfma_f32_16x16x32_bf16 a[0:3], v[0:3], v[4:7], a[0:3]\n"       # in-place chain: 0 states
        "\ts_nop 7\n"
        "\tv_accvgpr_read_b32 v9, a0\n"
        "\tv_pk_add_f32 v[10:11], v[12:13], v[10:11]\n"
        "\tv_pk_mul_f32 v[2:3], s[2:3], v[2:3] op_sel_hi:[0,1]\n"
        "\ts_endpgm\n")
    lint = os.path.join(ROOT, "scripts", "isa_lint.py")
    r = subprocess.run([sys.executable, lint, str(bad)], capture_output=True, text=True)
    assert r.returncode == 1
    for rule, n in (("R1", 1), ("R2", 3), ("R3", 1)):
        m = re.search(rule + r": (\d+) findings", r.stdout)
        assert m and int(m.group(1)) >= n, (rule, r.stdout)
    assert "across the loop back-edge" in r.stdout
    r = subprocess.run([sys.executable, lint, str(good)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout


def test_no_cpu_fallback_and_no_foreign_backends_in_the_product_sources():
    for name, text in _sources():
        code = _code(text)
        for banned in ("__HIP_PLATFORM_AMD__", "__CUDACC__", "cuda_runtime", "hipify"):
            assert banned not in code, f"{name}: {banned}"
