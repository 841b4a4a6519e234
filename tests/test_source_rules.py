"""Static guards for rules of the HIP sources that cannot be checked by a small numerical test on the CPU.

The first rule is the fix for rare wrong tiles at full batch size on gfx950: every MFMA goes
through `mfma_acc` (tied-operand inline asm, in-place accumulation); the compiler builtin lets hipcc rename
accumulators into a dependent-MFMA sequence that is not run-to-run stable.  The dynamic check is
tests/test_gpu_fullsize_properties.py on the GPU; this one catches a regression before it gets there."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "is-dqn_amd", "csrc")


def _sources():
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".h", ".hip")):
            yield f, open(os.path.join(CSRC, f)).read()


def _code(text):
    """source without // comments (the rule is explained in comments that name the builtin)"""
    return "\n".join(line.split("//")[0] for line in text.split("\n"))


def test_every_mfma_goes_through_the_in_place_asm_helper():
    uses = 0
    for name, text in _sources():
        code = _code(text)
        assert "__builtin_amdgcn_mfma" not in code, f"{name}: use mfma_acc(), not the MFMA builtin (DESIGN.md section 5)"
        uses += len(re.findall(r"\bmfma_acc\(", code))
    assert uses >= 10  # gemm engine, three conv kernels, head chain


def test_kernels_with_mfma_loops_pad_before_reading_accumulators():
    for name, text in _sources():
        code = _code(text)
        if re.search(r"\bmfma_acc\(", code) and name != "gemm_core.h":
            assert "mfma_drain(" in code, f"{name}: MFMA loop without mfma_drain()"
            assert "mfma_init(" in code or "s_nop" in code, f"{name}: accumulators initialised without padding"


def test_no_cpu_fallback_and_no_foreign_backends_in_the_product_sources():
    for name, text in _sources():
        code = _code(text)
        for banned in ("__HIP_PLATFORM_AMD__", "__CUDACC__", "cuda_runtime", "hipify"):
            assert banned not in code, f"{name}: {banned}"
