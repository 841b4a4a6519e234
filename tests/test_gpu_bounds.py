"""Memory safety of the operand loaders (the class of round 2's abort: a tile-rounded load past the end of a caller tensor whose
value is never used -- right results, and a page fault only when that tensor ends a mapped segment).  A second build of the library
(-DISDQN_BOUNDS, csrc/gemm_core.h: ISDQN_BOUNDS_CHECK) checks every global load of the tile engine's loaders, the image fills, the
frame-id table, the prefetching epilogues and the head chain against the exact byte extents of the tensors the caller passes;
scripts/bounds_check.py drives every learn / loss / forward / acting / gradient-only entry point through it over the suite's shapes
(ragged batches, the LunarLander fc plan, both full-size BASELINE shapes, impala, BatchNorm).  No load may fall outside; a control
case with one tensor left unregistered must be caught."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_operand_load_leaves_the_tensors_it_was_given():
    sys.path.insert(0, os.path.join(ROOT, "is-dqn_amd"))
    import build

    lib = build.build(verbose=False, variant="bounds", defines=("ISDQN_BOUNDS",))  # (no-op when the build is current)
    env = dict(os.environ, ISDQN_HIP_LIB=lib)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "bounds_check.py")], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    rows = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(rows) >= 20, out.stdout
    control = [r for r in rows if r["omitted"]]
    assert len(control) == 1 and control[0]["bad"] == 1 and control[0]["site"] == 20, control  # (site 20: the head chain's action load)
    for r in rows:
        if not r["omitted"]:
            assert r["bad"] == 0, f"out-of-extent load in {r['case']}: site {r['site']} at {r['addr']}"
