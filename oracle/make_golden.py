"""Generate tests/golden/sum_tree.npz from the REFERENCE implementation.

Runs only in the build container (needs /root/reference).  The reference file
``slimdqn/sample_collection/sum_tree.py`` is numpy-only and is loaded by path
under a private module name; nothing of it is copied into this repository --
only inputs (tests/sumtree_cases.py, seeded) and its outputs are stored.

Usage:  python oracle/make_golden.py
"""
import hashlib
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.sumtree_cases import all_cases, replay  # noqa: E402

REF = "/root/reference/slimdqn/sample_collection/sum_tree.py"


def load_reference():
    spec = importlib.util.spec_from_file_location("_reference_sum_tree", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.SumTree


def main():
    SumTree = load_reference()
    out = {}
    for name, capacity, ops in all_cases():
        tree = SumTree(capacity)
        results = replay(tree, ops)
        nodes = tree._nodes
        out[f"{name}/depth"] = np.int64(tree._depth)
        out[f"{name}/first_leaf_offset"] = np.int64(tree._first_leaf_offset)
        out[f"{name}/n_nodes"] = np.int64(nodes.size)
        out[f"{name}/root"] = np.float64(tree.root)
        out[f"{name}/max_recorded_priority"] = np.float64(tree.max_recorded_priority)
        out[f"{name}/nodes_sha256"] = np.frombuffer(hashlib.sha256(nodes.tobytes()).digest(), dtype=np.uint8)
        if nodes.size <= 2047:
            out[f"{name}/nodes"] = nodes.copy()
        else:  # top of the tree (shared ancestors: where add order matters most)
            out[f"{name}/nodes_top"] = nodes[:1023].copy()
        for i, r in enumerate(results):
            out[f"{name}/query{i}"] = r
        out[f"{name}/n_queries"] = np.int64(len(results))
    path = os.path.join(ROOT, "tests", "golden", "sum_tree.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path)} bytes")


if __name__ == "__main__":
    main()
