"""CPU oracle for the iS-DQN hot path.  TEST INFRASTRUCTURE ONLY.

Everything under ``oracle/`` is a CPU restatement of the reference algorithm
(theovincent/iS-DQN, package ``slimdqn``) used as the *checker* for the HIP
path.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import it.  The product (``is-dqn_amd/``) never
imports this package and has no CPU fallback.

Pinning status
--------------
* ``sum_tree`` / ``samplers`` / ``replay_buffer``: pinned.  ``sum_tree`` is
  checked against golden vectors produced by importing the reference's own
  numpy-only ``slimdqn/sample_collection/sum_tree.py`` (``oracle/make_golden.py``
  -> ``tests/golden/sum_tree_*.npz``) and against every known answer of the
  reference's ``tests/test_sum_tree.py``, ``tests/test_samplers.py`` and
  ``tests/test_replay_buffer.py``.
* ``network`` / ``isdqn`` (Conv/LayerNorm/Dense/Adam numerics): **parity
  unpinned**.  The arithmetic lives in flax==0.10.2 / jax==0.4.30 /
  optax==0.2.4, none of which is installed here and the reference tests hold
  no golden numbers for it (``tests/test_isdqn.py`` only compares the agent
  with an inline restatement that calls the same Flax network).  The
  restatement follows the documented defaults of those versions and is
  cross-checked against an independent plain-numpy im2col statement.
"""
