"""CPU oracle for the franQ SAC/TQC update path.  TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement of the reference's algorithm for the hot path
(replay ring -> windowed minibatch -> SAC/TQC loss -> backward -> Adam -> polyak).
It is the *checker*, never the product:

  * only tests/, __graft_entry__.smoke() and bench.py's ``cpu_baseline`` leg may import it;
  * nothing under fastdeepqlearning_amd/ imports it, and the product path raises when the
    HIP library is missing instead of falling back to this code.

Parity pin: every function here is checked against golden vectors produced by running the
reference itself in the build container (tests/golden/make_golden.py ->
tests/golden/*.npz; see tests/test_oracle_vs_golden.py).  HER-vmap (her_vmap.py /
nstep_return_vmap.py) is "shim-pinned": those files need jax, which the image lacks, so
they ran on a numpy-backed stand-in of the few jax entry points they use
(tests/golden/_refimport.py); see DESIGN.md.
"""
