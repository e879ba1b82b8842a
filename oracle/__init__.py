"""CPU oracle for the MuRCL hot path.  TEST INFRASTRUCTURE - NOT PRODUCT CODE.

This package is a CPU restatement (plain fp32 PyTorch tensor ops for the floating
point modules, numpy / Python integers for patch selection) of the reference
algorithms on the hot path named in SURVEY.md section 8.  Every function cites the
reference file:line it follows.

Rules (enforced by tests/test_no_oracle_in_product.py):
  * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it;
  * nothing under murcl_amd/ imports it, and murcl_amd has no CPU fallback: the
    product path raises when the HIP extension is missing.

Parity status: PINNED.  The reference ships no tests or golden vectors
(SURVEY.md section 4), so the oracle is pinned against outputs of the reference
itself, generated in the build container by oracle/gen_goldens.py (which imports
/root/reference) and committed as tests/golden/*.npz; tests/test_oracle_goldens.py
re-checks the oracle against them on every run.
"""
