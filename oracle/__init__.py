"""CPU oracle for the GraphEcho hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A from-scratch restatement, in plain PyTorch-CPU / numpy / C, of the arithmetic that the reference's
models/*.py and utils/*.py perform (each function cites the reference file:line it follows).  It exists only
so that tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg can check and time the HIP path;
nothing under graphecho_amd/ imports it.

Parity pinning: the reference has no tests or golden vectors of its own, so the oracle is
pinned by fixtures generated here by importing the reference (tools/gen_golden.py -> tests/golden/*.npz) and
checked in tests/test_oracle_golden.py.  The functions are written functionally over a ``state_dict`` so the
same weights drive the reference, the oracle and the HIP modules.
"""
