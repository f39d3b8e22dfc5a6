"""CPU oracle — TEST INFRASTRUCTURE ONLY.

A CPU restatement (plain C, NumPy, PyTorch-CPU fp32) of the reference's
algorithms on the hot path, each function citing the reference file:line it
follows. Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s
`cpu_baseline` leg may import this package; the product package
`openseq2seq_amd` never does (tests/test_boundary.py enforces it).
"""
