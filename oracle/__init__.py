"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's per-frame VSR forward.
Importable only from tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke(); the product package never imports it.

Parity status: PINNED.  The restatement is checked against golden vectors
captured from the reference's own Python, imported in the development
container with I/O-only patches (oracle/ref_harness.py, oracle/make_golden.py;
fixtures under tests/golden/).  The three CUDA extensions cannot be built or
run here and the reference holds no fixtures for them: native_ops.c restates
the .cu text and is cross-checked against independent stock PyTorch ops.
"""
