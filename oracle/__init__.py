"""CPU oracle for the AutoSmoothQuant W8A8 linear hot path.

THIS PACKAGE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it.  Nothing under ``autosmoothquant_amd/`` imports,
links or executes anything from here; the product path fails loudly when the
HIP library is missing instead of falling back to this code.

Parity status: PINNED.  Every function is checked bit-for-bit against the
reference's own Python hot path (imported in the build container with a
6-line ``_CUDA`` stub) by ``tests/golden/make_golden.py``; the resulting
vectors are committed under ``tests/golden/`` and replayed by
``tests/test_oracle_golden.py`` on any machine.  The native half of the
reference (``csrc/int8gemm``) is a cuBLASLt wrapper and cannot be built here
(needs the CUDA toolkit); it is an exact INT8xINT8->INT32 GEMM (alpha=1,
beta=0, CUBLAS_COMPUTE_32I: csrc/int8gemm/cublasINT8MMWrapper.cc:231,276-277),
so any exact integer GEMM is a bit-identical stand-in.  ``oracle/_ref`` is
therefore not produced ("unbuildable" in the sense of the build rules).
"""
