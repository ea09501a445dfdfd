"""CPU oracle for the DenseFusion hot path -- TEST INFRASTRUCTURE ONLY.

Everything under ``oracle/`` is a CPU restatement of the reference algorithm (each function
cites the reference file:line it follows).  It exists to *check* the HIP path: only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.  The product package ``densefusion_amd`` never imports it and has no CPU fallback.

Parity pinning: the restatement is checked in ``tests/test_oracle_golden.py`` against golden
vectors produced by importing the reference's own Python modules in the build container
(``oracle/make_golden.py``; fixtures in ``tests/golden/``), against the Gohlke doctest values
for the two quaternion functions (lib/transformations.py:1257-1265,1287-1306) and against the
two PLY clouds the reference ships (ADD / ADD-S values).  The CUDA 1-NN op (lib/knn) cannot be
built here (needs nvcc + THC + torch-0.4 cffi): its restatement is pinned only through
lib/nn.py's ``nn_distance`` on small sizes and the eval_linemod call-site semantics.
"""
