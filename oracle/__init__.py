"""CPU oracle for the MVIN hot path -- TEST INFRASTRUCTURE ONLY.

PIN LEVEL: wiring pinned to the reference's own code, TensorFlow's arithmetic unpinned.
The reference (johnnyjana730/MVIN) ships no tests, golden vectors or recorded
outputs for this path, and its implementation needs TensorFlow 1.x which cannot
run in this image.  What can be run is run: tests/golden/make_ref_fixtures.py
executes the reference's model.py / aggregators.py / util.py / train.py UNMODIFIED
over a numpy stand-in for the TF1 symbols they use (tests/refpin/tf1_standin.py)
and its TF-free modules as they are, and the fixtures under tests/golden/ref/ pin
every restatement below (tests/test_ref_pins.py: fp64 to 1e-9 relative, fp32 to
1e-5*|ref|+1e-6, level ids / CSR / metrics / early-stop / ablation table exactly).
That removes the shared-misreading risk; it does not certify TF's own kernels
(each op of those fixtures is numpy's).  The oracle is two independent
restatements of the reference's graph, cross-checked against each other, against
those fixtures and against analytic known-answer cases:

* ``oracle.mirror_fp32``  - op-by-op torch-CPU fp32 mirror of the TF graph
  (same op order and the same materialised intermediates as
  src/model/MVIN/model.py and aggregators.py).  It doubles as the timed
  "TF-graph-equivalent CPU restatement" baseline in bench.py.
* ``oracle.equations_fp64`` - from-the-equations numpy fp64 version written
  per (user,item) pair with explicit tree recursion.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  Nothing under mvin_amd/ imports it.
"""
