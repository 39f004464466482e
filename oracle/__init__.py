"""CPU oracle for the MVIN hot path -- TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the reference (johnnyjana730/MVIN) ships no tests, golden
vectors or recorded outputs for this path, and its implementation needs
TensorFlow 1.x which cannot run in this image.  The oracle is therefore two
independent restatements of the reference's graph, cross-checked against each
other and against analytic known-answer cases:

* ``oracle.mirror_fp32``  - op-by-op torch-CPU fp32 mirror of the TF graph
  (same op order and the same materialised intermediates as
  src/model/MVIN/model.py and aggregators.py).  It doubles as the timed
  "TF-graph-equivalent CPU restatement" baseline in bench.py.
* ``oracle.equations_fp64`` - from-the-equations numpy fp64 version written
  per (user,item) pair with explicit tree recursion.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  Nothing under mvin_amd/ imports it.
"""
