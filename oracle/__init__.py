"""CPU oracle for the HSE_FaceRec_tf feature-extract hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in ``hse_facerec_tf_amd`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do, and only as the checker.

PARITY UNPINNED: the arithmetic of the reference path lives in TensorFlow 1.x
(``tf.Session.run`` at facerec_test.py:120 and facial_analysis.py:109), which is
not vendored in the reference and not installable here; the reference has no
tests, golden vectors or known-answer fixtures for this path.  This package is a
restatement of TensorFlow's *published* op semantics executed over the
reference's own frozen graph, cross-checked against an independent torch-CPU
lowering (``oracle/torch_cpu.py``) -- not against TensorFlow outputs.
"""
