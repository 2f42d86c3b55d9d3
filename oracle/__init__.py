"""Oracle = TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference algorithm (flkraus/bayesian-yolov3 inference path)
plus the tooling that executes the reference's own Python under an eager TensorFlow
stand-in to produce golden fixtures.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it, and only as the checker / reported CPU baseline.

PARITY STATUS: **unpinned at the TensorFlow-primitive boundary** (the reference ships no
tests, no golden vectors, and TensorFlow is not installable here).  Pinned structurally:
the reference's own graph-construction code is executed unmodified (``oracle/tf1_shim.py``)
and its pure-numpy helpers run as written; see DESIGN.md section "Oracle".
"""
