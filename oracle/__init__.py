"""oracle -- CPU restatement of the reference hot path.  TEST INFRASTRUCTURE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package.  The product (``stanford-ctc_amd/``) never does.

* :mod:`oracle.ctc` -- ctypes binding of ``ctc_ref.c`` (float64 C restatement of
  ``ctc_fast/ctc-loss/ctc_fast.pyx:13-187``).
* :mod:`oracle.brnn` -- NumPy float64 restatement of the BRNN forward/backward
  (``ctc_fast/debug-utils/rnnetcpu.py:54-150`` + the clip / mask / L2 deltas of
  ``ctc_fast/nnets/brnnet.py:117-249``).

Parity pinning: both are checked in ``tests/test_oracle_golden.py`` against
golden vectors produced by the reference's own code (``tests/golden/``).
"""
