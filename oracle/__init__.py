"""CPU oracle for the TF-IDF -> LSI -> MOFA hot path.

TEST INFRASTRUCTURE ONLY.  This package restates, on the CPU, the arithmetic the
reference (scverse/muon 0.1.9) performs on the hot path, so that the HIP
implementation in ``muon_amd`` can be checked against it.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it; ``muon_amd`` never does (tests/test_layout.py enforces this), and the
product path raises if the HIP extension is missing instead of falling back.

Pinning status:
  * tfidf  - PINNED: reproduces every golden value of the reference's tests
             (/root/reference/tests/test_atac_preproc.py:19-20,52,63-64) and the
             fixtures in tests/golden/tfidf_golden.npz, which were produced by
             executing the reference source itself (tests/golden/make_golden.py).
  * lsi    - PINNED against tests/golden/lsi_golden.npz (reference source executed
             here); the reference has no LSI test of its own.
  * mofa   - PARITY UNPINNED: the arithmetic lives in the third-party package
             mofapy2 (not vendored, no version pin, not installable here); the
             oracle restates the published MOFA+ update equations.
"""
