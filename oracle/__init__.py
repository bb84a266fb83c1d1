"""CPU restatement ("oracle") of ecSeg's metaseg hot path.  TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the product: only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` may import it, and only as the checker.  The product
(``ecseg_amd``) never imports this package and fails loudly when its HIP library is missing.

Every function cites the reference file:line (relative to UCRajkumar/ecSeg) it restates.

Parity status (see DESIGN.md "Oracle pinning"):

* tiling / stitching / quantised argmax / ``meta_inference`` / counting / overlay rows / CSV text:
  PINNED - bit-exact against golden vectors produced by running the reference's own functions
  (``tools/make_golden.py`` -> ``tests/golden/*.npz``), checked by ``tests/test_oracle_golden.py``.
* ``meta_preprocess`` / ``u16_to_u8`` (OpenCV 4.6 ``threshold(OTSU)`` / ``convertScaleAbs``) and the
  Keras forward pass (TensorFlow 2.8): PARITY UNPINNED - OpenCV and TensorFlow are third-party wheels
  that are not vendored in the reference and not installed here; their published algorithms are
  restated and anchored on the reference's call sites (src/image_tools.py:86-101, src/utils.py:115).
"""
