"""CPU oracle for the SAVP hot path -- TEST INFRASTRUCTURE ONLY.

This package is a from-source CPU restatement (torch-CPU / numpy, fp64-capable) of
the arithmetic that alexlee-gk/video_prediction performs on its SAVP
training/inference path.  Every function cites the reference file:line it
follows.

Rules (enforced by tests/test_abi_and_host.py::test_product_never_imports_oracle):
  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
    ``cpu_baseline`` leg may import anything from ``oracle/``;
  * the product package ``video_prediction_amd`` never imports it and has no
    CPU fallback: it raises when the HIP library is missing.

PARITY UNPINNED.  The reference's arithmetic lives in an un-vendored third-party
dependency (``tensorflow-gpu>=1.9.0``, /root/reference/requirements.txt:1) that
cannot run in this image (no TF wheel for cp310, no network), and the reference
ships no tests, golden vectors or checkpoints.  The only executable statements
of intent are two docstring identities (ops.py:652-679, ops.py:799-817); both
are reproduced in tests/test_oracle_identities.py.  TensorFlow kernel semantics
(SAME/VALID padding arithmetic, cross-correlation, conv2d_transpose as the
adjoint of conv2d, fused_batch_norm with biased variance, SYMMETRIC pad,
LSTMCell gate order, Adam epsilon placement) are restated from TF's published
definitions in ``oracle/tf_ops.py`` and pinned by naive-loop / hand-computed known
answers in tests/test_oracle_tf_semantics.py and tests/test_oracle_pins.py (SAME padding,
conv2d_transpose alignment, explicitly padded strided conv3d, depthwise order, SYMMETRIC
pad, LSTMCell gate order / forget bias, one spectral-norm step and its total gradient,
Adam's first two steps, CRC-32C / varint / TFRecord / V2-checkpoint wire formats from
their published specifications).  None of that is an output of TensorFlow itself.
"""
