"""oracle/ -- CPU restatement of tiny-audio's projector-training hot path.

TEST INFRASTRUCTURE ONLY.  This package is the *checker* for the HIP path in
``tiny_audio_amd``: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  Nothing under
``tiny_audio_amd/`` imports it, and the product path raises when the HIP
library is missing instead of falling back to this code.

It is a numpy restatement (float32 storage, float64 where the reference's own
library does so internally) of what the reference computes in PyTorch /
``transformers``; every function cites the reference ``file:line`` it follows
(``TF:`` = the ``transformers`` package the reference pins, 5.x).

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the reference
(``/root/reference/tiny_audio`` + ``transformers``) in the build container,
runs it on seeded inputs/weights produced by ``oracle.weights`` and stores the
outputs in ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks this
restatement against those vectors (and against the known-answer tests the
reference's own test-suite holds for length formulas and embed gathering).
"""
