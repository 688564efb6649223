"""CPU oracle for the AF3 / UALM audio-understanding forward pass.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: it may be
imported by ``tests/``, by ``__graft_entry__.smoke()`` and by the
``cpu_baseline`` leg of ``bench.py`` -- as the *checker* or the timed CPU
baseline -- and by nothing else.  The product path (``audio_intelligence_amd``)
never imports it and fails loudly when the HIP library is missing.

It is a build-owned restatement (PyTorch-CPU fp32 + numpy) of what the reference
computes on this path; every function cites the reference file:line it follows.
Parity pin: the restatement is checked against golden vectors captured by
importing the real reference in the build container
(``oracle/make_golden.py`` -> ``tests/golden/``; ``tests/test_oracle_golden.py``).
"""

from . import lengths, logmel, afwhisper, qwen2, ualm  # noqa: F401
