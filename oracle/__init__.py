"""CPU oracle for the MAE pretraining hot path -- TEST INFRASTRUCTURE ONLY.

This package is a plain PyTorch-CPU (fp32) restatement of the reference algorithm
(IGNF/MAESTRO, ``maestro/ssl``, ``maestro/layers``, ``maestro/train/model.py``) plus a restatement of
the third-party ``vit_pytorch.vit.Transformer`` (vit-pytorch 1.10.1, absent from the reference tree).
Every function cites the reference ``file:line`` it follows.

Rules:
  * Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
  * ``maestro_amd`` (the product) never imports it and has no CPU fallback.
  * Parity pinning: ``oracle/gen_golden.py`` runs the *reference itself* (imported from
    ``/root/reference`` with non-arithmetic stubs) and the oracle on the same inputs/weights/RNG draws;
    the vectors it writes to ``tests/golden/`` pin this oracle to the reference.  The ViT block is
    third-party arithmetic that is not in the reference tree -> that part is "parity unpinned" by the
    reference and is pinned instead against an independent fp64 restatement (``oracle/vit.py``).
"""
