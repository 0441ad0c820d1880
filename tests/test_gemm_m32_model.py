"""The numpy restatement of the experimental 32x32x16 GEMM tile's index algebra (scripts/model_m32.py: LDS image, fragment reads,
the MFMA's documented lane maps, staged epilogue) reproduces A W^T and spreads every fragment read over 16 distinct 16-byte slots
per ds_read_b128 lane group.  CPU only; the kernel itself is gated (tests/test_gemm_m32_gpu.py)."""

import runpy
from pathlib import Path


def test_m32_index_algebra_reproduces_the_product(capsys):
    runpy.run_path(str(Path(__file__).resolve().parent.parent / "scripts" / "model_m32.py"), run_name="__main__")
    out = capsys.readouterr().out
    assert "tile result equals A W^T" in out and "conflict-free" in out
