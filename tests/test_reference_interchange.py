"""Checkpoint interchange WITH the reference (build container only: needs /root/reference; skipped elsewhere).

Runs ``oracle/gen_golden.py ckpt`` in a child process (it installs import stubs) into a scratch directory: the reference
module writes a Lightning-layout checkpoint, and a checkpoint written by ``maestro_amd`` is loaded into the REFERENCE
``SSLModule`` with ``strict=True`` (``maestro/run_experiment.py:66-74``)."""

import gzip
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.skipif(not Path("/root/reference/maestro").is_dir(), reason="the reference tree is not on this box")
def test_checkpoints_go_both_ways(tmp_path, golden_dir):
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", MAESTRO_GOLDEN_DIR=str(tmp_path))
    r = subprocess.run([sys.executable, "-m", "oracle.gen_golden", "ckpt"], cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "loaded into the reference SSLModule with strict=True" in r.stdout
    assert "built from the reference's config objects" in r.stdout       # the one-line import swap of run_experiment.py
    fresh = gzip.decompress((tmp_path / "ref_written.ckpt.gz").read_bytes())
    stored = gzip.decompress((golden_dir / "ref_written.ckpt.gz").read_bytes())
    assert len(fresh) == len(stored)          # same tensors, same pickled hyper-parameters as the committed fixture
    assert not list(Path("/root/reference").rglob("__pycache__"))
