"""Test configuration: registers the ``gpu`` marker; CPU tests must pass without a GPU."""

import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def observed():
    """``observed(test, key, value)``: appends a measured error to ``gpurun_out/observed_errors.jsonl`` (scratch, merged back
    from the GPU box) -- the stated tolerances are set to <= 3x these values."""
    import json

    out = ROOT / "gpurun_out"

    def record(test: str, key: str, value: float) -> None:
        try:
            out.mkdir(exist_ok=True)
            with open(out / "observed_errors.jsonl", "a") as f:
                f.write(json.dumps({"test": test, "key": key, "value": float(value)}) + "\n")
        except OSError:
            pass

    return record
