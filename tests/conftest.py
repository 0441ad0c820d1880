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


# Run order of the GPU suite (the driver runs ``pytest -m gpu -x``: a stop must not hide core parity behind an opt-in feature).
# Core parity first -- the bench configuration, the full-width BASELINE configurations, the reference-generated goldens --, then
# the kernel-level tests, then the wider rows (supervised branch, staging, data parallel), and the opt-in paths (fp8) last.  Files not listed keep their alphabetical place between the kernel tests and the opt-in group.
_GPU_ORDER = [
    "test_bench_config_parity_gpu", "test_fullwidth_parity_gpu", "test_mae_gpu",
    "test_kernels_gpu", "test_gemm_gpu", "test_gemm_dma_gpu", "test_gemm_grouped_gpu", "test_heads_gpu",
    "test_sup_gpu", "test_staging_gpu", "test_fullsize_gpu", "test_abi_cpp_gpu", "test_ddp_gpu", "test_ddp_nccl_gpu",
]
_GPU_LAST = ["test_fp8_gpu"]


def pytest_collection_modifyitems(session, config, items):
    def rank(item):
        stem = Path(str(item.fspath)).stem
        if stem in _GPU_ORDER:
            return _GPU_ORDER.index(stem)
        if stem in _GPU_LAST:
            return len(_GPU_ORDER) + 1 + _GPU_LAST.index(stem)
        return len(_GPU_ORDER)
    items.sort(key=rank)          # stable: the order inside a file is kept


def pytest_runtest_logreport(report):
    """One line per finished GPU test under ``gpurun_out/`` (scratch, merged back from the GPU box): the suite has stretches of
    several minutes (two-rank and child-process tests) that print nothing, and a run that writes nothing for seven minutes is
    taken to be hung by the GPU runner."""
    if report.when != "call" or "gpu" not in getattr(report, "keywords", {}):
        return
    try:
        out = ROOT / "gpurun_out"
        out.mkdir(exist_ok=True)
        with open(out / "gputest_progress.log", "a") as f:
            f.write(f"{report.outcome:7s} {report.duration:7.2f}s {report.nodeid}\n")
    except OSError:
        pass


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
