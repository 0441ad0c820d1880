"""The C ABI driven from plain C++ (examples/abi_smoke.cpp: hipMalloc'd buffers, no torch, no Python): the GEMM is bit-exact
on integer data, mask selection equals the stable-rank definition, bad arguments are rejected with a message."""
import subprocess
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def test_cpp_caller_of_the_abi():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from maestro_amd.csrc.build import EXAMPLE_BIN, build, build_example
    if not EXAMPLE_BIN.exists():      # normally prebuilt by __graft_entry__.build(); hipcc is on the GPU box too
        build()
        build_example()
    r = subprocess.run([str(EXAMPLE_BIN)], capture_output=True, text=True, timeout=120, cwd=ROOT)   # a child process, not an exec
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "mh_gemm_bf16 300x192x96: 0 mismatches" in r.stdout and "abi_smoke ok" in r.stdout, r.stdout
