"""ctypes binding of libmaestro_hip.so (the C ABI in include/maestro_hip.h) for torch device tensors.

There is NO fallback: if the library is missing or a tensor is not on the GPU the call raises.  PyTorch is used
only for device memory and the current HIP stream.
"""

from __future__ import annotations

import ctypes
from pathlib import Path

import torch

_LIB_PATH = Path(__file__).resolve().parent / "lib" / "libmaestro_hip.so"
_lib = None

GEMM_NT, GEMM_NN, GEMM_TN = 0, 1, 2
OUT_F32, BIAS, GELU, RESIDUAL, DGELU, ATOMIC = 1, 2, 4, 8, 16, 32


class HipExtensionError(RuntimeError):
    pass


def lib() -> ctypes.CDLL:
    """Load the HIP extension; fail loudly when it is absent (no CPU/PyTorch fallback exists)."""
    global _lib
    if _lib is None:
        if not _LIB_PATH.exists():
            raise HipExtensionError(
                f"{_LIB_PATH} not found: build it with `python -m maestro_amd.csrc.build` "
                "(or __graft_entry__.build()).  The MAE hot path has no fallback implementation.")
        _lib = ctypes.CDLL(str(_LIB_PATH))
        _lib.mh_last_error.restype = ctypes.c_char_p
    return _lib


def _check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().mh_last_error().decode(errors="replace")
        raise HipExtensionError(f"{what} failed (rc={rc}): {msg}")


def ptr(t: torch.Tensor | None):
    if t is None:
        return ctypes.c_void_p(0)
    if not t.is_cuda:
        raise HipExtensionError("maestro_amd kernels need GPU tensors (no CPU fallback)")
    return ctypes.c_void_p(t.data_ptr())


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


_I, _F, _L = ctypes.c_int, ctypes.c_float, ctypes.c_long


def gemm(layout: int, M: int, N: int, K: int, A, lda: int, B, ldb: int, C, ldc: int, flags: int = 0, bias=None,
         res=None, ldr: int = 0, aux_in=None, aux_out=None, ldaux: int = 0) -> None:
    _check(lib().mh_gemm_bf16(_I(layout), _I(M), _I(N), _I(K), ptr(A), _I(lda), ptr(B), _I(ldb), ptr(C), _I(ldc),
                              _I(flags), ptr(bias), ptr(res), _I(ldr), ptr(aux_in), ptr(aux_out), _I(ldaux),
                              stream()), "mh_gemm_bf16")


def call(name: str, *args) -> None:
    """Generic checked call: tensors -> device pointers, ints/floats passed through as given ctypes."""
    conv = []
    for a in args:
        if a is None or isinstance(a, torch.Tensor):
            conv.append(ptr(a))
        else:
            conv.append(a)
    _check(getattr(lib(), name)(*conv, stream()), name)
