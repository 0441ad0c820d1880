"""ctypes binding of libmaestro_hip.so (the C ABI in include/maestro_hip.h) for torch device tensors.

There is NO fallback: if the library is missing or a tensor is not on the GPU the call raises.  PyTorch is used
only for device memory and the current HIP stream.  ``KernelTimer`` (bottom of the file) brackets the MFMA kernels
with HIP events on their launch stream for bench.py's roofline leg.
"""

from __future__ import annotations

import ctypes
import os
from pathlib import Path

import torch

_LIB_PATH = Path(__file__).resolve().parent / "lib" / "libmaestro_hip.so"
_lib = None

GEMM_NT, GEMM_NN, GEMM_TN = 0, 1, 2
OUT_F32, BIAS, GELU, RESIDUAL, DGELU, ATOMIC, COLSUM, AUX_DGELU, MULAUX, C8_E5M2, AUX_U8 = 1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024


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
        retired = [v for v in ("MAESTRO_GROUPED", "MH_GEMM_SPLITK", "MH_ATTN_BWD", "MAESTRO_CU_MASK") if os.environ.get(v)]
        if retired:     # their kernels / code paths were removed (rounds 5-6): refuse instead of silently running the default path
            raise HipExtensionError(f"{', '.join(retired)}: retired experiment switch(es) -- the variant no longer exists, this run "
                                    "would measure the default path (INTEGRATION.md, profiles/r05_experiments.md, r06_experiments.md)")
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


# ---- GEMM: kernel / tile choice (include/maestro_hip.h MH_TILE_*)
TILE_AUTO, TILE_REG_128, TILE_DMA_256, TILE_DMA_256x128, TILE_DMA_128x256, TILE_DMA_128, TILE_DMA_128x4 = -1, 0, 1, 2, 3, 4, 5
TILE_DMA_256_LOCKSTEP = 6   # MH_TILE_DMA_256 without the wave-group stagger (A/B experiments)
TILE_REG_64, TILE_REG_192 = 13, 14   # the register-staged kernel with 64 x 128 / 192 x 128 tiles
TILE_SK_192, TILE_SK_256, TILE_SK_DMA_256 = 15, 16, 17   # stream-K tiles (gemm_sk.hip, gemm_sk_dma.hip): routed to mh_gemm_bf16_sk with a per-stream workspace
SK_TILES = (TILE_SK_192, TILE_SK_256, TILE_SK_DMA_256)
TILE_PP_128 = 7             # persistent 128x128 tile, epilogue of tile t inside the main loop of tile t + 1 (gemm_pp.hip)
TILES = (TILE_REG_128, TILE_DMA_256, TILE_DMA_256x128, TILE_DMA_128x256, TILE_DMA_128, TILE_DMA_128x4)
_LAYOUT_NAME = {0: "NT", 1: "NN", 2: "TN"}
_TILE_NAME = {TILE_REG_128: "gemm_kernel<{}>", TILE_DMA_256: "gemm_dma_kernel<256x256,{}>",
              TILE_DMA_256x128: "gemm_dma_kernel<256x128,{}>", TILE_DMA_128x256: "gemm_dma_kernel<128x256,{}>",
              TILE_DMA_128: "gemm_dma_kernel<128x128,{}>", TILE_DMA_128x4: "gemm_dma_kernel<128x128q,{}>",
              TILE_DMA_256_LOCKSTEP: "gemm_dma_kernel<256x256,{}>", TILE_PP_128: "gemm_pp_kernel<{}>",
              TILE_REG_64: "gemm_kernel<{},64x128>", TILE_REG_192: "gemm_kernel<{},192x128>",
              TILE_SK_192: "gemm_sk_kernel<{},192x128>", TILE_SK_256: "gemm_sk_kernel<{},256x128>",
              TILE_SK_DMA_256: "gemm_sk_dma_kernel<256x256,{}>"}
_tile_choice: dict = {}     # (layout, M, N, K, flags) -> fastest tile, filled while tuning is on
_tuning = False


def set_gemm_tuning(on: bool) -> None:
    """While on, the first call of every distinct (layout, M, N, K, flags) times each eligible tile on the call's own
    operands (device-synchronising: eager launches only, never inside a hipGraph capture) and keeps the fastest."""
    global _tuning
    _tuning = bool(on)


def gemm_tile_choices() -> dict:
    return dict(_tile_choice)


class InStepTuner:
    """Picks, per GEMM signature of a step, the tile that is fastest INSIDE the step.

    Round 3 finding (profiles/README.md): the fastest tile in an isolated loop (hot operands, nothing else on the chip) is often
    not the fastest between the step's other kernels -- the isolated ranking cost the probe step 2.5 % -- while HIP-event times
    of the step's own eager launches predicted the whole-step A/B.  So the engine runs its first step several times with the
    same inputs and draws, one candidate tile per pass for every signature at once (``begin``), each launch bracketed by events on
    its stream; ``finish`` keeps a candidate only where it beats the library's rule (MH_TILE_AUTO) by ``margin`` over all
    launches of that signature, else the rule stays.  Choices land in the process-wide table ``_pick_tile`` consults."""

    CANDIDATES = (TILE_AUTO, TILE_REG_128, TILE_PP_128, TILE_REG_64, TILE_REG_192, TILE_DMA_256) + (
        (TILE_SK_DMA_256, TILE_SK_192, TILE_SK_256) if os.environ.get("MAESTRO_INSTEP_SK") == "1" else ())   # round 6: the stream-K tiles

    def __init__(self) -> None:
        self.cand, self.ev, self.bad = None, {}, set()

    def begin(self, cand) -> None:
        """``None``: launches run under the rule and are not timed (the warm-up pass)."""
        self.cand = cand

    def open(self, key):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.ev.setdefault(key, {}).setdefault(self.cand, []).append((e0, e1))
        return e0, e1

    def reject(self, key) -> None:
        self.bad.add((key, self.cand))

    def finish(self, margin: float = 0.03) -> dict:
        torch.cuda.synchronize()
        report = {}
        for key, per in self.ev.items():
            base_n = len(per.get(TILE_AUTO, ()))
            ms = {c: sum(a.elapsed_time(b) for a, b in evs) for c, evs in per.items()
                  if (key, c) not in self.bad and len(evs) == base_n and base_n}
            if TILE_AUTO not in ms:
                continue
            best = min(ms, key=ms.get)
            pick = best if ms[best] < (1.0 - margin) * ms[TILE_AUTO] else TILE_AUTO
            _tile_choice[key] = pick
            report[key] = (pick, ms)
        return report


_instep: InStepTuner | None = None


def set_instep_tuner(t: InStepTuner | None) -> None:
    global _instep
    _instep = t


# ---- stream-K GEMM (mh_gemm_bf16_sk): one workspace per (stream, tile, grid) -- launches on different streams may overlap and must not
# share flag words or partial slots.  Zeroed once at allocation; the kernel leaves its flag words zero.  Allocated on first use, so the
# first call of a new (stream, tile, grid) must be an eager one (the engine's warm-up passes run before any hipGraph capture).
_sk_ws: dict = {}
_sk_grid = None


def sk_grid() -> int:
    """Persistent workgroups of a stream-K launch: one per CU (MH_SK_GRID overrides, e.g. the CU count of a masked stream)."""
    global _sk_grid
    if _sk_grid is None:
        env = os.environ.get("MH_SK_GRID")
        _sk_grid = int(env) if env else torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count
    return _sk_grid


def sk_workspace(tile: int, grid: int) -> torch.Tensor:
    key = (torch.cuda.current_stream().cuda_stream, tile, grid)
    ws = _sk_ws.get(key)
    if ws is None:
        lib().mh_gemm_sk_workspace.restype = ctypes.c_long
        nbytes = lib().mh_gemm_sk_workspace(_I(tile), _I(grid))
        if nbytes <= 0:
            raise HipExtensionError(f"mh_gemm_sk_workspace({tile}, {grid}) = {nbytes}")
        if torch.cuda.is_current_stream_capturing():
            raise HipExtensionError("stream-K workspace requested for the first time inside a hipGraph capture: run the step eagerly once first")
        ws = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
        _sk_ws[key] = ws
    return ws


def sk_error_flag(ws: torch.Tensor) -> int:
    """1 if a finisher's wait for a partial ran into its bound in any launch that used this workspace (results are then wrong)."""
    return int(ws[4092:4096].view(torch.int32).item())


def gemm_sk(tile: int, layout: int, M: int, N: int, K: int, A, lda: int, B, ldb: int, C, ldc: int, flags: int = 0, bias=None,  # noqa: N803
            res=None, ldr: int = 0, grid: int | None = None) -> int:
    """Returns the library's code: 0 launched, -2 the problem is not served by the stream-K kernel (nothing launched)."""
    grid = grid or sk_grid()
    ws = sk_workspace(tile, grid)
    return lib().mh_gemm_bf16_sk(_I(tile), _I(layout), _I(M), _I(N), _I(K), ptr(A), _I(lda), ptr(B), _I(ldb), ptr(C), _I(ldc),
                                 _I(flags), ptr(bias), ptr(res), _I(ldr), ptr(ws), _L(ws.numel()), _I(grid), stream())


def _gemm_tile(tile, layout, M, N, K, A, lda, B, ldb, C, ldc, flags, bias, res, ldr, aux_in, aux_out, ldaux, colsum) -> int:  # noqa: N803
    if tile in SK_TILES:
        return gemm_sk(tile, layout, M, N, K, A, lda, B, ldb, C, ldc, flags, bias, res, ldr)
    return lib().mh_gemm_bf16_tile(_I(tile), _I(layout), _I(M), _I(N), _I(K), ptr(A), _I(lda), ptr(B), _I(ldb), ptr(C),
                                   _I(ldc), _I(flags), ptr(bias), ptr(res), _I(ldr), ptr(aux_in), ptr(aux_out), _I(ldaux),
                                   ptr(colsum), stream())


def _tune_gemm(key, args) -> int:
    best, best_ms = TILE_REG_128, float("inf")
    for tile in TILES:
        rc = _gemm_tile(tile, *args)
        if rc == -2:
            continue
        _check(rc, "mh_gemm_bf16_tile")
        _gemm_tile(tile, *args)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            _gemm_tile(tile, *args)
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1)
        if ms < best_ms:
            best, best_ms = tile, ms
    _tile_choice[key] = best
    return best


def _pick_tile(layout, M, N, K, flags, args) -> int:  # noqa: N803
    tflags = flags & ~COLSUM                 # the column-sum side output does not change the kernel's cost profile
    key = (layout, M, N, K, tflags)
    tile = _tile_choice.get(key)
    if tile is None:
        # Experiment switches are resolved HERE, on the host side: the C library reads no environment (its header promises no
        # process-global state).  MH_GEMM_TILE=<id> forces one tile; MH_GEMM_DMA=0 keeps everything on the register-staged
        # kernel, =1 sends every eligible problem to the 256x256 LDS-DMA tile; MH_DMA_STAGGER=0 picks its lockstep form.
        forced = os.environ.get("MH_GEMM_TILE")
        if forced is not None:
            t = int(forced)
            # the ablation builds of the ping-pong tile (MH_TILE_PP_128_DIAG1..5) leave out parts of the kernel: WRONG outputs.  The
            # shipped library does not contain them (the C ABI returns -2); a library built with MH_BUILD_FLAGS=-DMH_DIAG_TILES does
            if TILE_PP_128 < t <= TILE_PP_128 + 5 and os.environ.get("MH_ALLOW_DIAG_TILES") != "1":
                raise HipExtensionError(f"MH_GEMM_TILE={t} selects a diagnostic build of the ping-pong tile that writes wrong "
                                        "outputs (present only in a -DMH_DIAG_TILES library); set MH_ALLOW_DIAG_TILES=1 to run it on purpose")
            return t
        dma = os.environ.get("MH_GEMM_DMA", "")[:1]
        if os.environ.get("MH_GEMM_PP", "")[:1] == "0" and dma == "":   # A/B: the round-2 rule (no persistent ping-pong tile)
            return TILE_DMA_256 if _uses_dma(layout, M, N, K, flags) else TILE_REG_128
        if dma == "0":
            return TILE_REG_128
        lockstep = os.environ.get("MH_DMA_STAGGER", "")[:1] == "0"
        if (dma == "1" and layout != GEMM_TN and not (flags & ATOMIC)) or (lockstep and _uses_dma(layout, M, N, K, flags)):
            return TILE_DMA_256_LOCKSTEP if lockstep else TILE_DMA_256
        if _tuning and not (flags & ATOMIC):   # accumulating outputs cannot be re-run for timing: the library's rule decides
            targs = list(args)
            targs[10], targs[17] = tflags, None
            tile = _tune_gemm(key, tuple(targs))
        else:
            tile = TILE_AUTO
    return tile


def _uses_192(layout, M, N, K, flags) -> bool:  # noqa: N803
    """Mirror of the 192 x 128 rule in csrc/gemm.hip (labels kernel timings only)."""
    if layout == GEMM_TN or (flags & (COLSUM | ATOMIC)) or K < 1536 or K % 64 or _uses_dma(layout, M, N, K, flags):
        return False
    t128 = -(-M // 128) * -(-N // 128)
    return 512 < t128 <= 576 and -(-M // 192) * -(-N // 128) <= 512


def _uses_pp(layout, M, N, K, flags) -> bool:  # noqa: N803
    """Mirror of the MH_TILE_AUTO rule in csrc/gemm.hip + gemm_pp_dispatch's eligibility (labels kernel timings only)."""
    if os.environ.get("MH_GEMM_PP", "")[:1] == "0":
        return False
    served = flags in (0, BIAS | GELU | AUX_DGELU | AUX_U8, OUT_F32 | BIAS | RESIDUAL)
    if not served or layout == GEMM_TN or K % 64 or K < 512 or N % 128 or _uses_192(layout, M, N, K, flags):
        return False
    t128 = -(-M // 128) * -(-N // 128)
    if layout != GEMM_NT or t128 < 256:
        return False
    if flags & GELU:
        return t128 <= 2304
    return not _uses_dma(layout, M, N, K, flags) or (K < 1024 and t128 <= 8192)


def _auto_tile_name(layout, M, N, K, flags) -> int:  # noqa: N803
    if _uses_192(layout, M, N, K, flags):
        return TILE_REG_192
    if _uses_pp(layout, M, N, K, flags):
        return TILE_PP_128
    return TILE_DMA_256 if _uses_dma(layout, M, N, K, flags) else TILE_REG_128


def gemm(layout: int, M: int, N: int, K: int, A, lda: int, B, ldb: int, C, ldc: int, flags: int = 0, bias=None,  # noqa: N803
         res=None, ldr: int = 0, aux_in=None, aux_out=None, ldaux: int = 0, colsum=None, tile: int | None = None) -> None:
    """``tile``: one of TILE_* to force a kernel (tests / micro-benchmarks); default = the tuned choice for this problem
    signature if there is one, else the library's own rule (MH_TILE_AUTO)."""
    args = (layout, M, N, K, A, lda, B, ldb, C, ldc, flags, bias, res, ldr, aux_in, aux_out, ldaux, colsum)
    explicit = tile is not None
    if _instep is not None and _instep.cand is not None and not explicit and not (flags & ATOMIC):
        key = (layout, M, N, K, flags & ~COLSUM)
        e0, e1 = _instep.open(key)
        e0.record()
        rc = _gemm_tile(_instep.cand, *args)
        if rc == -2:                   # the candidate does not serve this problem: not a contender for this signature
            _instep.reject(key)
            rc = _gemm_tile(TILE_AUTO, *args)
        _check(rc, "mh_gemm_bf16_tile")
        e1.record()
        return
    if not explicit:
        tile = _pick_tile(layout, M, N, K, flags, args)
    ev = None
    if _timer is not None:
        named = tile if tile != TILE_AUTO else _auto_tile_name(layout, M, N, K, flags)
        if named in (TILE_REG_64, TILE_REG_192) and (layout == GEMM_TN or (flags & COLSUM)):
            named = TILE_REG_128       # (the tile height applies to K-minor A without column sums; the library falls back silently)
        ev = _timer.record(_TILE_NAME[named].format(_LAYOUT_NAME[layout]), 2.0 * M * N * K, (M, N, K))
        ev[0].record()
    rc = _gemm_tile(tile, *args)
    if rc == -2 and not explicit:      # e.g. MH_GEMM_TILE / MH_GEMM_DMA=1 forced a DMA tile onto a problem with a K tail
        # the A/B switches that exist to keep a kernel family OUT of the run (MH_GEMM_DMA=1: everything on the DMA tile or the
        # register tile; MH_GEMM_PP=0: the round-2 dispatch) must not fall back to MH_TILE_AUTO, whose rule may pick that family
        pinned = os.environ.get("MH_GEMM_DMA", "")[:1] == "1" or os.environ.get("MH_GEMM_PP", "")[:1] == "0"
        rc = _gemm_tile(TILE_REG_128 if pinned else TILE_AUTO, *args)
    if rc == -2:
        raise HipExtensionError(f"mh_gemm_bf16_tile: problem ({M}, {N}, {K}) does not qualify for tile {tile}")
    _check(rc, "mh_gemm_bf16_tile")
    if ev is not None:
        ev[1].record()


def call(name: str, *args) -> None:
    """Generic checked call: tensors -> device pointers, ints/floats passed through as given ctypes."""
    conv = []
    for a in args:
        if a is None or isinstance(a, torch.Tensor):
            conv.append(ptr(a))
        else:
            conv.append(a)
    _check(getattr(lib(), name)(*conv, stream()), name)


def _hbm_call(kind: str, nbytes: float, name: str, *args) -> None:
    """``call`` bracketed by HIP events when a ``KernelTimer`` is active: the HBM-bound sub-stages report achieved GB/s on
    their ALGORITHMIC bytes (bench.py's ``hbm_substages``)."""
    if _timer is None:
        return call(name, *args)
    e0, e1 = _timer.record_bytes(kind, nbytes)
    e0.record()
    call(name, *args)
    e1.record()


def _esz(t) -> int:
    return 0 if t is None else t.element_size()


# ------------------------------------------------------------------------------------------------ typed wrappers
def layernorm_fwd(x, x_L, x_off, gamma, beta, y, y_L, y_off, mean, rstd, B, n, dim, eps=1e-5):
    _hbm_call(f"ln_fwd<{dim}>", float(B) * n * dim * (4 + _esz(y)), "mh_layernorm_fwd", x, _I(x_L), _I(x_off), gamma, beta, y, _I(y_L),
              _I(y_off), _I(1 if y.dtype == torch.float32 else 0), mean, rstd, _I(B), _I(n), _I(dim), _F(eps))


def layernorm_fwd_fp8(x, x_L, x_off, gamma, beta, y, y_L, y_off, mean, rstd, B, n, dim, y8, y8_scale, y8_amax, eps=1e-5):
    """LayerNorm forward writing the bf16 output AND its e4m3 copy (times ``y8_scale``), absmax folded into ``y8_amax``."""
    call("mh_layernorm_fwd_fp8", x, _I(x_L), _I(x_off), gamma, beta, y, _I(y_L), _I(y_off), mean, rstd, _I(B), _I(n), _I(dim),
         _F(eps), y8, y8_scale, y8_amax)


def layernorm_bwd_workspace(rows, dim) -> int:
    f = lib().mh_layernorm_bwd_workspace
    f.restype = ctypes.c_long
    return int(f(_I(rows), _I(dim)))


def layernorm_bwd(dy, dy_L, dy_off, x, x_L, x_off, gamma, mean, rstd, dres, dx, dx_bf16, dgamma, dbeta, dcol, workspace,
                  B, n, dim):
    _hbm_call(f"ln_bwd<{dim}>", float(B) * n * dim * (_esz(dy) + 4 + _esz(dres) + 4 + _esz(dx_bf16)), "mh_layernorm_bwd", dy, _I(dy_L),
              _I(dy_off), _I(1 if dy.dtype == torch.float32 else 0), x, _I(x_L), _I(x_off), gamma, mean, rstd, dres, dx, dx_bf16,
              dgamma, dbeta, dcol, workspace, _I(B), _I(n), _I(dim))


def layernorm_bwd_partial(dy, dy_L, dy_off, x, x_L, x_off, gamma, mean, rstd, dres, dx, dx_bf16, workspace, B, n, dim):
    """LayerNorm backward that leaves (dgamma | dbeta | colsum(dx)) as per-block partial rows in ``workspace`` (see ColsumBatch)."""
    _hbm_call(f"ln_bwd<{dim}>", float(B) * n * dim * (_esz(dy) + 4 + _esz(dres) + 4 + _esz(dx_bf16)), "mh_layernorm_bwd_partial", dy,
              _I(dy_L), _I(dy_off), _I(1 if dy.dtype == torch.float32 else 0), x, _I(x_L), _I(x_off), gamma, mean, rstd, dres, dx,
              dx_bf16, workspace, _I(B), _I(n), _I(dim))


COLSUM_ROWS = 16   # MH_COLSUM_ROWS in include/maestro_hip.h


class _MhColsumJob(ctypes.Structure):
    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("rows", ctypes.c_int), ("cols", ctypes.c_int),
                ("ld", ctypes.c_int), ("reserved", ctypes.c_int)]


class ColsumBatch:
    """Descriptor table (built once: all buffers are static) for ``mh_colsum_batched``: jobs ``(src f32 [rows, ld], dst f32
    [cols], rows, cols, ld)`` -> ``dst += column sums of src``, all in one launch."""

    def __init__(self, jobs, device) -> None:
        if not jobs:
            raise HipExtensionError("ColsumBatch: no jobs")
        arr = (_MhColsumJob * len(jobs))()
        for i, (src, dst, rows, cols, ld) in enumerate(jobs):
            if src.dtype != torch.float32 or dst.dtype != torch.float32 or not src.is_cuda or not dst.is_cuda:
                raise HipExtensionError("ColsumBatch: f32 device tensors expected")
            if rows <= 0 or cols <= 0 or ld < cols or src.numel() < (rows - 1) * ld + cols or dst.numel() < cols:
                raise HipExtensionError(f"ColsumBatch job {i}: shape ({rows}, {cols}, ld {ld}) does not fit its buffers")
            arr[i] = _MhColsumJob(src.data_ptr(), dst.data_ptr(), rows, cols, ld, 0)
        self.keep = [t for job in jobs for t in job[:2]]     # the descriptors hold raw pointers
        self.table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)
        blocks = []                                          # one work item per workgroup: job << 48 | column block << 32 | row chunk
        for i, (_, _, rows, cols, _) in enumerate(jobs):
            if cols > 65535 * 256 or len(jobs) > 32767:
                raise HipExtensionError("ColsumBatch: too many jobs / columns for the work-item encoding")
            for cb in range(-(-cols // 256)):
                blocks += [(i << 48) | (cb << 32) | rc for rc in range(-(-rows // COLSUM_ROWS))]
        self.blocks = torch.tensor(blocks, dtype=torch.int64).to(device)
        self.n, self.n_blocks = len(jobs), len(blocks)

    def launch(self) -> None:
        call("mh_colsum_batched", self.table, _I(self.n), self.blocks, _I(self.n_blocks))


def attn_fwd(qkv, out, lse, B, N, H, D, scale):
    call("mh_attn_fwd", qkv, out, lse, _I(B), _I(N), _I(H), _I(D), _F(scale))


def attn_bwd(qkv, out, dout, lse, delta, dqkv, B, N, H, D, scale):
    call("mh_attn_bwd", qkv, out, dout, lse, delta, dqkv, _I(B), _I(N), _I(H), _I(D), _F(scale))


def patchify(img, cols, target, BD, Ctot, S, P, Kpad, norm_bands, n_groups, normalise, rescale_elev):
    # 4 B read per pixel + the bf16 GEMM columns (K padded to Kpad) + the fp32 loss target
    nb = float(BD) * S * S * Ctot * 4 + float(BD) * (S // P) ** 2 * (Kpad * 2 * (cols is not None) + Ctot * P * P * 4 * (target is not None))
    _hbm_call("patchify", nb, "mh_patchify", img, cols, target, _I(BD), _I(Ctot), _I(S), _I(P), _I(Kpad), norm_bands, _I(n_groups),
              _I(int(normalise)), _I(int(rescale_elev)))


def patchify_bands(img, cols, target, BD, Csrc, c0, Ctot, S, P, Kpad, norm_bands, n_groups, normalise, rescale_elev):
    """``patchify`` of channels ``c0 .. c0 + Ctot - 1`` of an image with ``Csrc`` channels (one band-group); ``cols`` or
    ``target`` may be None."""
    call("mh_patchify_bands", img, cols, target, _I(BD), _I(Csrc), _I(c0), _I(Ctot), _I(S), _I(P), _I(Kpad), norm_bands,
         _I(n_groups), _I(int(normalise)), _I(int(rescale_elev)))


def groupnorm_partial_size(BD, L, E) -> int:
    return int(lib().mh_groupnorm_partial_size(_I(BD), _I(L), _I(E)))


def groupnorm_stats(y, partial, stats, BD, L, E, eps=1e-5):
    call("mh_groupnorm_stats", y, partial, stats, _I(BD), _I(L), _I(E), _F(eps))


def embed_finish(y, stats, gamma, beta, pos, date, date_rows, date_off, xg, B, D, L, E, tok_off, Lgroup):
    call("mh_embed_finish", y, stats, gamma, beta, pos, date, _I(date_rows), _I(date_off), xg, _I(B), _I(D), _I(L), _I(E),
         _I(tok_off), _I(Lgroup))


def date_features(dates, ref_date, out, B, D, rows, row_off, fac):
    call("mh_date_features", dates, ref_date, out, _I(B), _I(D), _I(rows), _I(row_off), _F(fac))


def resize(src, dst, planes, Hin, Win, Hout, Wout, mode):
    call("mh_resize", src, dst, _L(planes), _I(Hin), _I(Win), _I(Hout), _I(Wout), _I(mode))


def rescale_elev(img, out, BD, C, S):
    call("mh_rescale_elev", img, out, _I(BD), _I(C), _I(S))


def embed_finish_bwd(dxg, y, stats, gamma, dyc, dgamma, dbeta, sums, B, D, L, E, tok_off, Lgroup):
    call("mh_embed_finish_bwd", dxg, y, stats, gamma, dyc, dgamma, dbeta, sums, _I(B), _I(D), _I(L), _I(E), _I(tok_off),
         _I(Lgroup))


def depatchify(patches, img, BD, C, S, P):
    call("mh_depatchify", patches, img, _I(BD), _I(C), _I(S), _I(P))


def mask_select(noise, struct_mask, visible_idx, masked_idx, inv, mask, B, L, k):
    call("mh_mask_select", noise, struct_mask, visible_idx, masked_idx, inv, mask, _I(B), _I(L), _I(k))


def gather_rows(src, idx, dst, B, src_L, n_idx, dim, dst_L, dst_off):
    call("mh_gather_rows", src, idx, dst, _I(B), _I(src_L), _I(n_idx), _I(dim), _I(dst_L), _I(dst_off))


def scatter_rows(ddst, idx, dsrc, B, src_L, n_idx, dim, dst_L, dst_off):
    call("mh_scatter_rows", ddst, idx, dsrc, _I(B), _I(src_L), _I(n_idx), _I(dim), _I(dst_L), _I(dst_off))


def expand_rows(src, inv, dst, B, L, n, dim):
    call("mh_expand_rows", src, inv, dst, _I(B), _I(L), _I(n), _I(dim))


def unmask_assemble(y, inv, mask_token, tok_slot, pos, date, date_row, n_date_rows, xdec, B, L, n_vis, Dd):
    call("mh_unmask_assemble", y, inv, mask_token, tok_slot, pos, date, date_row, _I(n_date_rows), xdec, _I(B), _I(L),
         _I(n_vis), _I(Dd))


def unmask_token_grad(dxdec, mask, tok_slot, dmask_token, B, L, Dd, slot, t_lo, t_hi):
    call("mh_unmask_token_grad", dxdec, mask, tok_slot, dmask_token, _I(B), _I(L), _I(Dd), _I(slot), _I(t_lo), _I(t_hi))


def unmask_assemble_per_sample(y, inv, mask_token, tok_slot_bl, pos, date, date_row, n_date_rows, xdec, B, L, n_vis, Dd):
    call("mh_unmask_assemble_per_sample", y, inv, mask_token, tok_slot_bl, pos, date, date_row, _I(n_date_rows), xdec, _I(B), _I(L),
         _I(n_vis), _I(Dd))


def unmask_token_grad_per_sample(dxdec, mask, tok_slot_bl, dmask_token, B, L, Dd, slot):
    call("mh_unmask_token_grad_per_sample", dxdec, mask, tok_slot_bl, dmask_token, _I(B), _I(L), _I(Dd), _I(slot))


def count_masked(mask, B, L, t_lo, t_hi, out):
    call("mh_count_masked", mask, _I(B), _I(L), _I(t_lo), _I(t_hi), out)


def count_masked_elems(mask, B, L, t_lo, t_hi, out, mult, accumulate):
    call("mh_count_masked_elems", mask, _I(B), _I(L), _I(t_lo), _I(t_hi), out, _I(mult), _I(int(accumulate)))


def masked_loss_bands(rec, target, mask_group, n_elems, weight, acc, drec, B, Lm, Lgroup, tok_off, PPC, p, tgt_C, tgt_c0, n_g):  # noqa: N803
    call("mh_masked_loss_bands", rec, target, mask_group, n_elems, _F(weight), acc, drec, _I(B), _I(Lm), _I(Lgroup),
         _I(tok_off), _I(PPC), _I(p), _I(tgt_C), _I(tgt_c0), _I(n_g))


def masked_loss(rec, target, mask_group, n_masked, weight, acc, drec, B, Lm, Lgroup, tok_off, PPC, p):
    # per element: reconstruction (bf16) + target (f32) read, d loss / d rec (bf16) written; + one mask byte per token
    _hbm_call("masked_loss", float(B) * Lm * (PPC * (_esz(rec) + 4 + _esz(drec)) + 1), "mh_masked_loss", rec, target, mask_group,
              n_masked, _F(weight), acc, drec, _I(B), _I(Lm), _I(Lgroup), _I(tok_off), _I(PPC), _I(p))


def dihedral(x, out, flags):
    """Per-sample flips / transpose of square rasters ``x [B, ..., S, S]`` into ``out`` (same shape); ``flags`` uint8 [B]."""
    if x.shape != out.shape or x.dtype != out.dtype or x.shape[-1] != x.shape[-2] or not x.is_contiguous():
        raise HipExtensionError("dihedral: contiguous square rasters of equal shape and dtype expected")
    B, S = x.shape[0], x.shape[-1]  # noqa: N806
    planes = x.numel() // (B * S * S)
    call("mh_dihedral", x, out, flags, _I(B), _L(planes), _I(S), _I(x.element_size()))


# ---- probe / finetune heads (csrc/heads.hip)
def token_resize(x, in_rows, in_off, out, out_rows, out_off, B, D, h, H, E):  # noqa: N803
    call("mh_token_resize", x, _L(in_rows), _I(in_off), out, _L(out_rows), _I(out_off), _I(B), _I(D), _I(h), _I(H), _I(E))


def token_resize_bwd(dout, out_rows, out_off, din, in_rows, in_off, B, D, h, H, E, accumulate=False):  # noqa: N803
    call("mh_token_resize_bwd", dout, _L(out_rows), _I(out_off), din, _L(in_rows), _I(in_off), _I(B), _I(D), _I(h), _I(H),
         _I(E), _I(int(accumulate)))


def attn_reduce_partial_rows(n_seq: int) -> int:
    lib().mh_attn_reduce_partial_rows.restype = ctypes.c_long
    return int(lib().mh_attn_reduce_partial_rows(_I(n_seq)))


def attn_reduce_fwd(kv, query, out, lse, n_batch, T, Lr, dim, heads=8):  # noqa: N803
    call("mh_attn_reduce_fwd", kv, query, out, lse, _I(n_batch), _I(T), _I(Lr), _I(dim), _I(heads))


def attn_reduce_bwd(kv, query, out, lse, dout, dkv, dq_partial, n_batch, T, Lr, dim, heads=8):  # noqa: N803
    call("mh_attn_reduce_bwd", kv, query, out, lse, dout, dkv, dq_partial, _I(n_batch), _I(T), _I(Lr), _I(dim), _I(heads))


def mean_reduce_fwd(x, out, n_batch, T, Lr, dim):  # noqa: N803
    call("mh_mean_reduce_fwd", x, out, _I(n_batch), _I(T), _I(Lr), _I(dim))


def mean_reduce_bwd(dout, dx, n_batch, T, Lr, dim):  # noqa: N803
    call("mh_mean_reduce_bwd", dout, dx, _I(n_batch), _I(T), _I(Lr), _I(dim))


def head_linear_fwd(x, W, bias, out, B, C, E):  # noqa: N803
    call("mh_head_linear_fwd", x, W, bias, out, _I(B), _I(C), _I(E))


def head_linear_bwd(x, W, dout, dx, dW, db, B, C, E):  # noqa: N803
    call("mh_head_linear_bwd", x, W, dout, dx, dW, db, _I(B), _I(C), _I(E))


def count_valid(target, missing_val, count):
    call("mh_count_valid", target, _I(target.element_size()), _L(target.numel()), _L(int(missing_val)), count)


def ce_loss(logits, target, missing_val, n_valid, acc, dlogits, B, g, P, C, ld=None):  # noqa: N803
    call("mh_ce_loss", logits, target, _I(target.element_size()), _L(int(missing_val)), n_valid, acc, dlogits,
         _I(1 if dlogits.dtype == torch.float32 else 0), _I(B), _I(g), _I(P), _I(C), _I(P * P * C if ld is None else ld))


def bce_loss(logits, target, missing_val, acc, dlogits, B, C):  # noqa: N803
    call("mh_bce_loss", logits, target, _F(float(missing_val)), acc, dlogits, _I(B), _I(C))


def zero_spans(base, spans, n_spans, max_len):
    call("mh_zero_spans", base, spans, _I(n_spans), _L(max_len))


def colsum(x, out, M, N, ld):
    call("mh_colsum", x, _I(1 if x.dtype == torch.float32 else 0), out, _I(M), _I(N), _I(ld))


def scale_dev(x, n, scale):
    """``x[:n] *= scale`` with ``scale`` a 1-element f32 device tensor (read on the device; no-op when it is 1)."""
    call("mh_scale_dev", x, _L(n), scale)


def cast_bf16(src, dst, n):
    call("mh_cast_bf16", src, dst, _L(n))


def pack_rows_bf16(w, dst, E, K, Kpad):
    call("mh_pack_rows_bf16", w, dst, _I(E), _I(K), _I(Kpad))


def unpack_rows_add(src, dst, E, K, Kpad):
    call("mh_unpack_rows_add", src, dst, _I(E), _I(K), _I(Kpad))


def adamw(p, g, m, v, p_bf16, n, lr, b1, b2, eps, wd, step, grad_scale=1.0):
    _hbm_call("adamw", float(n) * (16 + 12 + _esz(p_bf16)), "mh_adamw", p, g, m, v, p_bf16, _L(n), _F(lr), _F(b1), _F(b2), _F(eps), _F(wd),
              _I(step), _F(grad_scale))


def adamw_fp8(p, g, m, v, p_bf16, p_fp8, slot_map, scale, amax, n, lr, b1, b2, eps, wd, step, grad_scale=1.0):
    """AdamW that also writes the e4m3 weight shadows (delayed scaling; see ``mh_adamw_fp8``)."""
    call("mh_adamw_fp8", p, g, m, v, p_bf16, p_fp8, slot_map, scale, amax, _L(n), _F(lr), _F(b1), _F(b2), _F(eps), _F(wd), _I(step),
         _F(grad_scale))


_libm = None


def adamw_bias_corrections(b1: float, b2: float, step: int) -> tuple[float, float]:
    """``(1 - b1^t, sqrt(1 - b2^t))`` in the float32 arithmetic ``mh_adamw`` uses on the host (libm ``powf``), so that the
    device-scalar variant applies bit-identical updates."""
    import numpy as np
    global _libm
    if _libm is None:
        _libm = ctypes.CDLL("libm.so.6")
        _libm.powf.restype = ctypes.c_float
        _libm.powf.argtypes = [ctypes.c_float, ctypes.c_float]
    libm = _libm
    f = np.float32
    bc1 = f(1.0) - f(libm.powf(float(f(b1)), float(step)))
    bc2 = np.sqrt(f(1.0) - f(libm.powf(float(f(b2)), float(step))), dtype=np.float32)
    return float(bc1), float(bc2)


def adamw_dev(p, g, m, v, p_bf16, n, b1, b2, eps, wd, hyper):
    """AdamW with {lr, 1 - b1^t, sqrt(1 - b2^t), grad_scale, active} read from the device tensor ``hyper`` (graph-capturable)."""
    _hbm_call("adamw", float(n) * (16 + 12 + _esz(p_bf16)), "mh_adamw_dev", p, g, m, v, p_bf16, _L(n), _F(b1), _F(b2), _F(eps), _F(wd), hyper)


# ------------------------------------------------------------------------------------------------ kernel timing
class KernelTimer:
    """HIP-event timing of the MFMA kernels on the stream they are launched on (bench.py's roofline leg)."""

    def __init__(self) -> None:
        self.events: dict[str, list] = {}
        self.flops: dict[str, float] = {}
        self.count: dict[str, int] = {}
        self.shapes: dict = {}
        self.hbm: dict[str, list] = {}      # HBM-bound sub-stages: kind -> [(e0, e1, algorithmic bytes)]

    def record(self, kind: str, flops: float, shape=None):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.events.setdefault(kind, []).append((e0, e1))
        if shape is not None:
            self.shapes.setdefault((kind, shape), []).append((e0, e1, flops))
        self.flops[kind] = self.flops.get(kind, 0.0) + flops
        self.count[kind] = self.count.get(kind, 0) + 1
        return e0, e1

    def record_bytes(self, kind: str, nbytes: float):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.hbm.setdefault(kind, []).append((e0, e1, nbytes))
        return e0, e1

    def hbm_substages(self, peak_gbs: float) -> dict:
        """Per HBM-bound kernel: achieved GB/s = algorithmic bytes / HIP-event duration, summed over its launches."""
        torch.cuda.synchronize()
        out = {}
        for kind, evs in sorted(self.hbm.items()):
            ms = sum(a.elapsed_time(b) for a, b, _ in evs)
            nb = sum(n for _, _, n in evs)
            if ms > 0:
                out[kind] = {"gb_s": round(nb / ms / 1e6, 1), "frac": round(nb / ms / 1e6 / peak_gbs, 3), "launches": len(evs),
                             "avg_us": round(1e3 * ms / len(evs), 2), "mb_per_launch": round(nb / len(evs) / 1e6, 2)}
        return out

    def totals(self) -> dict[str, float]:
        torch.cuda.synchronize()
        return {k: sum(a.elapsed_time(b) for a, b in v) for k, v in self.events.items()}  # ms

    def summary(self, steps: int) -> dict:
        return {k: round(v / steps, 3) for k, v in self.totals().items()}

    def by_shape(self, steps: int) -> list:
        torch.cuda.synchronize()
        rows = []
        for (kind, shape), evs in self.shapes.items():
            ms = sum(a.elapsed_time(b) for a, b, _ in evs)
            fl = sum(f for _, _, f in evs)
            rows.append((ms / steps, kind, shape, len(evs) // steps, fl / (ms * 1e-3) / 1e12))
        return sorted(rows, reverse=True)

    def roofline(self, peak_tflops: float) -> dict:
        tot = self.totals()
        kind = max(tot, key=tot.get)
        achieved = self.flops[kind] / (tot[kind] * 1e-3) / 1e12
        return {"bound": "mfma", "kernel": kind, "achieved": round(achieved, 1), "peak": peak_tflops, "unit": "TFLOP/s",
                "frac": round(achieved / peak_tflops, 4), "launches": self.count[kind],
                "avg_launch_us": round(1e3 * tot[kind] / self.count[kind], 2),
                "flops_per_launch": round(self.flops[kind] / self.count[kind]), "traffic": None}


_timer: KernelTimer | None = None


def _uses_dma(layout, M, N, K, flags) -> bool:
    """Mirror of prefer_dma() in csrc/gemm.hip (labels kernel timings; decides where MH_DMA_STAGGER=0 applies)."""
    if layout == GEMM_TN or (flags & ATOMIC) or K % 32 or K < 256:
        return False
    tiles = -(-M // 256) * -(-N // 256)
    waves = -(-tiles // 256)
    return tiles >= 256 and 10 * tiles >= 9 * 256 * waves


def set_kernel_timer(t: KernelTimer | None) -> None:
    global _timer
    _timer = t


def kernel_timer_active() -> bool:
    return _timer is not None


_attn_fwd_raw, _attn_bwd_raw = attn_fwd, attn_bwd


def attn_fwd(qkv, out, lse, B, N, H, D, scale):  # noqa: F811
    if _timer is None:
        return _attn_fwd_raw(qkv, out, lse, B, N, H, D, scale)
    e0, e1 = _timer.record("attn_fwd", 4.0 * B * H * N * N * D, (B, N, H, D))
    e0.record()
    _attn_fwd_raw(qkv, out, lse, B, N, H, D, scale)
    e1.record()


def attn_bwd(qkv, out, dout, lse, delta, dqkv, B, N, H, D, scale):  # noqa: F811
    if _timer is None:
        return _attn_bwd_raw(qkv, out, dout, lse, delta, dqkv, B, N, H, D, scale)
    e0, e1 = _timer.record("attn_bwd", 10.0 * B * H * N * N * D, (B, N, H, D))  # algorithmic 5 matmuls (the two-kernel form recomputes 2 more)
    e0.record()
    _attn_bwd_raw(qkv, out, dout, lse, delta, dqkv, B, N, H, D, scale)
    e1.record()


# ------------------------------------------------------------------------------------------------ grouped wgrad
class _MhGroupedGemm(ctypes.Structure):
    _fields_ = [("A", ctypes.c_void_p), ("B", ctypes.c_void_p), ("C", ctypes.c_void_p), ("M", ctypes.c_int),
                ("N", ctypes.c_int), ("K", ctypes.c_int), ("lda", ctypes.c_int), ("ldb", ctypes.c_int), ("ldc", ctypes.c_int),
                ("reserved", ctypes.c_int), ("accumulate", ctypes.c_int)]


class GroupedTN:
    """Descriptor table (built once: all buffers are static) for ``mh_gemm_grouped_tn``: every entry is one
    dW[M, N] (f32) = A[K, M]^T B[K, N] problem; the launch covers all their 256x256 tiles.  A dW written by a single
    problem is stored; one shared by several problems (same encoder on several groups) is accumulated atomically."""

    @staticmethod
    def check(i, prob) -> int:
        """Validates one problem against the kernel's addressing limits; returns its number of 256x256 tiles."""
        A, B, C, M, N, K, lda, ldb, ldc = prob  # noqa: N806
        if M % 8 or N % 8 or lda % 8 or ldb % 8 or ldc % 4 or min(M, N, K) <= 0:
            raise HipExtensionError(f"grouped wgrad problem {i}: M, N, lda, ldb must be multiples of 8 ({M}, {N}, {lda}, {ldb})")
        if C.dtype != torch.float32 or A.dtype != torch.bfloat16 or B.dtype != torch.bfloat16:
            raise HipExtensionError("grouped wgrad: A, B bf16 and C f32 expected")
        if (-(-K // 32) * 32) * max(lda, ldb) * 2 + 65536 >= 1 << 31:
            raise HipExtensionError(f"grouped wgrad problem {i}: operand beyond the 2 GiB buffer-descriptor range")
        return -(-M // 256) * -(-N // 256)

    @staticmethod
    def count_tiles(problems) -> int:
        return sum(GroupedTN.check(i, p) for i, p in enumerate(problems))

    N_XCD = 8

    def __init__(self, problems, device) -> None:
        arr = (_MhGroupedGemm * len(problems))()
        self.keep = []
        writers = {}
        for prob in problems:
            writers[prob[2].data_ptr()] = writers.get(prob[2].data_ptr(), 0) + 1
        units = []   # (work, problem index, tiles_m, tiles_n)
        for i, prob in enumerate(problems):
            A, B, C, M, N, K, lda, ldb, ldc = prob  # noqa: N806
            n = self.check(i, prob)
            shared = int(writers[C.data_ptr()] > 1)   # several problems add into one (zeroed) dW: atomic epilogue
            arr[i] = _MhGroupedGemm(A.data_ptr(), B.data_ptr(), C.data_ptr(), M, N, K, lda, ldb, ldc, 0, shared)
            units.append((n * K, i, -(-M // 256), -(-N // 256)))
            self.keep += [A, B, C]
        # One tile queue per XCD.  Whole problems go to the least-loaded queue, largest first (work = tiles x K), so that a
        # problem's tiles run under one L2 at the same time and share their dY / X panels; a problem bigger than half a
        # queue's fair share is dealt out by rows of tiles (one dY panel each) instead.
        fair = sum(u[0] for u in units) / self.N_XCD
        pieces = []  # (work, K, [tile ids])
        for work, i, tm, tn in units:
            K = problems[i][5]  # noqa: N806
            rows = [[(i << 16) | (r << 8) | c for c in range(tn)] for r in range(tm)]
            if work > 0.5 * fair and tm > 1:
                pieces += [(tn * K, K, row) for row in rows]
            else:
                pieces.append((work, K, [t for row in rows for t in row]))
        load = [0.0] * self.N_XCD
        queues = [[] for _ in range(self.N_XCD)]
        for work, K, ids in sorted(pieces, key=lambda p: -p[0]):  # noqa: N806
            x = min(range(self.N_XCD), key=load.__getitem__)
            load[x] += work
            queues[x].append((K, ids))
        qlen = max(sum(len(ids) for _, ids in q) for q in queues)
        flat = torch.full((self.N_XCD, qlen), 0xFFFFFFFF, dtype=torch.int64)
        for x, q in enumerate(queues):
            ids = [t for _, tiles in sorted(q, key=lambda e: -e[0]) for t in tiles]   # longest tiles first
            flat[x, : len(ids)] = torch.tensor(ids, dtype=torch.int64)
        self.queues = flat.to(torch.uint32).to(device)
        raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
        self.table = raw.to(device)
        self.n, self.queue_len = len(problems), qlen
        self.tiles = sum(u[2] * u[3] for u in units)
        self.balance = max(load) / max(fair, 1e-9)       # 1.0 = perfectly even queues
        self.flops = sum(2.0 * M * N * K for (_, _, _, M, N, K, _, _, _) in problems)

    def launch(self) -> None:
        if _timer is None:
            call("mh_gemm_grouped_tn", self.table, _I(self.n), self.queues, _I(self.queue_len))
            return
        e0, e1 = _timer.record("gemm_dma_grouped_tn_kernel", self.flops, ("grouped", self.n, self.tiles))
        e0.record()
        call("mh_gemm_grouped_tn", self.table, _I(self.n), self.queues, _I(self.queue_len))
        e1.record()


# ------------------------------------------------------------------------------------------------ fp8 path (C5)
FP8_E4M3, FP8_E5M2 = 0, 1
FP8_MAX = {FP8_E4M3: 448.0, FP8_E5M2: 57344.0}
QCHUNK = 4096   # elements per work item of mh_quant_batched (csrc/quant.hip)


_FP8_TILE_HINT = {"256": 2048, "128": 4096, "128d": 8192}   # MH_GEMM_FP8_TILE_* (include/maestro_hip.h)


def gemm_fp8(M, N, K, A8, lda, B8, ldb, C, ldc, descale_a, descale_b, flags=0, a_format=FP8_E4M3, bias=None, res=None, ldr=0,  # noqa: N803
             aux_in=None, aux_out=None, ldaux=0, colsum=None, c8=None, ldc8=0, c8_scale=None, c8_amax=None) -> None:
    """``C = descale_a * descale_b * A8 @ B8^T`` (+ epilogue): A8 ``[M, K]`` / B8 ``[N, K]`` uint8 tensors holding OCP fp8."""
    ev = None
    if _timer is not None:
        ev = _timer.record("gemm_fp8_kernel", 2.0 * M * N * K, (M, N, K))
        ev[0].record()
    flags |= _FP8_TILE_HINT.get(os.environ.get("MH_FP8_TILE", ""), 0)      # (experiments / tests; the library reads no environment)
    _check(lib().mh_gemm_fp8(_I(M), _I(N), _I(K), ptr(A8), _I(lda), _I(a_format), ptr(B8), _I(ldb), ptr(C), _I(ldc), _I(flags),
                             ptr(descale_a), ptr(descale_b), ptr(bias), ptr(res), _I(ldr), ptr(aux_in), ptr(aux_out), _I(ldaux),
                             ptr(colsum), ptr(c8), _I(ldc8), ptr(c8_scale), ptr(c8_amax), stream()), "mh_gemm_fp8")
    if ev is not None:
        ev[1].record()


class _MhQuantJob(ctypes.Structure):
    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("dst_t", ctypes.c_void_p), ("n", ctypes.c_long),
                ("rows", ctypes.c_int), ("cols", ctypes.c_int), ("slot", ctypes.c_int), ("is_f32", ctypes.c_int),
                ("format", ctypes.c_int), ("reserved", ctypes.c_int)]


AMAX_PITCH = 32 * 64   # MH_FP8_AMAX_SUBSLOTS * MH_FP8_AMAX_STRIDE


class Fp8Scales:
    """Scale / descale / amax tables (one slot per quantised tensor), all on the device.  ``amax`` is [n_slots, AMAX_PITCH]: the
    kernels fold into one of 32 sub-slots of a slot's row (``MH_FP8_AMAX_PITCH`` in the header: same-line atomics are slow);
    ``absmax(slot)`` is the maximum over the row."""

    def __init__(self, n_slots: int, device) -> None:
        self.n = n_slots
        self.scale = torch.ones(n_slots, dtype=torch.float32, device=device)
        self.descale = torch.ones(n_slots, dtype=torch.float32, device=device)
        self.amax = torch.zeros(n_slots, AMAX_PITCH, dtype=torch.float32, device=device)

    def absmax(self, slot: int) -> float:
        return float(self.amax[slot].max())

    def update(self, lo: int = 0, hi: int | None = None, fmt: int = FP8_E4M3, margin: int = 1) -> None:
        """``scale = 2^(floor(log2(max / amax)) - margin)`` for slots ``lo .. hi-1``; resets their amax."""
        hi = self.n if hi is None else hi
        call("mh_fp8_update_scales", self.amax[lo:hi], self.scale[lo:hi], self.descale[lo:hi], _I(hi - lo), _F(FP8_MAX[fmt]),
             _I(margin))


class _MhTransposeJob(ctypes.Structure):
    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("rows", ctypes.c_int), ("cols", ctypes.c_int)]


class TransposeBatch:
    """``mh_transpose_u8_batched``: ``pairs`` = (src uint8 [rows, cols], dst uint8 [cols, rows]), rows and cols multiples of 64."""

    def __init__(self, pairs, device) -> None:
        arr = (_MhTransposeJob * len(pairs))()
        items, self.keep = [], []
        for i, (src, dst) in enumerate(pairs):
            rows, cols = src.shape
            if rows % 64 or cols % 64 or tuple(dst.shape) != (cols, rows) or not (src.is_contiguous() and dst.is_contiguous()) \
                    or src.dtype != torch.uint8 or dst.dtype != torch.uint8 or (src.data_ptr() | dst.data_ptr()) % 16:
                raise HipExtensionError("TransposeBatch: contiguous 16-byte aligned uint8 [rows, cols] -> [cols, rows], multiples of 64")
            arr[i] = _MhTransposeJob(src.data_ptr(), dst.data_ptr(), rows, cols)
            items += [(i << 32) | t for t in range((rows // 64) * (cols // 64))]
            self.keep += [src, dst]
        self.table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)
        self.items = torch.tensor(items, dtype=torch.int64).to(device)
        self.n_items = len(items)

    def launch(self) -> None:
        call("mh_transpose_u8_batched", self.table, self.items, _I(self.n_items))


class QuantBatch:
    """Job table for ``mh_quant_batched``: ``jobs`` = dicts(src, dst=None, dst_t=None, slot, format=FP8_E4M3); a transposed copy
    needs a 2-D ``src``.  ``launch(mode)``: 0 absmax only, 1 cast, 2 cast + absmax (delayed scaling)."""

    def __init__(self, jobs, scales: Fp8Scales, device) -> None:
        arr = (_MhQuantJob * len(jobs))()
        items, self.keep = [], []
        for i, jb in enumerate(jobs):
            src, dst, dst_t = jb["src"], jb.get("dst"), jb.get("dst_t")
            n = src.numel()
            if n % 4 or not src.is_contiguous() or src.dtype not in (torch.float32, torch.bfloat16):
                raise HipExtensionError("QuantBatch: contiguous f32 / bf16 sources with numel % 4 == 0 expected")
            rows, cols = (src.shape[0], n // src.shape[0]) if src.dim() >= 2 else (1, n)
            if dst_t is not None and (cols % 4 or dst_t.numel() != n):
                raise HipExtensionError("QuantBatch: the transposed copy needs cols % 4 == 0 and as many bytes as elements")
            if dst is not None and dst.numel() != n:
                raise HipExtensionError("QuantBatch: dst must hold one byte per source element")
            arr[i] = _MhQuantJob(src.data_ptr(), dst.data_ptr() if dst is not None else None,
                                 dst_t.data_ptr() if dst_t is not None else None, n, rows, cols, jb["slot"],
                                 int(src.dtype == torch.float32), jb.get("format", FP8_E4M3), 0)
            items += [(i << 32) | c for c in range(-(-n // QCHUNK))]
            self.keep += [t for t in (src, dst, dst_t) if t is not None]
        self.table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)
        self.items = torch.tensor(items, dtype=torch.int64).to(device)
        self.scales, self.n_items = scales, len(items)

    def launch(self, mode: int) -> None:
        call("mh_quant_batched", self.table, self.items, _I(self.n_items), self.scales.scale, self.scales.amax, _I(mode))
