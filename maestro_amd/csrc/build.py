"""Build libmaestro_hip.so (gfx950) from the .hip sources in this directory with hipcc.

In-tree build: the shared object lands in ``maestro_amd/lib/`` so it travels with the repo snapshot to the GPU box
(it is git-ignored, not gpurun-ignored).  Objects are cached per source by content hash.
"""

from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

CSRC = Path(__file__).resolve().parent
ROOT = CSRC.parent.parent
LIB_DIR = CSRC.parent / "lib"
LIB = LIB_DIR / "libmaestro_hip.so"
OBJ_DIR = CSRC / "build"
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wno-unused-result"] + os.environ.get("MH_BUILD_FLAGS", "").split()
# Per-file code-generation flags.  attn.hip: keep the MFMA accumulators in VGPRs -- the softmax reads every score on the
# VALU, and with AGPR accumulators each tile paid 64-160 v_accvgpr_read/write moves (VALU-bound kernels: -25..-40 % cycles).
# -fno-slp-vectorize + scalar source (ATTN_SCALAR_VALU, default 1): packed v_pk_fma_f32 / v_pk_add_f32 beside MFMAs are no faster
# than two scalar instructions on gfx950 and measured 0-6 % slower here (scripts/bench_attn.py: backward N = 576, D = 32:
# 160 -> 150 us; N = 1024: 379 -> 370; forward +-1 %).  MH_ATTN_FLAGS="-DATTN_SCALAR_VALU=0" rebuilds the packed form.
FILE_FLAGS = {"attn.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-slp-vectorize"] + os.environ.get("MH_ATTN_FLAGS", "").split(),
              # gemm_sk.hip runs one wave per SIMD (512-register budget): without this flag the compiler selects the AGPR form of the
              # MFMA, keeps the accumulators' loop phis in VGPRs and shuttles every accumulator through v_accvgpr_write / _read around
              # each MFMA (774 moves in the kernel, 8 per MFMA in the K loop); with it the K loop has none.
              "gemm_sk.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}
# Round 6: no packed fp32 VALU (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) in the GEMM files.  The f32x4 expressions of the epilogues
# were selected as packed instructions (4.9 k / 1.5 k / 23.7 k of them in gemm / gemm_pp / gemm_dma); beside MFMAs -- gemm_pp's epilogue runs
# inside the next tile's main loop, the other kernels' next to the co-resident workgroup's -- a packed instruction costs more issue time
# than the two scalar ones it replaces (MI355X_MICROARCH.md, constants table).  Same box, four alternating rounds of 30 C3 steps
# (profiles/r06_experiments.md #13): 17.56 -> 17.34 ms median, bit-identical loss; the flag on the other files changes nothing.
NO_PK_F32 = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
for _f in ("gemm.hip", "gemm_pp.hip", "gemm_dma.hip"):
    FILE_FLAGS[_f] = FILE_FLAGS.get(_f, []) + NO_PK_F32


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found; the MI355X kernels cannot be built")


def _digest(src: Path) -> str:
    h = hashlib.sha256()
    for f in [src, CSRC / "common.hpp", CSRC / "gemm_common.hpp", CSRC / "gemm_ring.hpp", CSRC / "gemm_reg.hpp", CSRC / "gemm_sk_common.hpp", ROOT / "include" / "maestro_hip.h"]:
        h.update(f.read_bytes())
    h.update(" ".join(FLAGS + FILE_FLAGS.get(src.name, [])).encode())
    return h.hexdigest()[:16]


def build(force: bool = False, verbose: bool = False) -> Path:
    hipcc = _hipcc()
    OBJ_DIR.mkdir(exist_ok=True)
    LIB_DIR.mkdir(exist_ok=True)
    sources = sorted(CSRC.glob("*.hip"))
    jobs, objs = [], []
    for src in sources:
        obj = OBJ_DIR / f"{src.stem}.{_digest(src)}.o"
        objs.append(obj)
        if force or not obj.exists():
            for old in OBJ_DIR.glob(f"{src.stem}.*.o"):
                old.unlink()
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = [hipcc, *FLAGS, *FILE_FLAGS.get(src.name, []), "-c", str(src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src.name}:\n{r.stdout}\n{r.stderr}")
        # (the host half of the compilation does not know the device feature NO_PK_F32 names and says so: not a diagnostic of ours)
        return "\n".join(ln for ln in r.stderr.splitlines() if "'-packed-fp32-ops' is not a recognized feature" not in ln)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, os.cpu_count() or 1)) as ex:
            for warn in ex.map(compile_one, jobs):
                if verbose and warn:
                    print(warn, file=sys.stderr)
    if jobs or not LIB.exists():
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", *map(str, objs), "-o", str(LIB)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


EXAMPLE_SRC = ROOT / "examples" / "abi_smoke.cpp"
EXAMPLE_BIN = ROOT / "examples" / "abi_smoke"


def build_example(force: bool = False) -> Path:
    """The plain C++ caller of the C ABI (no torch, no Python): linked against the in-tree library, found through an
    $ORIGIN-relative rpath so that it runs from the repo snapshot on the GPU box."""
    if not force and EXAMPLE_BIN.exists() and EXAMPLE_BIN.stat().st_mtime >= max(EXAMPLE_SRC.stat().st_mtime, LIB.stat().st_mtime):
        return EXAMPLE_BIN
    cmd = [_hipcc(), "-O2", "-std=c++17", f"--offload-arch={ARCH}", str(EXAMPLE_SRC), f"-I{ROOT / 'include'}", f"-L{LIB_DIR}",
           "-lmaestro_hip", "-Wl,-rpath,$ORIGIN/../maestro_amd/lib", "-o", str(EXAMPLE_BIN)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"building examples/abi_smoke failed:\n{r.stdout}\n{r.stderr}")
    return EXAMPLE_BIN


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_example(force="--force" in sys.argv))
