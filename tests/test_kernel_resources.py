"""Properties of the BUILT kernels that need no GPU: register spills / scratch / LDS from the code objects' metadata, and the
issue order of the straight-line LayerNorm and column-sum kernels from their ISA (`hipcc -S`): every load of a wave leaves before
the first `s_waitcnt vmcnt` -- the property whose absence cost the generic kernels a memory round trip per chunk
(profiles/r03_ln_straightline.txt)."""

import re
import shutil
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "scripts"))

# forms of the ping-pong GEMM that are built but not dispatched by default because they spill
KNOWN_SPILLS = {"gemm_pp_kernel<true, 1, 0>", "gemm_pp_kernel<true, 2, 0>", "gemm_pp_kernel<false, 2, 0>"}


@pytest.fixture(scope="module")
def kernels():
    import kernel_resources as kr
    if not kr.LLVM.exists() or not list((ROOT / "maestro_amd" / "csrc" / "build").glob("*.o")):
        pytest.skip("needs the built objects (python -m maestro_amd.csrc.build) and the ROCm LLVM tools")
    ks = kr.library_kernels()
    assert len(ks) > 100
    return ks


def test_no_kernel_on_the_default_path_spills_or_uses_scratch(kernels):
    bad = [k["kernel"] for k in kernels
           if (k["vgpr_spill_count"] or k["private_segment_fixed_size"]) and k["kernel"] not in KNOWN_SPILLS]   # (SGPR spills go to VGPR lanes)
    assert not bad, bad


def test_lds_fits_a_cu_and_workgroups_fit_their_registers(kernels):
    for k in kernels:
        assert k["group_segment_fixed_size"] <= 160 * 1024, k["kernel"]
        assert k["vgpr_count"] <= 512, k["kernel"]


def test_straight_line_layernorm_kernels_are_in_the_library(kernels):
    names = {k["kernel"]: k for k in kernels}
    for nv in (1, 2, 3, 4):
        for fp8 in ("false", "true"):       # with and without the e4m3 copy (mh_layernorm_fwd_fp8): one kernel, two instantiations
            form = f"ln_fwd_fast_kernel<{nv}, {fp8}>"
            assert form in names and names[form]["private_segment_fixed_size"] == 0, form
    for form in ("ln_bwd_fast_kernel<1, 4>", "ln_bwd_fast_kernel<2, 4>", "ln_bwd_fast_kernel<3, 4>", "ln_bwd_fast_kernel<4, 2>"):
        assert form in names and names[form]["vgpr_count"] <= 256 and names[form]["private_segment_fixed_size"] == 0, form


def test_stream_k_kernels_keep_their_accumulators_in_registers(kernels):
    """Round 6: the eight-wave stream-K kernel has 256 registers per lane, 128 of them accumulators; written the obvious way
    (accumulators updated inside one arm of a branch) it spilled 30-250 registers -- some of them the K loop's fragment addresses,
    reloaded from scratch behind vmcnt(0) in every K step (profiles/r06_experiments.md #3).  All four instantiations must stay
    spill-free inside two waves per SIMD; the four-wave kernels inside one wave per SIMD."""
    names = {k["kernel"]: k for k in kernels}
    for b_kmajor in ("false", "true"):
        for epi in (0, 1):
            k = names.get(f"gemm_sk_dma_kernel<{b_kmajor}, {epi}>")
            assert k is not None, sorted(n for n in names if "gemm_sk" in n)
            assert k["vgpr_spill_count"] == 0 and k["private_segment_fixed_size"] == 0 and k["vgpr_count"] <= 256, k
            for mt in (6, 8):
                k4 = names.get(f"gemm_sk_kernel<{b_kmajor}, {mt}, {epi}>")
                assert k4 is not None and k4["vgpr_spill_count"] == 0 and k4["private_segment_fixed_size"] == 0, (b_kmajor, mt, epi, k4)


def _kernel_bodies(asm: str):
    cur, body, out = None, [], {}
    for ln in asm.splitlines():
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            cur, body = m.group(1), []
            out[cur] = body
        elif cur is not None:
            body.append(ln)
            if "s_endpgm" in ln and ln.startswith("\t"):
                pass
    return out


def _loads_before_first_vm_wait(body):
    n = 0
    for ln in body:
        if re.search(r"\b(global|flat|buffer)_load", ln):
            n += 1
        elif re.search(r"s_waitcnt.*vmcnt", ln):
            return n
    return n


def test_straight_line_kernels_issue_their_loads_before_the_first_wait():
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(hipcc).exists():
        pytest.skip("needs hipcc")
    from maestro_amd.csrc import build as B
    flags = [f for f in B.FLAGS if f != "-fPIC"]
    r = subprocess.run([hipcc, *flags, "--cuda-device-only", "-S", "-o", "-", str(B.CSRC / "norm.hip")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    bodies = _kernel_bodies(r.stdout)

    def body_of(fragment):
        hits = [b for n, b in bodies.items() if fragment in n]
        assert len(hits) == 1, (fragment, [n for n in bodies if fragment in n])
        return hits[0]
    # backward, dim 768: gamma (3) + 4 rows x (x, dy, dres) x 3 chunks = 39 loads in flight before anything is waited for
    assert _loads_before_first_vm_wait(body_of("ln_bwd_fast_kernelILi3ELi4E")) >= 39
    assert _loads_before_first_vm_wait(body_of("ln_bwd_fast_kernelILi2ELi4E")) >= 26
    # forward: the whole row (and at least part of gamma / beta) before the first wait
    for fp8 in ("Lb0E", "Lb1E"):
        assert _loads_before_first_vm_wait(body_of("ln_fwd_fast_kernelILi3E" + fp8)) >= 3
        assert _loads_before_first_vm_wait(body_of("ln_fwd_fast_kernelILi2E" + fp8)) >= 2
    # batched column sums: the chunk's 16 row loads together
    assert _loads_before_first_vm_wait(body_of("colsum_batched_kernel")) >= 16
    # and the generic backward still shows the pattern the straight-line form removes (documents the finding; if a compiler
    # update ever fixes it, this line fails and the comment in norm.hip can go)
    generic = body_of("ln_bwd_kernelILi3E")
    assert sum("s_waitcnt vmcnt(0)" in ln for ln in generic) >= 4
