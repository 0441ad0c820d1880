"""Register / scratch / LDS use of every kernel of the built library, read from the code objects' metadata (no GPU):

    python scripts/kernel_resources.py [--all]

Reads maestro_amd/csrc/build/*.o (the per-source objects of `python -m maestro_amd.csrc.build`): the .hip_fatbin section is
unbundled with clang-offload-bundler and the AMDGPU metadata note is read with llvm-readelf.  Default output: kernels that spill,
use scratch, or run at one wave per SIMD; --all prints every kernel.  tests/test_kernel_resources.py asserts on the same data.
"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
LLVM = Path("/opt/rocm/lib/llvm/bin")
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"
FIELDS = ("vgpr_count", "agpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
          "group_segment_fixed_size", "max_flat_workgroup_size")


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return [re.sub(r"\(.*$", "", o.replace("(anonymous namespace)::", "").replace("void ", "")) for o in out]


def kernels_of(obj: Path):
    """[{name, vgpr_count, ...}] of one host object with an embedded gfx950 code object."""
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = Path(tmp) / "fat.bin", Path(tmp) / "dev.co"
        r = subprocess.run([LLVM / "llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", obj], capture_output=True, text=True)
        if r.returncode != 0 or not fat.exists() or fat.stat().st_size == 0:
            return []
        subprocess.run([LLVM / "clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}", f"--targets={TARGET}",
                        f"--output={co}"], check=True, capture_output=True)
        notes = subprocess.run([LLVM / "llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
    out, cur = [], None
    for ln in notes.splitlines():
        m = re.match(r"\s+-?\s*\.(\w+):\s+(\S+)\s*$", ln)
        if not m:
            continue
        key, val = m.groups()
        if key in FIELDS or key == "name":
            if cur is None or (key in cur):
                cur = {}
                out.append(cur)
            cur[key] = val if key == "name" else int(val)
    out = [k for k in out if "name" in k and "vgpr_count" in k]
    for k, d in zip(out, demangle([k["name"] for k in out])):
        k["kernel"] = d
    return out


def library_kernels():
    objs = sorted((ROOT / "maestro_amd" / "csrc" / "build").glob("*.o"))
    res = []
    for o in objs:
        for k in kernels_of(o):
            k["source"] = o.name.split(".")[0]
            res.append(k)
    return res


def waves_per_simd(k) -> int:
    regs = max(k["vgpr_count"], 1)       # gfx90a+: the unified total (architected + accumulation registers)
    return max(1, min(8, 512 // (((regs + 7) // 8) * 8)))


if __name__ == "__main__":
    ks = library_kernels()
    if not ks:
        raise SystemExit("no objects under maestro_amd/csrc/build: run `python -m maestro_amd.csrc.build` first")
    show_all = "--all" in sys.argv
    print(f"{len(ks)} kernels in {len({k['source'] for k in ks})} sources")
    print(f"{'kernel':70s} {'VGPR':>5s} {'AGPR':>5s} {'spill':>5s} {'scratch':>7s} {'LDS':>7s} waves/SIMD")
    for k in ks:
        flag = k["vgpr_spill_count"] or k["private_segment_fixed_size"] or waves_per_simd(k) == 1
        if show_all or flag:
            print(f"{k['source'] + ': ' + k['kernel']:70.70s} {k['vgpr_count']:5d} {k.get('agpr_count', 0):5d} {k['vgpr_spill_count']:5d} "
                  f"{k['private_segment_fixed_size']:7d} {k['group_segment_fixed_size']:7d} {waves_per_simd(k)}")
