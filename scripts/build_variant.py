"""Build ANOTHER libmaestro_hip.so next to the shipped one, for same-box A/B runs through scripts/ab_lib.py:
    python scripts/build_variant.py <name> [--rev <git rev>] [--flags "<extra hipcc flags>"] [--attn-flags "..."]
The csrc tree of <git rev> (default: the working tree) is copied to a scratch directory, compiled with the library's own flags
plus the extra ones, and linked to ab_old/<name>.so (git-ignored, shipped to the GPU box by gpurun)."""
import argparse, os, shutil, subprocess, sys, tempfile
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
ap = argparse.ArgumentParser()
ap.add_argument("name"); ap.add_argument("--rev", default=None); ap.add_argument("--flags", default=""); ap.add_argument("--attn-flags", default="")
a = ap.parse_args()
tmp = Path(tempfile.mkdtemp(prefix=f"variant_{a.name}_"))
(tmp / "maestro_amd").mkdir(); (tmp / "include").mkdir()
if a.rev:
    for sub in ("maestro_amd/csrc", "include"):
        out = subprocess.run(["git", "-C", str(ROOT), "archive", a.rev, sub], capture_output=True, check=True).stdout
        subprocess.run(["tar", "-x", "-C", str(tmp)], input=out, check=True)
else:
    shutil.copytree(ROOT / "maestro_amd" / "csrc", tmp / "maestro_amd" / "csrc", ignore=shutil.ignore_patterns("build", "__pycache__"), dirs_exist_ok=True)
    shutil.copytree(ROOT / "include", tmp / "include", dirs_exist_ok=True)
csrc = tmp / "maestro_amd" / "csrc"
flags = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-result"] + a.flags.split()
attn = ["-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-slp-vectorize"] + a.attn_flags.split()
srcs = sorted(csrc.glob("*.hip"))
def cc(src):
    obj = src.with_suffix(".o")
    r = subprocess.run(["hipcc", *flags, *(attn if src.name == "attn.hip" else []), "-c", str(src), "-o", str(obj)], capture_output=True, text=True)
    if r.returncode: raise RuntimeError(r.stderr)
    return obj
with ThreadPoolExecutor(4) as ex: objs = list(ex.map(cc, srcs))
out = ROOT / "ab_old" / f"{a.name}.so"; out.parent.mkdir(exist_ok=True)
subprocess.run(["hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", *map(str, objs), "-o", str(out)], check=True)
shutil.rmtree(tmp)
print(out)
