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
ap.add_argument("--only", default="", help="comma-separated source names the extra --flags apply to (default: every file)")
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
flags = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-result"]
only = set(filter(None, a.only.split(",")))
extra = lambda name: a.flags.split() if not only or name in only else []
sys.path.insert(0, str(ROOT))
from maestro_amd.csrc.build import FILE_FLAGS          # the library's own per-file flags (attn.hip, gemm_sk.hip)
file_flags = {k: list(v) for k, v in FILE_FLAGS.items()}
file_flags["attn.hip"] = file_flags.get("attn.hip", []) + a.attn_flags.split()
srcs = sorted(csrc.glob("*.hip"))
def cc(src):
    obj = src.with_suffix(".o")
    r = subprocess.run(["hipcc", *flags, *extra(src.name), *file_flags.get(src.name, []), "-c", str(src), "-o", str(obj)], capture_output=True, text=True)
    if r.returncode: raise RuntimeError(r.stderr)
    return obj
with ThreadPoolExecutor(4) as ex: objs = list(ex.map(cc, srcs))
out = ROOT / "ab_old" / f"{a.name}.so"; out.parent.mkdir(exist_ok=True)
subprocess.run(["hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", *map(str, objs), "-o", str(out)], check=True)
shutil.rmtree(tmp)
print(out)
