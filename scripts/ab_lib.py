"""A/B aid: run a script of this repository against ANOTHER build of libmaestro_hip.so (same ABI, e.g. one compiled with
-DMH_LN_FAST=0):  python scripts/ab_lib.py <path/to/lib.so> <script.py> [script args ...]
       python scripts/ab_lib.py <path/to/lib.so> -m pytest tests/test_kernels_gpu.py -k layernorm -q
The library is loaded lazily by maestro_amd.hip, so pointing its path elsewhere before the first call is enough."""
import os, runpy, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathlib import Path
from maestro_amd import hip
lib, script = Path(sys.argv[1]).resolve(), sys.argv[2]
assert lib.exists(), lib
hip._LIB_PATH = lib
if script == "-m":
    sys.argv = sys.argv[3:]
    runpy.run_module(sys.argv[0], run_name="__main__", alter_sys=True)
else:
    sys.argv = [script] + sys.argv[3:]
    runpy.run_path(script, run_name="__main__")
