"""Which loops of the HBM-bound kernels wait for EVERY load before issuing the next one?  (no GPU needed)

    python scripts/isa_waits.py [source ...]          (default: norm embed loss mask heads quant)

For every loop (label ... backward branch) of every non-MFMA kernel: global / flat / buffer loads per trip and the `s_waitcnt
vmcnt(N)` it contains.  A loop with one load per trip followed by `vmcnt(0)` is a chain of dependent memory round trips unless
enough other waves hide it; a loop with several loads and only `vmcnt(0)` waits between them serialises them.  This is the scan
that found the LayerNorm and column-sum kernels of profiles/r03_ln_straightline.txt.  Straight-line kernels (no loop) are listed
with the number of loads before their first vmcnt wait."""
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from maestro_amd.csrc import build as B  # noqa: E402


def demangle(name):
    out = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    return re.sub(r"\(.*$", "", out.replace("(anonymous namespace)::", "").replace("void ", ""))


def scan(src: Path):
    flags = [f for f in B.FLAGS + B.FILE_FLAGS.get(src.name, []) if f != "-fPIC"]
    asm = subprocess.run([B._hipcc(), *flags, "--cuda-device-only", "-S", "-o", "-", str(src)], capture_output=True, text=True)
    if asm.returncode != 0:
        raise SystemExit(asm.stderr)
    cur, body, kernels = None, [], []
    for ln in asm.stdout.splitlines():
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            cur, body = m.group(1), []
            kernels.append((cur, body))
        elif cur is not None:
            body.append(ln)
            if "s_endpgm" in ln:
                cur = None
    for name, body in kernels:
        instr, labels = [], {}
        for ln in body:
            m = re.match(r"^(\.LBB\w+):", ln)
            if m:
                labels[m.group(1)] = len(instr)
                continue
            m = re.match(r"^\t([a-z_0-9]+)(.*)", ln)
            if m and not ln.startswith("\t."):
                instr.append((m.group(1), m.group(2)))
        if any(o.startswith("v_mfma") for o, _ in instr):
            continue
        is_load = lambda o: re.match(r"(global|flat|buffer)_load", o) is not None  # noqa: E731
        rows = []
        for i, (o, rest) in enumerate(instr):
            m = re.search(r"(\.LBB\w+)", rest) if o.startswith(("s_cbranch", "s_branch")) else None
            if m and m.group(1) in labels and labels[m.group(1)] <= i:
                seg = instr[labels[m.group(1)]:i + 1]
                loads = sum(is_load(x) for x, _ in seg)
                waits = [int(re.search(r"vmcnt\((\d+)\)", r).group(1)) for x, r in seg if x == "s_waitcnt" and "vmcnt" in r]
                if loads:
                    rows.append((len(seg), loads, waits))
        first = 0
        for o, rest in instr:
            if is_load(o):
                first += 1
            elif o == "s_waitcnt" and "vmcnt" in rest:
                break
        total = sum(is_load(o) for o, _ in instr)
        print(f"{src.stem:6s} {demangle(name)[:58]:58s} loads {total:3d}, {first:3d} before the first vm wait", end="")
        for n, loads, waits in rows:
            flag = "  <-- every load waited for" if waits and all(w == 0 for w in waits) and len(waits) >= loads else ""
            print(f"\n         loop of {n:4d} instr: {loads:2d} loads, vmcnt waits {waits}{flag}", end="")
        print()


if __name__ == "__main__":
    for w in sys.argv[1:] or ["norm", "embed", "loss", "mask", "heads", "quant"]:
        scan(B.CSRC / f"{w}.hip")
