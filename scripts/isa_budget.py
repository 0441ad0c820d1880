"""Static issue budget of the MFMA kernels from their gfx950 ISA (no GPU needed).

    python scripts/isa_budget.py [gemm|gemm_pp|gemm_dma|attn ...]      (default: all four)

Compiles the .hip source to assembly with the build's own flags (`hipcc --cuda-device-only -S`), finds every loop of every
kernel (label ... backward branch) and prints, for the loops that contain MFMAs and for the whole kernel body, the instruction
counts by issue class and the cycles they cost ONE wave on its SIMD by the constants of /opt/skills/guides/MI355X_MICROARCH.md
("Cycles per instruction" table):

    v_mfma_f32_16x16x32_bf16   16 cycles, holds the SIMD's vector issue for 8 of them
    v_mfma_f32_32x32x16_bf16   32 cycles, holds the vector issue for 8 of them
    v_mfma_scale_*16x16x128*   32 cycles (twice the bf16 form of the same M x N), 8 held
    plain VALU                  4 cycles of vector issue;  v_exp / v_rcp / v_rsq / v_sqrt / v_log: 8

    mfma    = sum of MFMA cycles                  (the floor of the loop when nothing else is in the way)
    vector  = VALU issue cycles + 8 per MFMA      (what the vector issue port is busy for)
    model   = max(mfma, vector)                   (guide: "an MFMA gap runs ~ max(32 or 16, their sum)")

A loop whose `vector` exceeds `mfma` is VALU-bound however it is scheduled; `free` = mfma - vector is the VALU issue room that is
left under the MFMAs (negative: none).  LDS / VMEM / SALU instructions are counted but not priced (separate issue ports).
The numbers are per wave and per loop trip; they say nothing about waits (s_waitcnt / s_barrier counts are listed for that).
"""
import collections
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from maestro_amd.csrc import build as B  # noqa: E402

TRANS = ("v_exp", "v_rcp", "v_rsq", "v_sqrt", "v_log", "v_sin", "v_cos")


def mfma_cycles(op: str) -> int:
    if "scale" in op or "f8f6f4" in op:
        return 32 if "16x16" in op else 64
    if "32x32" in op:
        return 32
    return 16


def classify(op: str) -> str:
    if op.startswith("v_mfma") or op.startswith("v_smfmac"):
        return "mfma"
    if op.startswith(TRANS):
        return "trans"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_"):
        return "salu"
    return "other"


def budget(ops):
    n = collections.Counter(classify(o) for o in ops)
    mf = sum(mfma_cycles(o) for o in ops if classify(o) == "mfma")
    vec = 4 * n["valu"] + 8 * n["trans"] + 8 * n["mfma"]
    return n, mf, vec


def demangle(name: str) -> str:
    out = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    out = out.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*$", "", out)


def kernels(asm: str):
    cur, body = None, []
    for ln in asm.splitlines():
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            cur, body = m.group(1), []
            continue
        if cur is None:
            continue
        body.append(ln)
        if "s_endpgm" in ln and not any(re.match(r"^\.LBB", b) for b in body[-1:]):
            pass
        if ln.startswith("\t.section") or ln.startswith(".Lfunc_end"):
            yield cur, body
            cur = None


def analyse(src: Path):
    flags = B.FLAGS + B.FILE_FLAGS.get(src.name, [])
    flags = [f for f in flags if f != "-fPIC"]
    asm = subprocess.run([B._hipcc(), *flags, "--cuda-device-only", "-S", "-o", "-", str(src)], capture_output=True, text=True)
    if asm.returncode != 0:
        raise SystemExit(asm.stderr)
    for name, body in kernels(asm.stdout):
        instr = []          # (line index, opcode)
        labels = {}
        for i, ln in enumerate(body):
            m = re.match(r"^(\.LBB\w+):", ln)
            if m:
                labels[m.group(1)] = len(instr)
                continue
            m = re.match(r"^\t([a-z_0-9]+)", ln)
            if m and not ln.startswith("\t."):
                instr.append((m.group(1), ln))
        ops = [o for o, _ in instr]
        n, mf, vec = budget(ops)
        if not n["mfma"]:
            continue
        print(f"\n{demangle(name)}")
        print(f"  whole body : {n['mfma']:4d} MFMA  {n['valu']:5d} VALU {n['trans']:3d} trans  {n['lds']:4d} LDS {n['vmem']:4d} VMEM "
              f"{n['salu']:5d} SALU {n['wait']:4d} waits {n['barrier']:2d} barriers")
        loops = []
        for i, (o, ln) in enumerate(instr):
            m = re.search(r"s_c?branch\w*\s+(\.LBB\w+)", ln)
            if m and m.group(1) in labels and labels[m.group(1)] <= i:
                loops.append((labels[m.group(1)], i))
        seen = set()
        for a, b in sorted(loops, key=lambda t: (t[0], -t[1])):
            lo = ops[a:b + 1]
            ln_, lm, lv = budget(lo)
            if not ln_["mfma"] or (a, b) in seen:
                continue
            seen.add((a, b))
            kinds = collections.Counter(o for o in lo if classify(o) == "mfma")
            kind = ", ".join(f"{c} x {k.replace('v_mfma_', '')}" for k, c in kinds.items())
            print(f"  loop {a:5d}-{b:5d}: {kind}")
            print(f"      mfma {lm:5d} cyc | vector {lv:5d} cyc ({ln_['valu']} VALU, {ln_['trans']} trans) | free {lm - lv:6d} | model {max(lm, lv):5d} "
                  f"| {ln_['lds']} LDS {ln_['vmem']} VMEM {ln_['salu']} SALU {ln_['wait']} waits {ln_['barrier']} barriers")
        inside = set()
        for a, b in seen:
            inside.update(range(a, b + 1))
        rest = [o for i, o in enumerate(ops) if i not in inside]
        rn, rm, rv = budget(rest)
        print(f"  outside the MFMA loops (prologue + epilogue, ALL epilogue forms of the template): {rn['mfma']} MFMA, {rn['valu']} VALU, "
              f"{rn['trans']} trans, {rn['lds']} LDS, {rn['vmem']} VMEM, {rn['wait']} waits")


if __name__ == "__main__":
    which = sys.argv[1:] or ["gemm", "gemm_pp", "gemm_dma", "attn"]
    for w in which:
        print(f"==================== {w}.hip")
        analyse(B.CSRC / f"{w}.hip")
