"""In-step A/B of a GEMM tile override without rebuilding the library: pre-populates maestro_amd.hip's per-signature tile table for the
signatures matched below, then runs bench.py's main with the remaining arguments.
  python scripts/ab_tile_override.py <rule> [bench args]     rule: none | nnpp (plain NN dgrads on the persistent ping-pong tile) | nn768 (NN, N = 768, K >= 2304, M = 8192 -> DMA-fed 4-wave 128 x 128 tile)"""
import runpy
import sys
from pathlib import Path

root = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(root))
from maestro_amd import hip  # noqa: E402

rule = sys.argv[1]
inner = hip._pick_tile
seen = {}


def pick(layout, M, N, K, flags, args):  # noqa: N803
    if rule == "nn768" and layout == 1 and N == 768 and K >= 2304 and M == 8192 and not (flags & (hip.MULAUX | hip.ATOMIC)):
        seen[(M, N, K, flags)] = seen.get((M, N, K, flags), 0) + 1
        return hip.TILE_DMA_128x4
    if rule == "nn768b" and layout == 1 and N == 768 and M in (8192, 3200) and not (flags & (hip.MULAUX | hip.ATOMIC)):
        seen[(M, N, K, flags)] = seen.get((M, N, K, flags), 0) + 1
        return hip.TILE_DMA_128x4 if K >= 2304 else hip.TILE_REG_64
    if rule == "nnpp" and layout == 1 and flags == 0 and K % 64 == 0 and K >= 512 and N % 128 == 0 and -(-M // 128) * (N // 128) >= 256:
        seen[(M, N, K, flags)] = seen.get((M, N, K, flags), 0) + 1
        return hip.TILE_PP_128
    if rule == "m3200_64" and M == 3200 and layout != 2 and not (flags & (hip.COLSUM | hip.ATOMIC)):
        seen[(M, N, K, flags)] = seen.get((M, N, K, flags), 0) + 1
        return hip.TILE_REG_64
    if rule == "n768_64" and N == 768 and M == 8192 and K <= 768 and layout != 2 and not (flags & (hip.COLSUM | hip.ATOMIC)):
        seen[(M, N, K, flags)] = seen.get((M, N, K, flags), 0) + 1
        return hip.TILE_REG_64
    if rule == "dec_pp" and layout == 0 and M == 32768 and not (flags & (hip.GELU | hip.COLSUM | hip.ATOMIC | hip.MULAUX)):
        seen[(M, N, K, flags)] = seen.get((M, N, K, flags), 0) + 1
        return hip.TILE_PP_128
    if rule == "dec_dma" and layout == 0 and M in (32768, 12800) and K % 32 == 0 and not (flags & (hip.COLSUM | hip.ATOMIC)):
        seen[(M, N, K, flags)] = seen.get((M, N, K, flags), 0) + 1
        return hip.TILE_DMA_256
    return inner(layout, M, N, K, flags, args)


hip._pick_tile = pick
sys.argv = [str(root / "bench.py")] + sys.argv[2:]
try:
    runpy.run_path(str(root / "bench.py"), run_name="__main__")
finally:
    print(f"[ab_tile_override] rule {rule}: overridden signatures {seen}", file=sys.stderr)
