import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maestro_amd import hip
dev = torch.device("cuda:0")
layout, M, N, K = (int(x) for x in sys.argv[1:5])
impl = sys.argv[5] if len(sys.argv) > 5 else "v1"
epi = sys.argv[6] if len(sys.argv) > 6 else "plain"      # "gelu": the fc1 epilogue (bias + GELU + byte-coded derivative)
A = torch.randn((M, K) if layout < 2 else (K, M), device=dev).bfloat16()
B = torch.randn((N, K) if layout == 0 else (K, N), device=dev).bfloat16()
C = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
os.environ["MH_GEMM_DMA"] = "0"
kw = {"tile": int(impl[4:])} if impl.startswith("tile") else {}      # "tile<k>": an explicit MH_TILE_* id (7 = the ping-pong tile, 8.. its diagnostic builds)
bias, aux = torch.randn(N, device=dev), torch.empty(M, N, device=dev, dtype=torch.uint8)
for _ in range(20):
    if epi == "gelu":
        hip.gemm(layout, M, N, K, A, A.shape[1], B, B.shape[1], C, N, hip.BIAS | hip.GELU | hip.AUX_DGELU | hip.AUX_U8, bias=bias,
                 aux_out=aux, ldaux=N, **kw)
    else:
        hip.gemm(layout, M, N, K, A, A.shape[1], B, B.shape[1], C, N, 0, **kw)
torch.cuda.synchronize()
