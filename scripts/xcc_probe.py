"""Which XCDs do the workgroups of a CU-masked stream land on?  A census kernel (hipModule-free: a tiny HIP source compiled with hipcc
at run time is avoided; we use the library's own mh_xcc_census entry) is not available, so this probe times a GEMM on masked streams
of 8 / 6 / 4 / 2 XCDs: the time must scale with the CU count if the mask is honoured."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maestro_amd import hip  # noqa: E402

dev = torch.device("cuda:0")
rt = ctypes.CDLL("libamdhip64.so")
ncu = torch.cuda.get_device_properties(dev).multi_processor_count
M, N, K = 16384, 3072, 768
A = torch.randn(M, K, device=dev).bfloat16()
W = torch.randn(N, K, device=dev).bfloat16()
C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)


def masked(xcds):
    words = [0] * ((ncu + 31) // 32)
    for i in range(ncu):
        if i % 8 in xcds:
            words[i // 32] |= 1 << (i % 32)
    arr = (ctypes.c_uint32 * len(words))(*words)
    h = ctypes.c_void_p()
    rc = rt.hipExtStreamCreateWithCUMask(ctypes.byref(h), ctypes.c_uint32(len(words)), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(h.value, device=dev)


for name, xcds in (("all 8", set(range(8))), ("6 (0-5)", set(range(6))), ("4 (0-3)", set(range(4))), ("2 (6-7)", {6, 7}), ("1 (3)", {3})):
    st = masked(xcds)
    with torch.cuda.stream(st):
        for _ in range(3):
            hip.gemm(0, M, N, K, A, K, W, K, C, N, 0, tile=hip.TILE_REG_128)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            hip.gemm(0, M, N, K, A, K, W, K, C, N, 0, tile=hip.TILE_REG_128)
        e1.record()
    torch.cuda.synchronize()
    print(f"XCDs {name:8s}: {e0.elapsed_time(e1) / 10 * 1e3:8.1f} us per GEMM ({M}, {N}, {K})", flush=True)
# inside a captured graph?
st = masked({6, 7})
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(st):
    hip.gemm(0, M, N, K, A, K, W, K, C, N, 0, tile=hip.TILE_REG_128)
    st.synchronize()
    with torch.cuda.graph(g, stream=st):
        for _ in range(10):
            hip.gemm(0, M, N, K, A, K, W, K, C, N, 0, tile=hip.TILE_REG_128)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
torch.cuda.synchronize()
print(f"graph captured on the 2-XCD stream, replayed on it: {e0.elapsed_time(e1) / 10 * 1e3:8.1f} us per GEMM (mask kept if ~ the 2-XCD time)")
