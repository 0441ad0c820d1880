import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from maestro_amd import hip
dev = torch.device("cuda:0")
here = os.path.dirname(os.path.abspath(__file__))
old = ctypes.CDLL(os.path.join(here, "old", "libold.so"))
def layer(Mtok, dim, mlp, inner):
    return [(dim, mlp, Mtok), (mlp, dim, Mtok), (dim, inner, Mtok), (3 * inner, dim, Mtok)]
sets = {"enc": [s for _ in range(9) for s in layer(8192, 768, 3072, 768) + layer(3200, 768, 3072, 768)],
        "dec": [s for _ in range(3) for s in layer(32768, 512, 3072, 512) + layer(12800, 512, 3072, 512)]}
for name, shapes in sets.items():
    probs = []
    for (M, N, K) in shapes:
        A = torch.randn(K, M, device=dev).bfloat16(); B = torch.randn(K, N, device=dev).bfloat16()
        probs.append((A, B, torch.zeros(M, N, device=dev), M, N, K, M, N, N))
    g = hip.GroupedTN(probs, dev)
    def new(): g.launch()
    def oldf():
        rc = old.mh_gemm_grouped_tn(ctypes.c_void_p(g.table.data_ptr()), ctypes.c_int(g.n), ctypes.c_void_p(g.queues.data_ptr()), ctypes.c_int(g.queue_len),
                                    ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0
    for label, f in (("old", oldf), ("new", new), ("old", oldf), ("new", new)):
        for _ in range(2): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"{name} {label}: {ms:6.3f} ms {g.flops/ms/1e9:6.1f} TF", flush=True)
for (M, N, K) in ((32768, 512, 3072), (32768, 3072, 512), (32768, 1536, 512)):
    A = torch.randn(M, K, device=dev).bfloat16(); W = torch.randn(N, K, device=dev).bfloat16(); C = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    for label, lib in (("old", old), ("new", hip.lib())):
        def f():
            rc = lib.mh_gemm_bf16_tile(ctypes.c_int(1), ctypes.c_int(0), ctypes.c_int(M), ctypes.c_int(N), ctypes.c_int(K), ctypes.c_void_p(A.data_ptr()), ctypes.c_int(K),
                                       ctypes.c_void_p(W.data_ptr()), ctypes.c_int(K), ctypes.c_void_p(C.data_ptr()), ctypes.c_int(N), ctypes.c_int(0), None, None, ctypes.c_int(0),
                                       None, None, ctypes.c_int(0), None, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            assert rc == 0
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        print(f"d256 NT ({M},{N},{K}) {label}: {2.0*M*N*K/(e0.elapsed_time(e1)/20)/1e9:6.1f} TF", flush=True)
