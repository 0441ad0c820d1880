"""Index-algebra check of csrc/gemm_m32.hip in numpy (no GPU): the LDS image, the 32x32x16 fragment reads, the MFMA's documented
operand / accumulator lane maps (cdna_hip_programming.md: A[row l&31][k 8(l>>5)+j], B[k 8(l>>5)+j][col l&31], D col = l&31,
row = (reg&3) + 8(reg>>2) + 4(l>>5)) and the staged epilogue are restated formula by formula and one 128 x 128 x 128 tile is pushed
through them; the result must equal A W^T.  Checks the REASONING behind the kernel (a transcription slip between this file and
the .hip is still possible), and the bank spread of the fragment read over ds_read_b128's lane groups."""
import numpy as np

BM = BN = 128; BK = 64
rng = np.random.default_rng(0)
K = 128
A = rng.integers(-3, 4, (BM, K)).astype(np.float32)
W = rng.integers(-2, 3, (BN, K)).astype(np.float32)

def slot(row, c): return (c ^ ((row >> 1) & 7)) << 4

def store_tile(P, k0):
    """256 threads: thread t writes chunk t&7 (8 bf16 = 16 B) of rows (t>>3) + 32 i; returns the byte-addressed image as fp32-per-bf16."""
    img = np.full(128 * 64, np.nan, np.float32)          # one entry per bf16 element: address = byte / 2
    for t in range(256):
        c, r = t & 7, t >> 3
        for i in range(4):
            row = r + 32 * i
            base = (row * 128 + slot(row, c)) // 2
            img[base:base + 8] = P[row, k0 + 8 * c:k0 + 8 * c + 8]
    assert not np.isnan(img).any()
    return img

def read_frag(img, rc0, s):
    """[64 lanes][8]: row rc0 + (l & 31), chunk 2 s + (l >> 5)."""
    out = np.empty((64, 8), np.float32)
    for l in range(64):
        row = rc0 + (l & 31)
        base = (row * 128 + slot(row, 2 * s + (l >> 5))) // 2
        out[l] = img[base:base + 8]
    return out

def mfma_32x32x16(a, b, acc):
    """a, b: [64][8] lane fragments; acc: [64][16].  D[i][j] += sum_k Amat[i][k] Bmat[k][j]."""
    Am = np.zeros((32, 16), np.float32); Bm = np.zeros((16, 32), np.float32)
    for l in range(64):
        r, h = l & 31, l >> 5
        Am[r, 8 * h:8 * h + 8] = a[l]
        Bm[8 * h:8 * h + 8, r] = b[l]
    D = Am @ Bm
    for l in range(64):
        for reg in range(16):
            acc[l, reg] += D[(reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5), l & 31]

C = np.zeros((BM, BN), np.float32)
acc = {(w, j, i): np.zeros((64, 16), np.float32) for w in range(4) for j in range(2) for i in range(2)}
for kt in range(K // BK):
    ta, tb = store_tile(A, kt * BK), store_tile(W, kt * BK)
    for w in range(4):
        wm, wn = (w >> 1) * 64, (w & 1) * 64
        for s in range(4):
            fa = [read_frag(ta, wm + 32 * i, s) for i in range(2)]
            fb = [read_frag(tb, wn + 32 * j, s) for j in range(2)]
            for j in range(2):
                for i in range(2):
                    mfma_32x32x16(fb[j], fa[i], acc[(w, j, i)])
# epilogue staging: st[mm * 68 + 32 jn + 8 q + 4 h + e] = acc[jn][im][4 q + e]; read back row-major (r, c..c+7)
for w in range(4):
    wm, wn = (w >> 1) * 64, (w & 1) * 64
    for im in range(2):
        st = np.full(32 * 68, np.nan, np.float32)
        for l in range(64):
            mm, h = l & 31, l >> 5
            for jn in range(2):
                for q in range(4):
                    for e in range(4):
                        st[mm * 68 + 32 * jn + 8 * q + 4 * h + e] = acc[(w, jn, im)][l, 4 * q + e]
        for l in range(64):
            c = (l & 7) * 8
            for p in range(4):
                r = p * 8 + (l >> 3)
                C[wm + 32 * im + r, wn + c:wn + c + 8] = st[r * 68 + c:r * 68 + c + 8]
assert np.array_equal(C, A @ W.T), np.abs(C - A @ W.T).max()
print("tile result equals A W^T")

# bank spread of one fragment read: ds_read_b128 lane groups, bank row = 256 B, a 16-B slot = 4 banks
groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
groups += [[l + 32 for l in g] for g in groups]
for rc0 in (0, 32, 64, 96):
    for s in range(4):
        for g in groups:
            slots = {((rc0 + (l & 31)) * 128 + slot(rc0 + (l & 31), 2 * s + (l >> 5))) % 256 // 16 for l in g}
            assert len(slots) == 16, (rc0, s, g, slots)
print("fragment reads: 16 distinct 16-B slots per ds_read_b128 lane group (conflict-free)")
# the same read on gemm.hip's image (XOR (row & 7)) would be 2-way:
worst = max(16 // len({((l & 31) * 128 + ((((0) ^ ((l & 31) & 7)) << 4))) % 256 // 16 for l in g}) for g in groups)
print(f"with the (row & 7) swizzle of gemm_reg.hpp the same read is {worst}-way")
