// EXPERIMENTAL (round 3, written without GPU time left: compiled and ISA-checked, NOT yet run on hardware; reachable only through
// the explicit tile MH_TILE_M32_128 of mh_gemm_bf16_tile, never picked by MH_TILE_AUTO; tests/test_gemm_m32_gpu.py is gated
// behind MAESTRO_TEST_EXPERIMENTAL=1).
//
// gemm_kernel's 128 x 128 x 64 NT tile on v_mfma_f32_32x32x16_bf16 instead of v_mfma_f32_16x16x32_bf16, to test one hypothesis on
// the real fc1 epilogue (profiles/r03_isa_budget.txt): a 16x16x32 MFMA holds the SIMD's vector issue port for 8 of its 16 cycles,
// a 32x32x16 MFMA for 8 of its 32 at the same flops per cycle, so the same tile blocks the port for 128 instead of 256 of a K
// step's 512 MFMA cycles -- and the bias + erf-GELU + byte-coded GELU' epilogue (1020 VALU + 130 transcendental instructions
// per wave and tile), which does not fit beside 16x16x32 MFMAs (+33 % predicted, +30 % measured), nearly fits beside these (+8 %).
// Two workgroups per CU as in gemm_kernel: one's epilogue runs under the other's main loop on the same SIMDs.
//
// What differs from gemm.hip (whose building blocks -- buffer loads, the software pipeline, the staged row-major epilogue -- are
// reused or restated):
//   * LDS image of a K-minor operand tile: 128-B rows, 16-B chunk index XOR ((row >> 1) & 7) instead of XOR (row & 7).  A 32x32x16
//     fragment read takes ONE chunk of 32 consecutive rows (lane l: row l & 31, chunk 2 s + (l >> 5)); ds_read_b128 is served in
//     the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+32): with (row & 7) the 16 rows of a group fall on 8 distinct
//     16-B slots (2-way), with ((row >> 1) & 7) -- the row's parity already picks the half of the 256-B bank row -- on 16.
//   * accumulators: 2 x 2 blocks of 32 x 32 per wave, operands swapped (D' = W_tile A_tile^T) so that a lane owns, for its output
//     row m = l & 31, the columns 8 q + 4 (l >> 5) + e (register 4 q + e): four runs of 4 consecutive columns, staged through the
//     wave's private LDS region with 16-byte writes and read back row-major exactly as gemm_epilogue_store does.
//   * epilogues: bf16 output with optional bias, GELU, saved GELU' (bf16 or byte code).  Everything else (fp32 output, residual,
//     MULAUX / DGELU / column sums, fp8 copies, split-K) -> -2 (the caller keeps the other kernels).
#include "gemm_reg.hpp"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16_t;

__device__ __forceinline__ int m32_slot(int row, int c) { return (c ^ ((row >> 1) & 7)) << 4; }

// registers -> LDS (thread t: chunk t & 7 of rows (t >> 3) + 32 i), the image described above
__device__ __forceinline__ void m32_store_tile(unsigned char* tile, const u32x4 (&v)[4]) {
    const int t = threadIdx.x, c = t & 7, r = t >> 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = r + 32 * i;
        *reinterpret_cast<u32x4*>(tile + row * 128 + m32_slot(row, c)) = v[i];
    }
}

// LDS -> 32x32x16 fragment: row rc0 + (l & 31), k = 16 s + 8 (l >> 5) + j, j = 0..7
__device__ __forceinline__ bf16x8 m32_read_frag(const unsigned char* tile, int rc0, int s) {
    const int l = threadIdx.x & 63, row = rc0 + (l & 31);
    return *reinterpret_cast<const bf16x8*>(tile + row * 128 + m32_slot(row, 2 * s + (l >> 5)));
}

// The staged bf16 epilogue of gemm_common.hpp (gemm_epilogue_store, RP = 32) for the 32 x 32 accumulator layout: passes of 32
// rows through the wave's 32 x 68-float region; the read side (8 lanes per 128-byte row segment) is the same.
__device__ __forceinline__ void m32_epilogue(const GemmParams& p, const f32x16_t (&acc)[2][2], float* st, int m_base, int n_base) {
    const int l = threadIdx.x & 63, mm = l & 31, h = l >> 5;
#pragma unroll
    for (int im = 0; im < 2; ++im) {
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<f32x4*>(st + mm * 68 + 32 * jn + 8 * q + 4 * h) =
                    (f32x4){acc[jn][im][4 * q], acc[jn][im][4 * q + 1], acc[jn][im][4 * q + 2], acc[jn][im][4 * q + 3]};
        const int c = (l & 7) * 8, n = n_base + c;
        f32x4 b_lo = {0, 0, 0, 0}, b_hi = {0, 0, 0, 0};
        if ((p.flags & MH_GEMM_BIAS) && n < p.N) {
            b_lo = *reinterpret_cast<const f32x4*>(p.bias + n);
            b_hi = *reinterpret_cast<const f32x4*>(p.bias + n + 4);
        }
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int r = pass * 8 + (l >> 3), m = m_base + 32 * im + r;
            f32x4 lo = *reinterpret_cast<const f32x4*>(st + r * 68 + c);
            f32x4 hi = *reinterpret_cast<const f32x4*>(st + r * 68 + c + 4);
            if (m < p.M && n < p.N) {
                lo += b_lo; hi += b_hi;
                if (p.flags & MH_GEMM_GELU) {
                    f32x4 c_lo, d_lo, c_hi, d_hi;
                    gelu_cdf_pdf4(lo, c_lo, d_lo);
                    gelu_cdf_pdf4(hi, c_hi, d_hi);
                    if (p.aux_out) {
                        f32x4 a_lo = lo, a_hi = hi;
                        if (p.flags & MH_GEMM_AUX_DGELU) { a_lo = lo * d_lo + c_lo; a_hi = hi * d_hi + c_hi; }
                        if (p.flags & MH_GEMM_AUX_U8) {
                            *reinterpret_cast<u32x2*>(reinterpret_cast<uint8_t*>(p.aux_out) + (size_t)m * p.ldaux + n) =
                                pack_dgelu_u8x8(a_lo, a_hi);
                        } else {
                            u32x4 pk = {pack_bf2(a_lo[0], a_lo[1]), pack_bf2(a_lo[2], a_lo[3]), pack_bf2(a_hi[0], a_hi[1]),
                                        pack_bf2(a_hi[2], a_hi[3])};
                            *reinterpret_cast<u32x4*>(p.aux_out + (size_t)m * p.ldaux + n) = pk;
                        }
                    }
                    lo *= c_lo; hi *= c_hi;
                }
                u32x4 pk = {pack_bf2(lo[0], lo[1]), pack_bf2(lo[2], lo[3]), pack_bf2(hi[0], hi[1]), pack_bf2(hi[2], hi[3])};
                *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(p.C) + (size_t)m * p.ldc + n) = pk;
            }
        }
    }
}

__global__ __launch_bounds__(NT, 2) void gemm_m32_kernel(GemmParams p) {
    constexpr int A_BYTES = TILE_BYTES, B_BYTES = TILE_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * A_BYTES + 2 * B_BYTES];  // A0 A1 B0 B1
    unsigned char* const sb = smem + 2 * A_BYTES;
    const int nwg = p.tiles_m * p.tiles_n;
    const int id = xcd_remap(blockIdx.x, nwg);          // the raster of gemm_kernel: XCD runs over 8-tile-high groups
    constexpr int GROUP_M = 8;
    const int per_group = GROUP_M * p.tiles_n;
    const int group = id / per_group, in_group = id - group * per_group;
    const int first_m = group * GROUP_M;
    const int gsz = min(p.tiles_m - first_m, GROUP_M);
    const int tile_m = first_m + in_group % gsz, tile_n = in_group / gsz;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int nk = p.K / BK;                             // host: K % 64 == 0, K >= 128

    const int w = threadIdx.x >> 6;
    const int wm = (w >> 1) * 64, wn = (w & 1) * 64;

    f32x16_t acc[2][2];   // [jn (32 columns)][im (32 rows)]
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][i][e] = 0.f;

    u32x4 ra[4], rb[4];
    // one half of a K step = two 16-wide k slices: 8 fragment reads, 8 MFMAs (32 cycles each)
    auto read_half = [&](const unsigned char* ta, const unsigned char* tb, int half, bf16x8 (&fa)[2][2], bf16x8 (&fb)[2][2]) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[s][i] = m32_read_frag(ta, wm + 32 * i, 2 * half + s);
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[s][j] = m32_read_frag(tb, wn + 32 * j, 2 * half + s);
        }
    };
    auto mfma_half = [&](const bf16x8 (&fa)[2][2], const bf16x8 (&fb)[2][2]) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[s][j], fa[s][i], acc[j][i], 0, 0, 0);
    };

    const __amdgpu_buffer_rsrc_t ra_src = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, (short)0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb_src = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, (short)0, (int)p.b_bytes, 0x00020000);
    int va[4], vb[4];
    tile_offsets<false>(p.lda, m0, va);
    tile_offsets<false>(p.ldb, n0, vb);
    constexpr int step = BK * 2;   // bytes per K step (both operands K-minor)
    int a_off = 0, b_off = 0;
    load_tile_fast(ra_src, va, a_off, ra);
    load_tile_fast(rb_src, vb, b_off, rb);
    m32_store_tile(smem, ra);
    m32_store_tile(sb, rb);
    a_off += step; b_off += step;
    load_tile_fast(ra_src, va, a_off, ra);
    load_tile_fast(rb_src, vb, b_off, rb);
    __syncthreads();
    // steady state, gemm_kernel's schedule with this tile's counts: [8 fragment reads] [8 x (1 MFMA, 1 LDS write, 1 buffer load)]
    // [8 fragment reads] [8 MFMAs], one barrier per K step
    int kt0 = 0;
    for (; kt0 + 2 < nk; ++kt0) {
        const int cur = kt0 & 1;
        const unsigned char* ta = smem + cur * A_BYTES;
        const unsigned char* tb = sb + cur * B_BYTES;
        bf16x8 fa[2][2], fb[2][2];
        read_half(ta, tb, 0, fa, fb);
        mfma_half(fa, fb);
        m32_store_tile(smem + (cur ^ 1) * A_BYTES, ra);
        m32_store_tile(sb + (cur ^ 1) * B_BYTES, rb);
        a_off += step; b_off += step;
        load_tile_fast(ra_src, va, a_off, ra);
        load_tile_fast(rb_src, vb, b_off, rb);
        read_half(ta, tb, 1, fa, fb);
        mfma_half(fa, fb);
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
        __syncthreads();
    }
    for (int kt = kt0; kt < nk; ++kt) {   // the last two steps: nothing left to load
        const int cur = kt & 1;
        const unsigned char* ta = smem + cur * A_BYTES;
        const unsigned char* tb = sb + cur * B_BYTES;
        bf16x8 fa[2][2], fb[2][2];
        read_half(ta, tb, 0, fa, fb);
        mfma_half(fa, fb);
        if (kt + 1 < nk) {
            m32_store_tile(smem + (cur ^ 1) * A_BYTES, ra);
            m32_store_tile(sb + (cur ^ 1) * B_BYTES, rb);
        }
        read_half(ta, tb, 1, fa, fb);
        mfma_half(fa, fb);
        __syncthreads();
    }
    float* st = reinterpret_cast<float*>(smem) + w * (32 * 68);
    m32_epilogue(p, acc, st, m0 + wm, n0 + wn);
}

}  // namespace

// -2 = not eligible (nothing launched): the caller keeps another tile
int gemm_m32_dispatch(int layout, int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C, int ldc, int flags,
                      const float* bias, void* aux_out, int ldaux, void* stream) {
    const int allowed = MH_GEMM_BIAS | MH_GEMM_GELU | MH_GEMM_AUX_DGELU | MH_GEMM_AUX_U8;
    if (layout != 0 || (flags & ~allowed) || K % BK != 0 || K < 2 * BK || N % 8 != 0 || ldc % 8 != 0) return -2;
    GemmParams p;
    p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = C;
    p.bias = bias; p.res = nullptr; p.aux_in = nullptr; p.aux_out = (bf16_t*)aux_out; p.colsum = nullptr;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldr = 0; p.ldaux = ldaux; p.flags = flags;
    p.tiles_m = ceil_div(M, BM); p.tiles_n = ceil_div(N, BN);
    p.k_per_split = K;
    const long a_ext = ((long)(M - 1) * lda + K) * 2, b_ext = ((long)(N - 1) * ldb + K) * 2;
    const long a_reach = (long)(p.tiles_m * BM) * lda * 2, b_reach = (long)(p.tiles_n * BN) * ldb * 2;
    if (a_reach + 4096 >= (1L << 31) || b_reach + 4096 >= (1L << 31)) return -2;
    p.fast = 1;
    p.a_bytes = (unsigned)a_ext; p.b_bytes = (unsigned)b_ext;
    hipLaunchKernelGGL(gemm_m32_kernel, dim3(p.tiles_m * p.tiles_n), dim3(NT), 0, (hipStream_t)stream, p);
    MH_LAUNCH_CHECK();
    return 0;
}
