// fp8 MFMA GEMM for gfx950 (BASELINE configs[4]: "ViT-Base MAE fp8 MFMA path"): C = descale_a * descale_b * A8 B8^T with
// the fused epilogues of mh_gemm_bf16, operands OCP e4m3 (weights, activations) or e5m2 (gradients), fp32 accumulation.
//
//   MFMA  v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales (E8M0 127): the block-scaled form is the one that runs at
//         the fp8 rate on CDNA4 (K = 128 per instruction, 2x the bf16 FLOP per clock; the plain fp8 16x16x32 form runs at the
//         bf16 rate).  Per-TENSOR scaling: the quantisers (quant.hip) multiply by a power-of-two scale kept on the device, the
//         epilogue multiplies the accumulators by the two descale factors.
//   tile  256 x 256 x 128 per 512-thread workgroup (8 waves, 2 x 4, 128 x 64 per wave = 8 x 4 MFMA tiles) or 128 x 128 x 128
//         per 256-thread workgroup (small problems, two workgroups per CU), both operands
//         K-minor ("NT": the dgrad uses a transposed fp8 weight shadow instead of a K-major read), so per output FLOP the
//         kernel moves HALF the operand bytes of the bf16 kernels through L2 -> LDS -- the path that bounds those kernels.
//   LDS   2-stage ring of (A + B tile: 64 KiB / 32 KiB), operands by LDS-DMA (buffer_load ... lds, 8 one-KiB pieces per wave and K
//         step), one raw s_barrier per K step; 128-byte rows, 16-byte chunk position p holds source chunk p ^ (row & 7)
//         (swizzle on the per-lane SOURCE offset and again on the fragment reads: conflict-free ds_read_b128).
//   operand map (checked with exact integer data, tests/test_fp8_gpu.py): lane l holds row (l & 15), K block (l >> 4) of 32
//         consecutive bytes; C / D as every 16 x 16 MFMA: column l & 15, rows 4 (l >> 4) + r.
#include "gemm_common.hpp"

namespace {

constexpr int BK8 = 128;

// Tile geometry: WM x WN waves, each a (16 MT) x 64 accumulator block.
//   <2,4,8> 256 x 256, 512 threads, 128 KiB ring, one workgroup per CU: the large decoder / joint problems
//   <2,2,4> 128 x 128, 256 threads,  64 KiB ring, TWO workgroups per CU (one's epilogue overlaps the other's main loop): the
//           per-group encoder problems of C5 (M = 512 .. 4608 token rows: 54 .. 216 tiles of 256 x 256 would leave most of the
//           256 CUs idle).  At one byte per element this tile moves the same operand bytes per FLOP as the bf16 256 x 256 tile.
//   <2,2,4,4> the same 128 x 128 tile with a FOUR-stage ring (128 KiB, one workgroup per CU, three K steps in flight) for long-K
//           launches with at most one tile per CU anyway: measured (scripts/bench_fp8_gemm.py, isolated) 4608 x 768 x 3072:
//           24.2 -> 21.9 us, 4608 x 512 x 3072: 23.2 -> 20.9; but 3-4 % SLOWER below 128 tiles or at K = 768 (512 x 768 x 3072:
//           18.8 -> 19.6 us -- those launches are bound by the exposed wait -> barrier -> fragment-read -> MFMA chain of a
//           single wave per SIMD, 0.8 us per K step, not by the DMA round trip), hence the narrow dispatch rule below.
template <int WM_, int WN_, int MT_, int S8_ = 2>
struct Tile8 {
    static constexpr int WM = WM_, WN = WN_, MT = MT_, S8 = S8_;
    static constexpr int BM = 16 * MT * WM, BN = 64 * WN, NW = WM * WN, NT = 64 * NW;
    static constexpr int A_BYTES = BM * BK8, B_BYTES = BN * BK8, STAGE_BYTES = A_BYTES + B_BYTES, LDS_BYTES = S8 * STAGE_BYTES;
    static constexpr int PA = A_BYTES / 1024 / NW, PB = B_BYTES / 1024 / NW;   // one-KiB DMA pieces per wave and K step
    static constexpr int MIN_WAVES = NT >= 512 ? 2 : 2;                        // waves per SIMD the register budget must allow
    static_assert(PA * NW * 1024 == A_BYTES && PB * NW * 1024 == B_BYTES, "pieces must divide evenly over the waves");
    static_assert(LDS_BYTES >= NW * 32 * 68 * 4, "the ring doubles as epilogue staging");
};
typedef Tile8<2, 4, 8> T8_256;
typedef Tile8<2, 2, 4> T8_128;
typedef Tile8<2, 2, 4, 4> T8_128D;

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((address_space(3))) void lds_void8;

template <int N>
__device__ __forceinline__ void wait_vm8() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// fragment of rows (rc0 + lane & 15): 32 bytes along k = 32 (lane >> 4) + j
__device__ __forceinline__ i32x8 read_frag8(const unsigned char* img, int rc0) {
    const int l = threadIdx.x & 63, row = rc0 + (l & 15), g = l >> 4, sw = row & 7;
    const u32x4 lo = *reinterpret_cast<const u32x4*>(img + row * 128 + (((2 * g) ^ sw) << 4));
    const u32x4 hi = *reinterpret_cast<const u32x4*>(img + row * 128 + (((2 * g + 1) ^ sw) << 4));
    return (i32x8){(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
}

// A_E5M2: the A operand (activations / gradients, the MFMA's second source here) is e5m2 instead of e4m3
// (Body as a __device__ function template, the kernel a thin wrapper: the host pass of hipcc 7.2 does not emit the launch stub
// of a kernel template whose own body holds the LDS-DMA builtin inside a lambda.)
template <class T, bool A_E5M2>
__device__ __forceinline__ void gemm_fp8_body(const GemmParams& p, unsigned char* smem) {
    constexpr int MT = T::MT, NW = T::NW, PA = T::PA, PB = T::PB;
    const int nwg = p.tiles_m * p.tiles_n;
    const int id = xcd_remap(blockIdx.x, nwg);
    int tile_m, tile_n;
    raster_tile<4>(p, id, tile_m, tile_n);
    const int m0 = tile_m * T::BM, n0 = tile_n * T::BN;
    const int nk = p.K / BK8;

    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63;
    const int wm = (w / T::WN) * (16 * MT), wn = (w % T::WN) * 64;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, (short)0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, (short)0, (int)p.b_bytes, 0x00020000);
    int va[PA], vb[PB];   // per-lane source byte offsets of this wave's pieces (piece = 8 rows x 128 B), k = 0
#pragma unroll
    for (int h = 0; h < PA; ++h) {
        const int row = (w + NW * h) * 8 + (l >> 3), pos = l & 7;
        va[h] = (m0 + row) * p.lda + ((pos ^ (row & 7)) << 4);
    }
#pragma unroll
    for (int h = 0; h < PB; ++h) {
        const int row = (w + NW * h) * 8 + (l >> 3), pos = l & 7;
        vb[h] = (n0 + row) * p.ldb + ((pos ^ (row & 7)) << 4);
    }
    constexpr int S8 = T::S8, DEPTH = S8 - 1;     // K steps in flight
    auto issue = [&](int t) {
        unsigned char* slot = smem + (t % S8) * T::STAGE_BYTES;
#pragma unroll
        for (int h = 0; h < PA; ++h)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void8*)(slot + (w + NW * h) * 1024), 16, va[h], t * BK8, 0, 0);
#pragma unroll
        for (int h = 0; h < PB; ++h)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void8*)(slot + T::A_BYTES + (w + NW * h) * 1024), 16, vb[h],
                                                     t * BK8, 0, 0);
    };

    f32x4 acc[4][MT];   // [j (n tile)][i (m tile)]
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
        if (d < nk) issue(d);
    for (int t = 0; t < nk; ++t) {
        // my pieces of step t have landed: at most the later steps' pieces (PA + PB per step and wave) may still be in flight
        if constexpr (DEPTH == 1) {
            wait_vm8<0>();
        } else {
            const int later = min(nk - 1 - t, DEPTH - 1);
            if (later >= 2) wait_vm8<2 * (PA + PB)>();
            else if (later == 1) wait_vm8<PA + PB>();
            else wait_vm8<0>();
        }
        __builtin_amdgcn_s_barrier();     // everybody's have; step t-1 has been read by everybody -> its slot can be refilled
        const unsigned char* ta = smem + (t % S8) * T::STAGE_BYTES;
        const unsigned char* tb = ta + T::A_BYTES;
        i32x8 fb[4], fa[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = read_frag8(tb, wn + 16 * j);
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = read_frag8(ta, wm + 16 * i);
        __builtin_amdgcn_sched_barrier(0);
        if (t + DEPTH < nk) issue(t + DEPTH);     // streams under this and the next steps' MFMAs
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int half = 0; half < MT / 4; ++half) {     // four m-tiles at a time: A fragments are 8 registers each
            if (half > 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) fa[i] = read_frag8(ta, wm + 64 * half + 16 * i);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[j][4 * half + i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(
                        fb[j], fa[i], acc[j][4 * half + i], 0, A_E5M2 ? 1 : 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        }
    }
    __builtin_amdgcn_s_barrier();   // all reads of the ring are done: reuse it as epilogue staging
    float* st = reinterpret_cast<float*>(smem) + w * (32 * 68);
    gemm_epilogue_store<MT>(p, acc, st, m0 + wm, n0 + wn);
}

template <class T, bool A_E5M2>
__global__ __launch_bounds__(T::NT, T::S8 > 2 ? 1 : 2) void gemm_fp8_kernel(GemmParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[T::LDS_BYTES];   // the ONLY LDS object
    gemm_fp8_body<T, A_E5M2>(p, smem);
}

template <class T>
void launch_fp8(GemmParams& p, int a_format, hipStream_t s) {
    p.tiles_m = ceil_div(p.M, T::BM); p.tiles_n = ceil_div(p.N, T::BN);
    dim3 grid(p.tiles_m * p.tiles_n), block(T::NT);
    if (a_format == MH_FP8_E5M2) hipLaunchKernelGGL((gemm_fp8_kernel<T, true>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((gemm_fp8_kernel<T, false>), grid, block, 0, s, p);
}

}  // namespace

extern "C" int mh_gemm_fp8(int M, int N, int K, const void* A8, int lda, int a_format, const void* B8, int ldb, void* C, int ldc,
                           int flags, const float* descale_a, const float* descale_b, const float* bias, const float* res,
                           int ldr, const void* aux_in, void* aux_out, int ldaux, float* colsum, void* c8, int ldc8,
                           const float* c8_scale, float* c8_amax, void* stream) {
    MH_CHECK_ARG(M > 0 && N > 0 && K >= BK8 && K % BK8 == 0, "mh_gemm_fp8: K must be a positive multiple of 128 (%d %d %d)", M, N, K);
    MH_CHECK_ARG(A8 && B8 && C && descale_a && descale_b, "mh_gemm_fp8: null operand / descale pointer");
    MH_CHECK_ARG(a_format == MH_FP8_E4M3 || a_format == MH_FP8_E5M2, "mh_gemm_fp8: a_format %d", a_format);
    MH_CHECK_ARG(lda % 16 == 0 && ldb % 16 == 0 && lda >= K && ldb >= K, "mh_gemm_fp8: lda / ldb must be multiples of 16 and >= K");
    MH_CHECK_ARG(((uintptr_t)A8 | (uintptr_t)B8 | (uintptr_t)C) % 16 == 0, "mh_gemm_fp8: bases must be 16-B aligned");
    MH_CHECK_ARG(N % 8 == 0 && ldc % 8 == 0, "mh_gemm_fp8: N, ldc %% 8 == 0");
    MH_CHECK_ARG(!(flags & MH_GEMM_ATOMIC), "mh_gemm_fp8: no atomic accumulate");
    MH_CHECK_ARG((flags & MH_GEMM_OUT_F32) || !(flags & MH_GEMM_RESIDUAL), "mh_gemm_fp8: residual epilogue needs f32 output");
    MH_CHECK_ARG(!(flags & MH_GEMM_OUT_F32) || !(flags & (MH_GEMM_GELU | MH_GEMM_DGELU | MH_GEMM_MULAUX | MH_GEMM_COLSUM)),
                 "mh_gemm_fp8: GELU / aux / colsum epilogues need bf16 output");
    MH_CHECK_ARG(!(flags & MH_GEMM_BIAS) || bias, "mh_gemm_fp8: bias flag without pointer");
    MH_CHECK_ARG(!(flags & MH_GEMM_RESIDUAL) || (res && ldr % 4 == 0), "mh_gemm_fp8: residual needs pointer, ldr %% 4 == 0");
    MH_CHECK_ARG(!(flags & (MH_GEMM_DGELU | MH_GEMM_MULAUX)) || (aux_in && ldaux % 8 == 0), "mh_gemm_fp8: aux_in / ldaux");
    MH_CHECK_ARG(!(flags & MH_GEMM_AUX_DGELU) || ((flags & MH_GEMM_GELU) && aux_out), "mh_gemm_fp8: aux_dgelu needs GELU + aux_out");
    MH_CHECK_ARG(!(flags & MH_GEMM_GELU) || !aux_out || ldaux % 8 == 0, "mh_gemm_fp8: ldaux %% 8");
    MH_CHECK_ARG(!(flags & MH_GEMM_AUX_U8) || ((flags & (MH_GEMM_AUX_DGELU | MH_GEMM_MULAUX)) && !(flags & MH_GEMM_DGELU)),
                 "mh_gemm_fp8: MH_GEMM_AUX_U8 applies to the saved GELU derivative only (AUX_DGELU / MULAUX)");
    MH_CHECK_ARG(!(flags & MH_GEMM_COLSUM) || colsum, "mh_gemm_fp8: colsum flag without pointer");
    MH_CHECK_ARG(!c8 || (!(flags & MH_GEMM_OUT_F32) && c8_scale && ldc8 % 8 == 0 && (uintptr_t)c8 % 8 == 0),
                 "mh_gemm_fp8: the fp8 output copy needs a bf16-output epilogue, a scale and ldc8 %% 8 == 0");
    GemmParams p;
    p.A = (const bf16_t*)A8; p.B = (const bf16_t*)B8; p.C = C;
    p.bias = bias; p.res = res; p.aux_in = (const bf16_t*)aux_in; p.aux_out = (bf16_t*)aux_out; p.colsum = colsum;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldr = ldr; p.ldaux = ldaux; p.flags = flags;
    p.k_per_split = K; p.fast = 1;
    const long a_ext = (long)(M - 1) * lda + K, b_ext = (long)(N - 1) * ldb + K;   // bytes: rows beyond M / N read as zero
    MH_CHECK_ARG((long)ceil_div(M, 256) * 256 * lda + 65536 < (1L << 31) && (long)ceil_div(N, 256) * 256 * ldb + 65536 < (1L << 31),
                 "mh_gemm_fp8: operand beyond the 2 GiB buffer-descriptor range");
    p.a_bytes = (unsigned)a_ext; p.b_bytes = (unsigned)b_ext;
    p.descale_a = descale_a; p.descale_b = descale_b;
    p.c8 = (uint8_t*)c8; p.c8_scale = c8_scale; p.c8_amax = c8_amax; p.ldc8 = ldc8;
    // 128 x 128 tiles (two workgroups per CU: one's epilogue under the other's main loop, and room for other streams' kernels
    // beside it) unless the problem is long in K and fills the chip with 256 x 256 tiles for several rounds -- measured
    // (scripts/bench_fp8_gemm.py): 128^2 is as fast or faster on every shape of the C5 / C3 steps (e.g. 8192 x 768 x 3072:
    // 1403 vs 855 TFLOP/s; 32768 x 3072 x 512: 1090 vs 1061), 256^2 wins at 16384 x 4096 x 4096 (2140 vs 1796).
    // MH_GEMM_FP8_TILE_* bits in `flags` force one (experiments, tests; the library reads no environment).
    const long tiles256 = (long)ceil_div(M, 256) * ceil_div(N, 256), tiles128 = (long)ceil_div(M, 128) * ceil_div(N, 128);
    const int force = flags & (MH_GEMM_FP8_TILE_256 | MH_GEMM_FP8_TILE_128 | MH_GEMM_FP8_TILE_128D);
    p.flags = flags & ~force;
    const bool big = force ? (force & MH_GEMM_FP8_TILE_256) != 0 : (tiles256 >= 768 && K >= 2048);
    const bool deep = force ? (force & MH_GEMM_FP8_TILE_128D) != 0 : (tiles128 > 128 && tiles128 <= 256 && K >= 2048);   // (256 CUs: at most one tile per CU)
    if (big) launch_fp8<T8_256>(p, a_format, (hipStream_t)stream);
    else if (deep) launch_fp8<T8_128D>(p, a_format, (hipStream_t)stream);
    else launch_fp8<T8_128>(p, a_format, (hipStream_t)stream);
    MH_LAUNCH_CHECK();
    return 0;
}
