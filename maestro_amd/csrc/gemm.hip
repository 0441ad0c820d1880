// bf16 MFMA GEMM for gfx950 with fused epilogues (see include/maestro_hip.h: mh_gemm_bf16).
//
// Tile 128x128x64 per 256-thread workgroup (4 waves, 2x2, 64x64 per wave = 4x4 v_mfma_f32_16x16x32_bf16 tiles).
// Operands are register-staged into a double-buffered, swizzled LDS image (one barrier per K step):
//   * K-minor operand ([rows][k], k contiguous in memory): 128-B LDS rows, 16-B chunk index XOR (row & 7),
//     fragments by ds_read_b128 (conflict-free: 16-lane groups hit 16 distinct 16-B slots).
//   * K-major operand ([k][cols], cols contiguous: dgrad B = W, wgrad A = dY and B = X): 256-B LDS rows,
//     32-B chunk index XOR f(k), fragments by ds_read_b64_tr_b16 (hardware transpose) -- no transposed copies
//     of weights or activations are ever materialised in HBM.
// The MFMA is issued with swapped operands (D' = B_tile * A_tile^T) so every lane owns 4 CONSECUTIVE output
// columns of one row: bias/residual/aux are read and C is written with 8/16-byte vectors.
// Split-K + fp32 atomics (LDS-transposed so each wave instruction adds 256 contiguous bytes) serve the wgrad.
#include "gemm_reg.hpp"

namespace {

// Measured alternatives that lost on MI355X (kept out of the code): a second register set / 2-step-deep prefetch (halves
// occupancy, ~2x slower); one LDS buffer + two barriers per K step at 4 workgroups per CU (NN/TN spill under the
// 128-VGPR cap, NT no faster); select-based branch-free predication (slower than exec-mask branches).
//
// MT = 16-row blocks per wave: the tile is (32 MT) x 128 -- 4: 128 x 128 (two workgroups per CU); 2: 64 x 128 (48 KiB of LDS,
// three workgroups per CU); 6: 192 x 128 (80 KiB, two per CU).  The other two shapes exist for the N = 512 / 768 outputs of the
// transformer blocks (out-proj, fc2 and three of the four dgrads), whose 128 x 128 tilings fill the 512 workgroup slots badly
// (M = 8192: 384 tiles; M = 3200: 150; M = 11392: 534): see the dispatch rule in mh_gemm_bf16_tile.
template <bool A_KMAJOR, bool B_KMAJOR, int MT = 4>
__global__ __launch_bounds__(NT, MT == 2 ? 3 : 2) void gemm_kernel(GemmParams p) {
    static_assert(!A_KMAJOR || MT == 4, "the K-major A image is 128 columns wide");
    constexpr int TBM = 32 * MT, NA = MT;                      // tile rows; 16-byte A pieces per thread and K step
    constexpr int A_BYTES = TBM * 128, B_BYTES = TILE_BYTES;   // one K step of each operand in LDS
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * A_BYTES + 2 * B_BYTES];  // A0 A1 B0 B1
    unsigned char* const sb = smem + 2 * A_BYTES;
    const int nwg = p.tiles_m * p.tiles_n;
    // XCD-aware + grouped rasterisation: every XCD gets a contiguous run of ids, and ids walk GROUP_M m-tiles before
    // moving to the next n-tile, so the ~64 workgroups co-resident on one XCD (32 CUs x 2) cover an 8x8 super-tile:
    // 8 A stripes + 8 B panels (~3 MB at K=768) stay in that XCD's 4 MB L2 instead of a whole B matrix.
    const int id = xcd_remap(blockIdx.x, nwg);
    constexpr int GROUP_M = MT == 2 ? 16 : 8;
    int tile_m, tile_n;
    raster_tile<GROUP_M>(p, id, tile_m, tile_n);
    const int m0 = tile_m * TBM, n0 = tile_n * BN;
    const int kbeg = blockIdx.y * p.k_per_split;
    const int kend = min(p.K, kbeg + p.k_per_split);
    const int nk = (kend - kbeg + BK - 1) / BK;

    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int wm = (w >> 1) * (16 * MT), wn = (w & 1) * 64;

    f32x4 acc[4][MT];  // [j (n tile)][i (m tile)]
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    u32x4 ra[NA], rb[4];
    auto mfma_block = [&](const bf16x8 (&fa)[MT], const bf16x8 (&fb)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < MT; ++i)
                acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[j][i], 0, 0, 0);
    };
    auto compute = [&](const unsigned char* ta, const unsigned char* tb) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 fa[MT], fb[4];
#pragma unroll
            for (int i = 0; i < MT; ++i) fa[i] = read_frag<A_KMAJOR>(ta, wm + 16 * i, s);
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = read_frag<B_KMAJOR>(tb, wn + 16 * j, s);
            mfma_block(fa, fb);
        }
    };
    auto compute_half = [&](const unsigned char* ta, const unsigned char* tb, int s) {
        bf16x8 fa[MT], fb[4];
#pragma unroll
        for (int i = 0; i < MT; ++i) fa[i] = read_frag<A_KMAJOR>(ta, wm + 16 * i, s);
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = read_frag<B_KMAJOR>(tb, wn + 16 * j, s);
        __builtin_amdgcn_s_setprio(1);   // favour the wave that has its fragments: +0.5-1.5 % (it issues its MFMAs back to back)
        mfma_block(fa, fb);
        __builtin_amdgcn_s_setprio(0);
    };
    if (p.fast) {
        const __amdgpu_buffer_rsrc_t ra_src = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, (short)0, (int)p.a_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rb_src = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, (short)0, (int)p.b_bytes, 0x00020000);
        int va[NA], vb[4];
        tile_offsets<A_KMAJOR, NA>(p.lda, m0, va);
        tile_offsets<B_KMAJOR>(p.ldb, n0, vb);
        const int a_step = A_KMAJOR ? BK * p.lda * 2 : BK * 2;   // bytes per K step
        const int b_step = B_KMAJOR ? BK * p.ldb * 2 : BK * 2;
        int a_off = A_KMAJOR ? kbeg * p.lda * 2 : kbeg * 2;
        int b_off = B_KMAJOR ? kbeg * p.ldb * 2 : kbeg * 2;
        // Software pipeline (one register set, two LDS buffers): tile kt+1 is written to LDS in the MIDDLE of step kt
        // (between the two MFMA halves) and the loads of tile kt+2 are re-issued right behind it, so every load has
        // a full step of latency cover and the barrier at the end of the step is not preceded by a vmcnt wait.
        load_tile_fast(ra_src, va, a_off, ra);
        load_tile_fast(rb_src, vb, b_off, rb);
        store_tile<A_KMAJOR, NA>(smem, ra);
        store_tile<B_KMAJOR>(sb, rb);
        if (nk > 1) {
            a_off += a_step; b_off += b_step;
            load_tile_fast(ra_src, va, a_off, ra);
            load_tile_fast(rb_src, vb, b_off, rb);
        }
        __syncthreads();
        // Steady state (branch-free body, ONE scheduling region): the LDS stores of tile kt+1 and the global loads of tile
        // kt+2 are spread over the first half's MFMAs (one store + one load per group of MFMAs, pinned with
        // sched_group_barrier) instead of sitting in a block between the halves, where the VGPR -> LDS store path (~13
        // cycles per ds_write_b128, shared by the workgroups of the CU) stalled the wave's MFMA stream: +10..35 % on
        // every shape and layout (e.g. M = 8192, N = 768, K = 3072: 653 -> 772 TFLOP/s), bit-identical results.
        constexpr int NM = 4 * MT, NS = NA + 4;                      // MFMAs per half; LDS stores = global loads per step
        constexpr int NR = (A_KMAJOR ? 8 : MT) + (B_KMAJOR ? 8 : 4); // fragment reads per half (ds_read_b128 / pairs of tr reads)
        int kt0 = 0;
        for (; kt0 + 2 < nk; ++kt0) {
            const int cur = kt0 & 1;
            const unsigned char* ta = smem + cur * A_BYTES;
            const unsigned char* tb = sb + cur * B_BYTES;
            bf16x8 fa[MT], fb[4];
#pragma unroll
            for (int i = 0; i < MT; ++i) fa[i] = read_frag<A_KMAJOR>(ta, wm + 16 * i, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = read_frag<B_KMAJOR>(tb, wn + 16 * j, 0);
            mfma_block(fa, fb);
            store_tile<A_KMAJOR, NA>(smem + (cur ^ 1) * A_BYTES, ra);
            store_tile<B_KMAJOR>(sb + (cur ^ 1) * B_BYTES, rb);
            a_off += a_step; b_off += b_step;
            load_tile_fast(ra_src, va, a_off, ra);
            load_tile_fast(rb_src, vb, b_off, rb);
#pragma unroll
            for (int i = 0; i < MT; ++i) fa[i] = read_frag<A_KMAJOR>(ta, wm + 16 * i, 1);
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = read_frag<B_KMAJOR>(tb, wn + 16 * j, 1);
            mfma_block(fa, fb);
            // schedule: [first-half fragment reads] [NS x (MFMAs, 1 DS write, 1 VMEM read)] [second-half reads] [NM MFMA]
            __builtin_amdgcn_sched_group_barrier(0x100, NR, 0);
#pragma unroll
            for (int r = 0; r < NS; ++r) {
                __builtin_amdgcn_sched_group_barrier(0x008, NM / NS, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
            if constexpr (NM % NS != 0) __builtin_amdgcn_sched_group_barrier(0x008, NM - (NM / NS) * NS, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, NR, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
            __syncthreads();
        }
        for (int kt = kt0; kt < nk; ++kt) {   // the last two steps: nothing left to load / store
            const int cur = kt & 1;
            const unsigned char* ta = smem + cur * A_BYTES;
            const unsigned char* tb = sb + cur * B_BYTES;
            compute_half(ta, tb, 0);
            if (kt + 1 < nk) {
                store_tile<A_KMAJOR, NA>(smem + (cur ^ 1) * A_BYTES, ra);
                store_tile<B_KMAJOR>(sb + (cur ^ 1) * B_BYTES, rb);
                if (kt + 2 < nk) {
                    a_off += a_step; b_off += b_step;
                    load_tile_fast(ra_src, va, a_off, ra);
                    load_tile_fast(rb_src, vb, b_off, rb);
                }
            }
            compute_half(ta, tb, 1);
            __syncthreads();
        }
    } else {
        // general path (K tail inside a K-minor operand; tiny GEMMs only): predicated loads, one LDS buffer, two barriers per step
        unsigned char* ta = smem;
        unsigned char* tb = sb;
        load_tile<A_KMAJOR, NA>(p.A, p.lda, m0, p.M, kbeg, kend, ra);
        load_tile<B_KMAJOR>(p.B, p.ldb, n0, p.N, kbeg, kend, rb);
        store_tile<A_KMAJOR, NA>(ta, ra);
        store_tile<B_KMAJOR>(tb, rb);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const bool more = kt + 1 < nk;
            if (more) {
                load_tile<A_KMAJOR, NA>(p.A, p.lda, m0, p.M, kbeg + (kt + 1) * BK, kend, ra);
                load_tile<B_KMAJOR>(p.B, p.ldb, n0, p.N, kbeg + (kt + 1) * BK, kend, rb);
            }
            compute(ta, tb);
            __syncthreads();
            if (more) {
                store_tile<A_KMAJOR, NA>(ta, ra);
                store_tile<B_KMAJOR>(tb, rb);
            }
            __syncthreads();
        }
    }

    // epilogues (gemm_common.hpp): operand buffers are free now (the last loop barrier has been passed); each wave stages
    // through its private 32 x 68 float region (4 x 8704 B = 34 KiB <= the 48 KiB of the smallest tile)
    float* st = reinterpret_cast<float*>(smem) + w * (32 * 68);
    if (p.flags & MH_GEMM_ATOMIC) gemm_epilogue_atomic<MT>(p, acc, st, m0 + wm, n0 + wn);
    else gemm_epilogue_store<MT>(p, acc, st, m0 + wm, n0 + wn);
}

}  // namespace

// Dispatch rule for the large-tile LDS-DMA kernel (gemm_dma.hip), from scripts/bench_tiles.py on MI355X: it wins (x1.05 to
// x1.2, x1.8 on the K = 512 dgrad) for NT and NN problems whose 256x256 tiles fill the 256 CUs in whole waves (>= 90 % wave
// efficiency: the M = 32768 decoder shapes); it loses when the tile count quantises badly (M = 8192 encoder shapes) and for
// split-K wgrads (more splits -> more fp32 atomic passes).  The library reads no environment: a caller that wants another
// kernel passes an explicit tile to mh_gemm_bf16_tile (maestro_amd/hip.py maps MH_GEMM_DMA / MH_GEMM_TILE onto that).
static bool prefer_dma(int layout, int M, int N, int K, int flags) {
    if (layout == 2 || (flags & MH_GEMM_ATOMIC) || K % 32 != 0 || K < 256) return false;
    const long tiles = (long)ceil_div(M, 256) * ceil_div(N, 256);
    if (tiles < 256) return false;
    const long waves = (tiles + 255) / 256;
    return 10 * tiles >= 9 * 256 * waves;
}

static int gemm_dispatch(int tile, int layout, int M, int N, int K, const void* A, int lda, const void* B, int ldb,
                         void* C, int ldc, int flags, const float* bias, const float* res, int ldr,
                         const void* aux_in, void* aux_out, int ldaux, float* colsum, void* stream);

extern "C" int mh_gemm_bf16_tile(int tile, int layout, int M, int N, int K, const void* A, int lda, const void* B, int ldb,
                                 void* C, int ldc, int flags, const float* bias, const float* res, int ldr,
                                 const void* aux_in, void* aux_out, int ldaux, float* colsum, void* stream) {
    return gemm_dispatch(tile, layout, M, N, K, A, lda, B, ldb, C, ldc, flags, bias, res, ldr, aux_in, aux_out, ldaux, colsum, stream);
}

static int gemm_dispatch(int tile, int layout, int M, int N, int K, const void* A, int lda, const void* B, int ldb,
                         void* C, int ldc, int flags, const float* bias, const float* res, int ldr,
                         const void* aux_in, void* aux_out, int ldaux, float* colsum, void* stream) {
    MH_CHECK_ARG(tile >= MH_TILE_AUTO && tile <= MH_TILE_REG_192, "mh_gemm_bf16: tile %d", tile);
    MH_CHECK_ARG(layout >= 0 && layout <= 2, "mh_gemm_bf16: layout %d", layout);
    MH_CHECK_ARG(!(flags & ~0x7ff), "mh_gemm_bf16: unknown flag bits 0x%x", flags & ~0x7ff);
    MH_CHECK_ARG(M > 0 && N > 0 && K > 0, "mh_gemm_bf16: empty problem %d %d %d", M, N, K);
    MH_CHECK_ARG(A && B && C, "mh_gemm_bf16: null operand");
    MH_CHECK_ARG(lda % 8 == 0 && ldb % 8 == 0, "mh_gemm_bf16: lda/ldb must be multiples of 8 (%d, %d)", lda, ldb);
    MH_CHECK_ARG(N % 4 == 0 && ldc % 4 == 0, "mh_gemm_bf16: N and ldc must be multiples of 4 (%d, %d)", N, ldc);
    MH_CHECK_ARG((flags & MH_GEMM_OUT_F32) || (N % 8 == 0 && ldc % 8 == 0), "mh_gemm_bf16: bf16 output needs N, ldc %% 8 == 0");
    MH_CHECK_ARG((flags & MH_GEMM_OUT_F32) || !(flags & MH_GEMM_RESIDUAL), "mh_gemm_bf16: residual epilogue needs f32 output");
    MH_CHECK_ARG(!(flags & MH_GEMM_OUT_F32) || !(flags & (MH_GEMM_GELU | MH_GEMM_DGELU | MH_GEMM_MULAUX)), "mh_gemm_bf16: GELU / aux epilogues need bf16 output");
    MH_CHECK_ARG(((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) % 16 == 0, "mh_gemm_bf16: bases must be 16-B aligned");
    if (layout == 2) {
        MH_CHECK_ARG(M % 8 == 0 && N % 8 == 0, "mh_gemm_bf16: TN needs M, N multiples of 8 (%d, %d)", M, N);
    } else {
        MH_CHECK_ARG(K % 8 == 0, "mh_gemm_bf16: K must be a multiple of 8 (%d)", K);
        if (layout == 1) MH_CHECK_ARG(N % 8 == 0, "mh_gemm_bf16: NN needs N multiple of 8 (%d)", N);
    }
    MH_CHECK_ARG(!(flags & MH_GEMM_BIAS) || bias, "mh_gemm_bf16: bias flag without pointer");
    MH_CHECK_ARG(!(flags & MH_GEMM_RESIDUAL) || (res && ldr % 4 == 0), "mh_gemm_bf16: residual needs pointer, ldr%%4==0");
    MH_CHECK_ARG(!(flags & (MH_GEMM_DGELU | MH_GEMM_MULAUX)) || (aux_in && ldaux % 8 == 0), "mh_gemm_bf16: dgelu / mulaux need aux_in, ldaux %% 8 == 0");
    MH_CHECK_ARG(!(flags & MH_GEMM_AUX_DGELU) || ((flags & MH_GEMM_GELU) && aux_out), "mh_gemm_bf16: aux_dgelu needs the GELU epilogue and aux_out");
    MH_CHECK_ARG(!((flags & MH_GEMM_DGELU) && (flags & MH_GEMM_MULAUX)), "mh_gemm_bf16: dgelu and mulaux exclude each other");
    MH_CHECK_ARG(!(flags & MH_GEMM_AUX_U8) || ((flags & (MH_GEMM_AUX_DGELU | MH_GEMM_MULAUX)) && !(flags & MH_GEMM_DGELU)),
                 "mh_gemm_bf16: MH_GEMM_AUX_U8 applies to the saved GELU derivative only (AUX_DGELU / MULAUX)");
    MH_CHECK_ARG(!(flags & MH_GEMM_GELU) || !aux_out || ldaux % 8 == 0, "mh_gemm_bf16: ldaux %% 8");
    MH_CHECK_ARG(!(flags & MH_GEMM_ATOMIC) || (flags & MH_GEMM_OUT_F32), "mh_gemm_bf16: atomic needs f32 output");
    MH_CHECK_ARG(!(flags & MH_GEMM_COLSUM) || (colsum && !(flags & MH_GEMM_OUT_F32)), "mh_gemm_bf16: colsum needs a pointer and bf16 output");
    MH_CHECK_ARG(!(flags & MH_GEMM_ATOMIC) || !(flags & ~(MH_GEMM_ATOMIC | MH_GEMM_OUT_F32)),
                 "mh_gemm_bf16: atomic accumulate excludes other epilogues");

    if (tile >= MH_TILE_PP_128 && tile <= MH_TILE_PP_128_DIAG5)
        return gemm_pp_dispatch(layout, M, N, K, A, lda, B, ldb, C, ldc, flags, bias, res, ldr, aux_in, aux_out, ldaux, colsum, stream,
                                tile - MH_TILE_PP_128);
    if (tile > MH_TILE_REG_128 && tile < MH_TILE_PP_128)   // explicit DMA tile: -2 when not eligible (the caller picks another tile)
        return gemm_dma_dispatch(tile, layout, M, N, K, A, lda, B, ldb, C, ldc, flags, bias, res, ldr, aux_in, aux_out, ldaux, colsum, stream);
    // Persistent 128x128 tile with the epilogue inside the next tile's main loop (gemm_pp.hip), scripts/bench_pp.py on the C3
    // step's shapes with their real epilogues: +6...12 % on the plain bf16 NT outputs (qkv), +3...10 % on fp32 + residual
    // (out-proj, fc2) and on the plain NN dgrads, +3 % on fc1; it LOSES 8-20 % on the fc2 dgrad (MULAUX + column sums: 256
    // VGPRs, spills) -> never picked there.  Against the 256x256 LDS-DMA tile (M = 32768) it wins the short-K NT problems
    // (qkv 61.0 vs 68.4 us, out-proj 38.8 vs 43.3, fc1 165 vs 173) and loses the long-K ones (fc2 136 vs 125).
    const bool dma = tile == MH_TILE_AUTO && prefer_dma(layout, M, N, K, flags);
    // 192 x 128 tiles where the 128 x 128 tiling overshoots the 512 workgroup slots by a few tiles (M = 11392, N = 768: 534 tiles,
    // i.e. a second, almost empty round; 360 tiles of 192 x 128 run in one): fc2 78.8 -> 76.5 us, fc1 dgrad 73.6 -> 69.5, qkv dgrad
    // 56.8 -> 54.1 (scripts/bench_pp.py).  The 64 x 128 form (three workgroups per CU) gains 1-4 % at M = 8192, N = 768 and loses
    // elsewhere: explicit tile only.
    if (tile == MH_TILE_AUTO && !dma && layout != 2 && !(flags & (MH_GEMM_COLSUM | MH_GEMM_ATOMIC)) && K >= 1536 && K % BK == 0) {
        const long t128 = (long)ceil_div(M, 128) * ceil_div(N, 128);
        if (t128 > 512 && t128 <= 576 && (long)ceil_div(M, 192) * ceil_div(N, 128) <= 512) tile = MH_TILE_REG_192;
    }
    // Round 3, second pass -- the rule below follows the kernels' times INSIDE the step (bench.py --shapes on the pretrain, probe and
    // finetune steps, profiles/README.md), which differ from the isolated loops above (operands are cold, the fp32 residual is
    // read, two group streams share the chip): the ping-pong tile keeps the NT problems between one round of workgroups and
    // 8192 tiles (qkv +6 %, out-proj +11 %, fc2 +4 % at M = 32768; +2 ... 5 % at M = 8192), loses the fc1 (GELU) epilogue beyond
    // ~2300 tiles (M = 32768: -7 % against the LDS-DMA tile, M = 12800: -2 %), loses 27 % on the segmentation head (M = 557056,
    // 52224 tiles, 1.7 GB of output: the LDS-DMA tile's staged full-line stores), and is a tie or a loss on every NN (dgrad)
    // problem in the step (M = 3200: -9 %), so those stay with the one-tile-per-workgroup / LDS-DMA kernels.
    const long t128_all = (long)ceil_div(M, 128) * ceil_div(N, 128);
    const bool pp_ok = layout == 0 && t128_all >= 256 &&
                       ((flags & MH_GEMM_GELU) ? t128_all <= 2304 : (!dma || (K < 1024 && t128_all <= 8192)));
    if (tile == MH_TILE_AUTO && !(flags & MH_GEMM_MULAUX) && pp_ok) {
        const int rc = gemm_pp_dispatch(layout, M, N, K, A, lda, B, ldb, C, ldc, flags, bias, res, ldr, aux_in, aux_out, ldaux, colsum, stream);
        if (rc != -2) return rc;   // -2: not eligible -> the kernels below
    }
    if (dma) {
        const int rc = gemm_dma_dispatch(MH_TILE_DMA_256, layout, M, N, K, A, lda, B, ldb, C, ldc, flags, bias, res, ldr, aux_in,
                                         aux_out, ldaux, colsum, stream);
        if (rc != -2) return rc;   // -2: not eligible -> general kernel below
    }
    GemmParams p;
    p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = C;
    p.bias = bias; p.res = res; p.aux_in = (const bf16_t*)aux_in; p.aux_out = (bf16_t*)aux_out; p.colsum = colsum;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldr = ldr; p.ldaux = ldaux; p.flags = flags;
    // tile rows: 128 unless the caller (or the rule above) asked for the 64- / 192-row form (K-minor A, no column sums)
    const int mt = (layout != 2 && !(flags & MH_GEMM_COLSUM)) ? (tile == MH_TILE_REG_64 ? 2 : tile == MH_TILE_REG_192 ? 6 : 4) : 4;
    const int tbm = 32 * mt;
    p.tiles_m = ceil_div(M, tbm); p.tiles_n = ceil_div(N, BN);
    int splits = 1;
    if (flags & MH_GEMM_ATOMIC) {  // fill the 256 CUs (2 workgroups each) when the output has few tiles
        const int tiles = p.tiles_m * p.tiles_n;
        const int ksteps = ceil_div(K, BK);
        splits = max(1, min(min(512 / max(tiles, 1), ksteps / 4), 32));
    }
    const int ksteps_per = ceil_div(ceil_div(K, BK), splits);
    p.k_per_split = ksteps_per * BK;
    splits = ceil_div(K, p.k_per_split);
    // Buffer-load fast path: K-minor operands must have no K tail inside a K step (it would wrap into the next row
    // instead of reading zero); K-major operands zero-fill beyond K by the descriptor extent.  Split ranges are whole
    // K steps, so the only tail is the global one.
    const bool a_kmajor = layout == 2, b_kmajor = layout != 0;
    const long a_ext = a_kmajor ? ((long)(K - 1) * lda + M) * 2 : ((long)(M - 1) * lda + K) * 2;
    const long b_ext = b_kmajor ? ((long)(K - 1) * ldb + N) * 2 : ((long)(N - 1) * ldb + K) * 2;
    const long a_reach = a_kmajor ? (long)(ceil_div(K, BK) * BK) * lda * 2 : (long)(p.tiles_m * tbm) * lda * 2;
    const long b_reach = b_kmajor ? (long)(ceil_div(K, BK) * BK) * ldb * 2 : (long)(p.tiles_n * BN) * ldb * 2;
    const bool tail_ok = (a_kmajor || K % BK == 0) && (b_kmajor || K % BK == 0);
    p.fast = tail_ok && a_reach + 4096 < (1L << 31) && b_reach + 4096 < (1L << 31);
    p.a_bytes = (unsigned)a_ext; p.b_bytes = (unsigned)b_ext;
    dim3 grid(p.tiles_m * p.tiles_n, splits), block(NT);
    hipStream_t s = (hipStream_t)stream;
    switch (layout * 8 + mt) {
        case 0 * 8 + 4: hipLaunchKernelGGL((gemm_kernel<false, false, 4>), grid, block, 0, s, p); break;
        case 0 * 8 + 2: hipLaunchKernelGGL((gemm_kernel<false, false, 2>), grid, block, 0, s, p); break;
        case 0 * 8 + 6: hipLaunchKernelGGL((gemm_kernel<false, false, 6>), grid, block, 0, s, p); break;
        case 1 * 8 + 4: hipLaunchKernelGGL((gemm_kernel<false, true, 4>), grid, block, 0, s, p); break;
        case 1 * 8 + 2: hipLaunchKernelGGL((gemm_kernel<false, true, 2>), grid, block, 0, s, p); break;
        case 1 * 8 + 6: hipLaunchKernelGGL((gemm_kernel<false, true, 6>), grid, block, 0, s, p); break;
        default: hipLaunchKernelGGL((gemm_kernel<true, true, 4>), grid, block, 0, s, p); break;
    }
    MH_LAUNCH_CHECK();
    return 0;
}

extern "C" int mh_gemm_bf16(int layout, int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C,
                            int ldc, int flags, const float* bias, const float* res, int ldr, const void* aux_in,
                            void* aux_out, int ldaux, float* colsum, void* stream) {
    return mh_gemm_bf16_tile(MH_TILE_AUTO, layout, M, N, K, A, lda, B, ldb, C, ldc, flags, bias, res, ldr, aux_in, aux_out,
                             ldaux, colsum, stream);
}
