// Large-tile bf16 MFMA GEMM for gfx950 fed by LDS-DMA (buffer_load ... lds), same interface/epilogues as gemm.hip.
//
//   tile 256 x 256 x 32 per 512-thread workgroup (8 waves, 2(M) x 4(N), 128 x 64 per wave = 8 x 4 MFMA 16x16x32 tiles)
//   LDS  4-stage ring of (A 16 KiB + B 16 KiB) = 128 KiB; operands go HBM/L2 -> LDS directly (no VGPR staging, no
//        ds_write): each wave issues 4 one-KiB DMA pieces per K step, 3 K steps ahead of their use
//   sync one raw s_barrier per K step; every wave waits for ITS OWN pieces of step t with a counted s_waitcnt vmcnt(8|4|0)
//        (the two younger steps stay in flight across the barrier), then the barrier makes everybody's pieces of step t
//        visible and certifies that step t-1 has been fully read, so its ring slot is refilled (step t+3) right after it
//   LDS reads per MFMA: 12 ds_read per 32 MFMA (0.375 vs 0.5 for the 128^2 kernel); HBM/L2 bytes per FLOP halved.
//
// LDS images are lane-linear per DMA piece (hardware writes M0 + 16*lane), so the bank-conflict swizzles are applied on
// the per-lane SOURCE offset and again on the fragment reads (same involution):
//   K-minor operand tile [256 rows][32 k]: 64-B rows, 16-B chunk position p holds source chunk p ^ f((row>>2)&3)
//   K-major operand tile [32 k][256 cols]: 512-B rows, 32-B chunk position q holds source chunk q ^ ((k&3)|((k>>3&1)<<2)),
//          fragments by ds_read_b64_tr_b16 (hardware transpose)
// Out-of-range rows / K rows read as zero through the buffer descriptor's extent (no predication anywhere).
#include "gemm_ring.hpp"

// MH_DMA_PRIO (A/B aid): 0 = raise the priority around every math phase (default); 1 = static priority for the second-dispatched half
// (waves 4-7), no flips; 2 = no priority at all
#ifndef MH_DMA_PRIO
#define MH_DMA_PRIO 0
#endif
#if MH_DMA_PRIO == 0
#define MH_PRIO_UP() __builtin_amdgcn_s_setprio(1)
#define MH_PRIO_DOWN() __builtin_amdgcn_s_setprio(0)
#else
#define MH_PRIO_UP() ((void)0)
#define MH_PRIO_DOWN() ((void)0)
#endif

namespace {

// (CALLER only makes the instantiations of the two kernels distinct: the host pass of hipcc 7.2 rejects the second request
// for one and the same specialization with a bogus "substitution failure".)
// STAGGER (8-wave tile only): waves 4-7 run half a K step behind waves 0-3 (see the main loop).  MH_DMA_STAGGER=0 in the
// environment selects the lockstep instantiation at run time (A/B measurements: scripts/bench_tiles.py).
template <class T, bool A_KMAJOR, bool B_KMAJOR, int CALLER, bool STAGGER>
__device__ __forceinline__ void gemm_dma_tile(const GemmParams& p, int tile_m, int tile_n, int kbeg, int kend,
                                              unsigned char* smem) {
    constexpr int S = T::S, PA = T::PA, PB = T::PB, NW = T::NW;
    static_assert(T::LDS_BYTES >= NW * 32 * 68 * 4, "the ring doubles as epilogue staging");
    const int m0 = tile_m * T::BM, n0 = tile_n * T::BN;
    const int nk = (kend - kbeg + BK - 1) / BK;

    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63;
    constexpr int MT = T::MT;
    const int wm = (w / T::WN) * (16 * MT), wn = (w % T::WN) * 64;

    const __amdgpu_buffer_rsrc_t ra_src = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, (short)0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb_src = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, (short)0, (int)p.b_bytes, 0x00020000);
    int va[PA], vb[PB];
    piece_offsets<A_KMAJOR, T::BM, NW, PA>(p.lda, m0, w, l, va);
    piece_offsets<B_KMAJOR, T::BN, NW, PB>(p.ldb, n0, w, l, vb);
    const int a_step = A_KMAJOR ? BK * p.lda * 2 : BK * 2;   // source bytes per K step
    const int b_step = B_KMAJOR ? BK * p.ldb * 2 : BK * 2;
    const int a_off0 = A_KMAJOR ? kbeg * p.lda * 2 : kbeg * 2;
    const int b_off0 = B_KMAJOR ? kbeg * p.ldb * 2 : kbeg * 2;

    auto issue = [&](int t) {  // DMA the operand tiles of K step t into ring slot t % S (PA + PB pieces per wave)
        unsigned char* slot = smem + (t % S) * T::STAGE_BYTES;
#pragma unroll
        for (int h = 0; h < PA; ++h)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra_src, (lds_void*)(slot + (w + NW * h) * 1024), 16, va[h],
                                                     a_off0 + t * a_step, 0, 0);
#pragma unroll
        for (int h = 0; h < PB; ++h)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb_src, (lds_void*)(slot + T::A_BYTES + (w + NW * h) * 1024), 16, vb[h],
                                                     b_off0 + t * b_step, 0, 0);
    };

    f32x4 acc[4][MT];  // [j (n tile)][i (m tile)]
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int t = 0; t < S - 1 && t < nk; ++t) issue(t);

    // my pieces of step `step` have landed once at most the S - 2 younger steps' DMA instructions are still pending
    auto wait_step = [&](int step) {
        const int younger = min(S - 2, nk - 1 - step);
        if (younger >= 2) wait_vmcnt<2 * (PA + PB)>();
        else if (younger == 1) wait_vmcnt<PA + PB>();
        else wait_vmcnt<0>();
    };
    bf16x8 fa[MT], fb[4];
    auto load_step = [&](int t) {
        const unsigned char* ta = smem + (t % S) * T::STAGE_BYTES;
        const unsigned char* tb = ta + T::A_BYTES;
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = read_frag<B_KMAJOR, T::BN>(tb, wn + 16 * j);
#pragma unroll
        for (int i = 0; i < MT; ++i) fa[i] = read_frag<A_KMAJOR, T::BM>(ta, wm + 16 * i);
        // Order matters: (1) all 24 fragment reads of the K step go out first (left alone, the scheduler interleaves them two
        // fragments at a time to save registers and every group of 4 MFMAs then waits on LDS: +2 %); (2) only then the DMA
        // pieces that refill the slot step t-1 vacated -- an LDS-DMA instruction costs the wave ~100 issue cycles, and issued
        // in front of the reads (as this kernel first did) it held the whole step back: grouped wgrad 754 -> 1050 TFLOP/s.
        __builtin_amdgcn_sched_barrier(0);
        if (t + S - 1 < nk) issue(t + S - 1);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto math_step = [&]() {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[j][i], 0, 0, 0);
    };

    if constexpr (NW == 8 && STAGGER) {
        // Two waves share every SIMD (w and w + 4).  Run in lockstep they both read / issue DMA (no MFMA on the SIMD for
        // ~500 cycles) and then both compute (the matrix pipe serialises their MFMAs): measured 1125 ns per K step against
        // 657 ns for the MFMAs alone and 652 ns for the DMA stream alone (scripts/bench_persist_ablate.py).  Here waves 4-7
        // run HALF A STEP behind waves 0-3 -- two barriers per K step, each wave alternates a load phase (fragment reads +
        // DMA issue) and a math phase (32 MFMAs) -- so one partner's MFMAs cover the other's reads and DMA issue:
        //     phase 2t   : waves 0-3 load step t      | waves 4-7 compute step t-1
        //     phase 2t+1 : waves 0-3 compute step t   | waves 4-7 load step t      ; everyone waits for ITS pieces of step t+1
        // Pieces of step t+1 are first read in phase 2t+2, behind the barrier that follows those waits; the slot refilled in
        // a load phase of step t (step t+S-1 -> slot of step t-1) was last read in phase 2t-1.  Same K order per output
        // element as the lockstep loop: bit-identical results.
        wait_step(0);
        __builtin_amdgcn_s_barrier();
#if MH_DMA_PRIO == 1
        if (w >= 4) __builtin_amdgcn_s_setprio(1);
#endif
        if (w < 4) {                   // wave-uniform role; both roles execute 2 nk barriers
            for (int t = 0; t < nk; ++t) {
                load_step(t);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                MH_PRIO_UP(); math_step(); MH_PRIO_DOWN();
                if (t + 1 < nk) wait_step(t + 1);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            __builtin_amdgcn_s_barrier();          // phase 0: nothing to compute yet
            __builtin_amdgcn_sched_barrier(0);
            for (int t = 0; t < nk; ++t) {
                load_step(t);                      // phase 2t+1
                if (t + 1 < nk) wait_step(t + 1);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                MH_PRIO_UP(); math_step(); MH_PRIO_DOWN();   // phase 2t+2
                if (t + 1 < nk) {
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // (the last math phase touches no LDS: the ring is free for the epilogue staging of the early waves)
        }
    } else {
        for (int t = 0; t < nk; ++t) {
            wait_step(t);
            __builtin_amdgcn_s_barrier();   // everybody's pieces of step t are in LDS; step t-1 has been read by everybody
            load_step(t);
            math_step();                    // (3) the MFMAs, whose operands arrive while the DMA instructions issue
        }
        __builtin_amdgcn_s_barrier();   // all reads of the ring are done: reuse it as epilogue staging
    }

    float* st = reinterpret_cast<float*>(smem) + w * (32 * 68);
    if (p.flags & MH_GEMM_ATOMIC) gemm_epilogue_atomic<MT>(p, acc, st, m0 + wm, n0 + wn);
    else gemm_epilogue_store<MT>(p, acc, st, m0 + wm, n0 + wn);
}

template <class T, bool A_KMAJOR, bool B_KMAJOR, bool STAGGER>
// Second launch bound = workgroups per CU: without it the 128- and 256-thread tiles were given 304 VGPRs, i.e. ONE wave per
// SIMD and one workgroup per CU instead of the two (four) their LDS footprint was sized for.
__global__ __launch_bounds__(T::NT, T::MIN_WG) void gemm_dma_kernel(GemmParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[T::LDS_BYTES];  // the ONLY LDS object
    const int nwg = p.tiles_m * p.tiles_n;
    const int id = xcd_remap(blockIdx.x, nwg);
    int tile_m, tile_n;
    raster_tile<4>(p, id, tile_m, tile_n);
    const int kbeg = blockIdx.y * p.k_per_split;
    gemm_dma_tile<T, A_KMAJOR, B_KMAJOR, 0, STAGGER>(p, tile_m, tile_n, kbeg, min(p.K, kbeg + p.k_per_split), smem);
}

typedef Tile<2, 4, 4> T256;      // 256 x 256
typedef Tile<2, 2, 3> T256x128;  // 256 x 128
typedef Tile<1, 4, 3> T128x256;  // 128 x 256
typedef Tile<1, 2, 4> T128;      // 128 x 128, two 128 x 64 waves
typedef Tile<2, 2, 4, 4> T128q;  // 128 x 128, four 64 x 64 waves
// (round 3 also tried Tile<4, 2, 4, 4> / Tile<2, 4, 4, 4>: 256 x 128 / 128 x 256 with EIGHT 64 x 64 waves, staggered, one workgroup
//  per CU -- 20-45 % slower than the 128^2 kernels on every shape of the step, profiles/r03_gemm_pp_ablation.txt: at 0.5 LDS reads
//  and 3 DMA pieces per 16 MFMAs a wave's load phase is longer than its partner's math phase)

// Grouped weight-gradient launch: ONE grid over the 256x256 tiles of many independent TN problems
// (dW_i[M_i, N_i] = dY_i^T X_i, K_i = tokens), no split-K.  Workgroup b (placed on XCD b % 8 by the hardware) runs entry
// b / 8 of that XCD's host-built tile queue, so that the tiles of one problem run side by side under one L2 and sweep
// K together: a dY / X panel is then fetched from HBM once per problem instead of once per tile.
template <class T, bool STAGGER>
__global__ __launch_bounds__(T::NT) void gemm_dma_grouped_tn_kernel(const MhGroupedGemm* __restrict__ table, int n_problems,
                                                                    const uint32_t* __restrict__ queues, int queue_len) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[T::LDS_BYTES];
    const uint32_t e = queues[(size_t)(blockIdx.x & 7) * queue_len + (blockIdx.x >> 3)];   // uniform: scalar load
    if (e == 0xFFFFFFFFu || (int)(e >> 16) >= n_problems) return;
    const MhGroupedGemm g = table[e >> 16];
    GemmParams p;
    p.A = (const bf16_t*)g.A; p.B = (const bf16_t*)g.B; p.C = g.C;
    p.bias = nullptr; p.res = nullptr; p.aux_in = nullptr; p.aux_out = nullptr; p.colsum = nullptr;
    p.M = g.M; p.N = g.N; p.K = g.K; p.lda = g.lda; p.ldb = g.ldb; p.ldc = g.ldc; p.ldr = 0; p.ldaux = 0;
    p.flags = MH_GEMM_OUT_F32 | (g.accumulate ? MH_GEMM_ATOMIC : 0);
    p.tiles_m = (g.M + T::BM - 1) / T::BM; p.tiles_n = (g.N + T::BN - 1) / T::BN; p.k_per_split = g.K; p.fast = 1;
    p.a_bytes = (unsigned)(((long)(g.K - 1) * g.lda + g.M) * 2);
    p.b_bytes = (unsigned)(((long)(g.K - 1) * g.ldb + g.N) * 2);
    const int tile_m = (e >> 8) & 0xff, tile_n = e & 0xff;
    if (tile_m >= p.tiles_m || tile_n >= p.tiles_n) return;
    gemm_dma_tile<T, true, true, 1, STAGGER>(p, tile_m, tile_n, 0, g.K, smem);
}

template <class T>
int launch_dma(int layout, GemmParams& p, hipStream_t s, bool stagger = true) {
    const int M = p.M, N = p.N, K = p.K;
    const bool a_kmajor = layout == 2, b_kmajor = layout != 0;
    p.tiles_m = ceil_div(M, T::BM); p.tiles_n = ceil_div(N, T::BN);
    const long a_reach = a_kmajor ? (long)(ceil_div(K, BK) * BK) * p.lda * 2 : (long)(p.tiles_m * T::BM) * p.lda * 2;
    const long b_reach = b_kmajor ? (long)(ceil_div(K, BK) * BK) * p.ldb * 2 : (long)(p.tiles_n * T::BN) * p.ldb * 2;
    if (a_reach + 65536 >= (1L << 31) || b_reach + 65536 >= (1L << 31)) return -2;
    int splits = 1;
    if (p.flags & MH_GEMM_ATOMIC) {
        const int tiles = p.tiles_m * p.tiles_n, ksteps = ceil_div(K, BK);
        const int slots = 256 * (T::LDS_BYTES > 80 * 1024 ? 1 : 2);
        splits = max(1, min(min(slots / max(tiles, 1), ksteps / 8), 32));
    }
    const int ksteps_per = ceil_div(ceil_div(K, BK), splits);
    p.k_per_split = ksteps_per * BK;
    splits = ceil_div(K, p.k_per_split);
    dim3 grid(p.tiles_m * p.tiles_n, splits), block(T::NT);
    if (T::NW == 8 && stagger) {
        switch (layout) {
            case 0: hipLaunchKernelGGL((gemm_dma_kernel<T, false, false, T::NW == 8>), grid, block, 0, s, p); break;
            case 1: hipLaunchKernelGGL((gemm_dma_kernel<T, false, true, T::NW == 8>), grid, block, 0, s, p); break;
            default: hipLaunchKernelGGL((gemm_dma_kernel<T, true, true, T::NW == 8>), grid, block, 0, s, p); break;
        }
        return 0;
    }
    switch (layout) {
        case 0: hipLaunchKernelGGL((gemm_dma_kernel<T, false, false, false>), grid, block, 0, s, p); break;
        case 1: hipLaunchKernelGGL((gemm_dma_kernel<T, false, true, false>), grid, block, 0, s, p); break;
        default: hipLaunchKernelGGL((gemm_dma_kernel<T, true, true, false>), grid, block, 0, s, p); break;
    }
    return 0;
}

}  // namespace

extern "C" int mh_gemm_grouped_tn(const MhGroupedGemm* table_device, int n_problems, const uint32_t* tile_queues,
                                  int queue_len, void* stream) {
    MH_CHECK_ARG(table_device && tile_queues && n_problems > 0 && n_problems < 65536 && queue_len > 0,
                 "mh_gemm_grouped_tn: bad arguments");
    hipLaunchKernelGGL((gemm_dma_grouped_tn_kernel<T256, true>), dim3(8 * queue_len), dim3(T256::NT), 0, (hipStream_t)stream,
                       table_device, n_problems, tile_queues, queue_len);
    MH_LAUNCH_CHECK();
    return 0;
}

// Called by mh_gemm_bf16 / mh_gemm_bf16_tile (gemm.hip) after argument validation.  Returns -2 (without touching the
// error string) when the problem does not qualify for the DMA path, so that the caller can use the general kernel.
int gemm_dma_dispatch(int tile, int layout, int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C,
                      int ldc, int flags, const float* bias, const float* res, int ldr, const void* aux_in, void* aux_out,
                      int ldaux, float* colsum, void* stream) {
    const bool a_kmajor = layout == 2, b_kmajor = layout != 0;
    if ((!a_kmajor || !b_kmajor) && K % BK != 0) return -2;   // a K tail inside a K-minor row would wrap, not read zero
    if (lda % 8 || ldb % 8) return -2;
    GemmParams p;
    p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = C;
    p.bias = bias; p.res = res; p.aux_in = (const bf16_t*)aux_in; p.aux_out = (bf16_t*)aux_out; p.colsum = colsum;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldr = ldr; p.ldaux = ldaux; p.flags = flags;
    const long a_ext = a_kmajor ? ((long)(K - 1) * lda + M) * 2 : ((long)(M - 1) * lda + K) * 2;
    const long b_ext = b_kmajor ? ((long)(K - 1) * ldb + N) * 2 : ((long)(N - 1) * ldb + K) * 2;
    p.fast = 1; p.a_bytes = (unsigned)a_ext; p.b_bytes = (unsigned)b_ext;
    hipStream_t s = (hipStream_t)stream;
    int rc;
    switch (tile) {
        case MH_TILE_DMA_256: rc = launch_dma<T256>(layout, p, s); break;
        case MH_TILE_DMA_256x128: rc = launch_dma<T256x128>(layout, p, s); break;
        case MH_TILE_DMA_128x256: rc = launch_dma<T128x256>(layout, p, s); break;
        case MH_TILE_DMA_128: rc = launch_dma<T128>(layout, p, s); break;
        case MH_TILE_DMA_128x4: rc = launch_dma<T128q>(layout, p, s); break;
        case MH_TILE_DMA_256_LOCKSTEP: rc = launch_dma<T256>(layout, p, s, false); break;
        default: return -2;
    }
    if (rc) return rc;
    MH_LAUNCH_CHECK();
    return 0;
}
