// Stream-K on the eight-wave 256 x 256 x 32 LDS-DMA tile (mh_gemm_bf16_sk, tile MH_TILE_SK_DMA_256): the main loop of gemm_dma.hip
// -- a 4-stage ring filled by buffer_load ... lds, two wave groups half a K step out of phase, 128 x 64 wave tiles -- run by
// PERSISTENT workgroups (one per CU) over an even split of tiles x (K / 32) units, for the long-K, narrow-N problems of the
// transformer blocks (fc2, out-proj, the fc1 / qkv / out-proj dgrads: the nn.Linear call sites of vit_pytorch's Attention /
// FeedForward built at /root/reference/maestro/ssl/mae.py:135-174).  Why this tile: it is the structure of this tree that turns a
// CU-second into the most FLOPs (two waves per SIMD cover each other's LDS / DMA issue; 0.375 fragment reads per MFMA; half the
// operand bytes per FLOP of a 128 x 128 tile: 1.2-1.35 PFLOP/s on long K), but an N = 768 output has 3 tile columns -- 96 tiles for
// 256 CUs at M = 8192 -- and its one-tile-per-workgroup launch pays a ring fill and an exposed epilogue per tile.  Here
//   * every CU gets the same number of K steps whatever the tile count (work split, partial hand-off, epilogues:
//     gemm_sk_common.hpp);
//   * the ring never drains: the DMA cursor runs S - 1 = 3 steps ahead of the MFMAs across segment boundaries, so a workgroup's
//     next tile is already landing while it stores the finished one (the epilogues work from the accumulator layout, no LDS);
//   * workgroups reach their tile ends at different times, so the fp32 + residual epilogues (256 KiB read + 256 KiB written per
//     tile) no longer hit HBM in one burst while the MFMAs idle.
// Ring protocol and LDS images: gemm_dma.hip / gemm_ring.hpp (same fragment reads and MFMA order: the same fp32 sums per output
// element as MH_TILE_DMA_256 on a tile owned by one workgroup; a shared tile adds its K ranges in ascending order).
// Barriers: both wave groups execute two per K step (gemm_dma.hip); a segment end adds ONE more in both groups, at matching places
// -- group 0 (waves 0-3) right behind the second barrier of its step, group 1 behind its MFMAs of that step: the same interval.
#include "gemm_ring.hpp"
#include "gemm_sk_common.hpp"

namespace {

typedef Tile<2, 4, 4> TSK;   // 256 x 256, eight 128 x 64 waves, 4-stage ring (128 KiB)

template <bool B_KMAJOR, int EPI>
__global__ __launch_bounds__(TSK::NT, 1) void gemm_sk_dma_kernel(GemmParams p, SkArgs sk) {
    using T = TSK;
    constexpr int S = T::S, PA = T::PA, PB = T::PB, NW = T::NW, MT = T::MT;
    __shared__ __attribute__((aligned(16))) unsigned char smem[T::LDS_BYTES];
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;
    const int wm = (w / T::WN) * (16 * MT), wn = (w % T::WN) * 64;

    const SkSplit sp(sk, blockIdx.x);
    if (sp.empty) return;
    const int lw = sp.lw, nk = sp.nk, nseg = sp.nseg;
    int nst = 0;                                            // K steps of this workgroup's stream
    for (int q = 0; q < nseg; ++q) {
        int tile, kb, ke;
        sp.segment(q, tile, kb, ke);
        nst += ke - kb;
    }
    auto origin = [&](int tile, int& m0, int& n0) {
        int tm, tn;
        raster_tile<4>(p, tile, tm, tn);
        m0 = tm * T::BM;
        n0 = tn * T::BN;
    };

    // ---- DMA stream: per-lane source offsets relative to a tile's origin + a uniform cursor (segment, K step)
    const __amdgpu_buffer_rsrc_t ra_src = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, (short)0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb_src = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, (short)0, (int)p.b_bytes, 0x00020000);
    int va[PA], vb[PB];
    piece_offsets<false, T::BM, NW, PA>(p.lda, 0, w, l, va);
    piece_offsets<B_KMAJOR, T::BN, NW, PB>(p.ldb, 0, w, l, vb);
    const int a_step = BK * 2, b_step = B_KMAJOR ? BK * p.ldb * 2 : BK * 2;
    int ld_q = 0, ld_kt, ld_ke, ld_a_t, ld_b_t, ld_t = 0;   // ld_t: stream step the next issue() fills (ring slot ld_t % S)
    auto ld_origin = [&](int tile) {
        int m0, n0;
        origin(tile, m0, n0);
        ld_a_t = m0 * p.lda * 2;                            // (in the voffset: the descriptor clips rows >= M)
        ld_b_t = B_KMAJOR ? n0 * 2 : n0 * p.ldb * 2;
    };
    {
        int tile;
        sp.segment(0, tile, ld_kt, ld_ke);
        ld_origin(tile);
    }
    auto issue = [&]() {   // the operand tiles of stream step ld_t into ring slot ld_t % S (PA + PB one-KiB pieces per wave), cursor + 1
        unsigned char* slot = smem + (ld_t & (S - 1)) * T::STAGE_BYTES;
#pragma unroll
        for (int h = 0; h < PA; ++h)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra_src, (lds_void*)(slot + (w + NW * h) * 1024), 16, va[h] + ld_a_t, ld_kt * a_step, 0, 0);
#pragma unroll
        for (int h = 0; h < PB; ++h)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb_src, (lds_void*)(slot + T::A_BYTES + (w + NW * h) * 1024), 16, vb[h] + ld_b_t,
                                                     ld_kt * b_step, 0, 0);
        ++ld_t;
        if (++ld_kt == ld_ke && ld_q + 1 < nseg) {
            // (the empty volatile asm keeps this a BRANCH: if-converted, the next segment's tile coordinates -- ~90 scalar instructions,
            //  two integer divisions -- were computed speculatively in every K step: 106 SALU instructions per step against 19)
            asm volatile("" ::: "memory");
            int tile;
            sp.segment(++ld_q, tile, ld_kt, ld_ke);
            ld_origin(tile);
        }
    };
    static_assert((S & (S - 1)) == 0, "ring slots are taken modulo a power of two");

    // ---- compute cursor
    int c_q = 0, c_tile, c_kb, c_ke, c_kt, c_m0, c_n0;
    sp.segment(0, c_tile, c_kb, c_ke);
    c_kt = c_kb;
    origin(c_tile, c_m0, c_n0);

    f32x4 acc[4][MT];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const SkEpiDesc ed = sk_epi_desc<EPI>(p, sk, (long)4 * MT * T::NT * 16);

    for (int t = 0; t < S - 1 && t < nst; ++t) issue();

    // my pieces of step `step` have landed once at most the S - 2 younger steps' DMA instructions are still pending.  (Behind a
    // segment end the epilogue's loads and stores sit between the pieces in issue order: the counted wait then also waits for most
    // of them -- stricter than needed, never looser: the pieces it is about are older than two steps' pieces either way.)
    auto wait_step = [&](int step) {
        const int younger = min(S - 2, nst - 1 - step);
        if (younger >= 2) wait_vmcnt<2 * (PA + PB)>();
        else if (younger == 1) wait_vmcnt<PA + PB>();
        else wait_vmcnt<0>();
    };
    bf16x8 fa[MT], fb[4];
    auto load_step = [&](int t) {
        const unsigned char* ta = smem + (t & (S - 1)) * T::STAGE_BYTES;
        const unsigned char* tb = ta + T::A_BYTES;
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = read_frag<B_KMAJOR, T::BN>(tb, wn + 16 * j, l);
#pragma unroll
        for (int i = 0; i < MT; ++i) fa[i] = read_frag<false, T::BM>(ta, wm + 16 * i, l);
        __builtin_amdgcn_sched_barrier(0);          // fragment reads first, then the DMA pieces that refill the vacated slot
        if (t + S - 1 < nst) issue();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto math_step = [&]() {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[j][i], 0, 0, 0);
    };
    // the step just computed was the segment's last: hand the tile over (both wave groups, each at its place: see the header)
    auto segment_end = [&]() {
        // Shaped for the register allocator (256 registers per lane, 128 of them accumulators): the accumulators are MODIFIED only by
        // unconditional straight-line code (the partial sums: an empty range of contributors unless this is the finisher of a shared
        // tile) and only READ inside the branches.  Updated inside one arm of a branch they became 128 phis and the kernel spilled
        // 30-250 registers, some of them the K loop's addresses, reloaded from scratch behind vmcnt(0) in every K step.
        const bool fin = c_ke == nk;
        const int wf = (fin && c_kb > 0) ? sp.first_owner(sk, c_tile) : lw;
        if (wf < lw) {
            if (w == 0 && sk_lane() == 0) sk_wait_flags(sk, wf, lw);
            __builtin_amdgcn_s_barrier();
        }
        sk_add_partials<MT, T::NT>(ed, acc, wf, lw, w);
        if (fin) {
            sk_epilogue<MT, EPI>(p, ed, acc, c_m0 + wm, c_n0 + wn);
        } else {
            sk_store_partial<MT, T::NT>(ed, acc, lw, w);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // every storing wave: its partial has left the CU
            __builtin_amdgcn_s_barrier();
            if (w == 0 && sk_lane() == 0) __hip_atomic_store(sk.flags + lw, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __builtin_amdgcn_sched_barrier(0);   // the zeros must not be materialised (in 128 NEW registers) above the epilogue's last uses
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < MT; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        __builtin_amdgcn_sched_barrier(0);
        if (c_q + 1 < nseg) {
            sp.segment(++c_q, c_tile, c_kb, c_ke);
            c_kt = c_kb;
            origin(c_tile, c_m0, c_n0);
        }
    };

    wait_step(0);
    __builtin_amdgcn_s_barrier();
    if (w < 4) {
        for (int t = 0; t < nst; ++t) {
            load_step(t);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1); math_step(); __builtin_amdgcn_s_setprio(0);
            if (t + 1 < nst) wait_step(t + 1);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (++c_kt == c_ke) segment_end();
        }
    } else {
        __builtin_amdgcn_s_barrier();          // phase 0: nothing to compute yet
        __builtin_amdgcn_sched_barrier(0);
        for (int t = 0; t < nst; ++t) {
            load_step(t);
            if (t + 1 < nst) wait_step(t + 1);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1); math_step(); __builtin_amdgcn_s_setprio(0);
            if (++c_kt == c_ke) segment_end();
            if (t + 1 < nst) {
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

}  // namespace

int gemm_sk_dma_launch(int layout, int epi, GemmParams& p, void* workspace, int grid, void* stream) {
    using T = TSK;
    const bool b_kmajor = layout == 1;
    if (p.K % BK != 0 || p.K < 4 * BK || p.N % T::BN != 0 || p.lda % 8 || p.ldb % 8) return -2;
    p.tiles_m = ceil_div(p.M, T::BM); p.tiles_n = p.N / T::BN; p.k_per_split = p.K; p.fast = 1;
    const long a_ext = ((long)(p.M - 1) * p.lda + p.K) * 2;
    const long b_ext = b_kmajor ? ((long)(p.K - 1) * p.ldb + p.N) * 2 : ((long)(p.N - 1) * p.ldb + p.K) * 2;
    const long a_reach = (long)(p.tiles_m * T::BM) * p.lda * 2, b_reach = b_kmajor ? (long)p.K * p.ldb * 2 : (long)p.N * p.ldb * 2;
    const long c_reach = (long)(p.tiles_m * T::BM) * p.ldc * (epi == SK_EPI_F32 ? 4 : 2);
    const long r_reach = epi == SK_EPI_F32 ? (long)(p.tiles_m * T::BM) * p.ldr * 4 : 0;
    const long lim = (1L << 31) - 65536;
    if (a_reach >= lim || b_reach >= lim || c_reach >= lim || r_reach >= lim) return -2;
    p.a_bytes = (unsigned)a_ext; p.b_bytes = (unsigned)b_ext;
    SkArgs sk;
    sk.flags = reinterpret_cast<int*>(workspace);
    sk.err = sk.flags + SK_FLAG_BYTES / 4 - 1;
    sk.ws = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + SK_FLAG_BYTES);
    sk.nk = p.K / BK;
    sk.tiles = p.tiles_m * p.tiles_n;
    const long units = (long)sk.tiles * sk.nk;
    sk.P = (int)(grid < units ? grid : units);
    dim3 g(sk.P), b(T::NT);
    hipStream_t s = (hipStream_t)stream;
    if (epi == SK_EPI_F32) {
        if (b_kmajor) hipLaunchKernelGGL((gemm_sk_dma_kernel<true, SK_EPI_F32>), g, b, 0, s, p, sk);
        else hipLaunchKernelGGL((gemm_sk_dma_kernel<false, SK_EPI_F32>), g, b, 0, s, p, sk);
    } else {
        if (b_kmajor) hipLaunchKernelGGL((gemm_sk_dma_kernel<true, SK_EPI_BF16>), g, b, 0, s, p, sk);
        else hipLaunchKernelGGL((gemm_sk_dma_kernel<false, SK_EPI_BF16>), g, b, 0, s, p, sk);
    }
    return 0;
}
