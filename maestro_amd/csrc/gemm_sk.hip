// Stream-K bf16 GEMM for gfx950: ONE four-wave workgroup per CU, (32 MT) x 128 x 64 tiles with a (16 MT) x 64 wave tile,
// a hand-placed software pipeline, and an in-kernel fix-up of the tiles that two or more workgroups share
// (mh_gemm_bf16_sk; tiles MH_TILE_SK_192 / MH_TILE_SK_256).  Written for the long-K, narrow-N problems of the transformer
// blocks -- fc2, out-proj and three of the four dgrads: the nn.Linear call sites of vit_pytorch's Attention / FeedForward
// constructed at /root/reference/maestro/ssl/mae.py:135-174 -- whose 128 x 128 tilings leave a quarter of the workgroup
// slots empty (M = 8192, N = 768: 384 tiles on 512 slots) and whose 64 x 64 wave tiles read 0.5 LDS fragments per MFMA.
//
// Work split, hand-off of the shared tiles' partials and the epilogues: gemm_sk_common.hpp.
//
// Pipeline (per K step of 64, per wave; MT = 6: 48 MFMA 16x16x32, 20 ds_read_b128, 10 ds_write_b128, 10 buffer_load x4):
//   operands are register-staged into a double-buffered swizzled LDS image (gemm_reg.hpp: the images, fragment reads and MFMA
//   operand order of gemm.hip, so the same fp32 sums per output element); global loads run two K steps ahead, the LDS stores
//   of step s + 1 and the loads of step s + 2 sit between the MFMAs of step s; the fragments of the SECOND half of step s are
//   read under the MFMAs of its first half and the fragments of the first half of step s + 1 under the last MFMAs of step s,
//   which are deferred past the step's barrier for that purpose (one wave per SIMD: nobody else covers a wave's read
//   latency), one barrier per K step.  The stream does not stop at a
//   segment boundary: the next segment's first step is in LDS and its second in registers when the epilogue starts.
// Epilogues work straight from the accumulator layout (no LDS staging: the operand buffers stay live): fp32 + bias + residual
// (fc2, out-proj) and plain bf16 (dgrads; two v_permlane16_swap make 16-byte row segments), as in gemm_pp.hip.
#include "gemm_reg.hpp"
#include "gemm_sk_common.hpp"
#include <type_traits>
#include <algorithm>

namespace {

template <bool B_KMAJOR, int MT, int EPI>
__global__ __launch_bounds__(NT, 1) void gemm_sk_kernel(GemmParams p, SkArgs sk) {
    constexpr int TBM = 32 * MT, A_BYTES = TBM * 128, B_BYTES = TILE_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * A_BYTES + 2 * B_BYTES];   // A0 A1 B0 B1
    unsigned char* const sb = smem + 2 * A_BYTES;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, g = l >> 4, lm = l & 15;
    const int wm = (w >> 1) * (16 * MT), wn = (w & 1) * 64;

    // ---- this workgroup's units (gemm_sk_common.hpp)
    const SkSplit sp(sk, blockIdx.x);
    if (sp.empty) return;                                   // (uniform; more workgroups than units)
    const int lw = sp.lw, nk = sp.nk, nseg = sp.nseg;
    auto segment = [&](int q, int& tile, int& kb, int& ke) { sp.segment(q, tile, kb, ke); };
    auto origin = [&](int tile, int& m0, int& n0) {
        int tm, tn;
        raster_tile<8>(p, tile, tm, tn);
        m0 = tm * TBM;
        n0 = tn * BN;
    };

    // ---- operand stream: loop-invariant per-thread byte offsets + a uniform cursor (segment, K step)
    const __amdgpu_buffer_rsrc_t ra_src = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, (short)0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb_src = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, (short)0, (int)p.b_bytes, 0x00020000);
    const int a_v0 = ((tid >> 3) * p.lda + (tid & 7) * 8) * 2;
    const int b_v0 = B_KMAJOR ? ((tid >> 4) * p.ldb + (tid & 15) * 8) * 2 : ((tid >> 3) * p.ldb + (tid & 7) * 8) * 2;
    const int a_i = 32 * p.lda * 2, b_i = (B_KMAJOR ? 16 : 32) * p.ldb * 2;
    const int a_step = BK * 2, b_step = B_KMAJOR ? BK * p.ldb * 2 : BK * 2;
    int ld_q = 0, ld_kt, ld_ke, ld_m0, ld_n0;
    {
        int tile;
        segment(0, tile, ld_kt, ld_ke);
        origin(tile, ld_m0, ld_n0);
    }
    u32x4 ra[MT], rb[4];
    auto issue_loads = [&]() {
        const int a_t = ld_m0 * p.lda * 2, b_t = B_KMAJOR ? ld_n0 * 2 : ld_n0 * p.ldb * 2;
        const int a_k = ld_kt * a_step, b_k = ld_kt * b_step;
#pragma unroll
        for (int i = 0; i < MT; ++i) ra[i] = __builtin_amdgcn_raw_buffer_load_b128(ra_src, a_v0 + (a_t + i * a_i), a_k, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) rb[i] = __builtin_amdgcn_raw_buffer_load_b128(rb_src, b_v0 + (b_t + i * b_i), b_k, 0);
    };
    auto advance = [&]() {   // cursor -> next K step of the stream; past the end it stays on the last step (never consumed)
        if (++ld_kt == ld_ke) {
            if (ld_q + 1 < nseg) {
                int tile;
                segment(++ld_q, tile, ld_kt, ld_ke);
                origin(tile, ld_m0, ld_n0);
            } else {
                --ld_kt;
            }
        }
    };

    f32x4 acc[4][MT];
    bf16x8 fa0[MT], fb0[4];      // first-half fragments of the current step
    int sidx = 0;

    // ---- one K step = ONE scheduling region that ends at the step's barrier (MFMAs are not memory operations: around a barrier in the
    // middle of the body the compiler moved them freely and the pinned groups no longer matched).  The second half of a step is cut
    // in two: IA row blocks run right behind the first half, the other IB = MT - IA run at the HEAD of the next body, under the
    // fragment reads of that step's first half -- the LDS image they read was completed by the barrier in between:
    //   head  (1 MFMA of the previous step's last IB row blocks, 1 first-half fragment read) x NRH
    //   main  (1 first-half MFMA, 1 second-half fragment read) x NRH; (PER MFMAs, 1 LDS store of step s + 1, 1 global load of
    //         step s + 2) x NS; the remaining first-half MFMAs and the IA row blocks of the second half; barrier
    // MFMAs run row-block-major (i outer), fragments are read B first, then A in row-block order: a row block's A fragment is needed
    // 4 MFMAs after the one before it.  One wave per SIMD: nobody else covers a wave's LDS latency, so no read may sit right in
    // front of its first use.
    constexpr int NRH = MT + (B_KMAJOR ? 8 : 4);                        // DS reads per half
    constexpr int IB = (NRH + 3) / 4;                                   // row blocks of the second half deferred to the next head
    constexpr int IA = MT - IB;
    constexpr int NS = MT + 4;                                          // LDS stores = global loads per step
    constexpr int RM = 4 * MT + 4 * IA;                                 // MFMAs in main
    constexpr int PER = (RM - NRH - 4) / NS > 0 ? (RM - NRH - 4) / NS : 1;   // MFMAs per (store, load) pair
    static_assert(IA >= 1 && RM >= NRH + NS * PER, "schedule does not fit");
    bf16x8 fa1[MT], fb1[4];      // second-half fragments; row blocks IA .. MT - 1 are consumed by the next head
    auto kstep = [&]() {
        const int cur = sidx & 1;
        const unsigned char* ta = smem + cur * A_BYTES;
        const unsigned char* tb = sb + cur * B_BYTES;
        unsigned char* na = smem + (cur ^ 1) * A_BYTES;
        unsigned char* nb = sb + (cur ^ 1) * B_BYTES;
        // head
#pragma unroll
        for (int j = 0; j < 4; ++j) fb0[j] = read_frag<B_KMAJOR>(tb, wn + 16 * j, 0);
#pragma unroll
        for (int i = 0; i < MT; ++i) fa0[i] = read_frag<false>(ta, wm + 16 * i, 0);
#pragma unroll
        for (int i = IA; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb1[j], fa1[i], acc[j][i], 0, 0, 0);
        // main
        bf16x8 ga[MT], gb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) gb[j] = read_frag<B_KMAJOR>(tb, wn + 16 * j, 1);
#pragma unroll
        for (int i = 0; i < MT; ++i) ga[i] = read_frag<false>(ta, wm + 16 * i, 1);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb0[j], fa0[i], acc[j][i], 0, 0, 0);
        store_tile<false, MT>(na, ra);
        store_tile<B_KMAJOR>(nb, rb);
        issue_loads();
#pragma unroll
        for (int i = 0; i < IA; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gb[j], ga[i], acc[j][i], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) fb1[j] = gb[j];
#pragma unroll
        for (int i = 0; i < MT; ++i) fa1[i] = ga[i];
        // pins
#pragma unroll
        for (int r = 0; r < NRH; ++r) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        if constexpr (4 * IB - NRH > 0) __builtin_amdgcn_sched_group_barrier(0x008, 4 * IB - NRH, 0);
#pragma unroll
        for (int r = 0; r < NRH; ++r) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
#pragma unroll
        for (int r = 0; r < NS; ++r) {
            __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        if constexpr (RM - NRH - NS * PER > 0) __builtin_amdgcn_sched_group_barrier(0x008, RM - NRH - NS * PER, 0);
        ++sidx;
        __syncthreads();
    };
    auto ktail = [&]() {   // the deferred row blocks of a segment's last step; the next head then adds zeros
#pragma unroll
        for (int i = IA; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb1[j], fa1[i], acc[j][i], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) fb1[j] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int i = IA; i < MT; ++i) fa1[i] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
    };

    // ---- epilogue state (gemm_sk_common.hpp)
    const SkEpiDesc ed = sk_epi_desc<EPI>(p, sk, (long)4 * MT * NT * 16);

    // ---- stream prologue: step 0 into LDS buffer 0, step 1 into registers; no deferred row blocks yet
    issue_loads(); advance();
    store_tile<false, MT>(smem, ra);
    store_tile<B_KMAJOR>(sb, rb);
    issue_loads(); advance();
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) fb1[j] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < MT; ++i) fa1[i] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};

    for (int q = 0; q < nseg; ++q) {
        int tile, kb, ke, m0, n0;
        segment(q, tile, kb, ke);
        origin(tile, m0, n0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < MT; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int kt = kb; kt < ke; ++kt) {
            kstep();
            advance();
        }
        ktail();
        if (ke < nk) {
            // non-finishing segment: publish the partial
            sk_store_partial<MT, NT>(ed, acc, lw, w);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // every storing wave: its partial has left the CU
            __syncthreads();
            if (tid == 0) __hip_atomic_store(sk.flags + lw, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            continue;
        }
        if (kb > 0) {
            // finisher of a shared tile: the workgroups that own units [tile nk, tile nk + kb) each published one partial
            const int wf = sp.first_owner(sk, tile);
            if (tid == 0) sk_wait_flags(sk, wf, lw);
            __syncthreads();
            sk_add_partials<MT, NT>(ed, acc, wf, lw, w);
        }
        sk_epilogue<MT, EPI>(p, ed, acc, m0 + wm, n0 + wn);
    }
}

template <bool B_KMAJOR, int MT>
void launch_sk(int epi, const GemmParams& p, const SkArgs& sk, hipStream_t s) {
    dim3 g(sk.P), b(NT);
    if (epi == SK_EPI_F32) hipLaunchKernelGGL((gemm_sk_kernel<B_KMAJOR, MT, SK_EPI_F32>), g, b, 0, s, p, sk);
    else hipLaunchKernelGGL((gemm_sk_kernel<B_KMAJOR, MT, SK_EPI_BF16>), g, b, 0, s, p, sk);
}

}  // namespace

static long sk_slot_bytes(int tile) {   // one partial tile in accumulator order
    return tile == MH_TILE_SK_192 ? 6 * 16384 : tile == MH_TILE_SK_256 ? 8 * 16384 : tile == MH_TILE_SK_DMA_256 ? 256 * 256 * 4 : 0;
}

extern "C" long mh_gemm_sk_workspace(int tile, int grid) {
    if (!sk_slot_bytes(tile) || grid <= 0 || grid > 1008) return -1;
    return SK_FLAG_BYTES + (long)grid * sk_slot_bytes(tile);
}

extern "C" int mh_gemm_bf16_sk(int tile, int layout, int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C,
                               int ldc, int flags, const float* bias, const float* res, int ldr, void* workspace,
                               long workspace_bytes, int grid, void* stream) {
    const int mt = tile == MH_TILE_SK_192 ? 6 : tile == MH_TILE_SK_256 ? 8 : 0;
    MH_CHECK_ARG(sk_slot_bytes(tile), "mh_gemm_bf16_sk: tile %d is not a stream-K tile", tile);
    MH_CHECK_ARG(layout == 0 || layout == 1, "mh_gemm_bf16_sk: layout %d (NT / NN only)", layout);
    MH_CHECK_ARG(M > 0 && N > 0 && K > 0 && A && B && C, "mh_gemm_bf16_sk: empty problem or null operand");
    MH_CHECK_ARG(grid > 0 && grid <= 1008, "mh_gemm_bf16_sk: grid %d", grid);
    MH_CHECK_ARG(workspace && (uintptr_t)workspace % 16 == 0 && workspace_bytes >= mh_gemm_sk_workspace(tile, grid),
                 "mh_gemm_bf16_sk: workspace of %ld bytes needed (16-byte aligned, zeroed once)", mh_gemm_sk_workspace(tile, grid));
    MH_CHECK_ARG(((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) % 16 == 0 && lda % 8 == 0 && ldb % 8 == 0, "mh_gemm_bf16_sk: alignment");
    const int epi = flags == 0 ? SK_EPI_BF16 : flags == (MH_GEMM_OUT_F32 | MH_GEMM_BIAS | MH_GEMM_RESIDUAL) ? SK_EPI_F32 : -1;
    if (epi < 0) return -2;                                                // -2: not served (the caller picks another tile)
    MH_CHECK_ARG(epi != SK_EPI_F32 || (bias && res && ldr % 4 == 0 && ldc % 4 == 0), "mh_gemm_bf16_sk: fp32 epilogue needs bias, res, ldr / ldc %% 4");
    MH_CHECK_ARG(epi != SK_EPI_BF16 || ldc % 8 == 0, "mh_gemm_bf16_sk: bf16 output needs ldc %% 8 == 0");
    const bool b_kmajor = layout == 1;
    GemmParams p;
    p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = C;
    p.bias = bias; p.res = res; p.aux_in = nullptr; p.aux_out = nullptr; p.colsum = nullptr;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldr = ldr; p.ldaux = 0; p.flags = flags;
    if (tile == MH_TILE_SK_DMA_256) {
        const int rc = gemm_sk_dma_launch(layout, epi, p, workspace, grid, stream);
        if (rc) return rc;
        MH_LAUNCH_CHECK();
        return 0;
    }
    if (K % BK != 0 || K < 2 * BK || N % BN != 0) return -2;
    const int tbm = 32 * mt;
    p.tiles_m = ceil_div(M, tbm); p.tiles_n = N / BN; p.k_per_split = K; p.fast = 1;
    const long a_ext = ((long)(M - 1) * lda + K) * 2;
    const long b_ext = b_kmajor ? ((long)(K - 1) * ldb + N) * 2 : ((long)(N - 1) * ldb + K) * 2;
    const long a_reach = (long)(p.tiles_m * tbm) * lda * 2, b_reach = b_kmajor ? (long)K * ldb * 2 : (long)N * ldb * 2;
    const long c_reach = (long)(p.tiles_m * tbm) * ldc * (epi == SK_EPI_F32 ? 4 : 2);
    const long r_reach = epi == SK_EPI_F32 ? (long)(p.tiles_m * tbm) * ldr * 4 : 0;
    const long lim = (1L << 31) - 65536;
    if (a_reach >= lim || b_reach >= lim || c_reach >= lim || r_reach >= lim) return -2;
    p.a_bytes = (unsigned)a_ext; p.b_bytes = (unsigned)b_ext;
    SkArgs sk;
    sk.flags = reinterpret_cast<int*>(workspace);
    sk.err = sk.flags + SK_FLAG_BYTES / 4 - 1;
    sk.ws = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + SK_FLAG_BYTES);
    sk.nk = K / BK;
    sk.tiles = p.tiles_m * p.tiles_n;
    sk.P = (int)std::min<long>(grid, (long)sk.tiles * sk.nk);
    hipStream_t s = (hipStream_t)stream;
    if (mt == 6) { if (b_kmajor) launch_sk<true, 6>(epi, p, sk, s); else launch_sk<false, 6>(epi, p, sk, s); }
    else { if (b_kmajor) launch_sk<true, 8>(epi, p, sk, s); else launch_sk<false, 8>(epi, p, sk, s); }
    MH_LAUNCH_CHECK();
    return 0;
}
