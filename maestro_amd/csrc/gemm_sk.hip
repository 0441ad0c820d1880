// Stream-K bf16 GEMM for gfx950: ONE four-wave workgroup per CU, (32 MT) x 128 x 64 tiles with a (16 MT) x 64 wave tile,
// a hand-placed software pipeline, and an in-kernel fix-up of the tiles that two or more workgroups share
// (mh_gemm_bf16_sk; tiles MH_TILE_SK_192 / MH_TILE_SK_256).  Written for the long-K, narrow-N problems of the transformer
// blocks -- fc2, out-proj and three of the four dgrads: the nn.Linear call sites of vit_pytorch's Attention / FeedForward
// constructed at /root/reference/maestro/ssl/mae.py:135-174 -- whose 128 x 128 tilings leave a quarter of the workgroup
// slots empty (M = 8192, N = 768: 384 tiles on 512 slots) and whose 64 x 64 wave tiles read 0.5 LDS fragments per MFMA.
//
// Work split.  The launch owns U = tiles x (K / 64) units (one K step of one tile); persistent workgroup w (logical id: the
// XCD-aware remap of blockIdx.x, so that the workgroups of one XCD own a contiguous run) takes units [w U / P, (w + 1) U / P):
// at most one tile's tail, whole tiles, and one tile's head.  A segment that does not contain its tile's LAST K step is
// "non-finishing": its fp32 partial goes to the workgroup's slot of the workspace.  The workgroup that owns the last K step
// (the finisher) adds the partials of the workgroups before it in workgroup order -- a fixed order: results do not depend on
// timing -- and runs the epilogue.  Every workgroup has at most ONE non-finishing segment (its last one) and processes it
// FIRST, before anything it could wait for: a partial is published ~one tile time before its finisher asks for it, and no
// workgroup ever waits before it has published (no cycle of waits).
// Hand-off (guide, "Workgroup dispatch, XCD placement & inter-workgroup visibility", first row of the measured table): every
// partial byte is stored and loaded `sc1` (16-byte buffer accesses), every storing wave waits vmcnt(0), a workgroup barrier,
// ONE lane's agent-scope flag store; the finisher's lane 0 polls the flags with sc1 loads, the other waves load behind the
// barrier it then joins; the finisher clears the flags it consumed, so a workspace is all zeros between launches.
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
#include <type_traits>
#include <algorithm>

namespace {

enum { SK_EPI_BF16 = 0, SK_EPI_F32 = 1 };

struct SkArgs {
    float* ws;      // P partial-tile slots of MT x 16 KiB (accumulator order)
    int* flags;     // P arrival words; zero between launches
    int P;          // persistent workgroups = gridDim.x
    int nk;         // K steps per tile
    int tiles;      // output tiles
    int* err;       // set to 1 if a flag wait ran into its bound (a lost producer: results are wrong, nothing hangs)
};

__device__ __forceinline__ void sk_swap16(uint32_t& a, uint32_t& b) {
    const u32x2 r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    a = r[0]; b = r[1];
}

template <bool B_KMAJOR, int MT, int EPI>
__global__ __launch_bounds__(NT, 1) void gemm_sk_kernel(GemmParams p, SkArgs sk) {
    constexpr int TBM = 32 * MT, A_BYTES = TBM * 128, B_BYTES = TILE_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * A_BYTES + 2 * B_BYTES];   // A0 A1 B0 B1
    unsigned char* const sb = smem + 2 * A_BYTES;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, g = l >> 4, lm = l & 15;
    const int wm = (w >> 1) * (16 * MT), wn = (w & 1) * 64;

    // ---- this workgroup's units
    const int lw = xcd_remap(blockIdx.x, sk.P);
    const long U = (long)sk.tiles * sk.nk;
    const long ub = (long)lw * U / sk.P, ue = (long)(lw + 1) * U / sk.P;
    if (ub >= ue) return;                                   // (uniform; U < P: more workgroups than units)
    const int nk = sk.nk;
    const int t0 = (int)(ub / nk), t1 = (int)((ue - 1) / nk), nseg = t1 - t0 + 1;
    const int k_first = (int)(ub - (long)t0 * nk), k_last_end = (int)(ue - (long)t1 * nk);
    const bool rot = k_last_end < nk;                       // the last segment is non-finishing: it runs first
    // processing index q -> (tile, first K step, end K step)
    auto segment = [&](int q, int& tile, int& kb, int& ke) {
        const int s = rot ? (q == 0 ? nseg - 1 : q - 1) : q;
        tile = t0 + s;
        kb = s == 0 ? k_first : 0;
        ke = s == nseg - 1 ? k_last_end : nk;
    };
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

    // ---- epilogue state
    const __amdgpu_buffer_rsrc_t rc_dst = __builtin_amdgcn_make_buffer_rsrc(
        p.C, (short)0, (int)(((long)(p.M - 1) * p.ldc + p.N) * (EPI == SK_EPI_F32 ? 4 : 2)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.res, (short)0, EPI == SK_EPI_F32 ? (int)(((long)(p.M - 1) * p.ldr + p.N) * 4) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rbias = __builtin_amdgcn_make_buffer_rsrc((void*)p.bias, (short)0, p.bias ? p.N * 4 : 0, 0x00020000);
    constexpr int PART = 4 * MT * NT * 4;                   // floats per partial tile
    constexpr int SC1 = 16;                                 // cache-policy bit of the buffer instructions (sc1)
    const __amdgpu_buffer_rsrc_t rws = __builtin_amdgcn_make_buffer_rsrc((void*)sk.ws, (short)0, (int)((long)sk.P * PART * 4 > 0x7fffffffL ? 0x7fffffff : (long)sk.P * PART * 4), 0x00020000);
    const int lane_off = tid * 16;

    auto epilogue = [&](int m0, int n0) {
        const int row = m0 + wm + lm;
        if constexpr (EPI == SK_EPI_F32) {
            // lane (lm, g) owns 4 consecutive columns of row 16 i + lm in every 16-column block: 16-byte accesses, four lanes per
            // 64-byte row segment.  The residual rows of a batch of row blocks are requested before its first store.
            f32x4 bias4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                bias4[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rbias, (n0 + wn + 16 * j + 4 * g) * 4, 0, 0));
            constexpr int BATCH = MT % 3 == 0 ? 3 : 4;
#pragma unroll
            for (int i0 = 0; i0 < MT; i0 += BATCH) {
                f32x4 add[BATCH][4];
#pragma unroll
                for (int ii = 0; ii < BATCH; ++ii)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        add[ii][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                            rres, ((row + 16 * (i0 + ii)) * p.ldr + n0 + wn + 16 * j + 4 * g) * 4, 0, 0));
#pragma unroll
                for (int ii = 0; ii < BATCH; ++ii)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x4 v = acc[j][i0 + ii] + (bias4[j] + add[ii][j]);   // (the order of gemm_common.hpp's fp32 epilogue)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rc_dst,
                                                               ((row + 16 * (i0 + ii)) * p.ldc + n0 + wn + 16 * j + 4 * g) * 4, 0, 0);
                    }
            }
        } else {
            const int col_swp = wn + 4 * g + 12 * (g & 1);   // after the 16-lane-row swap: 8 consecutive columns of a block pair
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int jp = 0; jp < 2; ++jp) {
                    const f32x4 vx = acc[2 * jp][i], vy = acc[2 * jp + 1][i];
                    uint32_t x0 = pack_bf2(vx[0], vx[1]), x1 = pack_bf2(vx[2], vx[3]);
                    uint32_t y0 = pack_bf2(vy[0], vy[1]), y1 = pack_bf2(vy[2], vy[3]);
                    sk_swap16(x0, y0);
                    sk_swap16(x1, y1);
                    __builtin_amdgcn_raw_buffer_store_b128((u32x4){x0, x1, y0, y1}, rc_dst,
                                                           ((row + 16 * i) * p.ldc + n0 + col_swp + 32 * jp) * 2, 0, 0);
                }
        }
    };

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
            // non-finishing segment: publish the partial (accumulator order, 4 KiB per wave instruction)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < MT; ++i)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[j][i]), rws, lane_off + (j * MT + i) * (NT * 16),
                                                           lw * (PART * 4), SC1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // every storing wave: its partial has left the CU
            __syncthreads();
            if (tid == 0) __hip_atomic_store(sk.flags + lw, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            continue;
        }
        if (kb > 0) {
            // finisher of a shared tile: the workgroups that own units [tile nk, tile nk + kb) each published one partial
            const int wf = (int)((((long)tile * nk + 1) * sk.P - 1) / U);
            if (tid == 0) {
                for (int c = wf; c < lw; ++c) {
                    int spins = 0;
                    while (__hip_atomic_load(sk.flags + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
                        __builtin_amdgcn_s_sleep(4);
                        if (++spins > (1 << 22)) { __hip_atomic_store(sk.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                    }
                    __hip_atomic_store(sk.flags + c, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            __syncthreads();
            for (int c = wf; c < lw; ++c) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < MT; ++i)
                        acc[j][i] += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rws, lane_off + (j * MT + i) * (NT * 16),
                                                                                                      c * (PART * 4), SC1));
            }
        }
        epilogue(m0, n0);
    }
}

template <bool B_KMAJOR, int MT>
void launch_sk(int epi, const GemmParams& p, const SkArgs& sk, hipStream_t s) {
    dim3 g(sk.P), b(NT);
    if (epi == SK_EPI_F32) hipLaunchKernelGGL((gemm_sk_kernel<B_KMAJOR, MT, SK_EPI_F32>), g, b, 0, s, p, sk);
    else hipLaunchKernelGGL((gemm_sk_kernel<B_KMAJOR, MT, SK_EPI_BF16>), g, b, 0, s, p, sk);
}

constexpr long SK_FLAG_BYTES = 4096;   // up to 1008 workgroups + the error word (last int)

}  // namespace

extern "C" long mh_gemm_sk_workspace(int tile, int grid) {
    const int mt = tile == MH_TILE_SK_192 ? 6 : tile == MH_TILE_SK_256 ? 8 : 0;
    if (!mt || grid <= 0 || grid > 1008) return -1;
    return SK_FLAG_BYTES + (long)grid * mt * 16384;
}

extern "C" int mh_gemm_bf16_sk(int tile, int layout, int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C,
                               int ldc, int flags, const float* bias, const float* res, int ldr, void* workspace,
                               long workspace_bytes, int grid, void* stream) {
    const int mt = tile == MH_TILE_SK_192 ? 6 : tile == MH_TILE_SK_256 ? 8 : 0;
    MH_CHECK_ARG(mt, "mh_gemm_bf16_sk: tile %d is not a stream-K tile", tile);
    MH_CHECK_ARG(layout == 0 || layout == 1, "mh_gemm_bf16_sk: layout %d (NT / NN only)", layout);
    MH_CHECK_ARG(M > 0 && N > 0 && K > 0 && A && B && C, "mh_gemm_bf16_sk: empty problem or null operand");
    MH_CHECK_ARG(grid > 0 && grid <= 1008, "mh_gemm_bf16_sk: grid %d", grid);
    MH_CHECK_ARG(workspace && (uintptr_t)workspace % 16 == 0 && workspace_bytes >= mh_gemm_sk_workspace(tile, grid),
                 "mh_gemm_bf16_sk: workspace of %ld bytes needed (16-byte aligned, zeroed once)", mh_gemm_sk_workspace(tile, grid));
    MH_CHECK_ARG(((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) % 16 == 0 && lda % 8 == 0 && ldb % 8 == 0, "mh_gemm_bf16_sk: alignment");
    const int epi = flags == 0 ? SK_EPI_BF16 : flags == (MH_GEMM_OUT_F32 | MH_GEMM_BIAS | MH_GEMM_RESIDUAL) ? SK_EPI_F32 : -1;
    if (epi < 0 || K % BK != 0 || K < 2 * BK || N % BN != 0) return -2;   // -2: not served (the caller picks another tile)
    MH_CHECK_ARG(epi != SK_EPI_F32 || (bias && res && ldr % 4 == 0 && ldc % 4 == 0), "mh_gemm_bf16_sk: fp32 epilogue needs bias, res, ldr / ldc %% 4");
    MH_CHECK_ARG(epi != SK_EPI_BF16 || ldc % 8 == 0, "mh_gemm_bf16_sk: bf16 output needs ldc %% 8 == 0");
    const bool b_kmajor = layout == 1;
    GemmParams p;
    p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = C;
    p.bias = bias; p.res = res; p.aux_in = nullptr; p.aux_out = nullptr; p.colsum = nullptr;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldr = ldr; p.ldaux = 0; p.flags = flags;
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
