// Persistent 128x128x64 bf16 GEMM with TWO accumulator sets: the epilogue of output tile t runs INSIDE the main loop of tile
// t + 1 (mh_gemm_bf16_tile, tile = MH_TILE_PP_128; picked by MH_TILE_AUTO for the wide, short-K problems of the transformer
// blocks: qkv / fc1 forward, fc2 dgrad -- the call sites of vit_pytorch's Attention / FeedForward Linears constructed at
// /root/reference/maestro/ssl/mae.py:135-174).
//
// Why: at K = 512 ... 768 a 128x128 tile is 8 ... 12 K steps and its epilogue (bias, erf-GELU, the byte-coded GELU', packs,
// stores: ~15-22 VALU issue slots per output element, 3 B ... 8 B of HBM per element) costs 30-45 % of the launch
// (profiles/r02_gemm_pmc.txt: 49 -> 72 us with the fc1 epilogue, MFMA-busy 40 % -> 32 %); the second resident workgroup cannot
// hide it because it owes the same work to the same pipes at the same time.  Here a workgroup walks several output tiles:
//   * the operand stream is ONE continuous software pipeline over (tile, K step): register-staged tiles exactly as in
//     gemm.hip (same swizzled LDS images, same fragment reads, same MFMA order -> the same fp32 sums), the loads of K step
//     s + 2 and the LDS stores of step s + 1 sit between the MFMAs of step s also ACROSS a tile boundary (no per-tile prologue);
//   * at a tile boundary the 64 accumulators are moved to a second set; the first 8 K steps of the next tile each carry one
//     eighth of the finished tile's epilogue (two 16 x 16 accumulator blocks = 8 outputs per lane), pinned between the MFMAs
//     with sched_group_barrier; the main loop itself is LDS-store- and L2-ingest-bound (~40 % MFMA-busy), so the VALU and
//     store slots the epilogue needs are idle there;
//   * the epilogue works straight from the accumulator layout (no LDS staging: the operand buffers are in use): lane
//     (lm, g) owns 4 consecutive columns of row lm in every 16-column block; for bf16 / byte outputs two v_permlane16_swap
//     make that 8 consecutive columns of TWO neighbouring blocks -> 16-byte (bf16) / 8-byte (u8) stores, 64 contiguous bytes per
//     row and instruction; fp32 + residual output is stored as it stands (16 bytes per lane, 64 per row segment);
//   * every global access of the epilogue is a raw buffer access whose voffset carries the whole (row, column) offset, so
//     rows beyond M are dropped / read as zero by the descriptor: no exec-mask branches inside the scheduling regions.
// Workgroup -> tile map: the tile ids of gemm.hip's XCD-aware grouped raster are cut into 8 contiguous runs (one per XCD,
// block id % 8); the grid / 8 workgroups of an XCD sweep their run together, so the tiles in flight under one L2 are the same
// neighbours as in the one-tile-per-workgroup launch.
#include "gemm_reg.hpp"
#include <type_traits>

// MH_PP_PRIO = 1 (round 6): the wave raises its priority over the MFMA body of every K step and drops it before the barrier, as gemm.hip's
// K loop does: of the two co-resident workgroups the one whose fragments have arrived issues its MFMAs back to back.  Same box, three
// alternating rounds of 30 C3 steps: 17.24 / 17.12 / 17.15 -> 17.02 / 17.00 / 17.01 ms (profiles/r06_experiments.md #21).  0: A/B aid.
#ifndef MH_PP_PRIO
#define MH_PP_PRIO 1
#endif

namespace {

enum { EPI_BF16 = 0, EPI_GELU = 1, EPI_MULAUX = 2, EPI_F32 = 3 };
constexpr int EPI_STEPS = 8;   // K steps of tile t + 1 that carry tile t's epilogue (two accumulator blocks each)

__device__ __forceinline__ void swap16(uint32_t& a, uint32_t& b) {
    // rows of 16 lanes: the odd rows of a are exchanged with the even rows of b (v_permlane16_swap_b32)
    const u32x2 r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    a = r[0]; b = r[1];
}

// DIAG (diagnostic builds for the ablation in profiles/: what does the interleaved epilogue cost, and which part of it?):
//   0 the kernel;  1 no epilogue at all (main loop only: nothing is written);  2 the epilogue's arithmetic without its
//   global stores (results kept alive by an empty asm);  3 its stores without the GELU arithmetic (raw accumulators packed)
template <bool B_KMAJOR, int EPI, int DIAG = 0>
__global__ __launch_bounds__(NT, 2) void gemm_pp_kernel(GemmParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * TILE_BYTES];  // A0 A1 B0 B1
    const int T = p.tiles_m * p.tiles_n, G = gridDim.x;
    const int x = blockIdx.x & 7, slot = blockIdx.x >> 3, per_x = G >> 3;
    const int q = T >> 3, r8 = T & 7;
    const int run0 = x < r8 ? x * (q + 1) : r8 * (q + 1) + (x - r8) * q;   // this XCD's run of tile ids
    const int cnt = q + (x < r8 ? 1 : 0);
    if (slot >= cnt) return;   // (uniform: the whole workgroup)
    const int n_my = (cnt - slot + per_x - 1) / per_x;
    const int nk = p.K / BK;
    auto coords = [&](int t, int& m0, int& n0) {
        int tm, tn;
        raster_tile<8>(p, run0 + slot + t * per_x, tm, tn);
        m0 = tm * BM;
        n0 = tn * BN;
    };

    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, g = l >> 4, lm = l & 15;
    const int wm = (w >> 1) * 64, wn = (w & 1) * 64;

    // ---- operand stream: per-thread byte offsets inside a tile (loop invariant) + a scalar cursor (tile, K step)
    const __amdgpu_buffer_rsrc_t ra_src = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, (short)0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb_src = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, (short)0, (int)p.b_bytes, 0x00020000);
    const int a_v0 = ((tid >> 3) * p.lda + (tid & 7) * 8) * 2;                                   // row r, 16-B chunk c
    const int b_v0 = B_KMAJOR ? ((tid >> 4) * p.ldb + (tid & 15) * 8) * 2 : ((tid >> 3) * p.ldb + (tid & 7) * 8) * 2;
    const int a_i = 32 * p.lda * 2, b_i = (B_KMAJOR ? 16 : 32) * p.ldb * 2;                         // per i = 0..3
    const int a_step = BK * 2, b_step = B_KMAJOR ? BK * p.ldb * 2 : BK * 2;                        // per K step (scalar offset)
    int ld_kt = 0, ld_m0, ld_n0, nx_m0, nx_n0;   // cursor: K step and tile origin of the next fetch; origin of the tile after it
    coords(0, ld_m0, ld_n0);
    coords(min(1, n_my - 1), nx_m0, nx_n0);
    u32x4 ra[4], rb[4];
    auto issue_loads = [&]() {
        const int a_t = ld_m0 * p.lda * 2, b_t = B_KMAJOR ? ld_n0 * 2 : ld_n0 * p.ldb * 2;       // tile base (in the voffset:
        const int a_k = ld_kt * a_step, b_k = ld_kt * b_step;                                      //  the descriptor clips rows >= M)
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[i] = __builtin_amdgcn_raw_buffer_load_b128(ra_src, a_v0 + (a_t + i * a_i), a_k, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) rb[i] = __builtin_amdgcn_raw_buffer_load_b128(rb_src, b_v0 + (b_t + i * b_i), b_k, 0);
    };
    auto advance = [&]() {   // cursor -> the next K step of the stream (past the end: the last tile again, never consumed).
        const bool wrap = ++ld_kt == nk;   // Wraps once per tile (K step nk - 3 of tile t: nk >= 8); the tile loop below
        ld_kt = wrap ? 0 : ld_kt;          // refreshes (nx_m0, nx_n0) at the start of every tile, well before that.
        ld_m0 = wrap ? nx_m0 : ld_m0;
        ld_n0 = wrap ? nx_n0 : ld_n0;
    };

    f32x4 acc[4][4], prev[4][4];   // [j (n block)][i (m block)]
    int sidx = 0;                  // stream step (LDS buffer = sidx & 1)

    // ---- epilogue state of the FINISHED tile (per-lane byte offsets; everything else is scalar)
    const __amdgpu_buffer_rsrc_t rc_dst = __builtin_amdgcn_make_buffer_rsrc(
        p.C, (short)0, (int)(((long)(p.M - 1) * p.ldc + p.N) * (EPI == EPI_F32 ? 4 : 2)), 0x00020000);
    const __amdgpu_buffer_rsrc_t raux = __builtin_amdgcn_make_buffer_rsrc(
        EPI == EPI_GELU ? (void*)p.aux_out : EPI == EPI_MULAUX ? (void*)p.aux_in : (void*)p.res, (short)0,
        (int)(EPI == EPI_F32 ? ((long)(p.M - 1) * p.ldr + p.N) * 4 : EPI == EPI_BF16 ? 0 : (long)(p.M - 1) * p.ldaux + p.N), 0x00020000);
    const __amdgpu_buffer_rsrc_t rbias = __builtin_amdgcn_make_buffer_rsrc((void*)p.bias, (short)0, p.bias ? p.N * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rcs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.colsum, (short)0, EPI == EPI_MULAUX ? (int)((long)((p.M + 63) / 64) * p.N * 4) : 0, 0x00020000);
    const int w_s = __builtin_amdgcn_readfirstlane(w);   // the wave's index as a scalar: epi_setup below rebuilds its lane constants
    int e_c = 0, e_aux = 0, e_bias = 0, e_cs = 0;        // per-lane byte offsets of the finished tile (row wm + lm, pair 0, i = 0)
    f32x4 bias_x = {0, 0, 0, 0}, bias_y = {0, 0, 0, 0};  // bias of the current pair's two column blocks
    f32x4 cs_x = {0, 0, 0, 0}, cs_y = {0, 0, 0, 0};      // EPI_MULAUX: column sums over the wave's 64 rows
    u32x4 ld_x = {0, 0, 0, 0}, ld_y = {0, 0, 0, 0};      // residual (f32x4) / aux bytes (first dword) of the chunk in flight
    auto epi_setup = [&](int m0, int n0) {
        // The lane constants are rebuilt HERE, once per tile, from the lane id (mbcnt) behind an opaque asm: written against the kernel's
        // `lm` / `g` they stayed live across the whole K loop, and with the priority window (MH_PP_PRIO) the GELU kernel spilled three
        // of them -- one scratch reload behind a vmcnt(0) at the head of every tile.
        int lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        asm volatile("" : "+v"(lane));
        const int g = lane >> 4, lm = lane & 15;
        const int wm = (w_s >> 1) * 64, wn = (w_s & 1) * 64;
        const int col_nat = wn + 4 * g;                      // natural layout: block X of pair jp at + 32 jp, block Y at + 32 jp + 16
        const int col_swp = wn + 4 * g + 12 * (g & 1);       // after the 16-lane-row swap: 8 consecutive columns at + 32 jp
        const int row = m0 + wm + lm;
        if constexpr (EPI == EPI_F32) {
            e_c = (row * p.ldc + n0 + col_nat) * 4;
            e_aux = (row * p.ldr + n0 + col_nat) * 4;
        } else {
            e_c = (row * p.ldc + n0 + col_swp) * 2;
            e_aux = row * p.ldaux + n0 + (EPI == EPI_GELU ? col_swp : col_nat);
        }
        e_bias = (n0 + col_nat) * 4;
        e_cs = (((m0 + wm) >> 6) * p.N + n0 + col_nat) * 4;
    };
    // chunk c = 0..7: pair jp = c >> 2 (column blocks 2 jp, 2 jp + 1), row block i = c & 3
    auto epi_loads = [&](auto cc) {
        constexpr int c = decltype(cc)::value, jp = c >> 2, i = c & 3;
        if constexpr (DIAG == 1) return;
        if constexpr (EPI == EPI_GELU || EPI == EPI_F32) {
            if constexpr (i == 0) {
                bias_x = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rbias, e_bias + 128 * jp, 0, 0));
                bias_y = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rbias, e_bias + 128 * jp + 64, 0, 0));
            }
        }
        if constexpr (EPI == EPI_F32) {
            const int o = e_aux + (16 * i * p.ldr + 32 * jp) * 4;
            ld_x = __builtin_amdgcn_raw_buffer_load_b128(raux, o, 0, 0);
            ld_y = __builtin_amdgcn_raw_buffer_load_b128(raux, o + 64, 0, 0);
        }
        if constexpr (EPI == EPI_MULAUX) {
            const int o = e_aux + 16 * i * p.ldaux + 32 * jp;
            ld_x[0] = __builtin_amdgcn_raw_buffer_load_b32(raux, o, 0, 0);
            ld_y[0] = __builtin_amdgcn_raw_buffer_load_b32(raux, o + 16, 0, 0);
        }
    };
    auto epi_math = [&](auto cc, const f32x4 (&src)[4][4]) {
        constexpr int c = decltype(cc)::value, jp = c >> 2, i = c & 3;
        f32x4 vx = src[2 * jp][i], vy = src[2 * jp + 1][i];
        if constexpr (DIAG == 1) {   // keep the accumulators alive, do nothing with them
            asm volatile("" ::"v"(vx), "v"(vy));
            return;
        }
        if constexpr (EPI == EPI_F32) {
            const int o = e_c + (16 * i * p.ldc + 32 * jp) * 4;
            vx = vx + (bias_x + __builtin_bit_cast(f32x4, ld_x));   // (the order of gemm_common.hpp's fp32 epilogue: same bits)
            vy = vy + (bias_y + __builtin_bit_cast(f32x4, ld_y));
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, vx), rc_dst, o, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, vy), rc_dst, o + 64, 0, 0);
            return;
        } else {
            if constexpr (EPI == EPI_GELU) {
                vx += bias_x; vy += bias_y;
                f32x4 cx, dx, cy, dy;
                if constexpr (DIAG == 3) { cx = vx; dx = vx; cy = vy; dy = vy; }
                else { gelu_cdf_pdf4(vx, cx, dx); gelu_cdf_pdf4(vy, cy, dy); }
                const f32x4 gx = vx * dx + cx, gy = vy * dy + cy;      // GELU' = CDF + x PDF
                uint32_t bx = 0, by = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    bx = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(gx[e], 200.f, 26.f), e, bx);
                    by = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(gy[e], 200.f, 26.f), e, by);
                }
                swap16(bx, by);
                if constexpr (DIAG == 2) asm volatile("" ::"v"(bx), "v"(by));
                else __builtin_amdgcn_raw_buffer_store_b64((u32x2){bx, by}, raux, e_aux + 16 * i * p.ldaux + 32 * jp, 0, 0);
                vx *= cx; vy *= cy;
            }
            if constexpr (EPI == EPI_MULAUX) {
                vx *= unpack_dgelu_u8x4(ld_x[0]);
                vy *= unpack_dgelu_u8x4(ld_y[0]);
                if constexpr (i == 0) { cs_x = vx; cs_y = vy; } else { cs_x += vx; cs_y += vy; }
                if constexpr (i == 3) {   // the pair's 32 columns are complete over the wave's 64 rows: fold the 16 row lanes
#pragma unroll
                    for (int o = 1; o < 16; o <<= 1) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            cs_x[e] += __shfl_xor(cs_x[e], o, 64);
                            cs_y[e] += __shfl_xor(cs_y[e], o, 64);
                        }
                    }
                    // (lanes lm = 0 hold the sums; the others store the same values to the same addresses: harmless, branch-free)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, cs_x), rcs, e_cs + 128 * jp, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, cs_y), rcs, e_cs + 128 * jp + 64, 0, 0);
                }
            }
            uint32_t x0 = pack_bf2(vx[0], vx[1]), x1 = pack_bf2(vx[2], vx[3]);
            uint32_t y0 = pack_bf2(vy[0], vy[1]), y1 = pack_bf2(vy[2], vy[3]);
            swap16(x0, y0);
            swap16(x1, y1);
            if constexpr (DIAG == 2) asm volatile("" ::"v"(x0), "v"(x1), "v"(y0), "v"(y1));
            else __builtin_amdgcn_raw_buffer_store_b128((u32x4){x0, x1, y0, y1}, rc_dst, e_c + (16 * i * p.ldc + 32 * jp) * 2, 0, 0);
        }
    };

    // ---- one K step of the stream.  FIRST: first step of a tile (accumulators start from zero); CHUNK >= 0: carries chunk CHUNK
    // of the finished tile's epilogue (from `prev`)
    auto kstep = [&](auto first_c, auto chunk_c) {
        constexpr bool FIRST = decltype(first_c)::value;
        constexpr int CHUNK = decltype(chunk_c)::value;
        const int cur = sidx & 1;
        const unsigned char* ta = smem + cur * TILE_BYTES;
        const unsigned char* tb = smem + (2 + cur) * TILE_BYTES;
        bf16x8 fa[4], fb[4];
#if MH_PP_PRIO == 2     // (A/B: the window opens before the step's first fragment reads)
        __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = read_frag<false>(ta, wm + 16 * i, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = read_frag<B_KMAJOR>(tb, wn + 16 * j, 0);
        if constexpr (CHUNK >= 0) epi_loads(chunk_c);
#if MH_PP_PRIO == 1 || MH_PP_PRIO == 3
        __builtin_amdgcn_s_setprio(MH_PP_PRIO);      // (3, A/B: the highest level instead of 1)
#endif
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], FIRST ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[j][i], 0, 0, 0);
        store_tile<false>(smem + (cur ^ 1) * TILE_BYTES, ra);
        store_tile<B_KMAJOR>(smem + (2 + (cur ^ 1)) * TILE_BYTES, rb);
        issue_loads();
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = read_frag<false>(ta, wm + 16 * i, 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = read_frag<B_KMAJOR>(tb, wn + 16 * j, 1);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[j][i], 0, 0, 0);
        if constexpr (CHUNK >= 0) epi_math(chunk_c, prev);
        // schedule: [first-half fragment reads] [epilogue operand loads] [8 x (2 MFMA, 1 DS write, 1 VMEM read, VALU)]
        //           [second-half reads] [8 x (2 MFMA, VALU)] [epilogue stores]
        // The epilogue's VALU instructions are NOT pinned: measured on the fc1 shapes (scripts/bench_pp_ablate.py, us):
        //   no VALU groups (the scheduler's own placement) 54.9 / 78.8 / 68.2 / 184.5   <- this
        //   NV0 VALU behind every pair of MFMAs (DIAG 4)    57.7 / 80.7 / 70.4 / 184.7
        //   2 NV0 behind each group of fragment reads + the rest behind the MFMA pairs (DIAG 5)  57.1 / 80.4 / 70.0 / 190.8
        constexpr int NV0 = CHUNK < 0 ? 0 : EPI == EPI_GELU ? 14 : EPI == EPI_MULAUX ? 6 : EPI == EPI_F32 ? 2 : 2;   // VALU / slot
        constexpr int NV = DIAG == 4 ? NV0 : DIAG == 5 ? (NV0 * 4 + 6) / 7 : 0;
        constexpr int NR = DIAG == 5 ? NV0 * 2 : 0;                                   // VALU behind each read group
        __builtin_amdgcn_sched_group_barrier(0x100, B_KMAJOR ? 12 : 8, 0);
        if constexpr (NR > 0) __builtin_amdgcn_sched_group_barrier(0x402, NR, 0);
        constexpr int NL = CHUNK < 0 ? 0 : (EPI == EPI_F32 || EPI == EPI_MULAUX ? 2 : 0) + ((EPI == EPI_F32 || EPI == EPI_GELU) && (CHUNK & 3) == 0 ? 2 : 0);
        if constexpr (NL > 0) __builtin_amdgcn_sched_group_barrier(0x020, NL, 0);   // the epilogue's operand loads
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            if constexpr (NV > 0) __builtin_amdgcn_sched_group_barrier(0x402, NV, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x100, B_KMAJOR ? 12 : 8, 0);
        if constexpr (NR > 0) __builtin_amdgcn_sched_group_barrier(0x402, NR, 0);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            if constexpr (NV > 0) __builtin_amdgcn_sched_group_barrier(0x402, NV, 0);
        }
        ++sidx;
#if MH_PP_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        __syncthreads();
        advance();
    };
    using std::integral_constant;
    constexpr integral_constant<bool, true> kFirst{};
    constexpr integral_constant<bool, false> kNext{};
    constexpr integral_constant<int, -1> kNone{};

    // ---- stream prologue: step 0 into LDS buffer 0, step 1 into registers
    issue_loads(); advance();
    store_tile<false>(smem, ra);
    store_tile<B_KMAJOR>(smem + 2 * TILE_BYTES, rb);
    issue_loads();
    __syncthreads();
    // (from here on the cursor is advanced at the END of every step: it then names the step the NEXT issue_loads fetches)
    advance();

    int cm0, cn0;
    coords(0, cm0, cn0);
    kstep(kFirst, kNone);
    for (int kt = 1; kt < nk; ++kt) kstep(kNext, kNone);
    for (int t = 1; t < n_my; ++t) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) prev[j][i] = acc[j][i];
        epi_setup(cm0, cn0);
        coords(t, cm0, cn0);
        coords(min(t + 1, n_my - 1), nx_m0, nx_n0);
        kstep(kFirst, integral_constant<int, 0>{});
        kstep(kNext, integral_constant<int, 1>{});
        kstep(kNext, integral_constant<int, 2>{});
        kstep(kNext, integral_constant<int, 3>{});
        kstep(kNext, integral_constant<int, 4>{});
        kstep(kNext, integral_constant<int, 5>{});
        kstep(kNext, integral_constant<int, 6>{});
        kstep(kNext, integral_constant<int, 7>{});
        for (int kt = EPI_STEPS; kt < nk; ++kt) kstep(kNext, kNone);
    }
    // ---- the last tile's epilogue is exposed.  Round 5: ALL its operand loads (residual rows / GELU' bytes / bias) are requested
    // before the first store.  Written chunk by chunk (load, compute, store, load, ...) the compiler kept every chunk's loads behind
    // the previous chunk's stores (it cannot prove that `res` / `aux_in` and `C` do not overlap): eight dependent memory round trips
    // -- load latency + store completion, vmcnt(0) each -- at the end of every workgroup, ~10 us inside the step where the residual
    // stream is cold, on launches whose workgroups own one or two tiles (out-proj, fc2: 384 tiles for 512 workgroups).
    epi_setup(cm0, cn0);
    if constexpr (DIAG == 0 && (EPI == EPI_F32 || EPI == EPI_MULAUX)) {
        u32x4 tx[8], ty[8];
        f32x4 tbx[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}}, tby[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
        auto fetch = [&](auto cc) {
            constexpr int c = decltype(cc)::value;
            epi_loads(cc);
            tx[c] = ld_x; ty[c] = ld_y;
            if constexpr ((c & 3) == 0) { tbx[c >> 2] = bias_x; tby[c >> 2] = bias_y; }
        };
        auto finish = [&](auto cc) {
            constexpr int c = decltype(cc)::value;
            ld_x = tx[c]; ld_y = ty[c]; bias_x = tbx[c >> 2]; bias_y = tby[c >> 2];
            epi_math(cc, acc);
        };
        fetch(integral_constant<int, 0>{}); fetch(integral_constant<int, 1>{}); fetch(integral_constant<int, 2>{});
        fetch(integral_constant<int, 3>{}); fetch(integral_constant<int, 4>{}); fetch(integral_constant<int, 5>{});
        fetch(integral_constant<int, 6>{}); fetch(integral_constant<int, 7>{});
        finish(integral_constant<int, 0>{}); finish(integral_constant<int, 1>{}); finish(integral_constant<int, 2>{});
        finish(integral_constant<int, 3>{}); finish(integral_constant<int, 4>{}); finish(integral_constant<int, 5>{});
        finish(integral_constant<int, 6>{}); finish(integral_constant<int, 7>{});
    } else {
        auto tail = [&](auto cc) { epi_loads(cc); epi_math(cc, acc); };
        tail(integral_constant<int, 0>{}); tail(integral_constant<int, 1>{}); tail(integral_constant<int, 2>{});
        tail(integral_constant<int, 3>{}); tail(integral_constant<int, 4>{}); tail(integral_constant<int, 5>{});
        tail(integral_constant<int, 6>{}); tail(integral_constant<int, 7>{});
    }
}

template <bool B_KMAJOR>
void launch_pp(int epi, int grid, const GemmParams& p, hipStream_t s, int diag = 0) {
    dim3 g(grid), b(NT);
#ifdef MH_DIAG_TILES   // ablation builds only (MH_BUILD_FLAGS=-DMH_DIAG_TILES): the shipped library does not contain these kernels
    if constexpr (!B_KMAJOR) {   // diagnostic builds: NT only, fc1 (GELU) and plain bf16 epilogues
        if (diag == 1 && epi == EPI_GELU) { hipLaunchKernelGGL((gemm_pp_kernel<false, EPI_GELU, 1>), g, b, 0, s, p); return; }
        if (diag == 2 && epi == EPI_GELU) { hipLaunchKernelGGL((gemm_pp_kernel<false, EPI_GELU, 2>), g, b, 0, s, p); return; }
        if (diag == 3 && epi == EPI_GELU) { hipLaunchKernelGGL((gemm_pp_kernel<false, EPI_GELU, 3>), g, b, 0, s, p); return; }
        if (diag == 4 && epi == EPI_GELU) { hipLaunchKernelGGL((gemm_pp_kernel<false, EPI_GELU, 4>), g, b, 0, s, p); return; }
        if (diag == 5 && epi == EPI_GELU) { hipLaunchKernelGGL((gemm_pp_kernel<false, EPI_GELU, 5>), g, b, 0, s, p); return; }
        if (diag == 1 && epi == EPI_BF16) { hipLaunchKernelGGL((gemm_pp_kernel<false, EPI_BF16, 1>), g, b, 0, s, p); return; }
    }
#else
    (void)diag;
#endif
    switch (epi) {
        case EPI_BF16: hipLaunchKernelGGL((gemm_pp_kernel<B_KMAJOR, EPI_BF16>), g, b, 0, s, p); break;
        case EPI_GELU: hipLaunchKernelGGL((gemm_pp_kernel<B_KMAJOR, EPI_GELU>), g, b, 0, s, p); break;
        case EPI_MULAUX: hipLaunchKernelGGL((gemm_pp_kernel<B_KMAJOR, EPI_MULAUX>), g, b, 0, s, p); break;
        default: hipLaunchKernelGGL((gemm_pp_kernel<B_KMAJOR, EPI_F32>), g, b, 0, s, p); break;
    }
}

}  // namespace

// Which epilogue form a flag set maps to (-1: not served by this kernel)
static int pp_epilogue(int flags) {
    if (flags == 0) return EPI_BF16;
    if (flags == (MH_GEMM_BIAS | MH_GEMM_GELU | MH_GEMM_AUX_DGELU | MH_GEMM_AUX_U8)) return EPI_GELU;
    if (flags == (MH_GEMM_MULAUX | MH_GEMM_AUX_U8 | MH_GEMM_COLSUM)) return EPI_MULAUX;
    if (flags == (MH_GEMM_OUT_F32 | MH_GEMM_BIAS | MH_GEMM_RESIDUAL)) return EPI_F32;
    return -1;
}

// Called by mh_gemm_bf16_tile (gemm.hip) after argument validation.  -2 (error string untouched): the problem does not qualify
// (layout TN, K % 64 != 0 or K < 512, N % 128 != 0, another epilogue, operands beyond the 2 GiB buffer-descriptor range).
int gemm_pp_dispatch(int layout, int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C, int ldc, int flags,
                     const float* bias, const float* res, int ldr, const void* aux_in, void* aux_out, int ldaux, float* colsum,
                     void* stream, int diag) {
    const int epi = pp_epilogue(flags);
#ifndef MH_DIAG_TILES
    if (diag != 0) return -2;   // MH_TILE_PP_128_DIAG1..5 write wrong outputs on purpose: compiled only under -DMH_DIAG_TILES
#else
    if (diag != 0 && (layout != 0 || !(epi == EPI_GELU || (epi == EPI_BF16 && diag == 1)))) return -2;   // no such ablation build
#endif
    if (epi < 0 || layout == 2 || K % BK != 0 || K < EPI_STEPS * BK || N % BN != 0) return -2;
    if (epi == EPI_GELU && !aux_out) return -2;
    const bool b_kmajor = layout == 1;
    GemmParams p;
    p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = C;
    p.bias = bias; p.res = res; p.aux_in = (const bf16_t*)aux_in; p.aux_out = (bf16_t*)aux_out; p.colsum = colsum;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldr = ldr; p.ldaux = ldaux; p.flags = flags;
    p.tiles_m = ceil_div(M, BM); p.tiles_n = N / BN; p.k_per_split = K; p.fast = 1;
    const long a_ext = ((long)(M - 1) * lda + K) * 2;
    const long b_ext = b_kmajor ? ((long)(K - 1) * ldb + N) * 2 : ((long)(N - 1) * ldb + K) * 2;
    const long a_reach = (long)(p.tiles_m * BM) * lda * 2, b_reach = b_kmajor ? (long)K * ldb * 2 : (long)N * ldb * 2;
    const long c_reach = (long)(p.tiles_m * BM) * ldc * (epi == EPI_F32 ? 4 : 2);
    const long x_reach = epi == EPI_F32 ? (long)(p.tiles_m * BM) * ldr * 4 : (long)(p.tiles_m * BM) * ldaux;
    const long lim = (1L << 31) - 65536;
    if (a_reach >= lim || b_reach >= lim || c_reach >= lim || x_reach >= lim) return -2;
    p.a_bytes = (unsigned)a_ext; p.b_bytes = (unsigned)b_ext;
    const int tiles = p.tiles_m * p.tiles_n;
    // MH_PP_TILES_PER_WG (build time): 0 = fully persistent (512 workgroups walk all tiles); n > 0: a workgroup walks at most n tiles
    // (shorter-lived workgroups let the dispatcher interleave another stream's kernel sooner, at one exposed epilogue per n tiles)
#ifndef MH_PP_TILES_PER_WG
#define MH_PP_TILES_PER_WG 0
#endif
    int grid = tiles >= 512 ? 512 : (tiles + 7) / 8 * 8;   // two workgroups per CU; a multiple of 8 (XCD runs)
    if (MH_PP_TILES_PER_WG > 0) grid = max(grid, ((tiles + MH_PP_TILES_PER_WG - 1) / MH_PP_TILES_PER_WG + 7) / 8 * 8);
    if (b_kmajor) launch_pp<true>(epi, grid, p, (hipStream_t)stream);
    else launch_pp<false>(epi, grid, p, (hipStream_t)stream, diag);
    MH_LAUNCH_CHECK();
    return 0;
}
