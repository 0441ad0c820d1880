// Stream-K pieces shared by gemm_sk.hip (four-wave register-staged tiles) and gemm_sk_dma.hip (eight-wave 256 x 256 LDS-DMA tile):
// the work split over persistent workgroups, the hand-off of partial tiles, and the epilogues that work straight from the
// accumulator layout acc[j][i] (j = 16-column block 0..3, i = 16-row block 0..MT-1 of a (16 MT) x 64 wave tile; lane (lm = l & 15,
// g = l >> 4) owns row 16 i + lm, columns 16 j + 4 g .. + 3) -- no LDS staging, so a workgroup's operand ring stays live across a
// tile boundary.
//
// Work split.  A launch owns U = tiles x nk units (one K step of one tile); persistent workgroup w (logical id: the XCD-aware remap of
// blockIdx.x, so that the workgroups of one XCD own a contiguous run of tiles) takes units [w U / P, (w + 1) U / P): at most one tile's
// tail, whole tiles, and one tile's head.  A segment that does not contain its tile's LAST K step is "non-finishing": its fp32
// partial goes to the workgroup's slot of the workspace.  The workgroup that owns the last K step (the finisher) adds the partials
// of the workgroups before it in workgroup order -- a fixed order: results do not depend on timing -- and runs the epilogue.  Every
// workgroup has at most ONE non-finishing segment (its last one) and processes it FIRST, before anything it could wait for: a
// partial is published about one tile time before its finisher asks for it, and no workgroup waits before it has published.
// Hand-off (guide, "Workgroup dispatch, XCD placement & inter-workgroup visibility", first row of the measured table): every partial
// byte is stored and loaded `sc1` (16-byte buffer accesses), every storing wave waits vmcnt(0), a workgroup barrier, ONE lane's
// agent-scope flag store; the finisher's lane 0 polls the flags with sc1 loads, the other waves load behind the barrier it then
// joins; the finisher clears the flags it consumed, so a workspace is all zeros between launches.
#pragma once
#include "gemm_common.hpp"

namespace {

enum { SK_EPI_BF16 = 0, SK_EPI_F32 = 1 };
constexpr long SK_FLAG_BYTES = 4096;   // up to 1008 workgroups' flag words + the error word (last int)
constexpr int SK_SC1 = 16;             // cache-policy bit of the buffer instructions (sc1)

struct SkArgs {
    float* ws;      // P partial-tile slots (accumulator order)
    int* flags;     // P arrival words; zero between launches
    int P;          // persistent workgroups = gridDim.x
    int nk;         // K steps per tile
    int tiles;      // output tiles
    int* err;       // set to 1 if a flag wait ran into its bound (a lost producer: results are wrong, nothing hangs)
};

// This workgroup's share of the launch: segments in PROCESSING order (the non-finishing one first).
struct SkSplit {
    long U;
    int lw, nk, t0, nseg, k_first, k_last_end;
    bool rot, empty;
    __device__ __forceinline__ SkSplit(const SkArgs& sk, int bid) {
        lw = xcd_remap(bid, sk.P);
        nk = sk.nk;
        U = (long)sk.tiles * sk.nk;
        const long ub = (long)lw * U / sk.P, ue = (long)(lw + 1) * U / sk.P;
        empty = ub >= ue;
        t0 = (int)(ub / nk);
        const int t1 = empty ? t0 : (int)((ue - 1) / nk);
        nseg = t1 - t0 + 1;
        k_first = (int)(ub - (long)t0 * nk);
        k_last_end = (int)(ue - (long)t1 * nk);
        rot = k_last_end < nk;
    }
    __device__ __forceinline__ void segment(int q, int& tile, int& kb, int& ke) const {
        const int s = rot ? (q == 0 ? nseg - 1 : q - 1) : q;
        tile = t0 + s;
        kb = s == 0 ? k_first : 0;
        ke = s == nseg - 1 ? k_last_end : nk;
    }
    // first workgroup that owns a unit of `tile` (the contributors of a finisher are [first_owner(tile), lw))
    __device__ __forceinline__ int first_owner(const SkArgs& sk, int tile) const { return (int)((((long)tile * nk + 1) * sk.P - 1) / U); }
};

// Lane id without a live register: the work-item id VGPR need not survive the K loop (the eight-wave kernel spilled it and
// reloaded it from scratch -- behind a vmcnt(0) that drained the DMA ring -- in every K step)
__device__ __forceinline__ int sk_lane() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

__device__ __forceinline__ void sk_swap16(uint32_t& a, uint32_t& b) {
    const u32x2 r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    a = r[0]; b = r[1];
}

struct SkEpiDesc {
    __amdgpu_buffer_rsrc_t c, res, bias, ws;
};
template <int EPI>
__device__ __forceinline__ SkEpiDesc sk_epi_desc(const GemmParams& p, const SkArgs& sk, long slot_bytes) {
    SkEpiDesc d;
    d.c = __builtin_amdgcn_make_buffer_rsrc(p.C, (short)0, (int)(((long)(p.M - 1) * p.ldc + p.N) * (EPI == SK_EPI_F32 ? 4 : 2)), 0x00020000);
    d.res = __builtin_amdgcn_make_buffer_rsrc((void*)p.res, (short)0, EPI == SK_EPI_F32 ? (int)(((long)(p.M - 1) * p.ldr + p.N) * 4) : 0,
                                              0x00020000);
    d.bias = __builtin_amdgcn_make_buffer_rsrc((void*)p.bias, (short)0, p.bias ? p.N * 4 : 0, 0x00020000);
    const long ws_bytes = (long)sk.P * slot_bytes;
    d.ws = __builtin_amdgcn_make_buffer_rsrc((void*)sk.ws, (short)0, (int)(ws_bytes > 0x7fffffffL ? 0x7fffffffL : ws_bytes), 0x00020000);
    return d;
}

// Epilogue of the wave tile whose first row / column are (row0 = m0 + wm, col0 = n0 + wn).  Rows >= M are dropped / read as zero by
// the descriptors (the voffset carries the whole (row, column) offset): no exec-mask branches.
template <int MT, int EPI>
__device__ __forceinline__ void sk_epilogue(const GemmParams& p, const SkEpiDesc& d, const f32x4 (&acc)[4][MT], int row0, int col0) {
    int lane = sk_lane();
    asm volatile("" : "+v"(lane));     // opaque: the per-lane offsets below are computed HERE, not hoisted into (and kept live across) the K loop
    const int l = lane & 63, g = l >> 4, lm = l & 15;
    const int row = row0 + lm;
    if constexpr (EPI == SK_EPI_F32) {
        // 16-byte accesses, four lanes per 64-byte row segment.  The residual rows of a batch of row blocks are requested before its
        // first store (round 5: behind the stores the compiler keeps the next loads -- it cannot prove res and C distinct).
        if constexpr (MT >= 8) {
            // the eight-wave tile (256 registers per lane, 128 of them accumulators): one column block at a time, its residual rows
            // requested four row blocks ahead of their stores -- 4 + 16 temporaries
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int cofs = (col0 + 16 * j + 4 * g) * 4;
                const f32x4 b4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(d.bias, cofs, 0, 0));
#pragma unroll
                for (int i0 = 0; i0 < MT; i0 += 4) {
                    f32x4 add[4];
#pragma unroll
                    for (int ii = 0; ii < 4; ++ii)
                        add[ii] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(d.res, (row + 16 * (i0 + ii)) * p.ldr * 4 + cofs, 0, 0));
#pragma unroll
                    for (int ii = 0; ii < 4; ++ii) {
                        const f32x4 v = acc[j][i0 + ii] + (b4 + add[ii]);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), d.c, (row + 16 * (i0 + ii)) * p.ldc * 4 + cofs, 0, 0);
                    }
                }
            }
            return;
        }
        f32x4 bias4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            bias4[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(d.bias, (col0 + 16 * j + 4 * g) * 4, 0, 0));
        constexpr int BATCH = MT % 3 == 0 ? 3 : 4;
#pragma unroll
        for (int i0 = 0; i0 < MT; i0 += BATCH) {
            f32x4 add[BATCH][4];
#pragma unroll
            for (int ii = 0; ii < BATCH; ++ii)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    add[ii][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                        d.res, ((row + 16 * (i0 + ii)) * p.ldr + col0 + 16 * j + 4 * g) * 4, 0, 0));
#pragma unroll
            for (int ii = 0; ii < BATCH; ++ii)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 v = acc[j][i0 + ii] + (bias4[j] + add[ii][j]);   // (the order of gemm_common.hpp's fp32 epilogue)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), d.c,
                                                           ((row + 16 * (i0 + ii)) * p.ldc + col0 + 16 * j + 4 * g) * 4, 0, 0);
                }
        }
    } else {
        // two v_permlane16_swap turn a lane's 4 + 4 columns of two neighbouring 16-column blocks into 8 consecutive ones: 16-byte
        // stores, 64 contiguous bytes per row and instruction (as gemm_pp.hip)
        const int col_swp = col0 + 4 * g + 12 * (g & 1);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
                const f32x4 vx = acc[2 * jp][i], vy = acc[2 * jp + 1][i];
                uint32_t x0 = pack_bf2(vx[0], vx[1]), x1 = pack_bf2(vx[2], vx[3]);
                uint32_t y0 = pack_bf2(vy[0], vy[1]), y1 = pack_bf2(vy[2], vy[3]);
                sk_swap16(x0, y0);
                sk_swap16(x1, y1);
                __builtin_amdgcn_raw_buffer_store_b128((u32x4){x0, x1, y0, y1}, d.c, ((row + 16 * i) * p.ldc + col_swp + 32 * jp) * 2, 0, 0);
            }
    }
}

// The partial tile of a workgroup of NTHR threads in accumulator order (4 KiB per wave instruction): slot bytes = 4 MT NTHR 16.
template <int MT, int NTHR>
__device__ __forceinline__ void sk_store_partial(const SkEpiDesc& d, const f32x4 (&acc)[4][MT], int slot, int wave) {
    int lane = sk_lane();
    asm volatile("" : "+v"(lane));
    const int lane_off = (wave * 64 + lane) * 16;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[j][i]), d.ws, lane_off + (j * MT + i) * (NTHR * 16),
                                                   slot * (4 * MT * NTHR * 16), SK_SC1);
}
// acc += the partials of slots [first, last), summed in slot order.  The loop over the contributors runs INSIDE each column block,
// on MT temporaries: written as a loop around `acc += partial` the 4 MT accumulator registers became loop phis inside a
// conditional and the eight-wave kernel (256 registers per lane) spilled ~90 of them -- into the K loop's fragment addresses.
template <int MT, int NTHR>
__device__ __forceinline__ void sk_add_partials(const SkEpiDesc& d, f32x4 (&acc)[4][MT], int first, int last, int wave) {
    int lane = sk_lane();
    asm volatile("" : "+v"(lane));
    const int lane_off = (wave * 64 + lane) * 16;
    constexpr int CH = MT >= 8 ? 4 : MT;      // row blocks per pass (the eight-wave tile has 256 registers per lane: 16 + 16 temporaries)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i0 = 0; i0 < MT; i0 += CH) {
            f32x4 t[CH];
#pragma unroll
            for (int i = 0; i < CH; ++i) t[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            for (int c = first; c < last; ++c) {
#pragma unroll
                for (int i = 0; i < CH; ++i)
                    t[i] += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(d.ws, lane_off + (j * MT + i0 + i) * (NTHR * 16),
                                                                                            c * (4 * MT * NTHR * 16), SK_SC1));
            }
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                acc[j][i0 + i] += t[i];
                asm volatile("" : "+v"(acc[j][i0 + i]));   // the add happens HERE: sunk to the epilogue, all 4 MT sums of t stayed live beside acc
            }
        }
}
// The finisher's lane 0: wait for the contributors' flags [wf, lw) and clear them (bounded spin: a lost producer sets *err)
__device__ __forceinline__ void sk_wait_flags(const SkArgs& sk, int wf, int lw) {
    for (int c = wf; c < lw; ++c) {
        int spins = 0;
        while (__hip_atomic_load(sk.flags + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
            __builtin_amdgcn_s_sleep(4);
            if (++spins > (1 << 22)) { __hip_atomic_store(sk.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
        __hip_atomic_store(sk.flags + c, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

}  // namespace

// gemm_sk_dma.hip: the eight-wave 256 x 256 stream-K tile; -2 = not served
int gemm_sk_dma_launch(int layout, int epi, GemmParams& p, void* workspace, int grid, void* stream);
