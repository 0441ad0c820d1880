// Persistent grouped bf16 MFMA GEMM for gfx950: ONE launch over the tiles of several independent NT / NN problems (same
// layout, each with its own operands, shape and fused epilogue), e.g. layer-op l of every modality group's encoder
// (reference call sites maestro/ssl/mae.py:135-141,168-174: the per-group Transformers run the same op on different rows).
//
// Why: the per-problem launches of these shapes (M = 3200 .. 12800 token rows, N = 512 .. 3072) quantise badly on 256 CUs
// (96 .. 540 tiles of 256 x 256) and pay a fixed ~6 us per tile -- pipeline fill, exposed epilogue, workgroup turnover --
// which is 20-30 % of a K = 512 / 768 tile.  Here
//   * one 512-thread workgroup per CU stays resident and walks a host-built work list (worker w runs items w, w + G, ...);
//     items are sorted by decreasing cost, so every worker gets the same number of full 256 x 256 tiles and the remainder
//     is dealt out as half / quarter tiles (128 x 256, 256 x 128, 128 x 128: four tile shapes in one kernel) instead of one
//     more full round that leaves most CUs idle -- no split-K, no inter-workgroup communication, no atomics;
//   * the LDS-DMA ring never drains between two tiles of the same shape: the last S-1 K steps of a tile already stream
//     the first S-1 K steps of the worker's next tile, so the epilogue runs with the next main loop's operands in flight
//     and the next tile starts without a pipeline fill;
//   * the epilogue stages through an LDS region of its own (16-row passes), so it never waits for the ring.
// Main loop, operand images, swizzles and the fused epilogues are those of gemm_dma.hip (gemm_ring.hpp, gemm_common.hpp);
// results are bit-identical to mh_gemm_bf16 per problem (same K order per output element, same epilogue arithmetic).
#include "gemm_ring.hpp"

namespace {

constexpr int PS = 3;                          // ring stages (two K steps in flight)
constexpr int SLOT_BYTES = 32768;              // one ring slot = the largest stage (256 x 256 tile: A 16 KiB + B 16 KiB)
constexpr int NWAVES = 8, NTHREADS = 512;
constexpr int STAGE_FLOATS = 16 * 68;          // per-wave epilogue staging: 16-row passes, 68-float pitch
constexpr int LDS_TOTAL = PS * SLOT_BYTES + NWAVES * STAGE_FLOATS * 4;   // 98304 + 34816 = 133120 B

typedef Tile<2, 4, PS, 8> P256;       // 256 x 256, waves 2 (M) x 4 (N), 128 x 64 per wave
typedef Tile<2, 4, PS, 4> P128x256;   // 128 x 256,                       64 x 64 per wave
typedef Tile<4, 2, PS, 4> P256x128;   // 256 x 128, waves 4 x 2,          64 x 64 per wave
typedef Tile<4, 2, PS, 2> P128;       // 128 x 128,                       32 x 64 per wave
static_assert(P256::NT == NTHREADS && P128x256::NT == NTHREADS && P256x128::NT == NTHREADS && P128::NT == NTHREADS, "8 waves");
static_assert(P256::STAGE_BYTES <= SLOT_BYTES, "ring slot size");

template <class T> struct TileId;
template <> struct TileId<P256> { static constexpr int v = MH_GTILE_256; };
template <> struct TileId<P128x256> { static constexpr int v = MH_GTILE_128x256; };
template <> struct TileId<P256x128> { static constexpr int v = MH_GTILE_256x128; };
template <> struct TileId<P128> { static constexpr int v = MH_GTILE_128; };

// what the DMA issue of one tile needs: operand descriptors, per-lane source offsets of this wave's pieces, K stepping
template <class T>
struct Cursor {
    __amdgpu_buffer_rsrc_t ra, rb;
    int va[T::PA], vb[T::PB];
    int a_step, b_step, nk;
};

__device__ __forceinline__ void load_problem(const MhGemmProblem* __restrict__ probs, int idx, GemmParams& p) {
    const MhGemmProblem q = probs[idx];      // uniform index: scalar loads
    p.A = (const bf16_t*)q.A; p.B = (const bf16_t*)q.B; p.C = q.C;
    p.bias = q.bias; p.res = q.res; p.aux_in = (const bf16_t*)q.aux_in; p.aux_out = (bf16_t*)q.aux_out; p.colsum = q.colsum;
    p.M = q.M; p.N = q.N; p.K = q.K; p.lda = q.lda; p.ldb = q.ldb; p.ldc = q.ldc; p.ldr = q.ldr; p.ldaux = q.ldaux;
    p.flags = q.flags; p.tiles_m = 0; p.tiles_n = 0; p.k_per_split = q.K; p.fast = 1;
    p.a_bytes = q.a_bytes; p.b_bytes = q.b_bytes;
}

template <class T, bool B_KMAJOR>
__device__ __forceinline__ void setup_cursor(const GemmParams& p, int m0, int n0, int w, int l, Cursor<T>& c) {
    c.ra = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, (short)0, (int)p.a_bytes, 0x00020000);
    c.rb = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, (short)0, (int)p.b_bytes, 0x00020000);
    piece_offsets<false, T::BM, T::NW, T::PA>(p.lda, m0, w, l, c.va);
    piece_offsets<B_KMAJOR, T::BN, T::NW, T::PB>(p.ldb, n0, w, l, c.vb);
    c.a_step = BK * 2;
    c.b_step = B_KMAJOR ? BK * p.ldb * 2 : BK * 2;
    c.nk = (p.K + BK - 1) / BK;
}

__device__ __forceinline__ int ring_next(int s) { return s + 1 == PS ? 0 : s + 1; }

// One run of consecutive work-list items of tile shape T for this worker.  On return `i` / `item` name the worker's next
// item (of another shape) or `i >= n_items`.
// ABL (diagnostic builds only, scripts/bench_persist_ablate.py): 0 = the kernel; 1 = DMA stream only (no fragment reads, no
// MFMAs); 2 = fragment reads + MFMAs only (no DMA: stale LDS); 3 = MFMAs only (no DMA, no reads); outputs are garbage for != 0.
template <class T, bool B_KMAJOR, int ABL>
__device__ __forceinline__ void run_shape(const MhGemmProblem* __restrict__ probs, const uint2* __restrict__ items, int n_items,
                                          int stride, int& i, uint2& item, int& slot, unsigned char* smem) {
    constexpr int MT = T::MT, PA = T::PA, PB = T::PB;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63;
    const int wm = (w / T::WN) * (16 * MT), wn = (w % T::WN) * 64;
    float* st = reinterpret_cast<float*>(smem + PS * SLOT_BYTES) + w * STAGE_FLOATS;

    GemmParams p;
    Cursor<T> cur;
    load_problem(probs, item.x & 0xffff, p);
    int m0 = (int)(item.y & 0xffff) * 64, n0 = (int)(item.y >> 16) * 64;
    setup_cursor<T, B_KMAJOR>(p, m0, n0, w, l, cur);
    // DMA the operand tiles of K step t of the tile behind `c` into ring slot `sl` (PA + PB one-KiB pieces per wave).
    // (A lambda, not a function template: the host pass of hipcc 7.2 cannot instantiate a template around this builtin.)
    auto issue_step = [&](const Cursor<T>& c, int t, int sl) {
        unsigned char* base = smem + sl * SLOT_BYTES;
#pragma unroll
        for (int h = 0; h < PA; ++h)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(c.ra, (lds_void*)(base + (w + T::NW * h) * 1024), 16, c.va[h], t * c.a_step, 0, 0);
#pragma unroll
        for (int h = 0; h < PB; ++h)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(c.rb, (lds_void*)(base + T::A_BYTES + (w + T::NW * h) * 1024), 16, c.vb[h],
                                                     t * c.b_step, 0, 0);
    };
    // Fresh pipeline fill (first item of the worker, or the shape changed).  `slot` = ring slot of the K step about to be
    // consumed; it keeps rotating across tiles and shapes.  A wave that gets here has passed the barrier of the previous
    // tile's LAST K step, i.e. every wave has finished reading the steps before that one: the two slots written here were
    // read three and two steps ago and are free, while the slot of the last step (possibly still being read by a slower
    // wave) is only refilled behind the barrier of this tile's first step.
    if constexpr (ABL < 2) {
        issue_step(cur, 0, slot);
        if (cur.nk > 1) issue_step(cur, 1, ring_next(slot));
    }

    while (true) {
        const int ni = i + stride;
        const bool has_next = ni < n_items;
        uint2 nitem = {0xffffffffu, 0u};
        if (has_next) nitem = items[ni];
        const bool same = has_next && (int)((nitem.x >> 16) & 3) == TileId<T>::v;
        GemmParams pn;
        Cursor<T> nxt;
        int nm0 = 0, nn0 = 0;
        if (same) {
            load_problem(probs, nitem.x & 0xffff, pn);
            nm0 = (int)(nitem.y & 0xffff) * 64; nn0 = (int)(nitem.y >> 16) * 64;
            setup_cursor<T, B_KMAJOR>(pn, nm0, nn0, w, l, nxt);
        }

        f32x4 acc[4][MT];   // [j (n tile)][i (m tile)]
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int ii = 0; ii < MT; ++ii) acc[j][ii] = (f32x4){0.f, 0.f, 0.f, 0.f};

        const int nk = cur.nk;
        for (int t = 0; t < nk; ++t) {
            // my pieces of step t have landed once at most the ONE younger step's DMA instructions are still pending (S = 3);
            // a younger step exists unless this is the worker's last step before a pipeline drain.  (After an epilogue its
            // stores are younger still: the count below then over-waits for them, never under-waits.)
            const bool younger = (t + 1 < nk) || same;
            if (younger) wait_vmcnt<PA + PB>(); else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();   // everybody's pieces of step t are in LDS; the previous step has been read by everybody
            const unsigned char* ta = smem + slot * SLOT_BYTES;
            const unsigned char* tb = ta + T::A_BYTES;
            bf16x8 fa[MT], fb[4];
            if constexpr (ABL == 0 || ABL == 2) {
#pragma unroll
                for (int j = 0; j < 4; ++j) fb[j] = read_frag<B_KMAJOR, T::BN>(tb, wn + 16 * j);
#pragma unroll
                for (int ii = 0; ii < MT; ++ii) fa[ii] = read_frag<false, T::BM>(ta, wm + 16 * ii);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) fb[j] = __builtin_bit_cast(bf16x8, (u32x4){(uint32_t)t, (uint32_t)l, 0x3f803f80u, 0u});
#pragma unroll
                for (int ii = 0; ii < MT; ++ii) fa[ii] = __builtin_bit_cast(bf16x8, (u32x4){0x3f803f80u, (uint32_t)ii, (uint32_t)t, 0u});
            }
            // (1) all fragment reads, (2) the DMA pieces that refill the slot vacated by the previous step -- with the step two
            // ahead of this one, which belongs to the NEXT tile of the run once this tile's K range is exhausted --, (3) MFMAs
            __builtin_amdgcn_sched_barrier(0);
            const int fill = slot == 0 ? PS - 1 : slot - 1;       // = (slot + PS - 1) % PS
            if constexpr (ABL < 2) {
                if (t + PS - 1 < nk) issue_step(cur, t + PS - 1, fill);
                else if (same && t + PS - 1 - nk < nxt.nk) issue_step(nxt, t + PS - 1 - nk, fill);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (ABL != 1) {
#pragma unroll
                for (int ii = 0; ii < MT; ++ii)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[j][ii] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[ii], acc[j][ii], 0, 0, 0);
            }
            slot = ring_next(slot);
        }
        gemm_epilogue_store<MT, 16>(p, acc, st, m0 + wm, n0 + wn);

        i = ni;
        item = nitem;
        if (!same) return;
        p = pn; cur = nxt; m0 = nm0; n0 = nn0;
    }
}

template <bool B_KMAJOR, int ABL>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_persist_kernel(const MhGemmProblem* __restrict__ probs,
                                                                   const uint2* __restrict__ items, int n_items) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_TOTAL];   // the ONLY LDS object: ring + epilogue staging
    // workers that share an XCD (blockIdx % 8, hardware placement; speed only) take neighbouring items of every round
    const int per_xcd = gridDim.x >> 3;
    const int worker = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    int i = worker, slot = 0;
    if (i >= n_items) return;
    uint2 item = items[i];
    while (i < n_items) {
        switch ((item.x >> 16) & 3) {
            case MH_GTILE_256: run_shape<P256, B_KMAJOR, ABL>(probs, items, n_items, gridDim.x, i, item, slot, smem); break;
            case MH_GTILE_128x256: run_shape<P128x256, B_KMAJOR, ABL>(probs, items, n_items, gridDim.x, i, item, slot, smem); break;
            case MH_GTILE_256x128: run_shape<P256x128, B_KMAJOR, ABL>(probs, items, n_items, gridDim.x, i, item, slot, smem); break;
            default: run_shape<P128, B_KMAJOR, ABL>(probs, items, n_items, gridDim.x, i, item, slot, smem); break;
        }
    }
}

}  // namespace

extern "C" int mh_gemm_grouped_check(int layout, const MhGemmProblem* problems_host, int n_problems) {
    MH_CHECK_ARG(layout == 0 || layout == 1, "mh_gemm_grouped: layout %d (NT = 0 and NN = 1 only)", layout);
    MH_CHECK_ARG(problems_host && n_problems > 0 && n_problems < 65536, "mh_gemm_grouped: bad problem table");
    for (int i = 0; i < n_problems; ++i) {
        const MhGemmProblem& q = problems_host[i];
        const int fl = q.flags;
        MH_CHECK_ARG(q.A && q.B && q.C && q.M > 0 && q.N > 0 && q.K >= 2 * BK, "mh_gemm_grouped[%d]: empty problem or K < 64", i);
        MH_CHECK_ARG(q.K % BK == 0, "mh_gemm_grouped[%d]: K must be a multiple of 32 (%d)", i, q.K);
        MH_CHECK_ARG(q.lda % 8 == 0 && q.ldb % 8 == 0 && q.N % 8 == 0 && q.ldc % 8 == 0, "mh_gemm_grouped[%d]: lda, ldb, N, ldc %% 8", i);
        MH_CHECK_ARG(((uintptr_t)q.A | (uintptr_t)q.B | (uintptr_t)q.C) % 16 == 0, "mh_gemm_grouped[%d]: bases must be 16-B aligned", i);
        MH_CHECK_ARG(!(fl & MH_GEMM_ATOMIC), "mh_gemm_grouped[%d]: no atomic accumulate (no split-K here)", i);
        MH_CHECK_ARG((fl & MH_GEMM_OUT_F32) || !(fl & MH_GEMM_RESIDUAL), "mh_gemm_grouped[%d]: residual epilogue needs f32 output", i);
        MH_CHECK_ARG(!(fl & MH_GEMM_OUT_F32) || !(fl & (MH_GEMM_GELU | MH_GEMM_DGELU | MH_GEMM_MULAUX | MH_GEMM_COLSUM)),
                     "mh_gemm_grouped[%d]: GELU / aux / colsum epilogues need bf16 output", i);
        MH_CHECK_ARG(!(fl & MH_GEMM_BIAS) || q.bias, "mh_gemm_grouped[%d]: bias flag without pointer", i);
        MH_CHECK_ARG(!(fl & MH_GEMM_RESIDUAL) || (q.res && q.ldr % 4 == 0), "mh_gemm_grouped[%d]: residual needs pointer, ldr %% 4", i);
        MH_CHECK_ARG(!(fl & (MH_GEMM_DGELU | MH_GEMM_MULAUX)) || (q.aux_in && q.ldaux % 8 == 0), "mh_gemm_grouped[%d]: aux_in / ldaux", i);
        MH_CHECK_ARG(!(fl & MH_GEMM_AUX_DGELU) || ((fl & MH_GEMM_GELU) && q.aux_out), "mh_gemm_grouped[%d]: aux_dgelu needs GELU + aux_out", i);
        MH_CHECK_ARG(!(fl & MH_GEMM_GELU) || !q.aux_out || q.ldaux % 8 == 0, "mh_gemm_grouped[%d]: ldaux %% 8", i);
        MH_CHECK_ARG(!(fl & MH_GEMM_AUX_U8) || ((fl & (MH_GEMM_AUX_DGELU | MH_GEMM_MULAUX)) && !(fl & MH_GEMM_DGELU)),
                     "mh_gemm_grouped[%d]: MH_GEMM_AUX_U8 applies to the saved GELU derivative only", i);
        MH_CHECK_ARG(!((fl & MH_GEMM_DGELU) && (fl & MH_GEMM_MULAUX)), "mh_gemm_grouped[%d]: dgelu and mulaux exclude each other", i);
        MH_CHECK_ARG(!(fl & MH_GEMM_COLSUM) || q.colsum, "mh_gemm_grouped[%d]: colsum flag without pointer", i);
        // operand extents as the buffer descriptors will see them (rows beyond M / N and K rows beyond K read as zero)
        const long a_ext = ((long)(q.M - 1) * q.lda + q.K) * 2;
        const long b_ext = layout == 1 ? ((long)(q.K - 1) * q.ldb + q.N) * 2 : ((long)(q.N - 1) * q.ldb + q.K) * 2;
        const long a_reach = (long)(ceil_div(q.M, 256) * 256) * q.lda * 2;
        const long b_reach = layout == 1 ? (long)q.K * q.ldb * 2 : (long)(ceil_div(q.N, 256) * 256) * q.ldb * 2;
        MH_CHECK_ARG(a_reach + 65536 < (1L << 31) && b_reach + 65536 < (1L << 31), "mh_gemm_grouped[%d]: operand beyond the 2 GiB descriptor range", i);
        MH_CHECK_ARG(q.a_bytes == (unsigned)a_ext && q.b_bytes == (unsigned)b_ext,
                     "mh_gemm_grouped[%d]: a_bytes / b_bytes must be the operand extents (%ld, %ld)", i, a_ext, b_ext);
    }
    return 0;
}

extern "C" int mh_gemm_grouped(int layout, const MhGemmProblem* problems_device, int n_problems, const uint32_t* items_device,
                               int n_items, int n_workers, void* stream) {
    MH_CHECK_ARG(layout == 0 || layout == 1, "mh_gemm_grouped: layout %d (NT = 0 and NN = 1 only)", layout);
    MH_CHECK_ARG(problems_device && items_device && n_problems > 0 && n_problems < 65536 && n_items > 0,
                 "mh_gemm_grouped: bad arguments");
    MH_CHECK_ARG(n_workers >= 8 && n_workers % 8 == 0 && n_workers <= 4096, "mh_gemm_grouped: n_workers must be a multiple of 8 (%d)", n_workers);
    MH_CHECK_ARG((uintptr_t)items_device % 8 == 0, "mh_gemm_grouped: items must be 8-byte aligned");
    const uint2* items = reinterpret_cast<const uint2*>(items_device);
    if (layout == 0) hipLaunchKernelGGL((gemm_persist_kernel<false, 0>), dim3(n_workers), dim3(NTHREADS), 0, (hipStream_t)stream, problems_device, items, n_items);
    else hipLaunchKernelGGL((gemm_persist_kernel<true, 0>), dim3(n_workers), dim3(NTHREADS), 0, (hipStream_t)stream, problems_device, items, n_items);
    MH_LAUNCH_CHECK();
    return 0;
}

// Diagnostic builds of the NT kernel (not part of the ABI in include/maestro_hip.h; outputs are garbage): see ABL above.
extern "C" int mh_gemm_grouped_ablate(int ablate, const MhGemmProblem* problems_device, const uint32_t* items_device, int n_items,
                                      int n_workers, void* stream) {
    const uint2* items = reinterpret_cast<const uint2*>(items_device);
    dim3 g(n_workers), b(NTHREADS);
    hipStream_t s = (hipStream_t)stream;
    switch (ablate) {
        case 1: hipLaunchKernelGGL((gemm_persist_kernel<false, 1>), g, b, 0, s, problems_device, items, n_items); break;
        case 2: hipLaunchKernelGGL((gemm_persist_kernel<false, 2>), g, b, 0, s, problems_device, items, n_items); break;
        case 3: hipLaunchKernelGGL((gemm_persist_kernel<false, 3>), g, b, 0, s, problems_device, items, n_items); break;
        default: hipLaunchKernelGGL((gemm_persist_kernel<false, 0>), g, b, 0, s, problems_device, items, n_items); break;
    }
    MH_LAUNCH_CHECK();
    return 0;
}
