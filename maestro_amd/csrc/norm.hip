// LayerNorm forward/backward for the fp32 residual stream (see include/maestro_hip.h).
// One wave (64 lanes) per row, row held in registers as float4 (dim <= 2048), two-pass statistics, wavefront
// shuffles for the reductions.  Rows are addressed through (sample, j) maps so that "split / concat of group
// sequences" (reference mim.py:408-423) is done by addressing instead of copies:
//     row(b, j) = b * L + off + j,  j < n.
#include "gemm_common.hpp"

namespace {

struct RowMap { int L, off; };
__device__ __forceinline__ size_t map_row(RowMap m, int b, int j) { return (size_t)b * m.L + m.off + j; }

// NV = float4 per lane (dim <= 256*NV); registers are sized exactly for the row width (no spills).
template <int NV>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, RowMap xm, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, void* __restrict__ y, RowMap ym,
                                                     int y_is_f32, float* __restrict__ mean, float* __restrict__ rstd,
                                                     int B, int n, int dim, float eps, uint8_t* __restrict__ y8,
                                                     const float* __restrict__ y8_scale, float* __restrict__ y8_amax) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= B * n) return;
    const int b = row / n, j = row - b * n;
    const float* xr = x + map_row(xm, b, j) * dim;
    const int nv = dim >> 2;
    f32x4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        v[i] = (f32x4){0, 0, 0, 0};
        if (c < nv) v[i] = *reinterpret_cast<const f32x4*>(xr + 4 * c);
        s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    const float mu = wave_sum(s) / dim;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (lane + 64 * i < nv) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mu; q += d * d; }
        }
    }
    const float rs = rsqrtf(wave_sum(q) / dim + eps);
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
    const size_t yrow = map_row(ym, b, j) * dim;
    const float s8 = y8 ? *y8_scale : 0.f;     // fp8 path: an e4m3 copy of the output (the next GEMM's A operand) + its absmax
    float amax8 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + 4 * c);
            const f32x4 bt = *reinterpret_cast<const f32x4*>(beta + 4 * c);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mu) * rs * g[e] + bt[e];
            if (y_is_f32) {
                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(y) + yrow + 4 * c) = o;
            } else {
                u32x2 pk = {pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3])};
                *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(y) + yrow + 4 * c) = pk;
            }
            if (y8) {
#pragma unroll
                for (int e = 0; e < 4; ++e) amax8 = amax_fold(amax8, o[e]);
                *reinterpret_cast<uint32_t*>(y8 + yrow + 4 * c) = pack_e4m3x4(o, s8);
            }
        }
    }
    if (y8 && y8_amax) {
        amax8 = wave_amax(amax8);
        if (lane == 0 && amax_nonzero(amax8)) atomic_max_pos(y8_amax, amax8);
    }
}

// Straight-line forward for the step's own case (dim == 256 * NV: every lane holds NV float4; bf16 output, optional e4m3 copy).
// Same operations in the same order as ln_fwd_kernel (results equal up to the compiler's fma contraction: a few fp32 ulp,
// tests/test_kernels_gpu.py), but no lane predicates and no output-type branches, so that the code is ONE scheduling region.  In the generic kernel's ISA (gfx950, -O3) every predicated chunk
// ends in `s_waitcnt vmcnt(0)`: the row's NV loads went out as NV dependent round trips, gamma / beta were fetched after
// the two reductions, chunk by chunk, and each chunk's wait also covered the previous chunk's STORES -- 7-8 memory round
// trips per row, one after the other.  Here the row, gamma and beta are requested back to back before anything is
// waited for, and the stores leave together at the end.
#ifndef MH_LN_FAST
#define MH_LN_FAST 1      // -DMH_LN_FAST=0 (MH_BUILD_FLAGS) builds the library without the straight-line forms (A/B aid)
#endif
#ifndef MH_LN_DPP
#define MH_LN_DPP 1
#endif
// Wave sum of the straight-line kernels (all 64 lanes active there).  MH_LN_DPP = 1 (default): each 16-lane row all-reduces by
// four DPP rotations (row_ror:8, 4, 2, 1: one v_add_f32_dpp each, no LDS crossbar round trip), then the four row sums meet
// through v_readlane.  With the loads no longer serialized the twelve ds_bpermute round trips per row show: forward -4 ... -7 %
// per launch, backward -1 ... -9 % (profiles/r03_ln_straightline.txt); 0 keeps the shuffle butterfly of wave_sum (the sums then
// differ from the generic kernels' in their last bits only by fma contraction; with 1 also by the order of the 64 partial sums).
__device__ __forceinline__ float ln_wave_sum(float v) {
#if MH_LN_DPP
#define MH_ROR(n) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + (n), 0xf, 0xf, false))
    v += MH_ROR(8); v += MH_ROR(4); v += MH_ROR(2); v += MH_ROR(1);
#undef MH_ROR
    const int iv = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
    return (r0 + r1) + (r2 + r3);
#else
    return wave_sum(v);
#endif
}
// FP8 = true (mh_layernorm_fwd_fp8): the same kernel also writes the e4m3 copy of the output (the next GEMM's A operand) and
// folds |y| into the tensor's amax row; the bf16 output is the SAME instruction sequence, hence bit-identical to FP8 = false.
template <int NV, bool FP8>
__global__ __launch_bounds__(256) void ln_fwd_fast_kernel(const float* __restrict__ x, RowMap xm, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, bf16_t* __restrict__ y, RowMap ym,
                                                          float* __restrict__ mean, float* __restrict__ rstd, int B, int n,
                                                          float eps, uint8_t* __restrict__ y8, const float* __restrict__ y8_scale,
                                                          float* __restrict__ y8_amax) {
    constexpr int dim = 256 * NV;
    const int row = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= B * n) return;
    const int b = row / n, j = row - b * n;
    const float* xr = x + map_row(xm, b, j) * dim;
    f32x4 v[NV], g[NV], bt[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = *reinterpret_cast<const f32x4*>(xr + 4 * (lane + 64 * i));
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        g[i] = *reinterpret_cast<const f32x4*>(gamma + 4 * (lane + 64 * i));
        bt[i] = *reinterpret_cast<const f32x4*>(beta + 4 * (lane + 64 * i));
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    const float mu = ln_wave_sum(s) / dim;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mu; q += d * d; }
    }
    const float rs = rsqrtf(ln_wave_sum(q) / dim + eps);
    const size_t yrow = map_row(ym, b, j) * dim;
    bf16_t* yr = y + yrow;
    float s8 = 0.f;
    if constexpr (FP8) s8 = *y8_scale;
    float amax8 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mu) * rs * g[i][e] + bt[i][e];
        const u32x2 pk = {pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3])};
        *reinterpret_cast<u32x2*>(yr + 4 * (lane + 64 * i)) = pk;
        if constexpr (FP8) {
#pragma unroll
            for (int e = 0; e < 4; ++e) amax8 = amax_fold(amax8, o[e]);
            *reinterpret_cast<uint32_t*>(y8 + yrow + 4 * (lane + 64 * i)) = pack_e4m3x4(o, s8);
        }
    }
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
    if constexpr (FP8) {
        if (y8_amax) {
            amax8 = wave_amax(amax8);
            if (lane == 0 && amax_nonzero(amax8)) atomic_max_pos(y8_amax, amax8);
        }
    }
}

// Backward. Each wave walks ROWS_PER_WAVE rows keeping per-column partials of dgamma, dbeta and colsum(dx) in
// registers; the block reduces them through LDS and writes ONE partial row [3*dim] to the workspace (plain stores);
// ln_bwd_reduce_kernel then sums the partial rows (few atomics per column, no same-address storm).
#ifndef LN_ROWS
#define LN_ROWS 4
#endif
constexpr int ROWS_PER_WAVE = LN_ROWS;  // 4: 16 rows per block, >= 2 blocks per CU at M = 8192 (8 was bandwidth-starved: 256 blocks)

template <int NV>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const void* __restrict__ dy, RowMap dym, int dy_is_f32,
                                                     const float* __restrict__ x, RowMap xm,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const float* __restrict__ dres,
                                                     float* __restrict__ dx, bf16_t* __restrict__ dx_bf16,
                                                     float* __restrict__ partial, int B, int n, int dim) {
    extern __shared__ __attribute__((aligned(16))) float red[];  // [4 waves][3][dim]
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nv = dim >> 2;
    f32x4 gsum[NV], bsum[NV], csum[NV], gm[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        gsum[i] = (f32x4){0, 0, 0, 0}; bsum[i] = (f32x4){0, 0, 0, 0}; csum[i] = (f32x4){0, 0, 0, 0};
        const int c = lane + 64 * i;
        gm[i] = c < nv ? *reinterpret_cast<const f32x4*>(gamma + 4 * c) : (f32x4){0, 0, 0, 0};
    }
    const int row0 = (blockIdx.x * 4 + w) * ROWS_PER_WAVE;
    const int rows = B * n;
    // Software pipeline over the wave's rows: every operand of row r+1 (x, dy and the residual gradient) is requested
    // before row r is reduced, so a wave keeps two rows of loads in flight instead of two dependent round trips per row.
    f32x4 xv_n[NV], rs_n[NV];
    u32x4 dy_n[NV];      // bf16: .xy hold 4 values; f32: all four lanes
    float mu_n = 0.f, rstd_n = 0.f;
    size_t xrow_n = 0;
    auto fetch = [&](int row) {
        const int b = row / n, j = row - b * n;
        xrow_n = map_row(xm, b, j) * dim;
        const size_t dyrow = map_row(dym, b, j) * dim;
        mu_n = mean[row]; rstd_n = rstd[row];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                xv_n[i] = *reinterpret_cast<const f32x4*>(x + xrow_n + 4 * c);
                if (dy_is_f32) {
                    dy_n[i] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const float*>(dy) + dyrow + 4 * c);
                } else {
                    const u32x2 pk = *reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16_t*>(dy) + dyrow + 4 * c);
                    dy_n[i] = (u32x4){pk[0], pk[1], 0u, 0u};
                }
                rs_n[i] = dres ? *reinterpret_cast<const f32x4*>(dres + xrow_n + 4 * c) : (f32x4){0, 0, 0, 0};
            }
        }
    };
    if (row0 < rows) fetch(row0);
    for (int rr = 0; rr < ROWS_PER_WAVE; ++rr) {
        const int row = row0 + rr;
        if (row >= rows) break;
        const size_t xrow = xrow_n;
        const float mu = mu_n, rs = rstd_n;
        f32x4 xh[NV], dz[NV], rsd[NV];
        u32x4 dyv[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) { xh[i] = xv_n[i]; dyv[i] = dy_n[i]; rsd[i] = rs_n[i]; }
        if (rr + 1 < ROWS_PER_WAVE && row + 1 < rows) fetch(row + 1);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            const f32x4 xv = xh[i];
            xh[i] = (f32x4){0, 0, 0, 0}; dz[i] = (f32x4){0, 0, 0, 0};
            if (c < nv) {
                f32x4 d;
                if (dy_is_f32) {
                    d = __builtin_bit_cast(f32x4, dyv[i]);
                } else {
                    d = (f32x4){__uint_as_float(dyv[i][0] << 16), __uint_as_float(dyv[i][0] & 0xffff0000u),
                                __uint_as_float(dyv[i][1] << 16), __uint_as_float(dyv[i][1] & 0xffff0000u)};
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    xh[i][e] = (xv[e] - mu) * rs;
                    dz[i][e] = d[e] * gm[i][e];
                    s1 += dz[i][e];
                    s2 += dz[i][e] * xh[i][e];
                    gsum[i][e] += d[e] * xh[i][e];
                    bsum[i][e] += d[e];
                }
            }
        }
        const float c1 = wave_sum(s1) / dim, c2 = wave_sum(s2) / dim;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = rs * (dz[i][e] - c1 - xh[i][e] * c2);
                o += rsd[i];
                csum[i] += o;
                *reinterpret_cast<f32x4*>(dx + xrow + 4 * c) = o;
                if (dx_bf16) {
                    u32x2 pk = {pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3])};
                    *reinterpret_cast<u32x2*>(dx_bf16 + xrow + 4 * c) = pk;
                }
            }
        }
    }
    if (!partial) return;
    float* rg = red + (size_t)w * 3 * dim;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            *reinterpret_cast<f32x4*>(rg + 4 * c) = gsum[i];
            *reinterpret_cast<f32x4*>(rg + dim + 4 * c) = bsum[i];
            *reinterpret_cast<f32x4*>(rg + 2 * dim + 4 * c) = csum[i];
        }
    }
    __syncthreads();
    float* prow = partial + (size_t)blockIdx.x * 3 * dim;
    for (int c = threadIdx.x; c < 3 * dim; c += 256)
        prow[c] = red[c] + red[3 * dim + c] + red[6 * dim + c] + red[9 * dim + c];
}

// Straight-line backward for the transformer blocks' own case: dim == 256 * NV, bf16 dy, a residual gradient to add, a bf16
// copy of dx to write, rows % ROWS_PER_WAVE == 0 (a wave's rows exist together).  Same operations, same order, same
// partial-row layout as ln_bwd_kernel (results equal up to fma contraction / packing: a few fp32 ulp).  Why it exists: in the generic kernel the run-time cases
// (dy type, dres / dx_bf16 present, lane < nv) are branches INSIDE the software pipeline; its ISA waits `vmcnt(0)` after
// each chunk's dy load (the bf16 words are moved into the f32-sized registers of the other case) and again at the loop
// head (store counts are not static across the branches), so the "prefetch" of the next row was NV + 1 dependent round
// trips and also waited for the previous row's stores: ~5 us per row and wave, whatever the bandwidth.  Here the wave's
// rows are unrolled, the operands of DEPTH rows are in flight before the first one is reduced, the waits carry exact
// counts, and the sched_barriers pin the issue order (loads of row r + DEPTH, then the arithmetic and stores of row r).
template <int NV, int DEPTH>
__global__ __launch_bounds__(256) void ln_bwd_fast_kernel(const bf16_t* __restrict__ dy, RowMap dym, const float* __restrict__ x,
                                                          RowMap xm, const float* __restrict__ gamma,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          const float* __restrict__ dres, float* __restrict__ dx,
                                                          bf16_t* __restrict__ dx_bf16, float* __restrict__ partial, int B, int n) {
    constexpr int dim = 256 * NV;
    static_assert(DEPTH >= 1 && DEPTH <= ROWS_PER_WAVE, "prefetch distance in rows");
    extern __shared__ __attribute__((aligned(16))) float red[];  // [4 waves][3][dim]
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    f32x4 gsum[NV], bsum[NV], csum[NV], gm[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        gsum[i] = (f32x4){0, 0, 0, 0}; bsum[i] = (f32x4){0, 0, 0, 0}; csum[i] = (f32x4){0, 0, 0, 0};
        gm[i] = *reinterpret_cast<const f32x4*>(gamma + 4 * (lane + 64 * i));
    }
    const int row0 = (blockIdx.x * 4 + w) * ROWS_PER_WAVE;
    if (row0 < B * n) {
        f32x4 xq[ROWS_PER_WAVE][NV], rq[ROWS_PER_WAVE][NV];
        u32x2 dq[ROWS_PER_WAVE][NV];
        float muq[ROWS_PER_WAVE], rsq[ROWS_PER_WAVE];
        size_t xrowq[ROWS_PER_WAVE];
        auto fetch = [&](int p) {
            const int row = row0 + p, b = row / n, j = row - b * n;
            xrowq[p] = map_row(xm, b, j) * dim;
            const size_t dyrow = map_row(dym, b, j) * dim;
            muq[p] = mean[row]; rsq[p] = rstd[row];
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c4 = 4 * (lane + 64 * i);
                xq[p][i] = *reinterpret_cast<const f32x4*>(x + xrowq[p] + c4);
                dq[p][i] = *reinterpret_cast<const u32x2*>(dy + dyrow + c4);
                rq[p][i] = *reinterpret_cast<const f32x4*>(dres + xrowq[p] + c4);
            }
        };
#pragma unroll
        for (int p = 0; p < DEPTH; ++p) fetch(p);
#pragma unroll
        for (int rr = 0; rr < ROWS_PER_WAVE; ++rr) {
            if (rr + DEPTH < ROWS_PER_WAVE) fetch(rr + DEPTH);
            __builtin_amdgcn_sched_barrier(0);
            const float mu = muq[rr], rs = rsq[rr];
            f32x4 xh[NV], dz[NV];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const u32x2 pk = dq[rr][i];
                const f32x4 d = {__uint_as_float(pk[0] << 16), __uint_as_float(pk[0] & 0xffff0000u),
                                 __uint_as_float(pk[1] << 16), __uint_as_float(pk[1] & 0xffff0000u)};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    xh[i][e] = (xq[rr][i][e] - mu) * rs;
                    dz[i][e] = d[e] * gm[i][e];
                    s1 += dz[i][e];
                    s2 += dz[i][e] * xh[i][e];
                    gsum[i][e] += d[e] * xh[i][e];
                    bsum[i][e] += d[e];
                }
            }
            const float c1 = ln_wave_sum(s1) / dim, c2 = ln_wave_sum(s2) / dim;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c4 = 4 * (lane + 64 * i);
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = rs * (dz[i][e] - c1 - xh[i][e] * c2);
                o += rq[rr][i];
                csum[i] += o;
                *reinterpret_cast<f32x4*>(dx + xrowq[rr] + c4) = o;
                const u32x2 pk = {pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3])};
                *reinterpret_cast<u32x2*>(dx_bf16 + xrowq[rr] + c4) = pk;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (!partial) return;
    float* rg = red + (size_t)w * 3 * dim;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c4 = 4 * (lane + 64 * i);
        *reinterpret_cast<f32x4*>(rg + c4) = gsum[i];
        *reinterpret_cast<f32x4*>(rg + dim + c4) = bsum[i];
        *reinterpret_cast<f32x4*>(rg + 2 * dim + c4) = csum[i];
    }
    __syncthreads();
    float* prow = partial + (size_t)blockIdx.x * 3 * dim;
    for (int c = threadIdx.x; c < 3 * dim; c += 256)
        prow[c] = red[c] + red[3 * dim + c] + red[6 * dim + c] + red[9 * dim + c];
}

#ifndef LN_BWD_DEPTH
#define LN_BWD_DEPTH 4    // rows of operands in flight per wave before the first is reduced (dim <= 768)
#endif
constexpr int RED_ROWS = 8;   // 8 partial rows per thread: short dependent-free load chains, 4x more (cheap) atomics
__global__ __launch_bounds__(256) void ln_bwd_reduce_kernel(const float* __restrict__ partial, int nblk, int dim,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                            float* __restrict__ dcol) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= 3 * dim) return;
    const int r0 = blockIdx.y * RED_ROWS, r1 = min(nblk, r0 + RED_ROWS);
    float s = 0.f;
    for (int r = r0; r < r1; ++r) s += partial[(size_t)r * 3 * dim + c];
    if (c < dim) atomicAdd(dgamma + c, s);
    else if (c < 2 * dim) atomicAdd(dbeta + (c - dim), s);
    else if (dcol) atomicAdd(dcol + (c - 2 * dim), s);
}

// One launch for many small column sums (LayerNorm partial rows, bias block sums): workgroup b runs the host-listed work item
// blocks[b] = (job, column block, row chunk), so jobs of very different shapes share a dense grid.
__global__ __launch_bounds__(256) void colsum_batched_kernel(const MhColsumJob* __restrict__ jobs, int n_jobs,
                                                             const uint64_t* __restrict__ blocks) {
    const uint64_t e = blocks[blockIdx.x];
    const int ji = (int)(e >> 48);
    if (ji >= n_jobs) return;
    const MhColsumJob j = jobs[ji];
    const int c = (int)((e >> 32) & 0xFFFF) * 256 + threadIdx.x;
    const int r0 = (int)(e & 0xFFFFFFFFu) * MH_COLSUM_ROWS;
    if (c >= j.cols || r0 >= j.rows) return;
    const int r1 = min(j.rows, r0 + MH_COLSUM_ROWS);
    // the chunk's loads go out together (the rolled loop was load -> s_waitcnt vmcnt(0) -> add: MH_COLSUM_ROWS dependent
    // round trips per thread); the adds keep their order
    const float* src = j.src + (size_t)r0 * j.ld + c;
    float v[MH_COLSUM_ROWS];
#pragma unroll
    for (int i = 0; i < MH_COLSUM_ROWS; ++i) v[i] = r0 + i < r1 ? src[(size_t)i * j.ld] : 0.f;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MH_COLSUM_ROWS; ++i) s += v[i];
    atomicAdd(j.dst + c, s);
}

}  // namespace

static int ln_nv(int dim) { const int v = (dim / 4 + 63) / 64; return v <= 4 ? v : 8; }

static int layernorm_fwd_impl(const float* x, int x_L, int x_off, const float* gamma, const float* beta, void* y, int y_L,
                              int y_off, int y_is_f32, float* mean, float* rstd, int B, int n, int dim, float eps, void* y8,
                              const float* y8_scale, float* y8_amax, void* stream);

extern "C" int mh_layernorm_fwd(const float* x, int x_L, int x_off, const float* gamma, const float* beta, void* y,
                                int y_L, int y_off, int y_is_f32, float* mean, float* rstd, int B, int n, int dim,
                                float eps, void* stream) {
    return layernorm_fwd_impl(x, x_L, x_off, gamma, beta, y, y_L, y_off, y_is_f32, mean, rstd, B, n, dim, eps, nullptr, nullptr,
                              nullptr, stream);
}

extern "C" int mh_layernorm_fwd_fp8(const float* x, int x_L, int x_off, const float* gamma, const float* beta, void* y,
                                    int y_L, int y_off, float* mean, float* rstd, int B, int n, int dim, float eps, void* y8,
                                    const float* y8_scale, float* y8_amax, void* stream) {
    MH_CHECK_ARG(y8 && y8_scale && ((uintptr_t)y8 % 4) == 0, "mh_layernorm_fwd_fp8: y8 (4-byte aligned) and its scale are required");
    return layernorm_fwd_impl(x, x_L, x_off, gamma, beta, y, y_L, y_off, 0, mean, rstd, B, n, dim, eps, y8, y8_scale, y8_amax, stream);
}

static int layernorm_fwd_impl(const float* x, int x_L, int x_off, const float* gamma, const float* beta, void* y, int y_L,
                              int y_off, int y_is_f32, float* mean, float* rstd, int B, int n, int dim, float eps, void* y8,
                              const float* y8_scale, float* y8_amax, void* stream) {
    MH_CHECK_ARG(x && gamma && beta && y && mean && rstd, "mh_layernorm_fwd: null pointer");
    MH_CHECK_ARG(dim % 4 == 0 && dim >= 4 && dim <= 2048, "mh_layernorm_fwd: dim %d unsupported", dim);
    MH_CHECK_ARG(B > 0 && n > 0 && x_off + n <= x_L && y_off + n <= y_L, "mh_layernorm_fwd: bad row map");
    const int rows = B * n;
    dim3 grid(ceil_div(rows, 4)), block(256);
    hipStream_t s = (hipStream_t)stream;
#define LN_FWD(NV) hipLaunchKernelGGL(ln_fwd_kernel<NV>, grid, block, 0, s, x, RowMap{x_L, x_off}, gamma, beta, y, \
                                      RowMap{y_L, y_off}, y_is_f32, mean, rstd, B, n, dim, eps, (uint8_t*)y8, y8_scale, y8_amax)
#define LN_FWD_FAST_(NV, FP8) hipLaunchKernelGGL((ln_fwd_fast_kernel<NV, FP8>), grid, block, 0, s, x, RowMap{x_L, x_off}, gamma, beta, \
                                                 (bf16_t*)y, RowMap{y_L, y_off}, mean, rstd, B, n, eps, (uint8_t*)y8, y8_scale, y8_amax)
#define LN_FWD_FAST(NV) do { if (y8) LN_FWD_FAST_(NV, true); else LN_FWD_FAST_(NV, false); } while (0)
    const int nvs = ln_nv(dim);
    if (MH_LN_FAST && nvs <= 4 && dim == 256 * nvs && !y_is_f32) {     // the step's own case: straight-line kernel (with or without the e4m3 copy)
        switch (nvs) { case 1: LN_FWD_FAST(1); break; case 2: LN_FWD_FAST(2); break; case 3: LN_FWD_FAST(3); break;
                       default: LN_FWD_FAST(4); }
    } else {
        switch (nvs) { case 1: LN_FWD(1); break; case 2: LN_FWD(2); break; case 3: LN_FWD(3); break;
                       case 4: LN_FWD(4); break; default: LN_FWD(8); }
    }
#undef LN_FWD_FAST
#undef LN_FWD_FAST_
#undef LN_FWD
    MH_LAUNCH_CHECK();
    return 0;
}

extern "C" long mh_layernorm_bwd_workspace(int rows, int dim) {
    return (long)ceil_div(rows, 4 * ROWS_PER_WAVE) * 3 * dim;
}

static int layernorm_bwd_impl(const void* dy, int dy_L, int dy_off, int dy_is_f32, const float* x, int x_L, int x_off,
                              const float* gamma, const float* mean, const float* rstd, const float* dres, float* dx,
                              void* dx_bf16, float* dgamma, float* dbeta, float* dcol, float* workspace, bool partial_only,
                              int B, int n, int dim, void* stream) {
    const int rows = B * n, nblk = ceil_div(rows, 4 * ROWS_PER_WAVE);
    const size_t lds = (size_t)4 * 3 * dim * sizeof(float);
    dim3 grid(nblk), block(256);
    hipStream_t s = (hipStream_t)stream;
    float* part = (dgamma || partial_only) ? workspace : nullptr;
#define LN_BWD(NV) hipLaunchKernelGGL(ln_bwd_kernel<NV>, grid, block, lds, s, dy, RowMap{dy_L, dy_off}, dy_is_f32, x, \
                                      RowMap{x_L, x_off}, gamma, mean, rstd, dres, dx, (bf16_t*)dx_bf16, part, B, n, dim)
#define LN_BWD_FAST(NV, DEPTH) hipLaunchKernelGGL((ln_bwd_fast_kernel<NV, DEPTH>), grid, block, lds, s, (const bf16_t*)dy, \
                                                  RowMap{dy_L, dy_off}, x, RowMap{x_L, x_off}, gamma, mean, rstd, dres, dx, \
                                                  (bf16_t*)dx_bf16, part, B, n)
    const int nvs = ln_nv(dim);
    if (MH_LN_FAST && nvs <= 4 && dim == 256 * nvs && !dy_is_f32 && dres && dx_bf16 && rows % ROWS_PER_WAVE == 0) {
        // the transformer blocks' own case: straight-line kernel, all of a wave's rows in flight (two at dim 1024: registers)
        constexpr int DEPTH = LN_BWD_DEPTH < ROWS_PER_WAVE ? LN_BWD_DEPTH : ROWS_PER_WAVE, DEPTH4 = 2 < ROWS_PER_WAVE ? 2 : ROWS_PER_WAVE;
        switch (nvs) { case 1: LN_BWD_FAST(1, DEPTH); break; case 2: LN_BWD_FAST(2, DEPTH); break;
                       case 3: LN_BWD_FAST(3, DEPTH); break; default: LN_BWD_FAST(4, DEPTH4); }
    } else {
        switch (nvs) { case 1: LN_BWD(1); break; case 2: LN_BWD(2); break; case 3: LN_BWD(3); break;
                       case 4: LN_BWD(4); break; default: LN_BWD(8); }
    }
#undef LN_BWD_FAST
#undef LN_BWD
    if (dgamma && !partial_only)
        hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3(ceil_div(3 * dim, 256), ceil_div(nblk, RED_ROWS)), dim3(256), 0, s,
                           workspace, nblk, dim, dgamma, dbeta, dcol);
    MH_LAUNCH_CHECK();
    return 0;
}

extern "C" int mh_layernorm_bwd(const void* dy, int dy_L, int dy_off, int dy_is_f32, const float* x, int x_L, int x_off,
                                const float* gamma, const float* mean, const float* rstd, const float* dres, float* dx,
                                void* dx_bf16, float* dgamma, float* dbeta, float* dcol, float* workspace, int B, int n,
                                int dim, void* stream) {
    MH_CHECK_ARG(dy && x && gamma && mean && rstd && dx, "mh_layernorm_bwd: null pointer");
    MH_CHECK_ARG((dgamma == nullptr) == (dbeta == nullptr), "mh_layernorm_bwd: dgamma/dbeta must come together");
    MH_CHECK_ARG(!dgamma || workspace, "mh_layernorm_bwd: parameter gradients need the workspace");
    MH_CHECK_ARG(!dcol || dgamma, "mh_layernorm_bwd: dcol needs dgamma/dbeta");
    MH_CHECK_ARG(dim % 4 == 0 && dim >= 4 && dim <= 2048, "mh_layernorm_bwd: dim %d unsupported", dim);
    MH_CHECK_ARG(B > 0 && n > 0 && x_off + n <= x_L && dy_off + n <= dy_L, "mh_layernorm_bwd: bad row map");
    return layernorm_bwd_impl(dy, dy_L, dy_off, dy_is_f32, x, x_L, x_off, gamma, mean, rstd, dres, dx, dx_bf16, dgamma, dbeta, dcol,
                              workspace, false, B, n, dim, stream);
}

extern "C" int mh_layernorm_bwd_partial(const void* dy, int dy_L, int dy_off, int dy_is_f32, const float* x, int x_L, int x_off,
                                        const float* gamma, const float* mean, const float* rstd, const float* dres, float* dx,
                                        void* dx_bf16, float* workspace, int B, int n, int dim, void* stream) {
    MH_CHECK_ARG(dy && x && gamma && mean && rstd && dx && workspace, "mh_layernorm_bwd_partial: null pointer");
    MH_CHECK_ARG(dim % 4 == 0 && dim >= 4 && dim <= 2048, "mh_layernorm_bwd_partial: dim %d unsupported", dim);
    MH_CHECK_ARG(B > 0 && n > 0 && x_off + n <= x_L && dy_off + n <= dy_L, "mh_layernorm_bwd_partial: bad row map");
    return layernorm_bwd_impl(dy, dy_L, dy_off, dy_is_f32, x, x_L, x_off, gamma, mean, rstd, dres, dx, dx_bf16, nullptr, nullptr,
                              nullptr, workspace, true, B, n, dim, stream);
}

extern "C" int mh_colsum_batched(const MhColsumJob* jobs_device, int n_jobs, const uint64_t* blocks_device, int n_blocks,
                                 void* stream) {
    MH_CHECK_ARG(jobs_device && blocks_device && n_jobs > 0 && n_jobs < 65536 && n_blocks > 0, "mh_colsum_batched: bad arguments");
    hipLaunchKernelGGL(colsum_batched_kernel, dim3(n_blocks), dim3(256), 0, (hipStream_t)stream, jobs_device, n_jobs, blocks_device);
    MH_LAUNCH_CHECK();
    return 0;
}
