// Pieces shared by the two GEMM kernels (gemm.hip: 128x128 register-staged; gemm_dma.hip: 256x256 LDS-DMA ring):
// the parameter block and the LDS-staged epilogues for a wave tile of (16*MT) x 64 accumulators held as acc[j][i]
// (j = n-tile 0..3, i = m-tile 0..MT-1; lane owns rows 16i + (lane&15), columns 16j + 4*(lane>>4) .. +3).
#pragma once
#include "common.hpp"
#include "../../include/maestro_hip.h"

struct GemmParams {
    const bf16_t* A; const bf16_t* B; void* C;
    const float* bias; const float* res; const bf16_t* aux_in; bf16_t* aux_out; float* colsum;
    int M, N, K, lda, ldb, ldc, ldr, ldaux, flags;
    int tiles_m, tiles_n, k_per_split;
    int fast;               // 1: buffer-load path (no K tail inside a K-minor operand, extents < 2 GiB)
    unsigned a_bytes, b_bytes;
    // fp8 path (gemm_fp8.hip; null / 0 for the bf16 kernels): the accumulators are multiplied by *descale_a x *descale_b (the
    // per-tensor dequantisation factors, device scalars) before the epilogue; c8 (optional, bf16-output epilogues): an OCP
    // e4m3 copy of the output scaled by *c8_scale, the operand of the next fp8 GEMM, with max |value| folded into *c8_amax
    const float* descale_a = nullptr; const float* descale_b = nullptr;
    uint8_t* c8 = nullptr; const float* c8_scale = nullptr; float* c8_amax = nullptr; int ldc8 = 0;
};

// Output-tile coordinates of raster id `id` (ids already XCD-remapped: every XCD owns a contiguous run).  Ids walk GROUP_M m-tiles
// before moving to the next n-tile, so the workgroups co-resident under one XCD's L2 cover a GROUP_M x (64 / GROUP_M) block of tiles
// and sweep the B operand (the weight matrix of a forward / dgrad GEMM) panel by panel.  Group g starts its sweep at n-tile
// x(g) tiles_n / 8, x(g) = the XCD whose run holds the group (round 5).  Unrotated, whenever the runs line up with the groups
// (M = 8192: 8 groups, 8 XCDs) all eight XCDs sweep the weight panels in the SAME order at the same time: every panel misses in all
// eight L2s at once, and inside the step -- where a layer's weights come from HBM, not from the Infinity Cache as in a relaunch loop
// -- the launch pays that latency at every tile transition: +11 us on a 42 us fc1 launch for 4.7 MB of weights.  Rotated, a panel is
// fetched from HBM by ONE XCD and found in the Infinity Cache by the others: fc1 72 -> 60 us, qkv 47 -> 37 us with cold operands,
// C3 step -1 % (profiles/r05_gap_table.md).  A bijection of the tile set for any shape; same fp32 sums per output element.
template <int GROUP_M>
__device__ __forceinline__ void raster_tile(const GemmParams& p, int id, int& tile_m, int& tile_n) {
    const int per_group = GROUP_M * p.tiles_n;
    const int group = id / per_group, in_group = id - group * per_group;
    const int first_m = group * GROUP_M;
    const int gsz = min(p.tiles_m - first_m, GROUP_M);
    tile_m = first_m + in_group % gsz;
    int n = in_group / gsz;
    const int groups = (p.tiles_m + GROUP_M - 1) / GROUP_M;
    n += ((group * 8) / groups) * p.tiles_n / 8;
    tile_n = n >= p.tiles_n ? n - p.tiles_n : n;
}

// 8 floats -> 8 OCP fp8 bytes, e4m3 (saturating at +-448: e4m3fn has no infinity, an overflow would become NaN) or e5m2
// (gradients; saturating at +-57344)
__device__ __forceinline__ u32x2 pack_fp8x8(f32x4 lo, f32x4 hi, float s, bool e5m2 = false) {
    const float lim = e5m2 ? 57344.f : 448.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        lo[e] = fminf(fmaxf(lo[e] * s, -lim), lim);
        hi[e] = fminf(fmaxf(hi[e] * s, -lim), lim);
    }
    int a, b;
    if (e5m2) {
        a = __builtin_amdgcn_cvt_pk_bf8_f32(lo[0], lo[1], 0, false);
        a = __builtin_amdgcn_cvt_pk_bf8_f32(lo[2], lo[3], a, true);
        b = __builtin_amdgcn_cvt_pk_bf8_f32(hi[0], hi[1], 0, false);
        b = __builtin_amdgcn_cvt_pk_bf8_f32(hi[2], hi[3], b, true);
    } else {
        a = __builtin_amdgcn_cvt_pk_fp8_f32(lo[0], lo[1], 0, false);
        a = __builtin_amdgcn_cvt_pk_fp8_f32(lo[2], lo[3], a, true);
        b = __builtin_amdgcn_cvt_pk_fp8_f32(hi[0], hi[1], 0, false);
        b = __builtin_amdgcn_cvt_pk_fp8_f32(hi[2], hi[3], b, true);
    }
    return (u32x2){(uint32_t)a, (uint32_t)b};
}
__device__ __forceinline__ uint32_t pack_e4m3x4(f32x4 v, float s) {   // 4 floats -> 4 saturating OCP e4m3 bytes
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fminf(fmaxf(v[e] * s, -448.f), 448.f);
    int a = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0, false);
    a = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], a, true);
    return (uint32_t)a;
}
// Fold an absmax (v >= 0 or NaN: the int order of the bits is the float order, NaN on top: see amax_fold) into a slot's amax ROW (MH_FP8_AMAX_PITCH floats).
// Atomics on one cache line retire at ~10 ns apiece whatever the address inside it (scripts/micro_amax.hip; a look-before-
// you-add needs an agent-scope load to see other CUs' maxima at all -- plain and nontemporal loads are served stale -- and
// still leaves the ~8000 waves resident at launch to storm the word: +80 us on a 7 us LayerNorm).  32 sub-slots 256 bytes
// apart spread the adds over L2 channels: +0.5 us, no look needed; mh_fp8_update_scales takes the maximum over the row.
__device__ __forceinline__ void atomic_max_pos(float* row, float v) {
    const unsigned k = ((blockIdx.x + blockIdx.y * gridDim.x) * (blockDim.x >> 6) + (threadIdx.x >> 6)) & (MH_FP8_AMAX_SUBSLOTS - 1);
    __hip_atomic_fetch_max(reinterpret_cast<int*>(row) + k * MH_FP8_AMAX_STRIDE, __float_as_int(v), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
}

// gemm_dma.hip: the LDS-DMA tile family (tile = MH_TILE_DMA_*); -2 = not eligible, nothing launched
int gemm_dma_dispatch(int tile, int layout, int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C,
                      int ldc, int flags, const float* bias, const float* res, int ldr, const void* aux_in, void* aux_out,
                      int ldaux, float* colsum, void* stream);

// gemm_pp.hip: the persistent 128x128 tile with a second accumulator set (tile = MH_TILE_PP_128); -2 = not eligible
int gemm_pp_dispatch(int layout, int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C, int ldc, int flags,
                     const float* bias, const float* res, int ldr, const void* aux_in, void* aux_out, int ldaux, float* colsum,
                     void* stream, int diag = 0);

// MH_GEMM_AUX_U8: the saved GELU derivative as a byte code, value = code / 200 - 0.13 (range [-0.129, 1.129] -> 0.2 .. 251.8)
__device__ __forceinline__ u32x2 pack_dgelu_u8x8(f32x4 lo, f32x4 hi) {
    unsigned a = 0, b = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        a = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(lo[e], 200.f, 26.f), e, a);   // (v_cvt_pk_u8_f32 rounds to nearest)
        b = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(hi[e], 200.f, 26.f), e, b);
    }
    return (u32x2){a, b};
}
__device__ __forceinline__ f32x4 unpack_dgelu_u8x4(unsigned w) {
    // ((float)(w >> 8k & 0xff) is one v_cvt_f32_ubyte<k>)
    return (f32x4){__builtin_fmaf((float)(w & 0xffu), 0.005f, -0.13f), __builtin_fmaf((float)((w >> 8) & 0xffu), 0.005f, -0.13f),
                   __builtin_fmaf((float)((w >> 16) & 0xffu), 0.005f, -0.13f), __builtin_fmaf((float)(w >> 24), 0.005f, -0.13f)};
}

// Epilogues.  fp32 output (bias / residual): straight from the accumulators, 16 bytes per lane, four lanes per 64-byte row
// segment (measured faster than an LDS transposition: 152 -> 125 us on the decoder fc2 GEMM).  bf16 output (bias / GELU /
// GELU' / aux): each wave transposes its tile through a private LDS region (passes of 32 rows, 68-float row pitch:
// conflict-free ds_write_b128 / ds_read_b128) so that the aux reads and the C / aux writes are done in ROW-MAJOR lane
// order: 8 lanes cover one 128-byte row segment with 16-byte accesses.
template <int MT, int RP = 32>   // RP = rows per staging pass (32: 32 x 68 floats per wave; 16: half of that)
__device__ __forceinline__ void gemm_epilogue_store(const GemmParams& p, const f32x4 (&acc)[4][MT], float* st, int m_base,
                                                    int n_base) {
    constexpr int TPP = RP / 16, NPASS = MT / TPP, SUB = RP / 8;   // m-tiles per pass, passes, 8-row groups per pass
    static_assert((RP == 16 || RP == 32) && MT % TPP == 0, "staging pass geometry");
    const int l = threadIdx.x & 63, g = l >> 4, lm = l & 15;
    const bool out_f32 = p.flags & MH_GEMM_OUT_F32;
    const float alpha = p.descale_a ? *p.descale_a * *p.descale_b : 1.f;   // fp8 operands: per-tensor dequantisation
    if (out_f32) {
        // fp32 output straight from the accumulators: lane (lm, g) owns 4 consecutive columns of row 16i + lm in every
        // n-tile, i.e. 16-byte accesses that four lanes extend to a 64-byte row segment; no LDS round trip.
        // Round 5: the residual rows of FOUR row blocks (16 loads) are requested before the first store.  The compiler cannot prove
        // that `res` and `C` do not overlap, so with the loads of row block i + 1 written behind the stores of row block i it kept
        // them there: MT dependent memory round trips per workgroup (8 on the 256 x 256 tile), each a full HBM latency inside the
        // step, where the residual stream of a layer is cold -- the decoder fc2 launch spent 38 of its 125 us in this epilogue.
        f32x4 bias4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n_base + 16 * j + 4 * g;
            bias4[j] = (f32x4){0, 0, 0, 0};
            if ((p.flags & MH_GEMM_BIAS) && n < p.N) bias4[j] = *reinterpret_cast<const f32x4*>(p.bias + n);
        }
        constexpr int IB = MT % 4 == 0 ? 4 : MT % 3 == 0 ? 3 : MT;      // row blocks per batch of residual loads (16 IB registers)
        static_assert(MT % IB == 0, "row blocks per wave");
#pragma unroll
        for (int i0 = 0; i0 < MT; i0 += IB) {
            f32x4 add[IB][4];
#pragma unroll
            for (int ii = 0; ii < IB; ++ii) {
                const int m = m_base + 16 * (i0 + ii) + lm;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = n_base + 16 * j + 4 * g;
                    add[ii][j] = bias4[j];
                    if ((p.flags & MH_GEMM_RESIDUAL) && m < p.M && n < p.N)
                        add[ii][j] += *reinterpret_cast<const f32x4*>(p.res + (size_t)m * p.ldr + n);
                }
            }
#pragma unroll
            for (int ii = 0; ii < IB; ++ii) {
                const int m = m_base + 16 * (i0 + ii) + lm;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = n_base + 16 * j + 4 * g;
                    if (m < p.M && n < p.N)
                        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + n) = acc[j][i0 + ii] * alpha + add[ii][j];
                }
            }
        }
        return;
    }
    f32x4 cs_lo = {0, 0, 0, 0}, cs_hi = {0, 0, 0, 0};   // MH_GEMM_COLSUM: this lane's 8 columns summed over its rows
    const float s8 = p.c8 ? *p.c8_scale : 0.f;
    float amax8 = 0.f;
    // The saved GELU' bytes (or pre-activations) of ALL passes are requested before the first store (round 5: behind the stores of
    // pass p the compiler could not move the loads of pass p + 1 -- aux_in / C may overlap for all it knows -- and every pass waited
    // for a cold load AND for the previous pass's stores to complete).
    u32x4 aux_all[NPASS][SUB];
    {
        const int n = n_base + (l & 7) * 8;
        if (p.flags & (MH_GEMM_DGELU | MH_GEMM_MULAUX)) {
#pragma unroll
            for (int pass_m = 0; pass_m < NPASS; ++pass_m)
#pragma unroll
                for (int pass = 0; pass < SUB; ++pass) {
                    const int m = m_base + RP * pass_m + pass * 8 + (l >> 3);
                    aux_all[pass_m][pass] = (u32x4){0, 0, 0, 0};
                    if (m < p.M && n < p.N) {
                        if (p.flags & MH_GEMM_AUX_U8) {
                            const u32x2 b8 = *reinterpret_cast<const u32x2*>(reinterpret_cast<const uint8_t*>(p.aux_in) + (size_t)m * p.ldaux + n);
                            aux_all[pass_m][pass] = (u32x4){b8[0], b8[1], 0, 0};
                        } else {
                            aux_all[pass_m][pass] = *reinterpret_cast<const u32x4*>(p.aux_in + (size_t)m * p.ldaux + n);
                        }
                    }
                }
        }
    }
#pragma unroll
    for (int pass_m = 0; pass_m < NPASS; ++pass_m) {
#pragma unroll
        for (int i = TPP * pass_m; i < TPP * pass_m + TPP; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *reinterpret_cast<f32x4*>(st + (16 * (i - TPP * pass_m) + lm) * 68 + 16 * j + 4 * g) = acc[j][i];
        {
            const int c = (l & 7) * 8, n = n_base + c;
            f32x4 b_lo = {0, 0, 0, 0}, b_hi = {0, 0, 0, 0};
            if ((p.flags & MH_GEMM_BIAS) && n < p.N) {
                b_lo = *reinterpret_cast<const f32x4*>(p.bias + n);
                b_hi = *reinterpret_cast<const f32x4*>(p.bias + n + 4);
            }
            const u32x4 (&aux_pre)[SUB] = aux_all[pass_m];
#pragma unroll
            for (int pass = 0; pass < SUB; ++pass) {
                const int r = pass * 8 + (l >> 3), m = m_base + RP * pass_m + r;
                f32x4 lo = *reinterpret_cast<const f32x4*>(st + r * 68 + c);
                f32x4 hi = *reinterpret_cast<const f32x4*>(st + r * 68 + c + 4);
                if (m < p.M && n < p.N) {
                    if (p.descale_a) { lo = lo * alpha + b_lo; hi = hi * alpha + b_hi; }   // fp8 operands: per-tensor dequantisation
                    else { lo += b_lo; hi += b_hi; }
                    if (p.flags & MH_GEMM_GELU) {
                        f32x4 c_lo, d_lo, c_hi, d_hi;      // CDF and PDF of the pre-activation (one exp2 + one rcp per element)
                        gelu_cdf_pdf4(lo, c_lo, d_lo);
                        gelu_cdf_pdf4(hi, c_hi, d_hi);
                        if (p.aux_out) {
                            f32x4 a_lo = lo, a_hi = hi;     // saved for the backward: the pre-activation, or GELU' = CDF + x PDF
                            if (p.flags & MH_GEMM_AUX_DGELU) { a_lo = lo * d_lo + c_lo; a_hi = hi * d_hi + c_hi; }
                            if (p.flags & MH_GEMM_AUX_U8) {
                                *reinterpret_cast<u32x2*>(reinterpret_cast<uint8_t*>(p.aux_out) + (size_t)m * p.ldaux + n) =
                                    pack_dgelu_u8x8(a_lo, a_hi);
                            } else {
                                u32x4 pk = {pack_bf2(a_lo[0], a_lo[1]), pack_bf2(a_lo[2], a_lo[3]), pack_bf2(a_hi[0], a_hi[1]),
                                            pack_bf2(a_hi[2], a_hi[3])};
                                *reinterpret_cast<u32x4*>(p.aux_out + (size_t)m * p.ldaux + n) = pk;
                            }
                        }
                        lo *= c_lo; hi *= c_hi;
                    }
                    if (p.flags & MH_GEMM_MULAUX) {
                        const u32x4 pk = aux_pre[pass];
                        if (p.flags & MH_GEMM_AUX_U8) {
                            lo *= unpack_dgelu_u8x4(pk[0]);
                            hi *= unpack_dgelu_u8x4(pk[1]);
                        } else {
                            lo *= (f32x4){__uint_as_float(pk[0] << 16), __uint_as_float(pk[0] & 0xffff0000u),
                                          __uint_as_float(pk[1] << 16), __uint_as_float(pk[1] & 0xffff0000u)};
                            hi *= (f32x4){__uint_as_float(pk[2] << 16), __uint_as_float(pk[2] & 0xffff0000u),
                                          __uint_as_float(pk[3] << 16), __uint_as_float(pk[3] & 0xffff0000u)};
                        }
                    }
                    if (p.flags & MH_GEMM_DGELU) {
                        const u32x4 pk = aux_pre[pass];
                        const f32x4 h_lo = {__uint_as_float(pk[0] << 16), __uint_as_float(pk[0] & 0xffff0000u),
                                            __uint_as_float(pk[1] << 16), __uint_as_float(pk[1] & 0xffff0000u)};
                        const f32x4 h_hi = {__uint_as_float(pk[2] << 16), __uint_as_float(pk[2] & 0xffff0000u),
                                            __uint_as_float(pk[3] << 16), __uint_as_float(pk[3] & 0xffff0000u)};
                        lo *= gelu_erf_grad4(h_lo);
                        hi *= gelu_erf_grad4(h_hi);
                    }
                    if (p.flags & MH_GEMM_COLSUM) { cs_lo += lo; cs_hi += hi; }   // (uniform branch: the fc1 epilogue is VALU-bound)
                    u32x4 pk = {pack_bf2(lo[0], lo[1]), pack_bf2(lo[2], lo[3]), pack_bf2(hi[0], hi[1]), pack_bf2(hi[2], hi[3])};
                    *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(p.C) + (size_t)m * p.ldc + n) = pk;
                    if (p.c8) {
                        *reinterpret_cast<u32x2*>(p.c8 + (size_t)m * p.ldc8 + n) = pack_fp8x8(lo, hi, s8, p.flags & MH_GEMM_C8_E5M2);
#pragma unroll
                        for (int e = 0; e < 4; ++e) amax8 = amax_fold(amax_fold(amax8, lo[e]), hi[e]);
                    }
                }
            }
            if ((p.flags & MH_GEMM_COLSUM) && (RP * (pass_m + 1)) % 64 == 0) {
                // one 64-row block done: lanes with equal (l & 7) hold the same 8 columns -> fold the 8 row groups and
                // store the block's partial column sums as row (m / 64) of the [ceil(M / 64), N] workspace (no atomics)
#pragma unroll
                for (int o = 8; o < 64; o <<= 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        cs_lo[e] += __shfl_xor(cs_lo[e], o, 64);
                        cs_hi[e] += __shfl_xor(cs_hi[e], o, 64);
                    }
                }
                const int m_blk = m_base + RP * (pass_m + 1) - 64;
                if (l < 8 && n < p.N && m_blk < p.M) {
                    float* dst = p.colsum + (size_t)(m_blk >> 6) * p.N + n;
                    *reinterpret_cast<f32x4*>(dst) = cs_lo;
                    *reinterpret_cast<f32x4*>(dst + 4) = cs_hi;
                }
                cs_lo = (f32x4){0, 0, 0, 0}; cs_hi = (f32x4){0, 0, 0, 0};
            }
        }
    }
    if (p.c8 && p.c8_amax) {
        amax8 = wave_amax(amax8);
        if (l == 0 && amax_nonzero(amax8)) atomic_max_pos(p.c8_amax, amax8);
    }
}

// Split-K / accumulate epilogue: fp32 atomic adds, transposed through the same LDS region (65-float pitch) so that one
// wave instruction adds one 256-byte row segment.
template <int MT>
__device__ __forceinline__ void gemm_epilogue_atomic(const GemmParams& p, const f32x4 (&acc)[4][MT], float* st, int m_base,
                                                     int n_base) {
    const int l = threadIdx.x & 63, g = l >> 4, lm = l & 15;
#pragma unroll
    for (int pass_m = 0; pass_m < MT / 2; ++pass_m) {
#pragma unroll
        for (int i = 2 * pass_m; i < 2 * pass_m + 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) st[(16 * (i - 2 * pass_m) + lm) * 65 + 16 * j + 4 * g + r] = acc[j][i][r];
        const int n = n_base + l;
        if (n < p.N) {
            for (int rr = 0; rr < 32; ++rr) {
                const int m = m_base + 32 * pass_m + rr;
                if (m < p.M) atomicAdd(reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + n, st[rr * 65 + l]);
            }
        }
    }
}
