// Fused (flash-style) multi-head attention forward / backward for gfx950; head_dim D in {32, 64}, no mask, no dropout.
// qkv bf16 [B, N, 3, H, D]; out / dout bf16 [B, N, H*D]; lse f32 [B, H, N]; arbitrary N (keys >= N masked to -inf).
//
// MFMA formulation (v_mfma_f32_16x16x32_bf16, D[row][col] = sum_k first[row][k] * second[k][col]):
//   forward / dQ kernels keep the QUERY on the lane:   S^T[key][q] = K . Q^T,  O^T[d][q] = V^T . P^T
//     -> softmax statistics (m, l), the rescale factor and delta/lse are per-lane scalars (no cross-lane traffic
//        except one 2-step max exchange per KV tile), and the P^T accumulator IS the B operand of the next MFMA.
//   dK/dV kernel keeps the KEY on the lane:             S[q][key] = Q . K^T,  dV^T[d][key] = dO^T . P,  dK^T = Q^T . dS
// K/V (or Q/dO) tiles of 64 rows are staged in LDS twice when needed: a "row image" read by ds_read_b128 and a
// "transpose image" read by ds_read_b64_tr_b16 (hardware transpose), both XOR-swizzled to be bank-conflict free.
#include "common.hpp"
#include "../../include/maestro_hip.h"
#include <type_traits>

namespace {

constexpr float LOG2E = 1.4426950408889634f;

// Softmax VALU arithmetic on four scores at a time.  ATTN_SCALAR_VALU=1 (with -fno-slp-vectorize) keeps every operation a
// scalar v_fma_f32 / v_add_f32 / v_mul_f32; 0 writes whole-vector expressions, which the compiler turns into packed
// v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32 (two floats per instruction).
#ifndef ATTN_SCALAR_VALU
#define ATTN_SCALAR_VALU 1
#endif
// ATTN_DIAG (diagnostic builds only, WRONG results; MH_ATTN_FLAGS="-DATTN_DIAG=n"): what is one VALU issue slot per score worth?
//   bit 0: forward without the row-sum adds;  bit 1: backward without the scale / lse FMA and without the "- delta" add
#ifndef ATTN_DIAG
#define ATTN_DIAG 0
#endif
// ATTN_SUM_MFMA = 1: the forward's softmax denominators come out of the matrix cores -- one more 16-row block of "V^T" whose row 0 is all
// ones, so that O^T's extra row 0 is sum_k P^T[k][q] -- instead of one v_add_f32 per score (the kernel is VALU-bound: 5 -> 4 issue slots per
// score at D = 32; the 4 extra MFMAs per key tile hold the issue port for 32 cycles).  The sum is then over the bf16-rounded probabilities,
// the same values the P V product uses.  D = 32 only: same-box A/B -6 % (N = 1024: 155 -> 145 us; N = 400: 36.5 -> 34.5); at D = 64 the
// kernel is less VALU-bound and the fifth accumulator block is a wash (-4 % ... +2 %), so the fp32 adds stay there.  0
// (MH_ATTN_FLAGS="-DATTN_SUM_MFMA=0") keeps the fp32 adds everywhere (A/B aid).
#ifndef ATTN_SUM_MFMA
#define ATTN_SUM_MFMA 1
#endif
// mh_attn_bwd's rule: 0 = always the two kernels.  Round 4 measured the single-pass kernel (variant 2 of mh_attn_bwd_variant) at parity
// on its best shape (N = 1024, D = 32: 372 vs 358 us) and slower elsewhere (profiles/r04_attn_bwd.txt): explicit variant only.
#ifndef MH_ATTN_FUSED_BWD_DEFAULT
#define MH_ATTN_FUSED_BWD_DEFAULT 0
#endif
__device__ __forceinline__ f32x4 fms4(f32x4 a, float c, f32x4 b) {   // a * c - b
#if ATTN_SCALAR_VALU
    return (f32x4){__builtin_fmaf(a[0], c, -b[0]), __builtin_fmaf(a[1], c, -b[1]), __builtin_fmaf(a[2], c, -b[2]),
                   __builtin_fmaf(a[3], c, -b[3])};
#else
    return a * (f32x4){c, c, c, c} - b;
#endif
}
__device__ __forceinline__ f32x4 scale4(f32x4 a, float c) {
#if ATTN_SCALAR_VALU
    return (f32x4){a[0] * c, a[1] * c, a[2] * c, a[3] * c};
#else
    return a * c;
#endif
}
__device__ __forceinline__ f32x4 mul4(f32x4 p, f32x4 a) {   // p * a
#if ATTN_SCALAR_VALU
    return (f32x4){p[0] * a[0], p[1] * a[1], p[2] * a[2], p[3] * a[3]};
#else
    return p * a;
#endif
}
typedef __attribute__((address_space(3))) unsigned char lds_u8;

template <int D> __device__ __forceinline__ int row_swz(int row) {
    if constexpr (D == 64) return row & 7;
    else return (0x78 >> (2 * ((row >> 2) & 3))) & 3;
}
template <int D> __device__ __forceinline__ int tr_swz(int row) {
    if constexpr (D == 64) return (row >> 1) & 3;
    else return (row >> 2) & 1;
}

// Stage a [64 rows][D] bf16 tile: global -> registers.  Row r of the tile is token (row0 + r); rows >= nrows give zeros.
template <int D>
__device__ __forceinline__ void tile_load(const bf16_t* __restrict__ base, size_t row_stride, int row0, int nrows, u32x4 (&v)[2]) {
    constexpr int CPR = D / 8;            // 16-B chunks per row
    constexpr int RPI = 256 / CPR;        // rows per pass
    const int t = threadIdx.x, c = t % CPR, r = t / CPR;
#pragma unroll
    for (int i = 0; i < 64 / RPI; ++i) {
        const int row = r + RPI * i;
        u32x4 z = {0, 0, 0, 0};
        if (row0 + row < nrows) z = *reinterpret_cast<const u32x4*>(base + (size_t)(row0 + row) * row_stride + c * 8);
        v[i] = z;
    }
}
template <int D>
__device__ __forceinline__ void tile_store_row(unsigned char* img, const u32x4 (&v)[2]) {
    constexpr int CPR = D / 8, RPI = 256 / CPR;
    const int t = threadIdx.x, c = t % CPR, r = t / CPR;
#pragma unroll
    for (int i = 0; i < 64 / RPI; ++i) {
        const int row = r + RPI * i;
        *reinterpret_cast<u32x4*>(img + row * (2 * D) + ((c ^ row_swz<D>(row)) << 4)) = v[i];
    }
}
template <int D>
__device__ __forceinline__ void tile_store_tr(unsigned char* img, const u32x4 (&v)[2]) {
    constexpr int CPR = D / 8, RPI = 256 / CPR;
    const int t = threadIdx.x, c = t % CPR, r = t / CPR;
#pragma unroll
    for (int i = 0; i < 64 / RPI; ++i) {
        const int row = r + RPI * i;
        *reinterpret_cast<u32x4*>(img + row * (2 * D) + ((((c >> 1) ^ tr_swz<D>(row)) << 5) | ((c & 1) << 4))) = v[i];
    }
}
// first/second-operand fragment of rows (r0 + lane&15), k = d in [32*ks + 8*g, +8)
template <int D>
__device__ __forceinline__ bf16x8 frag_row(const unsigned char* img, int r0, int ks) {
    const int l = threadIdx.x & 63, row = r0 + (l & 15), ch = 4 * ks + (l >> 4);
    return *reinterpret_cast<const bf16x8*>(img + row * (2 * D) + ((ch ^ row_swz<D>(row)) << 4));
}
// transposed fragment: lane (col = 16*ct + lane&15, g): element j<4 -> row 32u + 4g + j ; j>=4 -> row 32u + 16 + 4g + (j-4)
template <int D>
__device__ __forceinline__ bf16x8 frag_tr(const unsigned char* img, int ct, int u) {
    const int l = threadIdx.x & 63, g = l >> 4, q = (l & 15) >> 2, p = l & 3;
    const int r_lo = 32 * u + 4 * g + q, r_hi = r_lo + 16;
    const lds_u8* b = (const lds_u8*)img;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)(b + r_lo * (2 * D) + (((ct ^ tr_swz<D>(r_lo)) << 5) + p * 8)));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)(b + r_hi * (2 * D) + (((ct ^ tr_swz<D>(r_hi)) << 5) + p * 8)));
    s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, r);
}
// load a B/A-operand fragment straight from global: row `tok` (or zeros), d in [32*ks + 8*g, +8)
__device__ __forceinline__ bf16x8 frag_global(const bf16_t* __restrict__ base, size_t row_stride, int tok, int ntok, int ks) {
    const int g = (threadIdx.x & 63) >> 4;
    u32x4 z = {0, 0, 0, 0};
    if (tok < ntok) z = *reinterpret_cast<const u32x4*>(base + (size_t)tok * row_stride + 32 * ks + 8 * g);
    return __builtin_bit_cast(bf16x8, z);
}
// two accumulator tiles (rows 4g+r of tiles 2u, 2u+1) -> bf16x8 operand with the k-permutation of frag_tr
__device__ __forceinline__ bf16x8 pack_acc(const f32x4& a, const f32x4& b) {
    u32x4 r = {pack_bf2(a[0], a[1]), pack_bf2(a[2], a[3]), pack_bf2(b[0], b[1]), pack_bf2(b[2], b[3])};
    return __builtin_bit_cast(bf16x8, r);
}
__device__ __forceinline__ float exp2_fast(float x) { return __builtin_amdgcn_exp2f(x); }
// operand fragment times a constant, rounded to bf16 once more (the backward kernels fold scale x log2(e) into Q resp. K ONCE per
// workgroup, so that the S MFMA chain -- started from -lse log2(e) -- ends in the exp2 argument itself: round 5)
__device__ __forceinline__ bf16x8 scale_frag(bf16x8 f, float c) {
    const u32x4 u = __builtin_bit_cast(u32x4, f);
    u32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = pack_bf2(__uint_as_float(u[e] << 16) * c, __uint_as_float(u[e] & 0xffff0000u) * c);
    return __builtin_bit_cast(bf16x8, r);
}

// =============================================================================================== forward
template <int D>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                       float* __restrict__ lse, int N, int H, float scale) {
    constexpr int KS = D / 32, DT = D / 16, TB = 64 * D * 2;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * TB];  // K row image | V transpose image
    unsigned char* k_img = smem;
    unsigned char* v_img = smem + TB;
    // XCD-aware work order: workgroup ids go round-robin over the 8 XCDs, so consecutive ids (the blocks of ONE head) would
    // each pull that head's K / V through a different L2 (measured: 3.2x the algorithmic bytes at N = 1024).  xcd_remap
    // gives every XCD a contiguous run of (batch, head, block) items: a head's blocks, and the neighbouring head that shares
    // its 128-byte lines at D = 32, stay under one L2.
    const int gx = (N + 127) >> 7;
    const int wid = xcd_remap(blockIdx.x, gridDim.x);
    const int b = wid / (gx * H), h = (wid / gx) % H;
    const int q_blk = (wid % gx) * 128;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, g = l >> 4, lq = l & 15;
    const size_t rs = (size_t)3 * H * D;                              // token stride in qkv
    const bf16_t* qb = qkv + (size_t)b * N * rs + (size_t)h * D;       // q part
    const bf16_t* kb = qb + (size_t)H * D;
    const bf16_t* vb = qb + (size_t)2 * H * D;
    const int q0 = q_blk + 32 * w;

    bf16x8 qf[2][KS];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[qt][ks] = frag_global(qb, rs, q0 + 16 * qt + lq, N, ks);

    f32x4 o[2][DT];
    float m[2], lsum[2];
    constexpr bool SUM_MFMA = ATTN_SUM_MFMA && D == 32;
    f32x4 ol[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};      // SUM_MFMA: row 0 (lanes 0-15, element 0) = the running denominator of query column lq
    bf16x8 ones;
    {
        const short one = lq == 0 ? (short)0x3F80 : (short)0;     // first operand [row][k]: row 0 = 1.0 (bf16), rows 1-15 = 0
        const s16x8 r = {one, one, one, one, one, one, one, one};
        ones = __builtin_bit_cast(bf16x8, r);
    }
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        m[qt] = -INFINITY; lsum[qt] = 0.f;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[qt][dt] = (f32x4){0, 0, 0, 0};
    }
    const float c = scale * LOG2E;
    const int ntile = (N + 63) / 64;
    u32x4 rk[2], rv[2];
    tile_load<D>(kb, rs, 0, N, rk);
    tile_load<D>(vb, rs, 0, N, rv);
    // The KV loop is peeled: full tiles run a body without any key masking; only the last, partial tile (N % 64 != 0) pays
    // for the compares and selects (if-converted, they cost ~20 % of the VALU-bound loop when left in the common body).
    auto kv_tile = [&](int it, auto tail_tag) {
        constexpr bool TAIL = decltype(tail_tag)::value;
        const int kv0 = it * 64;
        __syncthreads();  // previous tile fully consumed
        tile_store_row<D>(k_img, rk);
        tile_store_tr<D>(v_img, rv);
        __syncthreads();
        if (it + 1 < ntile) {
            tile_load<D>(kb, rs, kv0 + 64, N, rk);
            tile_load<D>(vb, rs, kv0 + 64, N, rv);
        }
        if (q0 >= N) return;  // wave has no valid query: only helps staging
        f32x4 s[2][4];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) s[qt][kt] = (f32x4){0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const bf16x8 kf = frag_row<D>(k_img, 16 * kt, ks);
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) s[qt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[qt][ks], s[qt][kt], 0, 0, 0);
            }
        bf16x8 pf[2][2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            // running max on the RAW scores (scale > 0 commutes with max); scale and max folded into one FMA per element
            if constexpr (TAIL) {
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (kv0 + 16 * kt + 4 * g + r >= N) s[qt][kt][r] = -INFINITY;
            }
            float mx = m[qt];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {   // two v_max3_f32 per four scores
                mx = __builtin_fmaxf(__builtin_fmaxf(mx, s[qt][kt][0]), s[qt][kt][1]);
                mx = __builtin_fmaxf(__builtin_fmaxf(mx, s[qt][kt][2]), s[qt][kt][3]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float alpha = exp2_fast((m[qt] - mx) * c);  // first tile: exp2(-inf) = 0
            m[qt] = mx;
            const float mc = mx * c;
            // the softmax is VALU-bound (D = 32: ~4 VALU cycles per MFMA cycle): whole-vector expressions so that the
            // scale / shift and the row sum become packed v_pk_fma_f32 / v_pk_add_f32 (two floats per instruction)
            const f32x4 mc4 = {mc, mc, mc, mc};
            if constexpr (SUM_MFMA) {
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
                    const f32x4 t = fms4(s[qt][kt], c, mc4);
                    s[qt][kt] = (f32x4){exp2_fast(t[0]), exp2_fast(t[1]), exp2_fast(t[2]), exp2_fast(t[3])};
                }
                ol[qt][0] *= alpha;
            } else {
#if ATTN_SCALAR_VALU
            float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
#else
            f32x4 ps4 = {0.f, 0.f, 0.f, 0.f};
#endif
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const f32x4 t = fms4(s[qt][kt], c, mc4);
                const f32x4 pv = {exp2_fast(t[0]), exp2_fast(t[1]), exp2_fast(t[2]), exp2_fast(t[3])};
                s[qt][kt] = pv;
#if ATTN_DIAG & 1
                p0 = pv[0];
#elif ATTN_SCALAR_VALU
                p0 += pv[0]; p1 += pv[1]; p2 += pv[2]; p3 += pv[3];
#else
                ps4 += pv;
#endif
            }
#if ATTN_SCALAR_VALU
            const float ps = (p0 + p1) + (p2 + p3);
#else
            const float ps = (ps4[0] + ps4[1]) + (ps4[2] + ps4[3]);
#endif
            lsum[qt] = lsum[qt] * alpha + ps;
            }
            // (skipping these multiplies under a wave-uniform `alpha == 1` test -- bit-identical, 9 % fewer VALU instructions on tiles
            //  whose maximum did not move -- measured no gain: profiles/r04_experiments.txt)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) o[qt][dt] = scale4(o[qt][dt], alpha);
            pf[qt][0] = pack_acc(s[qt][0], s[qt][1]);
            pf[qt][1] = pack_acc(s[qt][2], s[qt][3]);
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const bf16x8 vf = frag_tr<D>(v_img, dt, u);
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) o[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[qt][u], o[qt][dt], 0, 0, 0);
            }
        if constexpr (SUM_MFMA) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) ol[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pf[qt][u], ol[qt], 0, 0, 0);
        }
    };
    const int nfull = N / 64;
    for (int it = 0; it < nfull; ++it) kv_tile(it, std::false_type{});
    if (nfull < ntile) kv_tile(nfull, std::true_type{});
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int q = q0 + 16 * qt + lq;
        float lt;
        if constexpr (SUM_MFMA) {
            lt = __shfl(ol[qt][0], lq, 64);               // lane lq (group 0) holds row 0 of the query's column
        } else {
            lt = lsum[qt];
            lt += __shfl_xor(lt, 16, 64);
            lt += __shfl_xor(lt, 32, 64);
        }
        if (q >= N) continue;
        const float inv = 1.f / lt;
        bf16_t* orow = out + ((size_t)b * N + q) * H * D + (size_t)h * D;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const f32x4 v = o[qt][dt] * inv;
            u32x2 pk = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
            *reinterpret_cast<u32x2*>(orow + 16 * dt + 4 * g) = pk;
        }
        if (g == 0) lse[((size_t)b * H + h) * N + q] = m[qt] * scale + logf(lt);  // m is a raw-score max
    }
}

// =============================================================================================== dQ
template <int D>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ out,
                                                          const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                          float* __restrict__ delta, bf16_t* __restrict__ dqkv, int N, int H,
                                                          float scale) {
    constexpr int KS = D / 32, DT = D / 16, TB = 64 * D * 2;
    __shared__ __attribute__((aligned(16))) unsigned char smem[3 * TB];  // K row | K transpose | V row
    unsigned char* k_row = smem;
    unsigned char* k_tr = smem + TB;
    unsigned char* v_row = smem + 2 * TB;
    // XCD-aware work order: workgroup ids go round-robin over the 8 XCDs, so consecutive ids (the blocks of ONE head) would
    // each pull that head's K / V through a different L2 (measured: 3.2x the algorithmic bytes at N = 1024).  xcd_remap
    // gives every XCD a contiguous run of (batch, head, block) items: a head's blocks, and the neighbouring head that shares
    // its 128-byte lines at D = 32, stay under one L2.
    const int gx = (N + 127) >> 7;
    const int wid = xcd_remap(blockIdx.x, gridDim.x);
    const int b = wid / (gx * H), h = (wid / gx) % H;
    const int q_blk = (wid % gx) * 128;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, g = l >> 4, lq = l & 15;
    const size_t rs = (size_t)3 * H * D, os = (size_t)H * D;
    const bf16_t* qb = qkv + (size_t)b * N * rs + (size_t)h * D;
    const bf16_t* kb = qb + (size_t)H * D;
    const bf16_t* vb = qb + (size_t)2 * H * D;
    const bf16_t* dob = dout + (size_t)b * N * os + (size_t)h * D;
    const bf16_t* ob = out + (size_t)b * N * os + (size_t)h * D;
    const int q0 = q_blk + 32 * w;

    bf16x8 qf[2][KS], dof[2][KS];
    float lse2[2], dlt[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int q = q0 + 16 * qt + lq;
        // delta[q] = sum_d O[q, d] dO[q, d] is computed HERE (it used to be a launch of its own in front of the two backward
        // kernels: 27 launches per step): the lane already holds its 8-element slices of dO, loads the matching slices of O,
        // and the four lane groups that share a query fold their partial sums; lane group 0 also stores it for the dK / dV
        // kernel, which runs behind this one on the stream.
        float part = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            qf[qt][ks] = scale_frag(frag_global(qb, rs, q, N, ks), scale * LOG2E);   // Q scale log2(e): S' = Q' K^T - lse log2(e) is the exp2 argument
            dof[qt][ks] = frag_global(dob, os, q, N, ks);
            const bf16x8 of = frag_global(ob, os, q, N, ks);
#pragma unroll
            for (int e = 0; e < 8; ++e) part += (float)of[e] * (float)dof[qt][ks][e];
        }
        part += __shfl_xor(part, 16, 64);
        part += __shfl_xor(part, 32, 64);
        lse2[qt] = q < N ? lse[((size_t)b * H + h) * N + q] * LOG2E : INFINITY;
        dlt[qt] = q < N ? part : 0.f;
        if (q < N && g == 0) delta[((size_t)b * H + h) * N + q] = part;
    }
    f32x4 dq[2][DT];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) dq[qt][dt] = (f32x4){0, 0, 0, 0};
    const int ntile = (N + 63) / 64;
    u32x4 rk[2], rv[2];
    tile_load<D>(kb, rs, 0, N, rk);
    tile_load<D>(vb, rs, 0, N, rv);
    auto kv_tile = [&](int it, auto tail_tag) {   // peeled like the forward: only the partial last tile masks keys
        constexpr bool TAIL = decltype(tail_tag)::value;
        const int kv0 = it * 64;
        __syncthreads();
        tile_store_row<D>(k_row, rk);
        tile_store_tr<D>(k_tr, rk);
        tile_store_row<D>(v_row, rv);
        __syncthreads();
        if (it + 1 < ntile) {
            tile_load<D>(kb, rs, kv0 + 64, N, rk);
            tile_load<D>(vb, rs, kv0 + 64, N, rv);
        }
        if (q0 >= N) return;
        // dP's accumulators start at -delta[q] (the lane's query: all four rows): the MFMA chain leaves dP - delta, and
        // dS = P (dP - delta) is ONE multiply per score instead of an add and a multiply (round 4; the softmax-gradient VALU work
        // bounds this kernel: 5.5 -> 4.5 issue slots per score).  Round 5: S's accumulators start at -lse[q] log2(e) and Q carries
        // scale log2(e), so the chain leaves the exp2 argument and the scale / shift FMA per score is gone too (4.5 -> 3.5).
        f32x4 s[2][4], dp[2][4];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                s[qt][kt] = (f32x4){-lse2[qt], -lse2[qt], -lse2[qt], -lse2[qt]};
#if ATTN_DIAG & 2
                dp[qt][kt] = (f32x4){0, 0, 0, 0};
#else
                dp[qt][kt] = (f32x4){-dlt[qt], -dlt[qt], -dlt[qt], -dlt[qt]};
#endif
            }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const bf16x8 kf = frag_row<D>(k_row, 16 * kt, ks);
                const bf16x8 vf = frag_row<D>(v_row, 16 * kt, ks);
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) {
                    s[qt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[qt][ks], s[qt][kt], 0, 0, 0);
                    dp[qt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, dof[qt][ks], dp[qt][kt], 0, 0, 0);
                }
            }
        if constexpr (TAIL) {   // last KV tile only: keys >= N get probability exp2(-inf) = 0
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (kv0 + 16 * kt + 4 * g + r >= N) s[qt][kt][r] = -INFINITY;
        }
        bf16x8 dsf[2][2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const f32x4 t = s[qt][kt];
                const f32x4 pv = {exp2_fast(t[0]), exp2_fast(t[1]), exp2_fast(t[2]), exp2_fast(t[3])};
                s[qt][kt] = mul4(pv, dp[qt][kt]);           // dS / scale = P (dP - delta); the factor is applied to dQ once at the end
            }
            dsf[qt][0] = pack_acc(s[qt][0], s[qt][1]);
            dsf[qt][1] = pack_acc(s[qt][2], s[qt][3]);
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const bf16x8 kt_f = frag_tr<D>(k_tr, dt, u);
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) dq[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt_f, dsf[qt][u], dq[qt][dt], 0, 0, 0);
            }
    };
    const int nfull = N / 64;
    for (int it = 0; it < nfull; ++it) kv_tile(it, std::false_type{});
    if (nfull < ntile) kv_tile(nfull, std::true_type{});
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int q = q0 + 16 * qt + lq;
        if (q >= N) continue;
        bf16_t* drow = dqkv + ((size_t)b * N + q) * rs + (size_t)h * D;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const f32x4 v = dq[qt][dt] * scale;
            u32x2 pk = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
            *reinterpret_cast<u32x2*>(drow + 16 * dt + 4 * g) = pk;
        }
    }
}

// =============================================================================================== dK, dV
// Workgroup = 128 keys (4 waves x 32 keys = 2 key tiles per wave); loops over 64-query tiles staged in LDS.
template <int D>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                           const float* __restrict__ lse, const float* __restrict__ delta,
                                                           bf16_t* __restrict__ dqkv, int N, int H, float scale) {
    constexpr int KS = D / 32, DT = D / 16, TB = 64 * D * 2;
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * TB + 2 * 64 * 4];  // Q row | Q tr | dO row | dO tr | lse2 | delta
    unsigned char* q_row = smem;
    unsigned char* q_tr = smem + TB;
    unsigned char* do_row = smem + 2 * TB;
    unsigned char* do_tr = smem + 3 * TB;
    float* s_lse = reinterpret_cast<float*>(smem + 4 * TB);
    float* s_dlt = s_lse + 64;
    // XCD-aware work order: workgroup ids go round-robin over the 8 XCDs, so consecutive ids (the blocks of ONE head) would
    // each pull that head's K / V through a different L2 (measured: 3.2x the algorithmic bytes at N = 1024).  xcd_remap
    // gives every XCD a contiguous run of (batch, head, block) items: a head's blocks, and the neighbouring head that shares
    // its 128-byte lines at D = 32, stay under one L2.
    const int gx = (N + 127) >> 7;
    const int wid = xcd_remap(blockIdx.x, gridDim.x);
    const int b = wid / (gx * H), h = (wid / gx) % H;
    const int k_blk = (wid % gx) * 128;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, g = l >> 4, lk = l & 15;
    const size_t rs = (size_t)3 * H * D, os = (size_t)H * D;
    const bf16_t* qb = qkv + (size_t)b * N * rs + (size_t)h * D;
    const bf16_t* kb = qb + (size_t)H * D;
    const bf16_t* vb = qb + (size_t)2 * H * D;
    const bf16_t* dob = dout + (size_t)b * N * os + (size_t)h * D;
    const int k0 = k_blk + 32 * w;

    bf16x8 kf[2][KS], vf[2][KS];   // second operands: col = key
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            kf[kt][ks] = scale_frag(frag_global(kb, rs, k0 + 16 * kt + lk, N, ks), scale * LOG2E);   // K scale log2(e): see the dQ kernel
            vf[kt][ks] = frag_global(vb, rs, k0 + 16 * kt + lk, N, ks);
        }
    f32x4 dk[2][DT], dv[2][DT];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) { dk[kt][dt] = (f32x4){0, 0, 0, 0}; dv[kt][dt] = (f32x4){0, 0, 0, 0}; }
    const int ntile = (N + 63) / 64;
    u32x4 rq[2], rd[2];
    tile_load<D>(qb, rs, 0, N, rq);
    tile_load<D>(dob, os, 0, N, rd);
    for (int it = 0; it < ntile; ++it) {
        const int q0 = it * 64;
        __syncthreads();
        tile_store_row<D>(q_row, rq);
        tile_store_tr<D>(q_tr, rq);
        tile_store_row<D>(do_row, rd);
        tile_store_tr<D>(do_tr, rd);
        if (threadIdx.x < 64) {
            const int q = q0 + threadIdx.x;
            s_lse[threadIdx.x] = q < N ? -lse[((size_t)b * H + h) * N + q] * LOG2E : -INFINITY;     // negated: S's initial accumulator
            s_dlt[threadIdx.x] = q < N ? -delta[((size_t)b * H + h) * N + q] : 0.f;   // negated: dP - delta as a packed add
        }
        __syncthreads();
        if (it + 1 < ntile) {
            tile_load<D>(qb, rs, q0 + 64, N, rq);
            tile_load<D>(dob, os, q0 + 64, N, rd);
        }
        if (k0 >= N) continue;
        // S[q][key], dP[q][key]: rows = queries 16*qt + 4g + r, col = key
        // (dP's accumulators start at -delta, S's at -lse log2(e) of their four query rows: see the dQ kernel; a query >= N starts
        //  at -inf: probability 0)
        f32x4 s[4][2], dp[4][2];
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) {
#if ATTN_DIAG & 2
            const f32x4 nd4 = {0, 0, 0, 0};
#else
            const f32x4 nd4 = *reinterpret_cast<const f32x4*>(s_dlt + 16 * qt + 4 * g);
#endif
            const f32x4 nl4 = *reinterpret_cast<const f32x4*>(s_lse + 16 * qt + 4 * g);
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) { s[qt][kt] = nl4; dp[qt][kt] = nd4; }
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int qt = 0; qt < 4; ++qt) {
                const bf16x8 qfr = frag_row<D>(q_row, 16 * qt, ks);
                const bf16x8 dfr = frag_row<D>(do_row, 16 * qt, ks);
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
                    s[qt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qfr, kf[kt][ks], s[qt][kt], 0, 0, 0);
                    dp[qt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dfr, vf[kt][ks], dp[qt][kt], 0, 0, 0);
                }
            }
        bf16x8 pf[2][2], dsf[2][2];  // [kt][u]: k index = query permutation of frag_tr
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                const f32x4 t = s[qt][kt];
                const f32x4 pv = {exp2_fast(t[0]), exp2_fast(t[1]), exp2_fast(t[2]), exp2_fast(t[3])};
                s[qt][kt] = pv;
                dp[qt][kt] = mul4(pv, dp[qt][kt]);                     // dS / scale = P (dP - delta) (scale applied to dK at the end)
            }
        }
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                pf[kt][u] = pack_acc(s[2 * u][kt], s[2 * u + 1][kt]);
                dsf[kt][u] = pack_acc(dp[2 * u][kt], dp[2 * u + 1][kt]);
            }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const bf16x8 dot = frag_tr<D>(do_tr, dt, u);
                const bf16x8 qt_f = frag_tr<D>(q_tr, dt, u);
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
                    dv[kt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dot, pf[kt][u], dv[kt][dt], 0, 0, 0);
                    dk[kt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt_f, dsf[kt][u], dk[kt][dt], 0, 0, 0);
                }
            }
    }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        const int key = k0 + 16 * kt + lk;
        if (key >= N) continue;
        bf16_t* krow = dqkv + ((size_t)b * N + key) * rs + (size_t)H * D + (size_t)h * D;
        bf16_t* vrow = krow + (size_t)H * D;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            u32x2 pk = {pack_bf2(dk[kt][dt][0] * scale, dk[kt][dt][1] * scale), pack_bf2(dk[kt][dt][2] * scale, dk[kt][dt][3] * scale)};
            u32x2 pv = {pack_bf2(dv[kt][dt][0], dv[kt][dt][1]), pack_bf2(dv[kt][dt][2], dv[kt][dt][3])};
            *reinterpret_cast<u32x2*>(krow + 16 * dt + 4 * g) = pk;
            *reinterpret_cast<u32x2*>(vrow + 16 * dt + 4 * g) = pv;
        }
    }
}


// =============================================================================================== single-pass backward (round 4)
// dQ, dK and dV in ONE kernel: S, P = exp2(.) and dS are computed ONCE per (query, key) pair instead of once in each of the two
// kernels above -- 5 matmuls instead of 7 and HALF the exp / softmax-gradient VALU work, which is what bounds the backward at
// D = 32 (profiles/r03_isa_budget.txt: vector issue : MFMA cycles = 2.1-2.5).
//
// Why this needs another structure: dK / dV want the KEY on the lane (the P / dS accumulators S[q][key] are then the second MFMA
// operands of dV^T = dO^T P and dK^T = Q^T dS), dQ wants the QUERY on the lane (dQ^T = K^T dS^T), and dQ sums over ALL keys.  So:
//   * one workgroup per (batch, head) walks every key block -- the dQ sum never leaves the CU: it lives in an fp32 LDS image
//     dQ^T[d][q] (pitch NP = 4 mod 64 floats: a wave's 16 consecutive queries x 4 row groups 4 NP apart fall on 64 distinct banks)
//     and is written out once at the end;
//   * 8 waves (two per SIMD, so that one wave's exp / VALU phase runs under the other's MFMAs); wave w owns keys
//     256 kb + 32 w .. + 31 of key block kb (K / V fragments and the dK / dV accumulators in registers, as in the dK / dV kernel);
//   * inner loop over 32-query tiles (Q / dO staged as row + transpose images, 2 x 2 x 32 x 2 D bytes);
//   * dS is TRANSPOSED through LDS: every wave stores its dS^T[key][q] (32 keys x 32 queries, bf16: four ds_write_b64 per lane) into
//     its 2 KiB staging image; one tile later -- behind the workgroup barrier the loop has anyway -- an OWNER wave per 16 x 16 tile
//     of dQ^T[d][q] reads all eight images back (ds_read_b64_tr_b16, exactly as frag_tr<32> reads any [k][n] image) as the second
//     operands of dQ^T += K^T[d][keys of the block] dS^T (K = 256 keys inside the MFMA chain: no cross-wave sum) and adds the tile
//     to the image with a plain read-modify-write: per key block every image element has exactly one writer.  (LDS float atomics
//     instead -- every wave adding its own 32-key partial -- were measured first: ds_add_f32 retires at ~190 cycles per wave
//     instruction, 2940 us against 440 us with the adds removed at N = 1024: profiles/r04_attn_bwd.txt.)
//     D = 64: the 8 tiles of a query tile have one owner wave each; D = 32: 4 tiles, waves 0-3 / 4-7 own them on even / odd tiles.
//     The K^T first operands (lane: row d, eight keys in frag_tr's k order, all 8 key groups of the block) are gathered from
//     global once per key block.
// LDS: 4 D NP + 256 D + 16 KiB + 4 NP + 128 bytes <= 160 KiB  ->  D = 32: N <= 1024 (D = 64 would fit N <= 448 but is not
// instantiated: see attn_bwd_impl).  delta = rowsum(dO * O) is computed in the prologue (and stored, as before).
template <int D, int NP>
__global__ __launch_bounds__(512) void attn_bwd_fused_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ out,
                                                             const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                             float* __restrict__ delta, bf16_t* __restrict__ dqkv, int N, int H,
                                                             float scale) {
    constexpr int KS = D / 32, DT = D / 16, QT = 32, IMG = QT * D * 2, CPR = D / 8;
    constexpr int TILES = 2 * DT;                                      // 16 x 16 tiles of dQ^T per query tile: 4 (D = 32) or 8
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * D * NP + 4 * IMG + 8 * 2048 + 4 * NP + 4 * QT];
    float* dq_img = reinterpret_cast<float*>(smem);                    // dQ^T[d][q], pitch NP
    unsigned char* q_row = smem + 4 * D * NP;
    unsigned char* q_tr = q_row + IMG;
    unsigned char* do_row = q_tr + IMG;
    unsigned char* do_tr = do_row + IMG;
    unsigned char* stage_all = do_tr + IMG;                            // 8 x [32 keys][32 q] bf16
    float* s_dlt = reinterpret_cast<float*>(stage_all + 8 * 2048);     // -delta[q], q < NP
    float* s_lse = s_dlt + NP;                                         // lse[q] log2(e) of the current query tile

    const int wid = xcd_remap(blockIdx.x, gridDim.x);
    const int b = wid / H, h = wid % H;
    const int t = threadIdx.x, w = t >> 6, l = t & 63, g = l >> 4, lk = l & 15;
    const size_t rs = (size_t)3 * H * D, os = (size_t)H * D;
    const bf16_t* qb = qkv + (size_t)b * N * rs + (size_t)h * D;
    const bf16_t* kb = qb + (size_t)H * D;
    const bf16_t* vb = qb + (size_t)2 * H * D;
    const bf16_t* dob = dout + (size_t)b * N * os + (size_t)h * D;
    const bf16_t* ob = out + (size_t)b * N * os + (size_t)h * D;
    const float* lse_bh = lse + ((size_t)b * H + h) * N;
    unsigned char* stage = stage_all + w * 2048;

    // ---- prologue: dQ image = 0; delta[q] = sum_d O[q][d] dO[q][d] (CPR consecutive lanes share a row)
    for (int i = t; i < D * NP; i += 512) dq_img[i] = 0.f;
    for (int i = t; i < NP; i += 512) s_dlt[i] = 0.f;
    __syncthreads();
    for (int base = 0; base < N * CPR; base += 512) {
        const int i = base + t, row = i / CPR, c = i % CPR;
        float part = 0.f;
        if (row < N) {
            const u32x4 ov = *reinterpret_cast<const u32x4*>(ob + (size_t)row * os + c * 8);
            const u32x4 dv4 = *reinterpret_cast<const u32x4*>(dob + (size_t)row * os + c * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                part += __uint_as_float(ov[e] << 16) * __uint_as_float(dv4[e] << 16);
                part += __uint_as_float(ov[e] & 0xffff0000u) * __uint_as_float(dv4[e] & 0xffff0000u);
            }
        }
#pragma unroll
        for (int o = 1; o < CPR; o <<= 1) part += __shfl_xor(part, o, 64);
        if (row < N && c == 0) {
            s_dlt[row] = -part;
            delta[((size_t)b * H + h) * N + row] = part;
        }
    }

    const float c = scale * LOG2E;
    const int ntile = (N + QT - 1) / QT;
    const int nkb = (N + 255) / 256;
    // Q / dO tile loader: thread t < 8 D moves one 16-byte chunk (t < 4 D: Q, else dO); threads < 32 also carry the tile's lse
    const bool ld_on = t < 8 * D;
    const bool ld_do = t >= 4 * D;
    const int ld_idx = t % (4 * D), ld_row = ld_idx / CPR, ld_c = ld_idx % CPR;
    const bf16_t* ld_base = ld_do ? dob : qb;
    const size_t ld_stride = ld_do ? os : rs;
    // Both fetches are UNCONDITIONAL loads of a clamped row (rows >= N are zeroed / set to +inf when they are consumed): a load
    // under a lane predicate sits behind an exec branch, the compiler can then no longer count it, and every wait in the loop
    // degrades to vmcnt(0) -- which would also wait for the tiles fetched last.
    auto tile_fetch = [&](int q0) -> u32x4 {
        const int row = min(q0 + ld_row, N - 1);
        return *reinterpret_cast<const u32x4*>(ld_base + (size_t)row * ld_stride + ld_c * 8);
    };
    auto lse_fetch = [&](int q0) -> float { return lse_bh[min(q0 + (t & (QT - 1)), N - 1)]; };     // (RAW value: no arithmetic here)
    // owner of dQ^T tile (dt_o, qt_o) of a query tile
    const int tile_o = TILES == 8 ? w : (w & 3);
    const int dt_o = tile_o >> 1, qt_o = tile_o & 1;

    for (int kblk = 0; kblk < nkb; ++kblk) {
        const int k0 = kblk * 256 + 32 * w;
        const bool active = k0 < N;                                    // wave-uniform
        bf16x8 kf[2][KS], vf[2][KS];                                   // second operands: col = key
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                kf[kt][ks] = frag_global(kb, rs, k0 + 16 * kt + lk, N, ks);
                vf[kt][ks] = frag_global(vb, rs, k0 + 16 * kt + lk, N, ks);
            }
        // K^T first operands of the owner's dQ product, one per 32-key group j of the block: lane (row d = 16 dt_o + lk, group g)
        // holds keys 4 g + i (i < 4) and 16 + 4 g + (i - 4) of the group: frag_tr's k order, in which the staged dS^T comes back
        bf16x8 ktf[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            s16x8 r;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int key = kblk * 256 + 32 * j + (i < 4 ? 4 * g + i : 16 + 4 * g + (i - 4));
                r[i] = key < N ? (short)kb[(size_t)key * rs + 16 * dt_o + lk] : (short)0;
            }
            ktf[j] = __builtin_bit_cast(bf16x8, r);
        }
        if (!active) {            // no keys left for this wave in the last block: its staging image must read as zeros
            *reinterpret_cast<u32x4*>(stage + 32 * l) = (u32x4){0, 0, 0, 0};
            *reinterpret_cast<u32x4*>(stage + 32 * l + 16) = (u32x4){0, 0, 0, 0};
        }
        f32x4 dk[2][DT], dv[2][DT];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) { dk[kt][dt] = (f32x4){0, 0, 0, 0}; dv[kt][dt] = (f32x4){0, 0, 0, 0}; }
        const bool key_tail = k0 + 32 > N;                              // wave-uniform: some of this wave's keys do not exist
        // dQ^T tile (dt_o, qt_o) of query tile `it` from the eight staged dS^T images, added to the image (one writer per element)
        auto dq_part = [&](int it) {
            if (TILES == 4 && ((it & 1) != (w >> 2))) return;
            f32x4 a = {0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < 8; ++j) a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktf[j], frag_tr<32>(stage_all + j * 2048, qt_o, 0), a, 0, 0, 0);
            float* col = dq_img + it * QT + 16 * qt_o + lk + (16 * dt_o + 4 * g) * NP;
#pragma unroll
            for (int r = 0; r < 4; ++r) col[r * NP] += a[r];
        };
        // Q / dO / lse of the next PF query tiles are in flight: with one workgroup per CU nobody else hides a miss, and a block's
        // Q / dO re-reads come from the Infinity Cache (32 CUs x 256 KiB of operands per XCD do not fit its 4 MiB L2): one tile
        // ahead (~1000 cycles of work) left ~2000 cycles of every tile exposed.  The loop is unrolled by PF so that every slot is a
        // fixed set of registers (rotating the slots by moves would read -- i.e. wait for -- the newest loads).
        constexpr int PF = 3;
        u32x4 nxt[PF];
        float lse_nxt[PF];
#pragma unroll
        for (int i = 0; i < PF; ++i) { nxt[i] = tile_fetch(i * QT); lse_nxt[i] = lse_fetch(i * QT); }
        auto tile_body = [&](int it, u32x4& slot, float& lse_slot) {
            const int q0 = it * QT;
            __syncthreads();                // tile it - 1 is done everywhere: its images are free, its eight dS^T images complete
            if (ld_on) {
                unsigned char* row_img = ld_do ? do_row : q_row;
                unsigned char* tr_img = ld_do ? do_tr : q_tr;
                const u32x4 v = q0 + ld_row < N ? slot : (u32x4){0, 0, 0, 0};
                *reinterpret_cast<u32x4*>(row_img + ld_row * (2 * D) + ((ld_c ^ row_swz<D>(ld_row)) << 4)) = v;
                *reinterpret_cast<u32x4*>(tr_img + ld_row * (2 * D) + ((((ld_c >> 1) ^ tr_swz<D>(ld_row)) << 5) | ((ld_c & 1) << 4))) = v;
            }
            if (t < QT) s_lse[t] = q0 + t < N ? lse_slot * LOG2E : INFINITY;     // query >= N: probability 0
            slot = tile_fetch(q0 + PF * QT);
            lse_slot = lse_fetch(q0 + PF * QT);
            if (it > 0) dq_part(it - 1);
            __syncthreads();                // images of tile `it` ready; the dS^T images of tile it - 1 have been consumed
            if (!active) return;
            // S[q][key], dP[q][key] - delta[q]: rows = queries 16 qt + 4 g + r, col = key 16 kt + lk
            f32x4 s[2][2], dp[2][2];
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                const f32x4 nd4 = *reinterpret_cast<const f32x4*>(s_dlt + q0 + 16 * qt + 4 * g);     // dP starts at -delta
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) { s[qt][kt] = (f32x4){0, 0, 0, 0}; dp[qt][kt] = nd4; }
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) {
                    const bf16x8 qfr = frag_row<D>(q_row, 16 * qt, ks);
                    const bf16x8 dfr = frag_row<D>(do_row, 16 * qt, ks);
#pragma unroll
                    for (int kt = 0; kt < 2; ++kt) {
                        s[qt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qfr, kf[kt][ks], s[qt][kt], 0, 0, 0);
                        dp[qt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dfr, vf[kt][ks], dp[qt][kt], 0, 0, 0);
                    }
                }
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                const f32x4 l4 = *reinterpret_cast<const f32x4*>(s_lse + 16 * qt + 4 * g);
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
                    const f32x4 tt = fms4(s[qt][kt], c, l4);
                    const f32x4 pv = {exp2_fast(tt[0]), exp2_fast(tt[1]), exp2_fast(tt[2]), exp2_fast(tt[3])};
                    s[qt][kt] = pv;
                    dp[qt][kt] = mul4(pv, dp[qt][kt]);                 // dS / scale (applied to dK and dQ at the end)
                }
            }
            if (key_tail) {       // a key >= N has K = V = 0: S = 0, P = exp2(-lse) != 0 -- harmless for dK / dV (those rows are never
                                  // stored) but dS = -P delta would reach dQ: zero it (one wave of the last key block only)
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
                    if (k0 + 16 * kt + lk >= N) {
#pragma unroll
                        for (int qt = 0; qt < 2; ++qt) dp[qt][kt] = (f32x4){0, 0, 0, 0};
                    }
            }
            // dS^T into the wave's staging image T[key][q] (4 consecutive queries = 8 bytes per store)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                const int key = 16 * kt + lk;
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) {
                    const u32x2 pk = {pack_bf2(dp[qt][kt][0], dp[qt][kt][1]), pack_bf2(dp[qt][kt][2], dp[qt][kt][3])};
                    *reinterpret_cast<u32x2*>(stage + key * 64 + ((qt ^ tr_swz<32>(key)) << 5) + 8 * g) = pk;
                }
            }
            // dV^T[d][key] += dO^T[d][q] P[q][key],  dK^T[d][key] += Q^T[d][q] dS[q][key]   (k = the tile's 32 queries)
            bf16x8 pf[2], dsf[2];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                pf[kt] = pack_acc(s[0][kt], s[1][kt]);
                dsf[kt] = pack_acc(dp[0][kt], dp[1][kt]);
            }
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const bf16x8 dot = frag_tr<D>(do_tr, dt, 0);
                const bf16x8 qt_f = frag_tr<D>(q_tr, dt, 0);
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
                    dv[kt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dot, pf[kt], dv[kt][dt], 0, 0, 0);
                    dk[kt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt_f, dsf[kt], dk[kt][dt], 0, 0, 0);
                }
            }
        };
        for (int it = 0; it < ntile; it += PF) {
            tile_body(it, nxt[0], lse_nxt[0]);
            if (it + 1 < ntile) tile_body(it + 1, nxt[1], lse_nxt[1]);
            if (it + 2 < ntile) tile_body(it + 2, nxt[2], lse_nxt[2]);
        }
        __syncthreads();                    // the last tile's dS^T images are complete
        dq_part(ntile - 1);
        if (active) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                const int key = k0 + 16 * kt + lk;
                if (key >= N) continue;
                bf16_t* krow = dqkv + ((size_t)b * N + key) * rs + (size_t)H * D + (size_t)h * D;
                bf16_t* vrow = krow + (size_t)H * D;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    u32x2 pk = {pack_bf2(dk[kt][dt][0] * scale, dk[kt][dt][1] * scale), pack_bf2(dk[kt][dt][2] * scale, dk[kt][dt][3] * scale)};
                    u32x2 pv = {pack_bf2(dv[kt][dt][0], dv[kt][dt][1]), pack_bf2(dv[kt][dt][2], dv[kt][dt][3])};
                    *reinterpret_cast<u32x2*>(krow + 16 * dt + 4 * g) = pk;
                    *reinterpret_cast<u32x2*>(vrow + 16 * dt + 4 * g) = pv;
                }
            }
        }
        __syncthreads();                    // the dS^T images are free (next key block stages / zeroes them); the image sums are final
    }
    for (int i = t; i < N * CPR; i += 512) {
        const int q = i / CPR, ch = i % CPR;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = dq_img[(8 * ch + e) * NP + q] * scale;
        const u32x4 pk = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
        *reinterpret_cast<u32x4*>(dqkv + ((size_t)b * N + q) * rs + (size_t)h * D + 8 * ch) = pk;
    }
}

}  // namespace

extern "C" int mh_attn_fwd(const void* qkv, void* out, float* lse, int B, int N, int H, int D, float scale, void* stream) {
    MH_CHECK_ARG(qkv && out && lse, "mh_attn_fwd: null pointer");
    MH_CHECK_ARG(B > 0 && N > 0 && H > 0 && (D == 32 || D == 64), "mh_attn_fwd: unsupported shape B=%d N=%d H=%d D=%d", B, N, H, D);
    MH_CHECK_ARG(H <= 65535 && B <= 65535, "mh_attn_fwd: grid limit");
    dim3 grid(ceil_div(N, 128) * H * B), block(256);
    if (D == 64) hipLaunchKernelGGL(attn_fwd_kernel<64>, grid, block, 0, (hipStream_t)stream, (const bf16_t*)qkv, (bf16_t*)out, lse, N, H, scale);
    else hipLaunchKernelGGL(attn_fwd_kernel<32>, grid, block, 0, (hipStream_t)stream, (const bf16_t*)qkv, (bf16_t*)out, lse, N, H, scale);
    MH_LAUNCH_CHECK();
    return 0;
}

// variant: 0 = the library's rule, 1 = the two kernels (dQ, then dK / dV), 2 = the single-pass kernel (-2 when the shape does not fit its LDS)
static int attn_bwd_impl(int variant, const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv,
                         int B, int N, int H, int D, float scale, void* stream) {
    MH_CHECK_ARG(qkv && out && dout && lse && delta && dqkv, "mh_attn_bwd: null pointer");
    MH_CHECK_ARG(B > 0 && N > 0 && H > 0 && (D == 32 || D == 64), "mh_attn_bwd: unsupported shape B=%d N=%d H=%d D=%d", B, N, H, D);
    MH_CHECK_ARG(variant >= 0 && variant <= 2, "mh_attn_bwd: variant %d", variant);
    hipStream_t s = (hipStream_t)stream;
    // the single-pass kernel is instantiated for D = 32 only: at D = 64 its register budget (256 per lane at two waves per SIMD:
    // K / V fragments, 16 dK / dV accumulators, the owner's eight K^T operands, three tiles of prefetch) spills, and it measured
    // 8-50 % slower than the two kernels on every D = 64 shape of the step (profiles/r04_attn_bwd.txt)
    const bool fits = D == 32 && N <= 1024;
    if (variant == 2 && !fits) return -2;
    const bool fused = variant == 2 || (variant == 0 && fits && MH_ATTN_FUSED_BWD_DEFAULT);
    if (fused) {
        dim3 grid(H * B), block(512);
#define FUSED(DD, NP) hipLaunchKernelGGL((attn_bwd_fused_kernel<DD, NP>), grid, block, 0, s, (const bf16_t*)qkv, (const bf16_t*)out, \
                                         (const bf16_t*)dout, lse, delta, (bf16_t*)dqkv, N, H, scale)
        if (N <= 512) FUSED(32, 516); else FUSED(32, 1028);
#undef FUSED
        MH_LAUNCH_CHECK();
        return 0;
    }
    dim3 grid(ceil_div(N, 128) * H * B), block(256);
    if (D == 64) {
        hipLaunchKernelGGL(attn_bwd_dq_kernel<64>, grid, block, 0, s, (const bf16_t*)qkv, (const bf16_t*)out, (const bf16_t*)dout, lse, delta, (bf16_t*)dqkv, N, H, scale);
        hipLaunchKernelGGL(attn_bwd_dkv_kernel<64>, grid, block, 0, s, (const bf16_t*)qkv, (const bf16_t*)dout, lse, delta, (bf16_t*)dqkv, N, H, scale);
    } else {
        hipLaunchKernelGGL(attn_bwd_dq_kernel<32>, grid, block, 0, s, (const bf16_t*)qkv, (const bf16_t*)out, (const bf16_t*)dout, lse, delta, (bf16_t*)dqkv, N, H, scale);
        hipLaunchKernelGGL(attn_bwd_dkv_kernel<32>, grid, block, 0, s, (const bf16_t*)qkv, (const bf16_t*)dout, lse, delta, (bf16_t*)dqkv, N, H, scale);
    }
    MH_LAUNCH_CHECK();
    return 0;
}

extern "C" int mh_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv,
                           int B, int N, int H, int D, float scale, void* stream) {
    return attn_bwd_impl(0, qkv, out, dout, lse, delta, dqkv, B, N, H, D, scale, stream);
}

extern "C" int mh_attn_bwd_variant(int variant, const void* qkv, const void* out, const void* dout, const float* lse, float* delta,
                                   void* dqkv, int B, int N, int H, int D, float scale, void* stream) {
    return attn_bwd_impl(variant, qkv, out, dout, lse, delta, dqkv, B, N, H, D, scale, stream);
}
