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
//   bit 0: forward without the row-sum adds;  bit 1: backward without the scale / lse FMA and without the "- delta" add;
//   bit 2: forward (D = 32) without the four denominator MFMAs per tile
#ifndef ATTN_DIAG
#define ATTN_DIAG 0
#endif
// ATTN_LAZY_MAX = 1 (round 6): the forward does NOT follow a running maximum while the scores stay in range.  Every query has a REFERENCE
// exponent m2 (log2 units, 0 at first); the exp2 argument of a score is s * scale * log2(e) - m2 (one FMA, as before), and as long as every
// lane's largest argument of the tile lies within ATTN_LAZY_RANGE of it the probabilities are exp2 of that as it stands: no cross-lane
// maximum, no exp2 for the correction factor, no rescaling of O and of the denominator (fp32 sums and bf16 probabilities keep their
// RELATIVE precision whatever the common factor 2^-m2 is; the normalisation removes it, lse = (m2 + log2 l) ln 2 carries it).  A tile
// that leaves the range (wave-uniform branch) moves the reference to its row maximum the classic way -- up at any tile, down only at the
// first one.  Same box, scripts/bench_attn.py: forward D = 32, N = 1024 155 -> 122 us, N = 400 34.9 -> 31.4; D = 64, N = 256 18.1 -> 16.2.
// What it costs: the row's largest probability is no longer exactly 1.0 but a bf16-rounded 2^t, so lse / the output carry up to 2^-9
// relative from it (observed lse 2e-3 -> 3e-3, output 1.9e-2 -> 2.3e-2 on tests/test_kernels_gpu.py's inputs).  Folding scale x log2(e)
// into Q (bf16) to drop the FMA as well measured no faster and costs 1.5e-2 on lse: not kept.
// 0 (MH_ATTN_FLAGS="-DATTN_LAZY_MAX=0") keeps the classic online softmax (A/B aid).
#ifndef ATTN_LAZY_MAX
#define ATTN_LAZY_MAX 1
#endif
#ifndef ATTN_LAZY_RANGE
#define ATTN_LAZY_RANGE 24.f
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
        m[qt] = ATTN_LAZY_MAX ? 0.f : -INFINITY; lsum[qt] = 0.f;      // lazy: the reference exponent (log2 units), else the raw-score maximum
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
            for (int kt = 0; kt < 4; ++kt) {
                s[qt][kt] = (f32x4){0, 0, 0, 0};
            }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const bf16x8 kf = frag_row<D>(k_img, 16 * kt, ks);
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) s[qt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[qt][ks], s[qt][kt], 0, 0, 0);
            }
        bf16x8 pf[2][2];
#if ATTN_LAZY_MAX
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            {
                const f32x4 m4 = {m[qt], m[qt], m[qt], m[qt]};
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) s[qt][kt] = fms4(s[qt][kt], c, m4);     // the exp2 argument relative to the reference
            }
            if constexpr (TAIL) {   // keys >= N: probability exp2(-inf) = 0
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (kv0 + 16 * kt + 4 * g + r >= N) s[qt][kt][r] = -INFINITY;
            }
            float mx = s[qt][0][0];                                  // this lane's largest exp2 argument of the tile
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                mx = __builtin_fmaxf(__builtin_fmaxf(mx, s[qt][kt][0]), s[qt][kt][1]);
                mx = __builtin_fmaxf(__builtin_fmaxf(mx, s[qt][kt][2]), s[qt][kt][3]);
            }
            // Wave-uniform: does any lane leave the range?  Upwards at any tile (the reference follows the row maximum, as the classic
            // form does: alpha <= 1).  Downwards ONLY at the first tile, where the reference is still the arbitrary 0 and nothing has been
            // accumulated: afterwards it lies within the range of a score that was really there, a tile far below it adds ~0 and is
            // left alone (moving down with mass accumulated would scale O up by 2^|shift|).  A NaN score compares false and flows
            // through exp2 into the output, as it does in the reference.
            const bool first = it == 0;
            const bool leave = mx > ATTN_LAZY_RANGE || (first && mx < -ATTN_LAZY_RANGE);
            if (__builtin_amdgcn_ballot_w64(leave) != 0) {
                mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                mx = first ? fmaxf(mx, -3.0e38f) : fmaxf(mx, 0.f);  // (a first tile has at least one valid key per row: mx is finite)
                const float alpha = first ? 1.f : exp2_fast(-mx);    // first tile: O and the denominator are still zero
                m[qt] += mx;
                const f32x4 mx4 = {mx, mx, mx, mx};
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) s[qt][kt] = fms4(s[qt][kt], 1.f, mx4);
                if constexpr (SUM_MFMA) ol[qt][0] *= alpha;
                else lsum[qt] *= alpha;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) o[qt][dt] = scale4(o[qt][dt], alpha);
            }
            float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const f32x4 t = s[qt][kt];
                const f32x4 pv = {exp2_fast(t[0]), exp2_fast(t[1]), exp2_fast(t[2]), exp2_fast(t[3])};
                s[qt][kt] = pv;
                if constexpr (!SUM_MFMA) { p0 += pv[0]; p1 += pv[1]; p2 += pv[2]; p3 += pv[3]; }
            }
            if constexpr (!SUM_MFMA) lsum[qt] += (p0 + p1) + (p2 + p3);
            pf[qt][0] = pack_acc(s[qt][0], s[qt][1]);
            pf[qt][1] = pack_acc(s[qt][2], s[qt][3]);
        }
#else
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
#endif
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const bf16x8 vf = frag_tr<D>(v_img, dt, u);
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) o[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[qt][u], o[qt][dt], 0, 0, 0);
            }
#if !(ATTN_DIAG & 4)       // (bit 2, WRONG results: the forward without the denominator MFMAs -- what are 4 of its 20 MFMAs per tile worth?)
        if constexpr (SUM_MFMA) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) ol[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pf[qt][u], ol[qt], 0, 0, 0);
        }
#else
        if constexpr (SUM_MFMA) { ol[0][0] = 1.f; ol[1][0] = 1.f; }
#endif
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
#if ATTN_LAZY_MAX
        if (g == 0) lse[((size_t)b * H + h) * N + q] = m[qt] * 0.6931471805599453f + logf(lt);  // m is the reference exponent, log2 units
#else
        if (g == 0) lse[((size_t)b * H + h) * N + q] = m[qt] * scale + logf(lt);  // m is a raw-score max
#endif
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
    // lse / delta of the NEXT query tile travel with its Q / dO rows (round 5): fetched between the two barriers of the tile that
    // needs them, as before, every workgroup sat through one dependent global-load latency per query tile with all four waves at
    // the barrier (16 times per workgroup at N = 1024)
    const float* const lse_bh = lse + ((size_t)b * H + h) * N;
    const float* const dlt_bh = delta + ((size_t)b * H + h) * N;
    float r_lse = 0.f, r_dlt = 0.f;       // (threads 0-63: the tile's query threadIdx.x)
    auto stat_load = [&](int q0) {
        if (threadIdx.x < 64) {
            const int q = min(q0 + (int)threadIdx.x, N - 1);        // clamped: unconditional loads, validity applied when stored
            r_lse = lse_bh[q];
            r_dlt = dlt_bh[q];
        }
    };
    stat_load(0);
    for (int it = 0; it < ntile; ++it) {
        const int q0 = it * 64;
        __syncthreads();
        tile_store_row<D>(q_row, rq);
        tile_store_tr<D>(q_tr, rq);
        tile_store_row<D>(do_row, rd);
        tile_store_tr<D>(do_tr, rd);
        if (threadIdx.x < 64) {
            const bool valid = q0 + (int)threadIdx.x < N;
            s_lse[threadIdx.x] = valid ? -r_lse * LOG2E : -INFINITY;     // negated: S's initial accumulator
            s_dlt[threadIdx.x] = valid ? -r_dlt : 0.f;                   // negated: dP's initial accumulator
        }
        __syncthreads();
        if (it + 1 < ntile) {
            tile_load<D>(qb, rs, q0 + 64, N, rq);
            tile_load<D>(dob, os, q0 + 64, N, rd);
            stat_load(q0 + 64);
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


// (Round 4's single-pass backward -- one workgroup per (batch, head), dQ summed in an fp32 LDS image -- measured at parity with the two
// kernels at N = 1024 and slower below (profiles/r04_attn_bwd.txt) and was removed in round 5.)

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

extern "C" int mh_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv,
                           int B, int N, int H, int D, float scale, void* stream) {
    MH_CHECK_ARG(qkv && out && dout && lse && delta && dqkv, "mh_attn_bwd: null pointer");
    MH_CHECK_ARG(B > 0 && N > 0 && H > 0 && (D == 32 || D == 64), "mh_attn_bwd: unsupported shape B=%d N=%d H=%d D=%d", B, N, H, D);
    hipStream_t s = (hipStream_t)stream;
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
