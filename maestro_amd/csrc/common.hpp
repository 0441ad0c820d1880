// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels of the MAE hot path.
// Wave = 64 lanes everywhere; no CUDA compatibility paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef uint16_t bf16_t;  // raw bf16 bits in global memory

// ----------------------------------------------------------------------------- error plumbing (host)
extern thread_local char mh_err_buf[512];
int mh_fail(int code, const char* fmt, ...);
#define MH_CHECK_ARG(cond, ...) do { if (!(cond)) return mh_fail(-1, __VA_ARGS__); } while (0)
#define MH_LAUNCH_CHECK() do { hipError_t e__ = hipGetLastError(); \
    if (e__ != hipSuccess) return mh_fail((int)e__, "%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e__)); } while (0)

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

// ----------------------------------------------------------------------------- bf16 <-> f32 (device)
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {  // round-to-nearest-even; NaN stays NaN via the cast path
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {   // ONE v_cvt_pk_bf16_f32 (round-to-nearest-even)
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}

// ----------------------------------------------------------------------------- wave / block reductions
// (Round 3 tried DPP reductions here -- row_shr:1,2,4,8 + row_bcast:15 / :31, one v_add_f32_dpp per step instead of six
//  ds_bpermute round trips: the LayerNorm / loss kernels take the same time with either form (scripts/bench_ln.py: they are bound
//  by their loads, not by the reductions), so the shuffle form stays.)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// Absmax bookkeeping of the fp8 path.  An absmax is held as a float whose BITS are compared as unsigned integers: on |x| the
// integer order is the float order, with NaN (0x7fc00000 and up) above +inf above every finite value -- so a non-finite
// element survives every fold, wave reduction and atomic max (fmaxf would drop a NaN and report a clean tensor) and reaches
// mh_fp8_update_scales, which poisons the tensor's scale with it: the divergence shows in the loss instead of being clamped
// to +-448 operands.
__device__ __forceinline__ float amax_fold(float acc, float x) {
    return __uint_as_float(max(__float_as_uint(acc), __float_as_uint(x) & 0x7fffffffu));
}
__device__ __forceinline__ float wave_amax(float v) {
    uint32_t u = __float_as_uint(v);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) u = max(u, (uint32_t)__shfl_xor((int)u, o, 64));
    return __uint_as_float(u);
}
__device__ __forceinline__ bool amax_nonzero(float v) { return __float_as_uint(v) != 0u; }
// Block-wide sum for blockDim.x = 64*NW threads; `red` is NW floats of LDS. Result broadcast to all threads.
template <int NW>
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) t += red[i];
    return t;
}

// GELU (exact-erf form of nn.GELU) and its derivative with erf from Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7, far
// below the bf16 output resolution): one v_exp + one v_rcp + 5 FMAs instead of the ~40-instruction erff.  The
// exp(-x^2/2) factor is shared between the CDF and the PDF term of the derivative.
__device__ __forceinline__ void gelu_cdf_pdf(float x, float& cdf, float& pdf) {
    const float ax = fabsf(x) * 0.70710678118654752f;                  // |x| / sqrt(2)
    const float t = __builtin_amdgcn_rcpf(1.f + 0.3275911f * ax);
    const float e = __builtin_amdgcn_exp2f(-ax * ax * 1.4426950408889634f);  // exp(-x^2/2)
    float poly = 1.061405429f;
    poly = poly * t - 1.453152027f;
    poly = poly * t + 1.421413741f;
    poly = poly * t - 0.284496736f;
    poly = poly * t + 0.254829592f;
    const float erf_abs = 1.f - poly * t * e;                          // erf(|x|/sqrt2)
    cdf = 0.5f * (1.f + copysignf(erf_abs, x));
    pdf = 0.3989422804014327f * e;
}
// Four elements at once for the GEMM epilogues, where this arithmetic is NOT hidden: at K = 768 the fc1 GEMM ran 49 us with a plain
// bias epilogue and 67 us with the GELU (scripts/bench_gelu_cost.py); inside the persistent kernel's main loop the arithmetic
// alone still adds 30 % (profiles/r03_gemm_pp_diag.txt: MFMA and VALU share the SIMD's issue port).
//   z = |x| / sqrt2, t = 1 / (1 + p z), h = 0.5 poly(t) exp(-z^2) = 1 - Phi(|x|);  Phi(x) = x >= 0 ? 1 - h : h
// MH_GELU_TERMS = 3 (default): Abramowitz-Stegun 7.1.25, |error| <= 2.5e-5 on erf (far below what the bf16 activation -- 4e-3
// relative -- and the byte-coded derivative -- 2.5e-3 absolute -- resolve); 5 (-DMH_GELU_TERMS=5 through MH_BUILD_FLAGS): A-S 7.1.26,
// |error| <= 1.5e-7, two FMAs per element more.  Round 3 measured the five-term form again, now inside the persistent kernel
// (profiles/r03_gemm_pp_diag.txt): the fc1 GEMMs take 6.6 ... 8.6 % longer (full / main-loop-only 1.44 / 1.40 / 1.62 / 1.83 against
// 1.34 / 1.31 / 1.50 / 1.69 on the four fc1 shapes), ~0.13 ms = 0.7 % of the C3 step: the epilogue's arithmetic is NOT free even
// when interleaved with the next tile's MFMAs, so the three-term form stays.
#ifndef MH_GELU_TERMS
#define MH_GELU_TERMS 3
#endif
__device__ __forceinline__ void gelu_cdf_pdf4(const f32x4 x, f32x4& cdf, f32x4& pdf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float e = __builtin_amdgcn_exp2f((x[i] * x[i]) * (-0.5f * 1.4426950408889634f));      // exp(-x^2 / 2)
#if MH_GELU_TERMS == 3
        const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(fabsf(x[i]), 0.47047f * 0.70710678118654752f, 1.f));
        const float poly = __builtin_fmaf(__builtin_fmaf(0.5f * 0.7478556f, t, 0.5f * -0.0958798f), t, 0.5f * 0.3480242f);
#else
        const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(fabsf(x[i]), 0.3275911f * 0.70710678118654752f, 1.f));
        float poly = __builtin_fmaf(0.5f * 1.061405429f, t, 0.5f * -1.453152027f);
        poly = __builtin_fmaf(poly, t, 0.5f * 1.421413741f);
        poly = __builtin_fmaf(poly, t, 0.5f * -0.284496736f);
        poly = __builtin_fmaf(poly, t, 0.5f * 0.254829592f);
#endif
        const float h = (poly * t) * e;
        cdf[i] = x[i] >= 0.f ? 1.f - h : h;
        pdf[i] = e * 0.3989422804014327f;
    }
}
__device__ __forceinline__ f32x4 gelu_erf4(const f32x4 x) {
    f32x4 cdf, pdf;
    gelu_cdf_pdf4(x, cdf, pdf);
    return x * cdf;
}
__device__ __forceinline__ f32x4 gelu_erf_grad4(const f32x4 x) {
    f32x4 cdf, pdf;
    gelu_cdf_pdf4(x, cdf, pdf);
    return x * pdf + cdf;
}
__device__ __forceinline__ float gelu_erf(float x) {
    float cdf, pdf;
    gelu_cdf_pdf(x, cdf, pdf);
    return x * cdf;
}
__device__ __forceinline__ float gelu_erf_grad(float x) {
    float cdf, pdf;
    gelu_cdf_pdf(x, cdf, pdf);
    return cdf + x * pdf;
}

// XCD-aware bijective remap of a 1-D block id: blocks that share an XCD (id % 8) get a contiguous chunk of tiles.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7, k = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
}
