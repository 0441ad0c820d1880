// Round 5 micro-benchmark (not product code): the streaming pattern of the AdamW kernel -- 4 fp32 read streams, 3 fp32 + 1 bf16 write streams,
// 176 M elements (C3) -- under four forms: plain / nontemporal accesses, one or two 16-byte quads per lane.  hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2(float a, float b) {
    uint32_t x = __float_as_uint(a), y = __float_as_uint(b);
    x += 0x7fffu + ((x >> 16) & 1u); y += 0x7fffu + ((y >> 16) & 1u);
    return (x >> 16) | (y & 0xffff0000u);
}
template <bool NT> __device__ __forceinline__ f32x4 ld(const float* p) {
    if constexpr (NT) return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
    else return *reinterpret_cast<const f32x4*>(p);
}
template <bool NT> __device__ __forceinline__ void st(float* p, f32x4 v) {
    if constexpr (NT) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
    else *reinterpret_cast<f32x4*>(p) = v;
}
template <bool NT> __device__ __forceinline__ void st2(uint16_t* p, u32x2 v) {
    if constexpr (NT) __builtin_nontemporal_store(v, reinterpret_cast<u32x2*>(p));
    else *reinterpret_cast<u32x2*>(p) = v;
}
__device__ __forceinline__ void upd(f32x4& pv, f32x4& mv, f32x4& vv, f32x4 gv, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2s) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        pv[e] *= 1.f - lr * wd;
        mv[e] = b1 * mv[e] + (1.f - b1) * gv[e];
        vv[e] = b2 * vv[e] + (1.f - b2) * gv[e] * gv[e];
        const float denom = sqrtf(vv[e]) / bc2s + eps;
        pv[e] -= (lr / bc1) * (mv[e] / denom);
    }
}
template <bool NTL, bool NTS, int Q>
__global__ __launch_bounds__(256) void k(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                         uint16_t* __restrict__ pb, long n) {
    const long base = ((long)blockIdx.x * 256 * Q + threadIdx.x) * 4;
    f32x4 gv[Q], pv[Q], mv[Q], vv[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const long i = base + q * 1024;
        if (i < n) { gv[q] = ld<NTL>(g + i); pv[q] = ld<NTL>(p + i); mv[q] = ld<NTL>(m + i); vv[q] = ld<NTL>(v + i); }
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const long i = base + q * 1024;
        if (i < n) {
            upd(pv[q], mv[q], vv[q], gv[q], 1e-3f, 0.9f, 0.95f, 1e-8f, 0.05f, 0.1f, 0.2f);
            st<NTS>(p + i, pv[q]); st<NTS>(m + i, mv[q]); st<NTS>(v + i, vv[q]);
            st2<NTS>(pb + i, (u32x2){pack2(pv[q][0], pv[q][1]), pack2(pv[q][2], pv[q][3])});
        }
    }
}
template <bool NTL, bool NTS, int Q> float run(float* p, float* g, float* m, float* v, uint16_t* pb, long n) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = (int)((n + 1024L * Q - 1) / (1024L * Q));
    float best = 1e9;
    for (int r = 0; r < 6; ++r) {
        hipEventRecord(e0);
        for (int it = 0; it < 3; ++it) hipLaunchKernelGGL((k<NTL, NTS, Q>), dim3(grid), dim3(256), 0, 0, p, g, m, v, pb, n);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms / 3 < best) best = ms / 3;
    }
    return best;
}
int main() {
    const long n = 176200000L / 1024 * 1024;
    float *p, *g, *m, *v; uint16_t* pb;
    hipMalloc(&p, n * 4); hipMalloc(&g, n * 4); hipMalloc(&m, n * 4); hipMalloc(&v, n * 4); hipMalloc(&pb, n * 2);
    hipMemset(p, 0, n * 4); hipMemset(g, 0, n * 4); hipMemset(m, 0, n * 4); hipMemset(v, 0, n * 4);
    const double bytes = (double)n * 30;
#define R(NTL, NTS, Q) { float ms = run<NTL, NTS, Q>(p, g, m, v, pb, n); printf("nt loads %d  nt stores %d  quads/lane %d : %.1f us  %.0f GB/s\n", NTL, NTS, Q, ms * 1e3, bytes / ms / 1e6); }
    R(false, false, 1) R(true, false, 1) R(false, true, 1) R(true, true, 1) R(false, false, 2) R(true, true, 2) R(false, false, 4) R(true, true, 4)
    R(false, false, 1) R(true, true, 1)
    return 0;
}
