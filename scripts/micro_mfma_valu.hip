// Micro-benchmark for round 4: how much VALU work fits under the MFMAs, by MFMA SHAPE?
//   hipcc -O3 -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form --offload-arch=gfx950 scripts/micro_mfma_valu.hip \
//         -o gpurun_out/micro_mfma_valu && gpurun_out/micro_mfma_valu
// (the two code-generation flags keep the loops clean: scalar v_fma_f32, accumulators in VGPRs -- checked in the ISA: the loop
//  bodies are `M M f*V` per unit for 16x16x32 and `M f*V` for 32x32x16)
//
// Why: profiles/r03_isa_budget.txt (static count from the ISA, guide constants).  v_mfma_f32_16x16x32_bf16 takes 16 cycles and
// holds the SIMD's vector issue port for 8 of them; v_mfma_f32_32x32x16_bf16 takes 32 and also holds 8, at the same flops per
// cycle.  Every MFMA kernel of the library is built on 16x16x32; the fc1 GEMM's GELU epilogue (5.3 VALU + 0.7 transcendental
// instructions per 32 MFMA cycles of its K = 768 tile = 27 cycles of vector issue) does not fit under them (+33 % predicted, +30 % measured).  On 32x32x16 the same VALU work
// should nearly fit (+8 % predicted).  This program measures exactly that before any kernel is rewritten:
//
//   one "unit" = 32 cycles of MFMA pipe = 2 x 16x16x32 or 1 x 32x32x16 (32768 flop per wave either way), followed in program
//   order by V independent v_fma_f32 (and T x (v_exp_f32 + v_mul_f32)); ITER units per wave, 4 independent accumulator sets, no memory traffic.
//   Printed: ns per unit and the slowdown against V = 0 for both shapes, at 1 and 2 waves per SIMD (the second wave of a SIMD
//   can issue VALU while the first one's MFMA runs -- the library's kernels live at 2 waves per SIMD).
// Reading: the V at which a shape's time starts to grow is its free VALU room per 32 MFMA cycles (prediction: 4 for 16x16x32
// -- (32 - 2 x 8) / 4 -- and 6 for 32x32x16 -- (32 - 8) / 4); the slope beyond it should be 4 cycles per instruction for both.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

// SHAPE 16: 2 x 16x16x32 per unit; SHAPE 32: 1 x 32x32x16 per unit.  V fma + T exp per unit, on registers of their own.
template <int SHAPE, int V, int T>
__global__ __launch_bounds__(256) void unit_kernel(float* __restrict__ out, int iters, float seed) {
    const int lane = threadIdx.x;
    bf16x8 a, b;
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(seed + 0.001f * (lane + j)); b[j] = (__bf16)(seed - 0.002f * (lane - j)); }
    constexpr int NVR = V > 0 ? V : 1, NTR = T > 0 ? T : 1;
    float x[NVR], t[NTR];
#pragma unroll
    for (int i = 0; i < NVR; ++i) x[i] = seed + i;
#pragma unroll
    for (int i = 0; i < NTR; ++i) t[i] = seed * 0.01f + i;
    const float k1 = 0.999f + seed * 1e-6f, k2 = 1e-3f;
    if constexpr (SHAPE == 16) {
        f32x4 acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {       // four units per trip, eight independent accumulators
                acc[2 * u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[2 * u], 0, 0, 0);
                acc[2 * u + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[2 * u + 1], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < V; ++i) x[i] = __builtin_fmaf(x[i], k1, k2);
#pragma unroll
                for (int i = 0; i < T; ++i) t[i] = __builtin_amdgcn_exp2f(t[i]) * 0.5f;
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);              // the unit's MFMAs ...
                __builtin_amdgcn_sched_group_barrier(0x002, V + 2 * T, 0);      // ... then its VALU / transcendental work
            }
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
#pragma unroll
        for (int i = 0; i < NVR; ++i) s += x[i];
#pragma unroll
        for (int i = 0; i < NTR; ++i) s += t[i];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    } else {
        f32x16 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < V; ++i) x[i] = __builtin_fmaf(x[i], k1, k2);
#pragma unroll
                for (int i = 0; i < T; ++i) t[i] = __builtin_amdgcn_exp2f(t[i]) * 0.5f;
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, V + 2 * T, 0);
            }
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
#pragma unroll
        for (int i = 0; i < NVR; ++i) s += x[i];
#pragma unroll
        for (int i = 0; i < NTR; ++i) s += t[i];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    }
}

template <int SHAPE, int V, int T>
static float run(float* out, int blocks, int iters) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((unit_kernel<SHAPE, V, T>), dim3(blocks), dim3(256), 0, 0, out, iters / 8, 1.0f);      // warm-up
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((unit_kernel<SHAPE, V, T>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    return best * 1e6f / (4.f * iters);      // ns per unit (each wave runs 4 units per trip)
}

template <int V, int T>
static void row(float* out, int blocks, int iters, float base16, float base32) {
    const float t16 = run<16, V, T>(out, blocks, iters), t32 = run<32, V, T>(out, blocks, iters);
    printf("  V = %2d fma + %d exp per 32 MFMA cycles:  16x16x32 %7.2f ns/unit (x%.2f)   32x32x16 %7.2f ns/unit (x%.2f)\n", V, T, t16,
           t16 / base16, t32, t32 / base32);
}

int main() {
    int dev = 0, cus = 0;
    CHECK(hipGetDevice(&dev));
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    float* out;
    CHECK(hipMalloc(&out, (size_t)cus * 8 * 256 * sizeof(float)));
    const int iters = 20000;
    for (int waves = 1; waves <= 2; ++waves) {      // 256-thread blocks = one wave per SIMD; `waves` blocks per CU
        const int blocks = cus * waves;
        printf("%d wave(s) per SIMD (%d blocks of 256 threads on %d CUs), %d units per wave\n", waves, blocks, cus, 4 * iters);
        const float b16 = run<16, 0, 0>(out, blocks, iters), b32 = run<32, 0, 0>(out, blocks, iters);
        printf("  MFMA only:                               16x16x32 %7.2f ns/unit (%.0f TFLOP/s)   32x32x16 %7.2f ns/unit (%.0f TFLOP/s)\n", b16,
               32768.0 * 4 * blocks / b16 * 1e-3, b32, 32768.0 * 4 * blocks / b32 * 1e-3);
        row<2, 0>(out, blocks, iters, b16, b32);
        row<4, 0>(out, blocks, iters, b16, b32);
        row<6, 0>(out, blocks, iters, b16, b32);
        row<8, 0>(out, blocks, iters, b16, b32);
        row<12, 0>(out, blocks, iters, b16, b32);
        row<16, 0>(out, blocks, iters, b16, b32);
        row<4, 1>(out, blocks, iters, b16, b32);       // ~ the fc1 epilogue's mix per 32 MFMA cycles (5 VALU + 1 exp / rcp)
        row<16, 4>(out, blocks, iters, b16, b32);      // ~ the D = 32 attention forward (20 VALU + 4 exp per 32 MFMA cycles)
    }
    CHECK(hipFree(out));
    return 0;
}
