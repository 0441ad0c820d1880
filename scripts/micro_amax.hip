// Micro-benchmark: what does "fold one absmax per wave into ONE word" cost on gfx950, and which look-before-you-add load is
// cheapest?  Each wave streams one 3 KB row (LayerNorm-like) and then folds its maximum.
//   hipcc -O3 --offload-arch=gfx950 scripts/micro_amax.hip -o gpurun_out/micro_amax && gpurun_out/micro_amax
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void fold_kernel(const float* __restrict__ x, float* __restrict__ y, float* amax, int rows) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    float mx = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        f32x4 v = *reinterpret_cast<const f32x4*>(x + (size_t)row * 768 + 4 * (lane + 64 * i));
        v = v * 1.5f;
        *reinterpret_cast<f32x4*>(y + (size_t)row * 768 + 4 * (lane + 64 * i)) = v;
        mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    }
    if (MODE == 0) return;
    for (int o = 32; o; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if (lane != 0) return;
    int* dst = reinterpret_cast<int*>(amax);
    if (MODE == 1) { if (mx > __builtin_nontemporal_load(amax)) atomicMax(dst, __float_as_int(mx)); }
    if (MODE == 2) { if (mx > __hip_atomic_load(amax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(dst, __float_as_int(mx)); }
    if (MODE == 3) { if (mx > *reinterpret_cast<volatile float*>(amax)) atomicMax(dst, __float_as_int(mx)); }
    if (MODE == 4) atomicMax(dst, __float_as_int(mx));
    if (MODE == 5) { if (mx > *amax) atomicMax(dst, __float_as_int(mx)); }
    if (MODE == 6) __hip_atomic_fetch_max(dst, __float_as_int(mx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // no return value used
}

// sub-slots: the workgroup folds into amax[(blockIdx.x % S) * stride] -- are same-LINE / same-CHANNEL atomics serialised too?
template <bool LOOK, bool BLOCK>
__global__ __launch_bounds__(256) void spread_kernel(const float* __restrict__ x, float* __restrict__ y, float* amax, int rows, int S,
                                                     int stride) {
    __shared__ float red[4];
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    float mx = 0.f;
    if (row < rows) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            f32x4 v = *reinterpret_cast<const f32x4*>(x + (size_t)row * 768 + 4 * (lane + 64 * i));
            v = v * 1.5f;
            *reinterpret_cast<f32x4*>(y + (size_t)row * 768 + 4 * (lane + 64 * i)) = v;
            mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
        }
    }
    for (int o = 32; o; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float* dst = amax + (size_t)(blockIdx.x % S) * stride;
    if (BLOCK) {
        if (lane == 0) red[threadIdx.x >> 6] = mx;
        __syncthreads();
        if (threadIdx.x != 0) return;
        mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    } else if (lane != 0) {
        return;
    }
    if (!LOOK || mx > __hip_atomic_load(dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(reinterpret_cast<int*>(dst), __float_as_int(mx));
}

template <bool LOOK, bool BLOCK>
float run_spread(const float* x, float* y, float* amax, int rows, int S, int stride, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < reps + 3; ++i) {
        if (i == 3) hipEventRecord(a, 0);
        hipMemsetAsync(amax, 0, (size_t)S * stride * 4, 0);
        hipLaunchKernelGGL((spread_kernel<LOOK, BLOCK>), dim3((rows + 3) / 4), dim3(256), 0, 0, x, y, amax, rows, S, stride);
    }
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / reps;
}

template <int MODE>
float run(const float* x, float* y, float* amax, int rows, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) { hipMemsetAsync(amax, 0, 4, 0); hipLaunchKernelGGL(fold_kernel<MODE>, dim3((rows + 3) / 4), dim3(256), 0, 0, x, y, amax, rows); }
    hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) { hipMemsetAsync(amax, 0, 4, 0); hipLaunchKernelGGL(fold_kernel<MODE>, dim3((rows + 3) / 4), dim3(256), 0, 0, x, y, amax, rows); }
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / reps;
}

int main() {
    const char* names[] = {"no fold", "nontemporal look", "agent-scope atomic load look", "volatile look", "atomicMax always",
                           "plain look", "fetch_max (no return) always"};
    for (int rows : {6400, 18432, 131072}) {
        float *x, *y, *amax;
        hipMalloc(&x, (size_t)rows * 768 * 4); hipMalloc(&y, (size_t)rows * 768 * 4); hipMalloc(&amax, 256);
        std::vector<float> h((size_t)rows * 768);
        unsigned s = 12345u;
        for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (float)(s >> 8) / 16777216.f - 0.5f; }
        hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        const int reps = rows > 200000 ? 10 : 50;
        float t[7];
        t[0] = run<0>(x, y, amax, rows, reps); t[1] = run<1>(x, y, amax, rows, reps); t[2] = run<2>(x, y, amax, rows, reps);
        t[3] = run<3>(x, y, amax, rows, reps); t[4] = run<4>(x, y, amax, rows, reps); t[5] = run<5>(x, y, amax, rows, reps);
        t[6] = run<6>(x, y, amax, rows, reps);
        float got; hipMemcpy(&got, amax, 4, hipMemcpyDeviceToHost);
        printf("rows %8d (amax %.4f)\n", rows, got);
        for (int m = 0; m < 7; ++m) printf("   %-32s %9.1f us  (+%.1f)\n", names[m], t[m], t[m] - t[0]);
        float* wide;
        hipMalloc(&wide, 64 * 4096 * 4);
        for (int S : {1, 8, 32, 64})
            for (int stride : {1, 16, 64, 1024}) {
                if (S == 1 && stride > 1) continue;
                printf("   S %2d stride %5d B: always/wave %8.1f  look/wave %8.1f  always/block %8.1f  look/block %8.1f us\n", S, stride * 4,
                       run_spread<false, false>(x, y, wide, rows, S, stride, reps), run_spread<true, false>(x, y, wide, rows, S, stride, reps),
                       run_spread<false, true>(x, y, wide, rows, S, stride, reps), run_spread<true, true>(x, y, wide, rows, S, stride, reps));
            }
        hipFree(wide);
        hipFree(x); hipFree(y); hipFree(amax);
    }
    return 0;
}
