// A plain C++ caller of the C ABI in include/maestro_hip.h -- no PyTorch, no Python: device memory from hipMalloc, plain
// pointers and sizes across the boundary.  What a non-Python host (or the reference behind its own FFI) would link against.
//   build: hipcc -O2 -std=c++17 --offload-arch=gfx950 examples/abi_smoke.cpp -Iinclude -Lmaestro_amd/lib -lmaestro_hip \
//                -Wl,-rpath,$PWD/maestro_amd/lib -o examples/abi_smoke      (python -m maestro_amd.csrc.build does it)
//   run:   examples/abi_smoke   -> exit code 0 and "abi_smoke ok" when every check is exact
// Checks (integer-valued data, so fp32 accumulation is exact and the comparison is bit-exact):
//   mh_gemm_bf16 NT with fp32 output + bias (the Linear of maestro/ssl/mae.py:145-154),
//   mh_mask_select against the stable-rank definition of maestro/ssl/mae.py:236-259.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <vector>

#include "maestro_hip.h"

#define HIP_OK(x)                                                                              \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } \
    } while (0)
#define MH_OK(x)                                                                               \
    do {                                                                                       \
        int rc_ = (x);                                                                         \
        if (rc_ != 0) { std::printf("ABI call failed (%d): %s\n", rc_, mh_last_error()); return 3; } \
    } while (0)

static uint16_t to_bf16(float f) {   // exact for the small integers used here
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return (uint16_t)(u >> 16);
}

int main() {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { std::printf("no GPU\n"); return 77; }
    std::printf("libmaestro_hip ABI version %d\n", mh_version());
    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));

    // ---- GEMM: C[M, N] = A[M, K] * B[N, K]^T + bias, M not a tile multiple
    const int M = 300, N = 192, K = 96;
    std::vector<float> a(M * K), b(N * K), bias(N), want((size_t)M * N);
    for (int i = 0; i < M * K; ++i) a[i] = (float)((i * 7 + 3) % 9 - 4);
    for (int i = 0; i < N * K; ++i) b[i] = (float)((i * 5 + 1) % 7 - 3);
    for (int n = 0; n < N; ++n) bias[n] = (float)(n % 11 - 5);
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
            float s = bias[n];
            for (int k = 0; k < K; ++k) s += a[m * K + k] * b[n * K + k];
            want[(size_t)m * N + n] = s;
        }
    std::vector<uint16_t> a16(a.size()), b16(b.size());
    std::transform(a.begin(), a.end(), a16.begin(), to_bf16);
    std::transform(b.begin(), b.end(), b16.begin(), to_bf16);
    void *dA, *dB;
    float *dC, *dBias;
    HIP_OK(hipMalloc(&dA, a16.size() * 2));
    HIP_OK(hipMalloc(&dB, b16.size() * 2));
    HIP_OK(hipMalloc(&dC, want.size() * 4));
    HIP_OK(hipMalloc(&dBias, bias.size() * 4));
    HIP_OK(hipMemcpy(dA, a16.data(), a16.size() * 2, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dB, b16.data(), b16.size() * 2, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dBias, bias.data(), bias.size() * 4, hipMemcpyHostToDevice));
    MH_OK(mh_gemm_bf16(/*layout NT*/ 0, M, N, K, dA, K, dB, K, dC, N, MH_GEMM_OUT_F32 | MH_GEMM_BIAS, dBias, nullptr, 0, nullptr,
                       nullptr, 0, nullptr, stream));
    std::vector<float> got(want.size());
    HIP_OK(hipStreamSynchronize(stream));
    HIP_OK(hipMemcpy(got.data(), dC, got.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < got.size(); ++i) bad += got[i] != want[i];
    std::printf("mh_gemm_bf16 %dx%dx%d: %zu mismatches\n", M, N, K, bad);
    if (bad) return 1;
    // bad arguments come back as a negative code with a message, nothing is launched
    if (mh_gemm_bf16(0, M, N, K, dA, K + 1, dB, K, dC, N, MH_GEMM_OUT_F32, nullptr, nullptr, 0, nullptr, nullptr, 0, nullptr, stream) >= 0) {
        std::printf("an lda that is not a multiple of 8 was accepted\n");
        return 1;
    }

    // ---- mask_select: k smallest noise values per row are masked, ties -> lower index first; index lists ascending
    const int B = 3, L = 37, k = 28;
    std::vector<float> noise(B * L);
    for (int i = 0; i < B * L; ++i) noise[i] = (float)((i * 2654435761u) % 23) / 23.f;   // plenty of ties
    float* dNoise;
    int *dVis, *dMsk, *dInv;
    uint8_t* dMask;
    HIP_OK(hipMalloc(&dNoise, noise.size() * 4));
    HIP_OK(hipMalloc(&dVis, B * (L - k) * 4));
    HIP_OK(hipMalloc(&dMsk, B * k * 4));
    HIP_OK(hipMalloc(&dInv, B * L * 4));
    HIP_OK(hipMalloc(&dMask, B * L));
    HIP_OK(hipMemcpy(dNoise, noise.data(), noise.size() * 4, hipMemcpyHostToDevice));
    MH_OK(mh_mask_select(dNoise, nullptr, dVis, dMsk, dInv, dMask, B, L, k, stream));
    HIP_OK(hipStreamSynchronize(stream));
    std::vector<int> vis(B * (L - k)), msk(B * k);
    std::vector<uint8_t> mask(B * L);
    HIP_OK(hipMemcpy(vis.data(), dVis, vis.size() * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(msk.data(), dMsk, msk.size() * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(mask.data(), dMask, mask.size(), hipMemcpyDeviceToHost));
    for (int r = 0; r < B; ++r) {
        std::vector<int> order(L);
        std::iota(order.begin(), order.end(), 0);
        std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return noise[r * L + x] < noise[r * L + y]; });
        std::vector<int> wm(order.begin(), order.begin() + k), wv(order.begin() + k, order.end());
        std::sort(wm.begin(), wm.end());
        std::sort(wv.begin(), wv.end());
        bool ok = std::equal(wm.begin(), wm.end(), msk.begin() + r * k) && std::equal(wv.begin(), wv.end(), vis.begin() + r * (L - k));
        for (int t = 0; t < L; ++t) ok = ok && mask[r * L + t] == (uint8_t)std::binary_search(wm.begin(), wm.end(), t);
        if (!ok) { std::printf("mh_mask_select: row %d differs from the stable-rank definition\n", r); return 1; }
    }
    std::printf("mh_mask_select %dx%d k=%d: exact\n", B, L, k);
    std::printf("abi_smoke ok\n");
    return 0;
}
