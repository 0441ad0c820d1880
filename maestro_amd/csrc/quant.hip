// fp8 quantisers of the C5 path (operands of mh_gemm_fp8): per-TENSOR power-of-two scales kept in device memory, so that
// the whole scaling loop -- absmax, scale update, cast -- is capturable in the step's hipGraphs and never read by the host.
//   scale table: one float slot per tensor: scale (multiplier before the cast), descale = 1 / scale (GEMM epilogue); amax: one
//                ROW of MH_FP8_AMAX_PITCH floats per slot (sub-slots against same-line atomics, see atomic_max_pos).
//   weights:     every optimizer step  absmax (batched) -> update -> cast (batched; optionally also the TRANSPOSED copy the
//                dgrad reads, so that both GEMM directions stay K-minor x K-minor)
//   activations: delayed scaling -- the cast of step t uses the scale derived from step t-1's amax and folds |x| of step t
//                into the amax slot; the first step runs with scale 1 (LayerNorm / GELU outputs are O(1): inside e4m3's
//                normal range).
#include "gemm_common.hpp"

namespace {

constexpr int QCHUNK = 4096;   // elements per workgroup

__device__ __forceinline__ uint32_t pack_fp8x4(f32x4 v, float s, bool e5m2) {
    const float lim = e5m2 ? 57344.f : 448.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fminf(fmaxf(v[e] * s, -lim), lim);
    int a;
    if (e5m2) {
        a = __builtin_amdgcn_cvt_pk_bf8_f32(v[0], v[1], 0, false);
        a = __builtin_amdgcn_cvt_pk_bf8_f32(v[2], v[3], a, true);
    } else {
        a = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0, false);
        a = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], a, true);
    }
    return (uint32_t)a;
}

__device__ __forceinline__ f32x4 load4(const void* src, int is_f32, long i) {
    if (is_f32) return *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(src) + i);
    const u32x2 pk = *reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16_t*>(src) + i);
    return (f32x4){__uint_as_float(pk[0] << 16), __uint_as_float(pk[0] & 0xffff0000u), __uint_as_float(pk[1] << 16),
                   __uint_as_float(pk[1] & 0xffff0000u)};
}

// work item = job << 32 | chunk; job: (src, dst, dst_t, n, rows, cols, slot, is_f32, format)
__global__ __launch_bounds__(256) void quant_batched_kernel(const MhQuantJob* __restrict__ jobs, const uint64_t* __restrict__ items,
                                                            const float* __restrict__ scale, float* __restrict__ amax, int mode) {
    __shared__ float red[4];
    const uint64_t it = items[blockIdx.x];
    const MhQuantJob jb = jobs[it >> 32];
    const long base = (long)(uint32_t)it * QCHUNK;
    const float s = mode == 0 ? 0.f : scale[jb.slot];
    float mx = 0.f;
#pragma unroll
    for (int r = 0; r < QCHUNK / 1024; ++r) {
        const long i = base + r * 1024 + threadIdx.x * 4;
        if (i < jb.n) {   // n % 4 == 0
            const f32x4 v = load4(jb.src, jb.is_f32, i);
            mx = amax_fold(amax_fold(amax_fold(amax_fold(mx, v[0]), v[1]), v[2]), v[3]);
            if (mode != 0) {
                const uint32_t pk = pack_fp8x4(v, s, jb.format == MH_FP8_E5M2);
                if (jb.dst) *reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(jb.dst) + i) = pk;
                if (jb.dst_t) {   // transposed copy [cols, rows] of a [rows, cols] tensor (cols % 4 == 0: the 4 values share a row)
                    const int row = (int)(i / jb.cols), col = (int)(i - (long)row * jb.cols);
                    uint8_t* t = reinterpret_cast<uint8_t*>(jb.dst_t);
#pragma unroll
                    for (int e = 0; e < 4; ++e) t[(size_t)(col + e) * jb.rows + row] = (uint8_t)(pk >> (8 * e));
                }
            }
        }
    }
    if (mode != 1) {   // 0: absmax only; 2: cast + absmax (delayed scaling); 1: cast only
        mx = wave_amax(mx);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
        __syncthreads();
        if (threadIdx.x == 0) {
            mx = amax_fold(amax_fold(amax_fold(red[0], red[1]), red[2]), red[3]);
            if (amax_nonzero(mx)) atomic_max_pos(amax + (size_t)jb.slot * MH_FP8_AMAX_PITCH, mx);
        }
    }
}

// scale = 2^(floor(log2(fmax / amax)) - margin) (1 while amax is 0; NaN when amax is inf / NaN: the divergence must show), descale = 1 / scale, amax reset to 0
__global__ __launch_bounds__(256) void update_scales_kernel(float* __restrict__ amax, float* __restrict__ scale,
                                                            float* __restrict__ descale, int n, float fmax8, int margin) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), l = threadIdx.x & 63;   // one wave per slot: maximum over its amax row
    if (i >= n) return;
    float* row = amax + (size_t)i * MH_FP8_AMAX_PITCH + (l & (MH_FP8_AMAX_SUBSLOTS - 1)) * MH_FP8_AMAX_STRIDE;
    const float a = wave_amax(*row);
    if (l < MH_FP8_AMAX_SUBSLOTS) *row = 0.f;
    if (l != 0) return;
    float s = 1.f;
    if (a > 0.f && a < 3.0e38f) {
        int e = (int)floorf(log2f(fmax8 / a)) - margin;
        e = max(-100, min(100, e));
        s = exp2f((float)e);
    } else if (amax_nonzero(a)) {
        s = __uint_as_float(0x7fc00000u);     // the tensor held an inf / NaN: poison its scale and descale -> NaN GEMM outputs, NaN loss
    }
    scale[i] = s;
    descale[i] = 1.f / s;
}

// dst [cols, rows] = src [rows, cols]^T, bytes; one 64 x 64 tile per workgroup through LDS (16-byte global accesses both ways)
__global__ __launch_bounds__(256) void transpose_u8_kernel(const MhTransposeJob* __restrict__ jobs, const uint64_t* __restrict__ items) {
    __shared__ __attribute__((aligned(16))) uint8_t t[64][80];
    const uint64_t it = items[blockIdx.x];
    const MhTransposeJob jb = jobs[it >> 32];
    const int tiles_c = jb.cols >> 6, tile = (int)(uint32_t)it;
    const int tr = tile / tiles_c, tc = tile - tr * tiles_c;
    const int r = threadIdx.x >> 2, q = threadIdx.x & 3;
    const uint8_t* src = reinterpret_cast<const uint8_t*>(jb.src);
    uint8_t* dst = reinterpret_cast<uint8_t*>(jb.dst);
    *reinterpret_cast<u32x4*>(&t[r][16 * q]) =
        *reinterpret_cast<const u32x4*>(src + (size_t)(tr * 64 + r) * jb.cols + tc * 64 + 16 * q);
    __syncthreads();
    u32x4 w;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        w[k] = (uint32_t)t[16 * q + 4 * k][r] | ((uint32_t)t[16 * q + 4 * k + 1][r] << 8) | ((uint32_t)t[16 * q + 4 * k + 2][r] << 16) |
               ((uint32_t)t[16 * q + 4 * k + 3][r] << 24);
    *reinterpret_cast<u32x4*>(dst + (size_t)(tc * 64 + r) * jb.rows + tr * 64 + 16 * q) = w;
}

}  // namespace

extern "C" int mh_transpose_u8_batched(const MhTransposeJob* jobs_device, const unsigned long* items_device, int n_items, void* stream) {
    MH_CHECK_ARG(jobs_device && items_device && n_items > 0, "mh_transpose_u8_batched: bad arguments");
    hipLaunchKernelGGL(transpose_u8_kernel, dim3(n_items), dim3(256), 0, (hipStream_t)stream, jobs_device,
                       reinterpret_cast<const uint64_t*>(items_device));
    MH_LAUNCH_CHECK();
    return 0;
}

extern "C" int mh_quant_batched(const MhQuantJob* jobs_device, const unsigned long* items_device_, int n_items, const float* scale,
                                float* amax, int mode, void* stream) {
    const uint64_t* items_device = reinterpret_cast<const uint64_t*>(items_device_);
    MH_CHECK_ARG(jobs_device && items_device && n_items > 0 && amax && (mode == 0 || scale) && mode >= 0 && mode <= 2,
                 "mh_quant_batched: bad arguments");
    hipLaunchKernelGGL(quant_batched_kernel, dim3(n_items), dim3(256), 0, (hipStream_t)stream, jobs_device, items_device, scale, amax,
                       mode);
    MH_LAUNCH_CHECK();
    return 0;
}

extern "C" int mh_fp8_update_scales(float* amax, float* scale, float* descale, int n, float format_max, int margin_log2,
                                    void* stream) {
    MH_CHECK_ARG(amax && scale && descale && n > 0 && format_max > 0.f, "mh_fp8_update_scales: bad arguments");
    hipLaunchKernelGGL(update_scales_kernel, dim3(ceil_div(n, 4)), dim3(256), 0, (hipStream_t)stream, amax, scale, descale, n,
                       format_max, margin_log2);
    MH_LAUNCH_CHECK();
    return 0;
}
