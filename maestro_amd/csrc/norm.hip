// LayerNorm forward/backward for the fp32 residual stream (see include/maestro_hip.h).
// One wave (64 lanes) per row, row held in registers as float4 (dim <= 2048), two-pass statistics, wavefront
// shuffles for the reductions.  Rows are addressed through (sample, j) maps so that "split / concat of group
// sequences" (reference mim.py:408-423) is done by addressing instead of copies:
//     row(b, j) = b * L + off + j,  j < n.
#include "common.hpp"
#include "../../include/maestro_hip.h"

namespace {

constexpr int MAXV = 8;  // float4 per lane -> dim <= 2048

struct RowMap { int L, off; };
__device__ __forceinline__ size_t map_row(RowMap m, int b, int j) { return (size_t)b * m.L + m.off + j; }

__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, RowMap xm, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, void* __restrict__ y, RowMap ym,
                                                     int y_is_f32, float* __restrict__ mean, float* __restrict__ rstd,
                                                     int B, int n, int dim, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= B * n) return;
    const int b = row / n, j = row - b * n;
    const float* xr = x + map_row(xm, b, j) * dim;
    const int nv = dim >> 2;
    f32x4 v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            v[i] = *reinterpret_cast<const f32x4*>(xr + 4 * c);
            s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        }
    }
    const float mu = wave_sum(s) / dim;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mu; q += d * d; }
        }
    }
    const float rs = rsqrtf(wave_sum(q) / dim + eps);
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
    const size_t yrow = map_row(ym, b, j) * dim;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
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
        }
    }
}

// Backward. Each wave walks ROWS_PER_WAVE rows, keeps dgamma/dbeta partials in registers, the block reduces them
// through LDS and issues one atomic per column.
constexpr int ROWS_PER_WAVE = 8;

__global__ __launch_bounds__(256) void ln_bwd_kernel(const void* __restrict__ dy, RowMap dym, int dy_is_f32,
                                                     const float* __restrict__ x, RowMap xm,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const float* __restrict__ dres,
                                                     float* __restrict__ dx, bf16_t* __restrict__ dx_bf16,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta, int B, int n,
                                                     int dim) {
    extern __shared__ __attribute__((aligned(16))) float red[];  // [4 waves][2][dim]
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nv = dim >> 2;
    f32x4 gsum[MAXV], bsum[MAXV], gm[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        gsum[i] = (f32x4){0, 0, 0, 0}; bsum[i] = (f32x4){0, 0, 0, 0};
        const int c = lane + 64 * i;
        gm[i] = c < nv ? *reinterpret_cast<const f32x4*>(gamma + 4 * c) : (f32x4){0, 0, 0, 0};
    }
    const int row0 = (blockIdx.x * 4 + w) * ROWS_PER_WAVE;
    for (int rr = 0; rr < ROWS_PER_WAVE; ++rr) {
        const int row = row0 + rr;
        if (row >= B * n) break;
        const int b = row / n, j = row - b * n;
        const size_t xrow = map_row(xm, b, j) * dim, dyrow = map_row(dym, b, j) * dim;
        const float mu = mean[row], rs = rstd[row];
        f32x4 xh[MAXV], dz[MAXV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                const f32x4 xv = *reinterpret_cast<const f32x4*>(x + xrow + 4 * c);
                f32x4 d;
                if (dy_is_f32) {
                    d = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(dy) + dyrow + 4 * c);
                } else {
                    const u32x2 pk = *reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16_t*>(dy) + dyrow + 4 * c);
                    d = (f32x4){__uint_as_float(pk[0] << 16), __uint_as_float(pk[0] & 0xffff0000u),
                                __uint_as_float(pk[1] << 16), __uint_as_float(pk[1] & 0xffff0000u)};
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
        for (int i = 0; i < MAXV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = rs * (dz[i][e] - c1 - xh[i][e] * c2);
                if (dres) o += *reinterpret_cast<const f32x4*>(dres + xrow + 4 * c);
                *reinterpret_cast<f32x4*>(dx + xrow + 4 * c) = o;
                if (dx_bf16) {
                    u32x2 pk = {pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3])};
                    *reinterpret_cast<u32x2*>(dx_bf16 + xrow + 4 * c) = pk;
                }
            }
        }
    }
    if (!dgamma) return;
    float* rg = red + (size_t)w * 2 * dim;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            *reinterpret_cast<f32x4*>(rg + 4 * c) = gsum[i];
            *reinterpret_cast<f32x4*>(rg + dim + 4 * c) = bsum[i];
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * dim; c += 256) {
        const float t = red[c] + red[2 * dim + c] + red[4 * dim + c] + red[6 * dim + c];
        if (c < dim) atomicAdd(dgamma + c, t); else atomicAdd(dbeta + (c - dim), t);
    }
}

}  // namespace

extern "C" int mh_layernorm_fwd(const float* x, int x_L, int x_off, const float* gamma, const float* beta, void* y,
                                int y_L, int y_off, int y_is_f32, float* mean, float* rstd, int B, int n, int dim,
                                float eps, void* stream) {
    MH_CHECK_ARG(x && gamma && beta && y && mean && rstd, "mh_layernorm_fwd: null pointer");
    MH_CHECK_ARG(dim % 4 == 0 && dim >= 4 && dim <= 4 * 64 * MAXV, "mh_layernorm_fwd: dim %d unsupported", dim);
    MH_CHECK_ARG(B > 0 && n > 0 && x_off + n <= x_L && y_off + n <= y_L, "mh_layernorm_fwd: bad row map");
    const int rows = B * n;
    hipLaunchKernelGGL(ln_fwd_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, RowMap{x_L, x_off},
                       gamma, beta, y, RowMap{y_L, y_off}, y_is_f32, mean, rstd, B, n, dim, eps);
    MH_LAUNCH_CHECK();
    return 0;
}

extern "C" int mh_layernorm_bwd(const void* dy, int dy_L, int dy_off, int dy_is_f32, const float* x, int x_L, int x_off,
                                const float* gamma, const float* mean, const float* rstd, const float* dres, float* dx,
                                void* dx_bf16, float* dgamma, float* dbeta, int B, int n, int dim, void* stream) {
    MH_CHECK_ARG(dy && x && gamma && mean && rstd && dx, "mh_layernorm_bwd: null pointer");
    MH_CHECK_ARG((dgamma == nullptr) == (dbeta == nullptr), "mh_layernorm_bwd: dgamma/dbeta must come together");
    MH_CHECK_ARG(dim % 4 == 0 && dim >= 4 && dim <= 4 * 64 * MAXV, "mh_layernorm_bwd: dim %d unsupported", dim);
    MH_CHECK_ARG(B > 0 && n > 0 && x_off + n <= x_L && dy_off + n <= dy_L, "mh_layernorm_bwd: bad row map");
    const int rows = B * n;
    const size_t lds = (size_t)4 * 2 * dim * sizeof(float);
    hipLaunchKernelGGL(ln_bwd_kernel, dim3(ceil_div(rows, 4 * ROWS_PER_WAVE)), dim3(256), lds, (hipStream_t)stream, dy,
                       RowMap{dy_L, dy_off}, dy_is_f32, x, RowMap{x_L, x_off}, gamma, mean, rstd, dres, dx,
                       (bf16_t*)dx_bf16, dgamma, dbeta, B, n, dim);
    MH_LAUNCH_CHECK();
    return 0;
}
