// Probe / finetune branch (maestro/ssl/mim.py:343-394, maestro/layers/head.py, maestro/train/base.py:98-151):
// token-grid bilinear resize onto the reference grid, attentive / mean reduction over a token axis, the small
// classification linear, and the cross-entropy / BCE-with-logits losses with missing-value filtering.
// All of it is HBM-bound gather / reduce work: coalesced rows of `dim` contiguous values, wavefront reductions.
#include "common.hpp"
#include "../../include/maestro_hip.h"

namespace {

// ------------------------------------------------------------------------------------------- bilinear token resize
// F.interpolate(mode="bilinear", align_corners=False) index map (mim.py:357-366), channel-last.
struct Lerp { int i0, i1; float l0, l1; };
__device__ __forceinline__ Lerp lerp_of(int dst, int n_in, int n_out) {
    const float scale = (float)n_in / (float)n_out;
    const float f = fmaxf(scale * (dst + 0.5f) - 0.5f, 0.f);
    Lerp r;
    r.i0 = min((int)f, n_in - 1);
    r.i1 = r.i0 + (r.i0 < n_in - 1 ? 1 : 0);
    r.l1 = f - r.i0; r.l0 = 1.f - r.l1;
    return r;
}

// in rows (b, in_off + d*h*h + y*h + x) of a [B, in_rows, E] buffer -> out rows (b, out_off + d*H*H + Y*H + X) of [B, out_rows, E]
__global__ __launch_bounds__(256) void token_resize_fwd_kernel(const float* __restrict__ in, long in_rows, int in_off,
                                                               float* __restrict__ out, long out_rows, int out_off, int D,
                                                               int h, int H, int E4, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = i % E4; long r = i / E4;
    const int X = r % H; r /= H; const int Y = r % H; r /= H; const int d = r % D; const long b = r / D;
    const Lerp ly = lerp_of(Y, h, H), lx = lerp_of(X, h, H);
    const f32x4* src = reinterpret_cast<const f32x4*>(in) + ((size_t)b * in_rows + in_off + (size_t)d * h * h) * E4 + c;
    const f32x4 v00 = src[(size_t)(ly.i0 * h + lx.i0) * E4], v01 = src[(size_t)(ly.i0 * h + lx.i1) * E4];
    const f32x4 v10 = src[(size_t)(ly.i1 * h + lx.i0) * E4], v11 = src[(size_t)(ly.i1 * h + lx.i1) * E4];
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = ly.l0 * (lx.l0 * v00[e] + lx.l1 * v01[e]) + ly.l1 * (lx.l0 * v10[e] + lx.l1 * v11[e]);
    reinterpret_cast<f32x4*>(out)[((size_t)b * out_rows + out_off + (size_t)d * H * H + (size_t)Y * H + X) * E4 + c] = o;
}

// Transposed map as a GATHER (deterministic, no atomics): source token (y, x) sums every destination that interpolates
// from it; din[b, in_off + ...] (+)= ...   accumulate = 1 adds to what din already holds.
__global__ __launch_bounds__(256) void token_resize_bwd_kernel(const float* __restrict__ dout, long out_rows, int out_off,
                                                               float* __restrict__ din, long in_rows, int in_off, int D,
                                                               int h, int H, int E4, long total, int accumulate) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = i % E4; long r = i / E4;
    const int x = r % h; r /= h; const int y = r % h; r /= h; const int d = r % D; const long b = r / D;
    const f32x4* src = reinterpret_cast<const f32x4*>(dout) + ((size_t)b * out_rows + out_off + (size_t)d * H * H) * E4 + c;
    f32x4 acc = {0, 0, 0, 0};
    for (int Y = 0; Y < H; ++Y) {
        const Lerp ly = lerp_of(Y, h, H);
        const float wy = (ly.i0 == y ? ly.l0 : 0.f) + (ly.i1 == y ? ly.l1 : 0.f);
        if (wy == 0.f) continue;
        for (int X = 0; X < H; ++X) {
            const Lerp lx = lerp_of(X, h, H);
            const float wx = (lx.i0 == x ? lx.l0 : 0.f) + (lx.i1 == x ? lx.l1 : 0.f);
            if (wx == 0.f) continue;
            acc += (wy * wx) * src[(size_t)(Y * H + X) * E4];
        }
    }
    f32x4* dst = reinterpret_cast<f32x4*>(din) + ((size_t)b * in_rows + in_off + (size_t)d * h * h + (size_t)y * h + x) * E4 + c;
    if (accumulate) acc += *dst;
    *dst = acc;
}

// ------------------------------------------------------------------------------------------- attentive reduction
// head.py:28-62 with heads = 8: sequence s = (b, l), token t in [0, T): row(s, t) = (b*T + t)*Lr + l of kv bf16
// [rows, 2*dim] (k | v).  One wave per sequence: lane = head*8 + j owns VPL = dim/64 consecutive channels of its head;
// the 8 lanes of a head share dots through 3 xor-shuffles; online softmax over t.  lse f32 [n_seq, 8] is kept for backward.
constexpr int AR_SEQ_PER_WAVE = 4;

template <int VPL>
__device__ __forceinline__ void load_bf16_row(const bf16_t* p, float (&v)[VPL]) {
    if constexpr (VPL % 2 == 0) {
#pragma unroll
        for (int i = 0; i < VPL / 2; ++i) {
            const uint32_t pk = reinterpret_cast<const uint32_t*>(p)[i];
            v[2 * i] = __uint_as_float(pk << 16); v[2 * i + 1] = __uint_as_float(pk & 0xffff0000u);
        }
    } else {
#pragma unroll
        for (int i = 0; i < VPL; ++i) v[i] = bf2f(p[i]);
    }
}

__device__ __forceinline__ float head_sum(float v) {   // sum over the 8 lanes of one head
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
    return v;
}

template <int VPL>
__global__ __launch_bounds__(256) void attn_reduce_fwd_kernel(const bf16_t* __restrict__ kv, const float* __restrict__ query,
                                                              float* __restrict__ out, float* __restrict__ lse, int n_seq,
                                                              int T, int Lr, float scale) {
    constexpr int dim = 64 * VPL;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float q[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) q[i] = query[lane * VPL + i] * scale;
    for (int rr = 0; rr < AR_SEQ_PER_WAVE; ++rr) {
        const int s = (blockIdx.x * 4 + w) * AR_SEQ_PER_WAVE + rr;
        if (s >= n_seq) break;
        const int b = s / Lr, l = s - b * Lr;
        float m = -INFINITY, sum = 0.f, acc[VPL];
#pragma unroll
        for (int i = 0; i < VPL; ++i) acc[i] = 0.f;
        for (int t = 0; t < T; ++t) {
            const bf16_t* row = kv + ((size_t)((size_t)b * T + t) * Lr + l) * (2 * dim) + lane * VPL;
            float k[VPL], v[VPL];
            load_bf16_row<VPL>(row, k);
            load_bf16_row<VPL>(row + dim, v);
            float dot = 0.f;
#pragma unroll
            for (int i = 0; i < VPL; ++i) dot += q[i] * k[i];
            dot = head_sum(dot);
            const float m_new = fmaxf(m, dot);
            const float c = __expf(m - m_new), p = __expf(dot - m_new);
            sum = sum * c + p;
#pragma unroll
            for (int i = 0; i < VPL; ++i) acc[i] = acc[i] * c + p * v[i];
            m = m_new;
        }
        const float inv = 1.f / sum;
#pragma unroll
        for (int i = 0; i < VPL; ++i) out[(size_t)s * dim + lane * VPL + i] = acc[i] * inv;
        if ((lane & 7) == 0) lse[(size_t)s * 8 + (lane >> 3)] = m + __logf(sum);
    }
}

// Backward: dkv bf16 [rows, 2*dim] (dk | dv), dq partial rows [gridDim.x, dim] (summed by mh_colsum afterwards).
template <int VPL>
__global__ __launch_bounds__(256) void attn_reduce_bwd_kernel(const bf16_t* __restrict__ kv, const float* __restrict__ query,
                                                              const float* __restrict__ out, const float* __restrict__ lse,
                                                              const float* __restrict__ dout, bf16_t* __restrict__ dkv,
                                                              float* __restrict__ dq_partial, int n_seq, int T, int Lr,
                                                              float scale) {
    constexpr int dim = 64 * VPL;
    __shared__ float red[4][dim];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float q[VPL], dq[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) { q[i] = query[lane * VPL + i] * scale; dq[i] = 0.f; }
    for (int rr = 0; rr < AR_SEQ_PER_WAVE; ++rr) {
        const int s = (blockIdx.x * 4 + w) * AR_SEQ_PER_WAVE + rr;
        if (s >= n_seq) break;
        const int b = s / Lr, l = s - b * Lr;
        float g[VPL], delta = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            g[i] = dout[(size_t)s * dim + lane * VPL + i];
            delta += g[i] * out[(size_t)s * dim + lane * VPL + i];
        }
        delta = head_sum(delta);
        const float ls = lse[(size_t)s * 8 + (lane >> 3)];
        for (int t = 0; t < T; ++t) {
            const size_t off = ((size_t)((size_t)b * T + t) * Lr + l) * (2 * dim) + lane * VPL;
            float k[VPL], v[VPL];
            load_bf16_row<VPL>(kv + off, k);
            load_bf16_row<VPL>(kv + off + dim, v);
            float dot = 0.f, dp = 0.f;
#pragma unroll
            for (int i = 0; i < VPL; ++i) { dot += q[i] * k[i]; dp += g[i] * v[i]; }
            dot = head_sum(dot); dp = head_sum(dp);
            const float p = __expf(dot - ls), ds = p * (dp - delta);
#pragma unroll
            for (int i = 0; i < VPL; ++i) {
                dkv[off + i] = f2bf(ds * q[i]);
                dkv[off + dim + i] = f2bf(p * g[i]);
                dq[i] += ds * k[i];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < VPL; ++i) red[w][lane * VPL + i] = dq[i] * scale;
    __syncthreads();
    for (int c = threadIdx.x; c < dim; c += 256)
        dq_partial[(size_t)blockIdx.x * dim + c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}

// ------------------------------------------------------------------------------------------- mean reduction ("linear" heads)
__global__ __launch_bounds__(256) void mean_reduce_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, int T,
                                                              int Lr, int dim4, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = i % dim4; const long s = i / dim4; const long b = s / Lr; const int l = s - b * Lr;
    f32x4 acc = {0, 0, 0, 0};
    for (int t = 0; t < T; ++t) acc += reinterpret_cast<const f32x4*>(x)[((size_t)(b * T + t) * Lr + l) * dim4 + c];
    reinterpret_cast<f32x4*>(out)[i] = acc * (1.f / T);
}
__global__ __launch_bounds__(256) void mean_reduce_bwd_kernel(const float* __restrict__ dout, float* __restrict__ dx, int T,
                                                              int Lr, int dim4, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;   // one thread per float4 of dx [rows, dim]
    if (i >= total) return;
    const int c = i % dim4; const long row = i / dim4;
    const int l = row % Lr; const long b = row / ((long)T * Lr);
    reinterpret_cast<f32x4*>(dx)[i] = reinterpret_cast<const f32x4*>(dout)[((size_t)b * Lr + l) * dim4 + c] * (1.f / T);
}

// ------------------------------------------------------------------------------------------- small classification linear
// out[b, c] = bias[c] + sum_e x[b, e] W[c, e]  (head.py:83,93; num_classes is small and odd: fp32, one wave per output)
__global__ __launch_bounds__(256) void head_linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                              const float* __restrict__ bias, float* __restrict__ out, int B,
                                                              int C, int E) {
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (o >= B * C) return;
    const int b = o / C, c = o - b * C;
    float s = 0.f;
    for (int e = lane; e < E; e += 64) s += x[(size_t)b * E + e] * W[(size_t)c * E + e];
    s = wave_sum(s);
    if (lane == 0) out[o] = s + bias[c];
}
// dx[b, e] = sum_c dout[b, c] W[c, e] (optional); dW[c, e] += sum_b dout[b, c] x[b, e]; db[c] += sum_b dout[b, c]
__global__ __launch_bounds__(256) void head_linear_bwd_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                              const float* __restrict__ dout, float* __restrict__ dx,
                                                              float* __restrict__ dW, float* __restrict__ db, int B, int C,
                                                              int E) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < C * E) {
        const int c = i / E, e = i - c * E;
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += dout[b * C + c] * x[(size_t)b * E + e];
        dW[i] += s;
        if (e == 0) {
            float t = 0.f;
            for (int b = 0; b < B; ++b) t += dout[b * C + c];
            db[c] += t;
        }
    }
    if (dx && i < B * E) {
        const int b = i / E, e = i - b * E;
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += dout[b * C + c] * W[(size_t)c * E + e];
        dx[i] = s;
    }
}

// ------------------------------------------------------------------------------------------- losses (base.py:98-151)
__device__ __forceinline__ long load_int(const void* p, long i, int bytes) {
    switch (bytes) {
        case 1: return reinterpret_cast<const int8_t*>(p)[i];
        case 2: return reinterpret_cast<const int16_t*>(p)[i];
        case 4: return reinterpret_cast<const int32_t*>(p)[i];
        default: return reinterpret_cast<const int64_t*>(p)[i];
    }
}

__global__ __launch_bounds__(256) void count_valid_kernel(const void* __restrict__ target, int tbytes, long n, long missing,
                                                          int* __restrict__ count) {
    __shared__ int red[4];
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    int v = (i < n && load_int(target, i, tbytes) != missing) ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0 && red[0] + red[1] + red[2] + red[3]) atomicAdd(count, red[0] + red[1] + red[2] + red[3]);
}

// Pixel (b, Y, X) of a [B, S, S] target raster <-> logits at patch layout [B*g*g, P*P*C] (token (Y/P)*g + X/P, columns
// ((Y%P)*P + X%P)*C + c: PixelifyBands' '(p1 p2 c)' order, embed.py:153-160); S = g*P.  Classification = g = P = 1.
// acc += mean over valid pixels of (logsumexp - logit[target]);  dlogits = (softmax - onehot) / n_valid, 0 where missing.
__global__ __launch_bounds__(256) void ce_loss_kernel(const float* __restrict__ logits, const void* __restrict__ target,
                                                      int tbytes, long missing, const int* __restrict__ n_valid,
                                                      float* __restrict__ acc, void* __restrict__ dlogits, int d_is_f32,
                                                      long n_pix, int g, int P, int C, int ld) {
    __shared__ float red[4];
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int nv = *n_valid;
    float loss = 0.f;
    if (i < n_pix) {
        const int S = g * P;
        const int X = i % S; const long r = i / S; const int Y = r % S; const long b = r / S;
        const size_t base = (size_t)(b * g * g + (size_t)(Y / P) * g + X / P) * ld + ((size_t)(Y % P) * P + X % P) * C;
        const long t = load_int(target, i, tbytes);
        const bool valid = t != missing && t >= 0 && t < C && nv > 0;
        float m = -INFINITY;
        if (valid) {
            for (int c = 0; c < C; ++c) m = fmaxf(m, logits[base + c]);
            float s = 0.f;
            for (int c = 0; c < C; ++c) s += __expf(logits[base + c] - m);
            const float lse = m + __logf(s), inv = 1.f / nv;
            loss = (lse - logits[base + t]) * inv;
            for (int c = 0; c < C; ++c) {
                const float d = (__expf(logits[base + c] - lse) - (c == t ? 1.f : 0.f)) * inv;
                if (d_is_f32) reinterpret_cast<float*>(dlogits)[base + c] = d;
                else reinterpret_cast<bf16_t*>(dlogits)[base + c] = f2bf(d);
            }
        } else {
            for (int c = 0; c < C; ++c) {
                if (d_is_f32) reinterpret_cast<float*>(dlogits)[base + c] = 0.f;
                else reinterpret_cast<bf16_t*>(dlogits)[base + c] = 0;
            }
        }
    }
    loss = wave_sum(loss);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = loss;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float t = (red[0] + red[1]) + (red[2] + red[3]);
        if (t != 0.f) atomicAdd(acc, t);
    }
}

// Multilabel: logits / targets f32 [B, C]; a row is used iff none of its targets equals missing (base.py:122-123);
// acc += mean over used rows x C of BCE-with-logits; dlogits f32 = (sigmoid(x) - t) / (n_rows * C).  One block.
__global__ __launch_bounds__(256) void bce_loss_kernel(const float* __restrict__ logits, const float* __restrict__ target,
                                                       float missing, float* __restrict__ acc, float* __restrict__ dlogits,
                                                       int B, int C) {
    __shared__ int n_rows;
    __shared__ float red[4];
    if (threadIdx.x == 0) n_rows = 0;
    __syncthreads();
    for (int b = threadIdx.x; b < B; b += 256) {
        bool ok = true;
        for (int c = 0; c < C; ++c) ok = ok && target[b * C + c] != missing;
        if (ok) atomicAdd(&n_rows, 1);
    }
    __syncthreads();
    const int nr = n_rows;
    const float inv = nr > 0 ? 1.f / ((float)nr * C) : 0.f;
    float loss = 0.f;
    for (int i = threadIdx.x; i < B * C; i += 256) {
        const int b = i / C;
        bool ok = true;
        for (int c = 0; c < C; ++c) ok = ok && target[b * C + c] != missing;
        float d = 0.f;
        if (ok) {
            const float x = logits[i], t = target[i];
            loss += (fmaxf(x, 0.f) - x * t + log1pf(__expf(-fabsf(x)))) * inv;   // stable BCE-with-logits
            d = (1.f / (1.f + __expf(-x)) - t) * inv;
        }
        dlogits[i] = d;
    }
    loss = wave_sum(loss);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = loss;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(acc, (red[0] + red[1]) + (red[2] + red[3]));
}

}  // namespace

extern "C" int mh_token_resize(const float* in, long in_rows, int in_off, float* out, long out_rows, int out_off, int B,
                               int D, int h, int H, int E, void* stream) {
    MH_CHECK_ARG(in && out && in != out && B > 0 && D > 0 && h > 0 && H > 0 && E % 4 == 0, "mh_token_resize: bad arguments");
    MH_CHECK_ARG(in_off + (long)D * h * h <= in_rows && out_off + (long)D * H * H <= out_rows, "mh_token_resize: rows out of range");
    const long total = (long)B * D * H * H * (E / 4);
    hipLaunchKernelGGL(token_resize_fwd_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, in, in_rows,
                       in_off, out, out_rows, out_off, D, h, H, E / 4, total);
    MH_LAUNCH_CHECK();
    return 0;
}

extern "C" int mh_token_resize_bwd(const float* dout, long out_rows, int out_off, float* din, long in_rows, int in_off,
                                   int B, int D, int h, int H, int E, int accumulate, void* stream) {
    MH_CHECK_ARG(dout && din && dout != din && B > 0 && D > 0 && h > 0 && H > 0 && E % 4 == 0, "mh_token_resize_bwd: bad arguments");
    MH_CHECK_ARG(in_off + (long)D * h * h <= in_rows && out_off + (long)D * H * H <= out_rows, "mh_token_resize_bwd: rows out of range");
    const long total = (long)B * D * h * h * (E / 4);
    hipLaunchKernelGGL(token_resize_bwd_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, dout, out_rows,
                       out_off, din, in_rows, in_off, D, h, H, E / 4, total, accumulate);
    MH_LAUNCH_CHECK();
    return 0;
}

extern "C" long mh_attn_reduce_partial_rows(int n_seq) { return ceil_div(n_seq, 4 * AR_SEQ_PER_WAVE); }

#define AR_DISPATCH(KERNEL, ...)                                                                                   \
    switch (dim / 64) {                                                                                            \
        case 3: hipLaunchKernelGGL(KERNEL<3>, grid, dim3(256), 0, s, __VA_ARGS__); break;                          \
        case 6: hipLaunchKernelGGL(KERNEL<6>, grid, dim3(256), 0, s, __VA_ARGS__); break;                          \
        case 12: hipLaunchKernelGGL(KERNEL<12>, grid, dim3(256), 0, s, __VA_ARGS__); break;                        \
        case 16: hipLaunchKernelGGL(KERNEL<16>, grid, dim3(256), 0, s, __VA_ARGS__); break;                        \
        default: return mh_fail(-1, "mh_attn_reduce: dim %d (192, 384, 768 or 1024: the reference's embed dims)", dim); \
    }

extern "C" int mh_attn_reduce_fwd(const void* kv, const float* query, float* out, float* lse, int n_batch, int T, int Lr,
                                  int dim, int heads, void* stream) {
    MH_CHECK_ARG(kv && query && out && lse && n_batch > 0 && T > 0 && Lr > 0, "mh_attn_reduce_fwd: bad arguments");
    MH_CHECK_ARG(heads == 8 && dim % 64 == 0, "mh_attn_reduce_fwd: heads must be 8 (head.py:31) and dim a multiple of 64");
    const int n_seq = n_batch * Lr;
    const float scale = 1.f / sqrtf((float)(dim / heads));
    dim3 grid(ceil_div(n_seq, 4 * AR_SEQ_PER_WAVE));
    hipStream_t s = (hipStream_t)stream;
    AR_DISPATCH(attn_reduce_fwd_kernel, (const bf16_t*)kv, query, out, lse, n_seq, T, Lr, scale)
    MH_LAUNCH_CHECK();
    return 0;
}

extern "C" int mh_attn_reduce_bwd(const void* kv, const float* query, const float* out, const float* lse, const float* dout,
                                  void* dkv, float* dq_partial, int n_batch, int T, int Lr, int dim, int heads, void* stream) {
    MH_CHECK_ARG(kv && query && out && lse && dout && dkv && dq_partial && n_batch > 0 && T > 0 && Lr > 0,
                 "mh_attn_reduce_bwd: bad arguments");
    MH_CHECK_ARG(heads == 8 && dim % 64 == 0, "mh_attn_reduce_bwd: heads must be 8 and dim a multiple of 64");
    const int n_seq = n_batch * Lr;
    const float scale = 1.f / sqrtf((float)(dim / heads));
    dim3 grid(ceil_div(n_seq, 4 * AR_SEQ_PER_WAVE));
    hipStream_t s = (hipStream_t)stream;
    AR_DISPATCH(attn_reduce_bwd_kernel, (const bf16_t*)kv, query, out, lse, dout, (bf16_t*)dkv, dq_partial, n_seq, T, Lr, scale)
    MH_LAUNCH_CHECK();
    return 0;
}

extern "C" int mh_mean_reduce_fwd(const float* x, float* out, int n_batch, int T, int Lr, int dim, void* stream) {
    MH_CHECK_ARG(x && out && n_batch > 0 && T > 0 && Lr > 0 && dim % 4 == 0, "mh_mean_reduce_fwd: bad arguments");
    const long total = (long)n_batch * Lr * (dim / 4);
    hipLaunchKernelGGL(mean_reduce_fwd_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, x, out, T, Lr,
                       dim / 4, total);
    MH_LAUNCH_CHECK();
    return 0;
}

extern "C" int mh_mean_reduce_bwd(const float* dout, float* dx, int n_batch, int T, int Lr, int dim, void* stream) {
    MH_CHECK_ARG(dout && dx && n_batch > 0 && T > 0 && Lr > 0 && dim % 4 == 0, "mh_mean_reduce_bwd: bad arguments");
    const long total = (long)n_batch * T * Lr * (dim / 4);
    hipLaunchKernelGGL(mean_reduce_bwd_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, dout, dx, T, Lr,
                       dim / 4, total);
    MH_LAUNCH_CHECK();
    return 0;
}

extern "C" int mh_head_linear_fwd(const float* x, const float* W, const float* bias, float* out, int B, int C, int E,
                                  void* stream) {
    MH_CHECK_ARG(x && W && bias && out && B > 0 && C > 0 && E > 0, "mh_head_linear_fwd: bad arguments");
    hipLaunchKernelGGL(head_linear_fwd_kernel, dim3(ceil_div(B * C, 4)), dim3(256), 0, (hipStream_t)stream, x, W, bias, out, B, C, E);
    MH_LAUNCH_CHECK();
    return 0;
}

extern "C" int mh_head_linear_bwd(const float* x, const float* W, const float* dout, float* dx, float* dW, float* db, int B,
                                  int C, int E, void* stream) {
    MH_CHECK_ARG(x && W && dout && dW && db && B > 0 && C > 0 && E > 0, "mh_head_linear_bwd: bad arguments");
    const int n = max(C * E, dx ? B * E : 0);
    hipLaunchKernelGGL(head_linear_bwd_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, x, W, dout, dx, dW, db,
                       B, C, E);
    MH_LAUNCH_CHECK();
    return 0;
}

extern "C" int mh_count_valid(const void* target, int target_bytes, long n, long missing_val, int* count, void* stream) {
    MH_CHECK_ARG(target && count && n > 0 && (target_bytes == 1 || target_bytes == 2 || target_bytes == 4 || target_bytes == 8),
                 "mh_count_valid: bad arguments");
    hipLaunchKernelGGL(count_valid_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, target, target_bytes, n,
                       missing_val, count);
    MH_LAUNCH_CHECK();
    return 0;
}

extern "C" int mh_ce_loss(const float* logits, const void* target, int target_bytes, long missing_val, const int* n_valid,
                          float* acc, void* dlogits, int dlogits_is_f32, int B, int g, int P, int C, int ld, void* stream) {
    MH_CHECK_ARG(logits && target && n_valid && acc && dlogits && B > 0 && g > 0 && P > 0 && C > 0 && ld >= P * P * C,
                 "mh_ce_loss: bad arguments");
    MH_CHECK_ARG(target_bytes == 1 || target_bytes == 2 || target_bytes == 4 || target_bytes == 8, "mh_ce_loss: target width");
    const long n_pix = (long)B * g * P * g * P;
    hipLaunchKernelGGL(ce_loss_kernel, dim3(ceil_div(n_pix, 256)), dim3(256), 0, (hipStream_t)stream, logits, target, target_bytes,
                       missing_val, n_valid, acc, dlogits, dlogits_is_f32, n_pix, g, P, C, ld);
    MH_LAUNCH_CHECK();
    return 0;
}

extern "C" int mh_bce_loss(const float* logits, const float* target, float missing_val, float* acc, float* dlogits, int B,
                           int C, void* stream) {
    MH_CHECK_ARG(logits && target && acc && dlogits && B > 0 && C > 0, "mh_bce_loss: bad arguments");
    hipLaunchKernelGGL(bce_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, target, missing_val, acc, dlogits, B, C);
    MH_LAUNCH_CHECK();
    return 0;
}
