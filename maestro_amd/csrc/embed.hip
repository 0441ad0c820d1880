// Patch extraction, target normalisation, patch-embed GroupNorm (+ encodings) forward/backward.
// See include/maestro_hip.h for the reference lines each entry point replaces.  All HBM-bound: coalesced row reads,
// LDS transposes where the output order differs from the input order, wavefront reductions for statistics.
#include "common.hpp"
#include "../../include/maestro_hip.h"

namespace {

// Generic block reduction for blockDim.x in {64, 128, 256}.
__device__ __forceinline__ float block_sum_any(float v, float* red) {
    v = wave_sum(v);
    const int nw = blockDim.x >> 6;
    if (nw == 1) return v;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}

// One block per patch (token).  LDS holds the patch in TARGET order [(p1*P+p2)*Ctot + c].
// Band windows (several band-groups per modality, maestro/layers/embed.py:18-34): the image has Csrc channels, the patch takes
// channels c0 .. c0 + Ctot - 1 of it; the elevation rescale always refers to channel 0 of the IMAGE.
__global__ void patchify_kernel(const float* __restrict__ img, bf16_t* __restrict__ cols, float* __restrict__ target,
                                int Csrc, int c0, int Ctot, int S, int P, int Kpad, const int* __restrict__ norm_bands,
                                int n_groups, int normalise, int rescale_elev) {
    extern __shared__ __attribute__((aligned(16))) float patch[];  // P*P*Ctot floats + 8 reduction slots
    const int g = S / P;
    const int tok = blockIdx.x;  // (bd, ph, pw)
    const int bd = tok / (g * g), pp = tok - bd * g * g, ph = pp / g, pw = pp - ph * g;
    const int PP = P * P, K = Ctot * PP;
    float* red = patch + K;
    const float* base0 = img + ((size_t)bd * Csrc) * S * S + (size_t)(ph * P) * S + pw * P;   // channel 0 of the image
    const float* base = base0 + (size_t)c0 * S * S;
    if ((P & 3) == 0 && (Kpad & 3) == 0) {
        // four consecutive pixels of one patch row per thread: 16-byte image loads, 8-byte bf16 stores
        for (int k = threadIdx.x * 4; k < Kpad; k += blockDim.x * 4) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (k < K) {
                const int c = k / PP, r = k - c * PP, p1 = r / P, p2 = r - p1 * P;
                v = *reinterpret_cast<const f32x4*>(base + (size_t)c * S * S + p1 * S + p2);
                if (rescale_elev && c0 + c >= 1) v = 30.f * (*reinterpret_cast<const f32x4*>(base0 + p1 * S + p2) - v);
#pragma unroll
                for (int e = 0; e < 4; ++e) patch[(r + e) * Ctot + c] = v[e];
            }
            if (cols) {
                u32x2 pk = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
                *reinterpret_cast<u32x2*>(cols + (size_t)tok * Kpad + k) = pk;
            }
        }
    } else {
        for (int k = threadIdx.x; k < Kpad; k += blockDim.x) {
            float v = 0.f;
            if (k < K) {
                const int c = k / PP, r = k - c * PP, p1 = r / P, p2 = r - p1 * P;
                v = base[(size_t)c * S * S + p1 * S + p2];
                if (rescale_elev && c0 + c >= 1) v = 30.f * (base0[p1 * S + p2] - v);
                patch[r * Ctot + c] = v;
            }
            if (cols) cols[(size_t)tok * Kpad + k] = f2bf(v);
        }
    }
    if (!target) return;
    __syncthreads();
    float* out = target + (size_t)tok * K;
    if (!normalise) {
        for (int e = threadIdx.x; e < K; e += blockDim.x) out[e] = patch[e];
        return;
    }
    // patch-group-wise statistics: unbiased variance, eps 1e-6 (reference model.py:226-229), two-pass; a thread owns whole
    // pixels of the group (no per-element div / mod)
    int c_lo = 0;
    for (int gi = 0; gi < n_groups; ++gi) {
        const int cg = norm_bands[gi], n = cg * PP;
        float s = 0.f;
        for (int pix = threadIdx.x; pix < PP; pix += blockDim.x)
            for (int cc = 0; cc < cg; ++cc) s += patch[pix * Ctot + c_lo + cc];
        const float mu = block_sum_any(s, red) / n;
        float q = 0.f;
        for (int pix = threadIdx.x; pix < PP; pix += blockDim.x)
            for (int cc = 0; cc < cg; ++cc) {
                const float d = patch[pix * Ctot + c_lo + cc] - mu;
                q += d * d;
            }
        const float var = block_sum_any(q, red) / (n - 1);
        const float inv = 1.f / sqrtf(var + 1.0e-6f);
        for (int pix = threadIdx.x; pix < PP; pix += blockDim.x)
            for (int cc = 0; cc < cg; ++cc) {
                const int idx = pix * Ctot + c_lo + cc;
                out[idx] = (patch[idx] - mu) * inv;
            }
        c_lo += cg;
    }
}

// P = 16 patches (aerial / spot rasters: 1024 x C elements per token), ONE WAVE per patch, no LDS, no barriers: lane l owns the four
// consecutive pixels (row l >> 2, columns 4 (l & 3) ..) of the patch in every channel, i.e. one 16-byte image load per channel,
// one 8-byte store per channel into the K-ordered bf16 columns (a wave writes 512 contiguous bytes) and -- because a lane holds
// ALL channels of its pixels -- 16 CT contiguous bytes of the pixel-major fp32 target (a wave writes the patch's 4 CT KiB in one
// run).  The patch-group statistics are two wave reductions per group.  The block-per-patch kernel above (LDS transpose, four
// block reductions = eight barriers per 4 KiB patch) ran at 2.0 TB/s on the C3 aerial launch (335 MB in 165 us).
template <int CT>
__global__ __launch_bounds__(256) void patchify_wave16_kernel(const float* __restrict__ img, bf16_t* __restrict__ cols,
                                                              float* __restrict__ target, int Csrc, int c0, int S, int Kpad,
                                                              const int* __restrict__ norm_bands, int n_groups, int normalise,
                                                              int rescale_elev, int n_tok) {
    constexpr int P = 16, PP = 256, K = CT * PP;
    const int tok = blockIdx.x * 4 + (threadIdx.x >> 6), l = threadIdx.x & 63;
    if (tok >= n_tok) return;
    const int g = S / P;
    const int bd = tok / (g * g), pp = tok - bd * g * g, ph = pp / g, pw = pp - ph * g;
    const int p1 = l >> 2, p2 = (l & 3) * 4;
    const size_t plane = (size_t)S * S;
    const float* base0 = img + (size_t)bd * Csrc * plane + (size_t)(ph * P + p1) * S + pw * P + p2;   // channel 0 of the image
    f32x4 v[CT];
    f32x4 elev = {0.f, 0.f, 0.f, 0.f};
    if (rescale_elev) elev = *reinterpret_cast<const f32x4*>(base0);
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        v[c] = *reinterpret_cast<const f32x4*>(base0 + (size_t)(c0 + c) * plane);
        if (rescale_elev && c0 + c >= 1) v[c] = 30.f * (elev - v[c]);
    }
    if (cols) {
        bf16_t* crow = cols + (size_t)tok * Kpad;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const u32x2 pk = {pack_bf2(v[c][0], v[c][1]), pack_bf2(v[c][2], v[c][3])};
            *reinterpret_cast<u32x2*>(crow + c * PP + 4 * l) = pk;
        }
        for (int k = K + 4 * l; k < Kpad; k += 256) *reinterpret_cast<u32x2*>(crow + k) = (u32x2){0u, 0u};   // K padding
    }
    if (!target) return;
    if (normalise) {   // patch-group-wise statistics: unbiased variance, eps 1e-6 (reference model.py:226-229), two-pass
        int c_lo = 0;
        for (int gi = 0; gi < n_groups; ++gi) {
            const int cg = norm_bands[gi], n = cg * PP;
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < CT; ++c)
                if (c >= c_lo && c < c_lo + cg) s += (v[c][0] + v[c][1]) + (v[c][2] + v[c][3]);
            const float mu = wave_sum(s) / n;
            float q = 0.f;
#pragma unroll
            for (int c = 0; c < CT; ++c)
                if (c >= c_lo && c < c_lo + cg) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float d = v[c][e] - mu; q += d * d; }
                }
            const float inv = 1.f / sqrtf(wave_sum(q) / (n - 1) + 1.0e-6f);
#pragma unroll
            for (int c = 0; c < CT; ++c)
                if (c >= c_lo && c < c_lo + cg) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[c][e] = (v[c][e] - mu) * inv;
                }
            c_lo += cg;
        }
    }
    float t[4 * CT];   // the lane's four pixels, channel-minor: element (4 l + e) * CT + c of the patch
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int c = 0; c < CT; ++c) t[e * CT + c] = v[c][e];
    float* out = target + (size_t)tok * K + (size_t)l * 4 * CT;
#pragma unroll
    for (int i = 0; i < CT; ++i) *reinterpret_cast<f32x4*>(out + 4 * i) = (f32x4){t[4 * i], t[4 * i + 1], t[4 * i + 2], t[4 * i + 3]};
}

// ---- GroupNorm(1, E) statistics over a whole (L x E) image: chunked partial sums, then a finalize.
constexpr int GN_CHUNK = 8192;

__global__ __launch_bounds__(256) void gn_partial_kernel(const float* __restrict__ y, float* __restrict__ partial,
                                                         long per_image, int nchunk) {
    __shared__ float red[8];
    const int bd = blockIdx.y, ch = blockIdx.x;
    const float* p = y + (size_t)bd * per_image;
    const long lo = (long)ch * GN_CHUNK, hi = min(per_image, lo + GN_CHUNK);
    float s = 0.f, q = 0.f;
    for (long i = lo + threadIdx.x * 4; i < hi; i += 256 * 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(p + i);
        s += (v[0] + v[1]) + (v[2] + v[3]);
        q += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    }
    s = block_sum<4>(s, red);
    q = block_sum<4>(q, red + 4);
    if (threadIdx.x == 0) {
        partial[((size_t)bd * nchunk + ch) * 2] = s;
        partial[((size_t)bd * nchunk + ch) * 2 + 1] = q;
    }
}

__global__ __launch_bounds__(64) void gn_finalize_kernel(const float* __restrict__ partial, float* __restrict__ stats,
                                                         long per_image, int nchunk, float eps) {
    const int bd = blockIdx.x;
    double s = 0.0, q = 0.0;
    for (int i = threadIdx.x; i < nchunk; i += 64) {
        s += partial[((size_t)bd * nchunk + i) * 2];
        q += partial[((size_t)bd * nchunk + i) * 2 + 1];
    }
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
    if (threadIdx.x == 0) {
        const double mu = s / per_image;
        const double var = fmax(q / per_image - mu * mu, 0.0);
        stats[bd * 2] = (float)mu;
        stats[bd * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

// ---- normalise + affine + positional + date encodings, written into the group sequence. One wave per token.
__global__ __launch_bounds__(256) void embed_finish_kernel(const float* __restrict__ y, const float* __restrict__ stats,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ pos, const float* __restrict__ date,
                                                           float* __restrict__ xg, int B, int D, int L, int E, int tok_off,
                                                           int Lgroup, int date_rows, int date_off) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= B * D * L) return;
    const int bd = row / L, l = row - bd * L, b = bd / D, d = bd - b * D;
    const float mu = stats[bd * 2], rs = stats[bd * 2 + 1];
    const float* yr = y + (size_t)row * E;
    float* o = xg + ((size_t)b * Lgroup + tok_off + d * L + l) * E;
    const float* pr = pos + (size_t)l * E;
    const float* dr = date ? date + ((size_t)b * date_rows + date_off + d) * 8 : nullptr;
    for (int c = lane * 4; c < E; c += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(yr + c);
        const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c);
        const f32x4 bt = *reinterpret_cast<const f32x4*>(beta + c);
        const f32x4 ps = *reinterpret_cast<const f32x4*>(pr + c);
        f32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = (v[e] - mu) * rs * g[e] + bt[e] + ps[e];
        if (dr && c >= E - 8) r += *reinterpret_cast<const f32x4*>(dr + (c - (E - 8)));
        *reinterpret_cast<f32x4*>(o + c) = r;
    }
}

// ---- backward pass 1: per image S1 = sum(dz), S2 = sum(dz*z); dgamma/dbeta (atomic). ROWS tokens per wave.
constexpr int EB_ROWS = 8;
// NV = ceil(E / 256) four-column chunks per lane (round 6): a lane owns the same columns in every row, so the dgamma / dbeta partial sums
// of its EB_ROWS rows stay in registers and touch LDS once at the end (they used to be read-modify-written in LDS for every element:
// two LDS round trips per element in a kernel that streams 8 bytes per element from HBM).  NV = 0: any E (the LDS form).
template <int NV>
__global__ __launch_bounds__(256) void embed_bwd_stats_kernel(const float* __restrict__ dxg, const float* __restrict__ y,
                                                              const float* __restrict__ stats, const float* __restrict__ gamma,
                                                              float* __restrict__ sums, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, int B, int D, int L, int E,
                                                              int tok_off, int Lgroup) {
    extern __shared__ __attribute__((aligned(16))) float red[];  // [4][2][E]
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int bd = blockIdx.y, b = bd / D, d = bd - b * D;
    const float mu = stats[bd * 2], rs = stats[bd * 2 + 1];
    float* rg = red + (size_t)w * 2 * E;
    float s1 = 0.f, s2 = 0.f;
    const int l0 = (blockIdx.x * 4 + w) * EB_ROWS;
    if constexpr (NV > 0) {
        f32x4 gz[NV], gd[NV], gm[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = lane * 4 + 256 * j;
            gz[j] = (f32x4){0, 0, 0, 0}; gd[j] = (f32x4){0, 0, 0, 0};
            gm[j] = c < E ? *reinterpret_cast<const f32x4*>(gamma + c) : (f32x4){0, 0, 0, 0};
        }
#pragma unroll 2
        for (int rr = 0; rr < EB_ROWS; ++rr) {
            const int l = l0 + rr;
            if (l >= L) break;
            const float* yr = y + ((size_t)bd * L + l) * E;
            const float* gr = dxg + ((size_t)b * Lgroup + tok_off + d * L + l) * E;
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int c = lane * 4 + 256 * j;
                if (c < E) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(yr + c);
                    const f32x4 dd = *reinterpret_cast<const f32x4*>(gr + c);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float z = (v[e] - mu) * rs, dz = dd[e] * gm[j][e];
                        s1 += dz; s2 += dz * z;
                        gz[j][e] += dd[e] * z;
                        gd[j][e] += dd[e];
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = lane * 4 + 256 * j;
            if (c < E) {
                *reinterpret_cast<f32x4*>(rg + c) = gz[j];
                *reinterpret_cast<f32x4*>(rg + E + c) = gd[j];
            }
        }
    } else {
        for (int c = lane; c < 2 * E; c += 64) rg[c] = 0.f;
        for (int rr = 0; rr < EB_ROWS; ++rr) {
            const int l = l0 + rr;
            if (l >= L) break;
            const float* yr = y + ((size_t)bd * L + l) * E;
            const float* gr = dxg + ((size_t)b * Lgroup + tok_off + d * L + l) * E;
            for (int c = lane * 4; c < E; c += 256) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(yr + c);
                const f32x4 dd = *reinterpret_cast<const f32x4*>(gr + c);
                const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float z = (v[e] - mu) * rs, dz = dd[e] * gm[e];
                    s1 += dz; s2 += dz * z;
                    rg[c + e] += dd[e] * z;       // lane-private columns: no race inside the wave
                    rg[E + c + e] += dd[e];
                }
            }
        }
    }
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    if (lane == 0) { atomicAdd(sums + bd * 2, s1); atomicAdd(sums + bd * 2 + 1, s2); }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * E; c += 256) {
        const float t = red[c] + red[2 * E + c] + red[4 * E + c] + red[6 * E + c];
        if (c < E) atomicAdd(dgamma + c, t); else atomicAdd(dbeta + (c - E), t);
    }
}

__global__ __launch_bounds__(256) void zero_f32_kernel(float* __restrict__ p, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = 0.f;
}

// ---- backward pass 2: dy = rstd * (dz - S1/n - z*S2/n) as bf16 (A operand of the patch-embed wgrad GEMM)
__global__ __launch_bounds__(256) void embed_bwd_apply_kernel(const float* __restrict__ dxg, const float* __restrict__ y,
                                                              const float* __restrict__ stats, const float* __restrict__ gamma,
                                                              const float* __restrict__ sums, bf16_t* __restrict__ dyc,
                                                              int B, int D, int L, int E, int tok_off, int Lgroup) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= B * D * L) return;
    const int bd = row / L, l = row - bd * L, b = bd / D, d = bd - b * D;
    const float mu = stats[bd * 2], rs = stats[bd * 2 + 1];
    const float n = (float)L * (float)E;
    const float c1 = sums[bd * 2] / n, c2 = sums[bd * 2 + 1] / n;
    const float* yr = y + (size_t)row * E;
    const float* gr = dxg + ((size_t)b * Lgroup + tok_off + d * L + l) * E;
    bf16_t* o = dyc + (size_t)row * E;
    for (int c = lane * 4; c < E; c += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(yr + c);
        const f32x4 dd = *reinterpret_cast<const f32x4*>(gr + c);
        const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + c);
        float r[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float z = (v[e] - mu) * rs;
            r[e] = rs * (dd[e] * gm[e] - c1 - z * c2);
        }
        u32x2 pk = {pack_bf2(r[0], r[1]), pack_bf2(r[2], r[3])};
        *reinterpret_cast<u32x2*>(o + c) = pk;
    }
}

// ---- patch layout -> image layout
__global__ __launch_bounds__(256) void depatchify_kernel(const float* __restrict__ patches, float* __restrict__ img,
                                                         int C, int S, int P, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;  // index into img [BD, C, S, S]
    if (i >= total) return;
    const int x = i % S; long r = i / S;
    const int yy = r % S; r /= S;
    const int c = r % C; const long bd = r / C;
    const int g = S / P, ph = yy / P, p1 = yy - ph * P, pw = x / P, p2 = x - pw * P;
    img[i] = patches[((bd * g + ph) * g + pw) * (long)(P * P * C) + (p1 * P + p2) * C + c];
}

// ---- date features (maestro/layers/utils.py:128-167), same fp32 operation order as the reference:
// out[b, row_off + d, :] = fac * [diff x4, sin(2pi doy/365.25), cos, sin(2pi hour/24), cos], diff = (year+doy')-(year_ref+doy_ref')
__global__ __launch_bounds__(256) void date_features_kernel(const int16_t* __restrict__ dates, const int16_t* __restrict__ ref,
                                                            float* __restrict__ out, int B, int D, int rows, int row_off, float fac) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * D) return;
    const int b = i / D, d = i - b * D;
    const int16_t* dt = dates + (size_t)i * 3;
    const float year = (float)dt[0], doy = (float)dt[1] / 365.25f, hour = (float)dt[2] / 24.0f;
    const float year_ref = (float)ref[b * 3], doy_ref = (float)ref[b * 3 + 1] / 365.25f;
    const float diff = (year + doy) - (year_ref + doy_ref);
    const float a = 6.283185307179586f * doy, h = 6.283185307179586f * hour;
    float* o = out + ((size_t)b * rows + row_off + d) * 8;
    o[0] = o[1] = o[2] = o[3] = diff * fac;
    o[4] = sinf(a) * fac; o[5] = cosf(a) * fac; o[6] = sinf(h) * fac; o[7] = cosf(h) * fac;
}

// ---- out = img with channels >= 1 replaced by 30 * (ch0 - ch)   (maestro/ssl/mim.py:433-436)
__global__ __launch_bounds__(256) void rescale_elev_kernel(const float* __restrict__ img, float* __restrict__ out, int C, long plane, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long px = i % plane; const long r = i / plane; const int c = r % C; const long bd = r / C;
    const float v = img[i];
    out[i] = c == 0 ? v : 30.f * (img[(bd * C) * plane + px] - v);
}

// ---- raster resize to image_size (maestro/ssl/mim.py:427-432 -> F.interpolate, align_corners=False): PyTorch's index maps
// mode 0 nearest: src = min(floor(dst * in/out), in-1); mode 1 bilinear: src = max((dst+0.5)*in/out - 0.5, 0);
// mode 2 bicubic: src = (dst+0.5)*in/out - 0.5 (not clamped), 4x4 taps at floor(src)-1..+2 clamped to the image, cubic
// convolution weights with A = -0.75 (PyTorch's upsample_bicubic2d)
__device__ __forceinline__ void cubic_weights(float t, float (&w)[4]) {
    const float A = -0.75f;
    const float x0 = t + 1.f, x3 = 2.f - t, x2 = 1.f - t;
    w[0] = ((A * x0 - 5.f * A) * x0 + 8.f * A) * x0 - 4.f * A;
    w[1] = ((A + 2.f) * t - (A + 3.f)) * t * t + 1.f;
    w[2] = ((A + 2.f) * x2 - (A + 3.f)) * x2 * x2 + 1.f;
    w[3] = ((A * x3 - 5.f * A) * x3 + 8.f * A) * x3 - 4.f * A;
}
__global__ __launch_bounds__(256) void resize_kernel(const float* __restrict__ in, float* __restrict__ out, int Hin, int Win,
                                                     int Hout, int Wout, int mode, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int x = i % Wout; const long r = i / Wout; const int y = r % Hout; const long plane = r / Hout;
    const float* src = in + plane * (long)Hin * Win;
    const float sy = (float)Hin / (float)Hout, sx = (float)Win / (float)Wout;
    if (mode == 0) {
        const int yy = min((int)floorf(y * sy), Hin - 1), xx = min((int)floorf(x * sx), Win - 1);
        out[i] = src[(long)yy * Win + xx];
        return;
    }
    if (mode == 2) {
        const float ry = sy * (y + 0.5f) - 0.5f, rx = sx * (x + 0.5f) - 0.5f;
        const int iy = (int)floorf(ry), ix = (int)floorf(rx);
        float wy[4], wx[4];
        cubic_weights(ry - iy, wy);
        cubic_weights(rx - ix, wx);
        float acc = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int yy = min(max(iy - 1 + a, 0), Hin - 1);
            float row = 0.f;
#pragma unroll
            for (int b = 0; b < 4; ++b) row += wx[b] * src[(long)yy * Win + min(max(ix - 1 + b, 0), Win - 1)];
            acc += wy[a] * row;
        }
        out[i] = acc;
        return;
    }
    const float fy = fmaxf(sy * (y + 0.5f) - 0.5f, 0.f), fx = fmaxf(sx * (x + 0.5f) - 0.5f, 0.f);
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < Hin - 1 ? 1 : 0), x1 = x0 + (x0 < Win - 1 ? 1 : 0);
    const float ly = fy - y0, lx = fx - x0, hy = 1.f - ly, hx = 1.f - lx;
    out[i] = hy * (hx * src[(long)y0 * Win + x0] + lx * src[(long)y0 * Win + x1]) +
             ly * (hx * src[(long)y1 * Win + x0] + lx * src[(long)y1 * Win + x1]);
}

// ---- per-sample dihedral augmentation of square rasters (maestro/dataset/dataset.py:224-257: flip axis 2, flip axis 3,
// swap axes 2 and 3 of the per-sample [D, C, H, W] array, in that order).  flags[b]: bit0 = flip rows, bit1 = flip columns,
// bit2 = transpose.  out[y, x] = in[yy, xx] with (yy, xx) = transpose ? (x, y) : (y, x), then xx -> S-1-xx if bit1,
// yy -> S-1-yy if bit0.  32x32 tiles through LDS so that the transposed case still reads and writes whole row segments.
template <typename T>
__global__ __launch_bounds__(256) void dihedral_kernel(const T* __restrict__ in, T* __restrict__ out,
                                                       const uint8_t* __restrict__ flags, int S, long planes) {
    __shared__ T tile[32][33];
    const int b = blockIdx.z, f = flags[b], tiles = (S + 31) / 32;
    const bool fh = f & 1, fw = f & 2, tr = f & 4;
    const int ty = blockIdx.x / tiles, tx = blockIdx.x % tiles;
    const size_t base = ((size_t)b * planes + blockIdx.y) * S * S;
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    if (!tr) {
        for (int j = ly; j < 32; j += 8) {
            const int y = ty * 32 + j, x = tx * 32 + lx;
            if (y < S && x < S) out[base + (size_t)y * S + x] = in[base + (size_t)(fh ? S - 1 - y : y) * S + (fw ? S - 1 - x : x)];
        }
        return;
    }
    for (int j = ly; j < 32; j += 8) {
        const int dx = tx * 32 + j, dy = ty * 32 + lx;      // destination column / row served by this source element
        if (dx < S && dy < S) tile[j][lx] = in[base + (size_t)(fh ? S - 1 - dx : dx) * S + (fw ? S - 1 - dy : dy)];
    }
    __syncthreads();
    for (int j = ly; j < 32; j += 8) {
        const int y = ty * 32 + j, x = tx * 32 + lx;
        if (y < S && x < S) out[base + (size_t)y * S + x] = tile[lx][j];
    }
}

}  // namespace

extern "C" int mh_resize(const float* in, float* out, long planes, int Hin, int Win, int Hout, int Wout, int mode, void* stream) {
    MH_CHECK_ARG(in && out && in != out && planes > 0 && Hin > 0 && Win > 0 && Hout > 0 && Wout > 0, "mh_resize: bad arguments");
    MH_CHECK_ARG(mode >= 0 && mode <= 2, "mh_resize: mode %d (0 nearest, 1 bilinear, 2 bicubic)", mode);
    const long total = planes * Hout * Wout;
    hipLaunchKernelGGL(resize_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, in, out, Hin, Win, Hout, Wout,
                       mode, total);
    MH_LAUNCH_CHECK();
    return 0;
}

extern "C" int mh_dihedral(const void* in, void* out, const uint8_t* flags, int B, long planes, int S, int elem_bytes,
                           void* stream) {
    MH_CHECK_ARG(in && out && in != out && flags && B > 0 && B < 65536 && planes > 0 && planes < 65536 && S > 0,
                 "mh_dihedral: bad arguments (out of place, B and planes per sample < 65536)");
    const int tiles = ceil_div(S, 32);
    dim3 grid(tiles * tiles, (unsigned)planes, B), block(256);
    hipStream_t s = (hipStream_t)stream;
    switch (elem_bytes) {
        case 1: hipLaunchKernelGGL(dihedral_kernel<uint8_t>, grid, block, 0, s, (const uint8_t*)in, (uint8_t*)out, flags, S, planes); break;
        case 2: hipLaunchKernelGGL(dihedral_kernel<uint16_t>, grid, block, 0, s, (const uint16_t*)in, (uint16_t*)out, flags, S, planes); break;
        case 4: hipLaunchKernelGGL(dihedral_kernel<uint32_t>, grid, block, 0, s, (const uint32_t*)in, (uint32_t*)out, flags, S, planes); break;
        case 8: hipLaunchKernelGGL(dihedral_kernel<uint64_t>, grid, block, 0, s, (const uint64_t*)in, (uint64_t*)out, flags, S, planes); break;
        default: return mh_fail(-1, "mh_dihedral: element size %d (1, 2, 4 or 8 bytes)", elem_bytes);
    }
    MH_LAUNCH_CHECK();
    return 0;
}

extern "C" int mh_date_features(const int16_t* dates, const int16_t* ref_date, float* out, int B, int D, int rows, int row_off,
                                float fac, void* stream) {
    MH_CHECK_ARG(dates && ref_date && out && row_off + D <= rows, "mh_date_features: bad arguments");
    hipLaunchKernelGGL(date_features_kernel, dim3(ceil_div(B * D, 256)), dim3(256), 0, (hipStream_t)stream, dates, ref_date, out, B, D,
                       rows, row_off, fac);
    MH_LAUNCH_CHECK();
    return 0;
}

extern "C" int mh_rescale_elev(const float* img, float* out, int BD, int C, int S, void* stream) {
    MH_CHECK_ARG(img && out && img != out, "mh_rescale_elev: bad arguments (out of place only)");
    const long plane = (long)S * S, total = (long)BD * C * plane;
    hipLaunchKernelGGL(rescale_elev_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, img, out, C, plane, total);
    MH_LAUNCH_CHECK();
    return 0;
}

extern "C" int mh_patchify(const float* img, void* cols, float* target, int BD, int Ctot, int S, int P, int Kpad,
                           const int* norm_bands, int n_norm_groups, int normalise, int rescale_elev, void* stream) {
    MH_CHECK_ARG(cols, "mh_patchify: null pointer");
    return mh_patchify_bands(img, cols, target, BD, Ctot, 0, Ctot, S, P, Kpad, norm_bands, n_norm_groups, normalise, rescale_elev,
                             stream);
}

extern "C" int mh_patchify_bands(const float* img, void* cols, float* target, int BD, int Csrc, int c0, int Ctot, int S, int P,
                                 int Kpad, const int* norm_bands, int n_norm_groups, int normalise, int rescale_elev,
                                 void* stream) {
    MH_CHECK_ARG(img && (cols || target), "mh_patchify: null pointer");
    MH_CHECK_ARG(c0 >= 0 && Ctot > 0 && c0 + Ctot <= Csrc, "mh_patchify_bands: band window [%d, %d) outside %d channels", c0, c0 + Ctot, Csrc);
    MH_CHECK_ARG(S % P == 0 && Kpad >= Ctot * P * P && Kpad % 8 == 0, "mh_patchify: bad geometry S=%d P=%d Kpad=%d", S, P, Kpad);
    MH_CHECK_ARG(!normalise || !target || (norm_bands && n_norm_groups > 0), "mh_patchify: norm_bands missing");
    const int K = Ctot * P * P, g = S / P;
    if (P == 16 && Ctot <= 4 && S % 4 == 0 && ((uintptr_t)img % 16) == 0 && (!cols || (uintptr_t)cols % 8 == 0) &&
        (!target || (uintptr_t)target % 16 == 0)) {   // one wave per patch (see patchify_wave16_kernel)
        const int n_tok = BD * g * g;
        dim3 grid(ceil_div(n_tok, 4)), block(256);
        hipStream_t s = (hipStream_t)stream;
#define PATCHIFY16(CT) hipLaunchKernelGGL(patchify_wave16_kernel<CT>, grid, block, 0, s, img, (bf16_t*)cols, target, Csrc, c0, S, Kpad, \
                                          norm_bands, n_norm_groups, normalise, rescale_elev, n_tok)
        switch (Ctot) { case 1: PATCHIFY16(1); break; case 2: PATCHIFY16(2); break; case 3: PATCHIFY16(3); break; default: PATCHIFY16(4); }
#undef PATCHIFY16
        MH_LAUNCH_CHECK();
        return 0;
    }
    const int threads = K <= 128 ? 64 : 256;
    const size_t lds = (size_t)(K + 8) * sizeof(float);
    MH_CHECK_ARG(lds <= 64 * 1024, "mh_patchify: patch too large for LDS (%d floats)", K);
    hipLaunchKernelGGL(patchify_kernel, dim3(BD * g * g), dim3(threads), lds, (hipStream_t)stream, img, (bf16_t*)cols,
                       target, Csrc, c0, Ctot, S, P, Kpad, norm_bands, n_norm_groups, normalise, rescale_elev);
    MH_LAUNCH_CHECK();
    return 0;
}

extern "C" int mh_groupnorm_stats(const float* y, float* partial, float* stats, int BD, int L, int E, float eps,
                                  void* stream) {
    MH_CHECK_ARG(y && partial && stats && E % 4 == 0, "mh_groupnorm_stats: bad arguments");
    const long per_image = (long)L * E;
    const int nchunk = ceil_div(per_image, GN_CHUNK);
    hipLaunchKernelGGL(gn_partial_kernel, dim3(nchunk, BD), dim3(256), 0, (hipStream_t)stream, y, partial, per_image, nchunk);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(BD), dim3(64), 0, (hipStream_t)stream, partial, stats, per_image, nchunk, eps);
    MH_LAUNCH_CHECK();
    return 0;
}

extern "C" int mh_groupnorm_partial_size(int BD, int L, int E) { return BD * ceil_div((long)L * E, GN_CHUNK) * 2; }

extern "C" int mh_embed_finish(const float* y, const float* stats, const float* gamma, const float* beta,
                               const float* pos, const float* date, int date_rows, int date_off, float* xg, int B, int D,
                               int L, int E, int tok_off, int Lgroup, void* stream) {
    MH_CHECK_ARG(y && stats && gamma && beta && pos && xg && E % 4 == 0 && E >= 8, "mh_embed_finish: bad arguments");
    MH_CHECK_ARG(tok_off + D * L <= Lgroup, "mh_embed_finish: modality does not fit its group sequence");
    hipLaunchKernelGGL(embed_finish_kernel, dim3(ceil_div((long)B * D * L, 4)), dim3(256), 0, (hipStream_t)stream, y, stats,
                       gamma, beta, pos, date, xg, B, D, L, E, tok_off, Lgroup, date_rows, date_off);
    MH_LAUNCH_CHECK();
    return 0;
}

extern "C" int mh_embed_finish_bwd(const float* dxg, const float* y, const float* stats, const float* gamma, void* dyc,
                                   float* dgamma, float* dbeta, float* sums, int B, int D, int L, int E, int tok_off,
                                   int Lgroup, void* stream) {
    MH_CHECK_ARG(dxg && y && stats && gamma && dyc && dgamma && dbeta && sums && E % 4 == 0, "mh_embed_finish_bwd: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    // (a KERNEL, not hipMemsetAsync: this entry point is captured into hipGraphs, and on ROCm 7.2 the 16-byte memset node of a
    //  [B = 2, D = 1] modality replayed wrongly -- garbage in `sums` from the second replay on, found in round 3 by the order of
    //  the GPU tests; every other node of the graph was fine)
    hipLaunchKernelGGL(zero_f32_kernel, dim3(ceil_div((long)B * D * 2, 256)), dim3(256), 0, s, sums, B * D * 2);
    {
        const dim3 grid(ceil_div(L, 4 * EB_ROWS), B * D), block(256);
        const size_t lds = (size_t)8 * E * sizeof(float);
#define MH_EB_LAUNCH(NV) hipLaunchKernelGGL(embed_bwd_stats_kernel<NV>, grid, block, lds, s, dxg, y, stats, gamma, sums, dgamma, dbeta, B, D, L, E, tok_off, Lgroup)
        switch ((E + 255) / 256) {
            case 1: MH_EB_LAUNCH(1); break;
            case 2: MH_EB_LAUNCH(2); break;
            case 3: MH_EB_LAUNCH(3); break;
            case 4: MH_EB_LAUNCH(4); break;
            default: MH_EB_LAUNCH(0); break;
        }
#undef MH_EB_LAUNCH
    }
    hipLaunchKernelGGL(embed_bwd_apply_kernel, dim3(ceil_div((long)B * D * L, 4)), dim3(256), 0, s, dxg, y, stats, gamma, sums,
                       (bf16_t*)dyc, B, D, L, E, tok_off, Lgroup);
    MH_LAUNCH_CHECK();
    return 0;
}

extern "C" int mh_depatchify(const float* patches, float* img, int BD, int C, int S, int P, void* stream) {
    MH_CHECK_ARG(patches && img && S % P == 0, "mh_depatchify: bad arguments");
    const long total = (long)BD * C * S * S;
    hipLaunchKernelGGL(depatchify_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, patches, img, C, S, P, total);
    MH_LAUNCH_CHECK();
    return 0;
}
