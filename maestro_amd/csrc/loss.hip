// Masked reconstruction loss at patch layout + its gradient, bias-gradient column sums, casts, fused AdamW.
#include "gemm_common.hpp"

namespace {

// rec/target f32 [T, PPC] (token rows of ONE modality: T = B * Lm); mask_group u8 [B, Lgroup]; token (b, t) of the
// modality sits at group position tok_off + t.  acc[0] += sum(e over masked), loss partial; drec (bf16) =
// coef * de/drec for masked tokens, 0 elsewhere, with coef = weight / (n_masked_tokens * PPC).
// One WAVE per token (4 tokens per block), 16-byte accesses; the block reduces its partial loss through LDS and issues one
// atomic (32768 -> 8192 atomics for the aerial modality; blocks whose 4 tokens are all visible issue none).
__global__ __launch_bounds__(256) void masked_loss_kernel(const float* __restrict__ rec, const float* __restrict__ target,
                                                          const uint8_t* __restrict__ mask_group, const int* __restrict__ n_masked,
                                                          float weight, float* __restrict__ acc, bf16_t* __restrict__ drec,
                                                          int B, int Lm, int Lgroup, int tok_off, int PPC, int p, int tgt_C,
                                                          int tgt_c0, int n_g, int denom_is_elems, int rows_per_wave) {
    // band window (several band-groups per modality): rec column k = pixel * n_g + c pairs with target column
    // pixel * tgt_C + tgt_c0 + c of the MODALITY's target rows (tgt_C channels); n_g == tgt_C: the plain case
    __shared__ float red[4];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float coef = weight / (denom_is_elems ? (float)(*n_masked) : (float)(*n_masked) * (float)PPC);
    const bool window = n_g != tgt_C;
    const int tgt_ld = window ? PPC / n_g * tgt_C : PPC;
    float s = 0.f;
    // several rows per wave: the block's ONE atomic into the loss word costs ~10 ns of serialised time (same-line atomics,
    // scripts/micro_amax.hip) -- 8192 blocks of 4 rows made the aerial modality's launch 80 us of atomics
    for (int rr = 0; rr < rows_per_wave; ++rr) {
        const int row = (blockIdx.x * 4 + w) * rows_per_wave + rr;
        if (row >= B * Lm) break;
        const int b = row / Lm, t = row - b * Lm;
        const bool masked = mask_group[(size_t)b * Lgroup + tok_off + t] != 0;
        const float* r = rec + (size_t)row * PPC;
        const float* g = target + (size_t)row * tgt_ld;
        bf16_t* d = drec ? drec + (size_t)row * PPC : nullptr;
        for (int c = lane * 4; c < PPC; c += 256) {
            float dd[4] = {0.f, 0.f, 0.f, 0.f};
            if (masked) {
                const f32x4 rv = *reinterpret_cast<const f32x4*>(r + c);
                f32x4 tv;
                if (!window) {
                    tv = *reinterpret_cast<const f32x4*>(g + c);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const int pix = (c + e) / n_g; tv[e] = g[pix * tgt_C + tgt_c0 + (c + e) - pix * n_g]; }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float diff = rv[e] - tv[e];
                    if (p == 2) { s += diff * diff; dd[e] = 2.f * diff * coef; }
                    else { s += fabsf(diff); dd[e] = (diff > 0.f ? coef : (diff < 0.f ? -coef : 0.f)); }
                }
            }
            if (d) {
                u32x2 pk = {pack_bf2(dd[0], dd[1]), pack_bf2(dd[2], dd[3])};
                *reinterpret_cast<u32x2*>(d + c) = pk;
            }
        }
    }
    s = wave_sum(s);
    if (lane == 0) red[w] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float t = (red[0] + red[1]) + (red[2] + red[3]);
        if (t != 0.f) atomicAdd(acc, t * coef);  // coef = weight / n_elems -> acc accumulates the weighted loss
        // a modality without a single masked token in the batch: the reference takes the mean of an empty selection
        // (maestro/train/model.py:241-243) -> NaN loss, zero gradient for this modality; reproduced, not guarded
        if (blockIdx.x == 0 && *n_masked == 0) atomicAdd(acc, __builtin_nanf(""));
    }
}

// 1024 threads: 16 waves walk the block's rows side by side, 64 lanes x 4 columns each, eight row loads in flight per lane.  The
// launch is bound by two things that pull in opposite directions -- memory-level parallelism (wants many resident waves) and the
// same-address atomics of the final add (M / rows_per_block per column: wants few, fat blocks) -- so a block is 16 waves on up to
// 1024 rows: 32768 x 768 bf16 31.7 -> see scripts/bench_small_reductions.py.
__global__ __launch_bounds__(1024) void colsum_kernel(const void* __restrict__ x, int is_f32, float* __restrict__ out, int M,
                                                      int N, int ld, int rows_per_block) {
    __shared__ f32x4 red[16][64];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    const int c = (blockIdx.x * 64 + lane) * 4;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
    f32x4 acc = {0, 0, 0, 0};
    if (c < N) {
        // independent row loads, eight in flight per lane.  The element type is tested OUTSIDE the loops: with the test inside,
        // the two cases met in one accumulator after every load and the ISA waited `vmcnt(0)` per row (one round trip each).
        if (is_f32) {
#pragma unroll 8
            for (int r = r0 + w; r < r1; r += nw)
                acc += *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(x) + (size_t)r * ld + c);
        } else {
#pragma unroll 8
            for (int r = r0 + w; r < r1; r += nw) {
                const u32x2 pk = *reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16_t*>(x) + (size_t)r * ld + c);
                acc += (f32x4){__uint_as_float(pk[0] << 16), __uint_as_float(pk[0] & 0xffff0000u),
                               __uint_as_float(pk[1] << 16), __uint_as_float(pk[1] & 0xffff0000u)};
            }
        }
    }
    red[w][lane] = acc;
    __syncthreads();
    if (w == 0 && c < N) {
        f32x4 t = red[0][lane];
        for (int i = 1; i < nw; ++i) t += red[i][lane];
#pragma unroll
        for (int e = 0; e < 4; ++e) atomicAdd(out + c + e, t[e]);
    }
}

// zero a list of (offset, length) float spans of one buffer: blockIdx.y = span, blockIdx.x walks it in 1024-float pieces
__global__ __launch_bounds__(256) void zero_spans_kernel(float* __restrict__ base, const long* __restrict__ spans) {
    const long off = spans[2 * blockIdx.y], len = spans[2 * blockIdx.y + 1];
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < len; i += (long)gridDim.x * 1024) {
        float* p = base + off + i;
        if (i + 3 < len && (reinterpret_cast<uintptr_t>(p) & 15) == 0) *reinterpret_cast<f32x4*>(p) = (f32x4){0, 0, 0, 0};
        else for (long e = i; e < min(len, i + 4); ++e) base[off + e] = 0.f;
    }
}

__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long n) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + i);
        u32x2 pk = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
        *reinterpret_cast<u32x2*>(dst + i) = pk;
    } else {
        for (long j = i; j < n; ++j) dst[j] = f2bf(src[j]);
    }
}

__global__ __launch_bounds__(256) void pack_rows_kernel(const float* __restrict__ w, bf16_t* __restrict__ dst, int E, int K, int Kpad) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)E * Kpad) return;
    const int e = i / Kpad, k = i - (long)e * Kpad;
    dst[i] = k < K ? f2bf(w[(size_t)e * K + k]) : (bf16_t)0;
}

__global__ __launch_bounds__(256) void unpack_rows_add_kernel(const float* __restrict__ src, float* __restrict__ dst, int E, int K, int Kpad) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)E * K) return;
    const int e = i / K, k = i - (long)e * K;
    atomicAdd(dst + i, src[(size_t)e * Kpad + k]);  // shared patch-embed weights may be updated from parallel streams
}

// torch.optim.AdamW step (decoupled weight decay, bias correction), 4 elements per thread, grid-stride free.
// fp8 shadows (C5 path): p8 = a flat uint8 buffer with the parameters' offsets; slot_map[i / 64] = scale slot of the weight
// element i belongs to (-1: not an fp8 GEMM operand).  The cast uses the scale derived from the PREVIOUS step's absmax and
// records this step's (delayed scaling: a weight moves by one learning-rate step at a time).
struct Fp8Shadow { uint8_t* p8; const short* slot_map; const float* scale; float* amax; };

// returns the fp8 scale slot of this lane's four elements (-1: none) and their absmax in *amax_out
__device__ __forceinline__ int adamw_update(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                             float* __restrict__ v, bf16_t* __restrict__ pb, long n, float lr, float b1, float b2,
                                             float eps, float wd, float bc1, float bc2_sqrt, float gscale,
                                             Fp8Shadow f8 = Fp8Shadow{nullptr, nullptr, nullptr, nullptr},
                                             float* amax_out = nullptr) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return -1;
    // Every stream of this kernel is touched once per step (30 bytes per parameter, 5.3 GB on C3): nontemporal loads AND stores keep
    // them from displacing each other in the L2 / Infinity Cache -- 925 -> 858 us on 176 M parameters (5.7 -> 6.2 TB/s); either half
    // alone gains 1 %, two or four quads per lane nothing (scripts/micro_adamw.hip, profiles/r05_experiments.md).
    const f32x4 gv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g + i)) * gscale;
    f32x4 pv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + i));
    f32x4 mv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(m + i));
    f32x4 vv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(v + i));
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        pv[e] *= 1.f - lr * wd;
        mv[e] = b1 * mv[e] + (1.f - b1) * gv[e];
        vv[e] = b2 * vv[e] + (1.f - b2) * gv[e] * gv[e];
        const float denom = sqrtf(vv[e]) / bc2_sqrt + eps;
        pv[e] -= (lr / bc1) * (mv[e] / denom);
    }
    __builtin_nontemporal_store(pv, reinterpret_cast<f32x4*>(p + i));
    __builtin_nontemporal_store(mv, reinterpret_cast<f32x4*>(m + i));
    __builtin_nontemporal_store(vv, reinterpret_cast<f32x4*>(v + i));
    if (pb) {
        u32x2 pk = {pack_bf2(pv[0], pv[1]), pack_bf2(pv[2], pv[3])};
        __builtin_nontemporal_store(pk, reinterpret_cast<u32x2*>(pb + i));
    }
    if (f8.p8) {
        const int slot = f8.slot_map[i >> 6];
        if (slot >= 0) {
            const float s8 = f8.scale[slot];
            float mx = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) mx = amax_fold(mx, pv[e]);     // keeps inf / NaN (common.hpp)
            __builtin_nontemporal_store(pack_e4m3x4(pv, s8), reinterpret_cast<uint32_t*>(f8.p8 + i));
            *amax_out = mx;
            return slot;
        }
    }
    return -1;
}
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, bf16_t* __restrict__ pb, long n, float lr, float b1,
                                                    float b2, float eps, float wd, float bc1, float bc2_sqrt, float gscale) {
    adamw_update(p, g, m, v, pb, n, lr, b1, b2, eps, wd, bc1, bc2_sqrt, gscale);
}
__global__ __launch_bounds__(256) void adamw_fp8_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, bf16_t* __restrict__ pb, long n, float lr, float b1,
                                                        float b2, float eps, float wd, float bc1, float bc2_sqrt, float gscale,
                                                        Fp8Shadow f8) {
    // absmax: 80 M lanes adding to one word per tensor took 22 ms per step -- fold per wave and per workgroup first (a workgroup
    // spans 1024 elements, i.e. one tensor except at the few tensor boundaries), then one add into a sub-slot of the amax row
    __shared__ float red_mx[4];
    __shared__ int red_slot[4];
    float mx = 0.f;
    const int slot = adamw_update(p, g, m, v, pb, n, lr, b1, b2, eps, wd, bc1, bc2_sqrt, gscale, f8, &mx);
    const int w = threadIdx.x >> 6;
    const int slot0 = __builtin_amdgcn_readfirstlane(slot);      // first ACTIVE lane: all lanes are active here
    const bool uniform = __all(slot == slot0);
    if (uniform) {
        mx = wave_amax(mx);
    } else if (slot >= 0 && amax_nonzero(mx)) {
        atomic_max_pos(f8.amax + (size_t)slot * MH_FP8_AMAX_PITCH, mx);   // a wave across a tensor boundary: rare
    }
    if ((threadIdx.x & 63) == 0) { red_mx[w] = uniform ? mx : 0.f; red_slot[w] = uniform ? slot0 : -1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (red_slot[0] == red_slot[1] && red_slot[0] == red_slot[2] && red_slot[0] == red_slot[3]) {
            const float m4 = amax_fold(amax_fold(amax_fold(red_mx[0], red_mx[1]), red_mx[2]), red_mx[3]);
            if (red_slot[0] >= 0 && amax_nonzero(m4)) atomic_max_pos(f8.amax + (size_t)red_slot[0] * MH_FP8_AMAX_PITCH, m4);
        } else {
            for (int k = 0; k < 4; ++k)
                if (red_slot[k] >= 0 && amax_nonzero(red_mx[k])) atomic_max_pos(f8.amax + (size_t)red_slot[k] * MH_FP8_AMAX_PITCH, red_mx[k]);
        }
    }
}
// per-step scalars from device memory: the launch is captured once and replayed with new values (see mh_adamw_dev)
__global__ __launch_bounds__(256) void adamw_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, bf16_t* __restrict__ pb, long n, float b1, float b2,
                                                        float eps, float wd, const float* __restrict__ hyper) {
    if (hyper[4] == 0.f) return;
    adamw_update(p, g, m, v, pb, n, hyper[0], b1, b2, eps, wd, hyper[1], hyper[2], hyper[3]);
}

// x *= *scale, skipped when the device scalar is exactly 1 (loss.backward() with autograd's default ones gradient): the
// Lightning bridge scales the flat gradient buffer by d loss without reading the scalar on the host.
__global__ __launch_bounds__(256) void scale_dev_kernel(float* __restrict__ x, long n, const float* __restrict__ scale) {
    const float s = *scale;
    if (s == 1.f) return;
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 4 <= n) {
        *reinterpret_cast<f32x4*>(x + i) = *reinterpret_cast<const f32x4*>(x + i) * s;
    } else {
        for (long j = i; j < n; ++j) x[j] *= s;
    }
}

}  // namespace

// rows per wave of masked_loss_kernel: about 1024 workgroups (= same-word atomics) per launch, at most 16 rows per wave
static int loss_rows_per_wave(long rows) { return (int)std::min(16L, std::max(1L, rows / 4096)); }

extern "C" int mh_scale_dev(float* x, long n, const float* scale, void* stream) {
    MH_CHECK_ARG(x && scale && n > 0 && ((uintptr_t)x % 16) == 0, "mh_scale_dev: bad arguments (x 16-byte aligned)");
    hipLaunchKernelGGL(scale_dev_kernel, dim3(ceil_div(n, 1024)), dim3(256), 0, (hipStream_t)stream, x, n, scale);
    MH_LAUNCH_CHECK();
    return 0;
}

extern "C" int mh_masked_loss(const float* rec, const float* target, const uint8_t* mask_group, const int* n_masked,
                              float weight, float* acc, void* drec, int B, int Lm, int Lgroup, int tok_off, int PPC, int p,
                              void* stream) {
    MH_CHECK_ARG(rec && target && mask_group && n_masked && acc, "mh_masked_loss: null pointer");
    MH_CHECK_ARG((p == 1 || p == 2) && PPC % 4 == 0 && tok_off + Lm <= Lgroup, "mh_masked_loss: bad arguments");
    const int rpw = loss_rows_per_wave((long)B * Lm);
    hipLaunchKernelGGL(masked_loss_kernel, dim3(ceil_div((long)B * Lm, 4 * rpw)), dim3(256), 0, (hipStream_t)stream, rec, target, mask_group, n_masked,
                       weight, acc, (bf16_t*)drec, B, Lm, Lgroup, tok_off, PPC, p, 1, 0, 1, 0, rpw);
    MH_LAUNCH_CHECK();
    return 0;
}

extern "C" int mh_masked_loss_bands(const float* rec, const float* target, const uint8_t* mask_group, const int* n_elems,
                                    float weight, float* acc, void* drec, int B, int Lm, int Lgroup, int tok_off, int PPC, int p,
                                    int tgt_C, int tgt_c0, int n_g, void* stream) {
    MH_CHECK_ARG(rec && target && mask_group && n_elems && acc, "mh_masked_loss_bands: null pointer");
    MH_CHECK_ARG((p == 1 || p == 2) && PPC % 4 == 0 && tok_off + Lm <= Lgroup, "mh_masked_loss_bands: bad arguments");
    MH_CHECK_ARG(n_g > 0 && PPC % n_g == 0 && tgt_c0 >= 0 && tgt_c0 + n_g <= tgt_C, "mh_masked_loss_bands: band window [%d, %d) of %d",
                 tgt_c0, tgt_c0 + n_g, tgt_C);
    const int rpw = loss_rows_per_wave((long)B * Lm);
    hipLaunchKernelGGL(masked_loss_kernel, dim3(ceil_div((long)B * Lm, 4 * rpw)), dim3(256), 0, (hipStream_t)stream, rec, target, mask_group, n_elems,
                       weight, acc, (bf16_t*)drec, B, Lm, Lgroup, tok_off, PPC, p, tgt_C, tgt_c0, n_g, 1, rpw);
    MH_LAUNCH_CHECK();
    return 0;
}

extern "C" int mh_colsum(const void* x, int x_is_f32, float* out, int M, int N, int ld, void* stream) {
    MH_CHECK_ARG(x && out && N % 4 == 0 && ld % 4 == 0, "mh_colsum: bad arguments");
    // up to 1024 rows per 16-wave block; fewer when that would leave most CUs without a block
    const int col_blocks = ceil_div(N, 256);
    int rows_per_block = 1024;
    while (rows_per_block > 128 && (long)col_blocks * ceil_div(M, rows_per_block) < 128) rows_per_block >>= 1;
    hipLaunchKernelGGL(colsum_kernel, dim3(col_blocks, ceil_div(M, rows_per_block)), dim3(1024), 0, (hipStream_t)stream, x,
                       x_is_f32, out, M, N, ld, rows_per_block);
    MH_LAUNCH_CHECK();
    return 0;
}

extern "C" int mh_zero_spans(float* base, const long* spans_device, int n_spans, long max_len, void* stream) {
    MH_CHECK_ARG(base && spans_device && n_spans > 0 && max_len > 0, "mh_zero_spans: bad arguments");
    dim3 grid((unsigned)min(64L, (max_len + 1023) / 1024), n_spans);
    hipLaunchKernelGGL(zero_spans_kernel, grid, dim3(256), 0, (hipStream_t)stream, base, spans_device);
    MH_LAUNCH_CHECK();
    return 0;
}

extern "C" int mh_cast_bf16(const float* src, void* dst, long n, void* stream) {
    MH_CHECK_ARG(src && dst && n > 0, "mh_cast_bf16: bad arguments");
    hipLaunchKernelGGL(cast_bf16_kernel, dim3(ceil_div(n, 1024)), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, n);
    MH_LAUNCH_CHECK();
    return 0;
}

extern "C" int mh_pack_rows_bf16(const float* w, void* dst, int E, int K, int Kpad, void* stream) {
    MH_CHECK_ARG(w && dst && Kpad >= K, "mh_pack_rows_bf16: bad arguments");
    hipLaunchKernelGGL(pack_rows_kernel, dim3(ceil_div((long)E * Kpad, 256)), dim3(256), 0, (hipStream_t)stream, w, (bf16_t*)dst, E, K, Kpad);
    MH_LAUNCH_CHECK();
    return 0;
}

extern "C" int mh_unpack_rows_add(const float* src, float* dst, int E, int K, int Kpad, void* stream) {
    MH_CHECK_ARG(src && dst && Kpad >= K, "mh_unpack_rows_add: bad arguments");
    hipLaunchKernelGGL(unpack_rows_add_kernel, dim3(ceil_div((long)E * K, 256)), dim3(256), 0, (hipStream_t)stream, src, dst, E, K, Kpad);
    MH_LAUNCH_CHECK();
    return 0;
}

extern "C" int mh_adamw(float* p, const float* g, float* m, float* v, void* p_bf16, long n, float lr, float b1, float b2,
                        float eps, float wd, int step, float grad_scale, void* stream) {
    MH_CHECK_ARG(p && g && m && v && n > 0 && n % 4 == 0 && step >= 1, "mh_adamw: bad arguments (n %% 4 == 0, step >= 1)");
    const float bc1 = 1.f - powf(b1, (float)step);
    const float bc2_sqrt = sqrtf(1.f - powf(b2, (float)step));
    hipLaunchKernelGGL(adamw_kernel, dim3(ceil_div(n, 1024)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (bf16_t*)p_bf16, n, lr,
                       b1, b2, eps, wd, bc1, bc2_sqrt, grad_scale);
    MH_LAUNCH_CHECK();
    return 0;
}

extern "C" int mh_adamw_fp8(float* p, const float* g, float* m, float* v, void* p_bf16, void* p_fp8, const short* slot_map,
                            const float* scale, float* amax, long n, float lr, float b1, float b2, float eps, float wd, int step,
                            float grad_scale, void* stream) {
    MH_CHECK_ARG(p && g && m && v && n > 0 && n % 4 == 0 && step >= 1, "mh_adamw_fp8: bad arguments (n %% 4 == 0, step >= 1)");
    MH_CHECK_ARG(p_fp8 && slot_map && scale && amax && ((uintptr_t)p_fp8 % 4) == 0, "mh_adamw_fp8: fp8 shadow, slot map, scale and amax tables are required");
    const float bc1 = 1.f - powf(b1, (float)step);
    const float bc2_sqrt = sqrtf(1.f - powf(b2, (float)step));
    hipLaunchKernelGGL(adamw_fp8_kernel, dim3(ceil_div(n, 1024)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (bf16_t*)p_bf16, n, lr,
                       b1, b2, eps, wd, bc1, bc2_sqrt, grad_scale, Fp8Shadow{(uint8_t*)p_fp8, slot_map, scale, amax});
    MH_LAUNCH_CHECK();
    return 0;
}

extern "C" int mh_adamw_dev(float* p, const float* g, float* m, float* v, void* p_bf16, long n, float b1, float b2, float eps,
                            float wd, const float* hyper, void* stream) {
    MH_CHECK_ARG(p && g && m && v && hyper && n > 0 && n % 4 == 0, "mh_adamw_dev: bad arguments (n %% 4 == 0)");
    hipLaunchKernelGGL(adamw_dev_kernel, dim3(ceil_div(n, 1024)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (bf16_t*)p_bf16,
                       n, b1, b2, eps, wd, hyper);
    MH_LAUNCH_CHECK();
    return 0;
}
