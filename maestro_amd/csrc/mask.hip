// Token masking: stable rank selection, visible-token gather / scatter, decoder-input assembly (+ backward).
// Integer/index work is bit-exact by construction (counting rank = stable argsort).
#include "common.hpp"
#include "../../include/maestro_hip.h"

namespace {

// One 1024-thread block per sample (the O(L^2) rank is the start of the forward's critical path: 16 waves, not 4).
// rank_i = #{j : n_j < n_i} + #{j < i : n_j == n_i}  (stable ascending order); masked <=> rank < k.  Positions inside the visible /
// masked lists are exclusive prefix counts (ascending index).  The rank loop reads the row four keys at a time (LDS broadcast
// reads of 16 bytes); the prefix counts are wave ballots + popcounts (round 2's serial count over j < i made the launch 91 us
// at L = 1024 -- two thirds of it in that loop).
__global__ __launch_bounds__(1024) void mask_select_kernel(const float* __restrict__ noise, const uint8_t* __restrict__ smask,
                                                          int* __restrict__ visible_idx, int* __restrict__ masked_idx,
                                                          int* __restrict__ inv, uint8_t* __restrict__ mask, int L, int k) {
    extern __shared__ __attribute__((aligned(16))) float sm[];  // noise[L4] (padded with +inf to a multiple of 4), then 16 wave totals
    const int L4 = (L + 3) & ~3;
    int* wtot = reinterpret_cast<int*>(sm + L4);
    const int b = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < L4; i += 1024) {
        float v = INFINITY;
        if (i < L) {
            v = noise[(size_t)b * L + i];
            if (smask && smask[(size_t)b * L + i]) v = 0.f;  // noise *= 1 - struct  (mae.py:240)
        }
        sm[i] = v;
    }
    __syncthreads();
    int base = 0;                                  // masked positions before this chunk of 1024 consecutive indices
    for (int i0 = 0; i0 < L; i0 += 1024) {
        const int i = i0 + threadIdx.x;
        int f = 0;
        if (i < L) {
            const float v = sm[i];
            int r = 0;
            for (int j = 0; j < L4; j += 4) {
                const f32x4 u = *reinterpret_cast<const f32x4*>(sm + j);
                r += (int)((u[0] < v) | ((u[0] == v) & (j < i)));
                r += (int)((u[1] < v) | ((u[1] == v) & (j + 1 < i)));
                r += (int)((u[2] < v) | ((u[2] == v) & (j + 2 < i)));
                r += (int)((u[3] < v) | ((u[3] == v) & (j + 3 < i)));
            }
            f = r < k;
        }
        const unsigned long long bal = __ballot(f);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        __syncthreads();                           // (the previous chunk's readers of wtot are done)
        if (lane == 0) wtot[w] = __popcll(bal);
        __syncthreads();
        int nm = base + before, tot = 0;
#pragma unroll
        for (int x = 0; x < 16; ++x) {
            const int c = wtot[x];
            nm += x < w ? c : 0;
            tot += c;
        }
        base += tot;
        if (i < L) {
            mask[(size_t)b * L + i] = (uint8_t)f;
            if (f) {
                masked_idx[(size_t)b * k + nm] = i;
                inv[(size_t)b * L + i] = -1;
            } else {
                visible_idx[(size_t)b * (L - k) + (i - nm)] = i;
                inv[(size_t)b * L + i] = i - nm;
            }
        }
    }
}

// dst[b, dst_off + j, :] = src[b, idx[b, j], :]   (one wave per row)
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, const int* __restrict__ idx,
                                                          float* __restrict__ dst, int B, int src_L, int n_idx, int dim,
                                                          int dst_L, int dst_off) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= B * n_idx) return;
    const int b = row / n_idx, j = row - b * n_idx;
    const float* s = src + ((size_t)b * src_L + idx[row]) * dim;
    float* d = dst + ((size_t)b * dst_L + dst_off + j) * dim;
    for (int c = lane * 4; c < dim; c += 256) *reinterpret_cast<f32x4*>(d + c) = *reinterpret_cast<const f32x4*>(s + c);
}

// dsrc[b, idx[b, j], :] = ddst[b, dst_off + j, :]  (dsrc pre-zeroed; indices are unique per sample)
__global__ __launch_bounds__(256) void scatter_rows_kernel(const float* __restrict__ ddst, const int* __restrict__ idx,
                                                           float* __restrict__ dsrc, int B, int src_L, int n_idx, int dim,
                                                           int dst_L, int dst_off) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= B * n_idx) return;
    const int b = row / n_idx, j = row - b * n_idx;
    float* s = dsrc + ((size_t)b * src_L + idx[row]) * dim;
    const float* d = ddst + ((size_t)b * dst_L + dst_off + j) * dim;
    for (int c = lane * 4; c < dim; c += 256) *reinterpret_cast<f32x4*>(s + c) = *reinterpret_cast<const f32x4*>(d + c);
}

// dst[b, t, :] = inv[b, t] >= 0 ? src[b, inv[b, t], :] : 0 -- the scatter of the visible rows' gradient back to the full
// sequence written as a gather over EVERY destination row, so the destination needs no memset in front of it
__global__ __launch_bounds__(256) void expand_rows_kernel(const float* __restrict__ src, const int* __restrict__ inv,
                                                          float* __restrict__ dst, int B, int L, int n, int dim) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= B * L) return;
    const int b = row / L, iv = inv[row];
    float* o = dst + (size_t)row * dim;
    const float* s = src + ((size_t)b * n + (iv < 0 ? 0 : iv)) * dim;
    for (int c = lane * 4; c < dim; c += 256) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (iv >= 0) v = *reinterpret_cast<const f32x4*>(s + c);
        *reinterpret_cast<f32x4*>(o + c) = v;
    }
}

// xdec[b,t,:] = (inv[b,t] < 0 ? mask_token[slot[t]] : y[b, inv[b,t], :]) + pos[t,:] + date8(b, date_row[t])
__global__ __launch_bounds__(256) void unmask_kernel(const float* __restrict__ y, const int* __restrict__ inv,
                                                     const float* __restrict__ mask_token, const int* __restrict__ tok_slot,
                                                     const float* __restrict__ pos, const float* __restrict__ date,
                                                     const int* __restrict__ date_row, int n_date_rows,
                                                     float* __restrict__ xdec, int B, int L, int n_vis, int Dd, int slot_stride) {
    // slot_stride: 0 = tok_slot [L] (a position's own mask token: the stable tie order); L = tok_slot [B, L] (per-sample map:
    // the reference's implementation-defined placement, mh_unmask_assemble_per_sample)
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= B * L) return;
    const int b = row / L, t = row - b * L;
    const int iv = inv[row];
    const float* s = iv < 0 ? mask_token + (size_t)tok_slot[(size_t)b * slot_stride + t] * Dd : y + ((size_t)b * n_vis + iv) * Dd;
    const float* pr = pos + (size_t)t * Dd;
    const float* dr = date ? date + ((size_t)b * n_date_rows + date_row[t]) * 8 : nullptr;
    float* o = xdec + (size_t)row * Dd;
    for (int c = lane * 4; c < Dd; c += 256) {
        f32x4 v = *reinterpret_cast<const f32x4*>(s + c) + *reinterpret_cast<const f32x4*>(pr + c);
        if (dr && c >= Dd - 8) v += *reinterpret_cast<const f32x4*>(dr + (c - (Dd - 8)));
        *reinterpret_cast<f32x4*>(o + c) = v;
    }
}

// dmask_token[:] (ONE row = the gradient of modality `slot`'s mask token) += sum over its masked tokens; each block reduces a strip of token rows in registers first.
// Round 3: 1024 threads and 512 rows per block instead of 256 / 128 -- the launch was bound by its memory-level parallelism (two
// row lanes x four loads per block) and by B*L/128 same-address atomics per column (36 MB in 46 us = 0.8 TB/s at 32768 x 512).
constexpr int UM_ROWS = 512, UM_NT = 1024;
__global__ __launch_bounds__(UM_NT) void unmask_bwd_token_kernel(const float* __restrict__ dxdec, const uint8_t* __restrict__ mask,
                                                                 const int* __restrict__ tok_slot, float* __restrict__ dmask_token,
                                                                 int B, int L, int Dd, int slot, int t_lo, int t_hi, int slot_stride) {
    // grid: ceil(B*(t_hi-t_lo) / UM_ROWS); the threads are (row lane, float4 column): Dd/4 <= 256 columns, UM_NT/(Dd/4) rows side
    // by side; four independent row loads in flight per thread (the row predicate is applied to the loaded value's use, not to
    // a branch around a dependent chain)
    __shared__ __attribute__((aligned(16))) float red[UM_NT * 4];
    __shared__ int row_off[UM_ROWS];               // element offset of a contributing row's start, -1 otherwise
    const int span = t_hi - t_lo, total = B * span;
    const int cols = Dd >> 2, lanes = UM_NT / cols;
    const int col = threadIdx.x % cols, sub = threadIdx.x / cols;
    const int r0 = blockIdx.x * UM_ROWS;
    // the row predicates first (mask byte + slot id are two loads the row's data load would otherwise wait for: with them in the
    // loop the launch was a chain of dependent memory latencies, 47 us for 36 MB whatever the block shape)
    for (int rr = threadIdx.x; rr < UM_ROWS; rr += UM_NT) {
        const int r = r0 + rr;
        int off = -1;
        if (r < total) {
            const int b = r / span, t = t_lo + (r - b * span);
            if (mask[(size_t)b * L + t] && tok_slot[(size_t)b * slot_stride + t] == slot) off = b * L + t;
        }
        row_off[rr] = off;
    }
    __syncthreads();
    f32x4 acc = {0, 0, 0, 0};
    if (sub < lanes) {
#pragma unroll 2
        for (int rr = sub; rr < UM_ROWS; rr += 4 * lanes) {
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v[u] = (f32x4){0, 0, 0, 0};
                const int ri = rr + u * lanes;
                const int off = ri < UM_ROWS ? row_off[ri] : -1;
                if (off >= 0) v[u] = *reinterpret_cast<const f32x4*>(dxdec + (size_t)off * Dd + 4 * col);
            }
            acc += (v[0] + v[1]) + (v[2] + v[3]);
        }
    }
    *reinterpret_cast<f32x4*>(red + 4 * threadIdx.x) = acc;
    __syncthreads();
    if (sub == 0) {
        for (int s = 1; s < lanes; ++s) acc += *reinterpret_cast<const f32x4*>(red + 4 * (s * cols + col));
#pragma unroll
        for (int e = 0; e < 4; ++e) if (acc[e] != 0.f) atomicAdd(dmask_token + 4 * col + e, acc[e]);
    }
}

__global__ __launch_bounds__(1024) void count_masked_kernel(const uint8_t* __restrict__ mask, int B, int L, int t_lo, int t_hi,
                                                            int* __restrict__ out, int mult, int accumulate) {
    // one block, 16 waves: wave w walks rows w, w + 16, ... with 64 consecutive bytes per load instruction (independent
    // loads, no index division); a single 256-thread block with a divide per element took 41 us for 32 x 1024 tokens
    __shared__ float red[16];
    const int span = t_hi - t_lo, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float c = 0.f;
    for (int b = w; b < B; b += 16) {
        const uint8_t* row = mask + (size_t)b * L + t_lo;
        for (int t = lane; t < span; t += 64) c += row[t] ? 1.f : 0.f;
    }
    c = block_sum<16>(c, red);
    if (threadIdx.x == 0) *out = (accumulate ? *out : 0) + (int)(c + 0.5f) * mult;
}

}  // namespace

extern "C" int mh_mask_select(const float* noise, const uint8_t* struct_mask, int* visible_idx, int* masked_idx, int* inv,
                              uint8_t* mask, int B, int L, int k, void* stream) {
    MH_CHECK_ARG(noise && visible_idx && masked_idx && inv && mask, "mh_mask_select: null pointer");
    MH_CHECK_ARG(B > 0 && L > 0 && k >= 0 && k <= L && L <= 8192, "mh_mask_select: bad sizes B=%d L=%d k=%d", B, L, k);
    hipLaunchKernelGGL(mask_select_kernel, dim3(B), dim3(1024), (size_t)((L + 3) & ~3) * 4 + 64, (hipStream_t)stream, noise, struct_mask,
                       visible_idx, masked_idx, inv, mask, L, k);
    MH_LAUNCH_CHECK();
    return 0;
}

extern "C" int mh_gather_rows(const float* src, const int* idx, float* dst, int B, int src_L, int n_idx, int dim, int dst_L,
                              int dst_off, void* stream) {
    MH_CHECK_ARG(src && idx && dst && dim % 4 == 0 && dst_off + n_idx <= dst_L, "mh_gather_rows: bad arguments");
    hipLaunchKernelGGL(gather_rows_kernel, dim3(ceil_div((long)B * n_idx, 4)), dim3(256), 0, (hipStream_t)stream, src, idx, dst, B,
                       src_L, n_idx, dim, dst_L, dst_off);
    MH_LAUNCH_CHECK();
    return 0;
}

extern "C" int mh_scatter_rows(const float* ddst, const int* idx, float* dsrc, int B, int src_L, int n_idx, int dim,
                               int dst_L, int dst_off, void* stream) {
    MH_CHECK_ARG(ddst && idx && dsrc && dim % 4 == 0 && dst_off + n_idx <= dst_L, "mh_scatter_rows: bad arguments");
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(ceil_div((long)B * n_idx, 4)), dim3(256), 0, (hipStream_t)stream, ddst, idx, dsrc,
                       B, src_L, n_idx, dim, dst_L, dst_off);
    MH_LAUNCH_CHECK();
    return 0;
}

extern "C" int mh_expand_rows(const float* src, const int* inv, float* dst, int B, int L, int n, int dim, void* stream) {
    MH_CHECK_ARG(src && inv && dst && B > 0 && L > 0 && n > 0 && dim % 4 == 0, "mh_expand_rows: bad arguments");
    hipLaunchKernelGGL(expand_rows_kernel, dim3(ceil_div((long)B * L, 4)), dim3(256), 0, (hipStream_t)stream, src, inv, dst, B, L,
                       n, dim);
    MH_LAUNCH_CHECK();
    return 0;
}

extern "C" int mh_unmask_assemble(const float* y, const int* inv, const float* mask_token, const int* tok_slot,
                                  const float* pos, const float* date, const int* date_row, int n_date_rows, float* xdec,
                                  int B, int L, int n_vis, int Dd, void* stream) {
    MH_CHECK_ARG(y && inv && mask_token && tok_slot && pos && xdec && Dd % 4 == 0 && Dd >= 8, "mh_unmask_assemble: bad arguments");
    MH_CHECK_ARG(!date || date_row, "mh_unmask_assemble: date without date_row");
    hipLaunchKernelGGL(unmask_kernel, dim3(ceil_div((long)B * L, 4)), dim3(256), 0, (hipStream_t)stream, y, inv, mask_token,
                       tok_slot, pos, date, date_row, n_date_rows, xdec, B, L, n_vis, Dd, 0);
    MH_LAUNCH_CHECK();
    return 0;
}

extern "C" int mh_unmask_assemble_per_sample(const float* y, const int* inv, const float* mask_token, const int* tok_slot_bl,
                                             const float* pos, const float* date, const int* date_row, int n_date_rows,
                                             float* xdec, int B, int L, int n_vis, int Dd, void* stream) {
    MH_CHECK_ARG(y && inv && mask_token && tok_slot_bl && pos && xdec && Dd % 4 == 0 && Dd >= 8, "mh_unmask_assemble_per_sample: bad arguments");
    MH_CHECK_ARG(!date || date_row, "mh_unmask_assemble_per_sample: date without date_row");
    hipLaunchKernelGGL(unmask_kernel, dim3(ceil_div((long)B * L, 4)), dim3(256), 0, (hipStream_t)stream, y, inv, mask_token,
                       tok_slot_bl, pos, date, date_row, n_date_rows, xdec, B, L, n_vis, Dd, L);
    MH_LAUNCH_CHECK();
    return 0;
}

extern "C" int mh_unmask_token_grad(const float* dxdec, const uint8_t* mask, const int* tok_slot, float* dmask_token, int B,
                                    int L, int Dd, int slot, int t_lo, int t_hi, void* stream) {
    MH_CHECK_ARG(dxdec && mask && tok_slot && dmask_token && Dd % 4 == 0 && Dd <= 1024, "mh_unmask_token_grad: bad arguments");
    MH_CHECK_ARG(0 <= t_lo && t_lo < t_hi && t_hi <= L, "mh_unmask_token_grad: bad token range");
    hipLaunchKernelGGL(unmask_bwd_token_kernel, dim3(ceil_div((long)B * (t_hi - t_lo), UM_ROWS)), dim3(UM_NT), 0,
                       (hipStream_t)stream, dxdec, mask, tok_slot, dmask_token, B, L, Dd, slot, t_lo, t_hi, 0);
    MH_LAUNCH_CHECK();
    return 0;
}

extern "C" int mh_unmask_token_grad_per_sample(const float* dxdec, const uint8_t* mask, const int* tok_slot_bl, float* dmask_token,
                                               int B, int L, int Dd, int slot, void* stream) {
    MH_CHECK_ARG(dxdec && mask && tok_slot_bl && dmask_token && Dd % 4 == 0 && Dd <= 1024 && B > 0 && L > 0,
                 "mh_unmask_token_grad_per_sample: bad arguments");
    hipLaunchKernelGGL(unmask_bwd_token_kernel, dim3(ceil_div((long)B * L, UM_ROWS)), dim3(UM_NT), 0, (hipStream_t)stream, dxdec, mask,
                       tok_slot_bl, dmask_token, B, L, Dd, slot, 0, L, L);
    MH_LAUNCH_CHECK();
    return 0;
}

extern "C" int mh_count_masked_elems(const uint8_t* mask, int B, int L, int t_lo, int t_hi, int* out, int mult, int accumulate,
                                     void* stream) {
    MH_CHECK_ARG(mask && out && 0 <= t_lo && t_lo < t_hi && t_hi <= L && mult > 0, "mh_count_masked_elems: bad arguments");
    hipLaunchKernelGGL(count_masked_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, mask, B, L, t_lo, t_hi, out, mult,
                       accumulate);
    MH_LAUNCH_CHECK();
    return 0;
}

extern "C" int mh_count_masked(const uint8_t* mask, int B, int L, int t_lo, int t_hi, int* out, void* stream) {
    MH_CHECK_ARG(mask && out && 0 <= t_lo && t_lo < t_hi && t_hi <= L, "mh_count_masked: bad arguments");
    hipLaunchKernelGGL(count_masked_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, mask, B, L, t_lo, t_hi, out, 1, 0);
    MH_LAUNCH_CHECK();
    return 0;
}
