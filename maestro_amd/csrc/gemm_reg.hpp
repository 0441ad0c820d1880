// The register-staged 128x128x64 tile's building blocks, shared by gemm.hip (one tile per workgroup) and gemm_pp.hip (persistent
// workgroups with a second accumulator set): global -> registers (buffer loads), registers -> swizzled LDS, LDS -> MFMA fragments.
//   * K-minor operand ([rows][k], k contiguous in memory): 128-B LDS rows, 16-B chunk index XOR (row & 7),
//     fragments by ds_read_b128 (conflict-free: 16-lane groups hit 16 distinct 16-B slots).
//   * K-major operand ([k][cols], cols contiguous: dgrad B = W, wgrad A = dY and B = X): 256-B LDS rows,
//     32-B chunk index XOR f(k), fragments by ds_read_b64_tr_b16 (hardware transpose).
#pragma once
#include "gemm_common.hpp"

namespace {

constexpr int BM = 128, BN = 128, BK = 64, NT = 256;
constexpr int TILE_BYTES = 128 * 64 * 2;  // one operand tile, 16 KiB

typedef __attribute__((address_space(3))) unsigned char lds_u8;

__device__ __forceinline__ int kmajor_f(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }

// ---- global -> registers (4 x 16 B per thread per operand tile)
// NI = 16-byte pieces per thread = (tile rows) / 32 for a K-minor operand (4: 128 rows; 2: 64; 6: 192); K-major tiles are 128 wide
template <bool KMAJOR, int NI = 4>
__device__ __forceinline__ void load_tile(const bf16_t* __restrict__ P, int ld, int row0, int nrows, int k0, int kend,
                                          u32x4 (&v)[NI]) {
    static_assert(!KMAJOR || NI == 4, "K-major operand tiles are 128 columns wide");
    const int t = threadIdx.x;
    if constexpr (!KMAJOR) {  // memory: P[row * ld + k]
        const int c = t & 7, r = t >> 3;
        const int gk = k0 + c * 8;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int grow = row0 + r + 32 * i;
            u32x4 z = {0, 0, 0, 0};
            if (grow < nrows && gk < kend) z = *reinterpret_cast<const u32x4*>(P + (size_t)grow * ld + gk);
            v[i] = z;
        }
    } else {  // memory: P[k * ld + col]
        const int c = t & 15, kk = t >> 4;
        const int gcol = row0 + c * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int gk = k0 + kk + 16 * i;
            u32x4 z = {0, 0, 0, 0};
            if (gk < kend && gcol < nrows) z = *reinterpret_cast<const u32x4*>(P + (size_t)gk * ld + gcol);
            v[i] = z;
        }
    }
}

// ---- fast path: buffer loads.  The 128-bit resource descriptor carries the exact byte extent of the operand, so rows
// beyond M / N (K-minor) or beyond K (K-major) read as zero in hardware; per-thread byte offsets are loop-invariant
// 32-bit VGPRs and the K advance is ONE scalar offset -> no per-step predication, no 64-bit vector address math.
template <bool KMAJOR, int NI = 4>
__device__ __forceinline__ void tile_offsets(int ld, int row0, int (&voff)[NI]) {
    static_assert(!KMAJOR || NI == 4, "K-major operand tiles are 128 columns wide");
    const int t = threadIdx.x;
    if constexpr (!KMAJOR) {
        const int c = t & 7, r = t >> 3;
#pragma unroll
        for (int i = 0; i < NI; ++i) voff[i] = ((row0 + r + 32 * i) * ld + c * 8) * 2;
    } else {
        const int c = t & 15, kk = t >> 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) voff[i] = ((kk + 16 * i) * ld + row0 + c * 8) * 2;
    }
}
template <int NI>
__device__ __forceinline__ void load_tile_fast(__amdgpu_buffer_rsrc_t rsrc, const int (&voff)[NI], int soff, u32x4 (&v)[NI]) {
#pragma unroll
    for (int i = 0; i < NI; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[i], soff, 0);
}

// ---- registers -> swizzled LDS tile
template <bool KMAJOR, int NI = 4>
__device__ __forceinline__ void store_tile(unsigned char* tile, const u32x4 (&v)[NI]) {
    static_assert(!KMAJOR || NI == 4, "K-major operand tiles are 128 columns wide");
    const int t = threadIdx.x;
    if constexpr (!KMAJOR) {
        const int c = t & 7, r = t >> 3;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int row = r + 32 * i;
            *reinterpret_cast<u32x4*>(tile + row * 128 + ((c ^ (row & 7)) << 4)) = v[i];
        }
    } else {
        const int c = t & 15, kk = t >> 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = kk + 16 * i;
            *reinterpret_cast<u32x4*>(tile + k * 256 + ((((c >> 1) ^ kmajor_f(k)) << 5) | ((c & 1) << 4))) = v[i];
        }
    }
}

// ---- LDS -> MFMA fragment: 8 bf16 along k for row/col (rc0 + lane&15), k = 32*s + 8*(lane>>4) + j
template <bool KMAJOR>
__device__ __forceinline__ bf16x8 read_frag(const unsigned char* tile, int rc0, int s) {
    const int l = threadIdx.x & 63;
    if constexpr (!KMAJOR) {
        const int row = rc0 + (l & 15), ch = 4 * s + (l >> 4);
        return *reinterpret_cast<const bf16x8*>(tile + row * 128 + ((ch ^ (row & 7)) << 4));
    } else {
        const int g = l >> 4, qrow = (l & 15) >> 2, p = l & 3, q = rc0 >> 4;
        const int k_lo = 32 * s + 8 * g + qrow, k_hi = k_lo + 4;
        const lds_u8* base = (const lds_u8*)tile;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(base + k_lo * 256 + (((q ^ kmajor_f(k_lo)) << 5) + p * 8)));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(base + k_hi * 256 + (((q ^ kmajor_f(k_hi)) << 5) + p * 8)));
        s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, r);
    }
}

}  // namespace
