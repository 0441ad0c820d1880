// LDS-DMA ring pieces of the large-tile GEMM kernels (gemm_dma.hip): tile geometry, the LDS swizzles, per-lane DMA source offsets and the
// fragment reads.  See the header comment of gemm_dma.hip for the ring protocol and the two operand images.
#pragma once
#include "gemm_common.hpp"

namespace {

constexpr int BK = 32;

// Tile configuration: WM x WN waves, each owning a (16 MT) x 64 accumulator block (MT = 8: 128 x 64, MT = 4: 64 x 64); S ring stages.
//   <2,4,4> 256x256, 512 threads, 128 KiB ring (1 workgroup / CU)      <2,2,3> 256x128 and <1,4,3> 128x256, 256 threads,
//   72 KiB ring (2 / CU)      <1,2,4> 128x128, 128 threads, 64 KiB ring (2 / CU)
//   <2,2,4,4> 128x128 with four 64x64 waves (the register-staged kernel's geometry, DMA-fed), 64 KiB ring (2 / CU)
template <int WM_, int WN_, int S_, int MT_ = 8>
struct Tile {
    static constexpr int WM = WM_, WN = WN_, S = S_, MT = MT_;
    static constexpr int BM = 16 * MT * WM, BN = 64 * WN, NW = WM * WN, NT = 64 * NW;
    static constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE_BYTES = A_BYTES + B_BYTES;
    static constexpr int PA = A_BYTES / 1024 / NW, PB = B_BYTES / 1024 / NW;   // 1-KiB DMA pieces per wave and K step
    static constexpr int LDS_BYTES = S * STAGE_BYTES;
    static constexpr int MIN_WG = NT >= 512 ? 1 : 512 / NT;   // workgroups per CU the register budget must allow (2 waves / SIMD)
    static_assert(BN >= 128, "the K-major swizzle needs at least 8 32-byte chunks per row");
    static_assert(PA * NW * 1024 == A_BYTES && PB * NW * 1024 == B_BYTES, "pieces must divide evenly over the waves");
};

typedef __attribute__((address_space(3))) unsigned char lds_u8;
typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ int kminor_sw(int row) { return (0x78 >> (2 * ((row >> 2) & 3))) & 3; }
__device__ __forceinline__ int kmajor_sw(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }

// Per-lane source byte offsets (relative to the operand base at k = 0) of the NP DMA pieces a wave issues per K step for
// an operand tile of BX rows / columns (NW waves share its BX / 16 pieces).
template <bool KMAJOR, int BX, int NW, int NP>
__device__ __forceinline__ void piece_offsets(int ld, int rc0, int w, int l, int (&voff)[NP]) {
#pragma unroll
    for (int h = 0; h < NP; ++h) {
        const int piece = w + NW * h;
        if constexpr (!KMAJOR) {                           // piece = 16 rows x 64 B
            const int row = piece * 16 + (l >> 2), pos = l & 3;
            voff[h] = ((rc0 + row) * ld + ((pos ^ kminor_sw(row)) << 3)) * 2;
        } else {                                           // piece = (512 / BX) k-rows x (2 BX) B
            constexpr int LPR = BX / 8;                    // lanes (16-byte chunks) per k-row
            const int k = piece * (64 / LPR) + l / LPR, s16 = l % LPR;
            const int q = (s16 >> 1) ^ kmajor_sw(k);
            voff[h] = (k * ld + rc0 + (((q << 1) | (s16 & 1)) << 3)) * 2;
        }
    }
}

// fragment of rows/cols (rc0 + lane&15) of a stage's operand image: 8 bf16 along k = 8*(lane>>4) + j
template <bool KMAJOR, int BX>
__device__ __forceinline__ bf16x8 read_frag(const unsigned char* img, int rc0, int l = threadIdx.x & 63) {
    if constexpr (!KMAJOR) {
        const int row = rc0 + (l & 15), g = l >> 4;
        return *reinterpret_cast<const bf16x8*>(img + row * 64 + ((g ^ kminor_sw(row)) << 4));
    } else {
        constexpr int ROWB = BX * 2;
        const int g = l >> 4, qrow = (l & 15) >> 2, p = l & 3, q = rc0 >> 4;
        const int k_lo = 8 * g + qrow, k_hi = k_lo + 4;
        const lds_u8* base = (const lds_u8*)img;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(base + k_lo * ROWB + (((q ^ kmajor_sw(k_lo)) << 5) + p * 8)));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(base + k_hi * ROWB + (((q ^ kmajor_sw(k_hi)) << 5) + p * 8)));
        s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, r);
    }
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

}  // namespace
