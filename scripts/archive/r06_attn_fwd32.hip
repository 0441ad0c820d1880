// Round 6, experiment 27 (NOT compiled into the library): the attention forward on v_mfma_f32_32x32x16_bf16.  Correct (the 26 attention
// tests of tests/test_kernels_gpu.py pass), 0-5 % slower than the 16x16x32 kernel -- see profiles/r06_experiments.md.  It was a block of
// maestro_amd/csrc/attn.hip between the forward and the dQ kernel and uses that file's helpers (tile_load, tile_store_row / _tr, row_swz, ...).
// =============================================================================================== forward, 32x32x16 MFMAs (experiment)
// The same forward on v_mfma_f32_32x32x16_bf16: an MFMA holds the SIMD's issue port for 8 cycles whatever its shape, so the 32 x 32
// form does the tile's products with half the issue time (12 instead of 20 MFMAs per 64-key tile at D = 32).  Lane (query = lane & 31,
// half = lane >> 5) holds S^T[key = 32 kt + 8 a + 4 half + r][query] in element 4 a + r of key tile kt: 32 scores of ONE query, the
// other half-wave holds the query's other 32 keys.  Elements 8 b .. 8 b + 7 are, as they stand, the second operand of the P V MFMA of
// key block (kt, b) under the key order k = 8 half + j <-> key 32 kt + 16 b + 8 (j >> 2) + 4 half + (j & 3), which frag_tr32 applies to V.
template <int D>
__device__ __forceinline__ bf16x8 frag_row32(const unsigned char* img, int r0, int kk) {
    const int l = threadIdx.x & 63, row = r0 + (l & 31), ch = 2 * kk + (l >> 5);
    return *reinterpret_cast<const bf16x8*>(img + row * (2 * D) + ((ch ^ row_swz<D>(row)) << 4));
}
template <int D>
__device__ __forceinline__ bf16x8 frag_tr32(const unsigned char* img, int r0, int cb0) {
    const int l = threadIdx.x & 63, G = l >> 4, q = (l & 15) >> 2, p = l & 3;
    const int cb = cb0 + (G & 1), r_lo = r0 + 4 * (G >> 1) + q, r_hi = r_lo + 8;
    const lds_u8* b = (const lds_u8*)img;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)(b + r_lo * (2 * D) + (((cb ^ tr_swz<D>(r_lo)) << 5) + p * 8)));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)(b + r_hi * (2 * D) + (((cb ^ tr_swz<D>(r_hi)) << 5) + p * 8)));
    s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, r);
}
__device__ __forceinline__ bf16x8 pack8(const f32x16& a, int e0) {
    u32x4 r = {pack_bf2(a[e0], a[e0 + 1]), pack_bf2(a[e0 + 2], a[e0 + 3]), pack_bf2(a[e0 + 4], a[e0 + 5]), pack_bf2(a[e0 + 6], a[e0 + 7])};
    return __builtin_bit_cast(bf16x8, r);
}

template <int D>
__global__ __launch_bounds__(256) void attn_fwd32_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                         float* __restrict__ lse, int N, int H, float scale) {
    constexpr int KK = D / 16, DT = D / 32, TB = 64 * D * 2;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * TB];  // K row image | V transpose image
    unsigned char* k_img = smem;
    unsigned char* v_img = smem + TB;
    const int gx = (N + 127) >> 7;
    const int wid = xcd_remap(blockIdx.x, gridDim.x);
    const int b = wid / (gx * H), h = (wid / gx) % H;
    const int q_blk = (wid % gx) * 128;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, half = l >> 5, lq = l & 31;
    const size_t rs = (size_t)3 * H * D;
    const bf16_t* qb = qkv + (size_t)b * N * rs + (size_t)h * D;
    const bf16_t* kb = qb + (size_t)H * D;
    const bf16_t* vb = qb + (size_t)2 * H * D;
    const int q0 = q_blk + 32 * w, q = q0 + lq;

    bf16x8 qf[KK];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
        u32x4 z = {0, 0, 0, 0};
        if (q < N) z = *reinterpret_cast<const u32x4*>(qb + (size_t)q * rs + 16 * kk + 8 * half);
        qf[kk] = __builtin_bit_cast(bf16x8, z);
    }
    constexpr bool SUM_MFMA = ATTN_SUM_MFMA && D == 32;
    f32x16 o[DT], ol;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        ol[e] = 0.f;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[dt][e] = 0.f;
    }
    bf16x8 ones;
    {
        const short one = lq == 0 ? (short)0x3F80 : (short)0;
        const s16x8 r = {one, one, one, one, one, one, one, one};
        ones = __builtin_bit_cast(bf16x8, r);
    }
    float m = 0.f, lsum = 0.f;         // the query's reference exponent (log2 units) and, without SUM_MFMA, this lane's share of the denominator
    const float c = scale * LOG2E;
    const int ntile = (N + 63) / 64;
    u32x4 rk[2], rv[2];
    tile_load<D>(kb, rs, 0, N, rk);
    tile_load<D>(vb, rs, 0, N, rv);
    auto kv_tile = [&](int it, auto tail_tag) {
        constexpr bool TAIL = decltype(tail_tag)::value;
        const int kv0 = it * 64;
        __syncthreads();
        tile_store_row<D>(k_img, rk);
        tile_store_tr<D>(v_img, rv);
        __syncthreads();
        if (it + 1 < ntile) {
            tile_load<D>(kb, rs, kv0 + 64, N, rk);
            tile_load<D>(vb, rs, kv0 + 64, N, rv);
        }
        if (q0 >= N) return;
        f32x16 s[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int e = 0; e < 16; ++e) s[kt][e] = 0.f;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
                s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row32<D>(k_img, 32 * kt, kk), qf[kk], s[kt], 0, 0, 0);
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float t = __builtin_fmaf(s[kt][e], c, -m);            // the exp2 argument relative to the reference
                if constexpr (TAIL) {
                    if (kv0 + 32 * kt + 8 * (e >> 2) + 4 * half + (e & 3) >= N) t = -INFINITY;
                }
                s[kt][e] = t;
            }
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int e = 0; e < 16; e += 2) mx = __builtin_fmaxf(__builtin_fmaxf(mx, s[kt][e]), s[kt][e + 1]);
        const bool first = it == 0;
        const bool leave = mx > ATTN_LAZY_RANGE || (first && mx < -ATTN_LAZY_RANGE);
        if (__builtin_amdgcn_ballot_w64(leave) != 0) {
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            mx = first ? fmaxf(mx, -3.0e38f) : fmaxf(mx, 0.f);
            const float alpha = first ? 1.f : exp2_fast(-mx);
            m += mx;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int e = 0; e < 16; ++e) s[kt][e] -= mx;
            if constexpr (SUM_MFMA) ol[0] *= alpha;
            else lsum *= alpha;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int e = 0; e < 16; ++e) o[dt][e] *= alpha;
        }
        float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int e = 0; e < 16; e += 4) {
                const float a0 = exp2_fast(s[kt][e]), a1 = exp2_fast(s[kt][e + 1]), a2 = exp2_fast(s[kt][e + 2]), a3 = exp2_fast(s[kt][e + 3]);
                s[kt][e] = a0; s[kt][e + 1] = a1; s[kt][e + 2] = a2; s[kt][e + 3] = a3;
                if constexpr (!SUM_MFMA) { p0 += a0; p1 += a1; p2 += a2; p3 += a3; }
            }
        if constexpr (!SUM_MFMA) lsum += (p0 + p1) + (p2 + p3);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
                const bf16x8 pf = pack8(s[kt], 8 * bb);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr32<D>(v_img, 32 * kt + 16 * bb, 2 * dt), pf, o[dt], 0, 0, 0);
                if constexpr (SUM_MFMA) ol = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, pf, ol, 0, 0, 0);
            }
    };
    const int nfull = N / 64;
    for (int it = 0; it < nfull; ++it) kv_tile(it, std::false_type{});
    if (nfull < ntile) kv_tile(nfull, std::true_type{});
    float lt;
    if constexpr (SUM_MFMA) lt = __shfl(ol[0], lq, 64);               // lane lq of the first half holds row 0 of the query's column
    else lt = lsum + __shfl_xor(lsum, 32, 64);
    if (q >= N) return;
    const float inv = 1.f / lt;
    bf16_t* orow = out + ((size_t)b * N + q) * H * D + (size_t)h * D;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            u32x2 pk = {pack_bf2(o[dt][4 * a] * inv, o[dt][4 * a + 1] * inv), pack_bf2(o[dt][4 * a + 2] * inv, o[dt][4 * a + 3] * inv)};
            *reinterpret_cast<u32x2*>(orow + 32 * dt + 8 * a + 4 * half) = pk;
        }
    if (half == 0) lse[((size_t)b * H + h) * N + q] = m * 0.6931471805599453f + logf(lt);
}

