/* C ABI of libmaestro_hip.so -- the MI355X (gfx950) kernels of MAESTRO's MAE pretraining hot path.
 *
 * The reference (IGNF/MAESTRO) is pure Python with no FFI of its own; the boundary it would bind
 * is therefore one entry point per fused stage of the per-step path.  Each declaration cites the reference
 * lines whose ATen/einops op sequence it replaces (paths relative to the reference root).
 *
 * Conventions (all entry points):
 *   - plain pointers + sizes, no torch types; every pointer is DEVICE memory owned by the caller unless noted;
 *   - asynchronous on `stream` (a hipStream_t passed as void*), never synchronises, allocates nothing, keeps no
 *     global mutable state, re-entrant across streams/threads; graph-capturable;
 *   - returns 0 on success, <0 for a bad argument, >0 = hipError_t; mh_last_error() gives a thread-local message;
 *   - bf16 = raw uint16 bits; "f32" = float; row-major with explicit leading dimensions (in elements).
 */
#ifndef MAESTRO_HIP_H
#define MAESTRO_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

const char* mh_last_error(void);
int mh_version(void);

/* ---------------------------------------------------------------------------------------------- GEMM (MFMA)
 * C[M,N] = op(A) * op(B) (+ epilogue), bf16 operands, fp32 accumulation on v_mfma_f32_16x16x32_bf16.
 *   layout 0 "NT": A[M,K] (lda), B[N,K] (ldb)      -- forward of nn.Linear / 1x1 conv / patch-embed conv
 *   layout 1 "NN": A[M,K] (lda), B[K,N] (ldb)      -- dgrad:  dX = dY * W
 *   layout 2 "TN": A[K,M] (lda), B[K,N] (ldb)      -- wgrad:  dW = dY^T * X   (K = tokens)
 * Replaces: vit_pytorch Attention.to_qkv / to_out / FeedForward Linear (call sites maestro/ssl/mae.py:135-174),
 * enc_to_dec Linear (mae.py:145-154), Patchify conv (maestro/layers/embed.py:48-60), Pixelify 1x1 conv
 * (embed.py:139-151) and their autograd transposes.
 * flags: bit0 out_f32 (else bf16) | bit1 +bias[N] | bit2 GELU(erf) (aux_out, if given, receives the bf16
 *        pre-activation) | bit3 += res[M,N] f32 (ldr) | bit4 *= gelu'(aux_in[M,N] bf16) | bit5 atomic accumulate
 *        into C (f32 only; split-K is applied automatically for layout 2).
 * Requirements: K % 8 == 0 (layouts 0/1), lda/ldb % 8 == 0, N % 4 == 0 and ldc % 4 == 0 (f32 out) or % 8 (bf16 out,
 * aux), 16-byte aligned bases; residual needs f32 output, GELU/GELU' need bf16 output. */
#define MH_GEMM_OUT_F32 1
#define MH_GEMM_BIAS 2
#define MH_GEMM_GELU 4
#define MH_GEMM_RESIDUAL 8
#define MH_GEMM_DGELU 16
#define MH_GEMM_ATOMIC 32
#define MH_GEMM_COLSUM 64   /* bf16 output only: colsum[(m / 64), n] = sum over the 64-row block of C[m, n] (f32 values before
                             * rounding); colsum is a [ceil(M / 64), N] workspace, plain stores; reduce it with mh_colsum */
#define MH_GEMM_AUX_DGELU 128 /* with MH_GEMM_GELU: aux_out receives GELU'(pre-activation) (bf16) instead of the pre-activation: the
                               * CDF / PDF are already at hand in the forward epilogue, so the backward only multiplies */
#define MH_GEMM_MULAUX 256    /* bf16 output only: C *= aux_in[M, N] (bf16) -- the backward of GELU with the saved derivative */
#define MH_GEMM_AUX_U8 1024   /* the saved GELU derivative (aux_out of MH_GEMM_AUX_DGELU, aux_in of MH_GEMM_MULAUX) is stored as one
                               * BYTE per element, code = round((GELU' + 0.13) * 200): GELU' lies in [-0.129, 1.129], the step of
                               * 0.005 is about what bf16 resolves near 1; halves the derivative's HBM bytes (ldaux in bytes) */
#define MH_GEMM_C8_E5M2 512   /* mh_gemm_fp8 only: the fp8 copy c8 of the output is e5m2 (a gradient: the next dgrad's A operand) */
/* mh_gemm_fp8 only, at most one of them: force a tile / ring form instead of the library's size rule (experiments, tests).
 * The library never reads the environment: every such choice is an argument (maestro_amd/hip.py maps MH_FP8_TILE onto these). */
#define MH_GEMM_FP8_TILE_256 2048    /* 256 x 256, 8 waves */
#define MH_GEMM_FP8_TILE_128 4096    /* 128 x 128, two-stage ring, two workgroups per CU */
#define MH_GEMM_FP8_TILE_128D 8192   /* 128 x 128, four-stage ring */
int mh_gemm_bf16(int layout, int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C, int ldc,
                 int flags, const float* bias, const float* res, int ldr, const void* aux_in, void* aux_out,
                 int ldaux, float* colsum, void* stream);

/* Same contract with an explicit kernel / tile choice (what mh_gemm_bf16 picks by itself with MH_TILE_AUTO):
 *   MH_TILE_REG_128      128x128x64 tile, register-staged double-buffered LDS (gemm.hip): every shape / tail
 *   MH_TILE_DMA_256      256x256x32 tile, 8 waves, 4-stage LDS-DMA ring (gemm_dma.hip)
 *   MH_TILE_DMA_256x128 / MH_TILE_DMA_128x256   4 waves, 3-stage ring;   MH_TILE_DMA_128   128x128, 2 waves, 4-stage ring
 *   MH_TILE_DMA_128x4    128x128, four 64x64 waves, 4-stage ring (the register-staged kernel's geometry, DMA-fed)
 *   MH_TILE_DMA_256_LOCKSTEP   MH_TILE_DMA_256 without the wave-group stagger (A/B experiments; bit-identical results)
 *   MH_TILE_PP_128       persistent workgroups over 128x128x64 tiles with a second accumulator set: the epilogue of tile t runs
 *                        inside the main loop of tile t + 1 (gemm_pp.hip); NT / NN, K %% 64 == 0, K >= 512, N %% 128 == 0, and
 *                        flags one of: 0 | BIAS+GELU+AUX_DGELU+AUX_U8 | MULAUX+AUX_U8+COLSUM | OUT_F32+BIAS+RESIDUAL
 * The DMA tiles return -2 (nothing launched, error string untouched) when the problem does not qualify (K %% 32 != 0 with
 * a K-minor operand, operands beyond the 2 GiB buffer-descriptor range): pick another tile.  The host side times the
 * eligible tiles once per distinct (layout, M, N, K, flags) and remembers the fastest (maestro_amd/hip.py). */
enum { MH_TILE_AUTO = -1, MH_TILE_REG_128 = 0, MH_TILE_DMA_256 = 1, MH_TILE_DMA_256x128 = 2, MH_TILE_DMA_128x256 = 3,
       MH_TILE_DMA_128 = 4, MH_TILE_DMA_128x4 = 5, MH_TILE_DMA_256_LOCKSTEP = 6, MH_TILE_PP_128 = 7,
       /* diagnostic builds of MH_TILE_PP_128 (NT only; outputs are NOT the GEMM's): main loop only / epilogue arithmetic without
        * its stores / stores without the GELU arithmetic -- the ablation under profiles/ (scripts/bench_pp_ablate.py).  They exist
        * only in a library built with -DMH_DIAG_TILES (MH_BUILD_FLAGS); the shipped library returns -2 for these ids (nothing
        * launched), so no caller of the C ABI can get a wrong-output kernel by passing a tile id. */
       MH_TILE_PP_128_DIAG1 = 8, MH_TILE_PP_128_DIAG2 = 9, MH_TILE_PP_128_DIAG3 = 10,
       MH_TILE_PP_128_DIAG4 = 11, MH_TILE_PP_128_DIAG5 = 12,  /* full epilogue, other instruction placements */
       /* MH_TILE_REG_128's kernel with 64 x 128 tiles (three workgroups per CU) / 192 x 128 tiles: for outputs whose 128 x 128
        * tiling fills the chip's workgroup slots badly (N = 512 / 768).  NT / NN without MH_GEMM_COLSUM; otherwise = REG_128. */
       MH_TILE_REG_64 = 13, MH_TILE_REG_192 = 14,
       /* stream-K tiles (gemm_sk.hip; mh_gemm_bf16_sk only -- they need a workspace): 192 x 128 / 256 x 128, one workgroup per CU */
       MH_TILE_SK_192 = 15, MH_TILE_SK_256 = 16,
       MH_TILE_SK_DMA_256 = 17 /* 256 x 256 x 32, eight waves, the LDS-DMA ring of MH_TILE_DMA_256 run by persistent workgroups */ };
int mh_gemm_bf16_tile(int tile, int layout, int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C,
                      int ldc, int flags, const float* bias, const float* res, int ldr, const void* aux_in, void* aux_out,
                      int ldaux, float* colsum, void* stream);

/* Stream-K GEMM (gemm_sk.hip): C[M, N] = A[M, K] B^T (layout 0, B [N, K]) or A B (layout 1, B [K, N]) with the long-K, narrow-N
 * problems of the transformer blocks in mind (fc2, out-proj, fc1 / qkv / out-proj dgrads: the nn.Linear call sites of
 * vit_pytorch's Attention / FeedForward built at maestro/ssl/mae.py:135-174).  `grid` persistent four-wave workgroups (one per
 * CU: pass the CU count of the device, or of the stream's CU mask) split tiles x (K / 64) units evenly; a tile shared by
 * several workgroups is summed in workgroup order by the one that owns its last K step (deterministic, no atomics on C).
 *   tile   MH_TILE_SK_DMA_256 (256 x 256 x 32, eight waves, LDS-DMA ring: gemm_sk_dma.hip; K %% 32 == 0, K >= 128, N %% 256 == 0) or the
 *          four-wave register-staged MH_TILE_SK_192 (192 x 128 x 64) / MH_TILE_SK_256 (256 x 128 x 64) (K %% 64 == 0, N %% 128 == 0)
 *   flags  0 (bf16 C) or MH_GEMM_OUT_F32 | MH_GEMM_BIAS | MH_GEMM_RESIDUAL (fp32 C = A B + bias + res); anything the tile does not
 *          serve returns -2 (nothing launched, error string untouched): use mh_gemm_bf16
 *   workspace  mh_gemm_sk_workspace(tile, grid) bytes, 16-byte aligned, owned by the caller, ZEROED ONCE before its first use (the
 *          kernel leaves its flag words zero); launches that may run concurrently (different streams) need different workspaces.
 * Same fp32 sums per output element as mh_gemm_bf16 when no tile is shared; a shared tile adds its K ranges in ascending order. */
long mh_gemm_sk_workspace(int tile, int grid);
int mh_gemm_bf16_sk(int tile, int layout, int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C, int ldc,
                    int flags, const float* bias, const float* res, int ldr, void* workspace, long workspace_bytes, int grid,
                    void* stream);

/* Grouped weight-gradient GEMM: ONE launch over the 256x256 tiles of many independent "TN" problems
 * dW_i[M_i, N_i] (f32) = A_i^T B_i with A_i [K_i, M_i] bf16 (= dY_i), B_i [K_i, N_i] bf16 (= X_i), K_i = tokens.
 * Used to issue all wgrads of a backward segment at once (no split-K, whole-chip tile occupancy).  `table` is a DEVICE
 * array of problems; M, N, lda, ldb %% 8 == 0.  accumulate = 0: plain stores (the entry is the only writer of C); 1: fp32
 * atomic adds into a zeroed C (several entries share one C, e.g. one encoder applied to several groups).
 * `tile_queues` is a DEVICE array [8][queue_len] of tile ids (problem << 16 | tile_m << 8 | tile_n, 0xFFFFFFFF = empty
 * slot), one queue per XCD: workgroup b runs entry [b %% 8][b / 8] (the hardware places workgroup b on XCD b %% 8), so the
 * host decides which tiles share an L2: keep the tiles of one problem in one queue (its dY / X panels are then fetched
 * once per XCD instead of once per tile) and balance sum(K) over the queues. */
typedef struct MhGroupedGemm {
    const void* A; const void* B; void* C;
    int M, N, K, lda, ldb, ldc, reserved, accumulate;
} MhGroupedGemm;
int mh_gemm_grouped_tn(const MhGroupedGemm* table_device, int n_problems, const uint32_t* tile_queues, int queue_len,
                       void* stream);

/* fp8 GEMM (BASELINE configs[4], "ViT-Base MAE fp8 MFMA path"): C[M, N] = (*descale_a) (*descale_b) A8[M, K] B8[N, K]^T
 * with the epilogues of mh_gemm_bf16 (flags, bias, res, aux_in / aux_out, colsum: same meaning; no MH_GEMM_ATOMIC).  Both
 * operands are K-minor bytes: B8 = a weight [N, K] in OCP e4m3 (the forward uses the weight as stored, the dgrad its
 * transposed fp8 shadow), A8 = activations in e4m3 or gradients in e5m2 (a_format).  K %% 128 == 0, lda / ldb %% 16 == 0 (bytes).
 * descale_a / descale_b: DEVICE scalars 1 / scale of the per-tensor quantisers below (mh_quant_batched).  Optional c8
 * (bf16-output epilogues only): an e4m3 copy of the output times *c8_scale (the next fp8 GEMM's A operand: fc1 -> fc2),
 * with max |output| folded into the amax row c8_amax (delayed scaling; see MH_FP8_AMAX_PITCH).  Replaces the nn.Linear calls inside vit_pytorch's Attention /
 * FeedForward (call sites maestro/ssl/mae.py:135-174) when the step runs with fp8 operands. */
enum { MH_FP8_E4M3 = 0, MH_FP8_E5M2 = 1 };
int mh_gemm_fp8(int M, int N, int K, const void* A8, int lda, int a_format, const void* B8, int ldb, void* C, int ldc,
                int flags, const float* descale_a, const float* descale_b, const float* bias, const float* res, int ldr,
                const void* aux_in, void* aux_out, int ldaux, float* colsum, void* c8, int ldc8, const float* c8_scale,
                float* c8_amax, void* stream);
/* Per-tensor fp8 quantisation, batched over a job table (DEVICE array): job = one tensor of n elements (n %% 4 == 0), f32 or
 * bf16 source, `slot` = its entry in the scale / amax tables.  items: DEVICE array of work items job << 32 | chunk (chunks of
 * 4096 elements).  mode 0: amax row of slot = max(itself, max |src|) only;  1: dst = fp8(src * scale[slot]) (+ the transposed
 * copy dst_t [cols, rows] of a [rows, cols] tensor when dst_t != NULL, cols %% 4 == 0);  2: cast AND fold max |src| into amax
 * (delayed scaling of activations: this step casts with the previous step's scale).  Values saturate at the format's largest
 * finite number.  mh_fp8_update_scales: scale = 2^(floor(log2(format_max / amax)) - margin_log2) (1 while amax is 0),
 * descale = 1 / scale, amax reset to 0 -- all on the device, capturable.
 * amax tables: ONE ROW of MH_FP8_AMAX_PITCH floats per slot, not one float -- same-cache-line atomics retire at about one per
 * 10 ns on MI355X (scripts/micro_amax.hip: 18432 LayerNorm rows folding into one word took 217 us instead of 19), so every
 * workgroup folds into sub-slot (its index %% MH_FP8_AMAX_SUBSLOTS) of the row, MH_FP8_AMAX_STRIDE floats (256 bytes) apart; a
 * slot's absmax is the maximum over its row, taken by mh_fp8_update_scales.  Every `amax` pointer of this header (c8_amax,
 * y8_amax, the tables of mh_quant_batched / mh_adamw_fp8) points at such rows. */
#define MH_FP8_AMAX_SUBSLOTS 32
#define MH_FP8_AMAX_STRIDE 64
#define MH_FP8_AMAX_PITCH (MH_FP8_AMAX_SUBSLOTS * MH_FP8_AMAX_STRIDE)
typedef struct MhQuantJob {
    const void* src; void* dst; void* dst_t;
    long n;
    int rows, cols, slot, is_f32, format, reserved;
} MhQuantJob;
int mh_quant_batched(const MhQuantJob* jobs_device, const unsigned long* items_device, int n_items, const float* scale,
                     float* amax, int mode, void* stream);
int mh_fp8_update_scales(float* amax, float* scale, float* descale, int n, float format_max, int margin_log2, void* stream);
/* Byte transposes, batched: dst [cols, rows] = src [rows, cols]^T for every job (rows, cols %% 64 == 0, 16-byte aligned) -- the
 * fp8 dgrad C = dY W needs W^T K-minor ([in, out] for an nn.Linear weight [out, in]): the e4m3 weight shadows are transposed
 * once per optimizer step.  items: DEVICE array of job << 32 | tile (64 x 64-byte tiles, row-major over the source). */
typedef struct MhTransposeJob { const void* src; void* dst; int rows, cols; } MhTransposeJob;
int mh_transpose_u8_batched(const MhTransposeJob* jobs_device, const unsigned long* items_device, int n_items, void* stream);

/* ---------------------------------------------------------------------------------------------- LayerNorm
 * y = (x - mean) * rstd * gamma + beta over the last dim; x f32 (residual stream), y bf16 (GEMM operand) or f32.
 * Rows are addressed as row(b, j) = b * L + off + j (j < n) on both sides, so the split / concat of group sequences
 * around the joint encoder (maestro/ssl/mim.py:408-423) and the per-modality ungroup (maestro/layers/utils.py:50-100)
 * are addressing, not copies.  Saves mean/rstd (f32 [B*n]).  Replaces nn.LayerNorm inside vit_pytorch
 * Attention.norm, FeedForward.net[0] and Transformer.norm (call sites maestro/ssl/mae.py:135-174).
 * dim % 4 == 0, dim <= 2048. */
int mh_layernorm_fwd(const float* x, int x_L, int x_off, const float* gamma, const float* beta, void* y, int y_L,
                     int y_off, int y_is_f32, float* mean, float* rstd, int B, int n, int dim, float eps, void* stream);
/* The same with a second output for the fp8 path: y8 = OCP e4m3 of (y * *y8_scale) in y's row map (the A operand of the next
 * mh_gemm_fp8), max |y| folded into the amax row y8_amax (optional; delayed scaling).  y is bf16 (the backward's wgrad reads it). */
int mh_layernorm_fwd_fp8(const float* x, int x_L, int x_off, const float* gamma, const float* beta, void* y, int y_L,
                         int y_off, float* mean, float* rstd, int B, int n, int dim, float eps, void* y8,
                         const float* y8_scale, float* y8_amax, void* stream);
/* dx (f32, x's row map) = (dres ? dres : 0) + LN-backward(dy); optional bf16 copy dx_bf16 (operand of the next
 * dgrad/wgrad GEMM).  dgamma/dbeta f32 [dim] are accumulated (+=) through `workspace` (f32,
 * mh_layernorm_bwd_workspace(B*n, dim) floats; per-block partial rows, then a short atomic reduce); dcol f32 [dim]
 * (optional) += column sums of dx = the bias gradient of the Linear whose output fed this residual.  Pass
 * dgamma = dbeta = NULL to skip all three.  dy is bf16 (dy_is_f32 = 0) or f32, addressed with its own row map. */
int mh_layernorm_bwd(const void* dy, int dy_L, int dy_off, int dy_is_f32, const float* x, int x_L, int x_off,
                     const float* gamma, const float* mean, const float* rstd, const float* dres, float* dx,
                     void* dx_bf16, float* dgamma, float* dbeta, float* dcol, float* workspace, int B, int n, int dim,
                     void* stream);
long mh_layernorm_bwd_workspace(int rows, int dim);
/* The same backward with the parameter-gradient reduce left to the caller: only writes the per-block partial rows
 * [mh_layernorm_bwd_workspace / (3 dim)][3 dim] = (dgamma | dbeta | colsum(dx)) into `workspace`; a later
 * mh_colsum_batched over many such workspaces replaces one small reduce launch per LayerNorm. */
int mh_layernorm_bwd_partial(const void* dy, int dy_L, int dy_off, int dy_is_f32, const float* x, int x_L, int x_off,
                             const float* gamma, const float* mean, const float* rstd, const float* dres, float* dx,
                             void* dx_bf16, float* workspace, int B, int n, int dim, void* stream);
/* Batched column sums: for every job, dst[c] += sum over r < rows of src[r * ld + c] (c < cols), all jobs in ONE launch
 * (fp32 atomics: dst must hold the value to add to).  jobs and blocks are device arrays; blocks lists the work items of
 * the launch, one per workgroup: job << 48 | column block (256 columns) << 32 | row chunk (MH_COLSUM_ROWS rows), so that
 * jobs of very different shapes share a dense grid. */
#define MH_COLSUM_ROWS 16
typedef struct {
    const float* src;
    float* dst;
    int rows, cols, ld, reserved;
} MhColsumJob;
int mh_colsum_batched(const MhColsumJob* jobs_device, int n_jobs, const uint64_t* blocks_device, int n_blocks, void* stream);

/* ---------------------------------------------------------------------------------------------- attention
 * Fused softmax(Q K^T * scale) V, no mask, no dropout (vit_pytorch Attention.forward; call sites mae.py:135-174).
 * qkv: bf16 [B, N, 3, H, D] (= to_qkv output, chunk(3) then 'b n (h d) -> b h n d'); out: bf16 [B, N, H*D];
 * lse: f32 [B, H, N] (natural-log sum-exp of the scaled scores, saved for the backward).  D in {32, 64}. */
int mh_attn_fwd(const void* qkv, void* out, float* lse, int B, int N, int H, int D, float scale, void* stream);
/* dqkv bf16 [B,N,3,H,D] from dout bf16 [B,N,H*D]; delta: f32 workspace [B,H,N].  Two kernels: dQ with the query on the MFMA lane
 * (it also computes delta = rowsum(dO * O)), then dK / dV with the key on the lane; S and P are recomputed in both from lse.  The
 * operand that sits in registers (Q resp. K) is multiplied by scale x log2(e) and rounded to bf16 once more, so the recomputed
 * probabilities differ from the forward's by one extra bf16 rounding of that operand (within the kernels' 1.5e-2 parity bound). */
int mh_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv,
                int B, int N, int H, int D, float scale, void* stream);
/* ---------------------------------------------------------------------------------------------- patch embed
 * Patch extraction of one modality (maestro/layers/embed.py:57-60 'b d c (h p1) (w p2)' + the loss target of
 * maestro/train/model.py:211-229): img f32 [BD, Ctot, S, S] ->
 *   cols   bf16 [BD*g*g, Kpad]  im2col rows in (c, p1, p2) order (= Conv2d weight flattening), zero padded;
 *   target f32  [BD*g*g, P*P*Ctot] columns (p1*P+p2)*Ctot + c (optional): patch-group-wise normalised per
 *          norm_bands group (device int array; unbiased variance, eps 1e-6) when normalise != 0, else raw pixels.
 * rescale_elev (maestro/ssl/mim.py:433-436): channels >= 1 become 30*(ch0 - ch) on the fly. */
int mh_patchify(const float* img, void* cols, float* target, int BD, int Ctot, int S, int P, int Kpad,
                const int* norm_bands, int n_norm_groups, int normalise, int rescale_elev, void* stream);
/* The same for ONE band-group of a modality with several (Patchify splits the channel axis by group sizes and gives every
 * group its own PatchifyBands conv, maestro/layers/embed.py:18-34): img f32 [BD, Csrc, S, S], the patch takes channels
 * c0 .. c0 + Ctot - 1; rescale_elev refers to channel 0 of the image.  cols or target may be NULL (the loss target of such a
 * modality is built once over ALL its channels -- the norm_bands groups of maestro/train/model.py:219-229 ignore the band-
 * groups -- with cols == NULL, c0 = 0, Ctot = Csrc). */
int mh_patchify_bands(const float* img, void* cols, float* target, int BD, int Csrc, int c0, int Ctot, int S, int P, int Kpad,
                      const int* norm_bands, int n_norm_groups, int normalise, int rescale_elev, void* stream);

/* GroupNorm(1, E) over the whole (tokens x E) image per (b, d) (embed.py:55,59-61): chunked partial sums ->
 * stats f32 [BD, 2] = (mean, rstd).  partial: f32 workspace of mh_groupnorm_partial_size() floats. */
int mh_groupnorm_stats(const float* y, float* partial, float* stats, int BD, int L, int E, float eps, void* stream);
int mh_groupnorm_partial_size(int BD, int L, int E);
/* normalise + per-channel affine + positional + date encodings (mim.py:232-252, utils.py:103-173), written straight
 * into the group sequence:  xg[b, tok_off + d*L + l, :] = (y - mean)*rstd*gamma + beta + pos[l, :] (+ date[b*D+d, :]
 * on the last 8 channels).  y: f32 [B*D*L, E] conv output incl. bias; pos f32 [L, E]; date f32 [B, date_rows, 8]
 * (row date_off + d belongs to this modality's date d) or NULL. */
int mh_embed_finish(const float* y, const float* stats, const float* gamma, const float* beta, const float* pos,
                    const float* date, int date_rows, int date_off, float* xg, int B, int D, int L, int E, int tok_off,
                    int Lgroup, void* stream);
/* Backward of the above w.r.t. the conv output: dyc bf16 [B*D*L, E] (A operand of the patch-embed wgrad GEMM);
 * dgamma/dbeta atomically accumulated; sums: f32 workspace [B*D, 2]. */
int mh_embed_finish_bwd(const float* dxg, const float* y, const float* stats, const float* gamma, void* dyc,
                        float* dgamma, float* dbeta, float* sums, int B, int D, int L, int E, int tok_off, int Lgroup,
                        void* stream);
/* Date features (maestro/layers/utils.py:128-167): dates int16 [B, D, 3] (year, day-of-year, hour), ref_date int16
 * [B, 1, 3] -> out f32 [B, rows, 8] rows [row_off, row_off + D): fac * [diff x4, sin/cos doy, sin/cos hour]. */
int mh_date_features(const int16_t* dates, const int16_t* ref_date, float* out, int B, int D, int rows, int row_off,
                     float fac, void* stream);
/* Input staging: resize rasters to image_size (maestro/ssl/mim.py:427-432, F.interpolate with align_corners=False).
 * in f32 [planes, Hin, Win] -> out f32 [planes, Hout, Wout]; mode 0 nearest, 1 bilinear, 2 bicubic (PyTorch index maps,
 * cubic convolution with A = -0.75). */
int mh_resize(const float* in, float* out, long planes, int Hin, int Win, int Hout, int Wout, int mode, void* stream);
/* Input staging: per-sample flips / transpose of square rasters (maestro/dataset/dataset.py:224-257: np.flip axis 2, np.flip
 * axis 3, np.swapaxes(2, 3) of the per-sample [D, C, H, W] array, applied in that order).  in/out [B, planes, S, S] of
 * elem_bytes-wide elements (1, 2, 4, 8), out of place; flags[b] bit0 = flip rows, bit1 = flip columns, bit2 = transpose.
 * Pure data movement: bit-exact. */
int mh_dihedral(const void* in, void* out, const uint8_t* flags, int B, long planes, int S, int elem_bytes, void* stream);
/* Elevation rescale copy (maestro/ssl/mim.py:433-436): out[:, c>=1] = 30 * (img[:, 0] - img[:, c]); out != img. */
int mh_rescale_elev(const float* img, float* out, int BD, int C, int S, void* stream);
/* Patch layout [BD*g*g, P*P*C] -> image [BD, C, S, S] ('(p1 p2 c) h w -> c (h p1) (w p2)', embed.py:153-160). */
int mh_depatchify(const float* patches, float* img, int BD, int C, int S, int P, void* stream);

/* ---------------------------------------------------------------------------------------------- masking
 * Random-mask token selection with STABLE tie order (maestro/ssl/mae.py:236-259; ties: DESIGN.md): noise f32 [B, L],
 * struct_mask u8 [B, L] or NULL (noise *= 1 - struct), k masked per row.  Outputs: visible_idx int32 [B, L-k]
 * ascending, masked_idx int32 [B, k] ascending, inv int32 [B, L] (position in the visible list or -1),
 * mask u8 [B, L] (1 = masked).  L <= 8192. */
int mh_mask_select(const float* noise, const uint8_t* struct_mask, int* visible_idx, int* masked_idx, int* inv,
                   uint8_t* mask, int B, int L, int k, void* stream);
/* Row gather  dst[b, dst_off + j, :] = src[b, idx[b, j], :]  (f32 rows of `dim`); dst rows live inside a sequence of
 * dst_L rows per sample (x[batch, unmasked_indices], mae.py:261, fused with the joint-encoder concat). */
int mh_gather_rows(const float* src, const int* idx, float* dst, int B, int src_L, int n_idx, int dim, int dst_L,
                   int dst_off, void* stream);
/* Transpose of mh_gather_rows into a ZEROED dsrc (indices unique per sample -> plain stores). */
int mh_scatter_rows(const float* ddst, const int* idx, float* dsrc, int B, int src_L, int n_idx, int dim, int dst_L,
                    int dst_off, void* stream);
/* The same transpose written as a gather over every destination row: dst[b, t, :] = inv[b, t] >= 0 ? src[b, inv[b, t], :] : 0
 * (inv = mh_mask_select's position map, -1 on masked tokens; src f32 [B, n, dim], dst f32 [B, L, dim]).  Backward of the
 * visible-token gather x[batch, unmasked_indices] (mae.py:261) without a memset of the destination. */
int mh_expand_rows(const float* src, const int* inv, float* dst, int B, int L, int n, int dim, void* stream);
/* Decoder input assembly (mae.py:266-287 + mim.py:254-274):
 *   xdec[b,t,:] = (inv[b,t] < 0 ? mask_token[tok_slot[t]] : y[b, inv[b,t], :]) + pos[t,:] + date[b, date_row[t], :8 tail]
 * y f32 [B, n_vis, Dd]; mask_token f32 [slots, Dd]; pos f32 [L, Dd]; date f32 [B, n_date_rows, 8] or NULL. */
int mh_unmask_assemble(const float* y, const int* inv, const float* mask_token, const int* tok_slot, const float* pos,
                       const float* date, const int* date_row, int n_date_rows, float* xdec, int B, int L, int n_vis,
                       int Dd, void* stream);
/* dmask_token[:] (f32 [Dd], the gradient row of ONE mask token) += sum of dxdec rows of masked tokens t in
 * [t_lo, t_hi) whose tok_slot == slot (atomic). */
int mh_unmask_token_grad(const float* dxdec, const uint8_t* mask, const int* tok_slot, float* dmask_token, int B, int L,
                         int Dd, int slot, int t_lo, int t_hi, void* stream);
/* The same two with a PER-SAMPLE slot map tok_slot_bl int32 [B, L]: masked position t of sample b receives mask token
 * tok_slot_bl[b, t].  Serves the reference's implementation-defined tie order (maestro/ssl/mae.py:274-286:
 * mask_rec.float().argsort(descending=True) orders the masked positions of a sample arbitrarily, so in a group of several
 * modalities a position can receive ANOTHER modality's token); the host builds the map from the same torch call
 * (MAEEngine.tie_order = "torch").  The gradient sums over all masked positions of the group whose map entry == slot. */
int mh_unmask_assemble_per_sample(const float* y, const int* inv, const float* mask_token, const int* tok_slot_bl,
                                  const float* pos, const float* date, const int* date_row, int n_date_rows, float* xdec,
                                  int B, int L, int n_vis, int Dd, void* stream);
int mh_unmask_token_grad_per_sample(const float* dxdec, const uint8_t* mask, const int* tok_slot_bl, float* dmask_token, int B,
                                    int L, int Dd, int slot, void* stream);
/* out[0] = number of masked tokens of all samples in group positions [t_lo, t_hi) (one modality). */
int mh_count_masked(const uint8_t* mask, int B, int L, int t_lo, int t_hi, int* out, void* stream);
/* out[0] = (accumulate ? out[0] : 0) + mult * that count: the masked ELEMENTS of a modality whose band-groups have different
 * patch sizes in elements (mult = P*P*bands of the group); the denominator of mh_masked_loss_bands. */
int mh_count_masked_elems(const uint8_t* mask, int B, int L, int t_lo, int t_hi, int* out, int mult, int accumulate, void* stream);

/* ---------------------------------------------------------------------------------------------- loss
 * Masked reconstruction loss at patch layout (maestro/train/model.py:195-247) for one modality: rec f32 [B*Lm, PPC]
 * (pixelify GEMM output), target f32 [B*Lm, PPC] (mh_patchify), mask_group u8 [B, Lgroup] (token (b,t) of the modality
 * sits at tok_off + t).  p = 1 (l1*) or 2 (l2*).  With coef = weight / (n_masked[0] * PPC):
 *   acc[0] += coef * sum(e over masked elements)   (weight = w_mod / sum_w  ->  acc accumulates the final loss)
 *   drec (bf16 [B*Lm, PPC], optional) = coef * d e / d rec on masked tokens, 0 elsewhere. */
int mh_masked_loss(const float* rec, const float* target, const uint8_t* mask_group, const int* n_masked, float weight,
                   float* acc, void* drec, int B, int Lm, int Lgroup, int tok_off, int PPC, int p, void* stream);
/* One band-group of a modality with several (Pixelify concatenates the groups' reconstructions along the channel axis and
 * the loss is ONE mean over the modality's masked elements, maestro/layers/embed.py:99-127, maestro/train/model.py:241-243):
 * rec f32 [B*Lm, PPC] with PPC = P*P*n_g, columns pixel * n_g + c; target = the MODALITY's rows [B*Lm, P*P*tgt_C], the group
 * reads columns pixel * tgt_C + tgt_c0 + c; coef = weight / n_elems[0] (mh_count_masked_elems over all groups). */
int mh_masked_loss_bands(const float* rec, const float* target, const uint8_t* mask_group, const int* n_elems, float weight,
                         float* acc, void* drec, int B, int Lm, int Lgroup, int tok_off, int PPC, int p, int tgt_C, int tgt_c0,
                         int n_g, void* stream);

/* ---------------------------------------------------------------------------------------------- probe / finetune heads
 * (maestro/ssl/mim.py:343-394 compute_logits, maestro/layers/head.py, maestro/train/base.py:98-151)
 *
 * Bilinear resize of a channel-last token grid onto the reference grid (mim.py:357-366, F.interpolate bilinear,
 * align_corners=False): rows (b, in_off + d*h*h + y*h + x) of in f32 [B, in_rows, E] -> rows (b, out_off + d*H*H + Y*H + X)
 * of out f32 [B, out_rows, E].  h == H is an exact copy.  The backward is the transposed map written as a gather
 * (deterministic); accumulate = 1 adds to din. */
int mh_token_resize(const float* in, long in_rows, int in_off, float* out, long out_rows, int out_off, int B, int D, int h,
                    int H, int E, void* stream);
int mh_token_resize_bwd(const float* dout, long out_rows, int out_off, float* din, long in_rows, int in_off, int B, int D,
                        int h, int H, int E, int accumulate, void* stream);
/* AttentiveReduce (head.py:28-62) after its LayerNorm + to_kv GEMM: kv bf16 [rows, 2*dim] (k | v); sequence (b, l),
 * b < n_batch, l < Lr, runs over t < T at row (b*T + t)*Lr + l  (PixelifyHead: T = dates, Lr = reference-grid tokens;
 * ClassificationHead: T = all tokens, Lr = 1).  out f32 [n_batch*Lr, dim] = softmax_t(scale * q.k_t) . v per head
 * (heads = 8, dim in {192, 384, 768, 1024}), lse f32 [n_batch*Lr, 8] kept for the backward.
 * Backward: dkv bf16 [rows, 2*dim]; dq_partial f32 [mh_attn_reduce_partial_rows(n_batch*Lr), dim], summed with mh_colsum. */
int mh_attn_reduce_fwd(const void* kv, const float* query, float* out, float* lse, int n_batch, int T, int Lr, int dim,
                       int heads, void* stream);
int mh_attn_reduce_bwd(const void* kv, const float* query, const float* out, const float* lse, const float* dout, void* dkv,
                       float* dq_partial, int n_batch, int T, int Lr, int dim, int heads, void* stream);
long mh_attn_reduce_partial_rows(int n_seq);
/* type_head = "linear": mean over the same token axis (head.py:77-78, 111-112); x f32 [rows, dim] -> out f32 [n_batch*Lr, dim]. */
int mh_mean_reduce_fwd(const float* x, float* out, int n_batch, int T, int Lr, int dim, void* stream);
int mh_mean_reduce_bwd(const float* dout, float* dx, int n_batch, int T, int Lr, int dim, void* stream);
/* ClassificationHead.linear (head.py:83,93), fp32: out[b, c] = bias[c] + x[b, :] . W[c, :]; backward ACCUMULATES dW, db
 * and writes dx (optional). */
int mh_head_linear_fwd(const float* x, const float* W, const float* bias, float* out, int B, int C, int E, void* stream);
int mh_head_linear_bwd(const float* x, const float* W, const float* dout, float* dx, float* dW, float* db, int B, int C, int E,
                       void* stream);
/* loss_pred (base.py:98-151).  count[0] += #targets != missing_val (target_bytes-wide signed integers).
 * mh_ce_loss: target raster [B, S, S] (S = g*P; classification: g = P = 1), logits f32 at PATCH layout [B*g*g, ld >= P*P*C]
 * with columns (p1*P + p2)*C + c (PixelifyBands order, embed.py:153-160); acc[0] += mean over valid pixels of cross entropy,
 * dlogits (bf16 or f32, same layout) = (softmax - onehot) / n_valid, 0 on missing pixels; n_valid = 0 -> loss 0.
 * mh_bce_loss: logits / target f32 [B, C]; rows with any target == missing_val are skipped (base.py:122-123);
 * acc[0] += mean BCE-with-logits, dlogits f32 [B, C]. */
int mh_count_valid(const void* target, int target_bytes, long n, long missing_val, int* count, void* stream);
int mh_ce_loss(const float* logits, const void* target, int target_bytes, long missing_val, const int* n_valid, float* acc,
               void* dlogits, int dlogits_is_f32, int B, int g, int P, int C, int ld, void* stream);
int mh_bce_loss(const float* logits, const float* target, float missing_val, float* acc, float* dlogits, int B, int C,
                void* stream);

/* ---------------------------------------------------------------------------------------------- misc
 * column sums: out[n] += sum_m x[m, n]  (bias gradients); x bf16 or f32; atomically accumulated. */
int mh_colsum(const void* x, int x_is_f32, float* out, int M, int N, int ld, void* stream);
/* Zero n_spans (offset, length) float ranges of one buffer (spans: device array of 2*n_spans longs; max_len = longest span).
 * Used to clear only the atomically accumulated gradient slots when the weight gradients are stored by the grouped GEMM. */
int mh_zero_spans(float* base, const long* spans_device, int n_spans, long max_len, void* stream);
/* x[0..n) *= *scale with the scalar read on the device; a no-op when it equals 1 (the autograd bridge of the Lightning
 * surface, maestro/train/base.py:242-247: loss.backward() hands over d loss as a device tensor). */
int mh_scale_dev(float* x, long n, const float* scale, void* stream);
/* f32 -> bf16 cast of a flat buffer (weight shadow copies). */
int mh_cast_bf16(const float* src, void* dst, long n, void* stream);
/* f32 [E, K] -> bf16 [E, Kpad] (zero padded rows: patch-embed conv weight [E, C*P*P]) and the transpose for grads:
 * dst f32 [E, K] += src f32 [E, Kpad][:, :K]. */
int mh_pack_rows_bf16(const float* w, void* dst, int E, int K, int Kpad, void* stream);
int mh_unpack_rows_add(const float* src, float* dst, int E, int K, int Kpad, void* stream);
/* Fused AdamW over a flat parameter buffer (torch.optim.AdamW semantics, maestro/train/model.py:135-140): decoupled
 * weight decay, bias correction, grads pre-multiplied by grad_scale; refreshes the bf16 shadow copy (optional).
 * n % 4 == 0, step >= 1. */
int mh_adamw(float* p, const float* g, float* m, float* v, void* p_bf16, long n, float lr, float b1, float b2,
             float eps, float wd, int step, float grad_scale, void* stream);
/* The same update that also refreshes the OCP e4m3 weight shadows of the fp8 path: p_fp8 = flat uint8 buffer with the
 * parameters' offsets (same n), slot_map[i / 64] = scale slot of element i's weight or -1 (short; parameters are 64-element
 * aligned), cast with scale[slot] (derived from the previous step's absmax by mh_fp8_update_scales), |new value| folded into
 * amax row `slot`.  The slices passed must start at a multiple of 64 elements of the flat buffer the map was built for. */
int mh_adamw_fp8(float* p, const float* g, float* m, float* v, void* p_bf16, void* p_fp8, const short* slot_map,
                 const float* scale, float* amax, long n, float lr, float b1, float b2, float eps, float wd, int step,
                 float grad_scale, void* stream);
/* The same update with the per-step scalars read from DEVICE memory, so that the launch can sit inside a captured hipGraph
 * (the optimizer step overlapped with the next forward): hyper = f32 [5] = {lr, 1 - b1^step, sqrt(1 - b2^step), grad_scale,
 * active}; active == 0 makes the launch a no-op (no update pending). */
int mh_adamw_dev(float* p, const float* g, float* m, float* v, void* p_bf16, long n, float b1, float b2, float eps,
                 float wd, const float* hyper, void* stream);

#ifdef __cplusplus
}
#endif
#endif
