/* libkokoro_hip.so — C ABI of the MI355X (gfx950) kernels for the Kokoro acoustic-model train step.
 *
 * The reference (igorshmukler/kokoro-ruslan) is pure PyTorch and has no FFI; every entry point
 * below therefore replaces an ATen call site of the reference's hot path (cited per function as
 * `file:line` relative to /root/reference/src/kokoro).  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer is a DEVICE pointer owned by the caller and must
 *    stay alive until `stream` has passed the call; kernels never allocate and never synchronise;
 *  - every function enqueues on `stream` (a hipStream_t passed as void*) and returns 0 on success,
 *    a negative KK_E* code or a positive hipError_t otherwise; `kk_last_error()` holds the message;
 *  - all float tensors are fp32 row-major; `math` selects the MFMA arithmetic of matmul-class
 *    kernels: KK_MATH_F32 (v_mfma_f32_32x32x2_f32, exact fp32 — the parity mode) or KK_MATH_BF16
 *    (operands rounded to bf16 at LDS staging, v_mfma_f32_32x32x16_bf16, fp32 accumulate);
 *  - parameter-gradient outputs ACCUMULATE (+=) into their destination: the caller zeroes the
 *    gradient arena once per accumulation cycle (reference: optimizer.zero_grad, trainer.py:2258).
 */
#ifndef KOKORO_HIP_H
#define KOKORO_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define KK_ABI_VERSION 2 /* 2 (round 6): kk_seg_sumsq takes a record workspace instead of `zeroed`; p_sumsq of kk_adamw_ema /
                          kk_weight_norm_project is an int64 Q34.30 sum; round 5 had already added parameters to kk_losses_fwd,
                          kk_losses_finalize, kk_opt_prepare, kk_adamw_ema, kk_rowdot_bwd and fields to KkOptCfg under version 1 */
#define KK_MATH_F32 0
#define KK_MATH_BF16 1
#define KK_EINVAL (-22)
#define KK_ENOTSUP (-95)
#define KK_SEG_ALIGN 1024 /* arena segments start on multiples of this many elements */

int kk_abi_version(void);
const char *kk_last_error(void);
/* Which kernel variant the LAST launching call of this thread took, e.g. "g16x<0,1,128,128,3,0,2,2,4>", "gemm16_w8<0,1,3>",
 * "attn_fwd3_q128", "attn_bwd_pair3" ("" before the first one).  The tile / generation policies are the library's own
 * (bytes through the busiest CU, sequence lengths, alignment); tests assert through this record that the shapes they mean to
 * cover really reach the kernel they name, so that a policy change cannot silently un-test a kernel. */
const char *kk_last_kernel(void);

/* ---- GEMM family (nn.Linear fwd/dgrad/wgrad: transformers.py:131-136,90-91; model.py:173,190) ----
 * C[M,N] = alpha * op(A)·op(B) (+bias[n]) (+residual[(m % res_mod), n]) (+ beta*C).
 * ta=0: A stored [M,K] (lda, K contiguous); ta=1: A stored [K,M] (M contiguous).
 * tb=0: B stored [N,K] (nn.Linear weight layout);  tb=1: B stored [K,N].
 * dtypes (KK_MATH_BF16 only): bit0 = A is stored as bf16, bit1 = B is bf16, bit2 = C is written as bf16
 * (pointers are then bf16 arrays passed through the float* parameters; lda/ldb/ldc stay in elements).
 * split_k>1 partitions K over blockIdx.y and accumulates with fp32 atomics (requires beta==1 or a
 * contiguous C that the call zero-fills when beta==0; bias/residual are added by slice 0). */
int kk_gemm(int ta, int tb, int64_t M, int64_t N, int64_t K, float alpha, const float *A, int64_t lda,
            const float *B, int64_t ldb, float beta, float *C, int64_t ldc, const float *bias,
            const float *residual, int64_t ldr, int64_t res_mod, int split_k, int math, int dtypes,
            void *stream);
/* Weight gradients of several nn.Linear layers in one launch (what autograd computes one by one for w_q/w_k/w_v/w_o,
 * transformers.py:131-136, and linear1/linear2, transformers.py:90-91): for i < n,
 *   dw_i[M_i, N_i] += dy_i[T_i, M_i]^T . x_i[T_i, N_i]      (bf16 dy / x, fp32 dw, like kk_gemm(ta=1, tb=1, beta=1)).
 * A layer's weight gradients have no consumer before the optimizer, so the engine queues them through the layer's
 * backward and issues them together (n <= 8): full-length reductions, no split-K atomics, one launch.  `descs` is a
 * HOST array read during the call.  The operands must stay untouched until the launch has run. */
typedef struct {
    const void *dy; int64_t lddy;
    const void *x;  int64_t ldx;
    float *dw;      int64_t lddw;
    int64_t M, N, T;
} KkWgradDesc;
int kk_gemm_wgrad_group(const KkWgradDesc *descs, int n, int split_k /* k-slices per problem, 0 = automatic */,
                        int overwrite /* 1: dw = product (the first micro-batch of an accumulation cycle: dw is not read; no k-slices), 0: dw += */,
                        void *ss_rec /* nullable; kk_seg_sumsq's record workspace */, const int32_t *ss_seg /* HOST, 2 per problem: arena segment id of dw's row 0, and the rows per
                        segment when dw spans several ADJACENT segments (the fused q|k|v view; a multiple of 128), else 0 */,
                        int32_t *ss_count /* HOST in/out: records of ss_rec in use.  When the launch writes every dw element exactly once from
                        one workgroup (no k-slices, write-through fp32 epilogue) each tile also leaves the sum of squares of what it stored as
                        a record [*ss_count ..) and *ss_count advances by the launch's tile count — kk_seg_sumsq(seg_skip, extra_records) then
                        does not read those tensors; otherwise *ss_count is left alone */, void *stream);
/* Attention projections with the per-head norm as the epilogue: raw[T, parts*heads*64] = x[T,K] . W^T (+bias), saved for
 * the backward, and y = per-head RMSNorm(64)(raw) * gains[part] (+ RoPE on the parts set in rope_mask, position = row % S)
 * from one launch — a 64x64 output tile is exactly 64 (row, head) vectors.  parts <= 12 column groups of heads*64 (q|k|v of
 * a fused projection, or k|v of every decoder layer's cross-attention), `gains` a HOST array of `parts` device pointers.
 * bf16 operands and outputs; same bits as kk_gemm + kk_headnorm_rope_fwd (transformers.py:246-277). */
int kk_gemm_qkv_headnorm(int64_t T, int parts, int heads, int64_t K, const void *x, int64_t ldx, const void *W,
                         const float *bias, void *raw, int64_t ldraw, void *y, int64_t ldy, int S,
                         const float *const *gains, int rope_mask, const float *cos_t, const float *sin_t, void *stream);
/* GLU feed-forward forward (GLUFeedForward.forward, transformers.py:105-108), fused: h1[T,2F] = x[T,K] . W[2F,K]^T + bias (saved for the backward, bf16) and the gated
 * product g[T,F] = gelu(h1[:, :F]) * h1[:, F:] * dropout mask (seed, site, p as in kk_glu_fwd) from one launch: every
 * workgroup owns a column block of BOTH halves.  bf16 operands and outputs.  Replaces kk_gemm + kk_glu_fwd. */
int kk_gemm_linear_glu(int64_t T, int64_t F, int64_t K, const void *x, int64_t ldx, const void *W, const float *bias,
                       void *h1, void *g, int64_t ldg, const uint32_t *seed, uint32_t site, float p, void *stream);
/* dX[M,N] = dY[M,K].W[K,N] of an attention OUTPUT projection (w_o backward, transformers.py:398 `self.w_o(context)`), bf16
 * operands and result, with the attention backward's row term as the epilogue: a 128x64 tile's columns are one head, so
 * Delta[b, head, q] = sum_d dX[b*S+q, 64*head+d] * O[b*S+q, 64*head+d] (fp32, [B, heads, S]) leaves with the tile.  Replaces
 * kk_gemm (dgrad) + the Delta pass inside kk_attn_bwd_dq / kk_attn_delta.  Only shapes that take the eight-wave 128x64 tile
 * (ask _supported). */
int kk_gemm_dgrad_delta_supported(int64_t M, int64_t N, int64_t K);
int kk_gemm_dgrad_delta(int64_t M, int64_t N, int64_t K, const void *dy, int64_t lddy, const void *W, int64_t ldw,
                        void *dx, int64_t lddx, const void *O, int64_t ldo, float *delta, int S, int heads, void *stream);
/* GLU feed-forward backward (autograd of transformers.py:105-108), fused: dG = dy[T,H] . W[H,F] (the linear2 dgrad; bf16 operands, W row-major [H,F]) with the
 * gate's backward as the epilogue — dh1[T,2F] is written directly from h1[T,2F] = [a | b] saved by the forward and the
 * gate's dropout mask (seed, site, p as in kk_glu_fwd); the column sums of dh1 (linear1's bias gradient) go to
 * partials[kk_gemm_dgrad_glu_blocks(T)][2F] for kk_partials_reduce.  Replaces kk_gemm + kk_glu_bwd + kk_colsum_acc. */
int kk_gemm_dgrad_glu_blocks(int64_t T);
int kk_gemm_dgrad_glu(int64_t T, int64_t F, int64_t H, const void *dy, int64_t lddy, const void *W, const void *h1,
                      void *dh1, float *partials, const uint32_t *seed, uint32_t site, float p, void *stream);
/* out[n] += sum_m X[m,n]  (bias gradients). */
int kk_colsum_acc(const float *X, int64_t ldx, int64_t M, int64_t N, float *out, int x_bf16, void *stream);

/* ---- activation storage ----
 * Tensors are fp32 unless a trailing `*_bf16` flag of the entry point says otherwise.  A non-zero flag means the
 * named operands (x / y / io = every activation operand of the call) are bf16 in HBM (2 bytes per element, same
 * strides counted in elements); statistics, gains, parameter gradients and index tensors stay fp32 / integer.
 * The fp32 parity mode passes 0 everywhere; the bf16 mode stores GEMM and attention operands as bf16 so that the
 * MFMA kernels load their fragments without conversion and every activation round trip moves half the bytes. */

/* ---- attention (F.scaled_dot_product_attention, transformers.py:393-398; masks :299-316) ----
 * Token-major operands: element (b, s, head, d) of X lives at X[(b*S + s)*ldx + head*64 + d]; head_dim is 64.
 * key_mask: optional uint8 [B,Sk], non-zero = key masked (−inf); causal: key > query masked.
 * LSE [B,heads,Sq] = log-sum-exp of the scaled scores (saved for backward).
 * Dropout on the probabilities (p_drop > 0): the keep mask is a pure function of (*seed, site, b, head, q, key), so
 * the two backward kernels regenerate it — pass them the same seed pointer, site and p_drop. */
int kk_attn_fwd(const float *Q, const float *K, const float *V, float *O, float *LSE, int B, int heads,
                int Sq, int Sk, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                const uint8_t *key_mask, int causal, float scale, const uint32_t *seed, uint32_t site,
                float p_drop, int math, int io_bf16, void *stream);
/* Delta[b,head,q] = sum_d dO·O (first step of the backward). */
int kk_attn_delta(const float *O, const float *dO, float *Delta, int B, int heads, int Sq, int64_t ldo,
                  int64_t lddo, int io_bf16, void *stream);
/* Optional epilogue of the two backward kernels (hn != NULL): the attention operand was y = RMSNorm64(raw)*gain (+ RoPE,
 * position = row within the sequence), and the kernel writes the gradient of RAW (into dQ, or dK / dV) instead of the
 * gradient of y, plus one row of 64 partial gain-gradient sums per workgroup into partials[kk_attn_bwd_blocks()][64]
 * (to be added up by kk_partials_reduce).  Replaces kk_headnorm_rope_bwd.  kk_attn_bwd_dq takes one descriptor (Q),
 * kk_attn_bwd_dkv an array of two (K, V). */
typedef struct {
    const void *raw; int64_t ldraw;     /* the projection output the norm was applied to, [rows, >= heads*64] */
    const float *gain;                  /* [64] */
    float *partials;                    /* [kk_attn_bwd_blocks(B, heads, S)][64] */
    const float *cos_t, *sin_t;         /* [S, 64] rotate-half RoPE tables (rope != 0): columns d and d + 32 are identical, as
                                           positional_encoding.py:129-150 builds them (emb = cat(freqs, freqs)); the bf16-storage
                                           kernels stage columns 0..31 only */
    int rope;
} KkAttnHeadNorm;
int kk_attn_bwd_blocks(int B, int heads, int S);   /* S = Sq for kk_attn_bwd_dq, Sk for kk_attn_bwd_dkv */
/* O == NULL: Delta is read.  O != NULL: Delta is computed here from dO and O (row stride ldo) and WRITTEN, so the
 * separate kk_attn_delta launch is not needed; kk_attn_bwd_dkv (launched after) reads it. */
int kk_attn_bwd_dq(const float *Q, const float *K, const float *V, const float *dO, const float *LSE,
                   float *Delta, float *dQ, int B, int heads, int Sq, int Sk, int64_t ldq, int64_t ldk,
                   int64_t ldv, int64_t lddo, int64_t lddq, const uint8_t *key_mask, int causal, float scale,
                   const uint32_t *seed, uint32_t site, float p_drop, int math, int io_bf16, const float *O,
                   int64_t ldo, const KkAttnHeadNorm *hn, void *stream);
int kk_attn_bwd_dkv(const float *Q, const float *K, const float *V, const float *dO, const float *LSE,
                    const float *Delta, float *dK, float *dV, int B, int heads, int Sq, int Sk, int64_t ldq,
                    int64_t ldk, int64_t ldv, int64_t lddo, int64_t lddk, int64_t lddv,
                    const uint8_t *key_mask, int causal, float scale, const uint32_t *seed, uint32_t site,
                    float p_drop, int math, int io_bf16, const KkAttnHeadNorm *hn, void *stream);
/* dQ, dK and dV of one attention in ONE launch (bf16 storage, two key groups: the second-generation kernels of the two
 * calls above as the z = 0 / z = 1 halves of one grid; any other case runs them as two launches, in order).  Delta is an
 * INPUT (kk_gemm_dgrad_delta writes it with dO; or kk_attn_delta), so the halves are independent: on a causal launch the
 * CUs that finish a short dQ block pick up the long dK/dV blocks.  hn_q: one descriptor, hn_kv: two (both or neither).
 * Same call sites as the two calls above (transformers.py:393-398 backward). */
int kk_attn_bwd(const float *Q, const float *K, const float *V, const float *dO, const float *LSE, const float *Delta,
                float *dQ, float *dK, float *dV, int B, int heads, int Sq, int Sk, int64_t ldq, int64_t ldk,
                int64_t ldv, int64_t lddo, int64_t lddq, int64_t lddk, int64_t lddv, const uint8_t *key_mask,
                int causal, float scale, const uint32_t *seed, uint32_t site, float p_drop, int math, int io_bf16,
                const KkAttnHeadNorm *hn_q, const KkAttnHeadNorm *hn_kv, void *stream);
/* Stored dropout keep decisions (round 5).  The probabilities' dropout mask (SDPA dropout_p, transformers.py:393-398) is a counter hash
 * that every kernel can regenerate; in the third-generation backward the regeneration was ~40 % of the vector instructions.
 * kk_attn_fwd_kb = kk_attn_fwd that ALSO stores the keep decisions as packed bits (one per score: the 16 lane masks of every 32 x 32
 * unit the kernel computes, 128 bytes per unit at (((b * heads + head) * ceil(Sq / 32) + qu) * ceil(Sk / 32) + ku) * 128);
 * kk_attn_bwd_kb = kk_attn_bwd whose pair launch reads them instead of hashing (bit-identical gradients).  `keep` = a 16-byte aligned
 * device buffer of kk_attn_keep_bytes(B, heads, Sq, Sk) bytes (0 = this shape's forward stores none: pass NULL); same seed value,
 * site, p_drop, shape and masks in both calls.  NULL keep = exactly kk_attn_fwd / kk_attn_bwd. */
int64_t kk_attn_keep_bytes(int B, int heads, int Sq, int Sk);
int kk_attn_fwd_kb(const float *Q, const float *K, const float *V, float *O, float *LSE, int B, int heads, int Sq, int Sk,
                   int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, const uint8_t *key_mask, int causal, float scale,
                   const uint32_t *seed, uint32_t site, float p_drop, int math, int io_bf16, void *keep, void *stream);
int kk_attn_bwd_kb(const float *Q, const float *K, const float *V, const float *dO, const float *LSE, const float *Delta,
                   float *dQ, float *dK, float *dV, int B, int heads, int Sq, int Sk, int64_t ldq, int64_t ldk, int64_t ldv,
                   int64_t lddo, int64_t lddq, int64_t lddk, int64_t lddv, const uint8_t *key_mask, int causal, float scale,
                   const uint32_t *seed, uint32_t site, float p_drop, int math, int io_bf16, const KkAttnHeadNorm *hn_q,
                   const KkAttnHeadNorm *hn_kv, const void *keep, void *stream);
/* Keep bits from a launch of their own (round 6).  kk_attn_keep_gen fills the keep-bit arrays of up to 16 attention launches — the
 * same function of (seed value, site, b, head, q, key), the same layout as kk_attn_fwd_kb stores — in ONE pure-vector launch that needs
 * no LDS, so it runs beside the persistent encoder forward; kk_attn_fwd_rb is kk_attn_fwd reading those bits instead of hashing (and
 * storing) them: the same output bits with ~40 % fewer vector instructions per score unit.  kk_attn_bwd_kb reads the same arrays. */
/* Weight warming (round 6): the next kk_attn_fwd / _kb / _rb launch of the calling thread that takes the third-generation kernel also
 * touches one dword per 128-byte line of up to two read-only device buffers (the weight matrices of the GEMMs behind it) so that
 * every XCD's L2 holds them when those GEMMs start (an XCD's L2 keeps read-only lines across a kernel boundary).  One-shot. */
int kk_attn_warm_next(const void *w0, int64_t bytes0, const void *w1, int64_t bytes1);
typedef struct {
    void *keep;          /* kk_attn_keep_bytes(B, heads, Sq, Sk) bytes */
    uint32_t site;       /* the site the attention launch is given (engine: sub-layer site + 3) */
    float p;             /* its dropout probability, in (0, 1) */
    int B, heads, Sq, Sk, causal;
} KkKeepSite;
int kk_attn_keep_gen(const KkKeepSite *sites, int n, const uint32_t *seed, int seed_offset /* the bits of seed value *seed + seed_offset: 1 = the
                     next micro-batch's, generated a step ahead beside the optimizer pass */,
                     int max_workgroups /* grid cap (0 = 2048): how many wave slots the launch may take from what runs beside it */, void *stream);
int kk_attn_fwd_rb(const float *Q, const float *K, const float *V, float *O, float *LSE, int B, int heads,
                   int Sq, int Sk, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, const uint8_t *key_mask,
                   int causal, float scale, const uint32_t *seed, uint32_t site, float p_drop, int math, int io_bf16,
                   const void *keep, void *stream);
/* The same backward in two passes through a caller-owned workspace (same call sites; `ws` of at least kk_attn_bwd_ws_bytes(...)
 * bytes, 16-byte aligned, private to the stream for the duration of the call): the dK/dV kernel also stores dS = P o (dP - Delta)
 * as bf16 tiles (2 bytes per score), and dQ = dS . K is a pass without softmax work (+ the head-norm epilogue) — the pair launch
 * above computes the scores, exponentials and dropout masks once per kernel.  Identical dK, dV and gain partials of hn_kv; dQ and
 * hn_q's partials differ by the order of the sum over key units.  ws == NULL, a workspace that is too small or a launch the two-pass
 * kernels do not serve (fp32 storage, one key tile, causal with Sq != Sk, unaligned operands): kk_attn_bwd runs.
 * kk_attn_bwd_two_pass(): whether the two passes are the faster form of a shape (the extra 2 bytes per score against the second
 * softmax) — the caller's policy for handing over a workspace; 0 for every shape beside the present kk_attn_bwd kernels. */
int64_t kk_attn_bwd_ws_bytes(int B, int heads, int Sq, int Sk);
int kk_attn_bwd_two_pass(int B, int heads, int Sq, int Sk, int causal);
int kk_attn_bwd_ws(const float *Q, const float *K, const float *V, const float *dO, const float *LSE, const float *Delta,
                   float *dQ, float *dK, float *dV, int B, int heads, int Sq, int Sk, int64_t ldq, int64_t ldk,
                   int64_t ldv, int64_t lddo, int64_t lddq, int64_t lddk, int64_t lddv, const uint8_t *key_mask,
                   int causal, float scale, const uint32_t *seed, uint32_t site, float p_drop, int math, int io_bf16,
                   const KkAttnHeadNorm *hn_q, const KkAttnHeadNorm *hn_kv, void *ws, int64_t ws_bytes, void *stream);

/* ---- norms ----
 * LayerNorm (nn.LayerNorm eps 1e-5; transformers.py:461-462,518-520,612; model.py:122). */
int kk_layernorm_fwd(const float *x, const float *gamma, const float *beta, float *y, float *mean,
                     float *rstd, int64_t rows, int H, int y_bf16, void *stream);
/* Backward.  The column reductions (dgamma, dbeta / dgain) leave each workgroup either as device-scope atomics into
 * the gradient vectors (partials == NULL) or as one row of partials[kk_norm_bwd_blocks(rows,H)][2H] ([..][H] for
 * RMSNorm) with plain stores; kk_partials_reduce then adds the column sums of any number of such matrices to their
 * gradient vectors in one launch (the engine does this once per step: the atomics were ~40 % of the kernel). */
int kk_norm_bwd_blocks(int64_t rows, int H);
typedef struct KkReduceDesc {
    const float *src;   /* [nblocks][ncols] partial sums, rows `stride` floats apart (0: ncols) */
    float *dst0;        /* columns [0, split) are added to dst0[c] */
    float *dst1;        /* columns [split, ncols) to dst1[c - split] */
    int nblocks, ncols, split, stride;
} KkReduceDesc;
int kk_partials_reduce(const KkReduceDesc *descs /* device memory */, int n, int max_cols, void *stream);
int kk_layernorm_bwd(const float *dy, const float *x, const float *gamma, const float *mean,
                     const float *rstd, float *dx, int dx_accumulate, float *dgamma, float *dbeta,
                     float *partials, int64_t rows, int H, int dy_bf16, void *stream);
/* RMSNorm over the full row, eps = FLT_EPSILON (GLU output_norm, transformers.py:94,109-110), fused with the
 * residual add of the block: y = (residual? residual : 0) + x*rstd*gain. */
int kk_rmsnorm_fwd(const float *x, const float *gain, const float *residual, float *y, float *rstd,
                   int64_t rows, int H, int x_bf16, void *stream);
int kk_rmsnorm_bwd(const float *dy, const float *x, const float *gain, const float *rstd, float *dx,
                   float *dgain, float *partials, int64_t rows, int H, int x_bf16, void *stream);
/* Per-head (64-wide) RMSNorm + optional RoPE rotate-half (transformers.py:260-277; positional_encoding.py:196-209)
 * over up to three column groups ("parts", e.g. q|k|v of a fused projection): element (row, part, head, d) at
 * [row*ld + part*heads*64 + head*64 + d]; gain_j / dgain_j belong to part j; bit j of rope_mask enables RoPE for
 * part j; position = row % S; cos/sin tables are [>=S, 64].  dgain_j accumulate (+=). */
int kk_headnorm_rope_fwd(const float *x, int64_t ldx, float *y, int64_t ldy, int64_t rows, int heads, int S,
                         int parts, const float *gain0, const float *gain1, const float *gain2, int rope_mask,
                         const float *cos_t, const float *sin_t, int io_bf16, void *stream);
/* partials (optional): [parts][kk_headnorm_bwd_blocks(rows,heads)][64] partial gain gradients instead of atomics into
 * dgain_j; sum them with kk_partials_reduce (one descriptor per part, ncols = 64). */
int kk_headnorm_bwd_blocks(int64_t rows, int heads);
int kk_headnorm_rope_bwd(const float *dy, int64_t lddy, const float *x, int64_t ldx, float *dx, int64_t lddx,
                         int64_t rows, int heads, int S, int parts, const float *gain0, const float *gain1,
                         const float *gain2, float *dgain0, float *dgain1, float *dgain2, float *partials,
                         int rope_mask, const float *cos_t, const float *sin_t, int io_bf16, void *stream);

/* ---- GLU feed-forward gate (transformers.py:107-108; exact-erf GELU): g = gelu(h[:, :F]) * h[:, F:] ---- */
int kk_glu_fwd(const float *h, float *g, int64_t rows, int F, const uint32_t *seed, uint32_t site, float p,
               int io_bf16, void *stream);
int kk_glu_bwd(const float *dg, const float *h, float *dh, int64_t rows, int F, const uint32_t *seed, uint32_t site,
               float p, int io_bf16, void *stream);

/* ---- embeddings + sinusoid PE (model.py:375-378; positional_encoding.py:66-74) ---- */
int kk_embed_fwd(const int64_t *ids, const int64_t *stress, const float *emb, const float *stress_emb,
                 const float *pe, float *out, int B, int P, int H, float scale, const uint32_t *seed,
                 uint32_t site, float p, void *stream);
/* kk_embed_fwd + the key-padding mask (key_mask[tok] = ids[tok] == 0, nullable; model.py:372) + kk_layernorm_fwd of the result (the
 * first encoder layer's pre-norm, transformers.py:468) in ONE launch, a wave per token; same bits as the separate launches. */
int kk_embed_ln_fwd(const int64_t *ids, const int64_t *stress, const float *emb, const float *stress_emb, const float *pe,
                    float *out, int B, int P, int H, float scale, const uint32_t *seed, uint32_t site, float p,
                    uint8_t *key_mask, const float *ln_gamma, const float *ln_beta, void *y, int y_bf16, float *mean,
                    float *rstd, void *stream);
int kk_embed_bwd(const int64_t *ids, const int64_t *stress, const float *dout, float *demb,
                 float *dstress_emb, int B, int P, int H, float scale, const uint32_t *seed, uint32_t site, float p,
                 void *stream);

/* ---- length regulator (utils/lengths.py:16-96): integer index expansion + payload gather ----
 * idx[b,f] = #{j : cumsum(max(dur_b,0))[j] <= f} for f < min(sum dur_b, L), else -1; lens[b] = min(sum, L);
 * total[b] = sum dur_b (un-clipped).  Bit-exact contract. */
int kk_length_regulate_index(const int64_t *dur, int64_t *idx, int64_t *lens, int64_t *total, int B, int P,
                             int L, void *stream);
int kk_length_regulate_gather(const float *x, const int64_t *idx, float *out, int B, int P, int L, int H,
                              void *stream);
/* max over a non-negative int64 vector (trainer.py:2224 reads phoneme_durations.max()). */
int kk_max_i64(const int64_t *x, int64_t n, int64_t *out, void *stream);
/* Expanded length T' = max_b sum(dur_b) != mel length T (model.py:607-628; variance_predictor.py:354-372, 396-420;
 * losses.py:111,137): dst[r, c] = c < cols_src ? src[r, c] : 0 for c < cols_dst — truncates the T'-frame pitch / energy
 * predictions to the T columns the losses read, and zero-pads their loss gradients / the frame-level targets back to T'. */
int kk_pad2d_f32(const float *src, int64_t lds, int cols_src, float *dst, int64_t ldd, int cols_dst, int64_t rows,
                 void *stream);
/* frame padding mask of the expanded sequence: mask[b, f] = (f >= lens[b])  (variance_predictor.py:363-369). */
int kk_frame_mask(const int64_t *lens, uint8_t *mask, int B, int T, void *stream);

/* ---- variance adaptor pieces (variance_predictor.py:89-115, 363-437) ---- */
/* col[(b,l), c*3+k] = x[b, l+k-1, c] inside the 512-frame chunk of l, else 0. */
int kk_im2col3_fwd(const float *x, float *col, int B, int L, int C, int chunk, int col_bf16, void *stream);
int kk_im2col3_bwd(const float *dcol, float *dx, int B, int L, int C, int chunk, int dcol_bf16, void *stream);
/* GroupNorm(1,C) over (C x chunk frames) per sample per chunk + ReLU; chunks with < 2 frames yield zeros.
 * stats [B*nchunks, 2] = (mean, rstd). */
int kk_groupnorm_relu_fwd(const float *x, const float *gamma, const float *beta, float *y, float *stats,
                          double *scratch, int B, int L, int C, int chunk, const uint32_t *seed, uint32_t site,
                          float p, void *stream);
int kk_groupnorm_relu_bwd(const float *dy, const float *x, const float *y, const float *gamma,
                          const float *stats, float *dx, float *dgamma, float *dbeta, double *scratch, int B,
                          int L, int C, int chunk, float p, int dx_bf16 /* dx is written as bf16 */, void *stream);
/* out[r] = mask[r] ? 0 : dot(x[r,:], w) + b  (Linear(C->1) + masked_fill; also the stop head, model.py:562). */
int kk_rowdot_fwd(const float *x, const float *w, const float *b, const uint8_t *mask, float *out,
                  int64_t rows, int C, int L, int chunk, int x_bf16, void *stream);
/* partials == NULL: dw[C] and db[1] are ACCUMULATED with atomics (every workgroup adds to the same C + 1 addresses: ~15 us of
 * same-address serialisation at 256 workgroups).  partials != NULL (C % 4 == 0): nothing is added; workgroup g writes the plain row
 * partials[g][C + 4] = (dw | db | pad) and kk_partials_reduce (nblocks = kk_rowdot_bwd_blocks(rows), ncols = C + 1, split = C,
 * stride = C + 4) sums the rows into dw / db. */
int kk_rowdot_bwd_blocks(int64_t rows);
int kk_rowdot_bwd(const float *dout, const float *x, const float *w, const uint8_t *mask, float *dx,
                  float *dw, float *db, int64_t rows, int C, int L, int chunk, int x_bf16, float *partials, void *stream);
/* frame_mask[b,f] = f >= lens[b]; bucketize(right=False) + two embedding adds + masked_fill. */
int kk_bucket_embed_add_fwd(const float *x, const float *pitch, const float *energy, const float *pbins,
                            const float *ebins, const float *pemb, const float *eemb, const int64_t *lens,
                            float *out, int32_t *pidx, int32_t *eidx, uint8_t *frame_mask, int B, int T,
                            int H, int nbins, int out_bf16, void *stream);
/* kk_length_regulate_gather + kk_bucket_embed_add_fwd (+ kk_specaug when seed != NULL) in ONE launch, a wave per frame: xf = the
 * regulated encoder rows (fp32, the predictors' input), out = the cross-attention memory (fp32 / bf16) with the SpecAugment masks of
 * (seed, site) already applied.  Same bits as the three launches (variance_predictor.py:338-439, model.py:636-639). */
int kk_regulate_embed_fwd(const float *enc, const int64_t *idx, const float *pitch, const float *energy, const float *pbins,
                          const float *ebins, const float *pemb, const float *eemb, const int64_t *lens, float *xf, float *out,
                          int32_t *pidx, int32_t *eidx, uint8_t *frame_mask, int B, int P, int T, int H, int nbins, int out_bf16,
                          const uint32_t *seed, uint32_t site, int time_mask_max, int feat_mask_max, int n_time, int n_feat,
                          void *stream);
int kk_bucket_embed_add_bwd(const float *dout, const int32_t *pidx, const int32_t *eidx,
                            const uint8_t *frame_mask, float *dpemb, float *deemb, int B, int T, int H,
                            int nbins, void *stream);   /* nbins = rows of the two embedding tables */
/* The same gradient as a segmented sum (no LDS float atomics).  kk_bucket_sort (forward side, once per batch): order = int32 [2][rows],
 * the unmasked frames of the pitch (0) and energy (1) table sorted by bin; items = int32 [2][kk_bucket_sort_items(rows, nbins)][4]
 * (16-byte aligned) = (bin, begin, end, 0) pieces of at most 16 frames of one bin, unused entries empty.  kk_bucket_embed_add_bwd_sorted adds
 * the rows of every piece into its table row (one workgroup per piece; accumulates like kk_bucket_embed_add_bwd; nbins <= 1024). */
int kk_bucket_sort_items(int64_t rows, int nbins);
int kk_bucket_sort(const int32_t *pidx, const int32_t *eidx, const uint8_t *frame_mask, int64_t rows, int nbins, int32_t *order,
                   int32_t *items, void *stream);
int kk_bucket_embed_add_bwd_sorted(const float *dout, const int32_t *order, const int32_t *items, float *dpemb, float *deemb,
                                   int64_t rows, int H, int nbins, void *stream);
/* text key mask: mask[i] = (ids[i] == 0)  (model.py:586-587). */
int kk_ids_eq_zero(const int64_t *ids, uint8_t *mask, int64_t n, void *stream);
/* decoder input shift-right (model.py:519): out[b,0,:]=0, out[b,t,:]=mel[b,t-1,:]. */
int kk_shift_right(const float *mel, float *out, int B, int T, int M, void *stream);

/* ---- autoregressive decode (KokoroGenerator.generate, model/generator.py:24-127; incremental attention with a KV cache,
 * transformers.py:237-253): the per-frame launches take the frame index t from DEVICE memory (*t_dev), so that ONE captured
 * hipGraph of a decoder step is replayed for every frame.
 * prologue: frame_in[B,M] = mel_all[:, t, :] (mel_all [B, L1, M]: row 0 = the all-zero first input, row t+1 = frame t's output),
 *   pe_row[H] = pe[t], cos_row / sin_row [64] = the RoPE tables' row t (the step's K is rotated by its absolute position,
 *   transformers.py:268-277), key_mask[t] = 0 (the self-attention runs over the whole cache under a key mask).
 * cache_append: nrm [B, 3H] (normalised q | k | v of the step) -> q [B*H], kcache[t] and vcache[t] ([L][B*H], time-major).
 * epilogue: mel_all[:, t+1, :] = frame_out [B,M]; stop_all[t, :] = stop [B]; *t_dev = t + 1. */
int kk_decode_prologue(const float *mel_all, float *frame_in, const float *pe, float *pe_row, const float *cos_t, const float *sin_t,
                       float *cos_row, float *sin_row, uint8_t *key_mask, const int *t_dev, int B, int L1, int M, int H, void *stream);
int kk_decode_cache_append(const void *nrm, void *q, void *kcache, void *vcache, const int *t_dev, int B, int H, int bf16, void *stream);
int kk_decode_epilogue(const float *frame_out, const float *stop, float *mel_all, float *stop_all, int *t_dev, int B, int L1, int M,
                       void *stream);

/* ---- dropout / DropPath / SpecAugment (p > 0 training paths; masks from an in-kernel counter RNG) ----
 * out = (res ? res[row % res_mod (0: row)] : 0) + x * m1 * m2 * droppath(sample(row)), m_i in {0, 1/(1-p_i)}
 * (transformers.py:16-40,482-487,569-581; the FFN has two dropouts in series, :111).  *seed is read on the device. */
int kk_dropout_fwd(const float *x, const float *res, int64_t res_mod, float *out, int64_t rows, int H, int S,
                   const uint32_t *seed, uint32_t site1, float p1, uint32_t site2, float p2, uint32_t site_dp,
                   float dp_rate, void *stream);
/* Fused tail of an attention / feed-forward sub-layer (forward, p > 0 paths): x_out = res + dropout_p1(dropout_p2(
 * drop_path([RMSNorm_gain](y)))) and, when ln_gamma is given, n = LayerNorm(x_out) with its mean / rstd — one launch
 * instead of kk_rmsnorm_fwd + kk_dropout_fwd + kk_layernorm_fwd; identical masks, so kk_dropout_bwd is its backward.
 * gain == NULL: no RMSNorm (attention output projection).  y: fp32 or bf16 (y_bf16); n: fp32 or bf16 (n_bf16). */
int kk_sublayer_out_fwd(const float *y, int y_bf16, const float *gain, float *rstd_f, const float *res, float *x_out,
                        const float *ln_gamma, const float *ln_beta, float *n, int n_bf16, float *mean, float *rstd,
                        int64_t rows, int H, int S, const uint32_t *seed, uint32_t site1, float p1, uint32_t site2,
                        float p2, uint32_t site_dp, float dp_rate, void *stream);
/* y = x.W^T + bias (x bf16 [rows, K] with pitch ldx, W bf16 [512, K]: the attention output projection transformers.py:131-136 /
 * :437, or the feed-forward's linear2 :105-111) AND kk_sublayer_out_fwd of that y in ONE launch: a workgroup owns 32 whole rows, the
 * [rows, 512] projection never reaches HBM.  y_round != 0: y is rounded to bf16 before the tail (what kk_gemm with a bf16 C followed
 * by kk_sublayer_out_fwd(y_bf16 = 1) computes: the results are bit-identical to those two launches); y_out (optional, bf16
 * [rows, 512]): y itself, for a backward that needs it (the RMSNorm of the feed-forward).  The other arguments are
 * kk_sublayer_out_fwd's.  H must be 512 and K a multiple of 64: ask kk_linear_tail_supported() first. */
int kk_linear_tail_supported(int64_t rows, int H, int K);
int kk_linear_tail_pays(int64_t rows, int H, int K);   /* 1 where the one launch measured faster than the two (K = 512, whole rounds of workgroups) */
int kk_linear_tail_fwd(const void *x, int64_t ldx, const void *W, const float *bias, int K, void *y_out, int y_round,
                       const float *gain, float *rstd_f, const float *res, float *x_out, const float *ln_gamma,
                       const float *ln_beta, float *n, int n_bf16, float *mean, float *rstd, int64_t rows, int H, int S,
                       const uint32_t *seed, uint32_t site1, float p1, uint32_t site2, float p2, uint32_t site_dp,
                       float dp_rate, void *stream);
/* Backward of kk_sublayer_out_fwd, fused with what follows it in the backward pass: dres (+)= LayerNorm_bwd(dn); dz = dres *
 * masks; dy = RMSNorm_bwd(dz) (gain != NULL, FFN) or dz (attention output projection).  The column reductions go to
 * partials[kk_sublayer_in_bwd_blocks(rows)][4][H] = (dgamma | dbeta | column sums of dy | dgain) for kk_partials_reduce
 * (row stride 4H).  Replaces kk_layernorm_bwd + kk_dropout_bwd (+ kk_rmsnorm_bwd) + kk_colsum_acc. */
int kk_sublayer_in_bwd_blocks(int64_t rows);
int kk_sublayer_in_bwd(const float *dn, int dn_bf16, const float *x_out, const float *ln_gamma, const float *mean,
                       const float *rstd, float *dres, int accumulate, const float *y, const float *gain,
                       const float *rstd_f, float *dy, int y_bf16, float *partials, int64_t rows, int H, int S,
                       const uint32_t *seed, uint32_t site1, float p1, uint32_t site2, float p2, uint32_t site_dp,
                       float dp_rate, void *stream);
int kk_dropout_bwd(const float *dy, float *dx, int64_t rows, int H, int S, const uint32_t *seed, uint32_t site1,
                   float p1, uint32_t site2, float p2, uint32_t site_dp, float dp_rate, int dx_bf16, void *stream);
/* SpecAugment on the cross-attention memory, in place (trainer.py:1577-1604); call again on the memory gradient. */
int kk_specaug(float *x, int B, int T, int H, const uint32_t *seed, uint32_t site, int time_mask_max,
               int feat_mask_max, int n_time, int n_feat, int x_bf16, void *stream);

/* ---- the whole text-encoder forward as one persistent launch (model/model.py:375-388: the stack of FFT blocks,
 * transformers.py:452-490 = pre-LN self-attention with per-head RMSNorm + RoPE :260-277,393-398 and the GLU feed-forward
 * :86-112, each followed by dropout / DropPath / residual :482-487) ----
 * Replaces, per layer, kk_gemm_qkv_headnorm + kk_attn_fwd + kk_gemm + kk_sublayer_out_fwd + kk_gemm_linear_glu + kk_gemm +
 * kk_sublayer_out_fwd (bf16 storage) with the same outputs in the same buffers, so the per-kernel backward is unchanged.
 * Work is partitioned by batch item: workgroups b, b+8, ... carry item b through all layers and meet at group barriers
 * (csrc/kk_encstack.hip).  Limits: kk_encoder_stack_supported().  `sync` = 512 uint32 words, zero before the first call
 * (the launch leaves them zero; word 0 != 0 afterwards means a barrier timed out and the outputs are invalid). */
#define KK_ENC_MAX_LAYERS 8
typedef struct KkEncLayer {
    const void *w_qkv;                    /* [3H, H] bf16: w_q | w_k | w_v */
    const float *g_q, *g_k, *g_v;         /* per-head RMSNorm gains [64] */
    const void *w_o;  const float *b_o;   /* [H, H] bf16, [H] */
    const float *ln2_g, *ln2_b;           /* LayerNorm in front of the feed-forward */
    const void *w1;   const float *b1;    /* [2F, H] bf16, [2F] */
    const void *w2;   const float *b2;    /* [H, F] bf16, [H] */
    const float *ffn_gain;                /* RMSNorm(H) gain of the feed-forward output */
    const float *next_g, *next_b;         /* the LayerNorm that follows the layer (next layer's norm1 / the stack's final norm) */
    const void *y1;                       /* in: LayerNorm_1 output [N, H] bf16 (layer l > 0: layer l-1's next_y) */
    void *qkv_raw, *qkv_n;                /* [N, 3H] bf16 */
    void *ctx;  float *lse;               /* [N, H] bf16, [B, heads, S] */
    float *proj;                          /* [N, H] fp32 scratch (one per layer) */
    const float *x_in;                    /* residual stream in [N, H] fp32 */
    float *xm;  void *y2;  float *mean2, *rstd2;          /* after attention: stream, LayerNorm_2 output (bf16) and statistics */
    void *h1, *g, *f2;  float *rstd_f;    /* [N, 2F], [N, F], [N, H] bf16; 1/rms of f2 */
    float *xo;  void *next_y;  float *next_mean, *next_rstd;   /* layer output stream, following LayerNorm output + statistics */
    int next_y_bf16;                      /* storage of next_y (the stack's final norm is fp32) */
    uint32_t site;                        /* dropout call-site base of the layer (attention: +0..3, feed-forward: +8..12) */
    float p, dpr;                         /* dropout probability, stochastic-depth rate */
} KkEncLayer;
typedef struct KkEncStack {
    int B, S, H, F, heads, layers;        /* N = B*S rows */
    const uint8_t *key_mask;              /* [B, S], 1 = padded key (may be null) */
    const float *cos_t, *sin_t;           /* RoPE tables [>= S, 64] */
    const uint32_t *seed;                 /* step seed, read on the device */
    uint32_t *sync;                       /* 512 words, see above */
    int placement;                        /* 0: group = workgroup % 8 (the members of a group share an XCD under the observed
                                             round-robin dispatch); 1: group = workgroup / members (a group spread over all XCDs) —
                                             results are identical, 1 exists to test placement independence */
    uint64_t *trace;                      /* optional (tools): workgroup `trace_wg` stamps the 100 MHz clock after every phase and
                                             after every barrier: 2 * 7 * layers words per item */
    int trace_wg;
    KkEncLayer layer[KK_ENC_MAX_LAYERS];
} KkEncStack;
int kk_encoder_stack_supported(int B, int S, int H, int F, int heads, int layers);
int kk_encoder_stack_workgroups(void);
int kk_encoder_stack_fwd(const KkEncStack *desc /* host struct, read during the call */, void *stream);

/* ---- losses (training/losses.py:9-216) ----
 * acc: 12 doubles (5 sums, 5 counts, the number of non-finite prediction elements, 1 spare) zeroed by the call.
 * losses: 6 floats (total, mel, dur, stop, pitch, energy).
 * coef: 5 floats = d(total*loss_scale)/d(per-element loss) for the backward kernel.
 * guard (nullable): &opt_state[KK_OS_MICRO_BAD] — the reference's per-micro-batch guards (trainer.py:3233-3256 finite
 * outputs, :3274-3296 finite losses): a flagged micro-batch back-propagates nothing (coef = 0) and marks its accumulation
 * cycle, whose optimizer boundary kk_opt_prepare then skips (trainer.py:2304-2314 drops the accumulated gradients). */
typedef struct KkLossCfg {
    float w_dur, w_stop, w_pitch, w_energy;
    float delta_dur, delta_pitch, delta_energy, pos_weight;
    float loss_scale;             /* 1/accumulation divisor (trainer.py:2284-2294) */
    int adaptive;                 /* 1: also apply the batch-shape loss scale of trainer.py:2218-2242 on device */
} KkLossCfg;
int kk_losses_fwd(const float *mel_pred, const float *mel_tgt, const float *dur_pred, const int64_t *dur,
                  const float *stop_logit, const float *stop_tgt, const float *pitch_pred,
                  const float *pitch_tgt, const float *energy_pred, const float *energy_tgt,
                  const int64_t *mel_len, const int64_t *ph_len, int B, int T, int P, int M,
                  const KkLossCfg *cfg, const int64_t *max_dur /* device scalar or null */, double *acc,
                  float *losses, float *coef, double *guard, int flags, void *stream);
/* flags of kk_losses_fwd — the 12-double accumulator costs a zero-fill launch on the step's critical chain unless it is handed
 * round: bit 0 = acc is zero on entry (the call launches no zero-fill), bit 1 = the call's own finalize leaves acc zero.  A
 * single-GPU step passes 3 on a buffer allocated zero; a data-parallel step passes 1 and lets kk_losses_finalize(clear = 1) — the
 * last reader — clear it.  0 is the self-contained form (zero-fill inside, acc kept). */
#define KK_LOSS_ACC_ZEROED 1
#define KK_LOSS_ACC_CLEAR 2
/* Recompute losses[6] and coef[5] from acc (see kk_losses_fwd) — used by data-parallel runs after acc has been
 * SUM-reduced and *max_dur MAX-reduced over the ranks: normalisers become the global valid-element counts.
 * clear != 0: acc is zero afterwards. */
int kk_losses_finalize(double *acc, const KkLossCfg *cfg, const int64_t *max_dur, int T, float *losses,
                       float *coef, double *guard, int clear, void *stream);
int kk_losses_bwd(const float *mel_pred, const float *mel_tgt, const float *dur_pred, const int64_t *dur,
                  const float *stop_logit, const float *stop_tgt, const float *pitch_pred,
                  const float *pitch_tgt, const float *energy_pred, const float *energy_tgt,
                  const int64_t *mel_len, const int64_t *ph_len, int B, int T, int P, int M,
                  const KkLossCfg *cfg, const float *coef, float *dmel, float *ddur, float *dstop,
                  float *dpitch, float *denergy, void *stream);

/* ---- optimizer pass over the flat arena (trainer.py:1332-1407,2355-2405; runtime_policies.py:14-87;
 *      torch AdamW; trainer.py:1491-1517,882-912) ---- */
typedef struct KkOptCfg {
    /* schedule (trainer.py:691-772,1519-1575) */
    double learning_rate, max_lr, warmup_start_lr, warmup_target_lr, pct_start, div_factor, final_div_factor;
    int64_t warmup_steps, onecycle_steps;
    int use_warmup;
    /* AdamW */
    double beta1, beta2, eps;
    /* clipping */
    double max_grad_norm;
    int64_t mel_length;           /* batch mel length for the adaptive clip (trainer.py:2218-2242) */
    /* explosion tracker (trainer.py:1315-1330,2367-2405) */
    double expl_alpha, expl_abs_floor, expl_multiplier, expl_warmup_floor;
    int64_t expl_warmup_steps, expl_min_ema_steps;
    /* EMA + weight-norm */
    double ema_decay;
    double max_weight_norm;
    int64_t ema_update_every;     /* the EMA moves on successful steps 0, N, 2N, ... (trainer.py:1499-1502); <= 1: every step */
    /* legacy schedule (use_onecycle_lr = False, trainer.py:789-799): legacy_schedule != 0 replaces warm-up + OneCycle by
     * lr(segment) = eta_min + (learning_rate * lr_mult - eta_min) * legacy_cos, legacy_cos = the host's (1 + cos(pi T_cur / T_i)) / 2
     * of the current EPOCH (CosineAnnealingWarmRestarts is stepped once per epoch); a zero-filled struct is the OneCycle schedule */
    int64_t legacy_schedule;
    double legacy_cos, eta_min;
} KkOptCfg;
/* device-resident optimizer state (doubles): see kk_opt_state_* indices */
#define KK_OS_SKIPPED 0      /* boundaries skipped for non-finite grads */
#define KK_OS_EXPL_EMA 1
#define KK_OS_EXPL_EMA_STEPS 2
#define KK_OS_EXPL_STREAK 3
#define KK_OS_LAST_GRAD_NORM 4
#define KK_OS_LAST_CLIP_COEF 5
#define KK_OS_LAST_SKIP 6
#define KK_OS_LAST_BASE_LR 7
#define KK_OS_LAST_CLIP_NORM 8
#define KK_OS_EXPL_EMA_VALID 9
#define KK_OS_ATTEMPT 10      /* optimizer-step boundaries reached so far (successful = ATTEMPT - SKIPPED) */
#define KK_OS_BAD_SEG 11      /* diagnostics of the most recent skipped boundary: 1 + first segment with a non-finite norm, */
#define KK_OS_BAD_COUNT 12    /* how many segments had one, */
#define KK_OS_BAD_ATTEMPT 13  /* and the boundary index (ATTEMPT) at which it happened */
#define KK_OS_MICRO_BAD 14    /* set by kk_losses_*: a micro-batch of the current cycle had non-finite outputs / losses */
#define KK_OS_MICRO_BAD_TOTAL 15 /* micro-batches flagged so far */
#define KK_OS_SIZE 16
/* sumsq[seg] (double) = sum of squares of each arena segment of `buf`, as a PURE FUNCTION of the buffer: workgroup partials are
 * merged in arena order by a fixed tree (no atomics), so data-parallel replicas holding the same reduced gradient compute the same
 * bits (trainer.py:2355-2362 is a host-side sum with the same property).  Every segment is stored exactly once: no zero-fill.
 * ws: kk_seg_sumsq_ws_bytes(nblocks) bytes of scratch: the arena walk's records first, then room for kk_gemm_wgrad_group's tile
 * records (kk_seg_sumsq_rec_offset() bytes into ws, kk_seg_sumsq_rec_capacity() records).
 * seg_skip (nullable, int32[nseg], device): segments NOT read by the walk — their sums come from `extra_records` tile records that
 * the step's weight-gradient launches left behind the walk's (one GPU only: data parallel norms need the REDUCED gradient). */
int64_t kk_seg_sumsq_ws_bytes(int64_t nblocks);
int64_t kk_seg_sumsq_rec_offset(void);
int kk_seg_sumsq_rec_capacity(void);
int kk_seg_sumsq(const float *buf, const int32_t *block_seg, int64_t nblocks, double *sumsq, int nseg, void *ws,
                 const int32_t *seg_skip, int extra_records, void *stream);
/* One-thread-block kernel: per-parameter pre-clip, total norm, non-finite check, explosion tracker,
 * adaptive + global clip, LR schedule -> per-segment gradient scale / lr / step-size constants. */
int kk_opt_prepare(const double *grad_sumsq, const float *seg_preclip, const float *seg_lr_mult,
                   const float *seg_wd, int nseg, const int64_t *max_dur, const KkOptCfg *cfg,
                   double *opt_state, float *seg_gscale, float *seg_decay, float *seg_stepsize,
                   float *step_consts /* [4]: mode (0 step + EMA, 1 skipped, 2 step without EMA), sqrt(1-beta2^t), eps, base_lr */,
                   double *clear_a, double *clear_b /* nullable: per-segment accumulators [nseg] left zero by this launch — grad_sumsq
                   itself (kk_seg_sumsq stores every segment, so this is hygiene only), p_sumsq (8-byte words: zero bits are zero in both
                   types) for the kk_adamw_ema(zeroed = 1) that follows */, void *stream);
/* Fused single pass: p,g,m,v,(ema) -> p,m,v,(ema).  seg_flags bit0: AdamW-updated, bit1: EMA-tracked,
 * bit2: weight-norm target (its post-step sum of squares is accumulated into p_sumsq as a Q34.30 FIXED-POINT integer — integer adds
 * commute, so the sum and the projection decision taken from it do not depend on the order the blocks arrive in; zeroed == 0:
 * zero-filled by the call first, zeroed != 0: zero on entry, e.g. cleared by the kk_opt_prepare in front of it).
 * p_bf16 (optional): bf16 shadow of the arena, same element offsets, rewritten wherever p is (the bf16 mode's
 * GEMMs read weights from it). */
int kk_adamw_ema(float *p, const float *g, float *m, float *v, float *ema, const int32_t *block_seg,
                 int64_t nblocks, const float *seg_gscale, const float *seg_decay, const float *seg_stepsize,
                 const int32_t *seg_flags, const float *step_consts, float beta1, float beta2,
                 float ema_decay, int64_t *p_sumsq, int nseg, void *p_bf16, int zeroed, void *stream);
/* FFN weight-norm projection: for flagged segments with ||W|| > max: W *= max/||W||. */
int kk_weight_norm_project(float *p, const int32_t *block_seg, int64_t nblocks, const int64_t *p_sumsq /* Q34.30 */,
                           const int32_t *seg_flags, const float *step_consts, double max_norm, void *p_bf16,
                           void *stream);
/* dst (bf16) = src (fp32), n % 4 == 0: builds the weight shadow after a checkpoint load. */
int kk_cast_f32_bf16(const float *src, void *dst, int64_t n, void *stream);

/* ---- one XCD per batch item: a decoder sub-layer as ONE persistent launch (round 6; csrc/kk_chain.hip) ----
 * Between kk_chain_begin() and kk_chain_launch() the entry points called on this thread RECORD the launch they would have made
 * (same dispatch code, same argument blocks) instead of making it; kk_chain_launch runs the recorded kernels' bodies as the phases of
 * one launch of 256 workgroups — the workgroups of XCD x carry batch item x through all phases, group barriers in between, hand-over
 * through the XCD's L2 — or returns KK_ENOTSUP and launches NOTHING when it does not carry that sequence (the caller then issues the
 * launches again, outside a capture).  kind 0 = the decoder's self-attention sub-layer forward (transformers.py:543-560):
 * kk_gemm_qkv_headnorm, kk_attn_fwd[_kb], kk_gemm (w_o), kk_sublayer_out_fwd at B = 8, 512 or 1024 frames, hidden 512.
 * sync: 512 zero-initialised uint32 (word 0 != 0: a barrier timed out; word 1: workgroups found on another XCD than their item's).
 * flags bit 0: agent-scope barrier atomics (placement independent) instead of XCD-local ones.  trace: tools (16 uint64 per workgroup) or NULL. */
int kk_chain_begin(void *stream);
int kk_chain_launch(int kind, uint32_t *sync, uint64_t *trace, int flags, void *stream);
int kk_chain_abort(void *stream);

/* ---- data-parallel gradient exchange over RCCL / xGMI (new functionality: the reference has no distributed code,
 * SURVEY §0 fact 2; contract = "the same maths as one process seeing the global batch", §8e) ----
 * One communicator per process (one process per GPU).  RCCL is bound at run time: the library loads without it, and
 * kk_comm_load(path) lets the host name the instance to use (PyTorch-ROCm ships its own; default = the one already in
 * the process, else the loader's search path).  rank 0 creates the 128-byte id (kk_comm_unique_id) and hands it to the
 * other ranks by any side channel (the Python host uses the torch.distributed store).  Every collective is enqueued on
 * `comm_stream` and never synchronises: the calls are legal inside a hipGraph capture, so the exchange of a bucket is
 * a branch of the step's graph beside the rest of the backward.  dtype: 0 = fp32, 1 = bf16 payload. */
int kk_comm_load(const char *rccl_path /* may be null */);
int kk_comm_unique_id(void *id128);
int kk_comm_init(int rank, int world, const void *nccl_unique_id);
int kk_comm_world(void);                                           /* ranks of the live communicator, 0 = none */
int kk_comm_destroy(void);
/* in-place SUM all-reduce of one gradient bucket (ptr[0 .. count)) */
int kk_comm_reduce_bucket(void *ptr, int64_t count, int dtype, void *comm_stream);
/* n in-place SUM all-reduces base[begin[i] .. end[i]) (element indices; HOST arrays) as one RCCL group */
int kk_comm_reduce_ranges(void *base, const int64_t *begin, const int64_t *end, int n, int dtype, void *comm_stream);
/* Loss normalisers of ragged data-parallel shards: acc[n_acc] (fp64 sums + valid-element counts of kk_losses_fwd;
 * reference losses.py:40-46,82-105) SUM-reduced and *max_dur (trainer.py:2218-2242's inputs) MAX-reduced in place, one RCCL
 * group on `stream`; kk_losses_finalize then normalises by the global counts.  Legal inside a hipGraph capture. */
int kk_comm_loss_sync(double *acc, int n_acc, int64_t *max_dur, void *stream);
/* the two halves of the ring all-reduce on their own (recv of rank r = block r of the sum / the concatenation) */
int kk_comm_reduce_scatter(const void *send, void *recv, int64_t recv_count, int dtype, void *comm_stream);
int kk_comm_all_gather(const void *send, void *recv, int64_t send_count, int dtype, void *comm_stream);
/* y[i] = scale * float(x[i]) for a bf16 x: widens a bf16 gradient bucket after its exchange */
int kk_cast_bf16_f32(const void *x, float *y, int64_t n, float scale, void *stream);
/* The bf16 payload of a gradient bucket (reference: none — trainer.py has no data parallelism): dst[begin_i, end_i) =
 * cast(src[begin_i, end_i)) for i < n over two arrays of the same layout (the fp32 gradient arena and its bf16 twin), to_bf16 = 1
 * narrows, 0 widens (times scale).  ONE launch of at most 128 thin workgroups for all ranges: it runs on the communication stream beside
 * the backward.  begin / end are HOST arrays of element offsets (begin % 4 == 0), read during the call. */
int kk_cast_ranges(const void *src, void *dst, const int64_t *begin, const int64_t *end, int n, int to_bf16, float scale,
                   void *stream);

/* ---- misc ---- */
/* *slot = device wall clock (100 MHz ticks) at the time this launch executes: in-graph time stamps for timelines. */
int kk_timestamp(uint64_t *slot, void *stream);
/* (replaces the per-tensor `.to(device)` / copy of a batch dict, trainer.py:2257-2290)  n <= 16 device-to-device copies (dst[i] <- src[i], bytes[i] each; host arrays of device pointers) as one launch:
 * the hand-over of a batch's tensors into the buffers the captured step reads. */
int kk_copy_many(const void *const *src, void *const *dst, const int64_t *bytes, int n, void *stream);
/* Zero-fill of up to 160 ranges (16-byte aligned, multiples of 16 bytes) in ONE launch: the gradient arena at the start of an
 * accumulation cycle minus the weight-gradient matrices that the cycle's first kk_gemm_wgrad_group launches overwrite
 * (optimizer.zero_grad(), trainer.py:2257-2262, without touching 89 % of the arena). */
int kk_zero_many(void *const *dst, const int64_t *bytes, int n, void *stream);
int kk_axpby(float a, const float *x, float b, float *y, int64_t n, void *stream); /* y = a*x + b*y */
int kk_mfma_probe(float *out_f32 /*32*32*/, float *out_bf16 /*32*32*/, void *stream);

#ifdef __cplusplus
}
#endif
#endif
