// Argument block and shared constants of the bf16-storage GEMM cores (kk_gemm16.hip: 64x64 / 128x64 tiles, one 32x32
// accumulator per wave; kk_gemm16x.hip: the large-tile family, several accumulators per wave).
#pragma once
#include "kk_common.h"

struct G16Args {
    int M, N, K;
    float alpha, beta;
    const void *A, *B;
    const float *bias, *residual;
    void *C;
    int c_bf16;
    int64_t lda, ldb, ldc, ldr, res_mod;
    int k_per_split, atomic, splits, split_major;
    int tiles_m, tiles_n, xcd_swizzle;
    int m_fast;                                                 // tile order inside an XCD's run: 0 = n fastest, 1 = m fastest (see gemm16_body)
    int dbg;                                                    // tools builds only (KK_DBG): timing probes that change results
    int wt;                                                     // write-through stores of C and the epilogues' outputs (kk_common.h: kk_write_through)
    uint32_t a_bytes, b_bytes;
    // weight gradients written exactly once per element (fp32 C through the LDS transpose, no k-slices): each tile also leaves the sum of
    // squares of the FINAL values it stored as record `tile index` of ss_rec (segment id ss_seg) — kk_seg_sumsq then skips the tensor
    KkSegRec *ss_rec;
    int ss_seg, ss_rows;                                        // segment of dW's row 0; rows per segment when dW spans several ADJACENT segments
                                                                // (the fused q|k|v / k|v views: a multiple of the tile height), else 0
    // EPI == 1 (GLU backward epilogue): C is not written; see gemm16_kernel
    const __bf16 *glu_h;
    __bf16 *glu_dh;
    float *glu_partials;
    const uint32_t *glu_seed;
    uint32_t glu_site;
    float glu_p;
    // EPI == 3 (per-head RMSNorm + RoPE epilogue): C receives the raw projection, hn_y the normalised one
    const float *hn_gain[12];                                   // one gain vector per part (a part = hn_H columns: q | k | v | ...)
    const float *hn_cos, *hn_sin;
    __bf16 *hn_y;
    int64_t hn_ldy;
    int hn_S, hn_H, hn_rope_mask;
    // Delta epilogue (eight-wave 128x64 tile, bf16 C; the dgrad of an attention output projection): the tile's 64 columns are
    // one head of dO = dY.W_o, so Delta[b, head, q] = sum_d dO * O (the attention backward's row term) leaves with it
    const __bf16 *dl_o;
    float *dl_out;
    int64_t dl_ldo;
    int dl_S, dl_heads;
};

constexpr int BK = 64;

// cache policy of the ACTIVATION operand's buffer-load-to-LDS DMA in gemm16_body / g16x_body: 0 in the library's own launches;
// kk_chain.hip compiles the bodies with sc1 (16: L1 bypass, served by the XCD's L2) for tensors written earlier in the same launch
#ifndef KK_A_AUX
#define KK_A_AUX 0
#endif

// Several independent GEMMs of one operand layout in ONE launch (a layer's weight gradients): descriptor table in the kernel arguments.
constexpr int GROUP_MAX = 8;
struct G16Group {
    int n;
    int xcd_chunks;                                             // kk_gemm16x.hip: XCD x takes the x-th contiguous eighth of ALL problems' tiles (see g16x_group_kernel)
    int start[GROUP_MAX + 1];                                   // first workgroup of each problem
    G16Args p[GROUP_MAX];
};

// Large-tile family (kk_gemm16x.hip).  cfg: the workgroup tile; every one is run by eight waves.
enum { G16X_128x128 = 0, G16X_256x128 = 1, G16X_128x192 = 2, G16X_256x192 = 3, G16X_NCFG = 4 };
void kk_g16x_tile(int cfg, int *bm, int *bn);
// plain GEMM (bias / bf16 or fp32 C / Delta epilogue); a.tiles_m / tiles_n must already match the tile
int kk_g16x_plain(int cfg, int ta, int tb, const G16Args &a, hipStream_t s);
int kk_g16x_headnorm(int cfg, const G16Args &a, hipStream_t s);       // q|k|v projection + per-head RMSNorm (+ RoPE) epilogue
int kk_g16x_glu_fwd(const G16Args &a, hipStream_t s);                 // linear1 + GLU gate: 256 rows x (96 + 96) columns per workgroup
int kk_g16x_glu_bwd(const G16Args &a, hipStream_t s);                 // linear2 dgrad + GLU backward: 128 x 192
int kk_g16x_group(const G16Group &g, int grid, hipStream_t s);        // grouped weight gradients on 128 x 128 tiles
