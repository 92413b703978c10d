// Dropout, DropPath (stochastic depth) and SpecAugment for the training forward/backward.
//
//  residual dropout + DropPath   model/transformers.py:16-40 (drop_path), :482-487, :569-581: x + dropout(drop_path(y))
//  FFN output dropout            transformers.py:111 (a second dropout in front of the residual one)
//  decoder-input dropouts        model/model.py:525-531 (functional dropout, then PE add, then PE dropout)
//  SpecAugment on the memory     training/trainer.py:1577-1604, applied at model/model.py:636-639
// Masks come from the counter RNG in kk_common.h; the backward kernels regenerate them.  These are the p > 0 paths;
// with p == 0 the engine uses the fused GEMM / RMSNorm residual epilogues instead.
#include "kk_common.h"
#include <float.h>

namespace {

struct DropArgs {
    const uint32_t *seed;
    uint32_t site1, site2, site_dp;
    float p1, p2, dp_rate;
    int S;   // rows per sample (DropPath is per sample)
};

__device__ __forceinline__ float row_scale(const DropArgs &d, uint32_t seed, int64_t row) {
    if (d.dp_rate <= 0.f) return 1.f;
    return kk_drop_mul(seed, d.site_dp, (uint64_t)(row / d.S), kk_drop_threshold(d.dp_rate), 1.f / (1.f - d.dp_rate));
}

// out = (res ? res[(row % res_mod)] : 0) + x * m1 * m2 * droppath(row)      (fwd)
// dx  = dy * m1 * m2 * droppath(row)                                        (bwd: res == nullptr, x = dy)
template <typename TO>
__global__ __launch_bounds__(256) void dropout_kernel(const float *__restrict__ x, const float *__restrict__ res, int64_t res_mod,
                                                      TO *__restrict__ out, int64_t total4, int H, DropArgs d) {
    const uint32_t seed = *d.seed;
    const uint32_t t1 = kk_drop_threshold(d.p1), t2 = kk_drop_threshold(d.p2);
    const float k1 = d.p1 > 0.f ? 1.f / (1.f - d.p1) : 1.f, k2 = d.p2 > 0.f ? 1.f / (1.f - d.p2) : 1.f;
    const int H4 = H / 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / H4;
        const int c = (int)(i - row * H4) * 4;
        const float rs = row_scale(d, seed, row);
        const float4 v = ld4(x + row * H + c);
        float o[4] = {v.x, v.y, v.z, v.w};
        float m1[4], m2[4];
        kk_drop_mul4(seed, d.site1, (uint64_t)row * H + c, t1, k1, m1);
        kk_drop_mul4(seed, d.site2, (uint64_t)row * H + c, t2, k2, m2);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] *= rs * m1[e] * m2[e];
        if (res) {
            const int64_t rr = res_mod > 0 ? row % res_mod : row;
            const float4 r = ld4(res + rr * H + c);
            o[0] += r.x; o[1] += r.y; o[2] += r.z; o[3] += r.w;
        }
        stv4<TO>(out + row * H + c, make_float4(o[0], o[1], o[2], o[3]));
    }
}

// ---- fused sub-layer tail (forward): [RMSNorm(H)] -> dropout(s) -> DropPath -> + residual -> [LayerNorm of the sum] ----
// One wave per row, the row in registers: replaces kk_rmsnorm_fwd + kk_dropout_fwd + kk_layernorm_fwd (three launches
// and two extra round trips of the [rows, H] tensor) at the end of every attention / feed-forward sub-layer.  The
// masks are the same functions of (seed, site, element) as dropout_kernel's, so kk_dropout_bwd regenerates them.
struct SubOutArgs {
    const void *y;                 // sub-layer result (TY): output projection (fp32) or FFN linear2 output (bf16 / fp32)
    const float *gain;             // optional RMSNorm(H) gain (GLU output_norm, transformers.py:109-110) ...
    float *rstd_f;                 // ... and where its 1/rms goes (saved for backward)
    const float *res;              // residual stream in
    float *x_out;                  // residual stream out = res + masks * norm(y)
    const float *ln_gamma, *ln_beta;   // optional LayerNorm of x_out (the next sub-layer's pre-norm / the stack's final norm)
    void *n;                       // its output (TN)
    float *mean, *rstd;
    int64_t rows;
    int H;
    int wt;                        // write-through output stores (kk_write_through(rows))
    DropArgs d;
};

// one row, one wave (every lane active); called by the launch below and by the persistent chain (kk_chain.hip)
template <typename TY, typename TN, int NV>
__device__ __forceinline__ void sublayer_out_row(const SubOutArgs &a, const int64_t row) {
    const int lane = threadIdx.x & 63, H = a.H;
    const uint32_t seed = *a.d.seed;
    const uint32_t t1 = kk_drop_threshold(a.d.p1), t2 = kk_drop_threshold(a.d.p2);
    const float k1 = a.d.p1 > 0.f ? 1.f / (1.f - a.d.p1) : 1.f, k2 = a.d.p2 > 0.f ? 1.f / (1.f - a.d.p2) : 1.f;
    const TY *yr = static_cast<const TY *>(a.y) + row * H;
    // every global load of the row is issued here, before the first reduction: the kernel is three dependent phases
    // (RMS statistic -> residual add -> LayerNorm) and was paying one memory latency per phase
    float4 v[NV], rres[NV], gg[NV], lg[NV], lb[NV];
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane * 4 + 256 * i;
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        v[i] = c < H ? ldv4<TY>(yr + c) : z;
        rres[i] = c < H ? ld4(a.res + row * H + c) : z;
        gg[i] = (a.gain && c < H) ? ld4(a.gain + c) : z;
        lg[i] = (a.ln_gamma && c < H) ? ld4(a.ln_gamma + c) : z;
        lb[i] = (a.ln_gamma && c < H) ? ld4(a.ln_beta + c) : z;
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) q += v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
    float rs = 1.f;
    if (a.gain) {
        rs = 1.f / sqrtf(wave_sum(q) / (float)H + FLT_EPSILON);
        if (lane == 0) a.rstd_f[row] = rs;
    }
    const float dp = row_scale(a.d, seed, row);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane * 4 + 256 * i;
        if (c < H) {
            float o[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
            if (a.gain) {
                const float4 g = gg[i];
                o[0] = o[0] * rs * g.x; o[1] = o[1] * rs * g.y; o[2] = o[2] * rs * g.z; o[3] = o[3] * rs * g.w;
            }
            const float4 r = rres[i];
            const float rr[4] = {r.x, r.y, r.z, r.w};
            float m1[4], m2[4];
            kk_drop_mul4(seed, a.d.site1, (uint64_t)row * H + c, t1, k1, m1);
            kk_drop_mul4(seed, a.d.site2, (uint64_t)row * H + c, t2, k2, m2);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = o[e] * (dp * m1[e] * m2[e]) + rr[e];
            v[i] = make_float4(o[0], o[1], o[2], o[3]);
            st4_out(a.x_out + row * H + c, v[i], a.wt);
            s += o[0] + o[1] + o[2] + o[3];
        }
    }
    if (!a.ln_gamma) return;
    const float mean = wave_sum(s) / (float)H;
    float qq = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (lane * 4 + 256 * i < H) {
            const float e0 = v[i].x - mean, e1 = v[i].y - mean, e2 = v[i].z - mean, e3 = v[i].w - mean;
            qq += e0 * e0 + e1 * e1 + e2 * e2 + e3 * e3;
        }
    }
    const float rstd = 1.f / sqrtf(wave_sum(qq) / (float)H + 1e-5f);
    TN *nr = static_cast<TN *>(a.n) + row * H;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane * 4 + 256 * i;
        if (c < H) {
            const float4 g = lg[i], b = lb[i];
            stv4_out<TN>(nr + c, make_float4((v[i].x - mean) * rstd * g.x + b.x, (v[i].y - mean) * rstd * g.y + b.y,
                                         (v[i].z - mean) * rstd * g.z + b.z, (v[i].w - mean) * rstd * g.w + b.w), a.wt);
        }
    }
    if (lane == 0) {
        a.mean[row] = mean;
        a.rstd[row] = rstd;
    }
}

template <typename TY, typename TN, int NV>
__global__ __launch_bounds__(256) void sublayer_out_fwd_kernel(SubOutArgs a) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.rows) return;
    sublayer_out_row<TY, TN, NV>(a, row);
}

#ifdef KK_BODIES_ONLY
}  // namespace   (kk_chain.hip includes this file for SubOutArgs / sublayer_out_row only)
#else
template <typename TY, typename TN>
void launch_subout(const SubOutArgs &a, hipStream_t s) {
    const dim3 grid(kk_cdiv(a.rows, 4)), blk(256);
    const int nv = kk_cdiv(a.H, 256);
    kk_note_kernelf("sublayer_out_fwd<%d,%d,%d>", (int)sizeof(TY), (int)sizeof(TN), nv <= 1 ? 1 : (nv <= 2 ? 2 : (nv <= 4 ? 4 : 8)));
    if (kk_capture(kk_last_kernel(), a, grid, 256, 0)) return;
    if (nv <= 1) hipLaunchKernelGGL((sublayer_out_fwd_kernel<TY, TN, 1>), grid, blk, 0, s, a);
    else if (nv <= 2) hipLaunchKernelGGL((sublayer_out_fwd_kernel<TY, TN, 2>), grid, blk, 0, s, a);
    else if (nv <= 4) hipLaunchKernelGGL((sublayer_out_fwd_kernel<TY, TN, 4>), grid, blk, 0, s, a);
    else hipLaunchKernelGGL((sublayer_out_fwd_kernel<TY, TN, 8>), grid, blk, 0, s, a);
}

// ---- a Linear with H = 512 outputs AND the sub-layer tail behind it as ONE launch (row ownership) -------------------------------
//   y = x.W^T + b  (bf16 x bf16 -> fp32, [rounded to bf16 as the two-launch path stores it]);  then exactly sublayer_out_fwd_kernel.
// Replaces kk_gemm + kk_sublayer_out_fwd behind an attention output projection (K = 512) or a feed-forward's linear2 (K = F): the
// [rows, 512] projection never exists in HBM (-2 KB per row of traffic, one launch and its ~3 us of dependent-launch gap less).
//
// A LayerNorm needs whole rows, so a workgroup owns 32 rows x all 512 columns and the WHOLE weight matrix streams through every CU:
// 512 K bytes per CU out of the XCD's L2 (it is the same matrix for every workgroup: L2 hits) — that stream is the launch time, so
// the loop is built around keeping it busy, not around MFMA duty:
//   * wave w owns output columns [64 w, 64 w + 64): its 64 weight rows are read by nobody else, so each wave runs a PRIVATE ring of
//     two 8 KB slots (64 rows x 64 k) with its own DMA issue and counted vmcnt — no workgroup barrier anywhere in the k-loop.  A slot
//     is refilled as soon as the wave's own fragment reads of it have returned (two k-tiles of every wave in flight: 64-128 KB per CU);
//   * the 32-row x panel (shared by the eight waves) sits in two 16 KB buffers of 256 k each; for K <= 512 it is resident and the
//     loop has no barrier at all, beyond that the waves meet once per 256 k;
//   * everything the tail reads from HBM (residual rows, bias, gains) is fetched into registers BEFORE the loop.
// LDS: 32 KB (x) + 8 x 16 KB (weight rings) = the CU's 160 KB; the epilogue's fp32 tile [32][512] overlays it.
// Same accumulation order as the 128x64 GEMM tile (32x32x16 MFMAs, ascending k) and the tail's own arithmetic: bit-identical to the
// two launches it replaces (tests/test_kernels_gpu.py::test_linear_tail_fwd_matches_two_launches).
#define KK_LDS_PTR(p) ((__attribute__((address_space(3))) void *)(p))
constexpr int RT_BK = 64, RT_BM = 32, RT_H = 512, RT_WAVES = 8;
constexpr int RT_SLOT = 64 * RT_BK * 2;                        // a wave's 64 weight rows x 64 k
constexpr int RT_ACH = 4;                                      // k-tiles per x chunk
constexpr int RT_A_KT = RT_BM * RT_BK * 2;                     // 4 KB: one k-tile of the x panel
constexpr int RT_A_CHUNK = RT_ACH * RT_A_KT;
constexpr int RT_LDS = 2 * RT_A_CHUNK + RT_WAVES * 2 * RT_SLOT;
constexpr int RT_TP = 520;                                     // floats per row of the epilogue tile (4 rows apart = 32 banks apart)
static_assert(RT_LDS == 163840 && RT_BM * RT_TP * 4 <= RT_LDS, "the fused Linear + tail launch takes the CU's whole LDS");

struct RowTailArgs {
    const void *x, *W;             // bf16 [rows, K] (pitch ldx), bf16 [512, K]
    int64_t ldx;
    uint32_t x_bytes, w_bytes;
    int K, y_round;
    uint64_t *trace;               // tools: [8 waves][8] wall-clock stamps of workgroup 0
    const float *bias;
    void *y_out;                   // optional (bf16): y as the backward wants it (the feed-forward's f2)
    SubOutArgs t;                  // the tail's own arguments (y unused)
};

template <int N> __device__ __forceinline__ void rt_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <typename TN>
__global__ __launch_bounds__(512) void linear_tail_fwd_kernel(RowTailArgs a) {
    __shared__ __attribute__((aligned(16))) char smem[RT_LDS];
    const SubOutArgs &t = a.t;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int64_t m0 = (int64_t)blockIdx.x * RT_BM;
    const int nk = a.K / RT_BK, nch = (nk + RT_ACH - 1) / RT_ACH;
    constexpr int H = RT_H;
#ifdef KK_TUNING_HOOKS
    uint64_t *tr = (a.trace && blockIdx.x == 0 && lane == 0) ? a.trace + wave * 8 : nullptr;
#define RT_STAMP() do { if (tr) *tr++ = wall_clock64(); } while (0)
#else
#define RT_STAMP() do { } while (0)
#endif
    RT_STAMP();

    // ---- the tail's operands: this wave finishes rows m0 + 4 wave + {0..3}, a lane columns 4 lane + {0, 256} ----
    float4 rres[4][2], bia[2], gg[2], lg[2], lb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = lane * 4 + 256 * i;
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        bia[i] = a.bias ? ld4(a.bias + c) : z;
        gg[i] = t.gain ? ld4(t.gain + c) : z;
        lg[i] = t.ln_gamma ? ld4(t.ln_gamma + c) : z;
        lb[i] = t.ln_gamma ? ld4(t.ln_beta + c) : z;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t row = m0 + 4 * wave + r;
            rres[r][i] = row < t.rows ? ld4(t.res + row * H + c) : z;
        }
    }

    // ---- DMA plumbing ----
    const __amdgpu_buffer_rsrc_t RX = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.x), 0, (int)a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t RW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.W), 0, (int)a.w_bytes, 0x00020000);
    const int c8 = lane & 7, r8 = lane >> 3;
    // x: wave w moves k-tile (w >> 1) of a chunk, rows 16 (w & 1) + 8 e + r8
    uint32_t voa[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int row = 16 * (wave & 1) + 8 * e + r8;
        voa[e] = (uint32_t)(((m0 + row) * a.ldx + (wave >> 1) * RT_BK + ((c8 ^ ((row >> 1) & 7)) << 3)) * 2);
    }
    char *adst = smem + (wave >> 1) * RT_A_KT + (wave & 1) * 2048;
    auto issue_x = [&](int ch) {
        const uint32_t so = (uint32_t)ch * (RT_ACH * RT_BK * 2);
        char *d = adst + (ch & 1) * RT_A_CHUNK;
        if (ch * RT_ACH + (wave >> 1) < nk) {                   // (wave-uniform; a K that is not a multiple of 256 has a short last chunk)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(RX, KK_LDS_PTR(d), 16, voa[0], so, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(RX, KK_LDS_PTR(d + 1024), 16, voa[1], so, 0, 0);
        } else {                                                // keep the wave's count of outstanding operations uniform
            __builtin_amdgcn_raw_ptr_buffer_load_lds(RX, KK_LDS_PTR(d), 16, 0xFFFFFFF0u, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(RX, KK_LDS_PTR(d + 1024), 16, 0xFFFFFFF0u, 0, 0, 0);
        }
    };
    // W: this wave's rows 64 w + 8 j + r8, j = 0..7
    uint32_t vob[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int row = 8 * j + r8;
        vob[j] = (uint32_t)(((int64_t)(64 * wave + row) * a.K + ((c8 ^ ((row >> 1) & 7)) << 3)) * 2);
    }
    char *ring = smem + 2 * RT_A_CHUNK + wave * (2 * RT_SLOT);
    auto issue_w = [&](int kt) {
        const uint32_t so = (uint32_t)kt * (RT_BK * 2);
        char *d = ring + (kt & 1) * RT_SLOT;
#pragma unroll
        for (int j = 0; j < 8; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(RW, KK_LDS_PTR(d + j * 1024), 16, vob[j], so, 0, 0);
    };
    // fragment reads (k-contiguous images, 128-byte rows, chunk ^ (row >> 1) & 7)
    const uint32_t fbase = (uint32_t)(l31 * (RT_BK * 2));
    uint32_t fx[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fx[ks] = fbase + (uint32_t)(((2 * ks + half) ^ ((l31 >> 1) & 7)) << 4);
    const uint32_t a_lds = (uint32_t)(uintptr_t)KK_LDS_PTR(smem), w_lds = (uint32_t)(uintptr_t)KK_LDS_PTR(ring);

    issue_x(0);
    if (nch > 1) issue_x(1);
    issue_w(0);
    if (nk > 1) issue_w(1);
    if (nk > 1) rt_wait_vm<8>(); else rt_wait_vm<0>();
    RT_STAMP();
    __builtin_amdgcn_s_barrier();                               // the x chunks of every wave have landed
    asm volatile("" ::: "memory");
    RT_STAMP();

    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    bool x_issued = false;                                      // an x chunk was issued in the previous iteration (2 operations in the FIFO)
    for (int kt = 0; kt < nk; ++kt) {
        if (kt > 0) {                                           // this wave's tile kt has landed: only younger operations may be outstanding
            const bool more = kt + 1 < nk;
            if (x_issued) { if (more) rt_wait_vm<10>(); else rt_wait_vm<2>(); }
            else { if (more) rt_wait_vm<8>(); else rt_wait_vm<0>(); }
        }
        x_issued = false;
        const int ch = kt / RT_ACH;
        if (kt > 0 && kt % RT_ACH == 0 && (ch >= 2 || ch + 1 < nch)) {
            __builtin_amdgcn_s_barrier();                       // chunk ch has landed for everyone; everyone is done with chunk ch - 1
            asm volatile("" ::: "memory");
            if (ch + 1 < nch) { issue_x(ch + 1); x_issued = true; }
        }
        const uint32_t ai = a_lds + (uint32_t)((ch & 1) * RT_A_CHUNK + (kt % RT_ACH) * RT_A_KT);
        const uint32_t bi = w_lds + (uint32_t)((kt & 1) * RT_SLOT);
        bf16x8 af[4], bf[4][2];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            asm volatile("ds_read_b128 %0, %1" : "=v"(af[ks]) : "v"(ai + fx[ks]));
            asm volatile("ds_read_b128 %0, %1" : "=v"(bf[ks][0]) : "v"(bi + fx[ks]));
            asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(bf[ks][1]) : "v"(bi + fx[ks]));
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks == 0) asm volatile("s_waitcnt lgkmcnt(9)" ::: "memory");
            else if (ks == 1) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
            else if (ks == 2) asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
            else {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (kt + 2 < nk) issue_w(kt + 2);               // the slot is free: every read of it has returned
            }
            asm volatile("" : "+v"(af[ks]), "+v"(bf[ks][0]), "+v"(bf[ks][1]));
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks], bf[ks][0], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks], bf[ks][1], acc[1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- the tile goes through LDS: accumulator layout (a lane = one column) -> a wave per row ----
    RT_STAMP();
    __builtin_amdgcn_s_barrier();                               // every wave is through with the panels
    asm volatile("" ::: "memory");
    float *tile = reinterpret_cast<float *>(smem);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) tile[frag_row(r, half) * RT_TP + 64 * wave + 32 * j + l31] = acc[j][r];
    __syncthreads();
    RT_STAMP();

    // The wave's four rows side by side: every step below is four independent chains (a row is three dependent wave reductions —
    // one after the other they were 6 us of this launch, 1.5 us per row at two waves per SIMD).  Rows past the end compute on zeros
    // and store nothing.
    const uint32_t seed = *t.d.seed;
    const uint32_t t1 = kk_drop_threshold(t.d.p1), t2 = kk_drop_threshold(t.d.p2);
    const float k1 = t.d.p1 > 0.f ? 1.f / (1.f - t.d.p1) : 1.f, k2 = t.d.p2 > 0.f ? 1.f / (1.f - t.d.p2) : 1.f;
    auto wave_sum4 = [](float (&x)[4]) {                        // wave_sum of each: four independent DPP chains
#pragma unroll
        for (int r = 0; r < 4; ++r) x[r] = wave_sum(x[r]);
    };
    const int64_t row0 = m0 + 4 * wave;
    float4 v[4][2];
    float q[4], rs[4], dp[4], sm[4], mean[4], qq[4], rstd[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t row = row0 + r;
        q[r] = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = lane * 4 + 256 * i;
            const float4 s4 = ld4(tile + (4 * wave + r) * RT_TP + c);
            float o[4] = {s4.x + bia[i].x, s4.y + bia[i].y, s4.z + bia[i].z, s4.w + bia[i].w};
            if (a.y_round) {
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (float)(__bf16)o[e];
            }
            v[r][i] = make_float4(o[0], o[1], o[2], o[3]);
            if (a.y_out && row < t.rows) stv4_out<__bf16>(static_cast<__bf16 *>(a.y_out) + row * H + c, v[r][i], t.wt);
            q[r] += o[0] * o[0] + o[1] * o[1] + o[2] * o[2] + o[3] * o[3];
        }
        rs[r] = 1.f;
        dp[r] = row_scale(t.d, seed, row);
    }
    if (t.gain) {
        wave_sum4(q);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            rs[r] = 1.f / sqrtf(q[r] / (float)t.H + FLT_EPSILON);
            if (lane == 0 && row0 + r < t.rows) t.rstd_f[row0 + r] = rs[r];
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t row = row0 + r;
        sm[r] = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = lane * 4 + 256 * i;
            float o[4] = {v[r][i].x, v[r][i].y, v[r][i].z, v[r][i].w};
            if (t.gain) {
                const float4 g = gg[i];
                o[0] = o[0] * rs[r] * g.x; o[1] = o[1] * rs[r] * g.y; o[2] = o[2] * rs[r] * g.z; o[3] = o[3] * rs[r] * g.w;
            }
            const float rr[4] = {rres[r][i].x, rres[r][i].y, rres[r][i].z, rres[r][i].w};
            float m1[4], m2[4];
            kk_drop_mul4(seed, t.d.site1, (uint64_t)row * H + c, t1, k1, m1);
            kk_drop_mul4(seed, t.d.site2, (uint64_t)row * H + c, t2, k2, m2);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = o[e] * (dp[r] * m1[e] * m2[e]) + rr[e];
            v[r][i] = make_float4(o[0], o[1], o[2], o[3]);
            if (row < t.rows) st4_out(t.x_out + row * H + c, v[r][i], t.wt);
            sm[r] += o[0] + o[1] + o[2] + o[3];
        }
    }
    if (t.ln_gamma) {
        wave_sum4(sm);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            mean[r] = sm[r] / (float)t.H;
            qq[r] = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float e0 = v[r][i].x - mean[r], e1 = v[r][i].y - mean[r], e2 = v[r][i].z - mean[r], e3 = v[r][i].w - mean[r];
                qq[r] += e0 * e0 + e1 * e1 + e2 * e2 + e3 * e3;
            }
        }
        wave_sum4(qq);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t row = row0 + r;
            rstd[r] = 1.f / sqrtf(qq[r] / (float)t.H + 1e-5f);
            if (row >= t.rows) continue;
            TN *nr = static_cast<TN *>(t.n) + row * H;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int c = lane * 4 + 256 * i;
                const float4 g = lg[i], b = lb[i];
                stv4_out<TN>(nr + c, make_float4((v[r][i].x - mean[r]) * rstd[r] * g.x + b.x, (v[r][i].y - mean[r]) * rstd[r] * g.y + b.y,
                                             (v[r][i].z - mean[r]) * rstd[r] * g.z + b.z, (v[r][i].w - mean[r]) * rstd[r] * g.w + b.w), t.wt);
            }
            if (lane == 0) {
                t.mean[row] = mean[r];
                t.rstd[row] = rstd[r];
            }
        }
    }
    RT_STAMP();
#ifdef KK_TUNING_HOOKS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RT_STAMP();
#endif
#undef RT_STAMP
}
#ifdef KK_TUNING_HOOKS
uint64_t *g_rt_trace = nullptr;
#endif

// ---- fused sub-layer tail (backward) ------------------------------------------------------------------------------
// The exact reverse of sublayer_out_fwd_kernel, one wave per row:
//   g   = (accumulate ? dres : 0) + LayerNorm_backward(dn; x_out, gamma, mean, rstd)      -> dres (gradient of the stream)
//   dz  = g * (the forward's masks)                                                        (kk_dropout_bwd)
//   FFN : dy = RMSNorm_backward(dz; y, gain, rstd_f)  (kk_rmsnorm_bwd)      attention: dy = dz
// and the four column reductions of the tail — LayerNorm gain / bias, RMSNorm gain, and the column sums of dy (the bias
// gradient of the Linear that produced y) — leave the workgroup as one row of partials[blockIdx.x][4][H]
// (dgamma | dbeta | dbias | dgain) for kk_partials_reduce.  Replaces 3-4 launches per sub-layer.
struct SubInArgs {
    const void *dn;                // gradient of the LayerNorm output (TN)
    const float *x_out, *ln_gamma, *mean, *rstd;
    float *dres;                   // residual-stream gradient, updated in place
    int accumulate;
    const void *y;                 // FFN: the RMSNorm input saved by the forward (TY); attention: unused
    const float *gain, *rstd_f;    // FFN only (gain == nullptr: attention)
    void *dy;                      // out (TY): gradient of y
    float *partials;               // [gridDim.x][4][H]
    int64_t rows;
    int H;
    int wt;                        // write-through output stores (kk_write_through(rows))
    DropArgs d;
};

// SIB_WAVES waves per workgroup, RIF rows IN FLIGHT per wave; 8 waves (two per SIMD hide each other's load latency) while the
// per-wave column-sum slabs fit 64 KB of LDS (H <= 512), fewer above.
// The column reductions want few, fat workgroups (one partial row each for kk_partials_reduce), so a wave walks several rows —
// and a row is one dependent round trip to HBM (loads -> two wave reductions -> stores).  Walking them one after the other
// made the launch a chain of round trips: 10.4 us for 23.5 MB at 4096 x 512 (2.3 TB/s, profiles/r02h).  RIF = 2: the loads of
// BOTH rows of a wave are issued before anything is reduced, so the launch is one round trip with twice the bytes in flight.
template <typename TN, typename TY, int NV, int SIB_WAVES, int RIF>
__global__ __launch_bounds__(64 * SIB_WAVES) void sublayer_in_bwd_kernel(SubInArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];   // [waves][4][H]: per-wave column sums, no LDS atomics
    const int lane = threadIdx.x & 63, H = a.H;
    const uint32_t seed = *a.d.seed;
    const uint32_t t1 = kk_drop_threshold(a.d.p1), t2 = kk_drop_threshold(a.d.p2);
    const float k1 = a.d.p1 > 0.f ? 1.f / (1.f - a.d.p1) : 1.f, k2 = a.d.p2 > 0.f ? 1.f / (1.f - a.d.p2) : 1.f;
    const bool ffn = a.gain != nullptr;
    float4 ag[NV], ab[NV], ac[NV], an[NV], gm[NV], gn[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane * 4 + 256 * i;
        ag[i] = ab[i] = ac[i] = an[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        gm[i] = c < H ? ld4(a.ln_gamma + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        gn[i] = (ffn && c < H) ? ld4(a.gain + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float invH = 1.f / (float)H;
    const TN *dn = static_cast<const TN *>(a.dn);
    const TY *yy = static_cast<const TY *>(a.y);
    TY *dy = static_cast<TY *>(a.dy);
    const int64_t stride = (int64_t)gridDim.x * SIB_WAVES;
    for (int64_t row0 = (int64_t)blockIdx.x * SIB_WAVES + (threadIdx.x >> 6); row0 < a.rows; row0 += stride * RIF) {
        float4 xh[RIF][NV], dg[RIF][NV], old[RIF][NV], yv[RIF][NV];
        float mu[RIF], rs[RIF], rsf[RIF];
#pragma unroll
        for (int r = 0; r < RIF; ++r) {                    // all loads of all rows in flight first
            const int64_t row = row0 + r * stride;
            const bool live = row < a.rows;
            mu[r] = live ? a.mean[row] : 0.f;
            rs[r] = live ? a.rstd[row] : 0.f;
            rsf[r] = (live && ffn) ? a.rstd_f[row] : 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = lane * 4 + 256 * i;
                const bool ok = live && c < H;
                const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                xh[r][i] = ok ? ld4(a.x_out + row * H + c) : z;
                dg[r][i] = ok ? ldv4<TN>(dn + row * H + c) : z;
                old[r][i] = (ok && a.accumulate) ? ld4(a.dres + row * H + c) : z;
                yv[r][i] = (ok && ffn) ? ldv4<TY>(yy + row * H + c) : z;
            }
        }
#pragma unroll
        for (int r = 0; r < RIF; ++r) {
            const int64_t row = row0 + r * stride;
            if (row >= a.rows) break;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const float4 d = dg[r][i];
                float4 &x = xh[r][i];
                x = make_float4((x.x - mu[r]) * rs[r], (x.y - mu[r]) * rs[r], (x.z - mu[r]) * rs[r], (x.w - mu[r]) * rs[r]);
                if (lane * 4 + 256 * i >= H) x = make_float4(0.f, 0.f, 0.f, 0.f);
                ag[i].x += d.x * x.x; ag[i].y += d.y * x.y; ag[i].z += d.z * x.z; ag[i].w += d.w * x.w;
                ab[i].x += d.x; ab[i].y += d.y; ab[i].z += d.z; ab[i].w += d.w;
                dg[r][i] = make_float4(d.x * gm[i].x, d.y * gm[i].y, d.z * gm[i].z, d.w * gm[i].w);
                s1 += dg[r][i].x + dg[r][i].y + dg[r][i].z + dg[r][i].w;
                s2 += dg[r][i].x * x.x + dg[r][i].y * x.y + dg[r][i].z * x.z + dg[r][i].w * x.w;
            }
            s1 = wave_sum(s1) * invH;
            s2 = wave_sum(s2) * invH;
            const float dp = row_scale(a.d, seed, row);
            float sk = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = lane * 4 + 256 * i;
                if (c < H) {
                    const float4 dgi = dg[r][i], xi = xh[r][i], oi = old[r][i];
                    float g[4] = {oi.x + rs[r] * (dgi.x - s1 - xi.x * s2), oi.y + rs[r] * (dgi.y - s1 - xi.y * s2),
                                  oi.z + rs[r] * (dgi.z - s1 - xi.z * s2), oi.w + rs[r] * (dgi.w - s1 - xi.w * s2)};
                    st4_out(a.dres + row * H + c, make_float4(g[0], g[1], g[2], g[3]), a.wt);
                    float m1[4], m2[4];
                    kk_drop_mul4(seed, a.d.site1, (uint64_t)row * H + c, t1, k1, m1);
                    kk_drop_mul4(seed, a.d.site2, (uint64_t)row * H + c, t2, k2, m2);
#pragma unroll
                    for (int e = 0; e < 4; ++e) g[e] *= dp * m1[e] * m2[e];
                    if (ffn) {                                  // dz -> RMSNorm backward (second pass below needs the row sum)
                        const float4 y4 = yv[r][i];
                        an[i].x += g[0] * y4.x * rsf[r]; an[i].y += g[1] * y4.y * rsf[r]; an[i].z += g[2] * y4.z * rsf[r]; an[i].w += g[3] * y4.w * rsf[r];
                        g[0] *= gn[i].x; g[1] *= gn[i].y; g[2] *= gn[i].z; g[3] *= gn[i].w;
                        sk += g[0] * y4.x + g[1] * y4.y + g[2] * y4.z + g[3] * y4.w;
                    }
                    dg[r][i] = make_float4(g[0], g[1], g[2], g[3]);     // reuse: dz (attention) or dz*gain (FFN)
                }
            }
            const float k = ffn ? wave_sum(sk) * invH * rsf[r] * rsf[r] * rsf[r] : 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = lane * 4 + 256 * i;
                if (c < H) {
                    float4 o = dg[r][i];
                    if (ffn) o = make_float4(rsf[r] * o.x - yv[r][i].x * k, rsf[r] * o.y - yv[r][i].y * k, rsf[r] * o.z - yv[r][i].z * k, rsf[r] * o.w - yv[r][i].w * k);
                    stv4_out<TY>(dy + row * H + c, o, a.wt);
                    ac[i].x += o.x; ac[i].y += o.y; ac[i].z += o.z; ac[i].w += o.w;
                }
            }
        }
    }
    float *mine = sm + (threadIdx.x >> 6) * 4 * H;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane * 4 + 256 * i;
        if (c < H) {
            st4(mine + c, ag[i]); st4(mine + H + c, ab[i]); st4(mine + 2 * H + c, ac[i]); st4(mine + 3 * H + c, an[i]);
        }
    }
    __syncthreads();
    for (int c = threadIdx.x * 4; c < 4 * H; c += 256 * SIB_WAVES) {
        float4 t = ld4(sm + c);
#pragma unroll
        for (int w = 1; w < SIB_WAVES; ++w) {
            const float4 u = ld4(sm + w * 4 * H + c);
            t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
        }
        st4(a.partials + (int64_t)blockIdx.x * 4 * H + c, t);
    }
}

template <typename TN, typename TY>
void launch_subin(const SubInArgs &a, int blocks, hipStream_t s) {
    const dim3 grid(blocks);
    const int nv = kk_cdiv(a.H, 256);
#define KK_SIB(NV, WV, RIF) hipLaunchKernelGGL((sublayer_in_bwd_kernel<TN, TY, NV, WV, RIF>), grid, dim3(64 * WV), (size_t)WV * 4 * a.H * sizeof(float), s, a)
    // RIF = 2 (both rows of a wave in flight at once, 213 registers): when a wave walks exactly two rows the launch becomes ONE round trip
    // to HBM.  Round 3 measured it slower in-step (4.165 vs 4.145 ms); re-measured in round 5 (interleaved, tools flavour): 8 x 512 (4096
    // rows, two per wave) 3.794 / 3.776 -> 3.746 / 3.759 ms, 8 x 1024 (four per wave: two trips either way, more registers) 6.217 -> 6.268 ms.
    // Taken where it is one trip; KK_SIB_RIF (tools) forces 1 or 2.
    static const int rif_env = kk_tune_env("KK_SIB_RIF", 0);
    const int rif2 = rif_env ? rif_env : (a.rows <= (int64_t)2 * blocks * 8 ? 2 : 1);
    if (nv <= 1) { if (rif2 == 2) KK_SIB(1, 8, 2); else KK_SIB(1, 8, 1); }
    else if (nv <= 2) { if (rif2 == 2) KK_SIB(2, 8, 2); else KK_SIB(2, 8, 1); }
    else if (nv <= 4) KK_SIB(4, 4, 1);
    else KK_SIB(8, 2, 1);
#undef KK_SIB
}

// SpecAugment: per sample `nt` time masks of t in [0, time_limit) frames at t0 in [0, max(1, T - t)) and `nf`
// feature masks of f in [0, max(1, fmax)) dims at f0 in [0, max(1, H - f)); masked positions are zeroed (in place).
template <typename TX>
__global__ __launch_bounds__(256) void specaug_kernel(TX *__restrict__ x, int64_t total4, int T, int H, const uint32_t *__restrict__ seedp,
                                                      uint32_t site, int tmax, int fmax, int nt, int nf) {
    const uint32_t seed = *seedp;
    const int time_limit = max(1, min(tmax, T / 4));
    const int H4 = H / 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / H4;
        const int c = (int)(i - row * H4) * 4;
        const int b = (int)(row / T), t = (int)(row - (int64_t)b * T);
        bool tm = false;
        bool fm[4] = {false, false, false, false};
        for (int k = 0; k < nt; ++k) {
            const int len = (int)(kk_hash(seed, site, (uint64_t)b * 64 + 2 * k) % (uint32_t)time_limit);
            const int t0 = (int)(kk_hash(seed, site, (uint64_t)b * 64 + 2 * k + 1) % (uint32_t)max(1, T - len));
            tm |= (t >= t0 && t < t0 + len);
        }
        for (int k = 0; k < nf; ++k) {
            const int len = (int)(kk_hash(seed, site, (uint64_t)b * 64 + 32 + 2 * k) % (uint32_t)max(1, fmax));
            const int f0 = (int)(kk_hash(seed, site, (uint64_t)b * 64 + 32 + 2 * k + 1) % (uint32_t)max(1, H - len));
#pragma unroll
            for (int e = 0; e < 4; ++e) fm[e] |= (c + e >= f0 && c + e < f0 + len);
        }
        if (tm || fm[0] || fm[1] || fm[2] || fm[3]) {
            float4 v = ldv4<TX>(x + row * H + c);
            if (tm || fm[0]) v.x = 0.f;
            if (tm || fm[1]) v.y = 0.f;
            if (tm || fm[2]) v.z = 0.f;
            if (tm || fm[3]) v.w = 0.f;
            stv4<TX>(x + row * H + c, v);
        }
    }
}

int launch_dropout(const float *x, const float *res, int64_t res_mod, float *out, int out_bf16, int64_t rows, int H, int S,
                   const uint32_t *seed, uint32_t site1, float p1, uint32_t site2, float p2, uint32_t site_dp, float dp_rate,
                   hipStream_t s, const char *name) {
    KK_REQUIRE(rows > 0 && H > 0 && H % 4 == 0 && S > 0 && seed, "%s: bad args", name);
    KK_REQUIRE(p1 >= 0.f && p1 < 1.f && p2 >= 0.f && p2 < 1.f && dp_rate >= 0.f && dp_rate < 1.f, "%s: probabilities must be in [0,1)", name);
    DropArgs d = {seed, site1, site2, site_dp, p1, p2, dp_rate, S};
    const int64_t total4 = rows * H / 4;
    int blocks = kk_cdiv(total4, 256);
    if (blocks > 4096) blocks = 4096;
    if (out_bf16) hipLaunchKernelGGL(dropout_kernel<__bf16>, dim3(blocks), dim3(256), 0, s, x, res, res_mod, reinterpret_cast<__bf16 *>(out), total4, H, d);
    else hipLaunchKernelGGL(dropout_kernel<float>, dim3(blocks), dim3(256), 0, s, x, res, res_mod, out, total4, H, d);
    KK_LAUNCH_CHECK(name);
    return 0;
}

}  // namespace

extern "C" int kk_dropout_fwd(const float *x, const float *res, int64_t res_mod, float *out, int64_t rows, int H, int S,
                              const uint32_t *seed, uint32_t site1, float p1, uint32_t site2, float p2, uint32_t site_dp,
                              float dp_rate, void *stream) {
    return launch_dropout(x, res, res_mod, out, 0, rows, H, S, seed, site1, p1, site2, p2, site_dp, dp_rate, (hipStream_t)stream,
                          "kk_dropout_fwd");
}

extern "C" int kk_sublayer_out_fwd(const float *y, int y_bf16, const float *gain, float *rstd_f, const float *res, float *x_out,
                                   const float *ln_gamma, const float *ln_beta, float *n, int n_bf16, float *mean, float *rstd,
                                   int64_t rows, int H, int S, const uint32_t *seed, uint32_t site1, float p1, uint32_t site2,
                                   float p2, uint32_t site_dp, float dp_rate, void *stream) {
    KK_REQUIRE(y && res && x_out && seed && rows > 0 && H > 0 && H % 4 == 0 && H <= 2048 && S > 0, "kk_sublayer_out_fwd: bad args");
    KK_REQUIRE(!gain || rstd_f, "kk_sublayer_out_fwd: the RMSNorm needs rstd_f");
    KK_REQUIRE(!ln_gamma || (ln_beta && n && mean && rstd), "kk_sublayer_out_fwd: the LayerNorm needs beta, n, mean, rstd");
    KK_REQUIRE(p1 >= 0.f && p1 < 1.f && p2 >= 0.f && p2 < 1.f && dp_rate >= 0.f && dp_rate < 1.f, "kk_sublayer_out_fwd: probabilities must be in [0,1)");
    SubOutArgs a;
    a.y = y; a.gain = gain; a.rstd_f = rstd_f; a.res = res; a.x_out = x_out; a.ln_gamma = ln_gamma; a.ln_beta = ln_beta; a.n = n;
    a.mean = mean; a.rstd = rstd; a.rows = rows; a.H = H; a.wt = kk_write_through(rows);
    a.d = {seed, site1, site2, site_dp, p1, p2, dp_rate, S};
    hipStream_t s = (hipStream_t)stream;
    if (y_bf16) { if (n_bf16) launch_subout<__bf16, __bf16>(a, s); else launch_subout<__bf16, float>(a, s); }
    else { if (n_bf16) launch_subout<float, __bf16>(a, s); else launch_subout<float, float>(a, s); }
    KK_LAUNCH_CHECK("kk_sublayer_out_fwd");
    return 0;
}

#ifdef KK_TUNING_HOOKS
extern "C" int kk_linear_tail_trace(void *buf) { g_rt_trace = static_cast<uint64_t *>(buf); return 0; }      // tools: 64 uint64 stamps of workgroup 0
#endif
extern "C" int kk_linear_tail_supported(int64_t rows, int H, int K) {
    static const int on = kk_tune_env("KK_LINEAR_TAIL", 1);
    return on && rows > 0 && rows <= (int64_t)1 << 21 && H == RT_H && K >= RT_BK && K % RT_BK == 0 && K <= 8192;
}

// Does the one-launch form beat kk_gemm + kk_sublayer_out_fwd at this shape?  Measured on MI355X (tools/probes/linear_tail_bench.py,
// dependent chains in a replayed graph; profiles/r05_linear_tail_bench.txt): a round of <= one workgroup per CU takes ~18.5 us at
// K = 512 whatever its row count (the weight matrix streams through every CU at ~54 B/clk, then every CU writes its rows at once),
// the two launches 4.5 us + 2.44 us per 1000 rows; K >= 1536 streams 3-4 x the bytes per CU and always loses (x 1.15 - 1.45).
extern "C" int kk_linear_tail_pays(int64_t rows, int H, int K) {
    if (!kk_linear_tail_supported(rows, H, K) || K != 512) return 0;
    int dev = 0, cus = 0;
    static const int ncu = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) ? cus : 256;
    const int64_t rounds = (kk_cdiv(rows, RT_BM) + ncu - 1) / ncu;
    return 18.5 * (double)rounds + 0.5 < 4.5 + 2.44e-3 * (double)rows;
}

extern "C" int kk_linear_tail_fwd(const void *x, int64_t ldx, const void *W, const float *bias, int K, void *y_out, int y_round,
                                  const float *gain, float *rstd_f, const float *res, float *x_out, const float *ln_gamma,
                                  const float *ln_beta, float *n, int n_bf16, float *mean, float *rstd, int64_t rows, int H, int S,
                                  const uint32_t *seed, uint32_t site1, float p1, uint32_t site2, float p2, uint32_t site_dp,
                                  float dp_rate, void *stream) {
    KK_REQUIRE(kk_linear_tail_supported(rows, H, K), "kk_linear_tail_fwd: unsupported shape rows=%lld H=%d K=%d (H = 512, K a multiple of 64)",
               (long long)rows, H, K);
    KK_REQUIRE(x && W && res && x_out && seed && S > 0 && ldx >= K && ldx % 8 == 0, "kk_linear_tail_fwd: bad args");
    KK_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(W) & 15) == 0, "kk_linear_tail_fwd: operands must be 16-byte aligned");
    KK_REQUIRE(!gain || rstd_f, "kk_linear_tail_fwd: the RMSNorm needs rstd_f");
    KK_REQUIRE(!ln_gamma || (ln_beta && n && mean && rstd), "kk_linear_tail_fwd: the LayerNorm needs beta, n, mean, rstd");
    KK_REQUIRE(p1 >= 0.f && p1 < 1.f && p2 >= 0.f && p2 < 1.f && dp_rate >= 0.f && dp_rate < 1.f, "kk_linear_tail_fwd: probabilities must be in [0,1)");
    const int64_t xb = ((rows - 1) * ldx + K) * 2;
    KK_REQUIRE(xb < ((int64_t)1 << 31), "kk_linear_tail_fwd: x spans more than 2 GB");
    RowTailArgs a;
    a.x = x; a.W = W; a.ldx = ldx; a.x_bytes = (uint32_t)xb; a.w_bytes = (uint32_t)((int64_t)RT_H * K * 2); a.K = K; a.y_round = y_round;
    a.trace = nullptr;
#ifdef KK_TUNING_HOOKS
    a.trace = g_rt_trace;
#endif
    a.bias = bias; a.y_out = y_out;
    SubOutArgs &t = a.t;
    t.y = nullptr; t.gain = gain; t.rstd_f = rstd_f; t.res = res; t.x_out = x_out; t.ln_gamma = ln_gamma; t.ln_beta = ln_beta; t.n = n;
    t.mean = mean; t.rstd = rstd; t.rows = rows; t.H = H; t.wt = kk_write_through(rows);
    t.d = {seed, site1, site2, site_dp, p1, p2, dp_rate, S};
    const dim3 grid(kk_cdiv(rows, RT_BM));
    kk_note_kernel("linear_tail_fwd");
    if (n_bf16) hipLaunchKernelGGL((linear_tail_fwd_kernel<__bf16>), grid, dim3(512), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((linear_tail_fwd_kernel<float>), grid, dim3(512), 0, (hipStream_t)stream, a);
    KK_LAUNCH_CHECK("kk_linear_tail_fwd");
    return 0;
}

extern "C" int kk_sublayer_in_bwd_blocks(int64_t rows) {
    int blocks = kk_cdiv(rows, 16);                       // the column-sum epilogue is per workgroup: few, fat workgroups
    return blocks > 256 ? 256 : (blocks < 1 ? 1 : blocks);
}

extern "C" int kk_sublayer_in_bwd(const float *dn, int dn_bf16, const float *x_out, const float *ln_gamma, const float *mean,
                                  const float *rstd, float *dres, int accumulate, const float *y, const float *gain,
                                  const float *rstd_f, float *dy, int y_bf16, float *partials, int64_t rows, int H, int S,
                                  const uint32_t *seed, uint32_t site1, float p1, uint32_t site2, float p2, uint32_t site_dp,
                                  float dp_rate, void *stream) {
    KK_REQUIRE(dn && x_out && ln_gamma && mean && rstd && dres && dy && partials && seed && rows > 0 && H > 0 && H % 4 == 0 &&
                   H <= 2048 && S > 0, "kk_sublayer_in_bwd: bad args");
    KK_REQUIRE(!gain || (y && rstd_f), "kk_sublayer_in_bwd: the RMSNorm backward needs y and rstd_f");
    SubInArgs a;
    a.dn = dn; a.x_out = x_out; a.ln_gamma = ln_gamma; a.mean = mean; a.rstd = rstd; a.dres = dres; a.accumulate = accumulate;
    a.y = y; a.gain = gain; a.rstd_f = rstd_f; a.dy = dy; a.partials = partials; a.rows = rows; a.H = H; a.wt = kk_write_through(rows);
    a.d = {seed, site1, site2, site_dp, p1, p2, dp_rate, S};
    hipStream_t s = (hipStream_t)stream;
    const int blocks = kk_sublayer_in_bwd_blocks(rows);
    if (dn_bf16) { if (y_bf16) launch_subin<__bf16, __bf16>(a, blocks, s); else launch_subin<__bf16, float>(a, blocks, s); }
    else { if (y_bf16) launch_subin<float, __bf16>(a, blocks, s); else launch_subin<float, float>(a, blocks, s); }
    KK_LAUNCH_CHECK("kk_sublayer_in_bwd");
    return 0;
}

extern "C" int kk_dropout_bwd(const float *dy, float *dx, int64_t rows, int H, int S, const uint32_t *seed, uint32_t site1,
                              float p1, uint32_t site2, float p2, uint32_t site_dp, float dp_rate, int dx_bf16, void *stream) {
    return launch_dropout(dy, nullptr, 0, dx, dx_bf16, rows, H, S, seed, site1, p1, site2, p2, site_dp, dp_rate, (hipStream_t)stream,
                          "kk_dropout_bwd");
}

extern "C" int kk_specaug(float *x, int B, int T, int H, const uint32_t *seed, uint32_t site, int time_mask_max,
                          int feat_mask_max, int n_time, int n_feat, int x_bf16, void *stream) {
    KK_REQUIRE(x && seed && B > 0 && T > 0 && H > 0 && H % 4 == 0, "kk_specaug: bad args");
    KK_REQUIRE(n_time >= 0 && n_time <= 16 && n_feat >= 0 && n_feat <= 16, "kk_specaug: at most 16 masks of each kind");
    const int64_t total4 = (int64_t)B * T * H / 4;
    int blocks = kk_cdiv(total4, 256);
    if (blocks > 4096) blocks = 4096;
    if (x_bf16)
        hipLaunchKernelGGL(specaug_kernel<__bf16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<__bf16 *>(x), total4,
                           T, H, seed, site, time_mask_max, feat_mask_max, n_time, n_feat);
    else
        hipLaunchKernelGGL(specaug_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, total4, T, H, seed, site,
                           time_mask_max, feat_mask_max, n_time, n_feat);
    KK_LAUNCH_CHECK("kk_specaug");
    return 0;
}
#endif  // KK_BODIES_ONLY
