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

template <typename TY, typename TN, int NV>
__global__ __launch_bounds__(256) void sublayer_out_fwd_kernel(SubOutArgs a) {
    const int lane = threadIdx.x & 63, H = a.H;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.rows) return;
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

template <typename TY, typename TN>
void launch_subout(const SubOutArgs &a, hipStream_t s) {
    const dim3 grid(kk_cdiv(a.rows, 4)), blk(256);
    const int nv = kk_cdiv(a.H, 256);
    if (nv <= 1) hipLaunchKernelGGL((sublayer_out_fwd_kernel<TY, TN, 1>), grid, blk, 0, s, a);
    else if (nv <= 2) hipLaunchKernelGGL((sublayer_out_fwd_kernel<TY, TN, 2>), grid, blk, 0, s, a);
    else if (nv <= 4) hipLaunchKernelGGL((sublayer_out_fwd_kernel<TY, TN, 4>), grid, blk, 0, s, a);
    else hipLaunchKernelGGL((sublayer_out_fwd_kernel<TY, TN, 8>), grid, blk, 0, s, a);
}

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
