// Dropout, DropPath (stochastic depth) and SpecAugment for the training forward/backward.
//
//  residual dropout + DropPath   model/transformers.py:16-40 (drop_path), :482-487, :569-581: x + dropout(drop_path(y))
//  FFN output dropout            transformers.py:111 (a second dropout in front of the residual one)
//  decoder-input dropouts        model/model.py:525-531 (functional dropout, then PE add, then PE dropout)
//  SpecAugment on the memory     training/trainer.py:1577-1604, applied at model/model.py:636-639
// Masks come from the counter RNG in kk_common.h; the backward kernels regenerate them.  These are the p > 0 paths;
// with p == 0 the engine uses the fused GEMM / RMSNorm residual epilogues instead.
#include "kk_common.h"

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
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint64_t idx = (uint64_t)row * H + c + e;
            o[e] *= rs * kk_drop_mul(seed, d.site1, idx, t1, k1) * kk_drop_mul(seed, d.site2, idx, t2, k2);
        }
        if (res) {
            const int64_t rr = res_mod > 0 ? row % res_mod : row;
            const float4 r = ld4(res + rr * H + c);
            o[0] += r.x; o[1] += r.y; o[2] += r.z; o[3] += r.w;
        }
        stv4<TO>(out + row * H + c, make_float4(o[0], o[1], o[2], o[3]));
    }
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
