// Fused loss reduction + closed-form loss gradients.
//
// Restates calculate_training_losses (reference training/losses.py:9-216; criteria training/trainer.py:410-444):
//   mel    mean over (valid frame x mel bin, finite) of |pred - target|                       (:37-46)
//   dur    Huber(delta) on (pred, log(d + 1)) over pos < phoneme_len and d > 0                (:48,82-98)
//   stop   BCEWithLogits(pos_weight) over valid frames, finite                               (:100-105)
//   pitch / energy  Huber(delta) over valid frames, finite                                   (:107-156)
//   clamp to 100/100/100/10/10 AFTER the mean, weighted sum                                  (:195-207)
// The reference does five boolean-mask gathers (each a host sync); here one streaming kernel produces 5 sums
// and 5 counts in fp64 accumulators, a one-thread kernel finishes the scalars and the per-element gradient
// coefficients, and one streaming kernel writes the five gradient tensors.  No host round trip.
#include "kk_common.h"
#include <math.h>

namespace {

struct LossArgs {
    const float *mel_pred, *mel_tgt, *dur_pred, *stop_logit, *stop_tgt, *pitch_pred, *pitch_tgt, *energy_pred, *energy_tgt;
    const int64_t *dur, *mel_len, *ph_len;
    int B, T, P, M;
    KkLossCfg cfg;
};

__device__ __forceinline__ float huber(float e, float delta) {
    const float a = fabsf(e);
    return a <= delta ? 0.5f * e * e : delta * (a - 0.5f * delta);
}
__device__ __forceinline__ float huber_grad(float e, float delta) { return fabsf(e) <= delta ? e : (e > 0.f ? delta : -delta); }
__device__ __forceinline__ float log_sigmoid(float z) { return fminf(z, 0.f) - log1pf(expf(-fabsf(z))); }
__device__ __forceinline__ float bce_logits(float z, float y, float pw) { return -(pw * y * log_sigmoid(z) + (1.f - y) * log_sigmoid(-z)); }

__global__ __launch_bounds__(256) void losses_fwd_kernel(LossArgs a, double *__restrict__ acc) {
    __shared__ double red[4];
    float s[5] = {0.f, 0.f, 0.f, 0.f, 0.f}, n[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    float bad = 0.f;       // non-finite prediction elements, padded positions included (finite-output guard, trainer.py:3233-3256)
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, stride = (int64_t)gridDim.x * 256;
    const int64_t nmel = (int64_t)a.B * a.T * a.M, nfr = (int64_t)a.B * a.T, nph = (int64_t)a.B * a.P;
    for (int64_t i = tid; i < nmel; i += stride) {
        const int64_t fr = i / a.M;
        const int b = (int)(fr / a.T), t = (int)(fr - (int64_t)b * a.T);
        const float pr = a.mel_pred[i];
        if (!isfinite(pr)) bad += 1.f;
        if (t < a.mel_len[b]) {
            const float v = fabsf(pr - a.mel_tgt[i]);
            if (isfinite(v)) { s[0] += v; n[0] += 1.f; }
        }
    }
    for (int64_t i = tid; i < nph; i += stride) {
        const int b = (int)(i / a.P), p = (int)(i - (int64_t)b * a.P);
        const int64_t d = a.dur[i];
        if (!isfinite(a.dur_pred[i])) bad += 1.f;
        if (p < a.ph_len[b] && d > 0) { s[1] += huber(a.dur_pred[i] - logf((float)d + 1.f), a.cfg.delta_dur); n[1] += 1.f; }
    }
    for (int64_t i = tid; i < nfr; i += stride) {
        const int b = (int)(i / a.T), t = (int)(i - (int64_t)b * a.T);
        if (!isfinite(a.stop_logit[i]) || !isfinite(a.pitch_pred[i]) || !isfinite(a.energy_pred[i])) bad += 1.f;
        if (t < a.mel_len[b]) {
            float v = bce_logits(a.stop_logit[i], a.stop_tgt[i], a.cfg.pos_weight);
            if (isfinite(v)) { s[2] += v; n[2] += 1.f; }
            v = huber(a.pitch_pred[i] - a.pitch_tgt[i], a.cfg.delta_pitch);
            if (isfinite(v)) { s[3] += v; n[3] += 1.f; }
            v = huber(a.energy_pred[i] - a.energy_tgt[i], a.cfg.delta_energy);
            if (isfinite(v)) { s[4] += v; n[4] += 1.f; }
        }
    }
    // the eleven block sums behind ONE barrier (they were eleven block reductions of two barriers each, the larger part of this
    // latency-bound launch): every wave reduces its eleven values (DPP), the first six threads add the four waves' and issue the atomics
    __shared__ double red11[11][4];
    (void)red;
    const int wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const double ds = wave_sum_d((double)s[k]), dn = wave_sum_d((double)n[k]);
        if ((threadIdx.x & 63) == 0) { red11[k][wv] = ds; red11[5 + k][wv] = dn; }
    }
    const double dbw = wave_sum_d((double)bad);
    if ((threadIdx.x & 63) == 0) red11[10][wv] = dbw;
    __syncthreads();
    if (threadIdx.x < 5) {
        const int k = threadIdx.x;
        const double ds = (red11[k][0] + red11[k][1]) + (red11[k][2] + red11[k][3]);
        const double dn = (red11[5 + k][0] + red11[5 + k][1]) + (red11[5 + k][2] + red11[5 + k][3]);
        if (dn > 0.0) { atomicAdd(&acc[k], ds); atomicAdd(&acc[5 + k], dn); }
    } else if (threadIdx.x == 5) {
        const double db = (red11[10][0] + red11[10][1]) + (red11[10][2] + red11[10][3]);
        if (db > 0.0) atomicAdd(&acc[10], db);
    }
}

// guard (nullable): 2 doubles of the step driver's state — [0] "a micro-batch of the current accumulation cycle had
// non-finite outputs or losses" (the reference then drops the cycle's gradients and takes no optimizer step,
// trainer.py:2304-2314; kk_opt_prepare honours and clears it), [1] how many micro-batches were flagged so far.
__global__ void losses_finalize_kernel(double *__restrict__ acc, KkLossCfg cfg, const int64_t *__restrict__ max_dur,
                                       int T, float *__restrict__ losses, float *__restrict__ coef, double *__restrict__ guard, int clear) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double scale = (double)cfg.loss_scale;
    if (cfg.adaptive) {                               // trainer.py:2218-2242 (second branch overrides the first)
        const double risk = fmax((double)T / 1400.0, max_dur ? (double)(*max_dur) / 150.0 : 0.0);
        if (risk > 1.0) scale *= fmax(0.25, 1.0 / risk);
    }
    const float cap[5] = {100.f, 100.f, 100.f, 10.f, 10.f};
    const float w[5] = {1.f, cfg.w_dur, cfg.w_stop, cfg.w_pitch, cfg.w_energy};
    float total = 0.f;
    for (int k = 0; k < 5; ++k) {
        const double cnt = acc[5 + k];
        float mean = cnt > 0.0 ? (float)(acc[k] / cnt) : 0.f;
        // torch.clamp(max=cap): value = min(mean, cap); gradient flows only while mean <= cap
        const bool open = cnt > 0.0 && mean <= cap[k];
        if (mean > cap[k]) mean = cap[k];
        losses[1 + k] = mean;
        total += mean * w[k];
        coef[k] = open ? (float)((double)w[k] * scale / cnt) : 0.f;
    }
    losses[0] = total;
    // finite-output guard (trainer.py:3233-3256: every element of the five predictions) and finite-loss guard (:3274-3296)
    bool ok = acc[10] == 0.0 && isfinite(total);
    for (int k = 0; k < 5; ++k) ok = ok && isfinite(losses[1 + k]);
    if (!ok && guard) {          // (without a guard slot the call is calculate_training_losses alone: elements masked, no veto)
        for (int k = 0; k < 5; ++k) coef[k] = 0.f;
        guard[0] = 1.0;
        guard[1] += 1.0;
    }
    if (clear)                   // the accumulator leaves the step zero: the next kk_losses_fwd needs no zero-fill launch in front of it
        for (int k = 0; k < 12; ++k) acc[k] = 0.0;
}

__global__ __launch_bounds__(256) void losses_bwd_kernel(LossArgs a, const float *__restrict__ coef, float *__restrict__ dmel,
                                                         float *__restrict__ ddur, float *__restrict__ dstop,
                                                         float *__restrict__ dpitch, float *__restrict__ denergy) {
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, stride = (int64_t)gridDim.x * 256;
    const int64_t nmel = (int64_t)a.B * a.T * a.M, nfr = (int64_t)a.B * a.T, nph = (int64_t)a.B * a.P;
    const float c0 = coef[0], c1 = coef[1], c2 = coef[2], c3 = coef[3], c4 = coef[4];
    for (int64_t i = tid; i < nmel; i += stride) {
        const int64_t fr = i / a.M;
        const int b = (int)(fr / a.T), t = (int)(fr - (int64_t)b * a.T);
        float g = 0.f;
        if (t < a.mel_len[b]) {
            const float e = a.mel_pred[i] - a.mel_tgt[i];
            if (isfinite(e)) g = e > 0.f ? c0 : (e < 0.f ? -c0 : 0.f);
        }
        dmel[i] = g;
    }
    for (int64_t i = tid; i < nph; i += stride) {
        const int b = (int)(i / a.P), p = (int)(i - (int64_t)b * a.P);
        const int64_t d = a.dur[i];
        float g = 0.f;
        if (p < a.ph_len[b] && d > 0) g = c1 * huber_grad(a.dur_pred[i] - logf((float)d + 1.f), a.cfg.delta_dur);
        ddur[i] = g;
    }
    for (int64_t i = tid; i < nfr; i += stride) {
        const int b = (int)(i / a.T), t = (int)(i - (int64_t)b * a.T);
        float gs = 0.f, gp = 0.f, ge = 0.f;
        if (t < a.mel_len[b]) {
            const float z = a.stop_logit[i], y = a.stop_tgt[i];
            if (isfinite(bce_logits(z, y, a.cfg.pos_weight))) {
                const float sg = 1.f / (1.f + expf(-z));
                gs = c2 * ((1.f - y) * sg - a.cfg.pos_weight * y * (1.f - sg));
            }
            float e = a.pitch_pred[i] - a.pitch_tgt[i];
            if (isfinite(e)) gp = c3 * huber_grad(e, a.cfg.delta_pitch);
            e = a.energy_pred[i] - a.energy_tgt[i];
            if (isfinite(e)) ge = c4 * huber_grad(e, a.cfg.delta_energy);
        }
        dstop[i] = gs; dpitch[i] = gp; denergy[i] = ge;
    }
}

LossArgs pack(const float *mel_pred, const float *mel_tgt, const float *dur_pred, const int64_t *dur,
              const float *stop_logit, const float *stop_tgt, const float *pitch_pred, const float *pitch_tgt,
              const float *energy_pred, const float *energy_tgt, const int64_t *mel_len, const int64_t *ph_len, int B,
              int T, int P, int M, const KkLossCfg *cfg) {
    LossArgs a;
    a.mel_pred = mel_pred; a.mel_tgt = mel_tgt; a.dur_pred = dur_pred; a.dur = dur; a.stop_logit = stop_logit;
    a.stop_tgt = stop_tgt; a.pitch_pred = pitch_pred; a.pitch_tgt = pitch_tgt; a.energy_pred = energy_pred;
    a.energy_tgt = energy_tgt; a.mel_len = mel_len; a.ph_len = ph_len; a.B = B; a.T = T; a.P = P; a.M = M; a.cfg = *cfg;
    return a;
}

}  // namespace

extern "C" int kk_losses_fwd(const float *mel_pred, const float *mel_tgt, const float *dur_pred, const int64_t *dur,
                             const float *stop_logit, const float *stop_tgt, const float *pitch_pred,
                             const float *pitch_tgt, const float *energy_pred, const float *energy_tgt,
                             const int64_t *mel_len, const int64_t *ph_len, int B, int T, int P, int M,
                             const KkLossCfg *cfg, const int64_t *max_dur, double *acc, float *losses, float *coef,
                             double *guard, int flags, void *stream) {
    KK_REQUIRE(B > 0 && T > 0 && P > 0 && M > 0 && cfg, "kk_losses_fwd: bad args");
    hipStream_t s = (hipStream_t)stream;
    if (!(flags & 1)) {                                         // bit 0: acc[0..12) == 0 on entry (a finalize with `clear` left it so)
        const int e = kk_zero_async(acc, 12 * sizeof(double), s);
        if (e != 0) return e;
    }
    LossArgs a = pack(mel_pred, mel_tgt, dur_pred, dur, stop_logit, stop_tgt, pitch_pred, pitch_tgt, energy_pred,
                      energy_tgt, mel_len, ph_len, B, T, P, M, cfg);
    int blocks = kk_cdiv((int64_t)B * T * M, 256 * 8);
    blocks = blocks > 1024 ? 1024 : (blocks < 1 ? 1 : blocks);
    hipLaunchKernelGGL(losses_fwd_kernel, dim3(blocks), dim3(256), 0, s, a, acc);
    hipLaunchKernelGGL(losses_finalize_kernel, dim3(1), dim3(64), 0, s, acc, *cfg, max_dur, T, losses, coef, guard, (flags >> 1) & 1);
    KK_LAUNCH_CHECK("kk_losses_fwd");
    return 0;
}

// Finish the scalars again from `acc` — data parallel: after the 5 sums + 5 counts have been SUM-all-reduced (and
// max_dur MAX-all-reduced), every rank normalises by the GLOBAL valid-element counts, so the summed gradients are the
// global-batch gradients also when the shards are ragged.  T = the global-batch mel length.
extern "C" int kk_losses_finalize(double *acc, const KkLossCfg *cfg, const int64_t *max_dur, int T, float *losses,
                                  float *coef, double *guard, int clear, void *stream) {
    KK_REQUIRE(acc && cfg && losses && coef && T > 0, "kk_losses_finalize: bad args");
    hipLaunchKernelGGL(losses_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, acc, *cfg, max_dur, T, losses, coef, guard, clear);
    KK_LAUNCH_CHECK("kk_losses_finalize");
    return 0;
}

extern "C" int kk_losses_bwd(const float *mel_pred, const float *mel_tgt, const float *dur_pred, const int64_t *dur,
                             const float *stop_logit, const float *stop_tgt, const float *pitch_pred,
                             const float *pitch_tgt, const float *energy_pred, const float *energy_tgt,
                             const int64_t *mel_len, const int64_t *ph_len, int B, int T, int P, int M,
                             const KkLossCfg *cfg, const float *coef, float *dmel, float *ddur, float *dstop,
                             float *dpitch, float *denergy, void *stream) {
    KK_REQUIRE(B > 0 && T > 0 && P > 0 && M > 0 && cfg, "kk_losses_bwd: bad args");
    LossArgs a = pack(mel_pred, mel_tgt, dur_pred, dur, stop_logit, stop_tgt, pitch_pred, pitch_tgt, energy_pred,
                      energy_tgt, mel_len, ph_len, B, T, P, M, cfg);
    int blocks = kk_cdiv((int64_t)B * T * M, 256 * 4);
    blocks = blocks > 2048 ? 2048 : (blocks < 1 ? 1 : blocks);
    hipLaunchKernelGGL(losses_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, coef, dmel, ddur, dstop,
                       dpitch, denergy);
    KK_LAUNCH_CHECK("kk_losses_bwd");
    return 0;
}
