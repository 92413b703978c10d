// Gather / elementwise kernels of the Kokoro train step (HBM-bound integer + fp32 work).
//
//  GLU gate                      model/transformers.py:107-108 (exact-erf nn.GELU, :51)
//  text+stress embedding, PE     model/model.py:375-378; model/positional_encoding.py:66-74
//  length regulator              utils/lengths.py:16-96 (index expansion is integer-exact)
//  variance-adaptor pieces       model/variance_predictor.py:89-115 (conv k=3 as im2col+GEMM, Linear(C->1)+mask),
//                                :363-368 (frame mask), :181-218,430-437 (bucketize + embedding adds + masked_fill)
//  decoder-input shift           model/model.py:519
#include "kk_common.h"

namespace {

// fp32 storage = the parity mode: exact erf.  bf16 storage: the fast pair of kk_common.h (the same function the GEMM
// epilogues use, so the fused and the unfused GLU give the same bits).
template <typename T> __device__ __forceinline__ void gelu_pair(float x, float &g, float &dg) {
    if constexpr (sizeof(T) == 2) kk_gelu_pair_fast(x, g, dg);
    else { g = kk_gelu(x); dg = kk_gelu_grad(x); }
}
template <typename T> __device__ __forceinline__ float gelu_f(float x) {
    if constexpr (sizeof(T) == 2) return kk_gelu_fast(x);
    else return kk_gelu(x);
}

// ------------------------------------------------------------------ GLU
// Dropout on the gated product (transformers.py:108) is fused: mask = f(seed, site, row*F + col).
struct Drop1 { const uint32_t *seed; uint32_t site; float p; };
__device__ __forceinline__ float4 drop4(const Drop1 &d, uint32_t seed, uint32_t thr, float ik, uint64_t idx0) {
    if (thr == 0u) return make_float4(1.f, 1.f, 1.f, 1.f);
    float m[4];
    kk_drop_mul4(seed, d.site, idx0, thr, ik, m);        // idx0 is a multiple of 4 at every call site
    return make_float4(m[0], m[1], m[2], m[3]);
}

template <typename T>
__global__ __launch_bounds__(256) void glu_fwd_kernel(const T *__restrict__ h, T *__restrict__ g, int64_t total4, int F, Drop1 d) {
    const int F4 = F / 4;
    const uint32_t thr = d.seed ? kk_drop_threshold(d.p) : 0u, seed = thr ? *d.seed : 0u;
    const float ik = thr ? 1.f / (1.f - d.p) : 1.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / F4;
        const int c = (int)(i - row * F4) * 4;
        const float4 a = ldv4<T>(h + row * 2 * F + c), b = ldv4<T>(h + row * 2 * F + F + c);
        const float4 m = drop4(d, seed, thr, ik, (uint64_t)row * F + c);
        stv4<T>(g + row * F + c, make_float4(gelu_f<T>(a.x) * b.x * m.x, gelu_f<T>(a.y) * b.y * m.y, gelu_f<T>(a.z) * b.z * m.z, gelu_f<T>(a.w) * b.w * m.w));
    }
}

template <typename T>
__global__ __launch_bounds__(256) void glu_bwd_kernel(const T *__restrict__ dg, const T *__restrict__ h,
                                                      T *__restrict__ dh, int64_t total4, int F, Drop1 dr) {
    const int F4 = F / 4;
    const uint32_t thr = dr.seed ? kk_drop_threshold(dr.p) : 0u, seed = thr ? *dr.seed : 0u;
    const float ik = thr ? 1.f / (1.f - dr.p) : 1.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / F4;
        const int c = (int)(i - row * F4) * 4;
        const float4 a = ldv4<T>(h + row * 2 * F + c), b = ldv4<T>(h + row * 2 * F + F + c);
        float4 d = ldv4<T>(dg + row * F + c);
        const float4 m = drop4(dr, seed, thr, ik, (uint64_t)row * F + c);
        d.x *= m.x; d.y *= m.y; d.z *= m.z; d.w *= m.w;
        float gx, gy, gz, gw, dx, dy, dz, dw;
        gelu_pair<T>(a.x, gx, dx); gelu_pair<T>(a.y, gy, dy); gelu_pair<T>(a.z, gz, dz); gelu_pair<T>(a.w, gw, dw);
        stv4<T>(dh + row * 2 * F + c, make_float4(d.x * b.x * dx, d.y * b.y * dy, d.z * b.z * dz, d.w * b.w * dw));
        stv4<T>(dh + row * 2 * F + F + c, make_float4(d.x * gx, d.y * gy, d.z * gz, d.w * gw));
    }
}

// ------------------------------------------------------------------ embedding + PE
__global__ __launch_bounds__(256) void embed_fwd_kernel(const int64_t *__restrict__ ids, const int64_t *__restrict__ stress,
                                                        const float *__restrict__ emb, const float *__restrict__ semb,
                                                        const float *__restrict__ pe, float *__restrict__ out, int64_t total4,
                                                        int P, int H, float scale, Drop1 d) {
    const int H4 = H / 4;
    const uint32_t thr = d.seed ? kk_drop_threshold(d.p) : 0u, seed = thr ? *d.seed : 0u;
    const float ik = thr ? 1.f / (1.f - d.p) : 1.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
        const int64_t tok = i / H4;
        const int c = (int)(i - tok * H4) * 4;
        const int p = (int)(tok % P);
        const float4 e = ld4(emb + ids[tok] * H + c), pv = ld4(pe + (int64_t)p * H + c);
        float4 o = make_float4(e.x * scale, e.y * scale, e.z * scale, e.w * scale);
        if (stress) { const float4 s = ld4(semb + stress[tok] * H + c); o.x += s.x; o.y += s.y; o.z += s.z; o.w += s.w; }
        o.x += pv.x; o.y += pv.y; o.z += pv.z; o.w += pv.w;
        const float4 m = drop4(d, seed, thr, ik, (uint64_t)tok * H + c);     // PE dropout (positional_encoding.py:74)
        o.x *= m.x; o.y *= m.y; o.z *= m.z; o.w *= m.w;
        st4(out + tok * H + c, o);
    }
}

// The encoder's whole prologue as ONE launch, a wave per token: the embedding row above (same expressions, same bits), the key-padding
// mask (ids == 0, model.py:372) and the first layer's pre-LayerNorm (the arithmetic of layernorm_fwd_kernel, kk_norm.hip).  These were
// three dependent launches in front of the encoder forward — the head of the step's critical path, ~7 us each with their gaps.
template <typename TY, int NV>
__global__ __launch_bounds__(256) void embed_ln_fwd_kernel(const int64_t *__restrict__ ids, const int64_t *__restrict__ stress,
                                                           const float *__restrict__ emb, const float *__restrict__ semb,
                                                           const float *__restrict__ pe, float *__restrict__ out, int64_t ntok, int P, int H,
                                                           float scale, Drop1 d, uint8_t *__restrict__ key_mask,
                                                           const float *__restrict__ gamma, const float *__restrict__ beta, TY *__restrict__ y,
                                                           float *__restrict__ mean_o, float *__restrict__ rstd_o) {
    const int lane = threadIdx.x & 63;
    const int64_t tok = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tok >= ntok) return;
    const uint32_t thr = d.seed ? kk_drop_threshold(d.p) : 0u, seed = thr ? *d.seed : 0u;
    const float ik = thr ? 1.f / (1.f - d.p) : 1.f;
    const int64_t id = ids[tok], sid = stress ? stress[tok] : 0;
    const int p = (int)(tok % P);
    if (key_mask && lane == 0) key_mask[tok] = id == 0 ? 1 : 0;
    float4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane * 4 + 256 * i;
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < H) {
            const float4 e = ld4(emb + id * H + c), pv = ld4(pe + (int64_t)p * H + c);
            float4 o = make_float4(e.x * scale, e.y * scale, e.z * scale, e.w * scale);
            if (stress) { const float4 sv = ld4(semb + sid * H + c); o.x += sv.x; o.y += sv.y; o.z += sv.z; o.w += sv.w; }
            o.x += pv.x; o.y += pv.y; o.z += pv.z; o.w += pv.w;
            const float4 m = drop4(d, seed, thr, ik, (uint64_t)tok * H + c);
            o.x *= m.x; o.y *= m.y; o.z *= m.z; o.w *= m.w;
            st4(out + tok * H + c, o);
            v[i] = o;
        }
        s += v[i].x + v[i].y + v[i].z + v[i].w;
    }
    const float mean = wave_sum(s) / (float)H;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane * 4 + 256 * i;
        if (c < H) {
            const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, dd = v[i].w - mean;
            q += a * a + b * b + cc * cc + dd * dd;
        }
    }
    const float rstd = 1.f / sqrtf(wave_sum(q) / (float)H + 1e-5f);
    TY *yr = y + tok * H;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane * 4 + 256 * i;
        if (c < H) {
            const float4 g = ld4(gamma + c), b = ld4(beta + c);
            float4 o;
            o.x = (v[i].x - mean) * rstd * g.x + b.x;
            o.y = (v[i].y - mean) * rstd * g.y + b.y;
            o.z = (v[i].z - mean) * rstd * g.z + b.z;
            o.w = (v[i].w - mean) * rstd * g.w + b.w;
            stv4<TY>(yr + c, o);
        }
    }
    if (lane == 0) {
        mean_o[tok] = mean;
        rstd_o[tok] = rstd;
    }
}

__global__ __launch_bounds__(256) void embed_bwd_kernel(const int64_t *__restrict__ ids, const int64_t *__restrict__ stress,
                                                        const float *__restrict__ dout, float *__restrict__ demb,
                                                        float *__restrict__ dsemb, int64_t total, int H, float scale, Drop1 dr) {
    const uint32_t thr = dr.seed ? kk_drop_threshold(dr.p) : 0u, seed = thr ? *dr.seed : 0u;
    const float ik = thr ? 1.f / (1.f - dr.p) : 1.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t tok = i / H;
        const int c = (int)(i - tok * H);
        const float d = dout[i] * kk_drop_mul(seed, dr.site, (uint64_t)i, thr, ik);
        atomicAdd(&demb[ids[tok] * H + c], d * scale);
        if (stress) { const int64_t s = stress[tok]; if (s != 0) atomicAdd(&dsemb[s * H + c], d); }   // padding_idx=0 (model.py:93)
    }
}

// ------------------------------------------------------------------ length regulator
// One workgroup per batch row: inclusive scan of max(dur,0) into LDS, then a per-frame upper_bound.
constexpr int LR_MAXP = 4096;
__global__ __launch_bounds__(256) void lr_index_kernel(const int64_t *__restrict__ dur, int64_t *__restrict__ idx,
                                                       int64_t *__restrict__ lens, int64_t *__restrict__ total, int P, int L) {
    __shared__ int64_t cum[LR_MAXP];
    __shared__ int64_t wtot[4];
    __shared__ int64_t carry;
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
    if (t == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < P; base += 256) {
        const int j = base + t;
        int64_t v = 0;
        if (j < P) { v = dur[(int64_t)b * P + j]; v = v > 0 ? v : 0; }
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {            // wave inclusive scan
            const int64_t n = __shfl_up(v, o, 64);
            if (lane >= o) v += n;
        }
        if (lane == 63) wtot[w] = v;
        __syncthreads();
        int64_t off = carry;
        for (int k = 0; k < w; ++k) off += wtot[k];
        if (j < P) cum[j] = v + off;
        __syncthreads();
        if (t == 0) carry += wtot[0] + wtot[1] + wtot[2] + wtot[3];
        __syncthreads();
    }
    const int64_t tot = carry;
    const int64_t len = tot < L ? tot : L;
    if (t == 0) { lens[b] = len; total[b] = tot; }
    for (int f = t; f < L; f += 256) {
        int64_t r = -1;
        if (f < len) {
            int lo = 0, hi = P;                       // first j with cum[j] > f  ==  #{j : cum[j] <= f}
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (cum[mid] <= f) lo = mid + 1; else hi = mid; }
            r = lo;
        }
        idx[(int64_t)b * L + f] = r;
    }
}

__global__ __launch_bounds__(256) void lr_gather_kernel(const float *__restrict__ x, const int64_t *__restrict__ idx,
                                                        float *__restrict__ out, int64_t total4, int P, int L, int H) {
    const int H4 = H / 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
        const int64_t fr = i / H4;
        const int c = (int)(i - fr * H4) * 4;
        const int64_t b = fr / L, j = idx[fr];
        st4(out + fr * H + c, j >= 0 ? ld4(x + (b * P + j) * H + c) : make_float4(0.f, 0.f, 0.f, 0.f));
    }
}

__global__ __launch_bounds__(256) void max_i64_kernel(const int64_t *__restrict__ x, int64_t n, int64_t *__restrict__ out) {
    __shared__ int64_t red[256];
    int64_t m = 0;
    for (int64_t i = threadIdx.x; i < n; i += 256) m = x[i] > m ? x[i] : m;
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] = red[threadIdx.x + s] > red[threadIdx.x] ? red[threadIdx.x + s] : red[threadIdx.x];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = red[0];
}

// ------------------------------------------------------------------ conv k=3 as im2col (per 512-frame chunk)
template <typename TO>
__global__ __launch_bounds__(256) void im2col3_fwd_kernel(const float *__restrict__ x, TO *__restrict__ col, int64_t total,
                                                          int L, int C, int chunk) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t fr = i / C;
        const int c = (int)(i - fr * C);
        const int l = (int)(fr % L);
        const int cb = (l / chunk) * chunk, ce = cb + chunk < L ? cb + chunk : L;
        TO *o = col + fr * 3 * C + c * 3;
        o[0] = (TO)((l - 1 >= cb) ? x[i - C] : 0.f);
        o[1] = (TO)x[i];
        o[2] = (TO)((l + 1 < ce) ? x[i + C] : 0.f);
    }
}

template <typename TI>
__global__ __launch_bounds__(256) void im2col3_bwd_kernel(const TI *__restrict__ dcol, float *__restrict__ dx, int64_t total,
                                                          int L, int C, int chunk) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t fr = i / C;
        const int c = (int)(i - fr * C);
        const int l = (int)(fr % L);
        const int cb = (l / chunk) * chunk, ce = cb + chunk < L ? cb + chunk : L;
        const TI *d = dcol + fr * 3 * C + c * 3;
        float v = (float)d[1];
        if (l + 1 < ce) v += (float)d[3 * C + 0];     // row l+1, tap k=0 read x[l]
        if (l - 1 >= cb) v += (float)d[-3 * C + 2];   // row l-1, tap k=2 read x[l]
        dx[i] = v;
    }
}

// ------------------------------------------------------------------ Linear(C->1) + masked_fill  (wave per row)
__device__ __forceinline__ bool row_dead(const uint8_t *mask, int64_t r, int L, int chunk) {
    if (mask && mask[r]) return true;
    if (chunk > 0) {
        const int l = (int)(r % L), cb = (l / chunk) * chunk;
        if ((L - cb < chunk ? L - cb : chunk) < 2) return true;
    }
    return false;
}

template <typename TX>
__global__ __launch_bounds__(256) void rowdot_fwd_kernel(const TX *__restrict__ x, const float *__restrict__ w,
                                                         const float *__restrict__ b, const uint8_t *__restrict__ mask,
                                                         float *__restrict__ out, int64_t rows, int C, int L, int chunk) {
    const int lane = threadIdx.x & 63;
    for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (int64_t)gridDim.x * 4) {
        float s = 0.f;
        for (int c = lane * 4; c < C; c += 256) {
            const float4 xv = ldv4<TX>(x + r * C + c), wv = ld4(w + c);
            s += xv.x * wv.x + xv.y * wv.y + xv.z * wv.z + xv.w * wv.w;
        }
        s = wave_sum(s);
        if (lane == 0) out[r] = row_dead(mask, r, L, chunk) ? 0.f : s + b[0];
    }
}

template <typename TX>
__global__ __launch_bounds__(256) void rowdot_bwd_kernel(const float *__restrict__ dout, const TX *__restrict__ x,
                                                         const float *__restrict__ w, const uint8_t *__restrict__ mask,
                                                         float *__restrict__ dx, float *__restrict__ dw, float *__restrict__ db,
                                                         int64_t rows, int C, int L, int chunk, int rows_per_block) {
    // thread t owns columns t, t+256, ... (C <= 1024); block walks a slab of rows.
    const int64_t rbeg = (int64_t)blockIdx.x * rows_per_block;
    const int64_t rend = rbeg + rows_per_block < rows ? rbeg + rows_per_block : rows;
    float aw[4] = {0.f, 0.f, 0.f, 0.f};
    float ab = 0.f;
    for (int64_t r = rbeg; r < rend; ++r) {
        const float d = row_dead(mask, r, L, chunk) ? 0.f : dout[r];
        if (threadIdx.x == 0) ab += d;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = threadIdx.x + 256 * k;
            if (c < C) {
                aw[k] += d * (float)x[r * C + c];
                if (dx) dx[r * C + c] = d * w[c];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = threadIdx.x + 256 * k;
        if (c < C) atomicAdd(&dw[c], aw[k]);
    }
    if (threadIdx.x == 0) atomicAdd(db, ab);
}

// The same backward, a wave per row and 16-byte accesses: the form above walks its rows one after the other with one 4-byte load in
// flight per thread behind the row's mask / dout round trip (22 us for 4096 x 256 on the side branch).  Here a wave takes RB rows per
// trip — their mask bytes, dout scalars and x rows are all in flight together — a lane owns 4 NV columns, and the four waves' weight
// gradient sums meet in LDS before the workgroup's C atomics.  Same grid (one workgroup per 16 rows): thinner, not burstier.
template <typename TX, int NV>
__global__ __launch_bounds__(256) void rowdot_bwd_rows_kernel(const float *__restrict__ dout, const TX *__restrict__ x,
                                                              const float *__restrict__ w, const uint8_t *__restrict__ mask,
                                                              float *__restrict__ dx, float *__restrict__ dw, float *__restrict__ db,
                                                              int64_t rows, int C, int L, int chunk, int rows_per_block,
                                                              float *__restrict__ partials) {
    constexpr int RB = 4;
    __shared__ float red[3][NV * 256 + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t rbeg = (int64_t)blockIdx.x * rows_per_block;
    const int64_t rend = rbeg + rows_per_block < rows ? rbeg + rows_per_block : rows;
    float4 wv[NV], aw[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane * 4 + 256 * i;
        wv[i] = c < C ? ld4(w + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        aw[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float ab = 0.f;
    for (int64_t r0 = rbeg + wave; r0 < rend; r0 += 4 * RB) {
        float d[RB];
        float4 xv[RB][NV];
#pragma unroll
        for (int u = 0; u < RB; ++u) {                            // every load of the trip first
            const int64_t r = r0 + 4 * u;
            const bool in = r < rend;
            const int64_t rr = in ? r : rbeg;
            bool dead = !in || (mask && mask[rr]);
            if (chunk > 0) {
                const int l = (int)(rr % L), cb = (l / chunk) * chunk;
                dead = dead || (L - cb < chunk ? L - cb : chunk) < 2;
            }
            d[u] = dout[rr];
            if (dead) d[u] = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = lane * 4 + 256 * i;
                xv[u][i] = c < C ? ldv4<TX>(x + rr * C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int u = 0; u < RB; ++u) {
            const int64_t r = r0 + 4 * u;
            if (r >= rend) continue;                              // (wave-uniform)
            ab += d[u];
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = lane * 4 + 256 * i;
                if (c < C) {
                    aw[i].x += d[u] * xv[u][i].x; aw[i].y += d[u] * xv[u][i].y; aw[i].z += d[u] * xv[u][i].z; aw[i].w += d[u] * xv[u][i].w;
                    if (dx) st4(dx + r * C + c, make_float4(d[u] * wv[i].x, d[u] * wv[i].y, d[u] * wv[i].z, d[u] * wv[i].w));
                }
            }
        }
    }
    // waves 1..3 hand their sums to wave 0
    if (wave > 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            float *p = &red[wave - 1][lane * 4 + 256 * i];
            p[0] = aw[i].x; p[1] = aw[i].y; p[2] = aw[i].z; p[3] = aw[i].w;
        }
        if (lane == 0) red[wave - 1][NV * 256] = ab;
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane * 4 + 256 * i;
            if (c < C) {
                float t[4] = {aw[i].x, aw[i].y, aw[i].z, aw[i].w};
#pragma unroll
                for (int k = 0; k < 3; ++k)
#pragma unroll
                    for (int e = 0; e < 4; ++e) t[e] += red[k][c + e];
                if (partials) {                                   // one plain row per workgroup for kk_partials_reduce: [dw (C) | db | pad]
                    st4(partials + (int64_t)blockIdx.x * (C + 4) + c, make_float4(t[0], t[1], t[2], t[3]));
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) atomicAdd(&dw[c + e], t[e]);
                }
            }
        }
        if (lane == 0) {
            const float tb = ab + red[0][NV * 256] + red[1][NV * 256] + red[2][NV * 256];
            if (partials) partials[(int64_t)blockIdx.x * (C + 4) + C] = tb;
            else atomicAdd(db, tb);
        }
    }
}

// ------------------------------------------------------------------ bucketize + embedding adds + frame mask (wave per frame)
__device__ __forceinline__ int bucketize_left(const float *__restrict__ bins, int n, float v) {
    int lo = 0, hi = n;                                  // #{i : bins[i] < v}   (torch.bucketize right=False)
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (bins[mid] < v) lo = mid + 1; else hi = mid; }
    return lo;
}

template <typename TO>
__global__ __launch_bounds__(256) void bucket_embed_add_fwd_kernel(
    const float *__restrict__ x, const float *__restrict__ pitch, const float *__restrict__ energy,
    const float *__restrict__ pbins, const float *__restrict__ ebins, const float *__restrict__ pemb,
    const float *__restrict__ eemb, const int64_t *__restrict__ lens, TO *__restrict__ out, int32_t *__restrict__ pidx,
    int32_t *__restrict__ eidx, uint8_t *__restrict__ fmask, int64_t rows, int T, int H, int nbins) {
    const int lane = threadIdx.x & 63;
    for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (int64_t)gridDim.x * 4) {
        const int b = (int)(r / T), f = (int)(r - (int64_t)b * T);
        const bool masked = f >= lens[b];
        const int pi = bucketize_left(pbins, nbins - 1, pitch[r]);
        const int ei = bucketize_left(ebins, nbins - 1, energy[r]);
        if (lane == 0) { pidx[r] = pi; eidx[r] = ei; fmask[r] = masked ? 1 : 0; }
        for (int c = lane * 4; c < H; c += 256) {
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!masked) {
                const float4 a = ld4(x + r * H + c), p = ld4(pemb + (int64_t)pi * H + c), e = ld4(eemb + (int64_t)ei * H + c);
                o = make_float4(a.x + p.x + e.x, a.y + p.y + e.y, a.z + p.z + e.z, a.w + p.w + e.w);
            }
            stv4<TO>(out + r * H + c, o);
        }
    }
}

// Length-regulator gather + the bucket-embedding adds + SpecAugment as ONE launch, a wave per frame: the three launches between the
// encoder's last LayerNorm and the cross-attention K/V GEMM (kk_length_regulate_gather, kk_bucket_embed_add_fwd, kk_specaug) sit on
// the step's critical chain and were a round trip of the [frames, H] tensor each.  Same expressions, same bits; the SpecAugment decisions
// are the same functions of (seed, site, sample, mask number) as specaug_kernel's (kk_dropout.hip): its backward is unchanged.
template <typename TO>
__global__ __launch_bounds__(256) void regulate_embed_fwd_kernel(
    const float *__restrict__ enc, const int64_t *__restrict__ idx, const float *__restrict__ pitch, const float *__restrict__ energy,
    const float *__restrict__ pbins, const float *__restrict__ ebins, const float *__restrict__ pemb, const float *__restrict__ eemb,
    const int64_t *__restrict__ lens, float *__restrict__ xf, TO *__restrict__ out, int32_t *__restrict__ pidx, int32_t *__restrict__ eidx,
    uint8_t *__restrict__ fmask, int64_t rows, int P, int T, int H, int nbins, const uint32_t *__restrict__ seedp, uint32_t site, int tmax,
    int fmax, int nt, int nf, int wt) {
    const int lane = threadIdx.x & 63;
    const uint32_t seed = seedp ? *seedp : 0u;
    const int time_limit = max(1, min(tmax, T / 4));
    for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (int64_t)gridDim.x * 4) {
        const int b = (int)(r / T), f = (int)(r - (int64_t)b * T);
        const int64_t j = idx[r];
        const bool masked = f >= lens[b];
        const int pi = bucketize_left(pbins, nbins - 1, pitch[r]);
        const int ei = bucketize_left(ebins, nbins - 1, energy[r]);
        if (lane == 0) { pidx[r] = pi; eidx[r] = ei; fmask[r] = masked ? 1 : 0; }
        bool tm = false;
        if (seedp) {
            for (int k = 0; k < nt; ++k) {
                const int len = (int)(kk_hash(seed, site, (uint64_t)b * 64 + 2 * k) % (uint32_t)time_limit);
                const int t0 = (int)(kk_hash(seed, site, (uint64_t)b * 64 + 2 * k + 1) % (uint32_t)max(1, T - len));
                tm |= (f >= t0 && f < t0 + len);
            }
        }
        for (int c = lane * 4; c < H; c += 256) {
            const float4 a = j >= 0 ? ld4(enc + ((int64_t)b * P + j) * H + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            st4_out(xf + r * H + c, a, wt);
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!masked) {
                const float4 p = ld4(pemb + (int64_t)pi * H + c), e = ld4(eemb + (int64_t)ei * H + c);
                o = make_float4(a.x + p.x + e.x, a.y + p.y + e.y, a.z + p.z + e.z, a.w + p.w + e.w);
            }
            if (seedp) {
                bool fm[4] = {tm, tm, tm, tm};
                for (int k = 0; k < nf; ++k) {
                    const int len = (int)(kk_hash(seed, site, (uint64_t)b * 64 + 32 + 2 * k) % (uint32_t)max(1, fmax));
                    const int f0 = (int)(kk_hash(seed, site, (uint64_t)b * 64 + 32 + 2 * k + 1) % (uint32_t)max(1, H - len));
#pragma unroll
                    for (int e2 = 0; e2 < 4; ++e2) fm[e2] |= (c + e2 >= f0 && c + e2 < f0 + len);
                }
                if (fm[0]) o.x = 0.f;
                if (fm[1]) o.y = 0.f;
                if (fm[2]) o.z = 0.f;
                if (fm[3]) o.w = 0.f;
            }
            stv4_out<TO>(out + r * H + c, o, wt);      // (write-through: 12 MB that the cross-attention K/V GEMM reads next)
        }
    }
}

__global__ __launch_bounds__(256) void bucket_embed_add_bwd_kernel(const float *__restrict__ dout, const int32_t *__restrict__ pidx,
                                                                   const int32_t *__restrict__ eidx, const uint8_t *__restrict__ fmask,
                                                                   float *__restrict__ dpemb, float *__restrict__ deemb, int64_t rows, int H) {
    const int lane = threadIdx.x & 63;
    for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (int64_t)gridDim.x * 4) {
        if (fmask[r]) continue;
        const int64_t po = (int64_t)pidx[r] * H, eo = (int64_t)eidx[r] * H;
        for (int c = lane; c < H; c += 64) {
            const float d = dout[r * H + c];
            atomicAdd(&dpemb[po + c], d);
            atomicAdd(&deemb[eo + c], d);
        }
    }
}

// The same scatter-add through LDS: a workgroup owns 16 columns of H and a quarter of the rows, adds its rows into two
// private [nbins][16] tables with LDS atomics (4096 rows fall into <= nbins buckets, so the global tables see nbins*16
// atomics per workgroup instead of rows*16), then flushes the non-zero entries.  64 us -> ~10 us at 4096 x 512.
constexpr int BE_CW = 16, BE_RG = 8;
__global__ __launch_bounds__(256) void bucket_embed_add_bwd_lds_kernel(const float *__restrict__ dout, const int32_t *__restrict__ pidx,
                                                                       const int32_t *__restrict__ eidx, const uint8_t *__restrict__ fmask,
                                                                       float *__restrict__ dpemb, float *__restrict__ deemb, int64_t rows,
                                                                       int H, int nbins) {
    extern __shared__ float be_tab[];                     // [2][nbins][BE_CW]
    float *tp = be_tab, *te = be_tab + nbins * BE_CW;
    for (int i = threadIdx.x; i < 2 * nbins * BE_CW; i += 256) be_tab[i] = 0.f;
    __syncthreads();
    const int c0 = blockIdx.x * BE_CW, col = threadIdx.x & (BE_CW - 1), rsub = threadIdx.x / BE_CW;
    const int64_t per = (rows + gridDim.y - 1) / gridDim.y, r0 = blockIdx.y * per, r1 = r0 + per < rows ? r0 + per : rows;
    if (c0 + col < H) {
        constexpr int RS = 256 / BE_CW, U = 8;            // U rows in flight per thread: the loop is load-latency bound
        for (int64_t rb = r0 + rsub; rb < r1; rb += RS * U) {
            float d[U];
            int pi[U], ei[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {                     // (four INDEPENDENT loads per row: behind `!fmask[r] ? ... :` they were two round trips)
                const int64_t r = rb + u * RS, rr = r < r1 ? r : r0;
                const uint8_t fm = fmask[rr];
                d[u] = dout[rr * H + c0 + col];
                pi[u] = pidx[rr];
                ei[u] = eidx[rr];
                if (r >= r1 || fm) pi[u] = -1;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (pi[u] < 0) continue;
                atomicAdd(&tp[pi[u] * BE_CW + col], d[u]);
                atomicAdd(&te[ei[u] * BE_CW + col], d[u]);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nbins * BE_CW; i += 256) {
        const int c = c0 + (i & (BE_CW - 1));
        if (c >= H) continue;
        const int64_t o = (int64_t)(i / BE_CW) * H + c;
        if (tp[i] != 0.f) atomicAdd(&dpemb[o], tp[i]);
        if (te[i] != 0.f) atomicAdd(&deemb[o], te[i]);
    }
}

// ---- the same scatter-add as a SEGMENTED sum: the frames of each table sorted by bin in the forward (kk_bucket_sort), the backward
// one workgroup per piece of at most BS_PIECE frames of ONE bin — row loads of 4 * blockDim contiguous floats, sums in registers, one global
// atomic per (piece, column).  No LDS float atomics (ds_add_f32 moves about one lane per two clocks per CU: the LDS form above spends
// half of its time there, profiles/r05_side_branch_kernels.txt).
constexpr int BS_PIECE = 16, BS_THREADS = 1024, BS_MAXBINS = 1024;
__global__ __launch_bounds__(BS_THREADS) void bucket_sort_kernel(const int32_t *__restrict__ pidx, const int32_t *__restrict__ eidx,
                                                                 const uint8_t *__restrict__ fmask, int64_t rows, int nbins, int max_items,
                                                                 int32_t *__restrict__ order, int32_t *__restrict__ items) {
    __shared__ int hist[BS_MAXBINS], cur[BS_MAXBINS], istart[BS_MAXBINS + 1];
    const int32_t *idx = blockIdx.x == 0 ? pidx : eidx;
    int32_t *ord = order + (int64_t)blockIdx.x * rows;
    int4 *it = reinterpret_cast<int4 *>(items) + (int64_t)blockIdx.x * max_items;
    for (int b = threadIdx.x; b < nbins; b += BS_THREADS) hist[b] = 0;
    __syncthreads();
    for (int64_t r = threadIdx.x; r < rows; r += BS_THREADS)
        if (!fmask[r]) atomicAdd(&hist[idx[r]], 1);
    __syncthreads();
    if (threadIdx.x == 0) {
        int s = 0, n = 0;
        for (int b = 0; b < nbins; ++b) {
            cur[b] = s;
            istart[b] = n;
            s += hist[b];
            n += (hist[b] + BS_PIECE - 1) / BS_PIECE;
        }
        istart[nbins] = n;
    }
    __syncthreads();
    for (int b = threadIdx.x; b < nbins; b += BS_THREADS) {
        const int base = cur[b], end = base + hist[b];
        int k = istart[b];
        for (int beg = base; beg < end; beg += BS_PIECE, ++k) it[k] = make_int4(b, beg, beg + BS_PIECE < end ? beg + BS_PIECE : end, 0);
    }
    for (int k = istart[nbins] + threadIdx.x; k < max_items; k += BS_THREADS) it[k] = make_int4(0, 0, 0, 0);      // empty pieces
    __syncthreads();
    for (int64_t r = threadIdx.x; r < rows; r += BS_THREADS)
        if (!fmask[r]) ord[atomicAdd(&cur[idx[r]], 1)] = (int32_t)r;
}

__global__ __launch_bounds__(256) void bucket_embed_add_bwd_sorted_kernel(const float *__restrict__ dout, const int32_t *__restrict__ order,
                                                                          const int32_t *__restrict__ items, float *__restrict__ dpemb,
                                                                          float *__restrict__ deemb, int64_t rows, int H, int max_items) {
    const int4 item = reinterpret_cast<const int4 *>(items)[(int64_t)blockIdx.y * max_items + blockIdx.x];
    const int beg = item.y, n = item.z - item.y;
    if (n <= 0) return;
    const int32_t *ord = order + (int64_t)blockIdx.y * rows + beg;
    float *dst = (blockIdx.y == 0 ? dpemb : deemb) + (int64_t)item.x * H;
    int rr[BS_PIECE];
#pragma unroll
    for (int u = 0; u < BS_PIECE; ++u) rr[u] = ord[u < n ? u : 0];             // (uniform: scalar loads)
    for (int c = threadIdx.x * 4; c < H; c += blockDim.x * 4) {
        float4 v[BS_PIECE];
#pragma unroll
        for (int u = 0; u < BS_PIECE; ++u) v[u] = ld4(dout + (int64_t)rr[u] * H + c);      // all of the piece's rows in flight
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < BS_PIECE; ++u)
            if (u < n) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
        atomicAdd(dst + c, acc.x);
        atomicAdd(dst + c + 1, acc.y);
        atomicAdd(dst + c + 2, acc.z);
        atomicAdd(dst + c + 3, acc.w);
    }
}

__global__ void ids_eq_zero_kernel(const int64_t *__restrict__ ids, uint8_t *__restrict__ mask, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        mask[i] = ids[i] == 0 ? 1 : 0;
}

__global__ __launch_bounds__(256) void shift_right_kernel(const float *__restrict__ mel, float *__restrict__ out, int64_t total,
                                                          int T, int M) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t fr = i / M;
        out[i] = (fr % T) == 0 ? 0.f : mel[i - M];
    }
}

// ---- autoregressive decode (model/generator.py:24-127, the KV-cache path of transformers.py:237-253) ------------------------------
// The launches of one decoder step read the frame index t from DEVICE memory, so that one captured hipGraph of the step is replayed
// for every frame of an utterance (the step's ~100 launches are otherwise host-bound: ~1 ms per frame through eager launches).
// prologue: the step's input frame, its positional-encoding row and RoPE rows into fixed buffers; key t of the self-attention
// caches becomes visible (the attention runs over the whole cache with a key mask, so its arguments do not change with t).
__global__ __launch_bounds__(256) void decode_prologue_kernel(const float *__restrict__ mel_all, float *__restrict__ frame_in,
                                                              const float *__restrict__ pe, float *__restrict__ pe_row,
                                                              const float *__restrict__ cos_t, const float *__restrict__ sin_t,
                                                              float *__restrict__ cos_row, float *__restrict__ sin_row,
                                                              uint8_t *__restrict__ key_mask, const int *__restrict__ t_dev, int B, int L1,
                                                              int M, int H) {
    const int t = *t_dev;
    for (int i = threadIdx.x; i < B * M; i += 256) {
        const int b = i / M, c = i - b * M;
        frame_in[i] = mel_all[((int64_t)b * L1 + t) * M + c];
    }
    for (int i = threadIdx.x; i < H; i += 256) pe_row[i] = pe[(int64_t)t * H + i];
    for (int i = threadIdx.x; i < 64; i += 256) {
        cos_row[i] = cos_t[(int64_t)t * 64 + i];
        sin_row[i] = sin_t[(int64_t)t * 64 + i];
    }
    if (threadIdx.x == 0) key_mask[t] = 0;
}
// the step's normalised q | k | v [B, 3H] -> the query buffer [B*H] and row t of the time-major K and V caches [L][B*H]
template <typename T>
__global__ __launch_bounds__(256) void decode_cache_append_kernel(const T *__restrict__ nrm, T *__restrict__ q, T *__restrict__ kc,
                                                                  T *__restrict__ vc, const int *__restrict__ t_dev, int B, int H) {
    const int t = *t_dev;
    const int64_t row = (int64_t)t * B * H;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < B * H; i += gridDim.x * 256) {
        const int b = i / H, c = i - b * H;
        const T *src = nrm + (int64_t)b * 3 * H + c;
        q[i] = src[0];
        kc[row + i] = src[H];
        vc[row + i] = src[2 * H];
    }
}
// epilogue: the step's mel frame -> row t+1 of the output (= the next step's input), its stop logits -> row t; t += 1
__global__ __launch_bounds__(256) void decode_epilogue_kernel(const float *__restrict__ frame_out, const float *__restrict__ stop,
                                                              float *__restrict__ mel_all, float *__restrict__ stop_all,
                                                              int *__restrict__ t_dev, int B, int L1, int M) {
    const int t = *t_dev;
    for (int i = threadIdx.x; i < B * M; i += 256) {
        const int b = i / M, c = i - b * M;
        mel_all[((int64_t)b * L1 + t + 1) * M + c] = frame_out[i];
    }
    for (int i = threadIdx.x; i < B; i += 256) stop_all[(int64_t)t * B + i] = stop[i];
    __syncthreads();
    if (threadIdx.x == 0) *t_dev = t + 1;
}

// Grid of a grid-stride element-wise launch: at most 1024 workgroups (4 per CU).  4096 was the cap until round 5; most of these launches
// run on the side branch or beside the persistent encoder launch, and a burst of 2048-4096 short workgroups costs the critical chain beside
// them more than the launch gains: 1024 is -0.6 % on the 8 x 512 step and -0.9 % at 8 x 1024, interleaved (profiles/r05_elem_grid_cap_ab.txt).
inline int grid_for(int64_t n, int cap = 4096) {
    static const int env_cap = kk_tune_env("KK_ELEM_GRID_CAP", 1024);   // (tools: 0 = the callers' own caps)
    if (env_cap > 0 && cap > env_cap) cap = env_cap;
    int b = kk_cdiv(n, 256);
    return b > cap ? cap : (b < 1 ? 1 : b);
}

}  // namespace

extern "C" int kk_glu_fwd(const float *h, float *g, int64_t rows, int F, const uint32_t *seed, uint32_t site, float p,
                          int io_bf16, void *stream) {
    KK_REQUIRE(rows > 0 && F > 0 && F % 4 == 0 && p >= 0.f && p < 1.f, "kk_glu_fwd: bad shape rows=%ld F=%d", (long)rows, F);
    const int64_t total4 = rows * F / 4;
    Drop1 d = {p > 0.f ? seed : nullptr, site, p};
    if (io_bf16)
        hipLaunchKernelGGL(glu_fwd_kernel<__bf16>, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream,
                           reinterpret_cast<const __bf16 *>(h), reinterpret_cast<__bf16 *>(g), total4, F, d);
    else
        hipLaunchKernelGGL(glu_fwd_kernel<float>, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, h, g, total4, F, d);
    KK_LAUNCH_CHECK("kk_glu_fwd");
    return 0;
}
extern "C" int kk_glu_bwd(const float *dg, const float *h, float *dh, int64_t rows, int F, const uint32_t *seed, uint32_t site,
                          float p, int io_bf16, void *stream) {
    KK_REQUIRE(rows > 0 && F > 0 && F % 4 == 0 && p >= 0.f && p < 1.f, "kk_glu_bwd: bad shape");
    const int64_t total4 = rows * F / 4;
    Drop1 d = {p > 0.f ? seed : nullptr, site, p};
    if (io_bf16)
        hipLaunchKernelGGL(glu_bwd_kernel<__bf16>, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream,
                           reinterpret_cast<const __bf16 *>(dg), reinterpret_cast<const __bf16 *>(h), reinterpret_cast<__bf16 *>(dh), total4, F, d);
    else
        hipLaunchKernelGGL(glu_bwd_kernel<float>, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, dg, h, dh, total4, F, d);
    KK_LAUNCH_CHECK("kk_glu_bwd");
    return 0;
}

extern "C" int kk_embed_fwd(const int64_t *ids, const int64_t *stress, const float *emb, const float *stress_emb,
                            const float *pe, float *out, int B, int P, int H, float scale, const uint32_t *seed,
                            uint32_t site, float p, void *stream) {
    KK_REQUIRE(B > 0 && P > 0 && H > 0 && H % 4 == 0 && p >= 0.f && p < 1.f, "kk_embed_fwd: bad shape");
    const int64_t total4 = (int64_t)B * P * H / 4;
    Drop1 d = {p > 0.f ? seed : nullptr, site, p};
    hipLaunchKernelGGL(embed_fwd_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, ids, stress, emb,
                       stress_emb, pe, out, total4, P, H, scale, d);
    KK_LAUNCH_CHECK("kk_embed_fwd");
    return 0;
}
extern "C" int kk_embed_ln_fwd(const int64_t *ids, const int64_t *stress, const float *emb, const float *stress_emb, const float *pe,
                               float *out, int B, int P, int H, float scale, const uint32_t *seed, uint32_t site, float p,
                               uint8_t *key_mask, const float *ln_gamma, const float *ln_beta, void *y, int y_bf16, float *mean,
                               float *rstd, void *stream) {
    KK_REQUIRE(B > 0 && P > 0 && H > 0 && H % 4 == 0 && H <= 2048 && p >= 0.f && p < 1.f, "kk_embed_ln_fwd: bad shape (H %% 4 == 0, H <= 2048)");
    KK_REQUIRE(ids && emb && pe && out && ln_gamma && ln_beta && y && mean && rstd, "kk_embed_ln_fwd: null pointer");
    const int64_t ntok = (int64_t)B * P;
    Drop1 d = {p > 0.f ? seed : nullptr, site, p};
    const dim3 grid(kk_cdiv(ntok, 4));
    const int nv = kk_cdiv(H, 256);
#define KK_EL(TY, NV) hipLaunchKernelGGL((embed_ln_fwd_kernel<TY, NV>), grid, dim3(256), 0, (hipStream_t)stream, ids, stress, emb, stress_emb, \
                                         pe, out, ntok, P, H, scale, d, key_mask, ln_gamma, ln_beta, static_cast<TY *>(y), mean, rstd)
    if (y_bf16) { if (nv <= 1) KK_EL(__bf16, 1); else if (nv <= 2) KK_EL(__bf16, 2); else if (nv <= 4) KK_EL(__bf16, 4); else KK_EL(__bf16, 8); }
    else { if (nv <= 1) KK_EL(float, 1); else if (nv <= 2) KK_EL(float, 2); else if (nv <= 4) KK_EL(float, 4); else KK_EL(float, 8); }
#undef KK_EL
    KK_LAUNCH_CHECK("kk_embed_ln_fwd");
    return 0;
}
extern "C" int kk_embed_bwd(const int64_t *ids, const int64_t *stress, const float *dout, float *demb,
                            float *dstress_emb, int B, int P, int H, float scale, const uint32_t *seed, uint32_t site,
                            float p, void *stream) {
    KK_REQUIRE(B > 0 && P > 0 && H > 0 && p >= 0.f && p < 1.f, "kk_embed_bwd: bad shape");
    const int64_t total = (int64_t)B * P * H;
    Drop1 d = {p > 0.f ? seed : nullptr, site, p};
    hipLaunchKernelGGL(embed_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, ids, stress, dout, demb,
                       dstress_emb, total, H, scale, d);
    KK_LAUNCH_CHECK("kk_embed_bwd");
    return 0;
}

extern "C" int kk_length_regulate_index(const int64_t *dur, int64_t *idx, int64_t *lens, int64_t *total, int B,
                                        int P, int L, void *stream) {
    KK_REQUIRE(B > 0 && P > 0 && L > 0, "kk_length_regulate_index: bad shape B=%d P=%d L=%d", B, P, L);
    KK_REQUIRE(P <= LR_MAXP, "kk_length_regulate_index: P=%d exceeds %d", P, LR_MAXP);
    hipLaunchKernelGGL(lr_index_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, dur, idx, lens, total, P, L);
    KK_LAUNCH_CHECK("kk_length_regulate_index");
    return 0;
}
extern "C" int kk_length_regulate_gather(const float *x, const int64_t *idx, float *out, int B, int P, int L, int H,
                                         void *stream) {
    KK_REQUIRE(B > 0 && P > 0 && L > 0 && H > 0 && H % 4 == 0, "kk_length_regulate_gather: bad shape");
    const int64_t total4 = (int64_t)B * L * H / 4;
    hipLaunchKernelGGL(lr_gather_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, x, idx, out, total4, P, L, H);
    KK_LAUNCH_CHECK("kk_length_regulate_gather");
    return 0;
}
extern "C" int kk_max_i64(const int64_t *x, int64_t n, int64_t *out, void *stream) {
    KK_REQUIRE(n > 0, "kk_max_i64: empty");
    hipLaunchKernelGGL(max_i64_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, x, n, out);
    KK_LAUNCH_CHECK("kk_max_i64");
    return 0;
}

extern "C" int kk_im2col3_fwd(const float *x, float *col, int B, int L, int C, int chunk, int col_bf16, void *stream) {
    KK_REQUIRE(B > 0 && L > 0 && C > 0 && chunk > 0, "kk_im2col3_fwd: bad shape");
    const int64_t total = (int64_t)B * L * C;
    if (col_bf16)
        hipLaunchKernelGGL(im2col3_fwd_kernel<__bf16>, dim3(grid_for(total, 8192)), dim3(256), 0, (hipStream_t)stream, x,
                           reinterpret_cast<__bf16 *>(col), total, L, C, chunk);
    else
        hipLaunchKernelGGL(im2col3_fwd_kernel<float>, dim3(grid_for(total, 8192)), dim3(256), 0, (hipStream_t)stream, x, col, total, L, C, chunk);
    KK_LAUNCH_CHECK("kk_im2col3_fwd");
    return 0;
}
extern "C" int kk_im2col3_bwd(const float *dcol, float *dx, int B, int L, int C, int chunk, int dcol_bf16, void *stream) {
    KK_REQUIRE(B > 0 && L > 0 && C > 0 && chunk > 0, "kk_im2col3_bwd: bad shape");
    const int64_t total = (int64_t)B * L * C;
    if (dcol_bf16)
        hipLaunchKernelGGL(im2col3_bwd_kernel<__bf16>, dim3(grid_for(total, 8192)), dim3(256), 0, (hipStream_t)stream,
                           reinterpret_cast<const __bf16 *>(dcol), dx, total, L, C, chunk);
    else
        hipLaunchKernelGGL(im2col3_bwd_kernel<float>, dim3(grid_for(total, 8192)), dim3(256), 0, (hipStream_t)stream, dcol, dx, total, L, C, chunk);
    KK_LAUNCH_CHECK("kk_im2col3_bwd");
    return 0;
}

extern "C" int kk_rowdot_fwd(const float *x, const float *w, const float *b, const uint8_t *mask, float *out,
                             int64_t rows, int C, int L, int chunk, int x_bf16, void *stream) {
    KK_REQUIRE(rows > 0 && C > 0 && C % 4 == 0 && L > 0, "kk_rowdot_fwd: bad shape");
    int blocks = kk_cdiv(rows, 4);
    if (blocks > 4096) blocks = 4096;
    if (x_bf16)
        hipLaunchKernelGGL(rowdot_fwd_kernel<__bf16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                           reinterpret_cast<const __bf16 *>(x), w, b, mask, out, rows, C, L, chunk);
    else
        hipLaunchKernelGGL(rowdot_fwd_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, w, b, mask, out, rows, C, L, chunk);
    KK_LAUNCH_CHECK("kk_rowdot_fwd");
    return 0;
}
// workgroups (= rows of the partial matrix [blocks][C + 4]: dw | db | pad) of a kk_rowdot_bwd launch over `rows` rows
extern "C" int kk_rowdot_bwd_blocks(int64_t rows) {
    int blocks = kk_cdiv(rows, 16);
    if (blocks > 1024) blocks = 1024;
    return kk_cdiv(rows, kk_cdiv(rows, blocks < 1 ? 1 : blocks));
}
extern "C" int kk_rowdot_bwd(const float *dout, const float *x, const float *w, const uint8_t *mask, float *dx,
                             float *dw, float *db, int64_t rows, int C, int L, int chunk, int x_bf16, float *partials, void *stream) {
    KK_REQUIRE(rows > 0 && C > 0 && C <= 1024 && L > 0, "kk_rowdot_bwd: bad shape (C <= 1024)");
    const int blocks = kk_rowdot_bwd_blocks(rows);
    const int rpb = kk_cdiv(rows, blocks);
    static const int vec = kk_tune_env("KK_ROWDOT_VEC", 1);
    const bool rows_form = vec && C % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (!dx || (reinterpret_cast<uintptr_t>(dx) & 15) == 0) &&
                           (reinterpret_cast<uintptr_t>(w) & 15) == 0;
    // (only the wave-per-row kernel writes partial rows: a call that asks for them and cannot take it must fail, not drop dw / db)
    KK_REQUIRE(!partials || (rows_form && (reinterpret_cast<uintptr_t>(partials) & 15) == 0),
               "kk_rowdot_bwd: partial rows need C %% 4 == 0 and 16-byte aligned x, dx, w and partial matrix");
    if (rows_form) {
        hipStream_t s = (hipStream_t)stream;
        const __bf16 *xb = reinterpret_cast<const __bf16 *>(x);
        const int nv = kk_cdiv(C, 256);
#define KK_RD(NV)                                                                                                                              \
    do {                                                                                                                                       \
        if (x_bf16) hipLaunchKernelGGL((rowdot_bwd_rows_kernel<__bf16, NV>), dim3(blocks), dim3(256), 0, s, dout, xb, w, mask, dx, dw, db, rows, C, L, chunk, rpb, partials); \
        else hipLaunchKernelGGL((rowdot_bwd_rows_kernel<float, NV>), dim3(blocks), dim3(256), 0, s, dout, x, w, mask, dx, dw, db, rows, C, L, chunk, rpb, partials);          \
    } while (0)
        if (nv <= 1) KK_RD(1); else if (nv <= 2) KK_RD(2); else KK_RD(4);
#undef KK_RD
        KK_LAUNCH_CHECK("kk_rowdot_bwd");
        return 0;
    }
    if (x_bf16)
        hipLaunchKernelGGL(rowdot_bwd_kernel<__bf16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dout,
                           reinterpret_cast<const __bf16 *>(x), w, mask, dx, dw, db, rows, C, L, chunk, rpb);
    else
        hipLaunchKernelGGL(rowdot_bwd_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dout, x, w, mask, dx, dw, db, rows,
                           C, L, chunk, rpb);
    KK_LAUNCH_CHECK("kk_rowdot_bwd");
    return 0;
}

extern "C" int kk_bucket_embed_add_fwd(const float *x, const float *pitch, const float *energy, const float *pbins,
                                       const float *ebins, const float *pemb, const float *eemb, const int64_t *lens,
                                       float *out, int32_t *pidx, int32_t *eidx, uint8_t *frame_mask, int B, int T,
                                       int H, int nbins, int out_bf16, void *stream) {
    KK_REQUIRE(B > 0 && T > 0 && H > 0 && H % 4 == 0 && nbins > 1, "kk_bucket_embed_add_fwd: bad shape");
    const int64_t rows = (int64_t)B * T;
    int blocks = kk_cdiv(rows, 4);
    if (blocks > 4096) blocks = 4096;
    if (out_bf16)
        hipLaunchKernelGGL(bucket_embed_add_fwd_kernel<__bf16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, pitch, energy,
                           pbins, ebins, pemb, eemb, lens, reinterpret_cast<__bf16 *>(out), pidx, eidx, frame_mask, rows, T, H, nbins);
    else
        hipLaunchKernelGGL(bucket_embed_add_fwd_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, pitch, energy, pbins,
                           ebins, pemb, eemb, lens, out, pidx, eidx, frame_mask, rows, T, H, nbins);
    KK_LAUNCH_CHECK("kk_bucket_embed_add_fwd");
    return 0;
}
extern "C" int kk_regulate_embed_fwd(const float *enc, const int64_t *idx, const float *pitch, const float *energy, const float *pbins,
                                     const float *ebins, const float *pemb, const float *eemb, const int64_t *lens, float *xf, float *out,
                                     int32_t *pidx, int32_t *eidx, uint8_t *frame_mask, int B, int P, int T, int H, int nbins, int out_bf16,
                                     const uint32_t *seed, uint32_t site, int time_mask_max, int feat_mask_max, int n_time, int n_feat,
                                     void *stream) {
    KK_REQUIRE(B > 0 && P > 0 && T > 0 && H > 0 && H % 4 == 0 && nbins > 1, "kk_regulate_embed_fwd: bad shape");
    KK_REQUIRE(enc && idx && pitch && energy && pbins && ebins && pemb && eemb && lens && xf && out && pidx && eidx && frame_mask,
               "kk_regulate_embed_fwd: null pointer");
    KK_REQUIRE(n_time >= 0 && n_time <= 16 && n_feat >= 0 && n_feat <= 16, "kk_regulate_embed_fwd: at most 16 masks of each kind");
    const int64_t rows = (int64_t)B * T;
    int blocks = kk_cdiv(rows, 4);
    if (blocks > 4096) blocks = 4096;
    if (out_bf16)
        hipLaunchKernelGGL(regulate_embed_fwd_kernel<__bf16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, enc, idx, pitch, energy, pbins,
                           ebins, pemb, eemb, lens, xf, reinterpret_cast<__bf16 *>(out), pidx, eidx, frame_mask, rows, P, T, H, nbins, seed,
                           site, time_mask_max, feat_mask_max, n_time, n_feat, kk_write_through(rows));
    else
        hipLaunchKernelGGL(regulate_embed_fwd_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, enc, idx, pitch, energy, pbins,
                           ebins, pemb, eemb, lens, xf, out, pidx, eidx, frame_mask, rows, P, T, H, nbins, seed, site, time_mask_max,
                           feat_mask_max, n_time, n_feat, kk_write_through(rows));
    KK_LAUNCH_CHECK("kk_regulate_embed_fwd");
    return 0;
}
extern "C" int kk_bucket_embed_add_bwd(const float *dout, const int32_t *pidx, const int32_t *eidx,
                                       const uint8_t *frame_mask, float *dpemb, float *deemb, int B, int T, int H,
                                       int nbins, void *stream) {
    KK_REQUIRE(B > 0 && T > 0 && H > 0 && nbins > 0, "kk_bucket_embed_add_bwd: bad shape");
    const int64_t rows = (int64_t)B * T;
    const size_t lds = (size_t)2 * nbins * BE_CW * sizeof(float);
    if (lds <= 64 * 1024) {
        hipLaunchKernelGGL(bucket_embed_add_bwd_lds_kernel, dim3(kk_cdiv(H, BE_CW), BE_RG), dim3(256), lds, (hipStream_t)stream, dout,
                           pidx, eidx, frame_mask, dpemb, deemb, rows, H, nbins);
        KK_LAUNCH_CHECK("kk_bucket_embed_add_bwd");
        return 0;
    }
    int blocks = kk_cdiv(rows, 4);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(bucket_embed_add_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dout, pidx, eidx,
                       frame_mask, dpemb, deemb, rows, H);
    KK_LAUNCH_CHECK("kk_bucket_embed_add_bwd");
    return 0;
}

extern "C" int kk_bucket_sort_items(int64_t rows, int nbins) { return nbins + (int)((rows + BS_PIECE - 1) / BS_PIECE); }
extern "C" int kk_bucket_sort(const int32_t *pidx, const int32_t *eidx, const uint8_t *frame_mask, int64_t rows, int nbins,
                              int32_t *order, int32_t *items, void *stream) {
    KK_REQUIRE(pidx && eidx && frame_mask && order && items && rows > 0 && rows < (1ll << 30) && nbins > 0 && nbins <= BS_MAXBINS,
               "kk_bucket_sort: bad args (at most %d bins)", BS_MAXBINS);
    KK_REQUIRE((reinterpret_cast<uintptr_t>(items) & 15) == 0, "kk_bucket_sort: items must be 16-byte aligned");
    hipLaunchKernelGGL(bucket_sort_kernel, dim3(2), dim3(BS_THREADS), 0, (hipStream_t)stream, pidx, eidx, frame_mask, rows, nbins,
                       kk_bucket_sort_items(rows, nbins), order, items);
    KK_LAUNCH_CHECK("kk_bucket_sort");
    return 0;
}
extern "C" int kk_bucket_embed_add_bwd_sorted(const float *dout, const int32_t *order, const int32_t *items, float *dpemb, float *deemb,
                                              int64_t rows, int H, int nbins, void *stream) {
    KK_REQUIRE(dout && order && items && dpemb && deemb && rows > 0 && H > 0 && H % 4 == 0 && nbins > 0 && nbins <= BS_MAXBINS,
               "kk_bucket_embed_add_bwd_sorted: bad args");
    const int max_items = kk_bucket_sort_items(rows, nbins);
    const int threads = H / 4 >= 256 ? 256 : (H / 4 + 63) / 64 * 64;
    hipLaunchKernelGGL(bucket_embed_add_bwd_sorted_kernel, dim3(max_items, 2), dim3(threads), 0, (hipStream_t)stream, dout, order, items,
                       dpemb, deemb, rows, H, max_items);
    KK_LAUNCH_CHECK("kk_bucket_embed_add_bwd_sorted");
    return 0;
}

extern "C" int kk_decode_prologue(const float *mel_all, float *frame_in, const float *pe, float *pe_row, const float *cos_t,
                                  const float *sin_t, float *cos_row, float *sin_row, uint8_t *key_mask, const int *t_dev, int B, int L1,
                                  int M, int H, void *stream) {
    KK_REQUIRE(mel_all && frame_in && pe && pe_row && cos_t && sin_t && cos_row && sin_row && key_mask && t_dev && B > 0 && L1 > 1 && M > 0 && H > 0,
               "kk_decode_prologue: bad args");
    hipLaunchKernelGGL(decode_prologue_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, mel_all, frame_in, pe, pe_row, cos_t, sin_t,
                       cos_row, sin_row, key_mask, t_dev, B, L1, M, H);
    KK_LAUNCH_CHECK("kk_decode_prologue");
    return 0;
}
extern "C" int kk_decode_cache_append(const void *nrm, void *q, void *kcache, void *vcache, const int *t_dev, int B, int H, int bf16,
                                      void *stream) {
    KK_REQUIRE(nrm && q && kcache && vcache && t_dev && B > 0 && H > 0, "kk_decode_cache_append: bad args");
    const int blocks = grid_for((int64_t)B * H, 64);
    if (bf16)
        hipLaunchKernelGGL(decode_cache_append_kernel<uint16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                           static_cast<const uint16_t *>(nrm), static_cast<uint16_t *>(q), static_cast<uint16_t *>(kcache),
                           static_cast<uint16_t *>(vcache), t_dev, B, H);
    else
        hipLaunchKernelGGL(decode_cache_append_kernel<uint32_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                           static_cast<const uint32_t *>(nrm), static_cast<uint32_t *>(q), static_cast<uint32_t *>(kcache),
                           static_cast<uint32_t *>(vcache), t_dev, B, H);
    KK_LAUNCH_CHECK("kk_decode_cache_append");
    return 0;
}
extern "C" int kk_decode_epilogue(const float *frame_out, const float *stop, float *mel_all, float *stop_all, int *t_dev, int B, int L1,
                                  int M, void *stream) {
    KK_REQUIRE(frame_out && stop && mel_all && stop_all && t_dev && B > 0 && L1 > 1 && M > 0, "kk_decode_epilogue: bad args");
    hipLaunchKernelGGL(decode_epilogue_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, frame_out, stop, mel_all, stop_all, t_dev, B, L1, M);
    KK_LAUNCH_CHECK("kk_decode_epilogue");
    return 0;
}

extern "C" int kk_ids_eq_zero(const int64_t *ids, uint8_t *mask, int64_t n, void *stream) {
    KK_REQUIRE(n > 0, "kk_ids_eq_zero: empty");
    hipLaunchKernelGGL(ids_eq_zero_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, ids, mask, n);
    KK_LAUNCH_CHECK("kk_ids_eq_zero");
    return 0;
}
extern "C" int kk_shift_right(const float *mel, float *out, int B, int T, int M, void *stream) {
    KK_REQUIRE(B > 0 && T > 0 && M > 0, "kk_shift_right: bad shape");
    const int64_t total = (int64_t)B * T * M;
    hipLaunchKernelGGL(shift_right_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, mel, out, total, T, M);
    KK_LAUNCH_CHECK("kk_shift_right");
    return 0;
}

// ------------------------------------------------------------------ batch hand-over
// Up to 16 device-to-device copies as ONE launch: a step's batch (ids, mel, durations, pitch, ... — nine small tensors)
// moves into the buffers the captured graphs read.  blockIdx.x walks 16 KiB chunks of the concatenation.
namespace {
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int COPY_MAX = 16;
constexpr int64_t COPY_CHUNK = 16384;
struct CopyMany {
    const char *src[COPY_MAX];
    char *dst[COPY_MAX];
    int64_t bytes[COPY_MAX];
    int start[COPY_MAX + 1];      // first chunk of each copy
    int n;
};
__global__ __launch_bounds__(256) void copy_many_kernel(CopyMany c) {
    int i = 0;
    while (i + 1 < c.n && (int)blockIdx.x >= c.start[i + 1]) ++i;
    const int64_t off = (int64_t)((int)blockIdx.x - c.start[i]) * COPY_CHUNK;
    const int64_t len = min(COPY_CHUNK, c.bytes[i] - off);
    const char *s = c.src[i] + off;
    char *d = c.dst[i] + off;
    if ((((uintptr_t)s | (uintptr_t)d) & 15) == 0) {
        const int64_t nv = len >> 4;
        for (int64_t j = threadIdx.x; j < nv; j += 256) reinterpret_cast<u32x4 *>(d)[j] = reinterpret_cast<const u32x4 *>(s)[j];
        for (int64_t j = (nv << 4) + threadIdx.x; j < len; j += 256) d[j] = s[j];
    } else {
        for (int64_t j = threadIdx.x; j < len; j += 256) d[j] = s[j];
    }
}
}  // namespace

extern "C" int kk_copy_many(const void *const *src, void *const *dst, const int64_t *bytes, int n, void *stream) {
    KK_REQUIRE(src && dst && bytes && n >= 1 && n <= COPY_MAX, "kk_copy_many: 1..16 copies per call");
    CopyMany c = {};
    c.n = n;
    for (int i = 0; i < n; ++i) {
        KK_REQUIRE(src[i] && dst[i] && bytes[i] > 0, "kk_copy_many: null pointer or empty copy");
        c.src[i] = static_cast<const char *>(src[i]);
        c.dst[i] = static_cast<char *>(dst[i]);
        c.bytes[i] = bytes[i];
        c.start[i + 1] = c.start[i] + (int)((bytes[i] + COPY_CHUNK - 1) / COPY_CHUNK);
    }
    hipLaunchKernelGGL(copy_many_kernel, dim3(c.start[n]), dim3(256), 0, (hipStream_t)stream, c);
    KK_LAUNCH_CHECK("kk_copy_many");
    return 0;
}

// ------------------------------------------------------------------ expanded length != mel length (model.py:607-628)
// When the durations of a batch expand to T' = max_b sum(dur) > T frames, the pitch / energy predictors see T' frames
// (variance_predictor.py:354-372) while the losses read their first T columns (losses.py:111,137) and the decoder memory
// is the first T frames.  These two helpers move [B, T'] <-> [B, T] rows and build the T'-frame padding mask.
namespace {
__global__ __launch_bounds__(256) void pad2d_kernel(const float *__restrict__ src, int64_t lds, int cols_src, float *__restrict__ dst,
                                                    int64_t ldd, int cols_dst, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / cols_dst;
        const int c = (int)(i - r * cols_dst);
        dst[r * ldd + c] = c < cols_src ? src[r * lds + c] : 0.f;
    }
}
__global__ __launch_bounds__(256) void frame_mask_kernel(const int64_t *__restrict__ lens, uint8_t *__restrict__ mask, int T, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / T;
        mask[i] = (i - b * T) >= lens[b] ? 1 : 0;
    }
}
}  // namespace

extern "C" int kk_pad2d_f32(const float *src, int64_t lds, int cols_src, float *dst, int64_t ldd, int cols_dst, int64_t rows,
                            void *stream) {
    KK_REQUIRE(src && dst && rows > 0 && cols_src > 0 && cols_dst > 0 && lds >= cols_src && ldd >= cols_dst, "kk_pad2d_f32: bad shape");
    const int64_t total = rows * cols_dst;
    hipLaunchKernelGGL(pad2d_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, src, lds, cols_src, dst, ldd, cols_dst, total);
    KK_LAUNCH_CHECK("kk_pad2d_f32");
    return 0;
}
extern "C" int kk_frame_mask(const int64_t *lens, uint8_t *mask, int B, int T, void *stream) {
    KK_REQUIRE(lens && mask && B > 0 && T > 0, "kk_frame_mask: bad shape");
    const int64_t total = (int64_t)B * T;
    hipLaunchKernelGGL(frame_mask_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, lens, mask, T, total);
    KK_LAUNCH_CHECK("kk_frame_mask");
    return 0;
}
