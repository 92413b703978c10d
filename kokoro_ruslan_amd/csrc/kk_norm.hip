// Normalisation kernels of the Kokoro train step — all HBM-bound, one wavefront per row.
//
//  LayerNorm      nn.LayerNorm(H), eps 1e-5           model/transformers.py:461-462,518-520,612; model.py:122
//  RMSNorm        nn.RMSNorm(H), eps = FLT_EPSILON    transformers.py:94,109-110 (GLU output_norm) + block residual
//  head-norm+RoPE nn.RMSNorm(64) per head + rotate-half  transformers.py:145-148,260-277; positional_encoding.py:196-209
//  GroupNorm(1,C) over (C x chunk frames) + ReLU      model/variance_predictor.py:55,77-87,102-106
//
// Rows are processed by a 64-lane wave holding the row in registers (float4 per lane per 256 columns), so a
// row is read once and written once; statistics are two-pass in registers (mean, then centred variance).
// Parameter gradients (gamma/beta/gain) are accumulated per workgroup in LDS and added to HBM with one
// atomic per column per workgroup.
#include "kk_common.h"
#include <float.h>

namespace {

constexpr int MAXV = 8;   // float4 per lane -> H <= 2048

__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// ------------------------------------------------------------------ LayerNorm
template <typename TY>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float *__restrict__ x, const float *__restrict__ gamma,
                                                            const float *__restrict__ beta, TY *__restrict__ y,
                                                            float *__restrict__ mean_o, float *__restrict__ rstd_o,
                                                            int64_t rows, int H) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *xr = x + row * H;
    float4 v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane * 4 + 256 * i;
        v[i] = c < H ? ld4(xr + c) : f4zero();
        s += v[i].x + v[i].y + v[i].z + v[i].w;
    }
    const float mean = wave_sum(s) / (float)H;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane * 4 + 256 * i;
        if (c < H) {
            const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
            q += a * a + b * b + cc * cc + d * d;
        }
    }
    const float rstd = 1.f / sqrtf(wave_sum(q) / (float)H + 1e-5f);
    TY *yr = y + row * H;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
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
        mean_o[row] = mean;
        rstd_o[row] = rstd;
    }
}

// Each wave takes LN_R rows per trip and issues all their loads before the first reduction, so a wave keeps
// LN_R x (x, dy, dx) rows in flight instead of one (the kernel is pure latency at 4 waves per CU otherwise).
// The per-column gain/bias gradients leave a workgroup either as 2H device-scope atomics (measured: ~29 ns per
// workgroup, 40 % of this kernel at 256 workgroups) or, when `partials` is given, as one plain row of
// partials[blockIdx.x][2H] that kk_partials_reduce sums later — once for all the norm layers of a step.
template <typename TD, int NV, int LN_R>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const TD *__restrict__ dy, const float *__restrict__ x,
                                                            const float *__restrict__ gamma, const float *__restrict__ mean,
                                                            const float *__restrict__ rstd, float *__restrict__ dx,
                                                            int dx_acc, float *__restrict__ dgamma, float *__restrict__ dbeta,
                                                            float *__restrict__ partials, int64_t rows, int H) {
    // [waves][2][H]: every wave leaves its column sums in a slab of its own (plain 16-byte stores; LDS float atomics move about one
    // lane per two clocks per CU: 4096 of them were ~3.4 us of this launch), the workgroup adds the four slabs after one barrier
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    float4 ag[NV], ab[NV], gm[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane * 4 + 256 * i;
        ag[i] = f4zero(); ab[i] = f4zero();
        gm[i] = c < H ? ld4(gamma + c) : f4zero();
    }
    const float invH = 1.f / (float)H;
    for (int64_t row0 = ((int64_t)blockIdx.x * wpb + (threadIdx.x >> 6)) * LN_R; row0 < rows; row0 += (int64_t)gridDim.x * wpb * LN_R) {
        float4 xv[LN_R][NV], dv[LN_R][NV], ov[LN_R][NV];
        float mu[LN_R], rs[LN_R];
#pragma unroll
        for (int r = 0; r < LN_R; ++r) {
            const int64_t row = row0 + r;
            const bool rv = row < rows;
            mu[r] = rv ? mean[row] : 0.f;
            rs[r] = rv ? rstd[row] : 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = lane * 4 + 256 * i;
                const bool ok = rv && c < H;
                xv[r][i] = ok ? ld4(x + row * H + c) : f4zero();
                dv[r][i] = ok ? ldv4<TD>(dy + row * H + c) : f4zero();
                ov[r][i] = (ok && dx_acc) ? ld4(dx + row * H + c) : f4zero();
            }
        }
#pragma unroll
        for (int r = 0; r < LN_R; ++r) {
            const int64_t row = row0 + r;
            if (row >= rows) break;                        // wave-uniform
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const float4 d = dv[r][i];
                float4 &xh = xv[r][i], &dg = dv[r][i];
                xh = make_float4((xh.x - mu[r]) * rs[r], (xh.y - mu[r]) * rs[r], (xh.z - mu[r]) * rs[r], (xh.w - mu[r]) * rs[r]);
                if (lane * 4 + 256 * i >= H) xh = f4zero();
                ag[i].x += d.x * xh.x; ag[i].y += d.y * xh.y; ag[i].z += d.z * xh.z; ag[i].w += d.w * xh.w;
                ab[i].x += d.x; ab[i].y += d.y; ab[i].z += d.z; ab[i].w += d.w;
                dg = make_float4(d.x * gm[i].x, d.y * gm[i].y, d.z * gm[i].z, d.w * gm[i].w);
                s1 += dg.x + dg.y + dg.z + dg.w;
                s2 += dg.x * xh.x + dg.y * xh.y + dg.z * xh.z + dg.w * xh.w;
            }
            s1 = wave_sum(s1) * invH;
            s2 = wave_sum(s2) * invH;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = lane * 4 + 256 * i;
                if (c < H) {
                    const float4 xh = xv[r][i], dg = dv[r][i];
                    float4 o = ov[r][i];
                    o.x += rs[r] * (dg.x - s1 - xh.x * s2);
                    o.y += rs[r] * (dg.y - s1 - xh.y * s2);
                    o.z += rs[r] * (dg.z - s1 - xh.z * s2);
                    o.w += rs[r] * (dg.w - s1 - xh.w * s2);
                    st4(dx + row * H + c, o);
                }
            }
        }
    }
    float *slab = sm + (size_t)(threadIdx.x >> 6) * 2 * H;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane * 4 + 256 * i;
        if (c < H) {
            st4(slab + c, ag[i]);
            st4(slab + H + c, ab[i]);
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * H; c += blockDim.x) {
        float s = sm[c];
        for (int w = 1; w < wpb; ++w) s += sm[(size_t)w * 2 * H + c];
        if (partials) partials[(int64_t)blockIdx.x * 2 * H + c] = s;
        else atomicAdd(c < H ? &dgamma[c] : &dbeta[c - H], s);
    }
}

// ------------------------------------------------------------------ RMSNorm (+ residual)
template <typename TX>
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(const TX *__restrict__ x, const float *__restrict__ gain,
                                                          const float *__restrict__ residual, float *__restrict__ y,
                                                          float *__restrict__ rstd_o, int64_t rows, int H) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float4 v[MAXV];
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane * 4 + 256 * i;
        v[i] = c < H ? ldv4<TX>(x + row * H + c) : f4zero();
        q += v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
    }
    const float rs = 1.f / sqrtf(wave_sum(q) / (float)H + FLT_EPSILON);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane * 4 + 256 * i;
        if (c < H) {
            const float4 g = ld4(gain + c);
            float4 o = make_float4(v[i].x * rs * g.x, v[i].y * rs * g.y, v[i].z * rs * g.z, v[i].w * rs * g.w);
            if (residual) { const float4 r = ld4(residual + row * H + c); o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w; }
            st4(y + row * H + c, o);
        }
    }
    if (lane == 0) rstd_o[row] = rs;
}

template <typename TX, int NV, int RR>
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const float *__restrict__ dy, const TX *__restrict__ x,
                                                          const float *__restrict__ gain, const float *__restrict__ rstd,
                                                          TX *__restrict__ dx, float *__restrict__ dgain, float *__restrict__ partials,
                                                          int64_t rows, int H) {
    extern __shared__ __attribute__((aligned(16))) float sm[];   // [H]
    for (int c = threadIdx.x; c < H; c += blockDim.x) sm[c] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    float4 ag[NV], gm[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        ag[i] = f4zero();
        gm[i] = lane * 4 + 256 * i < H ? ld4(gain + lane * 4 + 256 * i) : f4zero();
    }
    const float invH = 1.f / (float)H;
    // RR rows per wave per trip, all loads issued before the first reduction (see layernorm_bwd_kernel)
    for (int64_t row0 = ((int64_t)blockIdx.x * wpb + (threadIdx.x >> 6)) * RR; row0 < rows; row0 += (int64_t)gridDim.x * wpb * RR) {
        float4 xv[RR][NV], dv[RR][NV];
        float rs[RR];
#pragma unroll
        for (int r = 0; r < RR; ++r) {
            const int64_t row = row0 + r;
            const bool rv = row < rows;
            rs[r] = rv ? rstd[row] : 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = lane * 4 + 256 * i;
                const bool ok = rv && c < H;
                xv[r][i] = ok ? ldv4<TX>(x + row * H + c) : f4zero();
                dv[r][i] = ok ? ld4(dy + row * H + c) : f4zero();
            }
        }
#pragma unroll
        for (int r = 0; r < RR; ++r) {
            const int64_t row = row0 + r;
            if (row >= rows) break;
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const float4 d = dv[r][i], xx = xv[r][i];
                float4 &dg = dv[r][i];
                ag[i].x += d.x * xx.x * rs[r]; ag[i].y += d.y * xx.y * rs[r]; ag[i].z += d.z * xx.z * rs[r]; ag[i].w += d.w * xx.w * rs[r];
                dg = make_float4(d.x * gm[i].x, d.y * gm[i].y, d.z * gm[i].z, d.w * gm[i].w);
                s += dg.x * xx.x + dg.y * xx.y + dg.z * xx.z + dg.w * xx.w;
            }
            const float k = wave_sum(s) * invH * rs[r] * rs[r] * rs[r];
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = lane * 4 + 256 * i;
                const float4 dg = dv[r][i], xx = xv[r][i];
                if (c < H)
                    stv4<TX>(dx + row * H + c, make_float4(rs[r] * dg.x - xx.x * k, rs[r] * dg.y - xx.y * k,
                                                           rs[r] * dg.z - xx.z * k, rs[r] * dg.w - xx.w * k));
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane * 4 + 256 * i;
        if (c < H) { atomicAdd(&sm[c], ag[i].x); atomicAdd(&sm[c + 1], ag[i].y); atomicAdd(&sm[c + 2], ag[i].z); atomicAdd(&sm[c + 3], ag[i].w); }
    }
    __syncthreads();
    if (partials) {
        for (int c = threadIdx.x; c < H; c += blockDim.x) partials[(int64_t)blockIdx.x * H + c] = sm[c];
        return;
    }
    for (int c = threadIdx.x; c < H; c += blockDim.x) atomicAdd(&dgain[c], sm[c]);
}

// dst[c] += sum over the nblocks rows of a [nblocks][ncols] partial-sum matrix; one launch serves a whole list.
// grid (64-column slab, descriptor); 256 threads = 64 columns x 4 row groups.
__global__ __launch_bounds__(256) void partials_reduce_kernel(const KkReduceDesc *__restrict__ descs) {
    __shared__ float red[4][64];
    const KkReduceDesc d = descs[blockIdx.y];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
    if (blockIdx.x * 64 >= d.ncols) return;
    float s = 0.f;
    if (c < d.ncols) {
        const float *p = d.src + c;
        const int64_t stride = d.stride > 0 ? d.stride : d.ncols;
#pragma unroll 8
        for (int r = rg; r < d.nblocks; r += 4) s += p[(int64_t)r * stride];
    }
    red[rg][threadIdx.x & 63] = s;
    __syncthreads();
    if (rg == 0 && c < d.ncols) {
        float *dst = c < d.split ? d.dst0 + c : d.dst1 + (c - d.split);
        *dst += red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    }
}

// ------------------------------------------------------------------ per-head RMSNorm(64) + RoPE
// 16 lanes per (row, head) 64-vector, one float4 each (4 vectors per wave, 16 per workgroup); the sum of squares is
// a 4-step xor-shuffle inside the 16-lane group and rotate_half's partner element d^32 lives in lane^8.
// One launch covers up to three column groups ("parts": q|k|v of a fused projection) with their own gains and a
// per-part RoPE flag; blockIdx.y = part, so a thread's gain-gradient accumulator belongs to one gain vector.
// rotate_half(n)[d] = -n[d+32] (d<32), n[d-32] (d>=32)   (positional_encoding.py:152-157)
struct HeadNormArgs {
    const void *x, *dy;          // fp32 or bf16 (template parameter of the kernels)
    const float *cos_t, *sin_t;
    void *y, *dx;
    const float *gain[3];
    float *dgain[3];
    float *partials;             // optional [parts][gridDim.x][64] partial gain gradients (see layernorm_bwd_kernel)
    int64_t ldx, ldy, lddy, lddx, npairs;   // npairs = rows * heads (per part)
    int heads, S, rope_mask;
};

__device__ __forceinline__ float sum16(float v) { return kk_row16_sum(v); }
__device__ __forceinline__ float4 shfl8(const float4 &v) { return kk_shfl8(v); }

template <typename T>
__global__ __launch_bounds__(256) void headnorm_rope_fwd_kernel(HeadNormArgs a) {
    const T *x = static_cast<const T *>(a.x);
    T *y = static_cast<T *>(a.y);
    const int part = blockIdx.y, sub = threadIdx.x & 15, H = a.heads * 64;
    const bool rope = (a.rope_mask >> part) & 1;
    const float4 g = ld4(a.gain[part] + sub * 4);
    for (int64_t pr = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4); pr < a.npairs; pr += (int64_t)gridDim.x * 16) {
        const int64_t row = pr / a.heads;
        const int col = part * H + (int)(pr - row * a.heads) * 64 + sub * 4;
        const float4 v = ldv4<T>(x + row * a.ldx + col);
        const int pos = rope ? (int)(row % a.S) : 0;
        stv4<T>(y + row * a.ldy + col, kk_headnorm_rope(v, g, rope, a.cos_t + pos * 64, a.sin_t + pos * 64, sub));
    }
}

template <typename T>
__global__ __launch_bounds__(256) void headnorm_rope_bwd_kernel(HeadNormArgs a) {
    __shared__ float red[16][64];
    const T *x = static_cast<const T *>(a.x), *dy = static_cast<const T *>(a.dy);
    T *dx = static_cast<T *>(a.dx);
    const int part = blockIdx.y, sub = threadIdx.x & 15, H = a.heads * 64;
    const bool rope = (a.rope_mask >> part) & 1;
    const float4 g = ld4(a.gain[part] + sub * 4);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    constexpr int U = 4;     // (row, head) vectors in flight per 16-lane group: all loads of a trip precede its reductions
    for (int64_t pr0 = ((int64_t)blockIdx.x * 16 + (threadIdx.x >> 4)) * U; pr0 < a.npairs; pr0 += (int64_t)gridDim.x * 16 * U) {
        float4 vv[U], dd[U], cc[U], ss[U];
        int64_t off_x[U], off_o[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t pr = pr0 + u;
            const bool ok = pr < a.npairs;
            const int64_t row = ok ? pr / a.heads : 0;
            const int col = part * H + (int)((ok ? pr : 0) - row * a.heads) * 64 + sub * 4;
            off_x[u] = row * a.ldx + col;
            off_o[u] = row * a.lddx + col;
            vv[u] = ok ? ldv4<T>(x + off_x[u]) : make_float4(0.f, 0.f, 0.f, 0.f);
            dd[u] = ok ? ldv4<T>(dy + row * a.lddy + col) : make_float4(0.f, 0.f, 0.f, 0.f);
            if (rope) {
                const int pos = (int)(row % a.S);
                cc[u] = ld4(a.cos_t + pos * 64 + sub * 4);
                ss[u] = ld4(a.sin_t + pos * 64 + sub * 4);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (pr0 + u >= a.npairs) break;                 // uniform across the 16-lane group
            const float4 v = vv[u];
            float4 dn = dd[u];
            const float rs = 1.f / sqrtf(sum16(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w) * (1.f / 64.f) + FLT_EPSILON);
            if (rope) {     // dn[d] = dy[d] cos[d] + (d < 32 ? dy[d+32] sin[d+32] : -dy[d-32] sin[d-32])
                const float4 c = cc[u], sn = ss[u];
                const float4 o = shfl8(make_float4(dn.x * sn.x, dn.y * sn.y, dn.z * sn.z, dn.w * sn.w));
                const float sg = sub < 8 ? 1.f : -1.f;
                dn = make_float4(dn.x * c.x + sg * o.x, dn.y * c.y + sg * o.y, dn.z * c.z + sg * o.z, dn.w * c.w + sg * o.w);
            }
            acc.x += dn.x * v.x * rs; acc.y += dn.y * v.y * rs; acc.z += dn.z * v.z * rs; acc.w += dn.w * v.w * rs;
            const float4 dg = make_float4(dn.x * g.x, dn.y * g.y, dn.z * g.z, dn.w * g.w);
            const float k = sum16(dg.x * v.x + dg.y * v.y + dg.z * v.z + dg.w * v.w) * (1.f / 64.f) * rs * rs * rs;
            stv4<T>(dx + off_o[u], make_float4(rs * dg.x - v.x * k, rs * dg.y - v.y * k, rs * dg.z - v.z * k, rs * dg.w - v.w * k));
        }
    }
    st4(&red[threadIdx.x >> 4][sub * 4], acc);
    __syncthreads();
    if (threadIdx.x < 64) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) s += red[i][threadIdx.x];
        if (a.partials) a.partials[((int64_t)part * gridDim.x + blockIdx.x) * 64 + threadIdx.x] = s;
        else atomicAdd(&a.dgain[part][threadIdx.x], s);
    }
}

// ------------------------------------------------------------------ GroupNorm(1,C) per 512-frame chunk + ReLU
__device__ __forceinline__ int chunk_frames(int L, int chunk, int ci) {
    const int beg = ci * chunk;
    return (L - beg) < chunk ? (L - beg) : chunk;
}

// scratch[(b*nch+ci)*2 + {0,1}] += (sum, sumsq) of slice `blockIdx.x` of the chunk (contiguous frames*C floats).
__global__ __launch_bounds__(256) void gn_partial_kernel(const float *__restrict__ x, double *__restrict__ scratch, int L,
                                                         int C, int chunk, int nch, int slices) {
    __shared__ double red[4];
    const int bc = blockIdx.y, b = bc / nch, ci = bc % nch;
    const int64_t n = (int64_t)chunk_frames(L, chunk, ci) * C;
    const float *base = x + ((int64_t)b * L + (int64_t)ci * chunk) * C;
    const int64_t per = ((n / 4 + slices - 1) / slices) * 4;
    const int64_t beg = (int64_t)blockIdx.x * per, end = beg + per < n ? beg + per : n;
    float s = 0.f, q = 0.f;
    for (int64_t i = beg + threadIdx.x * 4; i < end; i += 1024) {
        const float4 v = ld4(base + i);
        s += v.x + v.y + v.z + v.w;
        q += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    const double ds = block_sum_256_d((double)s, red);
    const double dq = block_sum_256_d((double)q, red);
    if (threadIdx.x == 0 && beg < end) {
        atomicAdd(&scratch[bc * 2], ds);
        atomicAdd(&scratch[bc * 2 + 1], dq);
    }
}

__global__ void gn_finalize_kernel(const double *__restrict__ scratch, float *__restrict__ stats, int L, int C, int chunk,
                                   int nch, int total) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const double n = (double)chunk_frames(L, chunk, i % nch) * C;
    const double mean = scratch[i * 2] / n;
    double var = scratch[i * 2 + 1] / n - mean * mean;
    if (var < 0) var = 0;
    stats[i * 2] = (float)mean;
    stats[i * 2 + 1] = (float)(1.0 / sqrt(var + 1e-5));
}

__global__ __launch_bounds__(256) void gn_apply_relu_kernel(const float *__restrict__ x, const float *__restrict__ gamma,
                                                            const float *__restrict__ beta, const float *__restrict__ stats,
                                                            float *__restrict__ y, int64_t total4, int L, int C, int chunk, int nch,
                                                            const uint32_t *__restrict__ seedp, uint32_t site, float p) {
    // dropout after the ReLU (variance_predictor.py:106) is fused: dropped elements are stored as 0, kept ones scaled
    const uint32_t thr = seedp ? kk_drop_threshold(p) : 0u, seed = thr ? *seedp : 0u;
    const float ik = thr ? 1.f / (1.f - p) : 1.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
        const int64_t e = i * 4;
        const int c = (int)(e % C);
        const int64_t fr = e / C;
        const int l = (int)(fr % L), b = (int)(fr / L), ci = l / chunk;
        float4 o = f4zero();
        if (chunk_frames(L, chunk, ci) >= 2) {
            const float mu = stats[(b * nch + ci) * 2], rs = stats[(b * nch + ci) * 2 + 1];
            const float4 v = ld4(x + e), g = ld4(gamma + c), bt = ld4(beta + c);
            o.x = fmaxf((v.x - mu) * rs * g.x + bt.x, 0.f);
            o.y = fmaxf((v.y - mu) * rs * g.y + bt.y, 0.f);
            o.z = fmaxf((v.z - mu) * rs * g.z + bt.z, 0.f);
            o.w = fmaxf((v.w - mu) * rs * g.w + bt.w, 0.f);
            if (thr) {
                float m[4];
                kk_drop_mul4(seed, site, (uint64_t)e, thr, ik, m);
                o.x *= m[0]; o.y *= m[1]; o.z *= m[2]; o.w *= m[3];
            }
        }
        st4(y + e, o);
    }
}

// Backward pass 1: per (b,chunk) sums of dxhat and dxhat*xhat (double scratch) + dgamma/dbeta column sums.
// Block = 256 threads arranged as (256/C) frame-lanes x C columns; blockIdx.x = frame slab, blockIdx.y = (b,chunk).
__global__ __launch_bounds__(256) void gn_bwd_partial_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                             const float *__restrict__ y, const float *__restrict__ gamma,
                                                             const float *__restrict__ stats, double *__restrict__ scratch,
                                                             float *__restrict__ dgamma, float *__restrict__ dbeta, int L, int C,
                                                             int chunk, int nch, int slabs, float inv_keep) {
    __shared__ double red[4];
    const int bc = blockIdx.y, b = bc / nch, ci = bc % nch;
    const int nf = chunk_frames(L, chunk, ci);
    const int lanes = 256 / C, c = threadIdx.x % C, fl = threadIdx.x / C;
    const int per = (nf + slabs - 1) / slabs;
    const int fbeg = blockIdx.x * per, fend = fbeg + per < nf ? fbeg + per : nf;
    const float mu = stats[bc * 2], rs = stats[bc * 2 + 1], g = gamma[c];
    const int64_t base = ((int64_t)b * L + (int64_t)ci * chunk) * C;
    float ag = 0.f, ab = 0.f, s1 = 0.f, s2 = 0.f;
    if (nf >= 2)
#pragma unroll 4
        for (int f = fbeg + fl; f < fend; f += lanes) {             // (load-latency bound: few frames per block, loads unrolled)
            const int64_t o = base + (int64_t)f * C + c;
            const float d = y[o] > 0.f ? dy[o] * inv_keep : 0.f;   // y == 0 also where dropout removed the element
            const float xh = (x[o] - mu) * rs;
            ag += d * xh; ab += d;
            s1 += d * g; s2 += d * g * xh;
        }
    if (ag != 0.f || ab != 0.f) { atomicAdd(&dgamma[c], ag); atomicAdd(&dbeta[c], ab); }
    const double d1 = block_sum_256_d((double)s1, red);
    const double d2 = block_sum_256_d((double)s2, red);
    if (threadIdx.x == 0 && fbeg < fend) {
        atomicAdd(&scratch[bc * 2], d1);
        atomicAdd(&scratch[bc * 2 + 1], d2);
    }
}

// The same pass with 16-byte loads: a thread owns 4 channels, the workgroup's 1024 / C frame lanes walk the slab with every load of
// four frames in flight (the scalar form above keeps 12 four-byte loads in flight per thread: 25-33 us per launch on the side branch at
// 4096 x 256).  Same grid — the launch stays as thin as the comment at its call site wants it, it just holds its CUs for less time.
__global__ __launch_bounds__(256) void gn_bwd_partial_v4_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                                const float *__restrict__ y, const float *__restrict__ gamma,
                                                                const float *__restrict__ stats, double *__restrict__ scratch,
                                                                float *__restrict__ dgamma, float *__restrict__ dbeta, int L, int C,
                                                                int chunk, int nch, int slabs, float inv_keep) {
    __shared__ double red[4];
    __shared__ float colred[2][1024];                           // [dgamma | dbeta][frame lane][C]: lanes * C = 1024
    const int bc = blockIdx.y, b = bc / nch, ci = bc % nch;
    const int nf = chunk_frames(L, chunk, ci);
    const int C4 = C >> 2, lanes = 256 / C4, c = (threadIdx.x % C4) * 4, fl = threadIdx.x / C4;
    const int per = (nf + slabs - 1) / slabs;
    const int fbeg = blockIdx.x * per, fend = fbeg + per < nf ? fbeg + per : nf;
    const float mu = stats[bc * 2], rs = stats[bc * 2 + 1];
    const float4 g = ld4(gamma + c);
    const int64_t base = ((int64_t)b * L + (int64_t)ci * chunk) * C;
    float ag[4] = {0.f, 0.f, 0.f, 0.f}, ab[4] = {0.f, 0.f, 0.f, 0.f};
    float s1 = 0.f, s2 = 0.f;
    if (nf >= 2) {
        constexpr int U = 4;
        for (int f0 = fbeg + fl; f0 < fend; f0 += lanes * U) {
            float4 dv[U], xv[U], yv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int f = f0 + u * lanes, ff = f < fend ? f : fbeg;
                const int64_t o = base + (int64_t)ff * C + c;
                dv[u] = ld4(dy + o); xv[u] = ld4(x + o); yv[u] = ld4(y + o);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (f0 + u * lanes >= fend) continue;
                const float dd[4] = {dv[u].x, dv[u].y, dv[u].z, dv[u].w}, xx[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w};
                const float yy[4] = {yv[u].x, yv[u].y, yv[u].z, yv[u].w}, gg[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = yy[e] > 0.f ? dd[e] * inv_keep : 0.f;
                    const float xh = (xx[e] - mu) * rs;
                    ag[e] += d * xh; ab[e] += d;
                    s1 += d * gg[e]; s2 += d * gg[e] * xh;
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { colred[0][fl * C + c + e] = ag[e]; colred[1][fl * C + c + e] = ab[e]; }
    __syncthreads();
    if ((int)threadIdx.x < C) {
        float tg = 0.f, tb = 0.f;
        for (int l = 0; l < lanes; ++l) { tg += colred[0][l * C + threadIdx.x]; tb += colred[1][l * C + threadIdx.x]; }
        if (tg != 0.f || tb != 0.f) { atomicAdd(&dgamma[threadIdx.x], tg); atomicAdd(&dbeta[threadIdx.x], tb); }
    }
    const double d1 = block_sum_256_d((double)s1, red);
    const double d2 = block_sum_256_d((double)s2, red);
    if (threadIdx.x == 0 && fbeg < fend) {
        atomicAdd(&scratch[bc * 2], d1);
        atomicAdd(&scratch[bc * 2 + 1], d2);
    }
}

template <typename TO>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                           const float *__restrict__ y, const float *__restrict__ gamma,
                                                           const float *__restrict__ stats, const double *__restrict__ scratch,
                                                           TO *__restrict__ dx, int64_t total4, int L, int C, int chunk, int nch,
                                                           float inv_keep) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
        const int64_t e = i * 4;
        const int c = (int)(e % C);
        const int64_t fr = e / C;
        const int l = (int)(fr % L), b = (int)(fr / L), ci = l / chunk, bc = b * nch + ci;
        const int nf = chunk_frames(L, chunk, ci);
        float4 o = f4zero();
        if (nf >= 2) {
            const double n = (double)nf * C;
            const float m1 = (float)(scratch[bc * 2] / n), m2 = (float)(scratch[bc * 2 + 1] / n);
            const float mu = stats[bc * 2], rs = stats[bc * 2 + 1];
            float4 d = ld4(dy + e);
            d.x *= inv_keep; d.y *= inv_keep; d.z *= inv_keep; d.w *= inv_keep;
            const float4 xv = ld4(x + e), yv = ld4(y + e), g = ld4(gamma + c);
            o.x = rs * ((yv.x > 0.f ? d.x * g.x : 0.f) - m1 - (xv.x - mu) * rs * m2);
            o.y = rs * ((yv.y > 0.f ? d.y * g.y : 0.f) - m1 - (xv.y - mu) * rs * m2);
            o.z = rs * ((yv.z > 0.f ? d.z * g.z : 0.f) - m1 - (xv.z - mu) * rs * m2);
            o.w = rs * ((yv.w > 0.f ? d.w * g.w : 0.f) - m1 - (xv.w - mu) * rs * m2);
        }
        stv4<TO>(dx + e, o);
    }
}

inline int row_blocks(int64_t rows) { return kk_cdiv(rows, 4); }

}  // namespace

#define KK_CHECK_H(name)                                                                                  \
    KK_REQUIRE(rows > 0 && H > 0 && H % 4 == 0 && H <= 256 * MAXV, name ": unsupported rows=%ld H=%d", \
               (long)rows, H)

extern "C" int kk_layernorm_fwd(const float *x, const float *gamma, const float *beta, float *y, float *mean,
                                float *rstd, int64_t rows, int H, int y_bf16, void *stream) {
    KK_CHECK_H("kk_layernorm_fwd");
    if (y_bf16)
        hipLaunchKernelGGL(layernorm_fwd_kernel<__bf16>, dim3(row_blocks(rows)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta,
                           reinterpret_cast<__bf16 *>(y), mean, rstd, rows, H);
    else
        hipLaunchKernelGGL(layernorm_fwd_kernel<float>, dim3(row_blocks(rows)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta,
                           y, mean, rstd, rows, H);
    KK_LAUNCH_CHECK("kk_layernorm_fwd");
    return 0;
}

// workgroups (= rows of `partials`) that kk_layernorm_bwd / kk_rmsnorm_bwd launch for this shape
extern "C" int kk_norm_bwd_blocks(int64_t rows, int H) {
    const int nv = kk_cdiv(H, 256), R = nv <= 2 ? 4 : (nv <= 4 ? 2 : 1);
    int blocks = kk_cdiv(rows, 4 * R);
    if (blocks > 256) blocks = 256;
    return blocks < 1 ? 1 : blocks;
}

extern "C" int kk_partials_reduce(const KkReduceDesc *descs, int n, int max_cols, void *stream) {
    KK_REQUIRE(descs && n > 0 && max_cols > 0, "kk_partials_reduce: bad args");
    hipLaunchKernelGGL(partials_reduce_kernel, dim3(kk_cdiv(max_cols, 64), n), dim3(256), 0, (hipStream_t)stream, descs);
    KK_LAUNCH_CHECK("kk_partials_reduce");
    return 0;
}

extern "C" int kk_layernorm_bwd(const float *dy, const float *x, const float *gamma, const float *mean,
                                const float *rstd, float *dx, int dx_accumulate, float *dgamma, float *dbeta,
                                float *partials, int64_t rows, int H, int dy_bf16, void *stream) {
    KK_CHECK_H("kk_layernorm_bwd");
    const int nv = kk_cdiv(H, 256);
    const int blocks = kk_norm_bwd_blocks(rows, H);            // (rows in flight per wave: 4 / 2 / 1 by register budget)
    const size_t shm = (size_t)4 * 2 * H * sizeof(float);     // one [2][H] slab per wave
    hipStream_t st = (hipStream_t)stream;
#define KK_LN_BWD(TD, NV, R)                                                                                                  \
    hipLaunchKernelGGL((layernorm_bwd_kernel<TD, NV, R>), dim3(blocks), dim3(256), shm, st, reinterpret_cast<const TD *>(dy), x, gamma, \
                       mean, rstd, dx, dx_accumulate, dgamma, dbeta, partials, rows, H)
    if (dy_bf16) {
        if (nv <= 1) KK_LN_BWD(__bf16, 1, 4); else if (nv <= 2) KK_LN_BWD(__bf16, 2, 4); else if (nv <= 4) KK_LN_BWD(__bf16, 4, 2); else KK_LN_BWD(__bf16, 8, 1);
    } else {
        if (nv <= 1) KK_LN_BWD(float, 1, 4); else if (nv <= 2) KK_LN_BWD(float, 2, 4); else if (nv <= 4) KK_LN_BWD(float, 4, 2); else KK_LN_BWD(float, 8, 1);
    }
#undef KK_LN_BWD
    KK_LAUNCH_CHECK("kk_layernorm_bwd");
    return 0;
}

extern "C" int kk_rmsnorm_fwd(const float *x, const float *gain, const float *residual, float *y, float *rstd,
                              int64_t rows, int H, int x_bf16, void *stream) {
    KK_CHECK_H("kk_rmsnorm_fwd");
    if (x_bf16)
        hipLaunchKernelGGL(rmsnorm_fwd_kernel<__bf16>, dim3(row_blocks(rows)), dim3(256), 0, (hipStream_t)stream,
                           reinterpret_cast<const __bf16 *>(x), gain, residual, y, rstd, rows, H);
    else
        hipLaunchKernelGGL(rmsnorm_fwd_kernel<float>, dim3(row_blocks(rows)), dim3(256), 0, (hipStream_t)stream, x, gain, residual,
                           y, rstd, rows, H);
    KK_LAUNCH_CHECK("kk_rmsnorm_fwd");
    return 0;
}

extern "C" int kk_rmsnorm_bwd(const float *dy, const float *x, const float *gain, const float *rstd, float *dx,
                              float *dgain, float *partials, int64_t rows, int H, int x_bf16, void *stream) {
    KK_CHECK_H("kk_rmsnorm_bwd");
    const int nv = kk_cdiv(H, 256);
    const int blocks = kk_norm_bwd_blocks(rows, H);
    hipStream_t st = (hipStream_t)stream;
#define KK_RMS_BWD(TX, NV, R)                                                                                              \
    hipLaunchKernelGGL((rmsnorm_bwd_kernel<TX, NV, R>), dim3(blocks), dim3(256), H * sizeof(float), st, dy, reinterpret_cast<const TX *>(x), \
                       gain, rstd, reinterpret_cast<TX *>(dx), dgain, partials, rows, H)
    if (x_bf16) {
        if (nv <= 1) KK_RMS_BWD(__bf16, 1, 4); else if (nv <= 2) KK_RMS_BWD(__bf16, 2, 4); else if (nv <= 4) KK_RMS_BWD(__bf16, 4, 2); else KK_RMS_BWD(__bf16, 8, 1);
    } else {
        if (nv <= 1) KK_RMS_BWD(float, 1, 4); else if (nv <= 2) KK_RMS_BWD(float, 2, 4); else if (nv <= 4) KK_RMS_BWD(float, 4, 2); else KK_RMS_BWD(float, 8, 1);
    }
#undef KK_RMS_BWD
    KK_LAUNCH_CHECK("kk_rmsnorm_bwd");
    return 0;
}

extern "C" int kk_headnorm_rope_fwd(const float *x, int64_t ldx, float *y, int64_t ldy, int64_t rows, int heads, int S,
                                    int parts, const float *gain0, const float *gain1, const float *gain2, int rope_mask,
                                    const float *cos_t, const float *sin_t, int io_bf16, void *stream) {
    KK_REQUIRE(rows > 0 && heads > 0 && S > 0 && parts >= 1 && parts <= 3 && gain0, "kk_headnorm_rope_fwd: bad shape");
    KK_REQUIRE(rope_mask == 0 || (cos_t && sin_t), "kk_headnorm_rope_fwd: RoPE needs cos/sin tables");
    KK_REQUIRE(ldx % 4 == 0 && ldy % 4 == 0, "kk_headnorm_rope_fwd: strides must be multiples of 4");
    HeadNormArgs a = {};
    a.x = x; a.y = y; a.cos_t = cos_t; a.sin_t = sin_t; a.gain[0] = gain0; a.gain[1] = gain1; a.gain[2] = gain2;
    a.ldx = ldx; a.ldy = ldy; a.npairs = rows * heads; a.heads = heads; a.S = S; a.rope_mask = rope_mask;
    int blocks = kk_cdiv(a.npairs, 16 * 2);
    blocks = blocks > 2048 ? 2048 : (blocks < 1 ? 1 : blocks);
    if (io_bf16) hipLaunchKernelGGL(headnorm_rope_fwd_kernel<__bf16>, dim3(blocks, parts), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(headnorm_rope_fwd_kernel<float>, dim3(blocks, parts), dim3(256), 0, (hipStream_t)stream, a);
    KK_LAUNCH_CHECK("kk_headnorm_rope_fwd");
    return 0;
}

// workgroups per part (= rows of each [blocks][64] partial matrix) that kk_headnorm_rope_bwd launches
extern "C" int kk_headnorm_bwd_blocks(int64_t rows, int heads) {
    int blocks = kk_cdiv(rows * heads, 16 * 4 * 2);       // two trips of 4 vectors per 16-lane group
    return blocks > 512 ? 512 : (blocks < 1 ? 1 : blocks);
}

extern "C" int kk_headnorm_rope_bwd(const float *dy, int64_t lddy, const float *x, int64_t ldx, float *dx, int64_t lddx,
                                    int64_t rows, int heads, int S, int parts, const float *gain0, const float *gain1,
                                    const float *gain2, float *dgain0, float *dgain1, float *dgain2, float *partials,
                                    int rope_mask, const float *cos_t, const float *sin_t, int io_bf16, void *stream) {
    KK_REQUIRE(rows > 0 && heads > 0 && S > 0 && parts >= 1 && parts <= 3 && gain0 && dgain0, "kk_headnorm_rope_bwd: bad shape");
    KK_REQUIRE(rope_mask == 0 || (cos_t && sin_t), "kk_headnorm_rope_bwd: RoPE needs cos/sin tables");
    KK_REQUIRE(ldx % 4 == 0 && lddy % 4 == 0 && lddx % 4 == 0, "kk_headnorm_rope_bwd: strides must be multiples of 4");
    HeadNormArgs a = {};
    a.x = x; a.dy = dy; a.dx = dx; a.cos_t = cos_t; a.sin_t = sin_t;
    a.gain[0] = gain0; a.gain[1] = gain1; a.gain[2] = gain2; a.dgain[0] = dgain0; a.dgain[1] = dgain1; a.dgain[2] = dgain2;
    a.ldx = ldx; a.lddy = lddy; a.lddx = lddx; a.npairs = rows * heads; a.heads = heads; a.S = S; a.rope_mask = rope_mask;
    a.partials = partials;
    const int blocks = kk_headnorm_bwd_blocks(rows, heads);
    if (io_bf16) hipLaunchKernelGGL(headnorm_rope_bwd_kernel<__bf16>, dim3(blocks, parts), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(headnorm_rope_bwd_kernel<float>, dim3(blocks, parts), dim3(256), 0, (hipStream_t)stream, a);
    KK_LAUNCH_CHECK("kk_headnorm_rope_bwd");
    return 0;
}

extern "C" int kk_groupnorm_relu_fwd(const float *x, const float *gamma, const float *beta, float *y, float *stats,
                                     double *scratch, int B, int L, int C, int chunk, const uint32_t *seed, uint32_t site,
                                     float p, void *stream) {
    KK_REQUIRE(B > 0 && L > 0 && C > 0 && C % 4 == 0 && chunk > 0 && p >= 0.f && p < 1.f, "kk_groupnorm_relu_fwd: bad shape");
    hipStream_t s = (hipStream_t)stream;
    const int nch = kk_cdiv(L, chunk), total = B * nch;
    const int e = kk_zero_async(scratch, sizeof(double) * 2 * total, s);
    if (e != 0) return e;
    const int slices = 32;
    hipLaunchKernelGGL(gn_partial_kernel, dim3(slices, total), dim3(256), 0, s, x, scratch, L, C, chunk, nch, slices);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(kk_cdiv(total, 64)), dim3(64), 0, s, scratch, stats, L, C, chunk, nch, total);
    const int64_t total4 = (int64_t)B * L * C / 4;
    int blocks = kk_cdiv(total4, 256);
    { static const int ec = kk_tune_env("KK_ELEM_GRID_CAP", 1024); const int cap_ = ec > 0 ? ec : 4096; if (blocks > cap_) blocks = cap_; }      // (see grid_for, kk_elem.hip)
    hipLaunchKernelGGL(gn_apply_relu_kernel, dim3(blocks), dim3(256), 0, s, x, gamma, beta, stats, y, total4, L, C, chunk, nch,
                       p > 0.f ? seed : nullptr, site, p);
    KK_LAUNCH_CHECK("kk_groupnorm_relu_fwd");
    return 0;
}

extern "C" int kk_groupnorm_relu_bwd(const float *dy, const float *x, const float *y, const float *gamma,
                                     const float *stats, float *dx, float *dgamma, float *dbeta, double *scratch,
                                     int B, int L, int C, int chunk, float p, int dx_bf16, void *stream) {
    KK_REQUIRE(B > 0 && L > 0 && C > 0 && C % 4 == 0 && chunk > 0 && p >= 0.f && p < 1.f, "kk_groupnorm_relu_bwd: bad shape");
    const float inv_keep = 1.f / (1.f - p);
    KK_REQUIRE(C <= 256 && 256 % C == 0, "kk_groupnorm_relu_bwd: C=%d must divide 256", C);
    hipStream_t s = (hipStream_t)stream;
    const int nch = kk_cdiv(L, chunk), total = B * nch;
    const int e = kk_zero_async(scratch, sizeof(double) * 2 * total, s);
    if (e != 0) return e;
    // 16 slabs x (b, chunk): a thin grid on purpose.  This runs on the side branch beside the decoder backward; with 64 or
    // 128 slabs the kernel itself is 4x faster and the train step 1 % SLOWER (796K -> 790K -> 786K frames/s) — a burst of
    // workgroups disturbs the critical chain more than a long thin launch does; 8 and 4 slabs are slower again.
    const int slabs = 16;
    static const int v4 = kk_tune_env("KK_GN_BWD_V4", 1);
    if (v4 && C >= 4 && ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(gamma)) & 15) == 0)
        hipLaunchKernelGGL(gn_bwd_partial_v4_kernel, dim3(slabs, total), dim3(256), 0, s, dy, x, y, gamma, stats, scratch, dgamma,
                           dbeta, L, C, chunk, nch, slabs, inv_keep);
    else
        hipLaunchKernelGGL(gn_bwd_partial_kernel, dim3(slabs, total), dim3(256), 0, s, dy, x, y, gamma, stats, scratch, dgamma,
                           dbeta, L, C, chunk, nch, slabs, inv_keep);
    const int64_t total4 = (int64_t)B * L * C / 4;
    int blocks = kk_cdiv(total4, 256);
    { static const int ec = kk_tune_env("KK_ELEM_GRID_CAP", 1024); const int cap_ = ec > 0 ? ec : 4096; if (blocks > cap_) blocks = cap_; }      // (see grid_for, kk_elem.hip)
    if (dx_bf16)
        hipLaunchKernelGGL(gn_bwd_apply_kernel<__bf16>, dim3(blocks), dim3(256), 0, s, dy, x, y, gamma, stats, scratch,
                           reinterpret_cast<__bf16 *>(dx), total4, L, C, chunk, nch, inv_keep);
    else
        hipLaunchKernelGGL(gn_bwd_apply_kernel<float>, dim3(blocks), dim3(256), 0, s, dy, x, y, gamma, stats, scratch, dx, total4, L, C,
                           chunk, nch, inv_keep);
    KK_LAUNCH_CHECK("kk_groupnorm_relu_bwd");
    return 0;
}
