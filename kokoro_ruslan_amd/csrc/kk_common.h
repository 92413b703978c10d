// Shared helpers for the gfx950 kernels (wave = 64 lanes; MFMA 32x32 tiles).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdarg.h>
#include <stdio.h>
#include <type_traits>
#include "../../include/kokoro_hip.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;

int kk_fail(int code, const char *fmt, ...);
void kk_note_kernel(const char *name);         // kk_last_kernel(): `name` must have static storage duration
void kk_note_kernelf(const char *fmt, ...);    // ... a formatted name (interned)

// ---- launch capture (kk_chain.hip, round 6): between kk_chain_begin() and kk_chain_launch() the entry points of a sub-layer do not
// launch: their dispatch code runs as always (same tile policy, same argument blocks, same route record) and the launch it would
// have made — kernel name, grid, the by-value argument block — is RECORDED on the calling thread.  kk_chain_launch then runs the
// recorded bodies as the phases of ONE persistent launch (one batch item per XCD), or refuses.  Same arguments by construction, so
// the chained launch stores the bits the separate launches store.
#include <string.h>
constexpr int KK_CAPTURE_MAX = 16, KK_CAPTURE_ARG_BYTES = 1024;
struct KkCapturedLaunch {
    const char *kernel;                                         // the route record's name (kk_last_kernel form)
    unsigned grid[3], block;
    size_t lds, bytes;
    alignas(16) char args[KK_CAPTURE_ARG_BYTES];
};
struct KkLaunchCapture {
    int n, overflow;
    KkCapturedLaunch e[KK_CAPTURE_MAX];
};
KkLaunchCapture *kk_capture_target();                          // (kk_core.hip) non-null while this thread records
template <typename A> static inline bool kk_capture(const char *kernel, const A &a, dim3 grid, unsigned block, size_t lds) {
    static_assert(sizeof(A) <= KK_CAPTURE_ARG_BYTES && std::is_trivially_copyable<A>::value, "argument block too large to record");
    KkLaunchCapture *c = kk_capture_target();
    if (c == nullptr) return false;
    if (c->n >= KK_CAPTURE_MAX) { c->overflow = 1; return true; }
    KkCapturedLaunch &e = c->e[c->n++];
    e.kernel = kernel;
    e.grid[0] = grid.x; e.grid[1] = grid.y; e.grid[2] = grid.z;
    e.block = block; e.lds = lds; e.bytes = sizeof(A);
    memcpy(e.args, &a, sizeof(A));
    return true;
}

// A partial per-segment sum of squares (kk_seg_sumsq's records, kk_optim.hip): merged in array order by a fixed tree, so the order the
// records are WRITTEN in never matters.  The weight-gradient GEMMs write one per tile of a dW they have just finished (G16Args::ss_rec).
struct KkSegRec { double v; int32_t seg; int32_t pad; };       // seg < 0: empty slot

// Tuning switches.  The PRODUCT library reads no environment variable but KK_GEMM16_TUNE (one-time tile policy override, see
// kk_gemm16.hip): every A/B switch and every result-changing timing probe below exists only in a tools build
// (python -m kokoro_ruslan_amd.build --tuning, which defines KK_TUNING_HOOKS); in the product they fold to their defaults.
#include <stdlib.h>
#ifdef KK_TUNING_HOOKS
static inline int kk_tune_env(const char *name, int dflt) { const char *v = getenv(name); return v ? atoi(v) : dflt; }
#define KK_DBG(a, bits) (((a).dbg & (bits)) != 0)
#else
static inline int kk_tune_env(const char *, int dflt) { return dflt; }
#define KK_DBG(a, bits) (false)
#endif

#define KK_REQUIRE(cond, ...)                                   \
    do {                                                        \
        if (!(cond)) return kk_fail(KK_EINVAL, __VA_ARGS__);    \
    } while (0)

#define KK_LAUNCH_CHECK(name)                                                              \
    do {                                                                                   \
        hipError_t e__ = hipGetLastError();                                                \
        if (e__ != hipSuccess) return kk_fail((int)e__, "%s: %s", name, hipGetErrorString(e__)); \
    } while (0)

// bf16-storage GEMM core (kk_gemm16.hip); kk_gemm routes to it when both operands are bf16 and it is eligible
bool kk_gemm16_eligible(int ta, int tb, int64_t M, int64_t N, int64_t K, const void *A, int64_t lda, const void *B, int64_t ldb);
int kk_gemm16_launch(int ta, int tb, int64_t M, int64_t N, int64_t K, float alpha, const void *A, int64_t lda, const void *B,
                     int64_t ldb, float beta, void *C, int64_t ldc, int c_bf16, const float *bias, const float *residual,
                     int64_t ldr, int64_t res_mod, int split_k, int xcd_swizzle, hipStream_t s);
void kk_gemm16_tune(int thr128, int thr12864, int split_target);
int kk_gemm16_wgrad_group(const KkWgradDesc *d, int n, int split_k, int overwrite, int xcd_swizzle, hipStream_t s, void *ss_rec = nullptr,
                          const int32_t *ss_seg = nullptr, int32_t *ss_count = nullptr);
void kk_gemm16_tune_group(int split);
int kk_gemm16_qkv_headnorm(int64_t T, int parts, int heads, int64_t K, const void *x, int64_t ldx, const void *W, const float *bias,
                           void *raw, int64_t ldraw, void *y, int64_t ldy, int S, const float *const *gains, int rope_mask,
                           const float *cos_t, const float *sin_t, int xcd_swizzle, hipStream_t s);
// Zero `bytes` bytes at `p` (4-byte aligned, bytes % 4 == 0) with an ordinary kernel launch.  Used instead of
// hipMemsetAsync everywhere: inside a captured hipGraph that is launched again while its previous launch is still in
// flight, the runtime's memset node occasionally left the unaligned tail of the range (the last bytes % 32) holding
// garbage — measured on the 311-double gradient-norm vector (3 stale doubles, one NaN, ~12 skipped optimizer steps in
// 1 of 7 runs of 600 steps; tools/probes/skip_stress.py).
int kk_zero_async(void *p, size_t bytes, hipStream_t s);
int kk_gemm16_linear_glu(int64_t T, int64_t F, int64_t K, const void *x, int64_t ldx, const void *W, const float *bias, void *h1,
                         void *g, int64_t ldg, const uint32_t *seed, uint32_t site, float p, int xcd_swizzle, hipStream_t s);
// dX = dY.W of an attention output projection with Delta = rowsum_head(dX * O) as the epilogue (see kk_gemm16.hip)
bool kk_gemm16_dgrad_delta_supported(int64_t M, int64_t N, int64_t K);
int kk_gemm16_dgrad_delta(int64_t M, int64_t N, int64_t K, const void *dy, int64_t lddy, const void *W, int64_t ldw, void *dx,
                          int64_t lddx, const void *O, int64_t ldo, float *delta, int S, int heads, int xcd_swizzle, hipStream_t s);
// dX = dY.W fused with the GLU gate's backward (see kk_gemm16.hip)
int kk_gemm16_dgrad_glu(int64_t T, int64_t F, int64_t H, const void *dy, int64_t lddy, const void *W, const void *h1, void *dh1,
                        float *partials, const uint32_t *seed, uint32_t site, float p, int xcd_swizzle, hipStream_t s);

static inline int kk_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Cross-lane reductions by DPP (data-parallel primitives: the partner lane's value arrives as an operand modifier of a VALU instruction).
// __shfl_xor compiles to index arithmetic + ds_bpermute_b32 + s_waitcnt — a round trip through the LDS queue per step, ~150 clocks x 6
// dependent steps per wave reduction, and the row-per-wave kernels (sub-layer tails, LayerNorms, the encoder launch's tail phases) do
// two or three of those per row.  Here: quad_perm [1,0,3,2] and [2,3,0,1] (partners 1 and 2 away), row_half_mirror and row_mirror (at
// that point every lane of a quad / of eight lanes holds the same partial sum, so the mirrored lane is as good as the xor partner), then
// the four 16-lane rows' sums by v_readlane.  Every lane must be active.  (kk_dpp: v_mov_b32_dpp with bound_ctrl.)
#ifdef KK_OLD_SHFL      // (A/B flavour: the ds_bpermute forms)
template <int CTRL> __device__ __forceinline__ float kk_dpp(float v) {
    return __shfl_xor(v, CTRL == 0xB1 ? 1 : (CTRL == 0x4E ? 2 : (CTRL == 0x141 ? 4 : 8)), 64);
}
#else
template <int CTRL> __device__ __forceinline__ float kk_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
#endif
__device__ __forceinline__ float kk_row16_sum(float v) {       // sum over the lane's 16-lane row, in every lane of the row
    v += kk_dpp<0xB1>(v);
    v += kk_dpp<0x4E>(v);
    v += kk_dpp<0x141>(v);
    v += kk_dpp<0x140>(v);
    return v;
}
__device__ __forceinline__ float kk_row16_max(float v) {
    v = fmaxf(v, kk_dpp<0xB1>(v));
    v = fmaxf(v, kk_dpp<0x4E>(v));
    v = fmaxf(v, kk_dpp<0x141>(v));
    v = fmaxf(v, kk_dpp<0x140>(v));
    return v;
}
__device__ __forceinline__ float kk_readlane_f(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ float wave_sum(float v) {
#ifdef KK_OLD_SHFL
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
#else
    v = kk_row16_sum(v);
    return (kk_readlane_f(v, 0) + kk_readlane_f(v, 16)) + (kk_readlane_f(v, 32) + kk_readlane_f(v, 48));
#endif
}
template <int CTRL> __device__ __forceinline__ double kk_dpp_d(double v) {      // (two 32-bit DPP moves)
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, 0xF, 0xF, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, 0xF, 0xF, true);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double kk_readlane_d(double v, int lane) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, lane), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), lane);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double wave_sum_d(double v) {       // (as wave_sum: the block reductions of the loss / norm kernels do 11 of these)
#ifdef KK_OLD_SHFL
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
#else
    v += kk_dpp_d<0xB1>(v);
    v += kk_dpp_d<0x4E>(v);
    v += kk_dpp_d<0x141>(v);
    v += kk_dpp_d<0x140>(v);
    return (kk_readlane_d(v, 0) + kk_readlane_d(v, 16)) + (kk_readlane_d(v, 32) + kk_readlane_d(v, 48));
#endif
}
__device__ __forceinline__ float wave_max(float v) {
    v = kk_row16_max(v);
    return fmaxf(fmaxf(kk_readlane_f(v, 0), kk_readlane_f(v, 16)), fmaxf(kk_readlane_f(v, 32), kk_readlane_f(v, 48)));
}

// Block-wide sum for blockDim.x == 256 (4 waves). `red` is >= 4 floats of LDS. Result valid in every thread.
__device__ __forceinline__ float block_sum_256(float v, float *red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ double block_sum_256_d(double v, double *red) {
    v = wave_sum_d(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }

// Write-through (sc1) stores of a kernel's BIG outputs.  A kernel boundary writes the XCD L2s' dirty lines back before the next kernel
// may start (its readers sit on other XCDs): ~B / 6 TB/s behind B dirty bytes (MI355X_MICROARCH "boundary") — serial time on a chain of
// dependent launches.  A write-through store sends the bytes to memory while the kernel is still running.  Measured in-step, interleaved
// on one box (tools build, KK_WT_MAX_ROWS=0 against the default): 4.054 -> 3.994 ms (-1.5 %) at 8 x 512, 6.909 -> 6.837
// (-1.0 %) at 8 x 1024, 9.117 -> 9.112 (+-0) under dynamic batching at ~11 K rows per launch (long launches hide their write-back).
// Applied to the outputs of the sub-layer tails, every epilogue of the bf16 GEMM core (fp32 weight gradients through the same LDS
// transposition as the bf16 tiles, so that they too leave as 16-byte stores) and the second-generation attention kernels; NOT to the
// optimizer / norm / element-wise kernels (measured: no gain at 8 x 512, +0.9 % step time under dynamic batching).  Every entry point
// decides from its OWN shape (kk_write_through(rows of the launch)): no state, no switch in the product.
constexpr int KK_WT_MAX_ROWS = 16384;
static inline int kk_write_through(int64_t rows) {             // (tools build: KK_WT_MAX_ROWS=0 restores plain stores everywhere)
    static const int max_rows = kk_tune_env("KK_WT_MAX_ROWS", KK_WT_MAX_ROWS);
    return rows <= max_rows ? 1 : 0;
}
typedef unsigned int kk_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int kk_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void kk_st16_wt(void *p, kk_u32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void kk_st4_wt(void *p, unsigned v) {
    asm volatile("global_store_dword %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void kk_st8_wt(void *p, kk_u32x2 v) {
    asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
// 16- / 8-byte GLOBAL output stores; wt is wave-uniform (a kernel argument)
__device__ __forceinline__ void kk_store16(void *p, kk_u32x4 v, int wt) {
    if (wt) kk_st16_wt(p, v);
    else *reinterpret_cast<kk_u32x4 *>(p) = v;
}
__device__ __forceinline__ void kk_store8(void *p, kk_u32x2 v, int wt) {
    if (wt) kk_st8_wt(p, v);
    else *reinterpret_cast<kk_u32x2 *>(p) = v;
}
__device__ __forceinline__ void st4_out(float *p, float4 v, int wt) { kk_store16(p, __builtin_bit_cast(kk_u32x4, v), wt); }

// Storage-type generic 4-element access: activations are fp32 in the parity mode and bf16 in the bf16 mode.
template <typename T> __device__ __forceinline__ float4 ldv4(const T *p);
template <> __device__ __forceinline__ float4 ldv4<float>(const float *p) { return ld4(p); }
template <> __device__ __forceinline__ float4 ldv4<__bf16>(const __bf16 *p) {
    const bf16x4 v = *reinterpret_cast<const bf16x4 *>(p);
    return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}
template <typename T> __device__ __forceinline__ void stv4(T *p, float4 v);
template <> __device__ __forceinline__ void stv4<float>(float *p, float4 v) { st4(p, v); }
template <> __device__ __forceinline__ void stv4<__bf16>(__bf16 *p, float4 v) {
    bf16x4 o;
    o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;
    *reinterpret_cast<bf16x4 *>(p) = o;
}
template <typename T> __device__ __forceinline__ void stv4_out(T *p, float4 v, int wt);
template <> __device__ __forceinline__ void stv4_out<float>(float *p, float4 v, int wt) { st4_out(p, v, wt); }
template <> __device__ __forceinline__ void stv4_out<__bf16>(__bf16 *p, float4 v, int wt) {
    bf16x4 o;
    o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;
    kk_store8(p, __builtin_bit_cast(kk_u32x2, o), wt);
}
template <typename T> __device__ __forceinline__ float ldv1(const T *p) { return (float)(*p); }
template <typename T> __device__ __forceinline__ void stv1(T *p, float v) { *p = (T)v; }

// C/D fragment of a 32x32 MFMA tile: register r of lane l holds element
//   row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5),  col = l & 31.
__device__ __forceinline__ int frag_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// ---- counter-based RNG for dropout / DropPath / SpecAugment -------------------------------------------------------
// A mask bit is a pure function of (step seed read from device memory, call-site id, element index), so the backward
// kernels regenerate exactly the forward's masks and nothing mask-sized is ever stored.  The seed lives in HBM (not
// in a kernel argument) so that a captured hipGraph draws fresh masks on every replay.
__device__ __forceinline__ uint32_t kk_hash(uint32_t seed, uint32_t site, uint64_t idx) {
    uint32_t x = (uint32_t)idx ^ (seed * 0x9E3779B9u + site * 0x85EBCA6Bu + (uint32_t)(idx >> 32) * 0xC2B2AE35u);
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;      // "lowbias32" finaliser
    x ^= seed; x *= 0x9E3779B1u; x ^= x >> 15;
    return x;
}
__device__ __forceinline__ uint32_t kk_drop_threshold(float p) {       // drop iff hash < threshold
    return p <= 0.f ? 0u : (p >= 1.f ? 0xFFFFFFFFu : (uint32_t)((double)p * 4294967296.0));
}
// Element-wise dropout decisions: one hash serves the element PAIR (idx>>1) — 16-bit keep fields, the low one for the
// even element — and is two xorshift-multiply rounds with 24-bit multipliers (v_mul_u32_u24 is full rate; the 32-bit
// multiplies of kk_hash are quarter rate).  Statistically equivalent to lowbias32 on counters
// (tools/dropout_hash_quality.py).  p is quantised to 1/65536 (thr >> 16, rounded).
__device__ __forceinline__ uint32_t kk_pair_hash(uint32_t seed, uint32_t site, uint64_t idx) {
    uint32_t x = (uint32_t)(idx >> 1) ^ (seed * 0x9E3779B9u + site * 0x85EBCA6Bu);
    x ^= x >> 16; x = __umul24(x, 0xb5352du); x ^= x >> 13; x = __umul24(x, 0xca68b5u); x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t kk_thr16(uint32_t thr) { return thr >= 0xFFFF8000u ? 0xFFFFu : (thr + 0x8000u) >> 16; }
// multiplicative mask value: 0 when dropped, 1/(1-p) when kept (p == 0 -> always 1)
__device__ __forceinline__ float kk_drop_mul(uint32_t seed, uint32_t site, uint64_t idx, uint32_t thr, float inv_keep) {
    if (thr == 0u) return inv_keep;
    const uint32_t h = kk_pair_hash(seed, site, idx);
    return (((idx & 1) ? h >> 16 : h & 0xFFFFu) < kk_thr16(thr)) ? 0.f : inv_keep;
}
// four consecutive elements starting at a multiple of 4: two hashes
__device__ __forceinline__ void kk_drop_mul4(uint32_t seed, uint32_t site, uint64_t idx0, uint32_t thr, float inv_keep, float (&m)[4]) {
    m[0] = m[1] = m[2] = m[3] = inv_keep;
    if (thr == 0u) return;
    const uint32_t t = kk_thr16(thr), h0 = kk_pair_hash(seed, site, idx0), h1 = kk_pair_hash(seed, site, idx0 + 2);
    m[0] = (h0 & 0xFFFFu) < t ? 0.f : inv_keep;
    m[1] = (h0 >> 16) < t ? 0.f : inv_keep;
    m[2] = (h1 & 0xFFFFu) < t ? 0.f : inv_keep;
    m[3] = (h1 >> 16) < t ? 0.f : inv_keep;
}

// per-head RMSNorm(64) (+ RoPE) of one (row, head) vector held by 16 lanes, one float4 each (sub = lane & 15): shared by
// headnorm_rope_fwd_kernel and the q|k|v GEMM's epilogue so that the two give the same bits.
// rotate_half(n)[d] = -n[d+32] (d<32), n[d-32] (d>=32)   (positional_encoding.py:152-157)
__device__ __forceinline__ float kk_sum16(float v) { return kk_row16_sum(v); }      // (same additions as the xor butterfly: same bits)
__device__ __forceinline__ float4 kk_shfl8(const float4 &v) {       // the lane 8 away in the 16-lane row: row_ror:8
    return make_float4(kk_dpp<0x128>(v.x), kk_dpp<0x128>(v.y), kk_dpp<0x128>(v.z), kk_dpp<0x128>(v.w));
}
__device__ __forceinline__ float4 kk_headnorm_rope(const float4 &v, const float4 &g, bool rope, const float *cos_row,
                                                   const float *sin_row, int sub) {
    const float rs = 1.f / sqrtf(kk_sum16(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w) * (1.f / 64.f) + 1.1920928955078125e-7f);
    float4 n = make_float4(v.x * rs * g.x, v.y * rs * g.y, v.z * rs * g.z, v.w * rs * g.w);
    if (rope) {
        const float4 o = kk_shfl8(n), c = ld4(cos_row + sub * 4), sn = ld4(sin_row + sub * 4);
        const float sg = sub < 8 ? -1.f : 1.f;
        n = make_float4(n.x * c.x + sg * o.x * sn.x, n.y * c.y + sg * o.y * sn.y, n.z * c.z + sg * o.z * sn.z,
                        n.w * c.w + sg * o.w * sn.w);
    }
    return n;
}

// GELU and its derivative for the bf16 mode (results are rounded to bf16 anyway): erf by Abramowitz-Stegun 7.1.26
// (|error| <= 1.5e-7), whose exp(-z^2) with z = |x| / sqrt(2) IS the Gaussian exp(-x^2 / 2) the derivative needs — one
// v_exp_f32, one v_rcp_f32 and ~15 other VALU operations for both values, against ~130 for erff() + expf().  In the GLU
// epilogues of the GEMMs that arithmetic was a third of the kernel (the backward one: 2081 VALU instructions per wave
// after the last MFMA).  The fp32 parity mode keeps the exact forms below.
__device__ __forceinline__ void kk_gelu_pair_fast(float x, float &g, float &dg) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.f));
    const float e = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170368f);           // exp(-x^2 / 2)
    const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
    const float q = 0.5f * poly * e;                         // 0.5 erfc(|x| / sqrt 2)
    const float phi = x >= 0.f ? 1.f - q : q;                // Phi(x)
    g = x * phi;
    dg = fmaf(x * e, 0.39894228040143267794f, phi);
}
__device__ __forceinline__ float kk_gelu_fast(float x) {
    float g, dg;
    kk_gelu_pair_fast(x, g, dg);
    return g;
}
// (Round 6 measured a cheaper form — Phi(x) = 0.5 + x Q(x^2), Q of degree 8, no v_rcp / v_exp in the forward, |error| <= 1.4e-5 — because the GLU epilogues
//  spend 2 x 19 % of their SIMDs' time in the vector ALU: level in the step against THIS form built the same way (3.513 vs 3.517 ms at 8 x 512, 5.790 vs 5.790
//  at 8 x 1024, profiles/r06_gelu_poly_ab.txt), so the more accurate form stays.)
// exact-erf GELU (nn.GELU(), transformers.py:51) and its derivative
__device__ __forceinline__ float kk_gelu(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float kk_gelu_grad(float x) {
    return 0.5f * (1.f + erff(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * expf(-0.5f * x * x);
}
