// The whole text-encoder forward (all FFT blocks of model/model.py:375-388, transformers.py:452-490) as ONE persistent
// launch for gfx950.
//
// Why: at 8 x 64 phonemes the encoder works on 512 rows.  As 48 dependent launches (q|k|v, attention, w_o, tail, linear1,
// linear2, tail per layer) it is pure launch latency on the step's critical path — ~11 us per launch, 545 us per step
// with the chip idle (profiles/r02_step_timeline_timestamps_8x512.txt) — and nothing else in the forward can run beside it.
//
// How: a batch item never exchanges data with another one inside the encoder, so the work is partitioned by ITEM, not by
// tile: workgroups b, b+8, b+16, ... (the ones the dispatcher places on XCD b, private L2 and all) form a group of
// gridDim/8 members that carries item b (then b+8, ...) through every layer.  Inside a group a phase is split into
// units (32 rows x 32/64 output columns of a GEMM; a head of the attention; a row of a sub-layer tail); the members
// meet at a GROUP barrier between phases — one counter per group, never a grid-wide barrier — and hand activations
// over through HBM/L2 with write-through (sc1) stores and L1-bypassing (sc1) loads, the placement-independent form
// (correct wherever the workgroups land; same-XCD placement only makes it faster).  Everything the backward reads
// (raw / normalised q|k|v, context, log-sum-exp, LayerNorm outputs and statistics, h1, gate, f2, RMS statistics, the
// residual streams) is written exactly where the per-kernel path writes it, so the backward is unchanged, and the
// dropout / DropPath masks are the same functions of (seed, site, element).
//
// GEMM units at this size are latency-, not MFMA-bound: a unit (32 rows x 32/64 columns, the whole K) belongs to ONE
// wave, which streams its operand rows straight into MFMA fragments (contiguous per lane, no LDS staging, no workgroup
// barrier, the next chunk's loads in flight under the current chunk's MFMAs) and runs the epilogue (head norm + RoPE,
// bias, GLU gate + dropout) on its own row-major LDS tile; units are dealt workgroup-first, so a phase is one pass.
#include "kk_common.h"
#include <algorithm>
#include <float.h>
#include <math.h>

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int SC1 = 16;                      // cache-policy bit of the buffer builtins: sc1 (agent scope, write-through / L1 bypass)
constexpr int NTHREADS = 256, NWAVES = 4;
constexpr int KP = 72;                       // bf16 per row of the K tile in LDS (144-byte rows)
constexpr int SMAX = 128;                    // rows (phonemes) per item a group can carry

typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t mk(const void *p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, 0x7FFFFFFF, 0x00020000); }
// data produced INSIDE the launch by another workgroup: sc1 on both sides (no fences needed, any placement)
__device__ __forceinline__ u32x4 ld16(rsrc_t r, uint32_t off) { return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, SC1); }
__device__ __forceinline__ u32x2 ld8(rsrc_t r, uint32_t off) { return __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, SC1); }
__device__ __forceinline__ float ldf(rsrc_t r, uint32_t off) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, SC1)); }
__device__ __forceinline__ void st16(rsrc_t r, uint32_t off, u32x4 v) { __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, SC1); }
__device__ __forceinline__ void st8(rsrc_t r, uint32_t off, u32x2 v) { __builtin_amdgcn_raw_buffer_store_b64(v, r, off, 0, SC1); }
__device__ __forceinline__ void stf(rsrc_t r, uint32_t off, float v) { __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, off, 0, SC1); }
__device__ __forceinline__ float4 ldf4(rsrc_t r, uint32_t off) { return __builtin_bit_cast(float4, ld16(r, off)); }
__device__ __forceinline__ void stf4(rsrc_t r, uint32_t off, float4 v) { st16(r, off, __builtin_bit_cast(u32x4, v)); }
__device__ __forceinline__ u32x2 pack4(float4 v) {
    bf16x4 o;
    o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;
    return __builtin_bit_cast(u32x2, o);
}
__device__ __forceinline__ float4 unpack4(u32x2 u) {
    const bf16x4 v = __builtin_bit_cast(bf16x4, u);
    return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}

// ---- group barrier ---------------------------------------------------------------------------------------------------
// sync words (uint32): [0] = error flag (a spin ran out), [32 + 32 g] = arrivals of group g, [32 + 32 g + 16] = exits.
// The counters are zero between launches: the last member to leave a launch resets them (every member has passed every
// barrier by then), so a replayed hipGraph needs no memset node in front of the launch.
struct GroupSync {
    unsigned *arrive_ctr, *leave, *err;
    unsigned target, members;
    bool dead;
    __device__ __forceinline__ void init(unsigned *words, int group, int nmembers) {
        err = words;
        arrive_ctr = words + 32 + 32 * group;
        leave = arrive_ctr + 16;
        target = 0;
        members = (unsigned)nmembers;
        dead = false;
    }
    // Split barrier: arrive() publishes this workgroup's stores (every store issued before it is visible to every member
    // after its wait()); between the two a member does work that depends on nobody — it starts the DMA of the weights of
    // its next GEMM phase there, so that the issue time of up to 96 KB hides behind the arrival of the slowest member.
    __device__ __forceinline__ void arrive() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // each wave: its write-through stores have been acknowledged
        __syncthreads();
        target += members;
        if (threadIdx.x == 0 && !dead) __hip_atomic_fetch_add(arrive_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __device__ __forceinline__ void wait() {
        if (threadIdx.x == 0 && !dead) {
            unsigned spins = 0;
            while (__hip_atomic_load(arrive_ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 22)) {                        // seconds: a member is not resident / died — give up loudly
                    __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    dead = true;
                    break;
                }
                if ((spins & 1023u) == 0u && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                    dead = true;
                    break;
                }
            }
        }
        __syncthreads();
    }
    __device__ __forceinline__ void barrier() { arrive(); wait(); }
    __device__ __forceinline__ void exit() {
        if (threadIdx.x == 0) {
            const unsigned old = __hip_atomic_fetch_add(leave, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old == members - 1u) {
                __hip_atomic_store(arrive_ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(leave, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
};

// ---- GEMM phases: operands staged into LDS by the buffer-load-to-LDS DMA, 16x16x32 MFMA tiles ----------------------------
// A member's share of a phase is [rows of the item] x [its slice of the output columns]:  X (the item's activations,
// written by the group a phase earlier) and the member's weight rows are copied into LDS with fully coalesced 16-byte
// DMA reads — a wave instruction moves 1 KB that lies contiguously in memory AND in LDS — all of them in flight at
// once (no registers involved), one wait, one workgroup barrier, then MFMAs out of LDS.  (A first version fed the MFMA
// fragments straight from global memory, one row per lane: 64 distinct lines per load instruction, L1-bypassing for
// X — request-rate bound, 12-15 us per phase where the bytes need 1-2.)  The 16-byte chunk c of LDS row r holds the
// operand's chunk c ^ (r & 15): the fragment reads (ds_read_b128, 16 rows x the same k) are bank-conflict free.
// WEIGHTS DO NOT DEPEND ON THE PREVIOUS PHASE: a member starts the DMA of its next GEMM phase's weight rows inside the
// group barrier that ends the current one (between arriving and waiting), so that they land under the barrier and
// whatever non-GEMM phase lies in between; after a barrier only X is waited for.
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define LDS_PTR(p) ((__attribute__((address_space(3))) void *)(p))
constexpr int XS_BYTES = 65536;              // X panel: 64 rows x K = 512 (or 32 rows x K = 1536 reaching into WS)
constexpr int WS_BYTES = 98304;              // weight rows of the member: up to 96 x K = 512 / 16 x K = 1536 (at W2_OFF)
constexpr int W2_OFF = XS_BYTES + 49152;     // linear2's weight rows sit in the upper half of WS (its X panel needs 96 KB)
constexpr int TPW = 100;                     // floats per row of a wave's 16-row epilogue tile (96 columns + pad)
constexpr int TILE_BYTES = 16 * TPW * 4;

// rows [row0, row0 + rows) of a [*, K] bf16 matrix (pitch ld elements) -> LDS at dst (pitch K, swizzled); rows >= valid repeat
// row valid-1.  rows * K / 8 must be a multiple of 256 chunks.  INTRA: written earlier in this launch (sc1).  KT: K at
// compile time (0: run time) — the address arithmetic of an instruction must be a handful of operations, or the issue
// loop, not the memory system, sets the pace (a division per instruction made a 64 KB panel take 2 us to ISSUE).
template <bool INTRA, int KT>
__device__ __forceinline__ void stage_rows(char *dst, const void *src, int64_t ld, int64_t row0, int valid, int rows, int K) {
    const rsrc_t R = mk(src);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cpr = KT ? KT >> 3 : K >> 3, total = rows * cpr;
    const uint32_t base = (uint32_t)(row0 * ld * 2), pitch = (uint32_t)(ld * 2);
    if (KT == 512) {                                           // one row per instruction: the row index is wave-uniform
#pragma unroll 4
        for (int r = wave; r < rows; r += NWAVES) {
            const int rr = r < valid ? r : valid - 1;
            const uint32_t voff = base + (uint32_t)rr * pitch + (uint32_t)((lane ^ (r & 15)) << 4);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(R, LDS_PTR(dst + (size_t)r * 1024), 16, voff, 0, 0, INTRA ? SC1 : 0);
        }
        return;
    }
#pragma unroll 4
    for (int q0 = wave * 64; q0 < total; q0 += NTHREADS) {
        const int q = q0 + lane, r = q / cpr, ch = q - r * cpr;
        const int rr = r < valid ? r : valid - 1;
        const uint32_t voff = base + (uint32_t)rr * pitch + (uint32_t)((ch ^ (r & 15)) << 4);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(R, LDS_PTR(dst + (size_t)q0 * 16), 16, voff, 0, 0, INTRA ? SC1 : 0);
    }
}
__device__ __forceinline__ void dma_wait() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}
// acc[j] += X[xrow0 + (lane & 15)][k] * W[wrow0 + 16 j + (lane & 15)][k] for the k-steps [ks0, ks1) (32 k each), out of LDS.
// KT > 0: K and the step range are compile-time constants and all NCB blocks are live — the loop unrolls and the LDS reads
// of later steps run ahead of the MFMAs (with run-time bounds every MFMA waited for its own ds_read: ~140 cycles each).
template <int NCB, int KT, int KS0, int KS1>
__device__ __forceinline__ void wave_mma(f32x4 (&acc)[NCB], const char *XS, int xrow0, const char *WS, int wrow0, int ncb, int K, int ks0, int ks1) {
    const int lane = threadIdx.x & 63, l15 = lane & 15, kq = lane >> 4;
    const int Kc = KT ? KT : K;
    const char *xa = XS + (size_t)(xrow0 + l15) * Kc * 2, *wa = WS + (size_t)(wrow0 + l15) * Kc * 2;
    const size_t blk = (size_t)16 * Kc * 2;
    if constexpr (KT > 0) {
#pragma unroll
        for (int ks = KS0; ks < KS1; ++ks) {
            const int off = (((ks << 2) + kq) ^ l15) << 4;
            const bf16x8 a = *reinterpret_cast<const bf16x8 *>(xa + off);
            bf16x8 b[NCB];
#pragma unroll
            for (int j = 0; j < NCB; ++j) b[j] = *reinterpret_cast<const bf16x8 *>(wa + j * blk + off);
#pragma unroll
            for (int j = 0; j < NCB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b[j], acc[j], 0, 0, 0);
        }
    } else {
        for (int ks = ks0; ks < ks1; ++ks) {
            const int off = (((ks << 2) + kq) ^ l15) << 4;
            const bf16x8 a = *reinterpret_cast<const bf16x8 *>(xa + off);
#pragma unroll
            for (int j = 0; j < NCB; ++j) {
                if (j < ncb) {
                    const bf16x8 b = *reinterpret_cast<const bf16x8 *>(wa + j * blk + off);
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
                }
            }
        }
    }
}
// accumulators -> this wave's row-major LDS tile [16][TPW] (C layout: col = lane & 15, row = 4 (lane >> 4) + r)
template <int NCB>
__device__ __forceinline__ void acc_to_tile(float *tile, const f32x4 (&acc)[NCB], int ncb) {
    const int lane = threadIdx.x & 63, l15 = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int j = 0; j < NCB; ++j)
        if (j < ncb)
#pragma unroll
            for (int r = 0; r < 4; ++r) tile[(4 * kq + r) * TPW + 16 * j + l15] = acc[j][r];
    __builtin_amdgcn_wave_barrier();                           // (one wave: its LDS operations complete in order)
}

struct Ctx {
    const KkEncStack *a;
    int b, S, H, F, heads, RB;     // item, rows per item, dims, 32-row blocks per item
    int member, members;
    uint32_t seed;
    char *smem;                    // LDS: X panel | weight rows
    mutable uint64_t *sub;         // tools: finer clock stamps of the traced workgroup (trace + 2048), else null
    __device__ __forceinline__ void substamp() const { if (sub) *sub++ = wall_clock64(); }
    __device__ __forceinline__ float *tile() const { return reinterpret_cast<float *>(smem + (threadIdx.x >> 6) * TILE_BYTES); }
    __device__ __forceinline__ char *ws() const { return smem + XS_BYTES; }
};

// ---- phase 1: q|k|v projection + per-head RMSNorm (+ RoPE on q, k) ------------------------------------------------------
// unit = (part, head, row block): 32 rows x 64 columns.  transformers.py:131-136 (no bias), :260-277.
// weight rows of a member for each GEMM phase -> LDS (see the LDS map above); called a phase ahead
enum { G_QKV = 0, G_WO = 1, G_LIN1 = 2, G_LIN2 = 3 };
__device__ __forceinline__ int lin1_blocks(const Ctx &c, int &first) {     // 16-column blocks of the gate a member owns
    const int nb = c.F / 16, per = (nb + c.members - 1) / c.members;
    first = c.member * per;
    return max(0, min(per, nb - first));
}
template <int HT, int FT>
__device__ __forceinline__ void prefetch_weights(const Ctx &c, const KkEncLayer &L, int kind) {
    const int H = HT ? HT : c.H, F = FT ? FT : c.F;
    if (kind == G_QKV) {
        if (c.member < 3 * c.heads) stage_rows<false, HT>(c.ws(), L.w_qkv, H, (int64_t)c.member * 64, 64, 64, H);
    } else if (kind == G_WO) {
        if (c.member < H / 16) stage_rows<false, HT>(c.ws(), L.w_o, H, (int64_t)c.member * 16, 16, 16, H);
    } else if (kind == G_LIN1) {
        int first;
        const int cnt = lin1_blocks(c, first);
        if (cnt > 0) {
            stage_rows<false, HT>(c.ws(), L.w1, H, (int64_t)first * 16, cnt * 16, cnt * 16, H);
            stage_rows<false, HT>(c.ws() + (size_t)cnt * 16 * H * 2, L.w1, H, (int64_t)F + first * 16, cnt * 16, cnt * 16, H);
        }
    } else {
        if (c.member < H / 16) stage_rows<false, FT>(c.smem + W2_OFF, L.w2, F, (int64_t)c.member * 16, 16, 16, F);
    }
}

template <int HT, int FT>
__device__ __forceinline__ void phase_qkv(const Ctx &c, const KkEncLayer &L) {
    const int H = HT ? HT : c.H, S = c.S, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool active = c.member < 3 * c.heads;
    const rsrc_t RAW = mk(L.qkv_raw), NRM = mk(L.qkv_n);
    const int col0 = c.member * 64, prt = c.member / c.heads;
    for (int r0 = 0; r0 < S; r0 += 64) {                          // 64-row passes
        f32x4 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (active) stage_rows<true, HT>(c.smem, L.y1, H, (int64_t)c.b * S + r0, S - r0, 64, H);
        c.substamp();
        // (the epilogue's table rows are touched HERE, under the panel's DMA: read first in the epilogue they were a dependent L2 round
        //  trip per phase on the launch's critical chain)
        const int sub = lane & 15;
        const float *gain = prt == 0 ? L.g_q : (prt == 1 ? L.g_k : L.g_v);
        const float4 g = ld4(gain + sub * 4);
        float tw = 0.f;
        if (active && prt < 2) {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int r = r0 + wave * 16 + it * 4 + (lane >> 4), pos = r < S ? r : S - 1;
                tw += c.a->cos_t[pos * 64 + sub * 4] + c.a->sin_t[pos * 64 + sub * 4];      // warms the lines kk_headnorm_rope reads below
            }
        }
        asm volatile("" :: "v"(tw));
        dma_wait();
        c.substamp();
        const bool work = active && r0 + wave * 16 < S;
        if (work) wave_mma<4, HT, 0, HT / 32>(acc, c.smem, wave * 16, c.ws(), 0, 4, H, 0, H / 32);
        __syncthreads();                                           // X panel (and, after the last pass, the weights) are free
        c.substamp();
        if (work) {
            float *tile = c.tile();
            acc_to_tile<4>(tile, acc, 4);
            const bool rope = prt < 2;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int rl = it * 4 + (lane >> 4), r = r0 + wave * 16 + rl;
                const u32x2 raw = pack4(ld4(tile + rl * TPW + sub * 4));   // the backward (and the norm below) see the bf16 value
                const int pos = r < S ? r : S - 1;
                const float4 nn = kk_headnorm_rope(unpack4(raw), g, rope, c.a->cos_t + pos * 64, c.a->sin_t + pos * 64, sub);
                if (r < S) {
                    const uint32_t off = (uint32_t)((((int64_t)c.b * S + r) * 3 * H + col0 + sub * 4) * 2);
                    st8(RAW, off, raw);
                    st8(NRM, off, pack4(nn));
                }
            }
        }
        if (r0 + 64 < S) __syncthreads();                          // the tiles live in the X panel
    }
}

// ---- phase 2: attention of one head (all S <= 128 keys in one pass) ----------------------------------------------------
// Same orientation and dropout function as kk_attn.hip (attn_fwd_kernel / ProbDrop): S^T = K.Q^T with a lane owning one
// query, so the softmax is in-lane + one xor-32 exchange and P feeds the second MFMA as it lies in the accumulators.
struct ProbDropE {
    uint32_t thr, key, sk2;
    float inv_keep;
    __device__ __forceinline__ void init(uint32_t seed, uint32_t site, float p, int bh, int Sk) {
        thr = 0u;
        if (p > 0.f) {
            thr = (uint32_t)(p * 65536.f + 0.5f);
            thr = thr > 65535u ? 65535u : thr;
        }
        key = thr ? kk_hash(seed, site, (uint64_t)bh) : 0u;
        inv_keep = thr ? 65536.f / (float)(65536u - thr) : 1.f;
        sk2 = (uint32_t)(Sk + 1) >> 1;
    }
    __device__ __forceinline__ uint32_t hash(uint32_t x) const {
        x ^= key;
        x ^= x >> 16; x = __umul24(x, 0xb5352du); x ^= x >> 13; x = __umul24(x, 0xca68b5u); x ^= x >> 16;
        return x;
    }
};

template <int NSUB>     // 32-key sub-tiles (S <= 32 NSUB)
__device__ __forceinline__ void attn_head(const Ctx &c, const KkEncLayer &L, int hh, __bf16 *Ks, __bf16 *Vt) {
    constexpr int VP = NSUB * 32 + 8;                           // bf16 per row of the transposed V tile
    const int S = c.S, H = c.H, b = c.b;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
    const rsrc_t N = mk(L.qkv_n);
    const int64_t row_b = (int64_t)b * S;
    // this wave's query fragments first: their round trip overlaps the staging of K and V below
    const int q = wave * 32 + l31;
    const bool qvalid = q < S;
    const int qr = qvalid ? q : S - 1;
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
        qf[ks] = __builtin_bit_cast(bf16x8, ld16(N, (uint32_t)(((row_b + qr) * 3 * H + hh * 64 + ks * 16 + half * 8) * 2)));
    // K rows -> Ks[key][d], V rows -> Vt[d][key]; keys >= S are zero
    for (int p = threadIdx.x; p < NSUB * 32 * 4; p += NTHREADS) {
        const int key = p >> 2, seg = (p & 3) * 16;
        u32x4 v0 = {0u, 0u, 0u, 0u}, v1 = v0;
        if (key < S) {
            const uint32_t off = (uint32_t)(((row_b + key) * 3 * H + H + hh * 64 + seg) * 2);
            v0 = ld16(N, off);
            v1 = ld16(N, off + 16);
        }
        *reinterpret_cast<u32x4 *>(Ks + key * KP + seg) = v0;
        *reinterpret_cast<u32x4 *>(Ks + key * KP + seg + 8) = v1;
    }
    for (int p = threadIdx.x; p < NSUB * 8 * 16; p += NTHREADS) {
        const int rg = (p % (NSUB * 8)) * 4, dg = (p / (NSUB * 8)) * 4;
        u32x2 r[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            r[e] = u32x2{0u, 0u};
            if (rg + e < S) r[e] = ld8(N, (uint32_t)(((row_b + rg + e) * 3 * H + 2 * H + hh * 64 + dg) * 2));
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {                              // element e of rows 0..3 -> 4 contiguous keys of V^T row dg + e
            const int w = e >> 1, sh = 16 * (e & 1);
            u32x2 v;
            v[0] = ((r[0][w] >> sh) & 0xFFFFu) | (((r[1][w] >> sh) & 0xFFFFu) << 16);
            v[1] = ((r[2][w] >> sh) & 0xFFFFu) | (((r[3][w] >> sh) & 0xFFFFu) << 16);
            *reinterpret_cast<u32x2 *>(Vt + (dg + e) * VP + rg) = v;
        }
    }
    __syncthreads();
    if (wave * 32 < S) {
        const uint8_t *km = c.a->key_mask ? c.a->key_mask + (int64_t)b * S : nullptr;
        uint64_t kmb[(NSUB + 1) / 2];                               // bit j of word t: key 64 t + j is padding (wave-uniform)
#pragma unroll
        for (int t = 0; t < (NSUB + 1) / 2; ++t) {
            const int key = t * 64 + lane;
            kmb[t] = __ballot(km != nullptr && key < S && km[key < S ? key : 0] != 0);
        }
        const float c2 = 0.125f * 1.4426950408889634f;
        ProbDropE pd;
        pd.init(c.seed, L.site + 3, L.p, b * c.heads + hh, S);
        float p[NSUB][16];
        float mx = -INFINITY;
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) {
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
            if (sub * 32 < S) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const bf16x8 kf = *reinterpret_cast<const bf16x8 *>(Ks + (sub * 32 + l31) * KP + ks * 16 + half * 8);
                    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s, 0, 0, 0);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = sub * 32 + frag_row(r, half);
                const bool ok = key < S && !((kmb[sub >> 1] >> (key & 63)) & 1ull);
                p[sub][r] = ok ? s[r] : -INFINITY;
                mx = fmaxf(mx, p[sub][r]);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * c2;
        const float m = fmaxf(-1e30f, mx);
        float l = 0.f;
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub)
#pragma unroll
            for (int r = 0; r < 16; ++r) { p[sub][r] = __builtin_amdgcn_exp2f(fmaf(p[sub][r], c2, -m)); l += p[sub][r]; }
        l += __shfl_xor(l, 32, 64);
        f32x16 o[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) {
            if (sub * 32 >= S) continue;
            if (pd.thr) {                                          // softmax first, dropout after; 1/(1-p) at the store
                const uint32_t xb = (uint32_t)q * pd.sk2 + ((uint32_t)(sub * 32 + 4 * half) >> 1);
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const uint32_t hsh = pd.hash(xb + (uint32_t)(frag_row(r, 0) >> 1));
                    p[sub][r] = (hsh & 0xFFFFu) >= pd.thr ? p[sub][r] : 0.f;
                    p[sub][r + 1] = (hsh >> 16) >= pd.thr ? p[sub][r + 1] : 0.f;
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {                       // O^T[d][q] += V^T[d][key] P^T[key][q], keys in accumulator order
                bf16x8 pb;
#pragma unroll
                for (int j = 0; j < 8; ++j) pb[j] = (__bf16)p[sub][8 * s2 + j];
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const __bf16 *base = Vt + (db * 32 + l31) * VP + sub * 32 + 16 * s2 + 4 * half;
                    const bf16x4 lo = *reinterpret_cast<const bf16x4 *>(base), hi = *reinterpret_cast<const bf16x4 *>(base + 8);
                    bf16x8 va;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { va[e] = lo[e]; va[4 + e] = hi[e]; }
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, pb, o[db], 0, 0, 0);
                }
            }
        }
        if (qvalid) {
            const float inv = l > 0.f ? pd.inv_keep / l : 0.f;
            const rsrc_t C = mk(L.ctx);
            const uint32_t off = (uint32_t)(((row_b + q) * H + hh * 64) * 2);
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    st8(C, off + (uint32_t)(db * 32 + 8 * g + 4 * half) * 2,
                        pack4(make_float4(o[db][4 * g] * inv, o[db][4 * g + 1] * inv, o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv)));
            if (half == 0)
                L.lse[((int64_t)b * c.heads + hh) * S + q] = l > 0.f ? (m + __builtin_amdgcn_logf(l)) * 0.6931471805599453f : INFINITY;
        }
    }
    __syncthreads();
}

__device__ __forceinline__ void phase_attn(const Ctx &c, const KkEncLayer &L) {
    __bf16 *Ks = reinterpret_cast<__bf16 *>(c.smem);
    for (int hh = c.member; hh < c.heads; hh += c.members) {
        if (c.S <= 64) attn_head<2>(c, L, hh, Ks, Ks + 64 * KP);
        else attn_head<4>(c, L, hh, Ks, Ks + 128 * KP);
    }
}

// ---- phase 3: output projection (+ bias) -> fp32 ------------------------------------------------------------------------
template <int HT, int FT>
__device__ __forceinline__ void phase_wo(const Ctx &c, const KkEncLayer &L) {
    const int H = HT ? HT : c.H, S = c.S, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool active = c.member < H / 16;
    const rsrc_t Y = mk(L.proj);
    for (int r0 = 0; r0 < S; r0 += 64) {
        f32x4 acc[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
        if (active) stage_rows<true, HT>(c.smem, L.ctx, H, (int64_t)c.b * S + r0, S - r0, 64, H);
        c.substamp();
        const float4 bv_pre = active ? ld4(L.b_o + c.member * 16 + (lane & 3) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);      // (under the DMA)
        dma_wait();
        c.substamp();
        const bool work = active && r0 + wave * 16 < S;
        if (work) wave_mma<1, HT, 0, HT / 32>(acc, c.smem, wave * 16, c.ws(), 0, 1, H, 0, H / 32);
        __syncthreads();
        c.substamp();
        if (work) {
            float *tile = c.tile();
            acc_to_tile<1>(tile, acc, 1);
            const int rl = lane >> 2, cc = (lane & 3) * 4, r = r0 + wave * 16 + rl, col = c.member * 16 + cc;
            if (r < S) {
                float4 s = ld4(tile + rl * TPW + cc);
                const float4 bv = bv_pre;
                s.x += bv.x; s.y += bv.y; s.z += bv.z; s.w += bv.w;
                stf4(Y, (uint32_t)((((int64_t)c.b * S + r) * H + col) * 4), s);
            }
        }
        if (r0 + 64 < S) __syncthreads();
    }
}

// ---- phase 5: linear1 + GLU gate (+ dropout): h1 = [a | b] saved, g = gelu(a) * b * mask -----------------------------
template <int HT, int FT>
__device__ __forceinline__ void phase_lin1(const Ctx &c, const KkEncLayer &L) {
    const int H = HT ? HT : c.H, S = c.S, F = FT ? FT : c.F, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int first;
    const int cnt = lin1_blocks(c, first);                        // <= 3 blocks of 16 gate columns: a-rows then b-rows in LDS
    const bool active = cnt > 0;
    const rsrc_t H1 = mk(L.h1), G = mk(L.g);
    const uint32_t thr = kk_drop_threshold(L.p);
    const float ik = thr ? 1.f / (1.f - L.p) : 1.f;
    for (int r0 = 0; r0 < S; r0 += 64) {
        f32x4 acc[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (active) stage_rows<true, HT>(c.smem, L.y2, H, (int64_t)c.b * S + r0, S - r0, 64, H);
        c.substamp();
        float4 ba_pre[3], bb_pre[3];                                 // (under the DMA)
#pragma unroll
        for (int jj = 0; jj < 3; ++jj) {
            const int col = (first + (jj < cnt ? jj : 0)) * 16 + (lane & 3) * 4;
            ba_pre[jj] = active ? ld4(L.b1 + col) : make_float4(0.f, 0.f, 0.f, 0.f);
            bb_pre[jj] = active ? ld4(L.b1 + F + col) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        dma_wait();
        c.substamp();
        const bool work = active && r0 + wave * 16 < S;
        if (work) {
            if (HT > 0 && cnt == 3) wave_mma<6, HT, 0, HT / 32>(acc, c.smem, wave * 16, c.ws(), 0, 6, H, 0, H / 32);
            else wave_mma<6, 0, 0, 0>(acc, c.smem, wave * 16, c.ws(), 0, 2 * cnt, H, 0, H / 32);
        }
        __syncthreads();
        c.substamp();
        if (work) {
            float *tile = c.tile();
            acc_to_tile<6>(tile, acc, 2 * cnt);
            const int rl = lane >> 2, cc = (lane & 3) * 4, r = r0 + wave * 16 + rl;
            if (r < S) {
                const int64_t row = (int64_t)c.b * S + r;
#pragma unroll
                for (int jj = 0; jj < 3; ++jj) {
                    if (jj >= cnt) break;
                    const int col = (first + jj) * 16 + cc;
                    float4 av = ld4(tile + rl * TPW + 16 * jj + cc), bv = ld4(tile + rl * TPW + 16 * (cnt + jj) + cc);
                    const float4 ba = ba_pre[jj], bb = bb_pre[jj];
                    av.x += ba.x; av.y += ba.y; av.z += ba.z; av.w += ba.w;
                    bv.x += bb.x; bv.y += bb.y; bv.z += bb.z; bv.w += bb.w;
                    const u32x2 a16 = pack4(av), b16 = pack4(bv);    // what the backward will read
                    st8(H1, (uint32_t)((row * 2 * F + col) * 2), a16);
                    st8(H1, (uint32_t)((row * 2 * F + F + col) * 2), b16);
                    const float4 ar = unpack4(a16), br = unpack4(b16);
                    float mk4[4];
                    kk_drop_mul4(c.seed, L.site + 12, (uint64_t)row * F + col, thr, ik, mk4);
                    st8(G, (uint32_t)((row * F + col) * 2),
                        pack4(make_float4(kk_gelu_fast(ar.x) * br.x * mk4[0], kk_gelu_fast(ar.y) * br.y * mk4[1], kk_gelu_fast(ar.z) * br.z * mk4[2],
                                          kk_gelu_fast(ar.w) * br.w * mk4[3])));
                }
            }
        }
        if (r0 + 64 < S) __syncthreads();
    }
}

// ---- phase 6: linear2 (+ bias) -> bf16 -----------------------------------------------------------------------------------
// K = F does not fit beside its weights as a 64-row panel: 32-row passes (96 KB at F = 1536), the four waves = 2 row blocks x
// 2 halves of K, the halves summed through LDS.
template <int HT, int FT>
__device__ __forceinline__ void phase_lin2(const Ctx &c, const KkEncLayer &L) {
    const int H = HT ? HT : c.H, S = c.S, F = FT ? FT : c.F, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool active = c.member < H / 16;
    const rsrc_t Y = mk(L.f2);
    const int rbw = wave & 1, kh = wave >> 1, steps = F / 32;
    for (int r0 = 0; r0 < S; r0 += 32) {
        f32x4 acc[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
        if (active) stage_rows<true, FT>(c.smem, L.g, F, (int64_t)c.b * S + r0, S - r0, 32, F);
        c.substamp();
        const float4 b2_pre = active ? ld4(L.b2 + c.member * 16 + (lane & 3) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);      // (under the DMA)
        dma_wait();
        c.substamp();
        const bool work = active && r0 + rbw * 16 < S;
        if (work) {
            if (FT > 0 && kh == 0) wave_mma<1, FT, 0, FT / 64>(acc, c.smem, rbw * 16, c.smem + W2_OFF, 0, 1, F, 0, 0);
            else if (FT > 0) wave_mma<1, FT, FT / 64, FT / 32>(acc, c.smem, rbw * 16, c.smem + W2_OFF, 0, 1, F, 0, 0);
            else wave_mma<1, 0, 0, 0>(acc, c.smem, rbw * 16, c.smem + W2_OFF, 0, 1, F, kh * (steps / 2), (kh + 1) * (steps / 2));
        }
        __syncthreads();
        c.substamp();
        float *tile = c.tile();
        if (work) acc_to_tile<1>(tile, acc, 1);
        __syncthreads();
        if (work && kh == 0) {
            const float *other = reinterpret_cast<const float *>(c.smem + (wave + 2) * TILE_BYTES);
            const int rl = lane >> 2, cc = (lane & 3) * 4, r = r0 + rbw * 16 + rl, col = c.member * 16 + cc;
            if (r < S) {
                float4 s = ld4(tile + rl * TPW + cc);
                const float4 t = ld4(other + rl * TPW + cc), bv = b2_pre;
                s.x += t.x + bv.x; s.y += t.y + bv.y; s.z += t.z + bv.z; s.w += t.w + bv.w;
                st8(Y, (uint32_t)((((int64_t)c.b * S + r) * H + col) * 2), pack4(s));
            }
        }
        if (r0 + 32 < S) __syncthreads();
    }
}

// ---- phases 4 and 7: sub-layer tail, one wave per row (the arithmetic of sublayer_out_fwd_kernel, kk_dropout.hip) ------
//   x_out = res + masks * [RMSNorm](y);  n = LayerNorm(x_out)
template <bool FFN, typename TN, int NV>
__device__ __forceinline__ void tail_row(const Ctx &c, const KkEncLayer &L, int64_t row) {
    const int lane = threadIdx.x & 63, H = c.H;
    const rsrc_t Yr = mk(FFN ? (const void *)L.f2 : (const void *)L.proj), R = mk(FFN ? (const void *)L.xm : (const void *)L.x_in);
    const rsrc_t XO = mk(FFN ? L.xo : L.xm), NO = mk(FFN ? L.next_y : L.y2);
    const float *gain = FFN ? L.ffn_gain : nullptr;
    const float *lng = FFN ? L.next_g : L.ln2_g, *lnb = FFN ? L.next_b : L.ln2_b;
    const uint32_t site = L.site + (FFN ? 8u : 0u);
    const float p1 = L.p, p2 = FFN ? L.p : 0.f;
    const uint32_t t1 = kk_drop_threshold(p1), t2 = kk_drop_threshold(p2);
    const float k1 = p1 > 0.f ? 1.f / (1.f - p1) : 1.f, k2 = p2 > 0.f ? 1.f / (1.f - p2) : 1.f;
    float4 v[NV], rres[NV], gg[NV], lg[NV], lb[NV];
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int cc = lane * 4 + 256 * i;
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        v[i] = z; rres[i] = z; gg[i] = z; lg[i] = z; lb[i] = z;
        if (cc < H) {
            v[i] = FFN ? unpack4(ld8(Yr, (uint32_t)((row * H + cc) * 2))) : ldf4(Yr, (uint32_t)((row * H + cc) * 4));
            rres[i] = ldf4(R, (uint32_t)((row * H + cc) * 4));
            if (FFN) gg[i] = ld4(gain + cc);
            lg[i] = ld4(lng + cc);
            lb[i] = ld4(lnb + cc);
        }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) q += v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
    float rs = 1.f;
    if (FFN) {
        rs = 1.f / sqrtf(wave_sum(q) / (float)H + FLT_EPSILON);
        if (lane == 0) L.rstd_f[row] = rs;
    }
    float dp = 1.f;
    if (L.dpr > 0.f) dp = kk_drop_mul(c.seed, site + 2, (uint64_t)(row / c.S), kk_drop_threshold(L.dpr), 1.f / (1.f - L.dpr));
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int cc = lane * 4 + 256 * i;
        if (cc < H) {
            float o[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
            if (FFN) {
                const float4 g = gg[i];
                o[0] = o[0] * rs * g.x; o[1] = o[1] * rs * g.y; o[2] = o[2] * rs * g.z; o[3] = o[3] * rs * g.w;
            }
            const float rr[4] = {rres[i].x, rres[i].y, rres[i].z, rres[i].w};
            float m1[4], m2[4];
            kk_drop_mul4(c.seed, site, (uint64_t)row * H + cc, t1, k1, m1);
            kk_drop_mul4(c.seed, site + 1, (uint64_t)row * H + cc, t2, k2, m2);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = o[e] * (dp * m1[e] * m2[e]) + rr[e];
            v[i] = make_float4(o[0], o[1], o[2], o[3]);
            stf4(XO, (uint32_t)((row * H + cc) * 4), v[i]);
            s += o[0] + o[1] + o[2] + o[3];
        }
    }
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
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int cc = lane * 4 + 256 * i;
        if (cc < H) {
            const float4 g = lg[i], bb = lb[i];
            const float4 nn = make_float4((v[i].x - mean) * rstd * g.x + bb.x, (v[i].y - mean) * rstd * g.y + bb.y,
                                          (v[i].z - mean) * rstd * g.z + bb.z, (v[i].w - mean) * rstd * g.w + bb.w);
            if constexpr (sizeof(TN) == 2) st8(NO, (uint32_t)((row * H + cc) * 2), pack4(nn));
            else stf4(NO, (uint32_t)((row * H + cc) * 4), nn);
        }
    }
    if (lane == 0) {
        (FFN ? L.next_mean : L.mean2)[row] = mean;
        (FFN ? L.next_rstd : L.rstd2)[row] = rstd;
    }
}

template <bool FFN>
__device__ __forceinline__ void phase_tail(const Ctx &c, const KkEncLayer &L) {
    const int wave = threadIdx.x >> 6;
    for (int r = c.member + c.members * wave; r < c.S; r += c.members * NWAVES) {
        const int64_t row = (int64_t)c.b * c.S + r;
        if (FFN && !L.next_y_bf16) tail_row<FFN, float, 2>(c, L, row);
        else tail_row<FFN, __bf16, 2>(c, L, row);
    }
}

template <int HT, int FT>
__global__ __launch_bounds__(NTHREADS) void enc_stack_fwd_kernel(const KkEncStack a) {
    __shared__ __attribute__((aligned(16))) char smem[XS_BYTES + WS_BYTES];      // the whole 160 KB of the CU
    Ctx c;
    c.a = &a;
    c.S = a.S; c.H = a.H; c.F = a.F; c.heads = a.heads;
    c.RB = (a.S + 31) / 32;
    c.members = gridDim.x >> 3;
    c.member = a.placement ? blockIdx.x % c.members : blockIdx.x >> 3;
    c.seed = a.seed ? *a.seed : 0u;
    c.smem = smem;
    const int group = a.placement ? blockIdx.x / c.members : blockIdx.x & 7;
    GroupSync sy;
    sy.init(a.sync, group, c.members);
    uint64_t *tr = (a.trace && (int)blockIdx.x == a.trace_wg && threadIdx.x == 0) ? a.trace : nullptr;
    c.sub = tr ? tr + 2048 : nullptr;
    auto stamp = [&]() { if (tr) *tr++ = wall_clock64(); };
    stamp();
    if (group < a.B) prefetch_weights<HT, FT>(c, a.layer[0], G_QKV);
    for (int b = group; b < a.B; b += 8) {
        c.b = b;
        for (int l = 0; l < a.layers; ++l) {
            const KkEncLayer &L = a.layer[l];
            // whose q|k|v weights follow this layer's linear2: the next layer's, the next item's first layer's, or nobody's
            const int after = l + 1 < a.layers ? l + 1 : (b + 8 < a.B ? 0 : -1);
            // (the weights of the next GEMM phase start moving inside the barrier that ends the current one)
            phase_qkv<HT, FT>(c, L);
            stamp(); sy.arrive(); prefetch_weights<HT, FT>(c, L, G_WO); sy.wait(); stamp();
            phase_attn(c, L);
            stamp(); sy.barrier(); stamp();
            phase_wo<HT, FT>(c, L);
            stamp(); sy.arrive(); prefetch_weights<HT, FT>(c, L, G_LIN1); sy.wait(); stamp();
            phase_tail<false>(c, L);
            stamp(); sy.barrier(); stamp();
            phase_lin1<HT, FT>(c, L);
            stamp(); sy.arrive(); prefetch_weights<HT, FT>(c, L, G_LIN2); sy.wait(); stamp();
            phase_lin2<HT, FT>(c, L);
            stamp(); sy.arrive();
            if (after >= 0) prefetch_weights<HT, FT>(c, a.layer[after], G_QKV);
            sy.wait(); stamp();
            phase_tail<true>(c, L);
            stamp(); sy.barrier(); stamp();
        }
    }
    sy.exit();
}

}  // namespace

// One workgroup per CU, eight groups: the launch needs every workgroup resident at once (its barriers spin), so the grid
// follows the device's CU count (256 on MI355X; fewer under a CU mask or on a partitioned device -> fewer members per group,
// or "unsupported" when a member's share of a phase would no longer fit its LDS).
extern "C" int kk_encoder_stack_workgroups(void) {
    static int wgs = -1;
    if (wgs < 0) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
        wgs = cus >= 256 ? 256 : (cus / 8) * 8;
        // Residency: the launch is a plain one whose workgroups spin on each other, so every one of them must be
        // resident at once.  The grid never exceeds one workgroup per CU; ask the runtime that a CU admits one (registers, 160 KB of
        // LDS) for BOTH instantiations, and refuse the fused launch otherwise (the engine then takes the per-kernel sequence).  The
        // bounded spins + the error word remain the net for what the query cannot see (CU masks, a partitioned device).
        int nb0 = 0, nb1 = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb0, enc_stack_fwd_kernel<512, 1536>, NTHREADS, 0) != hipSuccess) nb0 = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb1, enc_stack_fwd_kernel<0, 0>, NTHREADS, 0) != hipSuccess) nb1 = 0;
        if (nb0 < 1 || nb1 < 1 || (int64_t)std::min(nb0, nb1) * cus < wgs) wgs = 0;
    }
    return wgs;
}

extern "C" int kk_encoder_stack_supported(int B, int S, int H, int F, int heads, int layers) {
    const int members = kk_encoder_stack_workgroups() / 8;
    // LDS map: a 64-row panel of K = H (<= 64 KB), up to 96 weight rows of K = H, a 32-row panel of K = F beside 16 rows of it;
    // per member at most one (part, head) of q|k|v, 16 columns of w_o / linear2, 3 blocks of 16 gate columns of linear1
    return members >= 1 && B > 0 && S > 0 && S <= SMAX && H > 0 && H <= 512 && H % 128 == 0 && F > 0 && F % 128 == 0 && F <= 1536 &&
           heads * 64 == H && layers > 0 && layers <= KK_ENC_MAX_LAYERS && 3 * heads <= members && H / 16 <= members &&
           (F / 16 + members - 1) / members <= 3;
}

extern "C" int kk_encoder_stack_fwd(const KkEncStack *d, void *stream) {
    KK_REQUIRE(d != nullptr, "kk_encoder_stack_fwd: null descriptor");
    KK_REQUIRE(kk_encoder_stack_supported(d->B, d->S, d->H, d->F, d->heads, d->layers),
               "kk_encoder_stack_fwd: unsupported shape B=%d S=%d H=%d F=%d heads=%d layers=%d (S <= %d, H <= 512, H and F multiples of 128, "
               "F <= 1536, head_dim 64)", d->B, d->S, d->H, d->F, d->heads, d->layers, SMAX);
    KK_REQUIRE(d->sync && d->cos_t && d->sin_t, "kk_encoder_stack_fwd: sync words and RoPE tables are required");
    for (int l = 0; l < d->layers; ++l) {
        const KkEncLayer &L = d->layer[l];
        KK_REQUIRE(L.w_qkv && L.g_q && L.g_k && L.g_v && L.w_o && L.b_o && L.ln2_g && L.ln2_b && L.w1 && L.b1 && L.w2 && L.b2 && L.ffn_gain &&
                       L.next_g && L.next_b && L.y1 && L.qkv_raw && L.qkv_n && L.ctx && L.lse && L.proj && L.x_in && L.xm && L.y2 && L.mean2 &&
                       L.rstd2 && L.h1 && L.g && L.f2 && L.rstd_f && L.xo && L.next_y && L.next_mean && L.next_rstd,
                   "kk_encoder_stack_fwd: layer %d has a null pointer", l);
        KK_REQUIRE(L.p >= 0.f && L.p < 1.f && L.dpr >= 0.f && L.dpr < 1.f, "kk_encoder_stack_fwd: probabilities must be in [0,1)");
        KK_REQUIRE(L.p == 0.f || d->seed, "kk_encoder_stack_fwd: dropout needs the seed");
    }
    // the model's own dimensions get the instantiation with compile-time loop bounds
    if (d->H == 512 && d->F == 1536)
        hipLaunchKernelGGL((enc_stack_fwd_kernel<512, 1536>), dim3(kk_encoder_stack_workgroups()), dim3(NTHREADS), 0, (hipStream_t)stream, *d);
    else
        hipLaunchKernelGGL((enc_stack_fwd_kernel<0, 0>), dim3(kk_encoder_stack_workgroups()), dim3(NTHREADS), 0, (hipStream_t)stream, *d);
    KK_LAUNCH_CHECK("kk_encoder_stack_fwd");
    return 0;
}
