// Fused (flash-style) multi-head attention, forward and backward, head_dim 64, for gfx950.
//
// Replaces F.scaled_dot_product_attention(Q,K,V, attn_mask=additive -inf bias) and its autograd backward
// (reference model/transformers.py:393-398; masks built at :299-316; causal mask model/model.py:434-437).
// The reference's SDPA math path materialises [B,h,Sq,Sk] scores/probabilities (268 MB per tensor at 8x1024);
// here nothing of size Sq x Sk ever leaves the CU.
//
// Work decomposition: one 256-thread workgroup per (batch, head, 128-row block); each of its 4 waves owns
// 32 rows.  K/V (or Q/dO in the dK/dV kernel) stream through LDS in 64-row tiles shared by the 4 waves.
//
// MFMA orientation ("swapped QK^T"): scores are computed TRANSPOSED, S^T[key][q] = K.Q^T, with the 32x32
// MFMA (C/D map: col = lane&31, row = (r&3)+8(r>>2)+4(lane>>5)).  A lane then owns one query column and 16 of
// the 32 keys of a sub-tile, so the softmax row reduction is 16 in-lane operations + one lane^32 exchange,
// and the probabilities are already laid out as the B operand of the second MFMA (O^T[d][q] = V^T.P^T) with
// no cross-lane traffic: the k-slots of that MFMA are bound to keys in the order the accumulator registers
// hold them, and the A operand (V^T) is read from a transposed LDS tile in that same order.
//
// KK_MATH_F32 runs the identical structure on v_mfma_f32_32x32x2_f32 (exact fp32) — the parity mode;
// KK_MATH_BF16 rounds Q/K/V/P/dS/dO to bf16 for the MFMAs and keeps scores, softmax and accumulators in fp32.
#include "kk_common.h"
#include <math.h>
#include <stdlib.h>
#include <algorithm>
#include <type_traits>

namespace {

template <bool BF16> struct ACfg;
template <> struct ACfg<true> { typedef __bf16 elem; static constexpr int LR = 72; };    // 144-byte rows
template <> struct ACfg<false> { typedef float elem; static constexpr int LR = 65; };

struct AttnArgs {
    const void *Q, *K, *V, *O, *dO;      // fp32, or bf16 when the kernel is instantiated with ST16 (bf16 storage)
    const float *LSE, *Delta;
    void *Out, *Out2;
    float *LSEo;
    float *DeltaOut;                     // dQ kernel: when set (with O), compute Delta = rowsum(dO * O) here and store it
    const uint8_t *key_mask;
    int B, heads, Sq, Sk, causal;
    int64_t ldq, ldk, ldv, ldo, lddo, ldout, ldout2;
    float scale;
    // dropout on the attention probabilities (SDPA dropout_p, transformers.py:396): mask = f(seed, site, element)
    const uint32_t *seed;
    uint32_t site;
    float p_drop;
    int xcd_map;
    int wt;                              // write-through stores of the [rows, 64] outputs (kk_common.h: kk_write_through(B * S))
    int dbg;                             // timing probes (KK_ATTN_DBG; results are wrong when set)
    int short_first;                     // attn_bwd_pair3: the dK/dV half of a causal launch hands out its SHORT blocks first (see there)
    void *dS;                            // kk_attn_bwd_ws: bf16 dS tiles, written by the dK/dV kernel, read by the dQ pass (kk_attn_bwd_dkv2.inc)
    // Packed keep decisions of the probability dropout (kk_attn_fwd_kb / kk_attn_bwd_kb): one bit per score, written by the
    // third-generation forward as the 16 ballots of every 32 x 32 unit it computes, read by the third-generation backward instead of
    // re-hashing — the hash was ~40 % of the backward's vector instructions.  Unit (qu, ku) of (b, head): 128 bytes at
    // (((b * heads + head) * nQU + qu) * nKU + ku) * 128, nQU = ceil(Sq / 32), nKU = ceil(Sk / 32); dword 2 r + h of a unit = bits over
    // the unit's 32 queries (bit = query) for key frag_row(r, h): the forward's ballot of accumulator register r, half h.
    void *keep;
    // weight warming (kk_attn_warm_next): the third-generation forward touches one dword per 128-byte line of up to two matrices the NEXT
    // launches multiply with — each XCD's workgroups share the lines out — during its last tile step, when its own DMAs are over: the
    // forward is vector-bound and its CUs' request slots are idle, and an XCD's L2 keeps read-only lines across the kernel boundary
    // (profiles/r06_l2_retention_probe.txt), so the GEMM behind it finds its weights L2-hot instead of in HBM
    const void *warm[2];
    uint32_t warm_bytes[2];
    int keep_rd;                         // forward: 1 = READ the keep bits (written by kk_attn_keep_gen beside the encoder forward) instead of hashing + storing them
    // backward kernels: the gradient of the per-head RMSNorm (+ RoPE) that produced Q (dQ kernel) / K and V (dK/dV
    // kernel: hn[0], hn[1]) as the epilogue — Out / Out2 then receive the gradient of the RAW projection
    KkAttnHeadNorm hn[2];
};

// Dropout on the probabilities.  The keep decision of element (b, head, q, key) is a 16-bit field of a 32-bit hash of
// (q, key >> 1), keyed by (seed, site, b, head): lanes that own a query get two decisions (key, key^1) per hash, and
// all three kernels evaluate the same function, so the backward regenerates the forward's mask exactly.  p is
// quantised to 1/65536 and 1/(1-p) is taken from the quantised value, so the mask stays unbiased.
struct ProbDrop {
    uint32_t thr, key, sk2;      // thr == 0: dropout off
    float inv_keep;
    template <typename A> __device__ __forceinline__ void init(A &a, int b, int hh) {
        thr = 0u;
        if (a.seed && a.p_drop > 0.f) {
            thr = (uint32_t)(a.p_drop * 65536.f + 0.5f);
            thr = thr > 65535u ? 65535u : thr;
        }
        key = thr ? kk_hash(*a.seed, a.site, (uint64_t)(b * a.heads + hh)) : 0u;
        inv_keep = thr ? 65536.f / (float)(65536u - thr) : 1.f;
        sk2 = (uint32_t)(a.Sk + 1) >> 1;
    }
    __device__ __forceinline__ uint32_t row(int q, int key0) const { return (uint32_t)q * sk2 + ((uint32_t)key0 >> 1); }
    // Two xorshift-multiply rounds with 24-bit multipliers: v_mul_u32_u24 is full rate, v_mul_lo_u32 quarter rate, and
    // the hash is a third of the softmax VALU work.  On the (q, key/2) counter lattice it tests like "lowbias32"
    // (keep rate, key/query/diagonal correlations at the 1e-3 noise floor, 8-bit pattern chi-square ~1; sweep in
    // tools/dropout_hash_quality.py); injective on counters below 2^24 (S <= 4096), distinct (b, head) differ by `key`.
    __device__ __forceinline__ uint32_t hash(uint32_t x) const {
        x ^= key;
        x ^= x >> 16; x = __umul24(x, 0xb5352du); x ^= x >> 13; x = __umul24(x, 0xca68b5u); x ^= x >> 16;
        return x;
    }
    // keep decisions; the 1/(1-p) of the kept elements is folded into an operand or an output scale by each kernel
    __device__ __forceinline__ bool keep_lo(uint32_t h) const { return (h & 0xFFFFu) >= thr; }   // even key
    __device__ __forceinline__ bool keep_hi(uint32_t h) const { return (h >> 16) >= thr; }       // odd key
};

// Edge sub-tiles without branches: bits 0 .. rel of a 32-bit word (rel < 0: none, rel >= 31: all).  A unit's visibility word is
// kk_low_bits(last visible element - first element of the unit) & ~(masked elements); a score is kept with v_bfe_i32 + v_and.
__device__ __forceinline__ uint32_t kk_low_bits(int rel) { return rel < 0 ? 0u : (rel >= 31 ? 0xFFFFFFFFu : (2u << rel) - 1u); }
// x with its bits ANDed by m (m = 0 or -1: v_bfe_i32 of a visibility / keep word).  Takes the value BY VALUE on purpose:
// __builtin_bit_cast applied directly to an element of an ext_vector (`bit_cast(int, acc[r])`) reads element 0 whatever r is (clang
// 19 / ROCm 7.2, seen in the disassembly: every select used the first accumulator register).
__device__ __forceinline__ float kk_andf(float x, int m) { return __builtin_bit_cast(float, __builtin_bit_cast(int, x) & m); }
__device__ __forceinline__ float kk_bfif(float x, int m, int other) { return __builtin_bit_cast(float, (__builtin_bit_cast(int, x) & m) | (~m & other)); }
typedef unsigned long long kk_u64x8 __attribute__((ext_vector_type(8)));
typedef const kk_u64x8 __attribute__((address_space(4))) kk_cu64x8;      // constant address space: a wave-uniform address gives s_load_dwordx16

// Workgroup -> (128-row block, batch*head).  The dispatcher places workgroup i (x fastest) on XCD i % 8, each with a private
// L2: in launch order the row blocks of one (batch, head) land on up to eight XCDs and every one of those L2s fetches that
// head's K and V (Q and dO in the dK/dV kernel) again.  With xcd_map set, the workgroups of XCD x are the blocks of the
// (batch, head) pairs = x (mod 8): a head's operands are fetched by one L2.  For causal launches the long blocks go first:
// per head (xcd_map = 1), or — xcd_map = 2, the causal pair launch of the backward — the longest blocks of ALL of an XCD's heads,
// then the second longest, ...: the dK/dV half of that launch is handed out as CUs finish their dQ block, and only in this order
// do the CUs that held the shortest dQ blocks receive the longest dK/dV blocks (5 block-units per CU instead of 7).
template <typename A> __device__ __forceinline__ void attn_block(A &a, int &bx, int &by, bool long_first_is_high) {
    bx = blockIdx.x; by = blockIdx.y;
    const int nx = gridDim.x, ny = gridDim.y;
    if (a.xcd_map && (ny & 7) == 0) {
        const int L = bx + nx * by, slot = L >> 3;
        if (a.xcd_map == 2) {                                  // block-major inside an XCD: ALL its longest blocks first (pair launch)
            const int per = ny >> 3;
            by = (L & 7) + 8 * (slot % per);
            bx = slot / per;
        } else {
            by = (L & 7) + 8 * (slot / nx);
            bx = slot % nx;
        }
    }
    if (a.causal && a.xcd_map) bx = long_first_is_high ? nx - 1 - bx : bx;
}

__device__ __forceinline__ float f4g(const float4 &v, int c) { return reinterpret_cast<const float *>(&v)[c]; }

// One row (this lane's row, lane&31) of a [32][64] fp32 matrix, held as an MFMA operand with k = d.
template <bool BF16> struct RowFrag;
template <> struct RowFrag<true> { bf16x8 v[4]; };    // v[ks][j] = X[row][16 ks + 8 half + j]
template <> struct RowFrag<false> { float v[32]; };   // v[ks]    = X[row][2 ks + half]

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

template <bool BF16>
__device__ __forceinline__ float rowfrag_dot(const RowFrag<BF16> &x, const RowFrag<BF16> &y) {   // this lane's 32 of the 64 d
    float s = 0.f;
    if constexpr (BF16) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) s += (float)x.v[ks][j] * (float)y.v[ks][j];
    } else {
#pragma unroll
        for (int j = 0; j < 32; ++j) s += x.v[j] * y.v[j];
    }
    return s;
}

template <bool BF16>
__device__ __forceinline__ void scale_rowfrag(RowFrag<BF16> &f, float k) {
    if constexpr (BF16) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) f.v[ks][j] = (__bf16)((float)f.v[ks][j] * k);
    } else {
#pragma unroll
        for (int j = 0; j < 32; ++j) f.v[j] *= k;
    }
}

template <bool BF16, typename T>
__device__ __forceinline__ void load_rowfrag(RowFrag<BF16> &f, const T *rowptr, int half) {
    if constexpr (BF16 && sizeof(T) == 2) {            // bf16 storage: the fragment is a plain 16-byte load
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if (rowptr) v = *reinterpret_cast<const u32x4 *>(rowptr + ks * 16 + half * 8);
            f.v[ks] = __builtin_bit_cast(bf16x8, v);
        }
    } else if constexpr (BF16) {
        const float *rp = reinterpret_cast<const float *>(rowptr);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
            if (rp) { a = ld4(rp + ks * 16 + half * 8); b = ld4(rp + ks * 16 + half * 8 + 4); }
#pragma unroll
            for (int e = 0; e < 4; ++e) { f.v[ks][e] = (__bf16)f4g(a, e); f.v[ks][4 + e] = (__bf16)f4g(b, e); }
        }
    } else {
        const float *rp = reinterpret_cast<const float *>(rowptr);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            if (rp) a = ld4(rp + 4 * j);
            f.v[2 * j] = half ? a.y : a.x;
            f.v[2 * j + 1] = half ? a.w : a.z;
        }
    }
}

// Staging of a [64][64] fp32 tile (row r at src + r*ld; rows >= nvalid read as zero) is split in two halves so the
// global loads of tile t+1 can be in flight while tile t is being multiplied: load_* fills 4 float4 registers,
// store_* converts and writes them to LDS.  "rows": LDS row-major S[64][LR];  "rows_T" (bf16 only): transposed
// St[d][row] (row contiguous), needed where the MFMA reduction runs over the tile's rows.
struct TileRegs { float4 r[4]; };

__device__ __forceinline__ void load_rows(TileRegs &t, const float *src, int64_t ld, int nvalid) {
    const int tl = threadIdx.x & 255, row = tl >> 2, seg = (tl & 3) * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) t.r[i] = row < nvalid ? ld4(src + (int64_t)row * ld + seg + 4 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
}

template <bool BF16>
__device__ __forceinline__ void store_rows(typename ACfg<BF16>::elem *S, const TileRegs &t) {
    constexpr int LR = ACfg<BF16>::LR;
    const int tl = threadIdx.x & 255, row = tl >> 2, seg = (tl & 3) * 16;
    if constexpr (BF16) {
        bf16x8 lo, hi;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            lo[e] = (__bf16)f4g(t.r[0], e); lo[4 + e] = (__bf16)f4g(t.r[1], e);
            hi[e] = (__bf16)f4g(t.r[2], e); hi[4 + e] = (__bf16)f4g(t.r[3], e);
        }
        *reinterpret_cast<bf16x8 *>(&S[row * LR + seg]) = lo;
        *reinterpret_cast<bf16x8 *>(&S[row * LR + seg + 8]) = hi;
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) S[row * LR + seg + 4 * i + e] = f4g(t.r[i], e);
    }
}

__device__ __forceinline__ void load_rows_T(TileRegs &t, const float *src, int64_t ld, int nvalid) {
    const int tl = threadIdx.x & 255, rg = (tl & 15) * 4, dg = (tl >> 4) * 4;
#pragma unroll
    for (int c = 0; c < 4; ++c) t.r[c] = (rg + c) < nvalid ? ld4(src + (int64_t)(rg + c) * ld + dg) : make_float4(0.f, 0.f, 0.f, 0.f);
}

__device__ __forceinline__ void store_rows_T(__bf16 *St, const TileRegs &t) {
    constexpr int LR = ACfg<true>::LR;
    const int tl = threadIdx.x & 255, rg = (tl & 15) * 4, dg = (tl >> 4) * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        bf16x4 v;
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = (__bf16)f4g(t.r[c], e);
        *reinterpret_cast<bf16x4 *>(&St[(dg + e) * LR + rg]) = v;
    }
}

// The same two staging patterns for tiles that are ALREADY bf16 in HBM (bf16 storage): no conversion, half the bytes.
struct TileRegs16 { u32x4 r[2]; };     // "rows" pattern: 16 contiguous bf16 of one row
struct TileRegs16T { u32x2 r[4]; };    // "rows_T" pattern: 4 rows x 4 contiguous bf16

__device__ __forceinline__ void load_rows(TileRegs16 &t, const __bf16 *src, int64_t ld, int nvalid) {
    const int tl = threadIdx.x & 255, row = tl >> 2, seg = (tl & 3) * 16;
    const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < 2; ++i) t.r[i] = row < nvalid ? *reinterpret_cast<const u32x4 *>(src + (int64_t)row * ld + seg + 8 * i) : z;
}
__device__ __forceinline__ void store_rows16(__bf16 *S, const TileRegs16 &t) {
    constexpr int LR = ACfg<true>::LR;
    const int tl = threadIdx.x & 255, row = tl >> 2, seg = (tl & 3) * 16;
    *reinterpret_cast<u32x4 *>(&S[row * LR + seg]) = t.r[0];
    *reinterpret_cast<u32x4 *>(&S[row * LR + seg + 8]) = t.r[1];
}
__device__ __forceinline__ void load_rows_T(TileRegs16T &t, const __bf16 *src, int64_t ld, int nvalid) {
    const int tl = threadIdx.x & 255, rg = (tl & 15) * 4, dg = (tl >> 4) * 4;
    const u32x2 z = {0u, 0u};
#pragma unroll
    for (int c = 0; c < 4; ++c) t.r[c] = (rg + c) < nvalid ? *reinterpret_cast<const u32x2 *>(src + (int64_t)(rg + c) * ld + dg) : z;
}
__device__ __forceinline__ void store_rows_T16(__bf16 *St, const TileRegs16T &t) {
    constexpr int LR = ACfg<true>::LR;
    const int tl = threadIdx.x & 255, rg = (tl & 15) * 4, dg = (tl >> 4) * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {        // element e of rows 0..3 -> 4 contiguous bf16 of transposed row dg+e
        const int w = e >> 1, sh = 16 * (e & 1);
        u32x2 v;
        v[0] = ((t.r[0][w] >> sh) & 0xFFFFu) | (((t.r[1][w] >> sh) & 0xFFFFu) << 16);
        v[1] = ((t.r[2][w] >> sh) & 0xFFFFu) | (((t.r[3][w] >> sh) & 0xFFFFu) << 16);
        *reinterpret_cast<u32x2 *>(&St[(dg + e) * LR + rg]) = v;
    }
}

// Uniform front end: Stage<BF16, ST16> picks the register type and the load/store pair for a tile.
template <bool BF16, bool ST16> struct Stage {
    typedef TileRegs R;
    typedef TileRegs RT;
    typedef float T;
    static __device__ __forceinline__ void st(typename ACfg<BF16>::elem *S, const R &r) { store_rows<BF16>(S, r); }
    static __device__ __forceinline__ void stT(__bf16 *S, const RT &r) { store_rows_T(S, r); }
};
template <> struct Stage<true, true> {
    typedef TileRegs16 R;
    typedef TileRegs16T RT;
    typedef __bf16 T;
    static __device__ __forceinline__ void st(__bf16 *S, const R &r) { store_rows16(S, r); }
    static __device__ __forceinline__ void stT(__bf16 *S, const RT &r) { store_rows_T16(S, r); }
};

// acc[row][col] += sum_d T[r0 + row][d] * F_col[d]: A operand = 32 rows of the LDS tile, B operand = RowFrag.
template <bool BF16>
__device__ __forceinline__ void mma_tile_x_frag(f32x16 &acc, const typename ACfg<BF16>::elem *T, int r0,
                                                const RowFrag<BF16> &f, int l31, int half) {
    constexpr int LR = ACfg<BF16>::LR;
    if constexpr (BF16) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 a = *reinterpret_cast<const bf16x8 *>(&T[(r0 + l31) * LR + ks * 16 + half * 8]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, f.v[ks], acc, 0, 0, 0);
        }
    } else {
#pragma unroll
        for (int ks = 0; ks < 32; ++ks)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(T[(r0 + l31) * LR + 2 * ks + half], f.v[ks], acc, 0, 0, 0);
    }
}

// out[db][d_local][col] += sum_{rows of sub-tile} X[row][db*32 + d_local] * p[row][col], where p[16] are this
// lane's accumulator-layout values (row_local = frag_row(r, half), col = lane&31).  bf16: Tx is the TRANSPOSED
// tile [d][row]; fp32: Tx is the row-major tile [row][d].
template <bool BF16>
__device__ __forceinline__ void mma_T_x_p(f32x16 (&out)[2], const typename ACfg<BF16>::elem *Tx, int sub0,
                                          const float (&p)[16], int l31, int half) {
    constexpr int LR = ACfg<BF16>::LR;
    if constexpr (BF16) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            bf16x8 b;
#pragma unroll
            for (int j = 0; j < 8; ++j) b[j] = (__bf16)p[8 * s2 + j];
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const __bf16 *base = &Tx[(db * 32 + l31) * LR + sub0 + 16 * s2 + 4 * half];
                const bf16x4 lo = *reinterpret_cast<const bf16x4 *>(base);
                const bf16x4 hi = *reinterpret_cast<const bf16x4 *>(base + 8);
                bf16x8 a;
#pragma unroll
                for (int e = 0; e < 4; ++e) { a[e] = lo[e]; a[4 + e] = hi[e]; }
                out[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, out[db], 0, 0, 0);
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = sub0 + frag_row(r, half);
#pragma unroll
            for (int db = 0; db < 2; ++db)
                out[db] = __builtin_amdgcn_mfma_f32_32x32x2f32(Tx[row * LR + db * 32 + l31], p[r], out[db], 0, 0, 0);
        }
    }
}

__device__ __forceinline__ void zero_acc(f32x16 &a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = 0.f;
}

// Store a transposed accumulator pair acc[db][r] (row = d, col = this lane's matrix row) to dst_row[0..63].
template <typename T>
__device__ __forceinline__ void store_row(T *dst_row, const f32x16 (&acc)[2], float mul, int half) {
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            stv4<T>(dst_row + db * 32 + 8 * g + 4 * half,
                make_float4(acc[db][4 * g] * mul, acc[db][4 * g + 1] * mul, acc[db][4 * g + 2] * mul, acc[db][4 * g + 3] * mul));
}

// Epilogue of the backward kernels: gradient of y = RMSNorm64(x)*gain (+ RoPE) for the (row, head) vector this lane
// pair holds (same math as headnorm_rope_bwd_kernel, kk_norm.hip; acc*mul is first rounded to the storage type, as
// the unfused path's store does).  A lane has d = db*32 + 8g + 4*half + e, so rotate_half's partner d^32 is its own
// acc[db^1] element and the two row reductions are a local sum plus one xor-32 shuffle.  The gain gradient needs
// column sums over the workgroup's 128 rows: every lane drops dn*x*rstd into colred[row][65] and hn_colsum() adds
// the columns after a barrier.  Returns nothing; `valid` lanes store dx.
template <typename T>
__device__ __forceinline__ void hn_bwd_row(const f32x16 (&acc)[2], float mul, bool valid, const T *raw_row, T *out_row,
                                           const KkAttnHeadNorm &h, int pos, int half, float *colred_row) {
    float dn[32], v[32];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 x4 = valid ? ldv4<T>(raw_row + db * 32 + 8 * g + 4 * half) : make_float4(0.f, 0.f, 0.f, 0.f);
            v[db * 16 + 4 * g] = x4.x; v[db * 16 + 4 * g + 1] = x4.y; v[db * 16 + 4 * g + 2] = x4.z; v[db * 16 + 4 * g + 3] = x4.w;
#pragma unroll
            for (int e = 0; e < 4; ++e) dn[db * 16 + 4 * g + e] = valid ? (float)(T)(acc[db][4 * g + e] * mul) : 0.f;
        }
    float ssq = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) ssq += v[i] * v[i];
    ssq += __shfl_xor(ssq, 32, 64);
    const float rs = 1.f / sqrtf(ssq * (1.f / 64.f) + 1.1920928955078125e-7f);
    if (h.rope) {     // dn[d] = dy[d] cos[d] + (d < 32 ? dy[d+32] sin[d+32] : -dy[d-32] sin[d-32])
        const int64_t pr = valid ? pos : 0;                      // (rows past the end of the sequence have no table row)
        const float *cr = h.cos_t + pr * 64, *sr = h.sin_t + pr * 64;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d = 8 * g + 4 * half;
            const float4 c0 = ld4(cr + d), c1 = ld4(cr + 32 + d), s0 = ld4(sr + d), s1 = ld4(sr + 32 + d);
            const float cl[4] = {c0.x, c0.y, c0.z, c0.w}, ch[4] = {c1.x, c1.y, c1.z, c1.w};
            const float sl[4] = {s0.x, s0.y, s0.z, s0.w}, sh[4] = {s1.x, s1.y, s1.z, s1.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float lo = dn[4 * g + e], hi = dn[16 + 4 * g + e];
                dn[4 * g + e] = lo * cl[e] + hi * sh[e];
                dn[16 + 4 * g + e] = hi * ch[e] - lo * sl[e];
            }
        }
    }
    float kdot = 0.f;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 g4 = ld4(h.gain + db * 32 + 8 * g + 4 * half);
            const float gg[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = db * 16 + 4 * g + e;
                colred_row[db * 32 + 8 * g + 4 * half + e] = dn[i] * v[i] * rs;
                dn[i] *= gg[e];                                  // dg
                kdot += dn[i] * v[i];
            }
        }
    kdot += __shfl_xor(kdot, 32, 64);
    const float k = kdot * (1.f / 64.f) * rs * rs * rs;
    if (valid) {
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int i = db * 16 + 4 * g;
                stv4<T>(out_row + db * 32 + 8 * g + 4 * half,
                        make_float4(rs * dn[i] - v[i] * k, rs * dn[i + 1] - v[i + 1] * k, rs * dn[i + 2] - v[i + 2] * k, rs * dn[i + 3] - v[i + 3] * k));
            }
    }
}
// column sums of colred[128][65] -> partials[workgroup][64] (threads 0..63 of the workgroup; call between barriers)
__device__ __forceinline__ void hn_colsum(const float *colred, float *partials) {
    if (threadIdx.x < 64) {
        float s0 = 0.f, s1 = 0.f;
#pragma unroll 8
        for (int r = 0; r < 128; r += 2) { s0 += colred[r * 65 + threadIdx.x]; s1 += colred[(r + 1) * 65 + threadIdx.x]; }
        partials[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 64 + threadIdx.x] = s0 + s1;
    }
}

// ------------------------------------------------------------------ forward
// G = 1: 4 waves, every wave sees every key tile.  G = 2: 8 waves (2 per SIMD — the second wave's MFMAs and LDS
// latencies hide under the first one's softmax VALU work and vice versa); wave group g takes the key tiles
// g, g+2, g+4, ... of the same 128 queries with its own running (max, sum, O), and the two partial softmaxes are
// merged through LDS at the end.  Each group stages its own tiles with its own 256 threads.
template <bool BF16, bool ST16, int G>
__global__ __launch_bounds__(256 * G) void attn_fwd_kernel(AttnArgs a) {
    using elem = typename ACfg<BF16>::elem;
    using SG = Stage<BF16, ST16>;
    using T = typename SG::T;
    constexpr int LR = ACfg<BF16>::LR, TILE = 64 * LR;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];        // [buffer][group][K | V]
    elem *smem = reinterpret_cast<elem *>(smem_raw);
    int bx_, by_;
    attn_block(a, bx_, by_, true);                         // (causal: blocks near the end of the sequence see the most keys)
    const int b = by_ / a.heads, hh = by_ % a.heads;
    const int qblk = bx_ * 128;
    const int lane = threadIdx.x & 63, wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), half = lane >> 5, l31 = lane & 31;
    const int wave = wave8 & 3, grp = wave8 >> 2;
    const int q = qblk + wave * 32 + l31;
    const bool qvalid = q < a.Sq;
    RowFrag<BF16> qf;
    load_rowfrag<BF16, T>(qf, qvalid ? static_cast<const T *>(a.Q) + ((int64_t)b * a.Sq + q) * a.ldq + hh * 64 : nullptr, half);
    f32x16 o[2];
    zero_acc(o[0]); zero_acc(o[1]);
    float m = -1e30f, l = 0.f;                            // running max in the log2 domain, running sum
    const float c2 = a.scale * 1.4426950408889634f;       // exp(x*scale) = exp2(x*c2)
    ProbDrop pd;
    pd.init(a, b, hh);
    const uint8_t *km = a.key_mask ? a.key_mask + (int64_t)b * a.Sk : nullptr;
    const int qmin = qblk + wave * 32;                    // smallest query of this wave
    int kend = a.Sk;
    if (a.causal && qblk + 128 < kend) kend = qblk + 128;
    const T *Kb = static_cast<const T *>(a.K) + (int64_t)b * a.Sk * a.ldk + hh * 64;
    const T *Vb = static_cast<const T *>(a.V) + (int64_t)b * a.Sk * a.ldv + hh * 64;
    // Register staging with a prefetch distance of TWO tiles: the kernel is bound by the latency of the K/V loads, not by
    // bandwidth or math (19 us at S=512 with one tile ahead: four dependent ~2 us round trips), so two register sets
    // alternate and a tile's loads have two tile-times to land before they are written to LDS.
    struct Regs {
        typename SG::R rk;
        typename std::conditional<BF16, typename SG::RT, typename SG::R>::type rv;
        uint32_t rkm;                                     // key-mask byte of key (tile start + lane)
    };
    Regs ra, rb;
    ra.rkm = rb.rkm = 0;
    auto issue = [&](Regs &t, int k0) {
        const int nvalid = a.Sk - k0 < 64 ? a.Sk - k0 : 64;
        load_rows(t.rk, Kb + (int64_t)k0 * a.ldk, a.ldk, nvalid);
        if constexpr (BF16) load_rows_T(t.rv, Vb + (int64_t)k0 * a.ldv, a.ldv, nvalid);
        else load_rows(t.rv, Vb + (int64_t)k0 * a.ldv, a.ldv, nvalid);
        t.rkm = km ? (lane < nvalid ? km[k0 + lane] : 0u) : 0u;
    };
    auto commit = [&](const Regs &t, int buf) {
        elem *dst = smem + (buf * G + grp) * 2 * TILE;
        SG::st(dst, t.rk);
        if constexpr (BF16) SG::stT(dst + TILE, t.rv);
        else SG::st(dst + TILE, t.rv);
    };
    constexpr int STEP = 64 * G;                          // this group's tiles: grp*64, grp*64 + STEP, ...
    const int kfirst = grp * 64;
    if (kfirst < kend) {
        issue(ra, kfirst);
        commit(ra, 0);
    }
    uint64_t kmbits = __ballot(ra.rkm != 0u), kmnext = 0;  // bit j: key (tile start + j) is masked (wave-uniform)
    if (kfirst + STEP < kend) issue(ra, kfirst + STEP);         // tile 1 -> set a
    if (kfirst + 2 * STEP < kend) issue(rb, kfirst + 2 * STEP); // tile 2 -> set b
    __syncthreads();
    int cur = 0;
    // one tile: multiply tile k0 from LDS buffer `cur`, then write tile k0+STEP (register set X) to the other buffer and
    // reuse X for tile k0+3*STEP
    auto tile_step = [&](Regs &X, int kk0) {
        const int k0 = kk0 + kfirst;
        const elem *Ks = smem + (cur * G + grp) * 2 * TILE, *Vx = Ks + TILE;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int kb = k0 + sub * 32;
            if (kb >= kend) continue;
            if (a.causal && kb > qmin + 31) continue;
            f32x16 s;
            zero_acc(s);
            mma_tile_x_frag<BF16>(s, Ks, sub * 32, qf, l31, half);
            const uint32_t kmsub = (uint32_t)(kmbits >> (sub * 32));
            // masks are only evaluated on edge sub-tiles: ragged end, causal diagonal, or a masked key among the 32
            const bool edge = kb + 32 > a.Sk || (a.causal && kb + 31 > qmin) || kmsub != 0u;
            float p[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) p[r] = s[r];
            if (edge) {
                const uint32_t kml = kmsub >> (4 * half);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kb + frag_row(r, half);
                    const bool ok = key < a.Sk && !(a.causal && key > q) && !((kml >> frag_row(r, 0)) & 1u);
                    p[r] = ok ? p[r] : -INFINITY;
                }
            }
            float mx = p[0];                               // max of the raw scores; the scale c2 > 0 commutes with max
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, p[r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * c2;
            const float mn = fmaxf(m, mx);
            if (__ballot(mn > m) != 0ull) {               // rescale only when some row's maximum moved (wave-uniform)
                const float alpha = __builtin_amdgcn_exp2f(m - mn);
                l *= alpha;
#pragma unroll
                for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
                m = mn;
            }
            float rs = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { p[r] = __builtin_amdgcn_exp2f(fmaf(p[r], c2, -m)); rs += p[r]; }
            rs += __shfl_xor(rs, 32, 64);
            l += rs;
            if (pd.thr) {                // the row sum l stays un-dropped: softmax first, dropout after; 1/(1-p) at the store
                const uint32_t xb = pd.row(q, kb + 4 * half);
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const uint32_t hsh = pd.hash(xb + (uint32_t)(frag_row(r, 0) >> 1));
                    p[r] = pd.keep_lo(hsh) ? p[r] : 0.f;
                    p[r + 1] = pd.keep_hi(hsh) ? p[r + 1] : 0.f;
                }
            }
            mma_T_x_p<BF16>(o, Vx, sub * 32, p, l31, half);
        }
        kmnext = 0;
        if (k0 + STEP < kend) {
            commit(X, cur ^ 1);
            kmnext = __ballot(X.rkm != 0u);
            if (k0 + 3 * STEP < kend) issue(X, k0 + 3 * STEP);
        }
        kmbits = kmnext;
        __syncthreads();
        cur ^= 1;
    };
    for (int kk0 = 0; kk0 < (KK_DBG(a, 32) ? 0 : kend); kk0 += 2 * STEP) {      // the bound is the same for both groups (barriers)
        tile_step(ra, kk0);
        if (kk0 + STEP < kend) tile_step(rb, kk0 + STEP);
    }
    if constexpr (G >= 2) {          // merge the key groups' partial softmaxes: groups 1 .. G-1 -> LDS -> group 0
        float *mb0 = reinterpret_cast<float *>(smem_raw) + (wave * 64 + lane) * 34;     // (the loop's last barrier is behind us)
        constexpr int GSTRIDE = 4 * 64 * 34;                                             // floats per group
        if (grp >= 1) {
            float *mb = mb0 + (grp - 1) * GSTRIDE;
            mb[0] = m; mb[1] = l;
#pragma unroll
            for (int r = 0; r < 16; ++r) { mb[2 + r] = o[0][r]; mb[18 + r] = o[1][r]; }
        }
        __syncthreads();
        if (grp >= 1) return;
#pragma unroll
        for (int g = 1; g < G; ++g) {
            const float *mb = mb0 + (g - 1) * GSTRIDE;
            const float m1 = mb[0], l1 = mb[1], mn = fmaxf(m, m1);
            const float a0 = __builtin_amdgcn_exp2f(m - mn), a1 = __builtin_amdgcn_exp2f(m1 - mn);
            l = l * a0 + l1 * a1;
            m = mn;
#pragma unroll
            for (int r = 0; r < 16; ++r) { o[0][r] = o[0][r] * a0 + mb[2 + r] * a1; o[1][r] = o[1][r] * a0 + mb[18 + r] * a1; }
        }
    }
    if (qvalid) {
        const float inv = l > 0.f ? pd.inv_keep / l : 0.f;
        store_row<T>(static_cast<T *>(a.Out) + ((int64_t)b * a.Sq + q) * a.ldout + hh * 64, o, inv, half);
        if (half == 0) a.LSEo[((int64_t)b * a.heads + hh) * a.Sq + q] = l > 0.f ? (m + __builtin_amdgcn_logf(l)) * 0.6931471805599453f : INFINITY;
    }
}

// ------------------------------------------------------------------ forward, second generation (bf16 storage)
// Same decomposition and the same arithmetic per score as attn_fwd_kernel<true, true, 2> (bit-identical dropout masks), but
//  * K and V tiles reach LDS by the buffer-load-to-LDS DMA (16 bytes per lane, no VGPR staging, no ds_write pass, no
//    software transpose): K as a [64 keys][64 d] image with XOR-ed 16-byte chunks (conflict-free ds_read_b128 fragments),
//    V exactly as it lies in memory with XOR-ed 32-byte blocks, its V^T fragments read by ds_read_b64_tr_b16 (the
//    hardware 4x16 transpose read) in the key order the accumulator registers hold the probabilities;
//  * NS stages per wave group, one raw s_barrier per 64-key tile, DMA waits by counted vmcnt;
//  * the wave is software-pipelined over 32-key units: the QK^T MFMAs of unit u+1 are issued BEFORE the softmax of
//    unit u, so the matrix pipe works under the VALU phase of the same wave (the first-generation kernel alternated
//    strictly: its phases add up).
typedef float f32x4_ __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
#define KK_LDS_PTR(p) ((__attribute__((address_space(3))) void *)(p))

__device__ __forceinline__ bf16x8 tr_pair(const s16x4 &lo, const s16x4 &hi) {
    s16x8 v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
    return __builtin_bit_cast(bf16x8, v);
}

// value of the lane 32 away combined with the own one, by v_permlane32_swap (a VALU instruction; __shfl_xor(.., 32) is a
// ds_bpermute round trip through the LDS queue): the swap of v with itself returns {own, partner} in some order
// (inline asm: with this compiler __builtin_amdgcn_permlane32_swap hands back its FIRST result for both elements of the
// returned pair — `v_add_f32 v, v9, v9` after the swap; the s_nops cover the VALU-write -> swap -> VALU-read wait states)
__device__ __forceinline__ void xor32_pair(float v, float &lo, float &hi) {
    lo = v; hi = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(lo), "+v"(hi));
}
__device__ __forceinline__ float xor32_sum(float v) {
    float a, b;
    xor32_pair(v, a, b);
    return a + b;
}
__device__ __forceinline__ float xor32_max(float v) {
    float a, b;
    xor32_pair(v, a, b);
    return fmaxf(a, b);
}

// ---- coalesced prologue / epilogue pieces of the second-generation kernels.  A row-per-lane access (one 128-byte head row
// per lane: the RowFrag loads, store_row) touches 32 lines per wave instruction and is bound by requests, not bytes
// (4 us of a 13 us forward launch were the Q loads and the O stores).
// DMA of a [128 rows][64] bf16 head tile into a 16 KB LDS image with XOR-ed 16-byte chunks, by 512 threads (two pieces each).
template <int NT = 512>                 // threads of the workgroup (512: two pieces each, 256: four)
__device__ __forceinline__ void dma_rows128(const __bf16 *base, int64_t ld, int nrows, char *img, int wave8) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(base), 0, nrows > 0 ? (int)((((int64_t)nrows - 1) * ld + 64) * 2) : 0, 0x00020000);
#pragma unroll
    for (int j = 0; j < 1024 / NT; ++j) {
        const int p = threadIdx.x + NT * j, row = p >> 3, pc = p & 7;
        const uint32_t vo = (uint32_t)(((int64_t)row * ld + ((pc ^ ((row >> 1) & 7)) * 8)) * 2);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, KK_LDS_PTR(img + wave8 * 1024 + j * (NT * 16)), 16, vo, 0, 0, 0);
    }
}
// this lane's row (row0 + lane&31, row0 a multiple of 16) of such an image as an MFMA operand with k = d (RowFrag layout)
__device__ __forceinline__ void rowfrag_from_image(RowFrag<true> &f, const char *img, int row0, int l31, int half) {
    const uint32_t base = (uint32_t)(uintptr_t)KK_LDS_PTR(img) + (uint32_t)((row0 + l31) * 128);
    const int swz = (l31 >> 1) & 7;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) asm volatile("ds_read_b128 %0, %1" : "=v"(f.v[ks]) : "v"(base + (uint32_t)(((2 * ks + half) ^ swz) * 16)));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(f.v[ks]));
}
// Store the wave's transposed accumulator pair (acc[db][r]: d = db*32 + 8(r>>2) + 4 half + (r&3), row = lane&31) times mul
// as 32 bf16 rows of 64 through a wave-private 4608-byte LDS tile: 16-byte global stores, eight lanes per 128-byte row.
__device__ __forceinline__ void store_rows_via_lds(__bf16 *dst_row0, int64_t ld, int nvalid, const f32x16 (&acc)[2], float mul,
                                                   char *tile, int lane, int wt) {
    const int l31 = lane & 31, half = lane >> 5;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            bf16x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (__bf16)(acc[db][4 * g + e] * mul);
            *reinterpret_cast<bf16x4 *>(tile + l31 * 144 + (db * 32 + 8 * g + 4 * half) * 2) = v;
        }
    __builtin_amdgcn_wave_barrier();                       // (one wave: its LDS operations complete in order)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = (lane >> 3) + 8 * j, c = lane & 7;
        const u32x4 v = *reinterpret_cast<const u32x4 *>(tile + row * 144 + c * 16);
        if (row < nvalid) kk_store16(dst_row0 + (int64_t)row * ld + c * 8, v, wt);
    }
}

template <int NS>
__global__ __launch_bounds__(512) void attn_fwd2_kernel(AttnArgs a) {
    typedef __bf16 T;
    constexpr int KIMG = 64 * 64 * 2, STAGE = 2 * KIMG, GSZ = NS * STAGE, STEP = 128;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];        // [group][stage][K image | V image], key-mask words
    uint64_t *kmb = reinterpret_cast<uint64_t *>(smem_raw + 2 * GSZ);       // [64] one word per 64-key tile
    int bx_, by_;
    attn_block(a, bx_, by_, true);                         // (causal: blocks near the end of the sequence see the most keys)
    const int b = by_ / a.heads, hh = by_ % a.heads;
    const int qblk = bx_ * 128;
    const int lane = threadIdx.x & 63, wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), half = lane >> 5, l31 = lane & 31;
    const int wave = wave8 & 3, grp = wave8 >> 2, tg = threadIdx.x & 255;
    const int q = qblk + wave * 32 + l31;
    const bool qvalid = q < a.Sq;
    const int qmin = qblk + wave * 32;
    int kend = a.Sk;
    if (a.causal && qblk + 128 < kend) kend = qblk + 128;
    int klim = kend;                                       // this wave multiplies the keys [0, klim)
    if (a.causal && qmin + 32 < klim) klim = qmin + 32;
    const int kfirst = grp * 64;
    const int nt = kfirst < kend ? (kend - kfirst + STEP - 1) / STEP : 0;       // this group's tiles
    const int nt0 = (kend + STEP - 1) / STEP;                                   // group 0's: the loop bound (barriers)
    int nu = 0;                                            // this wave's 32-key units: a prefix of the group's
    if (klim > kfirst) {
        const int full = (klim - kfirst) / STEP, rem = (klim - kfirst) - full * STEP;
        nu = 2 * full + (rem > 32 ? 2 : (rem > 0 ? 1 : 0));
    }
    RowFrag<true> qf;
    if (KK_DBG(a, 64)) return;                                // (timing probe: the launch alone)
    // probe (tools builds, bit 256): shader-clock stamps of workgroup 0's waves into the buffer at a.DeltaOut
    unsigned long long *trace = (KK_DBG(a, 256) && blockIdx.x == 0 && blockIdx.y == 0 && lane == 0) ? reinterpret_cast<unsigned long long *>(a.DeltaOut) + wave8 * 64 : nullptr;
    auto stamp = [&](int slot) { if (KK_DBG(a, 256) && trace != nullptr && slot < 64) trace[slot] = __builtin_amdgcn_s_memtime(); };
    stamp(0);
    char *qimg = smem_raw + 2 * GSZ + 512;                 // [128 queries][64] image (the oldest DMA: covered by every wait below)
    dma_rows128(static_cast<const T *>(a.Q) + ((int64_t)b * a.Sq + qblk) * a.ldq + hh * 64, a.ldq, a.Sq - qblk < 128 ? a.Sq - qblk : 128, qimg, wave8);
    const uint8_t *km = a.key_mask ? a.key_mask + (int64_t)b * a.Sk : nullptr;
    uint32_t kmv[8];
    if (km) {                                              // tile T's mask word: wave T % 8 (issued before the DMAs)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int key = (wave8 + 8 * i) * 64 + lane;
            kmv[i] = key < kend ? km[key] : 0u;
        }
    }
    // ---- DMA
    const T *Kb = static_cast<const T *>(a.K) + (int64_t)b * a.Sk * a.ldk + hh * 64;
    const T *Vb = static_cast<const T *>(a.V) + (int64_t)b * a.Sk * a.ldv + hh * 64;
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(Kb), 0, (int)((((int64_t)a.Sk - 1) * a.ldk + 64) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(Vb), 0, (int)((((int64_t)a.Sk - 1) * a.ldv + 64) * 2), 0x00020000);
    uint32_t kvo[2], vvo[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int p = tg + 256 * j, row = p >> 3, pc = p & 7;
        kvo[j] = (uint32_t)(((int64_t)row * a.ldk + ((pc ^ ((row >> 1) & 7)) * 8)) * 2);
        const int sw = 2 * ((row >> 1) & 1), g = (((pc >> 1) ^ sw) << 1) | (pc & 1);
        vvo[j] = (uint32_t)(((int64_t)row * a.ldv + g * 8) * 2);
    }
    char *gbase = smem_raw + grp * GSZ;
    const uint32_t ktile = (uint32_t)(STEP * a.ldk * 2), vtile = (uint32_t)(STEP * a.ldv * 2);
    const uint32_t kbeg = (uint32_t)(kfirst * a.ldk * 2), vbeg = (uint32_t)(kfirst * a.ldv * 2);
    auto issue_tile = [&](int t, int st) {                 // (offsets in the VGPR: the range check then covers the tile's rows)
        char *dst = gbase + st * STAGE + wave * 1024;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, KK_LDS_PTR(dst + j * 4096), 16, kvo[j] + kbeg + (uint32_t)t * ktile, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, KK_LDS_PTR(dst + KIMG + j * 4096), 16, vvo[j] + vbeg + (uint32_t)t * vtile, 0, 0, 0);
    };
#pragma unroll
    for (int t = 0; t < NS; ++t)
        if (t < nt) issue_tile(t, t);
    if (km) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint64_t bits = __ballot(kmv[i] != 0u);
            if (lane == 0) kmb[wave8 + 8 * i] = bits;
        }
    }
    // ---- fragment addresses (bytes inside a stage)
    const uint32_t gl = (uint32_t)(uintptr_t)KK_LDS_PTR(gbase);
    uint32_t ka[4], va[2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) ka[ks] = (uint32_t)(l31 * 128 + (((2 * ks + half) ^ ((l31 >> 1) & 7)) * 16));
    {
        const int L = lane & 15, kq = L >> 2, gi = (lane >> 4) & 1, sw = 2 * ((kq >> 1) & 1);
#pragma unroll
        for (int db = 0; db < 2; ++db) va[db] = (uint32_t)((4 * half + kq) * 128 + (((2 * db + gi) ^ sw) * 32) + 8 * (L & 3));
    }
    bf16x8 kf[4];
    s16x4 vlo[4], vhi[4];
    // (plain lambdas with literal offsets: inline-asm operands are not captured inside generic lambdas)
    auto read_k = [&](uint32_t img) {                      // img = LDS address of the unit's first K row
        if (KK_DBG(a, 16)) return;
        asm volatile("ds_read_b128 %0, %1" : "=v"(kf[0]) : "v"(img + ka[0]));
        asm volatile("ds_read_b128 %0, %1" : "=v"(kf[1]) : "v"(img + ka[1]));
        asm volatile("ds_read_b128 %0, %1" : "=v"(kf[2]) : "v"(img + ka[2]));
        asm volatile("ds_read_b128 %0, %1" : "=v"(kf[3]) : "v"(img + ka[3]));
    };
    auto read_v = [&](uint32_t img) {                      // img = LDS address of the unit's first V row
        if (KK_DBG(a, 16)) return;
        const uint32_t a0 = img + va[0], a1 = img + va[1];
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(vlo[0]) : "v"(a0));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=v"(vhi[0]) : "v"(a0));
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(vlo[1]) : "v"(a1));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=v"(vhi[1]) : "v"(a1));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(vlo[2]) : "v"(a0));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:3072" : "=v"(vhi[2]) : "v"(a0));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(vlo[3]) : "v"(a1));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:3072" : "=v"(vhi[3]) : "v"(a1));
    };
    auto wait_lds = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };
    auto qk = [&](f32x16 &s) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(kf[ks]));
        zero_acc(s);
        if (KK_DBG(a, 4)) { s[0] = (float)kf[0][0] + (float)kf[1][1] + (float)kf[2][2] + (float)kf[3][3]; return; }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf.v[ks], s, 0, 0, 0);
    };
    f32x16 o[2];
    zero_acc(o[0]); zero_acc(o[1]);
    float m = -1e30f, l = 0.f;
    const float c2 = a.scale * 1.4426950408889634f;
    ProbDrop pd;
    pd.init(a, b, hh);
    // softmax (+ dropout) of one unit: s -> two B operands of the PV MFMAs; the arithmetic of attn_fwd_kernel
    auto softmax_unit = [&](const f32x16 &s, int kb, uint32_t kmsub, bf16x8 (&pb)[2]) {
        if (KK_DBG(a, 2)) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int j = 0; j < 8; ++j) pb[s2][j] = (__bf16)s[8 * s2 + j];
            return;
        }
        const bool edge = kb + 32 > a.Sk || (a.causal && kb + 31 > qmin) || kmsub != 0u;
        float p[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) p[r] = s[r];
        if (edge) {
            const uint32_t kml = kmsub >> (4 * half);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kb + frag_row(r, half);
                const bool ok = key < a.Sk && !(a.causal && key > q) && !((kml >> frag_row(r, 0)) & 1u);
                p[r] = ok ? p[r] : -INFINITY;
            }
        }
        float mx = p[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, p[r]);
        mx = xor32_max(mx) * c2;
        const float mn = fmaxf(m, mx);
        if (__ballot(mn > m) != 0ull) {
            const float alpha = __builtin_amdgcn_exp2f(m - mn);
            l *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
            m = mn;
        }
        float rs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { p[r] = __builtin_amdgcn_exp2f(fmaf(p[r], c2, -m)); rs += p[r]; }
        rs = xor32_sum(rs);
        l += rs;
        if (pd.thr) {
            const uint32_t xb = pd.row(q, kb + 4 * half);
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const uint32_t hsh = pd.hash(xb + (uint32_t)(frag_row(r, 0) >> 1));
                p[r] = pd.keep_lo(hsh) ? p[r] : 0.f;
                p[r + 1] = pd.keep_hi(hsh) ? p[r + 1] : 0.f;
            }
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int j = 0; j < 8; ++j) pb[s2][j] = (__bf16)p[8 * s2 + j];
    };
    auto pv = [&](const bf16x8 (&pb)[2]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(vlo[i]), "+v"(vhi[i]));
        if (KK_DBG(a, 4)) { o[0][0] += (float)pb[0][0] + (float)pb[1][0] + (float)vlo[0][0] + (float)vhi[3][0]; return; }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int db = 0; db < 2; ++db)
                o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_pair(vlo[s2 * 2 + db], vhi[s2 * 2 + db]), pb[s2], o[db], 0, 0, 0);
    };
    if (KK_DBG(a, 128)) return;                               // (timing probe: launch + DMA issue, nothing waited for)
    stamp(1);
    // ---- prologue: tiles 0 and 1 landed (tile 2 may stay in flight), unit 0's scores, unit 1's K fragments
    if (NS >= 4 && nt >= 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (nt >= 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    stamp(2);
    rowfrag_from_image(qf, qimg, wave * 32, l31, half);
    f32x16 sa, sb;
    if (nu > 0) {
        read_k(gl);
        wait_lds();
        qk(sa);
        if (nu > 1) read_k(gl + 4096);
    }
    stamp(3);
    int st = 0;
    for (int t = 0; t < (KK_DBG(a, 32) ? 0 : nt0); ++t) {
        const int st1 = st + 1 == NS ? 0 : st + 1;
        stamp(4 + 6 * t);
        if (t > 0) {
            // tile t+1 landed: the only DMA younger than it is tile t+2 when NS == 4
            if (NS >= 4 && t + 2 < nt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            stamp(5 + 6 * t);
            if (!KK_DBG(a, 8)) __builtin_amdgcn_s_barrier();                  // ... for every wave, and every wave is done with tile t-1
            asm volatile("" ::: "memory");
            stamp(6 + 6 * t);
            if (t + NS - 1 < nt && !KK_DBG(a, 1)) issue_tile(t + NS - 1, st == 0 ? NS - 1 : st - 1);
            stamp(7 + 6 * t);
        }
        const int u0 = 2 * t;
        if (u0 < nu) {
            const int k0 = kfirst + t * STEP;
            const uint64_t kmbits = km ? kmb[k0 >> 6] : 0ull;
            const uint32_t cur = gl + st * STAGE, nxt = gl + st1 * STAGE;
            bf16x8 pb[2];
            // unit (t, 0): scores in sa; next unit (t, 1) -> sb
            if (u0 + 1 < nu) { wait_lds(); qk(sb); }
            __builtin_amdgcn_sched_barrier(0);
            read_v(cur + KIMG);
            softmax_unit(sa, k0, (uint32_t)kmbits, pb);
            wait_lds();
            pv(pb);
            __builtin_amdgcn_sched_barrier(0);
            stamp(8 + 6 * t);
            if (u0 + 2 < nu) read_k(nxt);
            if (u0 + 1 < nu) {
                // unit (t, 1): scores in sb; next unit (t+1, 0) -> sa
                if (u0 + 2 < nu) { wait_lds(); qk(sa); }
                __builtin_amdgcn_sched_barrier(0);
                read_v(cur + KIMG + 4096);
                softmax_unit(sb, k0 + 32, (uint32_t)(kmbits >> 32), pb);
                wait_lds();
                pv(pb);
                __builtin_amdgcn_sched_barrier(0);
                stamp(9 + 6 * t);
                if (u0 + 3 < nu) read_k(nxt + 4096);
            }
        }
        st = st1;
    }
    stamp(58);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    stamp(59);
    {                                // merge the two key groups' partial softmaxes: group 1 -> LDS -> group 0
        float *mb = reinterpret_cast<float *>(smem_raw) + (wave * 64 + lane) * 34;
        if (grp == 1) {
            mb[0] = m; mb[1] = l;
#pragma unroll
            for (int r = 0; r < 16; ++r) { mb[2 + r] = o[0][r]; mb[18 + r] = o[1][r]; }
        }
        __syncthreads();
        if (grp == 1) return;
        const float m1 = mb[0], l1 = mb[1], mn = fmaxf(m, m1);
        const float a0 = __builtin_amdgcn_exp2f(m - mn), a1 = __builtin_amdgcn_exp2f(m1 - mn);
        l = l * a0 + l1 * a1;
        m = mn;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o[0][r] = o[0][r] * a0 + mb[2 + r] * a1; o[1][r] = o[1][r] * a0 + mb[18 + r] * a1; }
    }
    stamp(60);
    const float inv = l > 0.f ? pd.inv_keep / l : 0.f;
    store_rows_via_lds(static_cast<T *>(a.Out) + ((int64_t)b * a.Sq + qmin) * a.ldout + hh * 64, a.ldout, a.Sq - qmin, o, inv,
                       smem_raw + 36864 + wave * 4608, lane, a.wt);
    if (qvalid && half == 0) a.LSEo[((int64_t)b * a.heads + hh) * a.Sq + q] = l > 0.f ? (m + __builtin_amdgcn_logf(l)) * 0.6931471805599453f : INFINITY;
    stamp(61);
}

#ifdef KK_TUNING_HOOKS
// Probe bit 4096 (tools): every workgroup of a launch leaves (first wave's entry, last wave's exit) in 10 ns ticks of the constant clock
// and its hardware id at stamp-buffer word 512 + 4 * linear workgroup index — the launch's dispatch ramp, the spread of workgroup
// durations and its tail, next to the rocprofv3 duration (tools/probes/attn_grid_timeline.py).
struct KkWgStamp {
    const AttnArgs &a;
    __device__ unsigned long long *slot() const {
        return reinterpret_cast<unsigned long long *>(a.DeltaOut) + 512 + 4 * (blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z));
    }
    __device__ bool on() const { return KK_DBG(a, 4096) && a.DeltaOut != nullptr && (threadIdx.x & 63) == 0; }
    __device__ explicit KkWgStamp(const AttnArgs &a_) : a(a_) {
        if (on()) {
            atomicMin(slot(), (unsigned long long)__builtin_amdgcn_s_memrealtime());
            if (threadIdx.x == 0) slot()[2] = (unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | 20) << 32 | __builtin_amdgcn_s_getreg((31 << 11) | 4);   // XCC_ID | HW_ID
        }
    }
    __device__ ~KkWgStamp() {
        if (on()) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            atomicMax(slot() + 1, (unsigned long long)__builtin_amdgcn_s_memrealtime());
        }
    }
};
#define KK_WG_STAMP(args) KkWgStamp wg_stamp_(args)
#else
#define KK_WG_STAMP(args)
#endif

// ------------------------------------------------------------------ forward, third generation (bf16 storage): two workgroups per CU
// What the stamps of attn_fwd2 said (tools/probes/attn_trace.py): the loop is VALU-bound — a wave64 VALU instruction occupies its
// SIMD for 4 clocks, ~250 of them per 32 x 32 score unit against 8 MFMAs — and its two waves per SIMD, phase-locked by the per-tile
// barrier of the two key groups, keep the VALU ~60 % busy; one 112 KB workgroup fits a CU, so the 512 workgroups of a 1024-frame
// launch run as two rounds, each with its own prologue, group imbalance and merge.  Here a workgroup needs <= 74 KB of LDS and <= 128
// registers, so TWO are resident per CU (four waves per SIMD, from independent workgroups: no common barrier), and the work is dealt
// so that every wave does the same amount: the 8 waves are QW query waves x KG key slots over ONE shared K/V ring whose tile is
// KT = 32 KG keys — in a tile step wave (qw, kg) computes exactly one 32 x 32 unit (queries 32 qw.., keys 32 kg.. of the tile).  No
// software pipelining inside a wave (the other three waves of the SIMD are the overlap), Q fragments re-read from LDS per unit.
// Same arithmetic per unit as attn_fwd2 (softmax per 32-key unit, same dropout function): the results differ from it only by the
// order in which the key slots' partial softmaxes are merged.
#ifndef KK_QKV_AUX
#define KK_QKV_AUX 0            // cache policy of the Q / K / V DMA loads (kk_chain.hip: sc1 = L1 bypass for tensors written earlier in the launch)
#endif
// CHAIN (kk_chain.hip): the body as one phase of a persistent launch — block coordinates from the caller, every wave stays to the end
// (no early exit in front of a workgroup barrier) and leaves through a barrier that frees the LDS for the next unit.
// KRD (kk_attn_fwd_rb): the dropout keep decisions are READ — the unit's 16 lane masks by two scalar loads from the array that
// kk_attn_keep_gen filled beside the encoder forward — instead of hashed and stored: -9 vector instructions per two scores and the
// 32 v_writelane + the store of a unit.  Same bits, same arithmetic: the output equals the hashing launch's bit for bit.
template <int QW, int KG, int NS, bool CHAIN = false, bool KRD = false>
__device__ __forceinline__ void attn_fwd3_body(const AttnArgs &a, int chain_bx = 0, int chain_by = 0) {
    typedef __bf16 T;
    constexpr int QB = 32 * QW, KT = 32 * KG, KIMG = KT * 128, STAGE = 2 * KIMG, RING = NS * STAGE;
    constexpr int KP = KT / 64, NPT = 2 * KP;                 // 16-byte pieces per thread: per operand, per tile
    static_assert(QW * KG == 8 && (QB * 8) % 512 == 0, "eight waves; whole Q pieces per thread");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];        // [stage][K image | V image] | key-mask words | Q image
    uint64_t *kmb = reinterpret_cast<uint64_t *>(smem_raw + RING);         // [64] one word per 64 keys
    char *qimg = smem_raw + RING + 512;                                    // [QB queries][64] image
    int bx_, by_;
    if constexpr (CHAIN) { bx_ = chain_bx; by_ = chain_by; }
    else attn_block(a, bx_, by_, true);
    const int b = by_ / a.heads, hh = by_ % a.heads;
    const int qblk = bx_ * QB;
    const int lane = threadIdx.x & 63, wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), half = lane >> 5, l31 = lane & 31;
    const int qw = wave8 % QW, kg = wave8 / QW;
    const int qmin = qblk + 32 * qw, q = qmin + l31;
    const bool qvalid = q < a.Sq;
    int kend = a.Sk;
    if (a.causal && qblk + QB < kend) kend = qblk + QB;
    int klim = kend;                                           // this wave multiplies the keys [0, klim)
    if (a.causal && qmin + 32 < klim) klim = qmin + 32;
    const int nt = (kend + KT - 1) / KT;
    // ---- Q rows of the block (the oldest DMA: covered by every wait below)
    {
        const int nrows = a.Sq - qblk < QB ? a.Sq - qblk : QB;
        const T *qb = static_cast<const T *>(a.Q) + ((int64_t)b * a.Sq + qblk) * a.ldq + hh * 64;
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(qb), 0, nrows > 0 ? (int)((((int64_t)nrows - 1) * a.ldq + 64) * 2) : 0, 0x00020000);
#pragma unroll
        for (int j = 0; j < QB * 8 / 512; ++j) {
            const int p = threadIdx.x + 512 * j, row = p >> 3, pc = p & 7;
            const uint32_t vo = (uint32_t)(((int64_t)row * a.ldq + ((pc ^ ((row >> 1) & 7)) * 8)) * 2);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, KK_LDS_PTR(qimg + wave8 * 1024 + j * 8192), 16, vo, 0, 0, KK_QKV_AUX);
        }
    }
    const uint8_t *km = a.key_mask ? a.key_mask + (int64_t)b * a.Sk : nullptr;
    uint32_t kmv[8];
    if (km) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int key = (wave8 + 8 * i) * 64 + lane;
            kmv[i] = key < kend ? km[key] : 0u;
        }
    }
    // ---- K / V tiles: every thread issues KP pieces of each
    const T *Kb = static_cast<const T *>(a.K) + (int64_t)b * a.Sk * a.ldk + hh * 64;
    const T *Vb = static_cast<const T *>(a.V) + (int64_t)b * a.Sk * a.ldv + hh * 64;
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(Kb), 0, (int)((((int64_t)a.Sk - 1) * a.ldk + 64) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(Vb), 0, (int)((((int64_t)a.Sk - 1) * a.ldv + 64) * 2), 0x00020000);
    uint32_t kvo[KP], vvo[KP];
#pragma unroll
    for (int j = 0; j < KP; ++j) {
        const int p = threadIdx.x + 512 * j, row = p >> 3, pc = p & 7;
        kvo[j] = (uint32_t)(((int64_t)row * a.ldk + ((pc ^ ((row >> 1) & 7)) * 8)) * 2);
        const int sw = 2 * ((row >> 1) & 1), g = (((pc >> 1) ^ sw) << 1) | (pc & 1);
        vvo[j] = (uint32_t)(((int64_t)row * a.ldv + g * 8) * 2);
    }
    const uint32_t ktile = (uint32_t)(KT * a.ldk * 2), vtile = (uint32_t)(KT * a.ldv * 2);
    auto issue_tile = [&](int t, int st) {
        char *dst = smem_raw + st * STAGE + wave8 * 1024;
#pragma unroll
        for (int j = 0; j < KP; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, KK_LDS_PTR(dst + j * 8192), 16, kvo[j] + (uint32_t)t * ktile, 0, 0, KK_QKV_AUX);
#pragma unroll
        for (int j = 0; j < KP; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, KK_LDS_PTR(dst + KIMG + j * 8192), 16, vvo[j] + (uint32_t)t * vtile, 0, 0, KK_QKV_AUX);
    };
#pragma unroll
    for (int t = 0; t < NS - 1; ++t)
        if (t < nt) issue_tile(t, t);
    if (km) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint64_t bits = __ballot(kmv[i] != 0u);
            if (lane == 0) kmb[wave8 + 8 * i] = bits;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the words are in LDS before this wave reaches the loop's raw s_barrier
    }
    // ---- fragment addresses (bytes inside an image)
    const uint32_t sl = (uint32_t)(uintptr_t)KK_LDS_PTR(smem_raw);
    const uint32_t ql = (uint32_t)(uintptr_t)KK_LDS_PTR(qimg) + (uint32_t)((32 * qw + l31) * 128);
    uint32_t ka[4], va[2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) ka[ks] = (uint32_t)(l31 * 128 + (((2 * ks + half) ^ ((l31 >> 1) & 7)) * 16));
    {
        const int L = lane & 15, kq = L >> 2, gi = (lane >> 4) & 1, sw = 2 * ((kq >> 1) & 1);
#pragma unroll
        for (int db = 0; db < 2; ++db) va[db] = (uint32_t)((4 * half + kq) * 128 + (((2 * db + gi) ^ sw) * 32) + 8 * (L & 3));
    }
    f32x16 o[2];
    zero_acc(o[0]); zero_acc(o[1]);
    float m = -1e30f, l = 0.f;
    const float c2 = a.scale * 1.4426950408889634f;
    const int klast = a.causal ? min(a.Sk - 1, q) : a.Sk - 1;  // the last key this lane's query sees
    ProbDrop pd;
    pd.init(a, b, hh);
    // keep bits (AttnArgs::keep): the unit's 32 ballot dwords are gathered into lanes 0..31 of ONE register (v_writelane) and stored
    // at the top of the NEXT tile step, BEFORE that step's tile DMAs are issued: the counted vmcnt waits above stay exact — the only
    // operations younger than the tile a step waits for are the DMAs of the tile after it (CDNA4 counts stores in vmcnt too).
    const int nKU = (a.Sk + 31) >> 5;
    uint32_t *kbh = (!KRD && a.keep != nullptr && pd.thr != 0u && qmin < a.Sq)         // (a wave whose 32 rows lie beyond Sq has no unit row in the array)
        ? reinterpret_cast<uint32_t *>(static_cast<char *>(a.keep) + ((int64_t)(b * a.heads + hh) * ((a.Sq + 31) >> 5) + (qmin >> 5)) * nKU * 128) : nullptr;
    // KRD: this wave's row of units (wave-uniform address: scalar loads); rows beyond Sq read any valid row (their output is not stored)
    const kk_cu64x8 *keep_row = KRD ? reinterpret_cast<const kk_cu64x8 *>(reinterpret_cast<uintptr_t>(
        static_cast<const char *>(a.keep) + ((int64_t)(b * a.heads + hh) * ((a.Sq + 31) >> 5) + (qmin < a.Sq ? qmin >> 5 : 0)) * nKU * 128)) : nullptr;
    uint32_t kw_pend = 0u;
    uint32_t warm_sink = 0u;                                   // destination of the weight-warming loads (never read)
    int kw_unit = -1;                                          // (wave-uniform) key unit whose words are pending in kw_pend
    auto flush_keep = [&]() {
        if (kw_unit >= 0) {
            if (lane < 32) {                                      // (write-through like the launch's other outputs: 2 - 8 MB of bits per launch)
                if (a.wt) kk_st4_wt(kbh + kw_unit * 32 + lane, kw_pend);
                else kbh[kw_unit * 32 + lane] = kw_pend;
            }
            kw_unit = -1;
        }
        asm volatile("" ::: "memory");
    };
    for (int t = 0; t < nt; ++t) {
        // tile t landed once at most the younger tiles' DMAs are outstanding (tile t + NS - 1 goes out behind this step's barrier)
        const int younger = min(nt - 1 - t, NS - 2);
        if (NS >= 3 && younger >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPT) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                         // ... for every wave, and every wave is done with tile t - 1
        asm volatile("" ::: "memory");
        flush_keep();
        if (t + NS - 1 < nt) issue_tile(t + NS - 1, (t + NS - 1) % NS);
        if (t == nt - 1 && a.warm_bytes[0] != 0u) {            // (no DMA is issued after this point: only the final vmcnt(0) waits for these)
            const uint32_t lin = blockIdx.x + gridDim.x * blockIdx.y, ngrp = max((gridDim.x * gridDim.y) >> 3, 1u), j = lin >> 3;    // (a launch of fewer than 8 workgroups: one group)
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                const uint32_t lines = a.warm_bytes[w] >> 7, per = (lines + ngrp - 1) / ngrp;                 // (lines == 0: per == 0, no trip)
                for (uint32_t i = wave8 * 64 + lane; i < per; i += 512) {
                    const uint32_t line = j * per + i;
                    if (line < lines) {
                        const char *ptr = static_cast<const char *>(a.warm[w]) + ((size_t)line << 7);
                        asm volatile("global_load_dword %0, %1, off" : "=v"(warm_sink) : "v"(ptr) : "memory");
                    }
                }
            }
        }
        const int k0 = t * KT + 32 * kg;
        if (k0 >= klim) continue;                             // (causal: nothing of this unit is visible to these queries)
        const uint32_t kimg = sl + (uint32_t)((t % NS) * STAGE + kg * 4096), vimg = kimg + KIMG;
        // S^T = K . Q^T
        bf16x8 kf[4], qf[4];
        asm volatile("ds_read_b128 %0, %1" : "=v"(kf[0]) : "v"(kimg + ka[0]));
        asm volatile("ds_read_b128 %0, %1" : "=v"(qf[0]) : "v"(ql + (ka[0] & 127)));
        asm volatile("ds_read_b128 %0, %1" : "=v"(kf[1]) : "v"(kimg + ka[1]));
        asm volatile("ds_read_b128 %0, %1" : "=v"(qf[1]) : "v"(ql + (ka[1] & 127)));
        asm volatile("ds_read_b128 %0, %1" : "=v"(kf[2]) : "v"(kimg + ka[2]));
        asm volatile("ds_read_b128 %0, %1" : "=v"(qf[2]) : "v"(ql + (ka[2] & 127)));
        asm volatile("ds_read_b128 %0, %1" : "=v"(kf[3]) : "v"(kimg + ka[3]));
        asm volatile("ds_read_b128 %0, %1" : "=v"(qf[3]) : "v"(ql + (ka[3] & 127)));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(kf[ks]), "+v"(qf[ks]));
        f32x16 s;
        zero_acc(s);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[ks], s, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        // V^T fragments in flight under the softmax
        s16x4 vlo[4], vhi[4];
        {
            const uint32_t a0 = vimg + va[0], a1 = vimg + va[1];
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(vlo[0]) : "v"(a0));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=v"(vhi[0]) : "v"(a0));
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(vlo[1]) : "v"(a1));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=v"(vhi[1]) : "v"(a1));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(vlo[2]) : "v"(a0));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:3072" : "=v"(vhi[2]) : "v"(a0));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(vlo[3]) : "v"(a1));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:3072" : "=v"(vhi[3]) : "v"(a1));
        }
        // softmax (+ dropout) of the unit, in place: the arithmetic of attn_fwd2's softmax_unit
        const uint32_t kmsub = km ? (uint32_t)(kmb[k0 >> 6] >> (k0 & 32)) : 0u;
        const bool edge = k0 + 32 > a.Sk || (a.causal && k0 + 31 > qmin) || kmsub != 0u;
        if (edge) {                                            // bit c of the word = key c of the unit is visible: no branches (kk_low_bits)
            const uint32_t aw = (kk_low_bits(klast - k0) & ~kmsub) >> (4 * half);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = __builtin_amdgcn_sbfe((int)aw, frag_row(r, 0), 1);
                s[r] = kk_bfif(s[r], m, (int)0xff800000);     // (v_bfi_b32: s or -inf)
            }
        }
        float mx = s[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
        mx = xor32_max(mx) * c2;
        const float mn = fmaxf(m, mx);
        if (__ballot(mn > m) != 0ull) {
            const float alpha = __builtin_amdgcn_exp2f(m - mn);
            l *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
            m = mn;
        }
        float rs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = __builtin_amdgcn_exp2f(fmaf(s[r], c2, -m)); rs += s[r]; }
        rs = xor32_sum(rs);
        l += rs;
        if constexpr (KRD) {
            if (pd.thr) {
                const kk_u64x8 mk0 = keep_row[(k0 >> 5) * 2], mk1 = keep_row[(k0 >> 5) * 2 + 1];
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_inverse_ballot_w64(r < 8 ? mk0[r & 7] : mk1[r & 7]) ? s[r] : 0.f;
            }
        } else if (pd.thr) {
            const uint32_t xb = pd.row(q, k0 + 4 * half);
            uint64_t mk[16];                                   // the comparisons' lane masks = the unit's ballots
            uint32_t hprev = 0u;
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                // (probe 1024: one hash per FOUR scores — every second hash is a rotation of the one before: wrong masks, the instruction
                //  count of VERDICT r4 item 2b's 8-bit decisions)
                const uint32_t hsh = (KK_DBG(a, 1024) && (r & 2)) ? ((hprev >> 8) | (hprev << 24)) : pd.hash(xb + (uint32_t)(frag_row(r, 0) >> 1));
                hprev = hsh;
                mk[r] = __builtin_amdgcn_uicmp(hsh & 0xFFFFu, pd.thr, 35);          // >= : keep_lo
                mk[r + 1] = __builtin_amdgcn_uicmp(hsh >> 16, pd.thr, 35);          //      keep_hi
                s[r] = __builtin_amdgcn_inverse_ballot_w64(mk[r]) ? s[r] : 0.f;
                s[r + 1] = __builtin_amdgcn_inverse_ballot_w64(mk[r + 1]) ? s[r + 1] : 0.f;
            }
            if (kbh != nullptr) {
                uint32_t kw = 0u;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(kw) : "s"((uint32_t)mk[r]), "n"(2 * r));
                    asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(kw) : "s"((uint32_t)(mk[r] >> 32)), "n"(2 * r + 1));
                }
                kw_pend = kw;
                kw_unit = __builtin_amdgcn_readfirstlane(k0 >> 5);
            }
        }
        bf16x8 pb[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int j = 0; j < 8; ++j) pb[s2][j] = (__bf16)s[8 * s2 + j];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(vlo[i]), "+v"(vhi[i]));
        // O^T += V^T . P^T
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int db = 0; db < 2; ++db)
                o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_pair(vlo[s2 * 2 + db], vhi[s2 * 2 + db]), pb[s2], o[db], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    flush_keep();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" :: "v"(warm_sink));
    __syncthreads();                                           // the ring is free: the key slots' partial softmaxes meet in it
    {
        float *mb = reinterpret_cast<float *>(smem_raw);       // [(kg - 1) * QW + qw][lane][34]
        if (kg > 0) {
            float *w = mb + (((kg - 1) * QW + qw) * 64 + lane) * 34;
            w[0] = m; w[1] = l;
#pragma unroll
            for (int r = 0; r < 16; ++r) { w[2 + r] = o[0][r]; w[18 + r] = o[1][r]; }
        }
        __syncthreads();
        if constexpr (!CHAIN) { if (kg > 0) return; }
        if (kg == 0) {
#pragma unroll
        for (int g = 1; g < KG; ++g) {
            const float *w = mb + (((g - 1) * QW + qw) * 64 + lane) * 34;
            const float m1 = w[0], l1 = w[1], mn = fmaxf(m, m1);
            const float a0 = __builtin_amdgcn_exp2f(m - mn), a1 = __builtin_amdgcn_exp2f(m1 - mn);
            l = l * a0 + l1 * a1;
            m = mn;
#pragma unroll
            for (int r = 0; r < 16; ++r) { o[0][r] = o[0][r] * a0 + w[2 + r] * a1; o[1][r] = o[1][r] * a0 + w[18 + r] * a1; }
        }
        }
    }
    __syncthreads();                                           // (the QW remaining waves: everyone has read the merge area — the store tiles reuse it)
    static_assert(RING >= (KG - 1) * QW * 64 * 34 * 4 && RING >= QW * 4608, "merge area / store tiles fit the ring");
    if (kg == 0) {
        const float inv = l > 0.f ? pd.inv_keep / l : 0.f;
        store_rows_via_lds(static_cast<T *>(a.Out) + ((int64_t)b * a.Sq + qmin) * a.ldout + hh * 64, a.ldout, a.Sq - qmin, o, inv,
                           smem_raw + qw * 4608, lane, a.wt);
        if (qvalid && half == 0) a.LSEo[((int64_t)b * a.heads + hh) * a.Sq + q] = l > 0.f ? (m + __builtin_amdgcn_logf(l)) * 0.6931471805599453f : INFINITY;
    }
    if constexpr (CHAIN) __syncthreads();                      // the next unit of this workgroup reuses the ring, the mask words and the Q image
}

#ifdef KK_BODIES_ONLY
}  // namespace   (kk_chain.hip includes this file for attn_fwd3_body only)
#else
// (plain kernels around the template body: hipcc's host pass did not emit the stub of the kernel TEMPLATE named in kk_attn_fwd, and
// rejected its explicit instantiation — the same host-pass trouble as g16x_group_kernel in kk_gemm16x.hip)
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void attn_fwd3_q128_kernel(AttnArgs a) { KK_WG_STAMP(a); attn_fwd3_body<4, 2, 3>(a); }
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void attn_fwd3_q64_kernel(AttnArgs a) { KK_WG_STAMP(a); attn_fwd3_body<2, 4, 2>(a); }
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void attn_fwd3_q128r_kernel(AttnArgs a) { KK_WG_STAMP(a); attn_fwd3_body<4, 2, 3, false, true>(a); }
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void attn_fwd3_q64r_kernel(AttnArgs a) { KK_WG_STAMP(a); attn_fwd3_body<2, 4, 2, false, true>(a); }

// ------------------------------------------------------------------ keep-bit generator (kk_attn_keep_gen)
// The dropout keep decisions of up to 16 attention launches as ONE pure-vector launch (no LDS): per 32 x 32 unit the 512 hashes the
// forward evaluates for it, stored as the same 16 ballots in the same layout (AttnArgs::keep).
// Issued on the decoder-head stream beside the persistent encoder forward, which is latency-bound and leaves the vector ALUs idle.
struct KeepGenArgs {
    KkKeepSite s[16];
    int64_t start[17];               // first unit of each site in the flattened unit list
    int n;
    const uint32_t *seed;
    uint32_t seed_offset;            // the bits are those of seed value *seed + seed_offset (1: the NEXT micro-batch's, generated beside the optimizer pass)
};
// Lane-local on purpose: a lane owns one (unit, key pair) — 16 lanes per unit, four units per wave — walks the unit's 32 queries,
// and builds the two dwords of that key pair (even key, odd key) bit by bit: the same 512 hashes per unit the forward's 64 lanes
// evaluate, but no ballots, no v_writelane, nothing wave-wide.  (A first version mirrored the forward — compare masks moved into one
// register by v_writelane — and wrote stale words whenever it ran beside other kernels: the compares that produce those SGPRs sat
// right in front of the inline-asm v_writelanes, where the hazard recogniser does not look.)
__global__ __launch_bounds__(256) void attn_keep_gen_kernel(const KeepGenArgs g) {
    const int lane = threadIdx.x & 63, j = lane & 15, hbit = j & 1, rr = j >> 1;
    const int64_t nquads = (int64_t)gridDim.x * 4, total = g.start[g.n];
    const uint32_t seed = *g.seed + g.seed_offset;
    const int kp_off = ((rr & 1) + 4 * (rr >> 1));               // frag_row(2 rr, 0) >> 1: the key pair of accumulator registers 2 rr, 2 rr + 1
    for (int64_t u0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4; u0 < total; u0 += nquads * 4) {
        const int64_t u = u0 + (lane >> 4);
        if (u >= total) continue;
        int i = 0;
        while (i + 1 < g.n && u >= g.start[i + 1]) ++i;
        const KkKeepSite &st = g.s[i];
        const int nQU = (st.Sq + 31) >> 5, nKU = (st.Sk + 31) >> 5;
        const int64_t v = u - g.start[i];
        const int ku = (int)(v % nKU), qu = (int)((v / nKU) % nQU), bh = (int)(v / ((int64_t)nKU * nQU));
        if (st.causal && ku > qu) continue;                    // (the forward never visits a unit above the diagonal)
        uint32_t thr = (uint32_t)(st.p * 65536.f + 0.5f);
        thr = thr > 65535u ? 65535u : thr;
        const uint32_t key = kk_hash(seed, st.site, (uint64_t)bh), sk2 = (uint32_t)(st.Sk + 1) >> 1;
        // the forward's lane (l31 = query, half) hashes  q * sk2 + ((k0 + 4 half) >> 1) + (frag_row(r, 0) >> 1)  for r = 0, 2, ..., 14
        uint32_t x0 = (uint32_t)(qu * 32) * sk2 + ((uint32_t)(ku * 32 + 4 * hbit) >> 1) + (uint32_t)kp_off;
        uint32_t lo = 0u, hi = 0u;
// (unrolled by 2: 39 registers, so that TWO waves of this kernel fit a SIMD beside the persistent encoder's two 216-register waves — 432 + 2 x 40 = 512;
//  unrolled by 8 it held 60 and fit one: -0.2 % of the step at 8 x 512 and -0.7 % at 8 x 1024 for the same bits, profiles/r06_keep_bits_gen_ab.txt)
#ifndef KK_KEEPGEN_UNROLL
#define KK_KEEPGEN_UNROLL 2
#endif
#pragma unroll KK_KEEPGEN_UNROLL
        for (int q = 0; q < 32; ++q) {
            uint32_t x = x0 ^ key;
            x ^= x >> 16; x = __umul24(x, 0xb5352du); x ^= x >> 13; x = __umul24(x, 0xca68b5u); x ^= x >> 16;
            lo |= ((x & 0xFFFFu) >= thr ? 1u : 0u) << q;           // even key: register 2 rr
            hi |= ((x >> 16) >= thr ? 1u : 0u) << q;               // odd key:  register 2 rr + 1
            x0 += sk2;
        }
        uint32_t *unit = reinterpret_cast<uint32_t *>(static_cast<char *>(st.keep) + (((int64_t)bh * nQU + qu) * nKU + ku) * 128);
        unit[2 * (2 * rr) + hbit] = lo;                        // dword 2 r + half = the ballot half of register r
        unit[2 * (2 * rr + 1) + hbit] = hi;
    }
}

// ------------------------------------------------------------------ decode: one query per (batch, head)
// Sq == 1 (the incremental path of transformers.py:237-253: a decoder step against the KV cache / against the memory), no dropout.
// The tiled kernels above would run one live row of a 128-row block; here a workgroup is one (batch, head): 16 waves = 256 key groups
// x 4 lanes (16 of the 64 dims each), scores kept in LDS between the two passes (max, then exp / sum / P.V), fp32 throughout.
template <typename T>
__global__ __launch_bounds__(1024) void attn_decode_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float dsm[];          // [Sk] scores | [16 waves][64] partial outputs | [32] reductions
    float *sc = dsm, *part = dsm + ((a.Sk + 3) & ~3), *red = part + 16 * 64;
    const int b = blockIdx.x / a.heads, hh = blockIdx.x % a.heads;
    const int tid = threadIdx.x, kg = tid >> 2, dq = (tid & 3) * 16, lane = tid & 63, wave = tid >> 6;
    const T *Q = static_cast<const T *>(a.Q) + (int64_t)b * a.ldq + hh * 64 + dq;
    const T *Kb = static_cast<const T *>(a.K) + (int64_t)b * a.Sk * a.ldk + hh * 64 + dq;
    const T *Vb = static_cast<const T *>(a.V) + (int64_t)b * a.Sk * a.ldv + hh * 64 + dq;
    const uint8_t *km = a.key_mask ? a.key_mask + (int64_t)b * a.Sk : nullptr;
    float qv[16];
#pragma unroll
    for (int i = 0; i < 16; i += 4) {
        const float4 t = ldv4<T>(Q + i);
        qv[i] = t.x; qv[i + 1] = t.y; qv[i + 2] = t.z; qv[i + 3] = t.w;
    }
    const float c2 = a.scale * 1.4426950408889634f;
    float mx = -INFINITY;
#pragma unroll 2
    for (int j = kg; j < a.Sk; j += 256) {                                // 256 key groups x 4 lanes (16 of the 64 dims each)
        const bool masked = km && km[j];
        float d = 0.f;
        if (!masked) {
            const T *kr = Kb + (int64_t)j * a.ldk;
            float4 t[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) t[i] = ldv4<T>(kr + 4 * i);
#pragma unroll
            for (int i = 0; i < 4; ++i) d += qv[4 * i] * t[i].x + qv[4 * i + 1] * t[i].y + qv[4 * i + 2] * t[i].z + qv[4 * i + 3] * t[i].w;
        }
        d += __shfl_xor(d, 1, 64);
        d += __shfl_xor(d, 2, 64);
        d = masked ? -INFINITY : d * c2;
        if ((tid & 3) == 0) sc[j] = d;
        mx = fmaxf(mx, d);
    }
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    float m = red[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) m = fmaxf(m, red[w]);
    const float mm = fmaxf(m, -1e30f);
    float acc[16], l = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll 2
    for (int j = kg; j < a.Sk; j += 256) {
        const float s = sc[j];
        if (s == -INFINITY) continue;
        const T *vr = Vb + (int64_t)j * a.ldv;
        float4 t[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) t[i] = ldv4<T>(vr + 4 * i);
        const float pj = __builtin_amdgcn_exp2f(s - mm);
        l += pj;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[4 * i] += pj * t[i].x; acc[4 * i + 1] += pj * t[i].y; acc[4 * i + 2] += pj * t[i].z; acc[4 * i + 3] += pj * t[i].w;
        }
    }
    // the 16 key groups of a wave (lanes with the same dims: lane ^ 4, 8, 16, 32), then the 16 waves through LDS
#pragma unroll
    for (int i = 0; i < 16; ++i) {
#pragma unroll
        for (int o = 4; o < 64; o <<= 1) acc[i] += __shfl_xor(acc[i], o, 64);
    }
    if (lane < 4) {
#pragma unroll
        for (int i = 0; i < 16; ++i) part[wave * 64 + dq + i] = acc[i];
    }
    l = wave_sum(l) * 0.25f;                                              // (the 4 lanes of a key group hold the same p)
    if (lane == 0) red[16 + wave] = l;
    __syncthreads();
    if (tid < 64) {
        float lt = 0.f, o = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) { lt += red[16 + w]; o += part[w * 64 + tid]; }
        o = lt > 0.f ? o / lt : 0.f;
        T *out = static_cast<T *>(a.Out) + (int64_t)b * a.ldout + hh * 64 + tid;
        *out = (T)o;
        if (tid == 0) a.LSEo[(int64_t)b * a.heads + hh] = lt > 0.f ? (mm + __builtin_amdgcn_logf(lt)) * 0.6931471805599453f : INFINITY;
    }
}

// ------------------------------------------------------------------ backward: dQ
// Same decomposition as the forward: a lane owns a query, wave group g sweeps the key tiles g, g+G, ...; with G = 2
// group 1's partial dQ is added to group 0's through LDS at the end.  Two register sets (prefetch distance 2).
template <bool BF16, bool ST16, int G>
__global__ __launch_bounds__(256 * G) void attn_bwd_dq_kernel(AttnArgs a) {
    using elem = typename ACfg<BF16>::elem;
    using SG = Stage<BF16, ST16>;
    using T = typename SG::T;
    constexpr int LR = ACfg<BF16>::LR, TILE = 64 * LR, NT = BF16 ? 3 : 2;   // K, V (+ K transposed for bf16)
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];        // [buffer][group][NT tiles]
    elem *smem = reinterpret_cast<elem *>(smem_raw);
    int bx_, by_;
    attn_block(a, bx_, by_, true);                         // (causal: blocks near the end of the sequence see the most keys)
    const int b = by_ / a.heads, hh = by_ % a.heads;
    const int qblk = bx_ * 128;
    const int lane = threadIdx.x & 63, wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), half = lane >> 5, l31 = lane & 31;
    const int wave = wave8 & 3, grp = wave8 >> 2;
    const int q = qblk + wave * 32 + l31;
    const bool qvalid = q < a.Sq;
    RowFrag<BF16> qf, dof;
    load_rowfrag<BF16, T>(qf, qvalid ? static_cast<const T *>(a.Q) + ((int64_t)b * a.Sq + q) * a.ldq + hh * 64 : nullptr, half);
    load_rowfrag<BF16, T>(dof, qvalid ? static_cast<const T *>(a.dO) + ((int64_t)b * a.Sq + q) * a.lddo + hh * 64 : nullptr, half);
    float dlt;
    if (a.DeltaOut) {       // Delta[b,head,q] = sum_d dO*O from the two row fragments already at hand (saves kk_attn_delta)
        RowFrag<BF16> of;
        load_rowfrag<BF16, T>(of, qvalid ? static_cast<const T *>(a.O) + ((int64_t)b * a.Sq + q) * a.ldo + hh * 64 : nullptr, half);
        dlt = rowfrag_dot<BF16>(dof, of);
        dlt += __shfl_xor(dlt, 32, 64);
        if (qvalid && half == 0 && grp == 0) a.DeltaOut[((int64_t)b * a.heads + hh) * a.Sq + q] = dlt;
    } else {
        dlt = qvalid ? a.Delta[((int64_t)b * a.heads + hh) * a.Sq + q] : 0.f;
    }
    ProbDrop pd;
    pd.init(a, b, hh);
    if (pd.thr) scale_rowfrag<BF16>(dof, pd.inv_keep);     // dP of a kept element carries 1/(1-p): fold it into dO once
    const float c2 = a.scale * 1.4426950408889634f;
    const float lse2 = qvalid ? a.LSE[((int64_t)b * a.heads + hh) * a.Sq + q] * 1.4426950408889634f : INFINITY;   // log2 domain
    f32x16 dq[2];
    zero_acc(dq[0]); zero_acc(dq[1]);
    const uint8_t *km = a.key_mask ? a.key_mask + (int64_t)b * a.Sk : nullptr;
    const int qmin = qblk + wave * 32;
    int kend = a.Sk;
    if (a.causal && qblk + 128 < kend) kend = qblk + 128;
    const T *Kb = static_cast<const T *>(a.K) + (int64_t)b * a.Sk * a.ldk + hh * 64;
    const T *Vb = static_cast<const T *>(a.V) + (int64_t)b * a.Sk * a.ldv + hh * 64;
    struct Regs {
        typename SG::R rk, rv;
        typename SG::RT rkt;
        uint32_t rkm;
    };
    Regs ra, rb;
    ra.rkm = rb.rkm = 0;
    auto issue = [&](Regs &t, int k0) {
        const int nvalid = a.Sk - k0 < 64 ? a.Sk - k0 : 64;
        load_rows(t.rk, Kb + (int64_t)k0 * a.ldk, a.ldk, nvalid);
        load_rows(t.rv, Vb + (int64_t)k0 * a.ldv, a.ldv, nvalid);
        if constexpr (BF16) load_rows_T(t.rkt, Kb + (int64_t)k0 * a.ldk, a.ldk, nvalid);
        t.rkm = km ? (lane < nvalid ? km[k0 + lane] : 0u) : 0u;
    };
    auto commit = [&](const Regs &t, int buf) {
        elem *dst = smem + (buf * G + grp) * NT * TILE;
        SG::st(dst, t.rk);
        SG::st(dst + TILE, t.rv);
        if constexpr (BF16) SG::stT(dst + 2 * TILE, t.rkt);
    };
    constexpr int STEP = 64 * G;
    const int kfirst = grp * 64;
    if (kfirst < kend) {
        issue(ra, kfirst);
        commit(ra, 0);
    }
    uint64_t kmbits = __ballot(ra.rkm != 0u), kmnext = 0;
    if (kfirst + STEP < kend) issue(ra, kfirst + STEP);
    if (kfirst + 2 * STEP < kend) issue(rb, kfirst + 2 * STEP);
    __syncthreads();
    int cur = 0;
    auto tile_step = [&](Regs &X, int kk0) {
        const int k0 = kk0 + kfirst;
        const elem *Ks = smem + (cur * G + grp) * NT * TILE, *Vs = Ks + TILE, *Kt = BF16 ? Ks + 2 * TILE : Ks;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int kb = k0 + sub * 32;
            if (kb >= kend) continue;
            if (a.causal && kb > qmin + 31) continue;
            f32x16 s, dp;
            zero_acc(s); zero_acc(dp);
            mma_tile_x_frag<BF16>(s, Ks, sub * 32, qf, l31, half);
            mma_tile_x_frag<BF16>(dp, Vs, sub * 32, dof, l31, half);
            const uint32_t kmsub = (uint32_t)(kmbits >> (sub * 32));
            const bool edge = kb + 32 > a.Sk || (a.causal && kb + 31 > qmin) || kmsub != 0u;
            float pv[16], ds[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) pv[r] = __builtin_amdgcn_exp2f(s[r] * c2 - lse2);
            if (edge) {
                const uint32_t kml = kmsub >> (4 * half);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kb + frag_row(r, half);
                    const bool ok = key < a.Sk && !(a.causal && key > q) && !((kml >> frag_row(r, 0)) & 1u);
                    pv[r] = ok ? pv[r] : 0.f;
                }
            }
            if (pd.thr) {
                const uint32_t xb = pd.row(q, kb + 4 * half);
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const uint32_t hsh = pd.hash(xb + (uint32_t)(frag_row(r, 0) >> 1));
                    ds[r] = pv[r] * ((pd.keep_lo(hsh) ? dp[r] : 0.f) - dlt);
                    ds[r + 1] = pv[r + 1] * ((pd.keep_hi(hsh) ? dp[r + 1] : 0.f) - dlt);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) ds[r] = pv[r] * (dp[r] - dlt);
            }
            mma_T_x_p<BF16>(dq, Kt, sub * 32, ds, l31, half);     // (the softmax scale is applied once, at the store)
        }
        kmnext = 0;
        if (k0 + STEP < kend) {
            commit(X, cur ^ 1);
            kmnext = __ballot(X.rkm != 0u);
            if (k0 + 3 * STEP < kend) issue(X, k0 + 3 * STEP);
        }
        kmbits = kmnext;
        __syncthreads();
        cur ^= 1;
    };
    for (int kk0 = 0; kk0 < (KK_DBG(a, 32) ? 0 : kend); kk0 += 2 * STEP) {      // the bound is the same for both groups (barriers)
        tile_step(ra, kk0);
        if (kk0 + STEP < kend) tile_step(rb, kk0 + STEP);
    }
    if constexpr (G == 2) {          // group 1's partial dQ -> LDS -> group 0
        float *mb = reinterpret_cast<float *>(smem_raw) + (wave * 64 + lane) * 33;
        if (grp == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { mb[r] = dq[0][r]; mb[16 + r] = dq[1][r]; }
        }
        __syncthreads();
        if (grp == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { dq[0][r] += mb[r]; dq[1][r] += mb[16 + r]; }
        }
    }
    T *out_row = static_cast<T *>(a.Out) + ((int64_t)b * a.Sq + q) * a.ldout + hh * 64;
    if (a.hn[0].raw == nullptr) {
        if (qvalid && grp == 0) store_row<T>(out_row, dq, a.scale, half);
        return;
    }
    float *colred = reinterpret_cast<float *>(smem_raw);          // [128 rows][65]
    __syncthreads();                                              // the staging tiles / merge buffer are free
    if (grp == 0)
        hn_bwd_row<T>(dq, a.scale, qvalid, static_cast<const T *>(a.hn[0].raw) + ((int64_t)b * a.Sq + q) * a.hn[0].ldraw + hh * 64,
                      out_row, a.hn[0], q, half, colred + (wave * 32 + l31) * 65);
    __syncthreads();
    hn_colsum(colred, a.hn[0].partials);
}

// ------------------------------------------------------------------ backward, second generation: shared pieces
// XOR value (on the 32-byte block index of a 128-byte row) of an image that is read BOTH as row fragments (ds_read_b128, the
// lane's own row) and through ds_read_b64_tr_b16 (four consecutive rows per 16-lane group): rows r and r+2 of a transpose
// read must differ in bit 1 of the block index (conflict-free), and the four values spread the row-fragment reads (2-way).
__device__ __forceinline__ int kk_xb(int r) { return (((r >> 1) & 1) << 1) | ((r >> 2) & 1); }

// Head-norm (+ RoPE) backward of the (row, head) vector this lane pair holds — hn_bwd_row with every operand in LDS: the
// raw projection tile and the RoPE rows were DMA'd there while the main loop ran, so the epilogue has no exposed global
// latency and no row-per-lane requests.  rawimg: [128][64] bf16, cosimg / sinimg: columns 0..31 of the table rows as
// [128][32] fp32 (rotate-half RoPE tables have identical halves, positional_encoding.py:129-150), all with the chunk XOR
// of dma_rows128.  The gradient of the raw projection comes back in the accumulator layout (out), for store_rows_via_lds.
// (core: the per-column contributions to the gain gradient come back in cr[32], accumulator order, for a caller whose colred buffer
//  shares LDS with the images and can only be written behind a barrier)
// the lane's 32 gain values (its columns db * 32 + 8 g + 4 half + e), fetched EARLY by the third-generation epilogues: eight dependent
// global loads in the middle of the row arithmetic were ~1 us of each head-norm epilogue
struct HnGain { float4 g4[8]; };
__device__ __forceinline__ HnGain hn_load_gain(const float *gain, int half) {
    HnGain r;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) r.g4[db * 4 + g] = ld4(gain + db * 32 + 8 * g + 4 * half);
    return r;
}
// (round 6: written on PAIRS — v_pk_mul_f32 / v_pk_fma_f32 on adjacent accumulator elements, one v_cvt_pk_bf16_f32 per two values, the
//  row mask on the packed word.  The compiler's own version of the scalar source was 737 vector instructions per call, a third of them
//  v_mov / v_cndmask to marshal pairs it had picked across the two halves of a row; these launches are bound by instruction issue —
//  profiles/r06_attn_pair_balance.txt — so the epilogues cost what they count.  Sums are taken pairwise: (even elements) + (odd elements).)
#ifndef KK_HN_CORE_V1
typedef float f32x2_ __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_ kk_unpack_bf16x2(uint32_t w) { return f32x2_{__uint_as_float(w << 16), __uint_as_float(w & 0xFFFF0000u)}; }
__device__ __forceinline__ void hn_bwd_row2_core(const f32x16 (&acc)[2], float mul, bool valid, const char *rawimg, const char *cosimg,
                                                 const char *sinimg, int row, bool rope, const HnGain &gn, int half, float (&cr)[32],
                                                 f32x16 (&out)[2]) {
    f32x2_ dn[16], v[16];                  // pair 8 db + 2 g + e2 = elements 4 g + 2 e2, + 1 of accumulator block db
    asm volatile("" : "+v"(row));          // (or the image addresses below are computed in the prologue and spilled across the main loop)
    const int swz = (row >> 1) & 7;
    const uint32_t vm = valid ? 0xFFFFFFFFu : 0u;
    const f32x2_ mul2 = {mul, mul};
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const kk_u32x2 w = *reinterpret_cast<const kk_u32x2 *>(rawimg + row * 128 + (((4 * db + g) ^ swz) * 16) + half * 8);
#pragma unroll
            for (int e2 = 0; e2 < 2; ++e2) {
                v[db * 8 + 2 * g + e2] = kk_unpack_bf16x2(w[e2]);
                const f32x2_ a2 = f32x2_{acc[db][4 * g + 2 * e2], acc[db][4 * g + 2 * e2 + 1]} * mul2;
                const uint32_t pk = __builtin_bit_cast(uint32_t, __builtin_convertvector(a2, bf16x2_)) & vm;      // the bf16 the consumer of this gradient sees
                dn[db * 8 + 2 * g + e2] = kk_unpack_bf16x2(pk);
            }
        }
    f32x2_ sq2 = v[0] * v[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) sq2 = __builtin_elementwise_fma(v[i], v[i], sq2);
    const float ssq = xor32_sum(sq2[0] + sq2[1]);
    const float rs = 1.f / sqrtf(ssq * (1.f / 64.f) + 1.1920928955078125e-7f);
    if (rope) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4_ c4 = *reinterpret_cast<const f32x4_ *>(cosimg + row * 128 + (((2 * g + half) ^ swz) * 16));
            const f32x4_ s4 = *reinterpret_cast<const f32x4_ *>(sinimg + row * 128 + (((2 * g + half) ^ swz) * 16));
#pragma unroll
            for (int e2 = 0; e2 < 2; ++e2) {
                const f32x2_ cc = {c4[2 * e2], c4[2 * e2 + 1]}, ss = {s4[2 * e2], s4[2 * e2 + 1]};
                const f32x2_ lo = dn[2 * g + e2], hi = dn[8 + 2 * g + e2];
                dn[2 * g + e2] = __builtin_elementwise_fma(lo, cc, hi * ss);
                dn[8 + 2 * g + e2] = __builtin_elementwise_fma(hi, cc, -(lo * ss));
            }
        }
    }
    const f32x2_ rs2 = {rs, rs};
    f32x2_ kd2 = {0.f, 0.f};
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 g4 = gn.g4[db * 4 + g];
            const f32x2_ gg[2] = {{g4.x, g4.y}, {g4.z, g4.w}};
#pragma unroll
            for (int e2 = 0; e2 < 2; ++e2) {
                const int i = db * 8 + 2 * g + e2;
                const f32x2_ c2 = (dn[i] * v[i]) * rs2;
                cr[2 * i] = c2[0];
                cr[2 * i + 1] = c2[1];
                dn[i] *= gg[e2];
                kd2 = __builtin_elementwise_fma(dn[i], v[i], kd2);
            }
        }
    const float kdot = xor32_sum(kd2[0] + kd2[1]);
    const float k = kdot * (1.f / 64.f) * rs * rs * rs;
    const f32x2_ k2 = {k, k};
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const f32x2_ o2 = __builtin_elementwise_fma(rs2, dn[i], -(v[i] * k2));
        out[i >> 3][2 * (i & 7)] = o2[0];
        out[i >> 3][2 * (i & 7) + 1] = o2[1];
    }
}
#else
__device__ __forceinline__ void hn_bwd_row2_core(const f32x16 (&acc)[2], float mul, bool valid, const char *rawimg, const char *cosimg,
                                                 const char *sinimg, int row, bool rope, const HnGain &gn, int half, float (&cr)[32],
                                                 f32x16 (&out)[2]) {
    float dn[32], v[32];
    asm volatile("" : "+v"(row));          // (or the image addresses below are computed in the prologue and spilled across the main loop)
    const int swz = (row >> 1) & 7;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const bf16x4 x4 = *reinterpret_cast<const bf16x4 *>(rawimg + row * 128 + (((4 * db + g) ^ swz) * 16) + half * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[db * 16 + 4 * g + e] = (float)x4[e];
                dn[db * 16 + 4 * g + e] = valid ? (float)(__bf16)(acc[db][4 * g + e] * mul) : 0.f;
            }
        }
    float ssq = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) ssq += v[i] * v[i];
    ssq = xor32_sum(ssq);
    const float rs = 1.f / sqrtf(ssq * (1.f / 64.f) + 1.1920928955078125e-7f);
    if (rope) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 c4 = *reinterpret_cast<const float4 *>(cosimg + row * 128 + (((2 * g + half) ^ swz) * 16));
            const float4 s4 = *reinterpret_cast<const float4 *>(sinimg + row * 128 + (((2 * g + half) ^ swz) * 16));
            const float cc[4] = {c4.x, c4.y, c4.z, c4.w}, ss[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float lo = dn[4 * g + e], hi = dn[16 + 4 * g + e];
                dn[4 * g + e] = lo * cc[e] + hi * ss[e];
                dn[16 + 4 * g + e] = hi * cc[e] - lo * ss[e];
            }
        }
    }
    float kdot = 0.f;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 g4 = gn.g4[db * 4 + g];
            const float gg[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = db * 16 + 4 * g + e;
                cr[i] = dn[i] * v[i] * rs;
                dn[i] *= gg[e];
                kdot += dn[i] * v[i];
            }
        }
    kdot = xor32_sum(kdot);
    const float k = kdot * (1.f / 64.f) * rs * rs * rs;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[db][r] = rs * dn[db * 16 + r] - v[db * 16 + r] * k;
}
#endif
__device__ __forceinline__ void hn_colred_store(const float (&cr)[32], int half, float *colred_row) {
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) colred_row[db * 32 + 8 * g + 4 * half + e] = cr[db * 16 + 4 * g + e];
}
// Column sums of cr over the 32 rows a half-wave holds (lane = row), in registers: a transposing butterfly — at the step with partner
// mask m a lane keeps the half of its live values whose index bit matches its own lane bit and hands the other half to its partner: 31
// exchange-adds.  The partner masks are taken in the order 8, 2, 1, 16, 4, so that the three big steps (16 + 8 + 4 exchanges) are DPP
// operand modifiers (row_ror:8, quad_perm) and only the last 2 + 1 are ds_bpermute round trips.  Afterwards lane l31 holds the sum over
// the 32 rows of cr[i], i = hn_colsum32_idx(l31): index bit 4 <- lane bit 3, 3 <- 1, 2 <- 0, 1 <- 4, 0 <- 2.
// Replaces, per head-norm epilogue, 32 LDS stores per lane into colred[128][65], two workgroup barriers and ONE wave adding up 128 rows.
template <int N, int MASK> __device__ __forceinline__ void hn_colsum_step(float (&v)[32], int l31) {      // N live values
    const bool up = (l31 & MASK) != 0;
#pragma unroll
    for (int j = 0; j < N / 2; ++j) {
        const float keep = up ? v[j + N / 2] : v[j], send = up ? v[j] : v[j + N / 2];
        float got;
        if constexpr (MASK == 8) got = kk_dpp<0x128>(send);
        else if constexpr (MASK == 2) got = kk_dpp<0x4E>(send);
        else if constexpr (MASK == 1) got = kk_dpp<0xB1>(send);
        else got = __shfl_xor(send, MASK, 64);
        v[j] = keep + got;
    }
}
__device__ __forceinline__ float hn_colsum32(float (&v)[32], int l31) {
    hn_colsum_step<32, 8>(v, l31);
    hn_colsum_step<16, 2>(v, l31);
    hn_colsum_step<8, 1>(v, l31);
    hn_colsum_step<4, 16>(v, l31);
    hn_colsum_step<2, 4>(v, l31);
    return v[0];
}
__device__ __forceinline__ int hn_colsum32_col(int l31, int half) {
    const int i = ((l31 >> 3) & 1) << 4 | ((l31 >> 1) & 1) << 3 | (l31 & 1) << 2 | ((l31 >> 4) & 1) << 1 | ((l31 >> 2) & 1);
    return (i >> 4) * 32 + 8 * ((i >> 2) & 3) + 4 * half + (i & 3);
}
// store_rows_via_lds through a tile of 16 rows (2304 bytes), two halves one after the other: fits the wave's OWN 4 KB of a dead
// 128-row image, so no workgroup barrier stands between the head-norm arithmetic and the stores.
__device__ __forceinline__ void store_rows_via_lds16(__bf16 *dst_row0, int64_t ld, int nvalid, const f32x16 (&acc)[2], float mul,
                                                     char *tile, int lane, int wt) {
    const int l31 = lane & 31, half = lane >> 5;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if ((l31 >> 4) == h) {
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    bf16x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (__bf16)(acc[db][4 * g + e] * mul);
                    *reinterpret_cast<bf16x4 *>(tile + (l31 & 15) * 144 + (db * 32 + 8 * g + 4 * half) * 2) = v;
                }
        }
        __builtin_amdgcn_wave_barrier();                       // (one wave: its LDS operations complete in order)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = (lane >> 3) + 8 * j, c = lane & 7;
            const u32x4 v = *reinterpret_cast<const u32x4 *>(tile + row * 144 + c * 16);
            if (16 * h + row < nvalid) kk_store16(dst_row0 + (int64_t)(16 * h + row) * ld + c * 8, v, wt);
        }
        __builtin_amdgcn_wave_barrier();
    }
}
__device__ __forceinline__ void hn_bwd_row2(const f32x16 (&acc)[2], float mul, bool valid, const char *rawimg, const char *cosimg,
                                            const char *sinimg, int row, bool rope, const float *gain, int half, float *colred_row,
                                            f32x16 (&out)[2]) {
    float cr[32];
    hn_bwd_row2_core(acc, mul, valid, rawimg, cosimg, sinimg, row, rope, hn_load_gain(gain, half), half, cr, out);
    hn_colred_store(cr, half, colred_row);
}
// the three epilogue images of a 128-row block (rows row0 .. of a sequence of S rows, position = row): raw | cos | sin
template <int NT = 512, typename HN> __device__ __forceinline__ void hn_dma_inputs(HN &h, int64_t seq_row0, int pos0, int nrows, int hh, char *img, int wave8) {
    dma_rows128<NT>(static_cast<const __bf16 *>(h.raw) + seq_row0 * h.ldraw + hh * 64, h.ldraw, nrows, img, wave8);
    if (h.rope) {       // (fp32 rows of 64 = 128 bf16-sized elements; the first 128 bytes of each)
        dma_rows128<NT>(reinterpret_cast<const __bf16 *>(h.cos_t + (int64_t)pos0 * 64), 128, nrows, img + 16384, wave8);
        dma_rows128<NT>(reinterpret_cast<const __bf16 *>(h.sin_t + (int64_t)pos0 * 64), 128, nrows, img + 32768, wave8);
    }
}

// ------------------------------------------------------------------ backward: dQ, second generation (bf16 storage)
// attn_bwd_dq_kernel<true, true, 2>'s arithmetic in attn_fwd2_kernel's structure: K / V tiles by DMA (K once, in an image
// that serves both the row fragments of S = K.Q^T and the transpose reads of dQ^T += K^T.dS^T), Q / dO / O rows by DMA (Delta
// from the fragments), the scores and dP of unit u+1 issued before the exponentials of unit u, the head-norm epilogue's
// operands prefetched into the prologue's LDS while the loop runs, 16-byte coalesced stores.
// (body of the dQ kernel: kk_attn_bwd_dq2.inc, included into attn_bwd_dq2_kernel and attn_bwd_pair2_kernel below)

// ------------------------------------------------------------------ backward: dK, dV
// A lane owns a key; the workgroup sweeps the query tiles.  G = 2: two wave groups take alternate query tiles of the
// same 128 keys (2 waves per SIMD, see attn_fwd_kernel) and group 1's dK / dV partial sums are added to group 0's
// through LDS at the end.  Staging: with G = 1 two register sets alternate and a tile's loads have two tile-times to land;
// with G = 2 the 256-register budget of 8 waves leaves room for one set (distance 1) — the second wave hides the rest.
template <bool BF16, bool ST16, int G>
__global__ __launch_bounds__(256 * G) void attn_bwd_dkv_kernel(AttnArgs a) {
    using elem = typename ACfg<BF16>::elem;
    using SG = Stage<BF16, ST16>;
    using T = typename SG::T;
    constexpr int LR = ACfg<BF16>::LR, TILE = 64 * LR, NT = BF16 ? 4 : 2;   // Q, dO (+ both transposed for bf16)
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];        // [buffer][group][NT tiles], then lse/delta rows
    elem *smem = reinterpret_cast<elem *>(smem_raw);
    float *stat = reinterpret_cast<float *>(smem_raw + (size_t)2 * G * NT * TILE * sizeof(elem));   // [buffer][group][2][64]
    int bx_, by_;
    attn_block(a, bx_, by_, false);                        // (causal: the first key blocks see the most queries)
    const int b = by_ / a.heads, hh = by_ % a.heads;
    const int kblk = bx_ * 128;
    const int lane = threadIdx.x & 63, wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), half = lane >> 5, l31 = lane & 31;
    const int wave = wave8 & 3, grp = wave8 >> 2, tl = threadIdx.x & 255;
    const int key = kblk + wave * 32 + l31;
    const bool kvalid = key < a.Sk;
    const bool kalive = kvalid && !(a.key_mask && a.key_mask[(int64_t)b * a.Sk + key]);
    RowFrag<BF16> kf, vf;
    load_rowfrag<BF16, T>(kf, kvalid ? static_cast<const T *>(a.K) + ((int64_t)b * a.Sk + key) * a.ldk + hh * 64 : nullptr, half);
    load_rowfrag<BF16, T>(vf, kvalid ? static_cast<const T *>(a.V) + ((int64_t)b * a.Sk + key) * a.ldv + hh * 64 : nullptr, half);
    f32x16 dk[2], dv[2];
    zero_acc(dk[0]); zero_acc(dk[1]); zero_acc(dv[0]); zero_acc(dv[1]);
    ProbDrop pd;
    pd.init(a, b, hh);
    if (pd.thr) scale_rowfrag<BF16>(vf, pd.inv_keep);       // dP = dO.V of a kept element carries 1/(1-p)
    const float c2 = a.scale * 1.4426950408889634f;
    const int kmaxw = kblk + wave * 32 + 31;                           // largest key of this wave
    const bool anydead = __ballot(!kalive) != 0ull;                    // masked / out-of-range keys in this wave
    const int qstart = a.causal ? (kblk / 64) * 64 : 0;
    const T *Qb = static_cast<const T *>(a.Q) + (int64_t)b * a.Sq * a.ldq + hh * 64;
    const T *dOb = static_cast<const T *>(a.dO) + (int64_t)b * a.Sq * a.lddo + hh * 64;
    const float *LSEb = a.LSE + ((int64_t)b * a.heads + hh) * a.Sq, *DLb = a.Delta + ((int64_t)b * a.heads + hh) * a.Sq;
    struct Regs {
        typename SG::R rq, rdo;
        typename SG::RT rqt, rdot;
        float lse, dlt;
    };
    constexpr int DIST = G == 1 ? 2 : 1;                   // prefetch distance in tiles (= register sets)
    Regs ra;
    typename std::conditional<DIST == 2, Regs, int>::type rb_store;
    Regs &rb = [&]() -> Regs & { if constexpr (DIST == 2) return rb_store; else return ra; }();
    auto issue = [&](Regs &t, int q0) {
        const int nvalid = a.Sq - q0 < 64 ? a.Sq - q0 : 64;
        load_rows(t.rq, Qb + (int64_t)q0 * a.ldq, a.ldq, nvalid);
        load_rows(t.rdo, dOb + (int64_t)q0 * a.lddo, a.lddo, nvalid);
        if constexpr (BF16) {
            load_rows_T(t.rqt, Qb + (int64_t)q0 * a.ldq, a.ldq, nvalid);
            load_rows_T(t.rdot, dOb + (int64_t)q0 * a.lddo, a.lddo, nvalid);
        }
        if (tl < 64) {
            const int qq = q0 + tl;
            t.lse = qq < a.Sq ? LSEb[qq] * 1.4426950408889634f : INFINITY;      // log2 domain
            t.dlt = qq < a.Sq ? DLb[qq] : 0.f;
        }
    };
    auto commit = [&](const Regs &t, int buf) {
        elem *dst = smem + (buf * G + grp) * NT * TILE;
        SG::st(dst, t.rq);
        SG::st(dst + TILE, t.rdo);
        if constexpr (BF16) {
            SG::stT(dst + 2 * TILE, t.rqt);
            SG::stT(dst + 3 * TILE, t.rdot);
        }
        if (tl < 64) {
            float *st = stat + (buf * G + grp) * 128;
            st[tl] = t.lse;
            st[64 + tl] = t.dlt;
        }
    };
    constexpr int STEP = 64 * G;
    const int qfirst = qstart + grp * 64;
    if (qfirst < a.Sq) {
        issue(ra, qfirst);
        commit(ra, 0);
    }
    if (qfirst + STEP < a.Sq) issue(ra, qfirst + STEP);
    if (DIST == 2 && qfirst + 2 * STEP < a.Sq) issue(rb, qfirst + 2 * STEP);
    __syncthreads();
    int cur = 0;
    auto tile_step = [&](Regs &X, int qq0) {
        const int q0 = qq0 + grp * 64;
        const elem *Qs = smem + (cur * G + grp) * NT * TILE, *dOs = Qs + TILE;
        const elem *Qt = BF16 ? Qs + 2 * TILE : Qs, *dOt = BF16 ? Qs + 3 * TILE : dOs;
        const float *lse_t = stat + (cur * G + grp) * 128, *dlt_t = lse_t + 64;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int qb = q0 + sub * 32;
            if (qb >= a.Sq) continue;
            if (a.causal && qb + 31 < kblk + wave * 32) continue;
            f32x16 s, dp;
            zero_acc(s); zero_acc(dp);
            mma_tile_x_frag<BF16>(s, Qs, sub * 32, kf, l31, half);
            mma_tile_x_frag<BF16>(dp, dOs, sub * 32, vf, l31, half);
            const bool edge = anydead || qb + 32 > a.Sq || (a.causal && kmaxw > qb);
            float p[16], ds[16];
            const float *lse_r = lse_t + sub * 32 + 4 * half, *dlt_r = dlt_t + sub * 32 + 4 * half;
#pragma unroll
            for (int r = 0; r < 16; ++r) p[r] = __builtin_amdgcn_exp2f(s[r] * c2 - lse_r[frag_row(r, 0)]);
            if (edge) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int qq = qb + frag_row(r, half);
                    const bool ok = kalive && qq < a.Sq && !(a.causal && key > qq);
                    p[r] = ok ? p[r] : 0.f;
                }
            }
            if (pd.thr) {
                const uint32_t xb = pd.row(qb + 4 * half, key);
                const bool odd = key & 1;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const uint32_t hsh = pd.hash(xb + (uint32_t)frag_row(r, 0) * pd.sk2);
                    const bool keep = odd ? pd.keep_hi(hsh) : pd.keep_lo(hsh);
                    ds[r] = p[r] * ((keep ? dp[r] : 0.f) - dlt_r[frag_row(r, 0)]);
                    p[r] = keep ? p[r] : 0.f;                         // dropped probabilities feed dV (1/(1-p) at the store)
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) ds[r] = p[r] * (dp[r] - dlt_r[frag_row(r, 0)]);
            }
            mma_T_x_p<BF16>(dv, dOt, sub * 32, p, l31, half);
            mma_T_x_p<BF16>(dk, Qt, sub * 32, ds, l31, half);
        }
        if (q0 + STEP < a.Sq) {
            commit(X, cur ^ 1);
            if (q0 + (DIST + 1) * STEP < a.Sq) issue(X, q0 + (DIST + 1) * STEP);
        }
        __syncthreads();
        cur ^= 1;
    };
    for (int qq0 = qstart; qq0 < (KK_DBG(a, 32) ? 0 : a.Sq); qq0 += 2 * STEP) {              // the bound is the same for both groups (barriers)
        tile_step(ra, qq0);
        if (qq0 + STEP < a.Sq) tile_step(rb, qq0 + STEP);
    }
    if constexpr (G == 2) {          // group 1's partial dK / dV -> LDS -> group 0
        float *mb = reinterpret_cast<float *>(smem_raw) + (wave * 64 + lane) * 65;      // 64 floats per lane (+1: bank spread)
        if (grp == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { mb[r] = dk[0][r]; mb[16 + r] = dk[1][r]; mb[32 + r] = dv[0][r]; mb[48 + r] = dv[1][r]; }
        }
        __syncthreads();
        if (grp == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { dk[0][r] += mb[r]; dk[1][r] += mb[16 + r]; dv[0][r] += mb[32 + r]; dv[1][r] += mb[48 + r]; }
        }
    }
    T *dk_row = static_cast<T *>(a.Out) + ((int64_t)b * a.Sk + key) * a.ldout + hh * 64;
    T *dv_row = static_cast<T *>(a.Out2) + ((int64_t)b * a.Sk + key) * a.ldout2 + hh * 64;
    if (a.hn[0].raw == nullptr) {
        if (kvalid && grp == 0) {
            store_row<T>(dk_row, dk, a.scale, half);
            store_row<T>(dv_row, dv, pd.inv_keep, half);
        }
        return;
    }
    float *colred = reinterpret_cast<float *>(smem_raw);          // [128 rows][65]
    const int64_t rrow = (int64_t)b * a.Sk + key;
    __syncthreads();
    if (grp == 0)
        hn_bwd_row<T>(dk, a.scale, kvalid, static_cast<const T *>(a.hn[0].raw) + rrow * a.hn[0].ldraw + hh * 64, dk_row, a.hn[0], key,
                      half, colred + (wave * 32 + l31) * 65);
    __syncthreads();
    hn_colsum(colred, a.hn[0].partials);
    __syncthreads();
    if (grp == 0)
        hn_bwd_row<T>(dv, pd.inv_keep, kvalid, static_cast<const T *>(a.hn[1].raw) + rrow * a.hn[1].ldraw + hh * 64, dv_row, a.hn[1], key,
                      half, colred + (wave * 32 + l31) * 65);
    __syncthreads();
    hn_colsum(colred, a.hn[1].partials);
}

// ------------------------------------------------------------------ backward: dK, dV, second generation (bf16 storage)
// attn_bwd_dkv_kernel<true, true, 2>'s arithmetic; Q and dO tiles reach LDS once each by DMA, in the image that serves both
// the row fragments (S^T = Q.K^T, dP^T = dO.V^T) and the transpose reads (dV^T += dO^T.P, dK^T += Q^T.dS) — the first
// generation staged four tiles (two of them transposed in registers) per step; lse / Delta rows by DMA; K and V rows, the
// head-norm epilogues' operands and the outputs as in attn_bwd_dq2_kernel.
// (body of the dK/dV kernel: kk_attn_bwd_dkv2.inc, included into attn_bwd_dkv2_kernel and attn_bwd_pair2_kernel below)

// The bodies live in .inc files because they must name a by-value KERNEL parameter: handed to a device function by reference
// (or read through a pointer to the kernarg segment) the same code spills 5-11 vector registers, and a spill is fatal beside
// LDS-DMA (scratch reloads queue behind the tile DMAs).
__global__ __launch_bounds__(512) void attn_bwd_dq2_kernel(AttnArgs a) {
#include "kk_attn_bwd_dq2.inc"
}
#ifdef KK_TUNING_HOOKS                                       // (second-generation dK/dV and pair kernels: A/B arms of the tools flavour, KK_ATTN_BWD3=0)
__global__ __launch_bounds__(512) void attn_bwd_dkv2_kernel(AttnArgs a) {
#include "kk_attn_bwd_dkv2.inc"
}
#endif
// dQ and dK/dV of one attention in ONE launch (grid z = 0: the dQ workgroups, z = 1: the dK/dV workgroups; Delta is an INPUT of
// both, see kk_gemm_dgrad_delta).  The two kernels are independent once Delta exists, each keeps one workgroup per CU (148 KB of
// LDS), and a causal launch is lopsided: dQ blocks near the end of the sequence see the most keys, dK/dV blocks near its start the
// most queries.  Dispatched in this order (x, y, then z; long blocks first inside each half) the CUs that finish a short dQ block
// pick up the long dK/dV blocks, so at S = 512 (one workgroup per CU and kernel) the pair takes about 5 block-units instead of
// 4 + 4, and a non-causal pair saves one launch's ramp and tail.
#ifdef KK_TUNING_HOOKS
__global__ __launch_bounds__(512) void attn_bwd_pair2_kernel(AttnArgs a_dq, AttnArgs a_dkv) {
    if (blockIdx.z == 0) {
#define a a_dq
#include "kk_attn_bwd_dq2.inc"
#undef a
    } else {
#define a a_dkv
#include "kk_attn_bwd_dkv2.inc"
#undef a
    }
}
#endif

// ------------------------------------------------------------------ backward, third generation: one wave group, two workgroups per CU
// The second-generation bodies as ONE 256-thread group each (kk_attn_bwd_dq3.inc / kk_attn_bwd_dkv3.inc): <= 70 KB of LDS, so two
// workgroups share a CU.  The waves per SIMD stay two (256 registers: the dK/dV half holds 64 accumulator registers per wave,
// DESIGN section 9) but they now belong to INDEPENDENT workgroups: no common barrier, one's prologue / epilogue under the other's
// loop, no merge of group partials, and the 2 x 256 workgroups of an 8 x 8 x 512^2 launch are resident at once (one round).  In the
// pair launch the dK/dV half of a causal launch hands out its SHORT blocks first: the i-th workgroup of each half land on the same
// CU, so every CU holds a long block of one kernel beside a short block of the other, concurrently.
__global__ __launch_bounds__(256, 2) void attn_bwd_dq3_kernel(AttnArgs a) {
#include "kk_attn_bwd_dq3.inc"
}
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv3_kernel(AttnArgs a) {
#include "kk_attn_bwd_dkv3.inc"
}
// Dispatch order: the dK/dV half (z = 0) goes out FIRST.  Its workgroups are the long ones (26.5 against 17.5 us at 512 x 512, four
// matmuls and two head-norm epilogues against three and one); dispatched last they are what a slot delayed by the side branch's
// workgroups finishes with.  Stand-alone the order makes no difference (32.5 us either way); inside the step it is -0.5 % at 8 x 512
// and -0.6 % at 8 x 1024 (interleaved, profiles/r05_attn_bwd_dispatch_order_ab.txt).  Probe bit 2048 restores dQ first.
__global__ __launch_bounds__(256, 2) void attn_bwd_pair3_kernel(AttnArgs a_dq, AttnArgs a_dkv) {
    KK_WG_STAMP(a_dq);
    if ((blockIdx.z == 1) != KK_DBG(a_dq, 2048)) {
#define a a_dq
#include "kk_attn_bwd_dq3.inc"
#undef a
    } else {
#define a a_dkv
#include "kk_attn_bwd_dkv3.inc"
#undef a
    }
}
// The pair launch that READS the dropout keep decisions the forward stored (AttnArgs::keep) instead of hashing them again: the same
// bodies compiled with KK_KEEP_BITS — same arithmetic on the same decisions, bit-identical outputs (tests), ~100 vector instructions
// per 32 x 32 unit less in each half.
#define KK_KEEP_BITS 1
__global__ __launch_bounds__(256, 2) void attn_bwd_pair3k_kernel(AttnArgs a_dq, AttnArgs a_dkv) {
    KK_WG_STAMP(a_dq);
    if ((blockIdx.z == 1) != KK_DBG(a_dq, 2048)) {             // (dK/dV first: see attn_bwd_pair3_kernel)
#define a a_dq
#include "kk_attn_bwd_dq3.inc"
#undef a
    } else {
#define a a_dkv
#include "kk_attn_bwd_dkv3.inc"
#undef a
    }
}
#undef KK_KEEP_BITS

// ------------------------------------------------------------------ backward in two passes (kk_attn_bwd_ws)
// The pair launch computes the scores, the exponentials, the dropout masks and dS TWICE (once per kernel: 7 S x S x 64 matmuls and
// ~560 vector instructions per 32 x 32 unit where the algorithm needs 5 and ~330), because dQ is a sum over keys and dK / dV sums
// over queries.  Here the dK/dV kernel — the same body — also stores dS as it feeds it to its dK MFMAs (bf16, [32 keys][32
// queries] tiles of 2 KB: 2 bytes per score, 32 MB per launch at 8 x 8 x 512^2, mostly served back by the Infinity Cache), and dQ =
// dS . K becomes a pass with NO vector work: four waves (one per 32 queries) stream K tiles and their dS tiles through a three-stage
// DMA ring and issue 4 MFMAs per unit, both operands by transpose reads (K^T as in the dQ kernel; dS^T [key][query] the same way: a
// 512-byte span of a tile per instruction, conflict free without a swizzle); the head-norm epilogue of the dQ kernel follows.
// dS is bit-identical to what the dQ kernel computes for itself (same MFMA sums, same rounding), so dQ differs from the pair
// launch's only by the order in which the key units are added (all of them in sequence here; two interleaved halves there).
#ifdef KK_TUNING_HOOKS
__global__ __launch_bounds__(512) void attn_bwd_dkv2s_kernel(AttnArgs a) {
#define KK_DKV_STORE_DS 1
#include "kk_attn_bwd_dkv2.inc"
#undef KK_DKV_STORE_DS
}
#endif
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv3s_kernel(AttnArgs a) {      // (the one-group body: what kk_attn_bwd's dK/dV half runs)
#define KK_DKV_STORE_DS 1
#include "kk_attn_bwd_dkv3.inc"
#undef KK_DKV_STORE_DS
}

__global__ __launch_bounds__(256) void attn_bwd_dqpass_kernel(AttnArgs a) {
    typedef __bf16 T;
    constexpr int NS = 3, KIMG = 64 * 64 * 2, DIMG = 2 * 4 * 2048, STAGE = KIMG + DIMG, NPT = 6;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];        // [stage][K image | 2 key units x 4 query units of dS]
    int bx_, by_;
    attn_block(a, bx_, by_, true);
    const int b = by_ / a.heads, hh = by_ % a.heads;
    const int qblk = bx_ * 128;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), half = lane >> 5, l31 = lane & 31;
    const int qmin = qblk + 32 * wave;
    const bool qvalid = qmin + l31 < a.Sq;
    int kend = a.Sk;
    if (a.causal && qblk + 128 < kend) kend = qblk + 128;
    int klim = kend;
    if (a.causal && qmin + 32 < klim) klim = qmin + 32;
    const int nt = (kend + 63) >> 6;
    const int nqu4 = ((a.Sq + 127) >> 7) << 2, nku = (a.Sk + 31) >> 5;
    const T *Kb = static_cast<const T *>(a.K) + (int64_t)b * a.Sk * a.ldk + hh * 64;
    const char *dsb = static_cast<const char *>(a.dS) + (int64_t)(b * a.heads + hh) * nku * nqu4 * 2048;
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(Kb), 0, (int)((((int64_t)a.Sk - 1) * a.ldk + 64) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(dsb), 0, (int)((int64_t)nku * nqu4 * 2048), 0x00020000);
    uint32_t kvo[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int p = threadIdx.x + 256 * j, row = p >> 3, pc = p & 7;
        kvo[j] = (uint32_t)(((int64_t)row * a.ldk + ((pc ^ (kk_xb(row) << 1)) * 8)) * 2);
    }
    const uint32_t ktile = (uint32_t)(64 * a.ldk * 2);
    const uint32_t drow = (uint32_t)(nqu4 * 2048), dq0 = (uint32_t)((qblk >> 5) * 2048 + threadIdx.x * 16);
    auto issue_tile = [&](int t, int st) {
        char *dst = smem_raw + st * STAGE + wave * 1024;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, KK_LDS_PTR(dst + j * 4096), 16, kvo[j] + (uint32_t)t * ktile, 0, 0, 0);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int j = 0; j < 2; ++j)           // the four query units' tiles of a key unit are one 8 KB run
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rd, KK_LDS_PTR(dst + KIMG + kk * 8192 + j * 4096), 16,
                                                         (uint32_t)(2 * t + kk) * drow + dq0 + (uint32_t)j * 4096, 0, 0, 0);
    };
#pragma unroll
    for (int t = 0; t < NS - 1; ++t)
        if (t < nt) issue_tile(t, t);
    const uint32_t sl = (uint32_t)(uintptr_t)KK_LDS_PTR(smem_raw);
    uint32_t ta[2], da;
    {
        const int L = lane & 15, kq = L >> 2, gi = (lane >> 4) & 1, xb = (((kq >> 1) & 1) << 1) | half;
#pragma unroll
        for (int db = 0; db < 2; ++db) ta[db] = (uint32_t)((4 * half + kq) * 128 + (((2 * db + gi) ^ xb) * 32) + 8 * (L & 3));
        da = (uint32_t)(KIMG + wave * 2048 + (4 * half + kq) * 64 + gi * 32 + 8 * (L & 3));
    }
    f32x16 dq[2];
    zero_acc(dq[0]); zero_acc(dq[1]);
    for (int t = 0; t < nt; ++t) {
        const int younger = min(nt - 1 - t, NS - 2);
        if (younger >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPT) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t + NS - 1 < nt) issue_tile(t + NS - 1, (t + NS - 1) % NS);
        const uint32_t stg = sl + (uint32_t)((t % NS) * STAGE);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            if (t * 64 + 32 * kk >= klim) continue;
            s16x4 tlo[4], thi[4], dlo[2], dhi[2];
            const uint32_t a0 = stg + kk * 4096 + ta[0], a1 = stg + kk * 4096 + ta[1], d0 = stg + kk * 8192 + da;
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(dlo[0]) : "v"(d0));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:512" : "=v"(dhi[0]) : "v"(d0));
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(tlo[0]) : "v"(a0));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=v"(thi[0]) : "v"(a0));
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(tlo[1]) : "v"(a1));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=v"(thi[1]) : "v"(a1));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=v"(dlo[1]) : "v"(d0));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1536" : "=v"(dhi[1]) : "v"(d0));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(tlo[2]) : "v"(a0));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:3072" : "=v"(thi[2]) : "v"(a0));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(tlo[3]) : "v"(a1));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:3072" : "=v"(thi[3]) : "v"(a1));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(tlo[i]), "+v"(thi[i]));
#pragma unroll
            for (int i = 0; i < 2; ++i) asm volatile("" : "+v"(dlo[i]), "+v"(dhi[i]));
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int db = 0; db < 2; ++db)
                    dq[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_pair(tlo[s2 * 2 + db], thi[s2 * 2 + db]), tr_pair(dlo[s2], dhi[s2]), dq[db], 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                           // the ring is free
    T *out0 = static_cast<T *>(a.Out) + ((int64_t)b * a.Sq + qmin) * a.ldout + hh * 64;
    char *otile = smem_raw + 36864 + wave * 4608;
    if (a.hn[0].raw == nullptr) {
        store_rows_via_lds(out0, a.ldout, a.Sq - qmin, dq, a.scale, otile, lane, a.wt);
        return;
    }
    const int nrows = a.Sq - qblk < 128 ? a.Sq - qblk : 128;
    hn_dma_inputs<256>(a.hn[0], (int64_t)b * a.Sq + qblk, qblk, nrows, hh, smem_raw, wave);       // raw | cos | sin: 48 KB
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    f32x16 dx[2];
    float cr[32];
    hn_bwd_row2_core(dq, a.scale, qvalid, smem_raw, smem_raw + 16384, smem_raw + 32768, wave * 32 + l31, a.hn[0].rope != 0, hn_load_gain(a.hn[0].gain, half), half, cr, dx);
    __syncthreads();                                           // every wave has read its image rows: colred and the store tiles lie over them
    float *colred = reinterpret_cast<float *>(smem_raw);       // [128 rows][65]
    hn_colred_store(cr, half, colred + (wave * 32 + l31) * 65);
    store_rows_via_lds(out0, a.ldout, a.Sq - qmin, dx, 1.f, otile, lane, a.wt);
    __syncthreads();
    hn_colsum(colred, a.hn[0].partials);
}

// Delta[b,h,q] = sum_d dO*O : one wave per (row, head).
template <typename T>
__global__ __launch_bounds__(256) void attn_delta_kernel(const T *__restrict__ O, const T *__restrict__ dO,
                                                         float *__restrict__ Delta, int64_t npairs, int heads, int Sq,
                                                         int64_t ldo, int64_t lddo) {
    const int lane = threadIdx.x & 63;
    for (int64_t pr = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); pr < npairs; pr += (int64_t)gridDim.x * 4) {
        const int64_t row = pr / heads;
        const int hd = (int)(pr - row * heads);
        const float v = wave_sum((float)O[row * ldo + hd * 64 + lane] * (float)dO[row * lddo + hd * 64 + lane]);
        if (lane == 0) {
            const int64_t b = row / Sq, q = row - b * Sq;
            Delta[(b * heads + hd) * Sq + q] = v;
        }
    }
}

int g_attn_groups = 2;
static int attn_v2_mask() {              // bit 0: forward, bit 1: dQ, bit 2: dK/dV second-generation kernels
    static const int v = kk_tune_env("KK_ATTN_V2", 7);
    return v;
}
#ifdef KK_TUNING_HOOKS
static void *g_attn_trace = nullptr;
extern "C" int kk_attn_trace(void *buf) { g_attn_trace = buf; return 0; }      // tools: destination of the stamps of probe bit 256
#endif
static int attn_dbg() {                  // timing probes of tools/probes (tools build only: results are wrong when set)
    static const int v = kk_tune_env("KK_ATTN_DBG", 0);
    return v;
}
static bool al16(const void *p) { return ((uintptr_t)p & 15) == 0; }
// The second-generation BACKWARD kernels also take sequences of ONE 64-row tile (33..64 queries and keys: the text encoder): wave
// group 1 then has no tile and only joins the barriers and the merge, but the DMA prologue, the LDS-transposed epilogues and the
// one-launch form (kk_attn_bwd) beat the first generation's 13 + 17 us there (KK_ATTN_V2_SMALL=0 in the tools build: old dispatch)
static bool attn_v2_small(int Sq, int Sk) {
    static const int v = kk_tune_env("KK_ATTN_V2_SMALL", 1);
    return v != 0 && Sq > 32 && Sk > 32;
}
static int attn_pair() {                 // KK_ATTN_PAIR=0: kk_attn_bwd issues the dQ and the dK/dV kernel as two launches
    static const int v = kk_tune_env("KK_ATTN_PAIR", 1);
    return v;
}
#ifdef KK_TUNING_HOOKS
static int attn_gen3() {                 // KK_ATTN_BWD3: bit 0 the one-group backward kernels (two workgroups per CU), bit 1 short-first dK/dV half
    static const int v = kk_tune_env("KK_ATTN_BWD3", 3);
    return v;
}
#else
static constexpr int attn_gen3() { return 3; }        // (the product carries the third generation only; the second-generation dK/dV / pair kernels are tools arms)
#endif
static int g_attn_cus() {
    static const int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) return v;
        return 256;
    }();
    return n;
}
static int attn_xcd_env() {
    static const int v = kk_tune_env("KK_ATTN_XCD", 2);
    return v;
}
// AttnArgs::xcd_map of a launch: 0 = launch order, 1 = a head's blocks on one XCD, 2 (default for causal launches) = that, with the
// blocks of an XCD's heads longest first (attn_block).  KK_ATTN_XCD=1 keeps 1 for the single-kernel causal launches.
static int attn_xcd_map(int causal = 0) {
    const int v = attn_xcd_env();
    return v == 2 ? (causal ? 2 : 1) : v;
}

// Launch KERNEL<BF16, ST16, G> with G*256 threads and its dynamic LDS (buffers x groups x NT tiles; above 64 KB the
// kernel attribute has to be raised once).
template <typename K>
int launch_attn(K kernel, dim3 grid, int G, size_t lds, hipStream_t s, const AttnArgs &a) {
    if (lds > 64 * 1024) {
        // (one table for every kernel of this signature: 64 slots for the ~20 kernels with more than 64 KB of dynamic LDS; a full table
        //  is an error, not a silent hipFuncSetAttribute per launch — ADVICE r4)
        static thread_local const void *raised[64];
        bool done = false;
        for (const void *p : raised) done = done || p == (const void *)kernel;
        if (!done) {
            hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return kk_fail((int)e, "attention: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e));
            bool noted = false;
            for (auto &p : raised)
                if (!p) { p = (const void *)kernel; noted = true; break; }
            if (!noted) return kk_fail(KK_EINVAL, "attention: the table of kernels with raised LDS limits is full");
        }
    }
    hipLaunchKernelGGL(kernel, grid, dim3(256 * G), lds, s, a);
    return 0;
}
#define KK_ATTN_LDS(BF16, G, NT, EXTRA) ((size_t)2 * (G) * (NT) * 64 * ACfg<BF16>::LR * sizeof(typename ACfg<BF16>::elem) + (EXTRA))
#define KK_ATTN_LAUNCH_X(KERNEL, BF16, ST16, G, NT, EXTRA)                                                                         \
    do {                                                                                                                           \
        int rc__ = (G) == 2 ? launch_attn(KERNEL<BF16, ST16, 2>, grid, 2, KK_ATTN_LDS(BF16, 2, NT, EXTRA), (hipStream_t)stream, a) \
                            : launch_attn(KERNEL<BF16, ST16, 1>, grid, 1, KK_ATTN_LDS(BF16, 1, NT, EXTRA), (hipStream_t)stream, a); \
        if (rc__) return rc__;                                                                                                     \
    } while (0)
#define KK_ATTN_LAUNCH(KERNEL, BF16, ST16, G, NT) KK_ATTN_LAUNCH_X(KERNEL, BF16, ST16, G, NT, 0)

int check_common(const char *name, int B, int heads, int Sq, int Sk, int math, const int64_t *lds, int nld) {
    KK_REQUIRE(B > 0 && heads > 0 && Sq > 0 && Sk > 0, "%s: bad shape B=%d heads=%d Sq=%d Sk=%d", name, B, heads, Sq, Sk);
    KK_REQUIRE(math == KK_MATH_F32 || math == KK_MATH_BF16, "%s: bad math mode", name);
    for (int i = 0; i < nld; ++i) KK_REQUIRE(lds[i] % 8 == 0 && lds[i] >= 64 * heads, "%s: row stride %ld unsupported", name, (long)lds[i]);
    KK_REQUIRE((int64_t)B * heads < 65536, "%s: B*heads too large", name);
    return 0;
}

int check_headnorm(const char *name, const KkAttnHeadNorm *hn, int n) {
    for (int i = 0; i < n; ++i) {
        KK_REQUIRE(hn[i].raw && hn[i].gain && hn[i].partials && hn[i].ldraw % 8 == 0, "%s: head-norm epilogue %d needs raw, gain, partials", name, i);
        KK_REQUIRE(!hn[i].rope || (hn[i].cos_t && hn[i].sin_t), "%s: head-norm epilogue %d: RoPE needs cos/sin tables", name, i);
    }
    return 0;
}

}  // namespace

// rows of the [workgroups][64] partial gain-gradient matrix a backward launch with a head-norm epilogue writes
extern "C" int kk_attn_bwd_blocks(int B, int heads, int S) { return kk_cdiv(S, 128) * B * heads; }

// Whether a forward launch of this shape stores keep bits (the third-generation kernels) and how many bytes they take; 0 = no.
extern "C" int64_t kk_attn_keep_bytes(int B, int heads, int Sq, int Sk) {
    if (B <= 0 || heads <= 0 || Sq <= 0 || Sk <= 128 || Sk > 4096 || g_attn_groups != 2 || !(attn_v2_mask() & 1)) return 0;
    const int64_t per_head = (int64_t)kk_cdiv(Sq, 32) * kk_cdiv(Sk, 32) * 128;
    return per_head < (1ll << 31) ? (int64_t)B * heads * per_head : 0;
}

static thread_local const void *g_warm_ptr[2] = {nullptr, nullptr};
static thread_local uint32_t g_warm_bytes[2] = {0u, 0u};
// The NEXT third-generation forward launch of this thread also warms these (up to two) read-only matrices into every XCD's L2 (see
// AttnArgs::warm).  One-shot: consumed by that launch, dropped by any other forward launch.
extern "C" int kk_attn_warm_next(const void *w0, int64_t bytes0, const void *w1, int64_t bytes1) {
    g_warm_ptr[0] = w0; g_warm_bytes[0] = (w0 && bytes0 > 0 && bytes0 < (1ll << 31)) ? (uint32_t)bytes0 : 0u;
    g_warm_ptr[1] = w1; g_warm_bytes[1] = (w1 && bytes1 > 0 && bytes1 < (1ll << 31)) ? (uint32_t)bytes1 : 0u;
    if (g_warm_bytes[0] == 0u) { g_warm_ptr[0] = g_warm_ptr[1]; g_warm_bytes[0] = g_warm_bytes[1]; g_warm_bytes[1] = 0u; }
    return 0;
}

static int attn_fwd_impl(const float *Q, const float *K, const float *V, float *O, float *LSE, int B, int heads,
                         int Sq, int Sk, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                         const uint8_t *key_mask, int causal, float scale, const uint32_t *seed, uint32_t site,
                         float p_drop, int math, int io_bf16, void *keep, void *stream, int keep_rd = 0) {
    KK_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "kk_attn_fwd: dropout probability must be in [0,1)");
    KK_REQUIRE(!io_bf16 || math == KK_MATH_BF16, "kk_attn_fwd: bf16 storage needs KK_MATH_BF16");
    const int64_t lds[4] = {ldq, ldk, ldv, ldo};
    if (int rc = check_common("kk_attn_fwd", B, heads, Sq, Sk, math, lds, 4)) return rc;
    AttnArgs a = {};
    a.Q = Q; a.K = K; a.V = V; a.Out = O; a.LSEo = LSE; a.key_mask = key_mask;
    a.B = B; a.heads = heads; a.Sq = Sq; a.Sk = Sk; a.causal = causal;
    a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldout = ldo; a.scale = scale;
    a.seed = p_drop > 0.f ? seed : nullptr; a.site = site; a.p_drop = p_drop; a.xcd_map = attn_xcd_map(causal); a.dbg = attn_dbg(); a.wt = kk_write_through((int64_t)B * std::max(Sq, Sk));
    a.keep = keep;
    a.keep_rd = (keep_rd && keep != nullptr && a.seed != nullptr) ? 1 : 0;
    a.warm[0] = g_warm_ptr[0]; a.warm[1] = g_warm_ptr[1];
    a.warm_bytes[0] = g_warm_bytes[0]; a.warm_bytes[1] = g_warm_bytes[1];
    g_warm_bytes[0] = g_warm_bytes[1] = 0u;                     // (one-shot)
#ifdef KK_TUNING_HOOKS
    if (a.dbg & (256 | 4096)) a.DeltaOut = static_cast<float *>(g_attn_trace);
#endif
    if (Sq == 1 && a.seed == nullptr && !causal && Sk <= 8192 && ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0 &&
        (((uintptr_t)Q | (uintptr_t)K | (uintptr_t)V) & 15) == 0) {  // a decoder step of the incremental path: one (batch, head) per workgroup
        const size_t lds = (size_t)(((Sk + 3) & ~3) + 16 * 64 + 32) * sizeof(float);
        if (io_bf16) hipLaunchKernelGGL(attn_decode_kernel<__bf16>, dim3(B * heads), dim3(1024), lds, (hipStream_t)stream, a);
        else hipLaunchKernelGGL(attn_decode_kernel<float>, dim3(B * heads), dim3(1024), lds, (hipStream_t)stream, a);
        KK_LAUNCH_CHECK("kk_attn_fwd");
        return 0;
    }
    dim3 grid(kk_cdiv(Sq, 128), B * heads);
    const int G = (Sk > 64 && g_attn_groups == 2) ? 2 : 1;          // one key tile: nothing to split
    // second-generation kernel (DMA-staged, software-pipelined): bf16 storage, two key groups, 16-byte aligned operands
    const int fwd_v2 = attn_v2_mask() & 1;
    if (io_bf16 && fwd_v2 && G == 2 && Sk <= 4096 && (((uintptr_t)Q | (uintptr_t)K | (uintptr_t)V | (uintptr_t)O) & 15) == 0 && (int64_t)Sk * std::max(ldk, ldv) * 2 < (1ll << 31)) {
        // third generation: two 74 KB / 128-register workgroups per CU.  128-query blocks x 2 key slots when that gives two workgroups
        // per CU, else 64-query blocks x 4 key slots (a 512-frame launch: 512 workgroups instead of 256)
        static const int fwd3 = kk_tune_env("KK_ATTN_FWD3", 1);
        if (fwd3 && Sk > 128) {
            const bool big = (int64_t)kk_cdiv(Sq, 128) * B * heads >= 2 * g_attn_cus();    // (two workgroups per CU)
            // (a conditional expression: hipcc does not emit the host stub of a kernel template named only inside an if / else chain)
            if (a.keep_rd) {
                kk_note_kernel((big || fwd3 == 2) ? "attn_fwd3_q128r" : "attn_fwd3_q64r");
                const int rcr = (big || fwd3 == 2)
                    ? launch_attn(attn_fwd3_q128r_kernel, dim3(kk_cdiv(Sq, 128), B * heads), 2, (size_t)3 * 16384 + 512 + 16384, (hipStream_t)stream, a)
                    : launch_attn(attn_fwd3_q64r_kernel, dim3(kk_cdiv(Sq, 64), B * heads), 2, (size_t)2 * 32768 + 512 + 8192, (hipStream_t)stream, a);
                if (rcr) return rcr;
                KK_LAUNCH_CHECK("kk_attn_fwd_rb");
                return 0;
            }
            kk_note_kernel((big || fwd3 == 2) ? "attn_fwd3_q128" : "attn_fwd3_q64");
            if (kk_capture(kk_last_kernel(), a, (big || fwd3 == 2) ? dim3(kk_cdiv(Sq, 128), B * heads) : dim3(kk_cdiv(Sq, 64), B * heads), 512,
                           (big || fwd3 == 2) ? (size_t)3 * 16384 + 512 + 16384 : (size_t)2 * 32768 + 512 + 8192)) return 0;
            const int rc3 = (big || fwd3 == 2)
                ? launch_attn(attn_fwd3_q128_kernel, dim3(kk_cdiv(Sq, 128), B * heads), 2, (size_t)3 * 16384 + 512 + 16384, (hipStream_t)stream, a)
                : launch_attn(attn_fwd3_q64_kernel, dim3(kk_cdiv(Sq, 64), B * heads), 2, (size_t)2 * 32768 + 512 + 8192, (hipStream_t)stream, a);
            if (rc3) return rc3;
            KK_LAUNCH_CHECK("kk_attn_fwd");
            return 0;
        }
        KK_REQUIRE(keep == nullptr, "kk_attn_fwd_kb / _rb: only the third-generation forward stores or reads keep bits (ask kk_attn_keep_bytes)");
        static const int ns2 = kk_tune_env("KK_ATTN_NS", 3);
        kk_note_kernel("attn_fwd2");
#ifdef KK_TUNING_HOOKS
        int rc2 = ns2 == 4 ? launch_attn(attn_fwd2_kernel<4>, grid, 2, (size_t)2 * 4 * 16384 + 512 + 16384, (hipStream_t)stream, a)
                           : launch_attn(attn_fwd2_kernel<3>, grid, 2, (size_t)2 * 3 * 16384 + 512 + 16384, (hipStream_t)stream, a);
#else
        (void)ns2;
        int rc2 = launch_attn(attn_fwd2_kernel<3>, grid, 2, (size_t)2 * 3 * 16384 + 512 + 16384, (hipStream_t)stream, a);
#endif
        if (rc2) return rc2;
        KK_LAUNCH_CHECK("kk_attn_fwd");
        return 0;
    }
    KK_REQUIRE(keep == nullptr, "kk_attn_fwd_kb: this launch (storage, alignment or shape) does not take the third-generation forward, which alone stores keep bits");
    kk_note_kernel("attn_fwd");
    if (io_bf16) KK_ATTN_LAUNCH(attn_fwd_kernel, true, true, G, 2);
    else if (math == KK_MATH_BF16) KK_ATTN_LAUNCH(attn_fwd_kernel, true, false, G, 2);
    else KK_ATTN_LAUNCH(attn_fwd_kernel, false, false, G, 2);
    KK_LAUNCH_CHECK("kk_attn_fwd");
    return 0;
}

extern "C" int kk_attn_fwd(const float *Q, const float *K, const float *V, float *O, float *LSE, int B, int heads,
                           int Sq, int Sk, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                           const uint8_t *key_mask, int causal, float scale, const uint32_t *seed, uint32_t site,
                           float p_drop, int math, int io_bf16, void *stream) {
    return attn_fwd_impl(Q, K, V, O, LSE, B, heads, Sq, Sk, ldq, ldk, ldv, ldo, key_mask, causal, scale, seed, site, p_drop, math, io_bf16,
                         nullptr, stream);
}
// kk_attn_fwd that also stores the dropout keep decisions (AttnArgs::keep; kk_attn_keep_bytes(B, heads, Sq, Sk) bytes, > 0 required)
extern "C" int kk_attn_fwd_kb(const float *Q, const float *K, const float *V, float *O, float *LSE, int B, int heads,
                              int Sq, int Sk, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                              const uint8_t *key_mask, int causal, float scale, const uint32_t *seed, uint32_t site,
                              float p_drop, int math, int io_bf16, void *keep, void *stream) {
    KK_REQUIRE(keep == nullptr || (al16(keep) && kk_attn_keep_bytes(B, heads, Sq, Sk) > 0), "kk_attn_fwd_kb: no keep bits for this shape (kk_attn_keep_bytes) or unaligned buffer");
    return attn_fwd_impl(Q, K, V, O, LSE, B, heads, Sq, Sk, ldq, ldk, ldv, ldo, key_mask, causal, scale, seed, site, p_drop, math, io_bf16,
                         keep, stream);
}

// kk_attn_fwd whose dropout keep decisions are READ from `keep` (filled by kk_attn_keep_gen for the same seed value, site, p and shape):
// same output bits as kk_attn_fwd / kk_attn_fwd_kb
extern "C" int kk_attn_fwd_rb(const float *Q, const float *K, const float *V, float *O, float *LSE, int B, int heads,
                              int Sq, int Sk, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                              const uint8_t *key_mask, int causal, float scale, const uint32_t *seed, uint32_t site,
                              float p_drop, int math, int io_bf16, const void *keep, void *stream) {
    KK_REQUIRE(keep != nullptr && al16(keep) && kk_attn_keep_bytes(B, heads, Sq, Sk) > 0 && p_drop > 0.f && seed != nullptr && io_bf16,
               "kk_attn_fwd_rb: needs dropout, bf16 storage and a keep-bit array of a shape that has one (kk_attn_keep_bytes)");
    return attn_fwd_impl(Q, K, V, O, LSE, B, heads, Sq, Sk, ldq, ldk, ldv, ldo, key_mask, causal, scale, seed, site, p_drop, math, io_bf16,
                         const_cast<void *>(keep), stream, 1);
}

// The keep bits of n <= 16 attention launches in one launch (see attn_keep_gen_kernel); sites: HOST array read during the call.
extern "C" int kk_attn_keep_gen(const KkKeepSite *sites, int n, const uint32_t *seed, int seed_offset, int max_workgroups, void *stream) {
    KK_REQUIRE(sites != nullptr && seed != nullptr && n > 0 && n <= 16, "kk_attn_keep_gen: 1..16 sites and the seed are required");
    KeepGenArgs g;
    g.n = n;
    g.seed = seed;
    g.seed_offset = (uint32_t)seed_offset;
    g.start[0] = 0;
    for (int i = 0; i < n; ++i) {
        const KkKeepSite &st = sites[i];
        KK_REQUIRE(st.keep != nullptr && al16(st.keep) && st.p > 0.f && st.p < 1.f && kk_attn_keep_bytes(st.B, st.heads, st.Sq, st.Sk) > 0,
                   "kk_attn_keep_gen: site %d has no keep-bit array (kk_attn_keep_bytes) or no dropout", i);
        g.s[i] = st;
        g.start[i + 1] = g.start[i] + (int64_t)st.B * st.heads * kk_cdiv(st.Sq, 32) * kk_cdiv(st.Sk, 32);
    }
    int64_t wgs = (g.start[n] + 15) / 16;                         // (a wave takes four units at a time)
    const int cap = max_workgroups > 0 ? max_workgroups : 2048;   // (thin: it runs beside another launch and must leave it its wave slots)
    if (wgs > cap) wgs = cap;
    kk_note_kernel("attn_keep_gen");
    hipLaunchKernelGGL(attn_keep_gen_kernel, dim3((unsigned)wgs), dim3(256), 0, (hipStream_t)stream, g);
    KK_LAUNCH_CHECK("kk_attn_keep_gen");
    return 0;
}

extern "C" int kk_attn_delta(const float *O, const float *dO, float *Delta, int B, int heads, int Sq, int64_t ldo,
                             int64_t lddo, int io_bf16, void *stream) {
    KK_REQUIRE(B > 0 && heads > 0 && Sq > 0, "kk_attn_delta: bad shape");
    const int64_t npairs = (int64_t)B * Sq * heads;
    int blocks = kk_cdiv(npairs, 4);
    if (blocks > 8192) blocks = 8192;
    if (io_bf16)
        hipLaunchKernelGGL(attn_delta_kernel<__bf16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const __bf16 *>(O),
                           reinterpret_cast<const __bf16 *>(dO), Delta, npairs, heads, Sq, ldo, lddo);
    else
        hipLaunchKernelGGL(attn_delta_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, O, dO, Delta, npairs, heads, Sq, ldo, lddo);
    KK_LAUNCH_CHECK("kk_attn_delta");
    return 0;
}

extern "C" int kk_attn_bwd_dq(const float *Q, const float *K, const float *V, const float *dO, const float *LSE,
                              float *Delta, float *dQ, int B, int heads, int Sq, int Sk, int64_t ldq,
                              int64_t ldk, int64_t ldv, int64_t lddo, int64_t lddq, const uint8_t *key_mask,
                              int causal, float scale, const uint32_t *seed, uint32_t site, float p_drop, int math,
                              int io_bf16, const float *O, int64_t ldo, const KkAttnHeadNorm *hn, void *stream) {
    KK_REQUIRE(!io_bf16 || math == KK_MATH_BF16, "kk_attn_bwd_dq: bf16 storage needs KK_MATH_BF16");
    const int64_t lds[5] = {ldq, ldk, ldv, lddo, lddq};
    if (int rc = check_common("kk_attn_bwd_dq", B, heads, Sq, Sk, math, lds, 5)) return rc;
    AttnArgs a = {};
    a.Q = Q; a.K = K; a.V = V; a.dO = dO; a.LSE = LSE; a.Delta = Delta; a.Out = dQ; a.key_mask = key_mask;
    a.B = B; a.heads = heads; a.Sq = Sq; a.Sk = Sk; a.causal = causal;
    a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.lddo = lddo; a.ldout = lddq; a.scale = scale;
    if (O) {                                  // Delta is an OUTPUT of this call (and still the input of kk_attn_bwd_dkv)
        KK_REQUIRE(ldo % 8 == 0 && ldo >= 64 * heads, "kk_attn_bwd_dq: row stride of O unsupported");
        a.O = O; a.ldo = ldo; a.DeltaOut = Delta;
    }
    a.seed = p_drop > 0.f ? seed : nullptr; a.site = site; a.p_drop = p_drop; a.xcd_map = attn_xcd_map(causal); a.dbg = attn_dbg(); a.wt = kk_write_through((int64_t)B * std::max(Sq, Sk));
    if (hn) {
        if (int rc = check_headnorm("kk_attn_bwd_dq", hn, 1)) return rc;
        a.hn[0] = hn[0];
    }
    dim3 grid(kk_cdiv(Sq, 128), B * heads);
    const int G = (Sk > 64 && g_attn_groups == 2) ? 2 : 1;
    if (io_bf16 && (attn_v2_mask() & 2) && (G == 2 || attn_v2_small(Sq, Sk)) && g_attn_groups == 2 && Sk <= 4096 && al16(Q) && al16(K) && al16(V) && al16(dO) && al16(dQ) && (!O || al16(O)) &&
        (!hn || (al16(hn->raw) && (!hn->rope || (al16(hn->cos_t) && al16(hn->sin_t))))) && (int64_t)Sk * std::max(ldk, ldv) * 2 < (1ll << 31)) {
        kk_note_kernel((!O && (attn_gen3() & 1)) ? "attn_bwd_dq3" : "attn_bwd_dq2");
        int rc2 = (!O && (attn_gen3() & 1)) ? launch_attn(attn_bwd_dq3_kernel, grid, 1, (size_t)3 * 16384 + 512 + 16384, (hipStream_t)stream, a)
                                            : launch_attn(attn_bwd_dq2_kernel, grid, 2, (size_t)2 * 3 * 16384 + 512 + 3 * 16384, (hipStream_t)stream, a);
        if (rc2) return rc2;
        KK_LAUNCH_CHECK("kk_attn_bwd_dq");
        return 0;
    }
    kk_note_kernel("attn_bwd_dq");
    if (io_bf16) KK_ATTN_LAUNCH(attn_bwd_dq_kernel, true, true, G, 3);
    else if (math == KK_MATH_BF16) KK_ATTN_LAUNCH(attn_bwd_dq_kernel, true, false, G, 3);
    else KK_ATTN_LAUNCH(attn_bwd_dq_kernel, false, false, G, 2);
    KK_LAUNCH_CHECK("kk_attn_bwd_dq");
    return 0;
}

extern "C" int kk_attn_bwd_dkv(const float *Q, const float *K, const float *V, const float *dO, const float *LSE,
                               const float *Delta, float *dK, float *dV, int B, int heads, int Sq, int Sk,
                               int64_t ldq, int64_t ldk, int64_t ldv, int64_t lddo, int64_t lddk, int64_t lddv,
                               const uint8_t *key_mask, int causal, float scale, const uint32_t *seed, uint32_t site,
                               float p_drop, int math, int io_bf16, const KkAttnHeadNorm *hn, void *stream) {
    KK_REQUIRE(!io_bf16 || math == KK_MATH_BF16, "kk_attn_bwd_dkv: bf16 storage needs KK_MATH_BF16");
    const int64_t lds[6] = {ldq, ldk, ldv, lddo, lddk, lddv};
    if (int rc = check_common("kk_attn_bwd_dkv", B, heads, Sq, Sk, math, lds, 6)) return rc;
    AttnArgs a = {};
    a.Q = Q; a.K = K; a.V = V; a.dO = dO; a.LSE = LSE; a.Delta = Delta; a.Out = dK; a.Out2 = dV; a.key_mask = key_mask;
    a.B = B; a.heads = heads; a.Sq = Sq; a.Sk = Sk; a.causal = causal;
    a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.lddo = lddo; a.ldout = lddk; a.ldout2 = lddv; a.scale = scale;
    a.seed = p_drop > 0.f ? seed : nullptr; a.site = site; a.p_drop = p_drop; a.xcd_map = attn_xcd_map(causal); a.dbg = attn_dbg(); a.wt = kk_write_through((int64_t)B * std::max(Sq, Sk));
    if (hn) {
        if (int rc = check_headnorm("kk_attn_bwd_dkv", hn, 2)) return rc;
        a.hn[0] = hn[0]; a.hn[1] = hn[1];
    }
    dim3 grid(kk_cdiv(Sk, 128), B * heads);
    const int G = (Sq > 64 && g_attn_groups == 2) ? 2 : 1;          // one query tile: nothing to split
    if (io_bf16 && (attn_v2_mask() & 4) && (G == 2 || attn_v2_small(Sq, Sk)) && g_attn_groups == 2 && al16(Q) && al16(K) && al16(V) && al16(dO) && al16(dK) && al16(dV) &&
        (!hn || (al16(hn[0].raw) && al16(hn[1].raw) && !hn[1].rope && (!hn[0].rope || (al16(hn[0].cos_t) && al16(hn[0].sin_t))))) &&
        (int64_t)Sq * std::max(ldq, lddo) * 2 < (1ll << 31)) {
        kk_note_kernel((attn_gen3() & 1) ? "attn_bwd_dkv3" : "attn_bwd_dkv2");
#ifdef KK_TUNING_HOOKS
        int rc2 = (attn_gen3() & 1) ? launch_attn(attn_bwd_dkv3_kernel, grid, 1, (size_t)71680, (hipStream_t)stream, a)
                                    : launch_attn(attn_bwd_dkv2_kernel, grid, 2, (size_t)2 * 3 * (16384 + 512) + 3 * 16384, (hipStream_t)stream, a);
#else
        int rc2 = launch_attn(attn_bwd_dkv3_kernel, grid, 1, (size_t)71680, (hipStream_t)stream, a);
#endif
        if (rc2) return rc2;
        KK_LAUNCH_CHECK("kk_attn_bwd_dkv");
        return 0;
    }
    kk_note_kernel("attn_bwd_dkv");
    if (io_bf16) KK_ATTN_LAUNCH_X(attn_bwd_dkv_kernel, true, true, G, 4, 2 * G * 128 * sizeof(float));
    else if (math == KK_MATH_BF16) KK_ATTN_LAUNCH_X(attn_bwd_dkv_kernel, true, false, 1, 4, 2 * 128 * sizeof(float));   // (G = 2 would spill)
    else KK_ATTN_LAUNCH_X(attn_bwd_dkv_kernel, false, false, G, 2, 2 * G * 128 * sizeof(float));
    KK_LAUNCH_CHECK("kk_attn_bwd_dkv");
    return 0;
}

// dQ, dK and dV in one launch (attn_bwd_pair2_kernel) when both second-generation kernels apply; otherwise the two launches
// above, in order.  Delta[b, head, q] = sum_d dO * O is an INPUT here (kk_gemm_dgrad_delta writes it with dO, or kk_attn_delta).
// hn_q / hn_kv: the head-norm backward epilogues of kk_attn_bwd_dq / kk_attn_bwd_dkv (both or neither).
static int attn_bwd_impl(const float *Q, const float *K, const float *V, const float *dO, const float *LSE, const float *Delta,
                         float *dQ, float *dK, float *dV, int B, int heads, int Sq, int Sk, int64_t ldq, int64_t ldk,
                         int64_t ldv, int64_t lddo, int64_t lddq, int64_t lddk, int64_t lddv, const uint8_t *key_mask,
                         int causal, float scale, const uint32_t *seed, uint32_t site, float p_drop, int math, int io_bf16,
                         const KkAttnHeadNorm *hn_q, const KkAttnHeadNorm *hn_kv, const void *keep, void *stream) {
    KK_REQUIRE(Delta != nullptr, "kk_attn_bwd: Delta is an input of this call");
    KK_REQUIRE((hn_q == nullptr) == (hn_kv == nullptr), "kk_attn_bwd: head-norm epilogues for both kernels or for neither");
    const int G = (Sk > 64 && Sq > 64 && g_attn_groups == 2) ? 2 : 1;
    const bool pair = io_bf16 && math == KK_MATH_BF16 && (attn_v2_mask() & 6) == 6 && attn_pair() && (G == 2 || attn_v2_small(Sq, Sk)) &&
                      g_attn_groups == 2 && Sk <= 4096 &&
                      kk_cdiv(Sq, 128) == kk_cdiv(Sk, 128) && al16(Q) && al16(K) && al16(V) && al16(dO) && al16(dQ) && al16(dK) && al16(dV) &&
                      (!hn_q || (al16(hn_q->raw) && (!hn_q->rope || (al16(hn_q->cos_t) && al16(hn_q->sin_t))))) &&
                      (!hn_kv || (al16(hn_kv[0].raw) && al16(hn_kv[1].raw) && !hn_kv[1].rope &&
                                  (!hn_kv[0].rope || (al16(hn_kv[0].cos_t) && al16(hn_kv[0].sin_t))))) &&
                      (int64_t)Sk * std::max(ldk, ldv) * 2 < (1ll << 31) && (int64_t)Sq * std::max(ldq, lddo) * 2 < (1ll << 31);
    if (!pair) {
        g_warm_bytes[0] = g_warm_bytes[1] = 0u;                 // (one-shot: a launch that cannot warm drops the request, it never waits for a later one)
        if (int rc = kk_attn_bwd_dq(Q, K, V, dO, LSE, const_cast<float *>(Delta), dQ, B, heads, Sq, Sk, ldq, ldk, ldv, lddo, lddq, key_mask,
                                    causal, scale, seed, site, p_drop, math, io_bf16, nullptr, 0, hn_q, stream))
            return rc;
        return kk_attn_bwd_dkv(Q, K, V, dO, LSE, Delta, dK, dV, B, heads, Sq, Sk, ldq, ldk, ldv, lddo, lddk, lddv, key_mask, causal,
                               scale, seed, site, p_drop, math, io_bf16, hn_kv, stream);
    }
    KK_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "kk_attn_bwd: dropout probability must be in [0,1)");
    const int64_t lds[7] = {ldq, ldk, ldv, lddo, lddq, lddk, lddv};
    if (int rc = check_common("kk_attn_bwd", B, heads, Sq, Sk, math, lds, 7)) return rc;
    struct { AttnArgs dq, dkv; } p = {};
    AttnArgs &a = p.dq;
    a.Q = Q; a.K = K; a.V = V; a.dO = dO; a.LSE = LSE; a.Delta = Delta; a.Out = dQ; a.key_mask = key_mask;
    a.B = B; a.heads = heads; a.Sq = Sq; a.Sk = Sk; a.causal = causal;
    a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.lddo = lddo; a.ldout = lddq; a.scale = scale;
    a.seed = p_drop > 0.f ? seed : nullptr; a.site = site; a.p_drop = p_drop; a.xcd_map = attn_xcd_map(causal); a.dbg = attn_dbg(); a.wt = kk_write_through((int64_t)B * std::max(Sq, Sk));
    if (a.xcd_map && causal) a.xcd_map = 2;                    // (the pair launch: always block-major when causal)
    p.dkv = a;
    if (attn_gen3() & 1) {                                     // the dQ half of the third-generation pair launch warms the next GEMMs' weights
        p.dq.warm[0] = g_warm_ptr[0]; p.dq.warm[1] = g_warm_ptr[1];
        p.dq.warm_bytes[0] = g_warm_bytes[0]; p.dq.warm_bytes[1] = g_warm_bytes[1];
    }
    g_warm_bytes[0] = g_warm_bytes[1] = 0u;                     // (one-shot)
    p.dkv.Out = dK; p.dkv.Out2 = dV; p.dkv.ldout = lddk; p.dkv.ldout2 = lddv;
    if (hn_q) {
        if (int rc = check_headnorm("kk_attn_bwd", hn_q, 1)) return rc;
        if (int rc = check_headnorm("kk_attn_bwd", hn_kv, 2)) return rc;
        p.dq.hn[0] = hn_q[0];
        p.dkv.hn[0] = hn_kv[0]; p.dkv.hn[1] = hn_kv[1];
    }
    if (attn_gen3() & 1) {                                     // one wave group per workgroup: two workgroups per CU
        const size_t lds3 = 71680;
        static thread_local bool raised3 = false;
        if (!raised3) {
            hipError_t e = hipFuncSetAttribute((const void *)attn_bwd_pair3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
            if (e != hipSuccess) return kk_fail((int)e, "kk_attn_bwd: cannot reserve %zu bytes of LDS: %s", lds3, hipGetErrorString(e));
            raised3 = true;
        }
        // (short blocks first only when both halves are resident at once — 2 workgroups per CU; with more rounds the longest-first
        //  order of the second generation is the faster one: 8 x 8 x 1024^2 causal 79 against 96 us)
        p.dkv.short_first = (causal && (attn_gen3() & 2) != 0 && (int64_t)kk_cdiv(Sq, 128) * B * heads <= g_attn_cus()) ? 1 : 0;
#ifdef KK_TUNING_HOOKS
        if (a.dbg & (256 | 4096)) { p.dq.DeltaOut = static_cast<float *>(g_attn_trace); p.dkv.DeltaOut = static_cast<float *>(g_attn_trace); }      // (stamp buffer: 8 rows x 64)
#endif
        if (keep != nullptr && p_drop > 0.f && Sk > 128 && kk_attn_keep_bytes(B, heads, Sq, Sk) > 0) {      // (exactly the launches whose forward stored the bits)
            static thread_local bool raised3k = false;
            if (!raised3k) {
                hipError_t e = hipFuncSetAttribute((const void *)attn_bwd_pair3k_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
                if (e != hipSuccess) return kk_fail((int)e, "kk_attn_bwd: cannot reserve %zu bytes of LDS: %s", lds3, hipGetErrorString(e));
                raised3k = true;
            }
            p.dq.keep = p.dkv.keep = const_cast<void *>(keep);
            kk_note_kernel("attn_bwd_pair3k");
            hipLaunchKernelGGL(attn_bwd_pair3k_kernel, dim3(kk_cdiv(Sq, 128), B * heads, 2), dim3(256), lds3, (hipStream_t)stream, p.dq, p.dkv);
            KK_LAUNCH_CHECK("kk_attn_bwd");
            return 0;
        }
        kk_note_kernel("attn_bwd_pair3");
        hipLaunchKernelGGL(attn_bwd_pair3_kernel, dim3(kk_cdiv(Sq, 128), B * heads, 2), dim3(256), lds3, (hipStream_t)stream, p.dq, p.dkv);
        KK_LAUNCH_CHECK("kk_attn_bwd");
        return 0;
    }
#ifdef KK_TUNING_HOOKS
    const size_t lds_bytes = std::max((size_t)2 * 3 * 16384 + 512 + 3 * 16384, (size_t)2 * 3 * (16384 + 512) + 3 * 16384);
    static thread_local bool raised = false;
    if (!raised) {
        hipError_t e = hipFuncSetAttribute((const void *)attn_bwd_pair2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return kk_fail((int)e, "kk_attn_bwd: cannot reserve %zu bytes of LDS: %s", lds_bytes, hipGetErrorString(e));
        raised = true;
    }
    kk_note_kernel("attn_bwd_pair2");
    hipLaunchKernelGGL(attn_bwd_pair2_kernel, dim3(kk_cdiv(Sq, 128), B * heads, 2), dim3(512), lds_bytes, (hipStream_t)stream, p.dq, p.dkv);
    KK_LAUNCH_CHECK("kk_attn_bwd");
    return 0;
#else
    return kk_fail(KK_EINVAL, "kk_attn_bwd: unreachable");
#endif
}

extern "C" int kk_attn_bwd(const float *Q, const float *K, const float *V, const float *dO, const float *LSE, const float *Delta,
                           float *dQ, float *dK, float *dV, int B, int heads, int Sq, int Sk, int64_t ldq, int64_t ldk,
                           int64_t ldv, int64_t lddo, int64_t lddq, int64_t lddk, int64_t lddv, const uint8_t *key_mask,
                           int causal, float scale, const uint32_t *seed, uint32_t site, float p_drop, int math, int io_bf16,
                           const KkAttnHeadNorm *hn_q, const KkAttnHeadNorm *hn_kv, void *stream) {
    return attn_bwd_impl(Q, K, V, dO, LSE, Delta, dQ, dK, dV, B, heads, Sq, Sk, ldq, ldk, ldv, lddo, lddq, lddk, lddv, key_mask, causal,
                         scale, seed, site, p_drop, math, io_bf16, hn_q, hn_kv, nullptr, stream);
}
// kk_attn_bwd reading the keep decisions kk_attn_fwd_kb stored for the SAME launch parameters (seed value, site, p_drop, shape): the
// pair launch then reads bits where it would hash (bit-identical results); every fall-back path ignores `keep` and hashes.
extern "C" int kk_attn_bwd_kb(const float *Q, const float *K, const float *V, const float *dO, const float *LSE, const float *Delta,
                              float *dQ, float *dK, float *dV, int B, int heads, int Sq, int Sk, int64_t ldq, int64_t ldk,
                              int64_t ldv, int64_t lddo, int64_t lddq, int64_t lddk, int64_t lddv, const uint8_t *key_mask,
                              int causal, float scale, const uint32_t *seed, uint32_t site, float p_drop, int math, int io_bf16,
                              const KkAttnHeadNorm *hn_q, const KkAttnHeadNorm *hn_kv, const void *keep, void *stream) {
    KK_REQUIRE(keep == nullptr || al16(keep), "kk_attn_bwd_kb: unaligned keep buffer");
    return attn_bwd_impl(Q, K, V, dO, LSE, Delta, dQ, dK, dV, B, heads, Sq, Sk, ldq, ldk, ldv, lddo, lddq, lddk, lddv, key_mask, causal,
                         scale, seed, site, p_drop, math, io_bf16, hn_q, hn_kv, keep, stream);
}

// Backward in two passes through a caller-owned workspace (see attn_bwd_dkv2s_kernel): the same contract and fall-backs as
// kk_attn_bwd, which is what runs when ws is null / too small or the launch is not eligible for the pair launch either.
extern "C" int64_t kk_attn_bwd_ws_bytes(int B, int heads, int Sq, int Sk) {
    if (B <= 0 || heads <= 0 || Sq <= 0 || Sk <= 0) return 0;
    return (int64_t)B * heads * kk_cdiv(Sk, 32) * (kk_cdiv(Sq, 128) * 4) * 2048;
}
// Whether the two passes are the faster form for this shape (advice to the caller, who hands kk_attn_bwd_ws a workspace only then;
// the entry point itself takes the two passes whenever it gets an adequate workspace and the kernels serve the launch).  The dQ
// pass reads the dS tiles back from the Infinity Cache (~11 B/clk/CU with every CU streaming), 17 us of a 37-47 us launch at
// 8 x 8 x 512^2 — so the two passes lose there and win where the pair launch is long and not lopsided: full attention from 1024^2 up
// (146 -> 128 us; causal 92 -> 90: left to the pair launch, whose halves balance each other).  KK_ATTN_BWD_TWO_PASS=2: always, 0: never.
static int attn_two_pass_mode() {
    static const int v = kk_tune_env("KK_ATTN_BWD_TWO_PASS", 1);
    return v;
}
extern "C" int kk_attn_bwd_two_pass(int B, int heads, int Sq, int Sk, int causal) {
    const int mode = attn_two_pass_mode();
    if (mode == 0 || B <= 0 || heads <= 0 || Sq <= 64 || Sk <= 64 || Sk > 4096 || (causal && Sq != Sk)) return 0;
    if (mode == 2) return 1;
    // (against the second-generation pair launch the two passes won for full attention from 1024^2 up: 142 -> 125 us; the
    //  third-generation pair launch runs that shape in 118 us, so with it the advice is "never")
    return !(attn_gen3() & 1) && !causal && (int64_t)Sq * Sk >= (1ll << 20);
}

extern "C" int kk_attn_bwd_ws(const float *Q, const float *K, const float *V, const float *dO, const float *LSE, const float *Delta,
                              float *dQ, float *dK, float *dV, int B, int heads, int Sq, int Sk, int64_t ldq, int64_t ldk,
                              int64_t ldv, int64_t lddo, int64_t lddq, int64_t lddk, int64_t lddv, const uint8_t *key_mask,
                              int causal, float scale, const uint32_t *seed, uint32_t site, float p_drop, int math, int io_bf16,
                              const KkAttnHeadNorm *hn_q, const KkAttnHeadNorm *hn_kv, void *ws, int64_t ws_bytes, void *stream) {
    KK_REQUIRE(Delta != nullptr, "kk_attn_bwd_ws: Delta is an input of this call");
    KK_REQUIRE((hn_q == nullptr) == (hn_kv == nullptr), "kk_attn_bwd_ws: head-norm epilogues for both kernels or for neither");
    const int two_pass = attn_two_pass_mode() != 0;             // (the shape policy is the caller's: kk_attn_bwd_two_pass)
    const int64_t per_head = (int64_t)kk_cdiv(Sk, 32) * (kk_cdiv(Sq, 128) * 4) * 2048;
    const bool ok = two_pass && ws && al16(ws) && ws_bytes >= kk_attn_bwd_ws_bytes(B, heads, Sq, Sk) && per_head < (1ll << 31) &&
                    io_bf16 && math == KK_MATH_BF16 && (attn_v2_mask() & 6) == 6 && g_attn_groups == 2 && Sk > 64 && Sq > 64 && Sk <= 4096 &&
                    (!causal || Sq == Sk) && al16(Q) && al16(K) && al16(V) && al16(dO) && al16(dQ) && al16(dK) && al16(dV) &&
                    (!hn_q || (al16(hn_q->raw) && (!hn_q->rope || (al16(hn_q->cos_t) && al16(hn_q->sin_t))))) &&
                    (!hn_kv || (al16(hn_kv[0].raw) && al16(hn_kv[1].raw) && !hn_kv[1].rope &&
                                (!hn_kv[0].rope || (al16(hn_kv[0].cos_t) && al16(hn_kv[0].sin_t))))) &&
                    (int64_t)Sk * std::max(ldk, ldv) * 2 < (1ll << 31) && (int64_t)Sq * std::max(ldq, lddo) * 2 < (1ll << 31);
    if (!ok)
        return kk_attn_bwd(Q, K, V, dO, LSE, Delta, dQ, dK, dV, B, heads, Sq, Sk, ldq, ldk, ldv, lddo, lddq, lddk, lddv, key_mask, causal,
                           scale, seed, site, p_drop, math, io_bf16, hn_q, hn_kv, stream);
    KK_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "kk_attn_bwd_ws: dropout probability must be in [0,1)");
    const int64_t lds[7] = {ldq, ldk, ldv, lddo, lddq, lddk, lddv};
    if (int rc = check_common("kk_attn_bwd_ws", B, heads, Sq, Sk, math, lds, 7)) return rc;
    AttnArgs a = {};
    a.Q = Q; a.K = K; a.V = V; a.dO = dO; a.LSE = LSE; a.Delta = Delta; a.Out = dK; a.Out2 = dV; a.key_mask = key_mask;
    a.B = B; a.heads = heads; a.Sq = Sq; a.Sk = Sk; a.causal = causal;
    a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.lddo = lddo; a.ldout = lddk; a.ldout2 = lddv; a.scale = scale;
    a.seed = p_drop > 0.f ? seed : nullptr; a.site = site; a.p_drop = p_drop; a.xcd_map = attn_xcd_map(causal); a.dbg = attn_dbg();
    a.wt = kk_write_through((int64_t)B * std::max(Sq, Sk));
    a.dS = ws;
    if (hn_q) {
        if (int rc = check_headnorm("kk_attn_bwd_ws", hn_q, 1)) return rc;
        if (int rc = check_headnorm("kk_attn_bwd_ws", hn_kv, 2)) return rc;
        a.hn[0] = hn_kv[0]; a.hn[1] = hn_kv[1];
    }
#ifdef KK_TUNING_HOOKS
    if (int rc = (attn_gen3() & 1) ? launch_attn(attn_bwd_dkv3s_kernel, dim3(kk_cdiv(Sk, 128), B * heads), 1, (size_t)71680, (hipStream_t)stream, a)
                                   : launch_attn(attn_bwd_dkv2s_kernel, dim3(kk_cdiv(Sk, 128), B * heads), 2, (size_t)2 * 3 * (16384 + 512) + 3 * 16384, (hipStream_t)stream, a))
        return rc;
#else
    if (int rc = launch_attn(attn_bwd_dkv3s_kernel, dim3(kk_cdiv(Sk, 128), B * heads), 1, (size_t)71680, (hipStream_t)stream, a)) return rc;
#endif
    KK_LAUNCH_CHECK("kk_attn_bwd_ws (dK, dV, dS)");
    a.Out = dQ; a.Out2 = nullptr; a.ldout = lddq; a.ldout2 = 0;
    a.hn[0] = KkAttnHeadNorm{}; a.hn[1] = KkAttnHeadNorm{};
    if (hn_q) a.hn[0] = hn_q[0];
    kk_note_kernel("attn_bwd_dkv3s+dqpass");
    if (int rc = launch_attn(attn_bwd_dqpass_kernel, dim3(kk_cdiv(Sq, 128), B * heads), 1, (size_t)3 * (8192 + 16384), (hipStream_t)stream, a))
        return rc;
    KK_LAUNCH_CHECK("kk_attn_bwd_ws (dQ pass)");
    return 0;
}
#endif  // KK_BODIES_ONLY
