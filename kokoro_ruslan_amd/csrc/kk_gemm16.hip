// bf16-storage MFMA GEMM core for gfx950: operands already bf16 in HBM, fp32 accumulate, LDS filled by the
// buffer-load-to-LDS DMA (16 bytes per lane, no VGPR staging, no ds_write pass).
//
// kk_gemm (kk_gemm.hip) routes to this core when both operands are stored as bf16 and the alignment rules below
// hold; it serves the same three Linear layouts (reference call sites listed at the top of kk_gemm.hip):
//   forward  Y = X.W^T        A k-contiguous, B k-contiguous
//   dgrad    dX = dY.W        A k-contiguous, B k-strided
//   wgrad    dW = dY^T.X      A k-strided,    B k-strided
//
// LDS images (BK = 64 reduction elements per stage, two stages):
//   k-contiguous operand: [rows][64] bf16, 128-byte rows, 16-byte chunk index XOR-ed with (row>>1)&7.  The DMA
//     writes LDS lane-linearly, so the XOR is applied to the GLOBAL source address (a permutation inside one
//     128-byte line: still fully coalesced); MFMA fragments are conflict-free ds_read_b128.
//   k-strided operand: [64 k][rows] bf16 exactly as it lies in memory (rows*2 bytes per k), 32-byte blocks XOR-ed
//     with a function of k; fragments (8 consecutive k per lane) come from two ds_read_b64_tr_b16 — the hardware
//     4x16 transpose read — so no software transpose exists anywhere.
// Out-of-range rows (tile edges, the K tail of a k-strided operand) are zero-filled by the buffer bounds check.
#include "kk_gemm16.h"
#include <algorithm>
#include <stdlib.h>

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
#define LDS_PTR(p) ((__attribute__((address_space(3))) void *)(p))



// One operand tile of ROWS x 64: DMA issue + fragment reads.
template <int ROWS, bool KS, int NT = 256, int AUX = 0> struct Operand {      // AUX: cache policy of the DMA loads (kk_gemm16.h: KK_A_AUX)
    static constexpr int BYTES = ROWS * BK * 2;
    static constexpr int NP = ROWS * 8 / NT;                   // 16-byte pieces per thread per tile (NT threads)
    static constexpr int PITCH = KS ? ROWS * 2 : BK * 2;        // bytes per LDS row
    uint32_t voff[NP];                                          // per-thread byte offset of each piece (tile 0)
    uint32_t kstep;                                             // bytes to advance per k-tile
    __amdgpu_buffer_rsrc_t rsrc;

    __device__ __forceinline__ void init(const void *base, uint32_t bytes, int64_t ld, int r0) {
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
        const int t = threadIdx.x;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int p = t + NT * j;
            if constexpr (!KS) {
                const int row = p >> 3, c = (p & 7) ^ ((row >> 1) & 7);
                voff[j] = (uint32_t)(((int64_t)(r0 + row) * ld + c * 8) * 2);
            } else {
                constexpr int PPR = ROWS / 8;                   // pieces per k-row
                const int k = p / PPR, q = p % PPR;
                const int s = ROWS == 128 ? 2 * (k & 3) : 2 * ((k >> 1) & 1);
                const int g = (((q >> 1) ^ s) << 1) | (q & 1);
                voff[j] = (uint32_t)(((int64_t)k * ld + r0 + g * 8) * 2);
            }
        }
        kstep = KS ? (uint32_t)(ld * BK * 2) : (uint32_t)(BK * 2);
    }
    // start the DMA of k-tile `kt` (absolute tile index) into the LDS image at `dst`
    __device__ __forceinline__ void issue(char *dst, int kt, int wave) const {
        const uint32_t so = (uint32_t)kt * kstep;
#pragma unroll
        for (int j = 0; j < NP; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(dst + (wave * 64 + NT * j) * 16), 16, voff[j], so, 0, AUX);
    }
};

struct Frag {
    bf16x8 v;            // k-contiguous operand
    s16x4 lo, hi;        // k-strided operand: k 0..3 and 4..7 of this lane's eight
};

// per-lane constants of the fragment reads
template <int ROWS, bool KS> struct FragAddr {
    uint32_t base;          // byte offset inside the image
    uint32_t x[4];          // KC: chunk offsets per ks;  KS: block offsets per 32-row block (ROWS/64 used... up to 4)
    __device__ __forceinline__ void init(int lane, int wave_row0) {
        const int l31 = lane & 31, half = lane >> 5;
        if constexpr (!KS) {
            const int swz = (l31 >> 1) & 7;
            base = (uint32_t)((wave_row0 + l31) * (BK * 2));
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) x[ks] = (uint32_t)(((2 * ks + half) ^ swz) * 16);
        } else {
            const int L = lane & 15, gi = (lane >> 4) & 1, kq = L >> 2;
            const int s = ROWS == 128 ? 2 * (kq & 3) : 2 * ((kq >> 1) & 1);
            base = (uint32_t)((8 * half + kq) * (ROWS * 2) + 8 * (L & 3));
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) x[rb] = (uint32_t)((((wave_row0 >> 4) + 2 * rb + gi) ^ s) * 32);
        }
    }
    // Fragment for 32-row block rb (relative to the wave's first row), k-slab ks (16 k): one ds_read_b128 (k-contiguous)
    // or two transpose reads (k-strided).  All of them are inline asm: the k-loop keeps the reads of the NEXT slabs in
    // flight under the MFMAs of this one and retires them with counted lgkmcnt waits (wait_slab), which only works
    // when every LDS read of the loop is in program order under our control.  (Through the builtin, hipcc also puts
    // an s_waitcnt vmcnt(0) in front of every transpose read while a DMA is in flight.)
    static constexpr int READS = KS ? 2 : 1;                    // LDS instructions per fragment
    __device__ __forceinline__ void load(Frag &f, const char *img, int rb, int ks) const {
        if constexpr (!KS) {
            const uint32_t addr = (uint32_t)(uintptr_t)LDS_PTR(img) + base + x[ks] + rb * 32 * (BK * 2);
            asm volatile("ds_read_b128 %0, %1" : "=v"(f.v) : "v"(addr));
        } else {
            const uint32_t addr = (uint32_t)(uintptr_t)LDS_PTR(img) + base + x[rb] + ks * 16 * (ROWS * 2);
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.lo) : "v"(addr));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.hi) : "v"(addr), "n"(4 * ROWS * 2));
        }
    }
};

__device__ __forceinline__ bf16x8 frag_value(const Frag &f, bool ks) {
    if (!ks) return f.v;
    s16x8 v;
    v[0] = f.lo[0]; v[1] = f.lo[1]; v[2] = f.lo[2]; v[3] = f.lo[3]; v[4] = f.hi[0]; v[5] = f.hi[1]; v[6] = f.hi[2]; v[7] = f.hi[3];
    return __builtin_bit_cast(bf16x8, v);
}
// Wait until at most PENDING younger LDS reads are outstanding (they return in order), then pin the slab's fragments
// behind the wait: the empty asm makes their registers data-dependent on this point, so no MFMA is scheduled above it.
template <int PENDING> __device__ __forceinline__ void wait_reads() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(PENDING) : "memory"); }
__device__ __forceinline__ void pin_frag(Frag &f, bool ks) {
    if (ks) asm volatile("" : "+v"(f.lo), "+v"(f.hi));
    else asm volatile("" : "+v"(f.v));
}

// NS LDS stages: NS-1 k-tiles are in flight while one is multiplied.  The DMA of a tile is waited for with a COUNTED
// vmcnt (the younger tiles stay in flight across the barrier), and the barrier is a raw s_barrier: __syncthreads()
// would drain vmcnt(0) because an LDS-DMA is a pending LDS write.
// EPI = 1: the GEMM is dG = dY.W2 of a GLU feed-forward (N = F columns) and the epilogue is the gate's backward
// (transformers.py:107-108): with h1 = [a | b] saved by the forward and m the gate's dropout mask,
//   dh1[:, c] = dG*m * b * gelu'(a),   dh1[:, F + c] = dG*m * gelu(a)
// are written directly (dG never exists in HBM), and the column sums of dh1 — linear1's bias gradient — leave the
// workgroup as plain rows partials[2*tile_m + wave_row][2F] for kk_partials_reduce.  Replaces kk_glu_bwd + kk_colsum_acc.
// EPI = 3: the GEMM is a q / k / v projection whose heads are 64 wide, so a 64x64 tile holds whole (row, head) vectors:
// the epilogue writes the projection (saved for the backward) AND its per-head RMSNorm (+ RoPE) — the attention's
// operands — through an LDS transpose of the tile.  Replaces kk_headnorm_rope_fwd (same math, same bits).
// EPI = 2: the GEMM is h1 = x.W1^T + b1 of a GLU feed-forward; a workgroup owns output columns [n0, n0+64) AND
// [F+n0, F+n0+64) (two B panels, two accumulators, the A tile is read from LDS once for both), so its epilogue writes
// h1 = [a | b] (saved for the backward) and the gated product g = gelu(a)*b*mask in one go.  Replaces kk_glu_fwd.
template <bool TA, bool TB, int BM, int BN, int NS, int EPI, int WAVES = 4, int WC = 2>
__device__ __forceinline__ void gemm16_body(const G16Args &a, const int wg, char *smem) {
    constexpr int WR = WAVES / WC;                              // waves along M x waves along N
    constexpr int MI = BM / (32 * WR), NI = BN / (32 * WC);     // 32x32 MFMA tiles per wave (wave tile = BM/WR x BN/WC)
    static_assert(EPI == 0 || WAVES == 4 || (WAVES == 8 && BM == 128 && BN == 64),
                  "the epilogue variants are written for four waves and for the eight-wave 128x64 tile");
    using OA = Operand<BM, TA, 64 * WAVES, KK_A_AUX>;
    using OB = Operand<BN, TB, 64 * WAVES>;
    constexpr int NB = EPI == 2 ? 2 : 1;                        // EPI == 2 multiplies A with TWO 64-row panels of B (see below)
    constexpr int STAGE = OA::BYTES + NB * OB::BYTES;
    constexpr int NPT = OA::NP + NB * OB::NP;                   // DMA instructions per thread per k-tile

    // Workgroup -> (tile, k-slice).  The dispatcher places workgroup i on XCD i % 8 (private 4 MiB L2 each).
    //  tile-major (default): every XCD sweeps a contiguous run of tiles (n fastest), all k-slices of a tile together;
    //  split-major (option, split-K with a multiple of 8 slices): slice = i % splits, so XCD x owns the k-slices
    //    = x (mod 8) of every tile and reads its part of A and B from HBM exactly once.  It cuts FETCH_SIZE of the
    //    512x512x4096 weight gradients 4.5x, yet the train step is 2 % SLOWER with it (566K vs 579K frames/s): these
    //    launches are latency-bound, not HBM-bound, and a tile's atomics then come from eight XCDs.  Left off.
    int tid_lin, ksl;
    if (a.split_major) {
        ksl = wg % a.splits;
        tid_lin = wg / a.splits;
    } else {
        const int ntiles = a.tiles_m * a.tiles_n;
        tid_lin = wg % ntiles;
        ksl = wg / ntiles;
        if (a.xcd_swizzle) {                                    // bijective for any tile count (see kk_gemm.hip)
            const int q = ntiles >> 3, r = ntiles & 7, xcd = tid_lin & 7, in = tid_lin >> 3;
            tid_lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + in;
        }
    }
    // An XCD's contiguous run of tiles covers a few rows of the tile grid in the FAST direction completely: it streams the whole
    // operand of that direction through its private L2 (all eight L2s do) and a slice of the other one.  n fastest: B re-read 8x,
    // A once; m fastest: the other way round.  The grouped weight gradients pick the direction that replicates the SMALLER
    // operand (linear2's dW is 512 x 1536: m fastest re-reads dY 8x = 32 MB instead of the gated activations 8x = 100 MB).
    const int m0 = (a.m_fast ? tid_lin % a.tiles_m : tid_lin / a.tiles_n) * BM;
    const int n0 = (a.m_fast ? tid_lin / a.tiles_m : tid_lin % a.tiles_n) * BN;
    const int kbeg = ksl * a.k_per_split;
    const int kend = min(a.K, kbeg + a.k_per_split);
    const int nk = (kend - kbeg + BK - 1) / BK, kt0 = kbeg / BK;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave / WC, wc = wave % WC, half = lane >> 5, l31 = lane & 31;

    OA oa;
    OB ob, ob2;
    oa.init(a.A, a.a_bytes, a.lda, m0);
    ob.init(a.B, a.b_bytes, a.ldb, n0);
    if constexpr (EPI == 2) ob2.init(a.B, a.b_bytes, a.ldb, n0 + a.N);
    FragAddr<BM, TA> fa;
    FragAddr<BN, TB> fb;
    fa.init(lane, wr * (BM / WR));
    fb.init(lane, wc * (BN / WC));

    f32x16 acc[MI][NI];
    f32x16 acc2;                                                // EPI == 2: the second B panel's accumulator (MI = NI = 1)
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[r] = 0.f;

#pragma unroll
    for (int p = 0; p < NS - 1; ++p)
        if (p < nk) {
            oa.issue(smem + p * STAGE, kt0 + p, wave);
            ob.issue(smem + p * STAGE + OA::BYTES, kt0 + p, wave);
            if constexpr (EPI == 2) ob2.issue(smem + p * STAGE + OA::BYTES + OB::BYTES, kt0 + p, wave);
        }
    int sc = 0, sn = NS - 1;                                    // stage being multiplied / stage being refilled
    for (int kt = 0; kt < nk; ++kt) {
        // this wave's pieces of tile kt have landed once at most the younger tiles' DMAs are outstanding
        const int younger = min(nk - 1 - kt, NS - 2);
        if (NS >= 4 && younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NPT) : "memory");
        else if (NS >= 3 && younger >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPT) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                           // everyone's pieces landed; everyone finished reading stage sn
        asm volatile("" ::: "memory");
        if (kt + NS - 1 < nk) {
            oa.issue(smem + sn * STAGE, kt0 + kt + NS - 1, wave);
            ob.issue(smem + sn * STAGE + OA::BYTES, kt0 + kt + NS - 1, wave);
            if constexpr (EPI == 2) ob2.issue(smem + sn * STAGE + OA::BYTES + OB::BYTES, kt0 + kt + NS - 1, wave);
        }
        const char *cur = smem + sc * STAGE;
        sn = sc;
        sc = sc + 1 == NS ? 0 : sc + 1;
        const char *Ai = cur, *Bi = cur + OA::BYTES;
        // The four 16-k slabs of the tile, software-pipelined inside the wave: the LDS reads of slabs ks+1..ks+AHEAD
        // are in flight while slab ks is multiplied (the lgkmcnt counter holds 15, hence AHEAD by reads per slab).
        constexpr int RPS = MI * FragAddr<BM, TA>::READS + (NI + (EPI == 2 ? 1 : 0)) * FragAddr<BN, TB>::READS;
        constexpr int AHEAD = 3 * RPS <= 15 ? 2 : (2 * RPS <= 15 ? 1 : 0);
        Frag af[4][MI], bf[4][NI], bf2[4];
        auto read_slab = [&](int ks) {
#pragma unroll
            for (int i = 0; i < MI; ++i) fa.load(af[ks][i], Ai, i, ks);
#pragma unroll
            for (int j = 0; j < NI; ++j) fb.load(bf[ks][j], Bi, j, ks);
            if constexpr (EPI == 2) fb.load(bf2[ks], Bi + OB::BYTES, 0, ks);
        };
#pragma unroll
        for (int ks = 0; ks < AHEAD; ++ks) read_slab(ks);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks + AHEAD < 4) read_slab(ks + AHEAD);
            if (ks + AHEAD < 4) wait_reads<AHEAD * RPS>();
            else if (ks + 1 < 4 && AHEAD == 2 && ks == 2) wait_reads<RPS>();
            else wait_reads<0>();
#pragma unroll
            for (int i = 0; i < MI; ++i) pin_frag(af[ks][i], TA);
#pragma unroll
            for (int j = 0; j < NI; ++j) pin_frag(bf[ks][j], TB);
            if constexpr (EPI == 2) pin_frag(bf2[ks], TB);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_value(af[ks][i], TA), frag_value(bf[ks][j], TB), acc[i][j], 0, 0, 0);
            if constexpr (EPI == 2)
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_value(af[ks][0], TA), frag_value(bf2[ks], TB), acc2, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);                  // keep the slab's MFMAs here, between the waits
        }
    }
    if (nk <= 0) return;

    if constexpr (EPI == 1) {
        static_assert(EPI == 0 || (BM / WR == 32 && BN / WC == 32), "the GLU epilogues expect one 32x32 MFMA tile per wave");
        // The epilogue moves 8 bytes per output element (h1 = [a | b] in, dh1 out) — as much HBM traffic as the GEMM itself.
        // In the accumulator layout a lane owns ONE column of 16 rows: 64 two-byte accesses per lane, 64-byte segments.  So
        // the wave's 32x32 tile goes through LDS once and a lane works on 8 consecutive columns of 2 rows: 16-byte loads
        // and stores, eight of them per lane.
        const int F = a.N;
        const uint32_t thr = a.glu_seed ? kk_drop_threshold(a.glu_p) : 0u, seed = thr ? *a.glu_seed : 0u;
        const float ik = thr ? 1.f / (1.f - a.glu_p) : 1.f;
        constexpr int TP = 36;                                  // floats per tile row (16-byte aligned rows)
        __builtin_amdgcn_s_barrier();                           // every wave is done with the last stage
        float *tile = reinterpret_cast<float *>(smem) + wave * 32 * TP;
#pragma unroll
        for (int r = 0; r < 16; ++r) tile[frag_row(r, half) * TP + l31] = acc[0][0][r];
        __builtin_amdgcn_wave_barrier();                        // (one wave: its LDS operations complete in order)
        const int c8 = (lane & 3) * 8, col = n0 + wc * 32 + c8;
        float sa[8], sb[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) sa[j] = sb[j] = 0.f;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int rl = it * 16 + (lane >> 2), row = m0 + wr * 32 + rl;
            if (row < a.M && col < F) {
                const float4 d0 = ld4(tile + rl * TP + c8), d1 = ld4(tile + rl * TP + c8 + 4);
                const float d[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
                const int64_t o = (int64_t)row * 2 * F + col;
                const bf16x8 av = *reinterpret_cast<const bf16x8 *>(a.glu_h + o), bv = *reinterpret_cast<const bf16x8 *>(a.glu_h + o + F);
                float mk[8];
                kk_drop_mul4(seed, a.glu_site, (uint64_t)row * F + col, thr, ik, *reinterpret_cast<float(*)[4]>(mk));
                kk_drop_mul4(seed, a.glu_site, (uint64_t)row * F + col + 4, thr, ik, *reinterpret_cast<float(*)[4]>(mk + 4));
                bf16x8 oa, ob;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float gv, gd;
                    kk_gelu_pair_fast((float)av[j], gv, gd);
                    const float dd = d[j] * mk[j];
                    const float da = dd * (float)bv[j] * gd, db = dd * gv;
                    oa[j] = (__bf16)da;
                    ob[j] = (__bf16)db;
                    sa[j] += da;
                    sb[j] += db;
                }
                kk_store16(a.glu_dh + o, __builtin_bit_cast(kk_u32x4, oa), a.wt);
                kk_store16(a.glu_dh + o + F, __builtin_bit_cast(kk_u32x4, ob), a.wt);
            }
        }
        // column sums over the wave's 32 rows: the 16 lanes that share (lane & 3)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            {   // lanes 4 and 8 away inside the 16-lane row by DPP rotations (only lanes 0..3 are read below: for them the same additions as
                // the xor butterfly), the rows 16 and 32 away by ds_bpermute
                sa[j] += kk_dpp<0x124>(sa[j]); sb[j] += kk_dpp<0x124>(sb[j]);
                sa[j] += kk_dpp<0x128>(sa[j]); sb[j] += kk_dpp<0x128>(sb[j]);
                sa[j] += __shfl_xor(sa[j], 16, 64); sb[j] += __shfl_xor(sb[j], 16, 64);
                sa[j] += __shfl_xor(sa[j], 32, 64); sb[j] += __shfl_xor(sb[j], 32, 64);
            }
        }
        const int prow = (m0 + wr * 32) / 32;                   // one partial row per 32 rows of dY, kk_gemm_dgrad_glu_blocks(T) of them
        if (lane < 4 && col < F && prow < 2 * ((a.M + 63) / 64)) {
            float *pr = a.glu_partials + (int64_t)prow * 2 * F;
            st4(pr + col, make_float4(sa[0], sa[1], sa[2], sa[3]));
            st4(pr + col + 4, make_float4(sa[4], sa[5], sa[6], sa[7]));
            st4(pr + F + col, make_float4(sb[0], sb[1], sb[2], sb[3]));
            st4(pr + F + col + 4, make_float4(sb[4], sb[5], sb[6], sb[7]));
        }
        return;
    }
    if constexpr (EPI == 3) {
        static_assert(EPI != 3 || (BN == 64 && BM / WR == 32 && BN / WC == 32), "the head-norm epilogue expects one 32x32 MFMA tile per wave, 64 columns");
        constexpr int PITCH = 72;                               // bf16 per LDS row: 144 B, rows land on different banks
        __bf16 *tile = reinterpret_cast<__bf16 *>(smem);
        __builtin_amdgcn_s_barrier();                           // every wave is done with the last stage
        {
            const int col = wc * 32 + l31;
            const float bv = a.bias ? a.bias[n0 + col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) tile[(wr * 32 + frag_row(r, half)) * PITCH + col] = (__bf16)(acc[0][0][r] + bv);
        }
        __syncthreads();
        const int sub = threadIdx.x & 15, part = n0 / a.hn_H;
        const bool rope = (a.hn_rope_mask >> part) & 1;
        const float4 g = ld4(a.hn_gain[part] + sub * 4);
        __bf16 *raw = static_cast<__bf16 *>(a.C);
        constexpr int RPI = 4 * WAVES;                          // rows per pass: 16 threads per (row, head) vector
#pragma unroll
        for (int it = 0; it < BM / RPI; ++it) {
            const int rl = it * RPI + (threadIdx.x >> 4), row = m0 + rl;
            const bf16x4 r4 = *reinterpret_cast<const bf16x4 *>(tile + rl * PITCH + sub * 4);
            const float4 v = make_float4((float)r4[0], (float)r4[1], (float)r4[2], (float)r4[3]);
            const int pos = rope ? (row < a.M ? row : a.M - 1) % a.hn_S : 0;
            const float4 n = kk_headnorm_rope(v, g, rope, a.hn_cos + pos * 64, a.hn_sin + pos * 64, sub);
            if (row < a.M) {
                kk_store8(raw + (int64_t)row * a.ldc + n0 + sub * 4, __builtin_bit_cast(kk_u32x2, r4), a.wt);
                bf16x4 n4;
                n4[0] = (__bf16)n.x; n4[1] = (__bf16)n.y; n4[2] = (__bf16)n.z; n4[3] = (__bf16)n.w;
                kk_store8(a.hn_y + (int64_t)row * a.hn_ldy + n0 + sub * 4, __builtin_bit_cast(kk_u32x2, n4), a.wt);
            }
        }
        return;
    }
    if constexpr (EPI == 2) {
        // (same transposition as EPI == 1: both accumulators through LDS, a lane stores 8 consecutive columns of 2 rows)
        const int F = a.N;
        const uint32_t thr = a.glu_seed ? kk_drop_threshold(a.glu_p) : 0u, seed = thr ? *a.glu_seed : 0u;
        const float ik = thr ? 1.f / (1.f - a.glu_p) : 1.f;
        constexpr int TP = 36;
        __builtin_amdgcn_s_barrier();                           // every wave is done with the last stage
        float *ta_ = reinterpret_cast<float *>(smem) + wave * 2 * 32 * TP, *tb_ = ta_ + 32 * TP;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            ta_[frag_row(r, half) * TP + l31] = acc[0][0][r];
            tb_[frag_row(r, half) * TP + l31] = acc2[r];
        }
        __builtin_amdgcn_wave_barrier();
        const int c8 = (lane & 3) * 8, col = n0 + wc * 32 + c8;
        if (col >= F) return;
        float ba[8], bb[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { ba[j] = a.bias ? a.bias[col + j] : 0.f; bb[j] = a.bias ? a.bias[F + col + j] : 0.f; }
        __bf16 *h = a.glu_dh, *g = static_cast<__bf16 *>(a.C);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int rl = it * 16 + (lane >> 2), row = m0 + wr * 32 + rl;
            if (row >= a.M) continue;
            const float4 a0 = ld4(ta_ + rl * TP + c8), a1 = ld4(ta_ + rl * TP + c8 + 4);
            const float4 b0 = ld4(tb_ + rl * TP + c8), b1 = ld4(tb_ + rl * TP + c8 + 4);
            const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w}, bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
            float mk[8];
            kk_drop_mul4(seed, a.glu_site, (uint64_t)row * F + col, thr, ik, *reinterpret_cast<float(*)[4]>(mk));
            kk_drop_mul4(seed, a.glu_site, (uint64_t)row * F + col + 4, thr, ik, *reinterpret_cast<float(*)[4]>(mk + 4));
            bf16x8 oa, ob, og;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                oa[j] = (__bf16)(av[j] + ba[j]);                 // what the backward will read
                ob[j] = (__bf16)(bv[j] + bb[j]);
                og[j] = (__bf16)(kk_gelu_fast((float)oa[j]) * (float)ob[j] * mk[j]);
            }
            const int64_t o = (int64_t)row * 2 * F + col;
            kk_store16(h + o, __builtin_bit_cast(kk_u32x4, oa), a.wt);
            kk_store16(h + o + F, __builtin_bit_cast(kk_u32x4, ob), a.wt);
            kk_store16(g + (int64_t)row * a.ldc + col, __builtin_bit_cast(kk_u32x4, og), a.wt);
        }
        return;
    }
    const bool lead = (ksl == 0);
    // bf16 C without accumulation or residual (most dgrads, linear2): the tile goes through LDS so that a lane stores 8
    // consecutive columns (16 bytes) of 2 rows instead of 16 two-byte values of one column
    constexpr bool WIDE_OK = NS * STAGE >= WAVES * 32 * 36 * 4 + 512;      // the staging area holds one 32x32 fp32 tile per wave (+ the Delta rows)
    constexpr bool DELTA_OK = EPI == 0 && WAVES == 8 && WC == 2 && BM == 128 && BN == 64;      // one 32x32 tile per wave, a head per workgroup
    if (WIDE_OK && a.c_bf16 && a.residual == nullptr && (a.ldc & 7) == 0 && (a.N & 7) == 0 && (reinterpret_cast<uintptr_t>(a.C) & 15) == 0) {
        constexpr int TP = 36;
        __builtin_amdgcn_s_barrier();                           // every wave is done with the last stage
        float *tile = reinterpret_cast<float *>(smem) + wave * 32 * TP;
        __bf16 *C = static_cast<__bf16 *>(a.C);
        const int c8 = (lane & 3) * 8;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) {
#pragma unroll
                for (int r = 0; r < 16; ++r) tile[frag_row(r, half) * TP + l31] = acc[i][j][r];
                __builtin_amdgcn_wave_barrier();
                const int col = n0 + wc * (BN / WC) + j * 32 + c8;
                float dsum[2] = {0.f, 0.f};                   // Delta epilogue: this lane's 8 columns of its 2 rows
                if (col < a.N) {
                    float bv[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) bv[e] = (a.bias != nullptr && lead) ? a.bias[col + e] : 0.f;
#pragma unroll
                    for (int it = 0; it < 2; ++it) {
                        const int rl = it * 16 + (lane >> 2), row = m0 + wr * (BM / WR) + i * 32 + rl;
                        if (row >= a.M) continue;
                        const float4 v0 = ld4(tile + rl * TP + c8), v1 = ld4(tile + rl * TP + c8 + 4);
                        const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                        bf16x8 o;
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = (__bf16)(a.alpha * v[e] + bv[e]);
                        kk_store16(C + (int64_t)row * a.ldc + col, __builtin_bit_cast(kk_u32x4, o), a.wt);
                        if constexpr (DELTA_OK) {
                            if (a.dl_out != nullptr) {              // (from the ROUNDED dO: what the attention kernels will read)
                                const bf16x8 ov = *reinterpret_cast<const bf16x8 *>(a.dl_o + (int64_t)row * a.dl_ldo + col);
#pragma unroll
                                for (int e = 0; e < 8; ++e) dsum[it] += (float)o[e] * (float)ov[e];
                            }
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
                if constexpr (DELTA_OK) {
                    if (a.dl_out != nullptr) {                      // (workgroup-uniform; N % 64 == 0 is checked by the entry point)
                        // a row's 32 columns of this wave: the 4 lanes that share lane >> 2; its other 32 are in wave wc ^ 1
                        float *red = reinterpret_cast<float *>(smem + WAVES * 32 * TP * 4);      // [WR][32] row sums of the wc = 1 waves
#pragma unroll
                        for (int it = 0; it < 2; ++it) {
                            dsum[it] += __shfl_xor(dsum[it], 1, 64);
                            dsum[it] += __shfl_xor(dsum[it], 2, 64);
                        }
                        if (wc == 1 && (lane & 3) == 0) {
                            red[wr * 32 + (lane >> 2)] = dsum[0];
                            red[wr * 32 + 16 + (lane >> 2)] = dsum[1];
                        }
                        __syncthreads();
                        if (wc == 0 && (lane & 3) == 0) {
#pragma unroll
                            for (int it = 0; it < 2; ++it) {
                                const int rl = it * 16 + (lane >> 2), row = m0 + wr * 32 + rl;
                                if (row < a.M) {
                                    const int bb = row / a.dl_S, q = row - bb * a.dl_S;
                                    a.dl_out[((int64_t)bb * a.dl_heads + n0 / 64) * a.dl_S + q] = dsum[it] + red[wr * 32 + rl];
                                }
                            }
                        }
                    }
                }
            }
        return;
    }
    // fp32 C written (or accumulated into) exactly once per element — the weight gradients — in a write-through launch: the same
    // transposition, so that a lane stores 8 consecutive columns of 2 rows as 16-byte write-through stores (a launch leaves up to
    // 31 MB of dW behind; as plain 4-byte stores they sit dirty in the L2s until the kernel boundary writes them back)
    if (a.wt && WIDE_OK && !a.c_bf16 && !a.atomic && a.residual == nullptr && (a.ldc & 3) == 0 && (a.N & 7) == 0 &&
        (reinterpret_cast<uintptr_t>(a.C) & 15) == 0) {
        constexpr int TP = 36;
        __builtin_amdgcn_s_barrier();                           // every wave is done with the last stage
        float *tile = reinterpret_cast<float *>(smem) + wave * 32 * TP;
        float *C = static_cast<float *>(a.C);
        const int c8 = (lane & 3) * 8;
        float ssq = 0.f;                                         // (ss_rec: this lane's share of the tile's sum of squares)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) {
#pragma unroll
                for (int r = 0; r < 16; ++r) tile[frag_row(r, half) * TP + l31] = acc[i][j][r];
                __builtin_amdgcn_wave_barrier();
                const int col = n0 + wc * (BN / WC) + j * 32 + c8;
                if (col < a.N) {
                    float bv[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) bv[e] = (a.bias != nullptr && lead) ? a.bias[col + e] : 0.f;
#pragma unroll
                    for (int it = 0; it < 2; ++it) {
                        const int rl = it * 16 + (lane >> 2), row = m0 + wr * (BM / WR) + i * 32 + rl;
                        if (row >= a.M) continue;
                        float *dst = C + (int64_t)row * a.ldc + col;
                        const float4 v0 = ld4(tile + rl * TP + c8), v1 = ld4(tile + rl * TP + c8 + 4);
                        float4 o0 = make_float4(a.alpha * v0.x + bv[0], a.alpha * v0.y + bv[1], a.alpha * v0.z + bv[2], a.alpha * v0.w + bv[3]);
                        float4 o1 = make_float4(a.alpha * v1.x + bv[4], a.alpha * v1.y + bv[5], a.alpha * v1.z + bv[6], a.alpha * v1.w + bv[7]);
                        if (a.beta != 0.f) {
                            const float4 d0 = ld4(dst), d1 = ld4(dst + 4);
                            o0 = make_float4(o0.x + a.beta * d0.x, o0.y + a.beta * d0.y, o0.z + a.beta * d0.z, o0.w + a.beta * d0.w);
                            o1 = make_float4(o1.x + a.beta * d1.x, o1.y + a.beta * d1.y, o1.z + a.beta * d1.z, o1.w + a.beta * d1.w);
                        }
                        kk_st16_wt(dst, __builtin_bit_cast(kk_u32x4, o0));
                        kk_st16_wt(dst + 4, __builtin_bit_cast(kk_u32x4, o1));
                        ssq += (o0.x * o0.x + o0.y * o0.y) + (o0.z * o0.z + o0.w * o0.w) + (o1.x * o1.x + o1.y * o1.y) + (o1.z * o1.z + o1.w * o1.w);
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        if (a.ss_rec != nullptr) {                              // (workgroup-uniform; splits == 1 by the caller) wave sums added in wave order
            double *wsum = reinterpret_cast<double *>(smem + WAVES * 32 * TP * 4);
            const double wv = wave_sum_d((double)ssq);
            if (lane == 0) wsum[wave] = wv;
            __syncthreads();
            if (threadIdx.x == 0) {
                double t = 0.0;
#pragma unroll
                for (int w = 0; w < WAVES; ++w) t += wsum[w];
                a.ss_rec[tid_lin] = KkSegRec{t, a.ss_seg + (a.ss_rows > 0 ? m0 / a.ss_rows : 0), 0};
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int col = n0 + wc * (BN / WC) + j * 32 + l31;
            if (col >= a.N) continue;
            const float bv = (a.bias != nullptr && lead) ? a.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wr * (BM / WR) + i * 32 + frag_row(r, half);
                if (row >= a.M) continue;
                float v = a.alpha * acc[i][j][r] + bv;
                if (a.residual != nullptr && lead) {
                    const int64_t rr = a.res_mod > 0 ? (int64_t)row % a.res_mod : (int64_t)row;
                    v += a.residual[rr * a.ldr + col];
                }
                if (a.c_bf16) {
                    static_cast<__bf16 *>(a.C)[(int64_t)row * a.ldc + col] = (__bf16)v;
                    continue;
                }
                float *dst = static_cast<float *>(a.C) + (int64_t)row * a.ldc + col;
                if (a.atomic) {
                    atomicAdd(dst, v);
                } else {
                    if (a.beta != 0.f) v += a.beta * (*dst);
                    *dst = v;
                }
            }
        }
}

#ifdef KK_BODIES_ONLY
}  // namespace   (kk_chain.hip includes this file for gemm16_body only)
#else
template <bool TA, bool TB, int BM, int BN, int NS, int EPI = 0>
__global__ __launch_bounds__(256) void gemm16_kernel(G16Args a) {
    __shared__ __attribute__((aligned(16))) char smem[NS * (BM + (EPI == 2 ? 2 : 1) * BN) * BK * 2];
    gemm16_body<TA, TB, BM, BN, NS, EPI>(a, blockIdx.x, smem);
}

// The same body run by EIGHT waves on the 128x64 tile (a 4x2 wave grid, one 32x32 MFMA chain per wave): twice the waves per
// byte of LDS, which is what these latency-bound launches respond to (see the grouped weight gradients below).
template <bool TA, bool TB, int NS>
__global__ __launch_bounds__(512) void gemm16_kernel_w8(G16Args a) {
    __shared__ __attribute__((aligned(16))) char smem[NS * (128 + 64) * BK * 2];
    gemm16_body<TA, TB, 128, 64, NS, 0, 8, 2>(a, blockIdx.x, smem);
}
template __global__ void gemm16_kernel_w8<false, false, 2>(G16Args);
template __global__ void gemm16_kernel_w8<false, true, 2>(G16Args);
template __global__ void gemm16_kernel_w8<true, false, 2>(G16Args);
template __global__ void gemm16_kernel_w8<true, true, 2>(G16Args);
template __global__ void gemm16_kernel_w8<false, false, 3>(G16Args);
template __global__ void gemm16_kernel_w8<false, true, 3>(G16Args);
template __global__ void gemm16_kernel_w8<true, false, 3>(G16Args);
template __global__ void gemm16_kernel_w8<true, true, 3>(G16Args);
// ... and the two GLU epilogues on that tile (linear1 + gate: two B panels, 96 KB of LDS; linear2 dgrad + gate backward)
template <bool TB, int NS, int EPI>
__global__ __launch_bounds__(512) void gemm16_kernel_w8_glu(G16Args a) {
    __shared__ __attribute__((aligned(16))) char smem[NS * (128 + (EPI == 2 ? 2 : 1) * 64) * BK * 2];
    gemm16_body<false, TB, 128, 64, NS, EPI, 8, 2>(a, blockIdx.x, smem);
}
// EPI 3 on the eight-wave tile (KK_G16_W8_HN=0: the four-wave 64x64 form)
template <int NS>
__global__ __launch_bounds__(512) void gemm16_kernel_w8_hn(G16Args a) {
    __shared__ __attribute__((aligned(16))) char smem[NS * (128 + 64) * BK * 2];
    gemm16_body<false, false, 128, 64, NS, 3, 8, 2>(a, blockIdx.x, smem);
}
template __global__ void gemm16_kernel_w8_hn<3>(G16Args);
template __global__ void gemm16_kernel_w8_glu<false, 3, 2>(G16Args);
template __global__ void gemm16_kernel_w8_glu<true, 3, 1>(G16Args);
int g16_w8_hn = kk_tune_env("KK_G16_W8_HN", 1);        // measured: +0.8 % on the step (interleaved A/B)
int g16_w8_glu = kk_tune_env("KK_G16_W8_GLU", 1);      // bit 0: dgrad + GLU backward, bit 1: linear1 + GLU; 4th part, interleaved: bit 0 +0.3 % (on), bit 1 -0.5 % (off)

template <int NS>
void launch_w8(int ta, int tb, const G16Args &a, dim3 grid, hipStream_t s) {
    kk_note_kernelf("gemm16_w8<%d,%d,%d>", ta, tb, NS);
    if (kk_capture(kk_last_kernel(), a, grid, 512, 0)) return;
    if (!ta && !tb) hipLaunchKernelGGL((gemm16_kernel_w8<false, false, NS>), grid, dim3(512), 0, s, a);
    else if (!ta && tb) hipLaunchKernelGGL((gemm16_kernel_w8<false, true, NS>), grid, dim3(512), 0, s, a);
    else if (ta && !tb) hipLaunchKernelGGL((gemm16_kernel_w8<true, false, NS>), grid, dim3(512), 0, s, a);
    else hipLaunchKernelGGL((gemm16_kernel_w8<true, true, NS>), grid, dim3(512), 0, s, a);
}
// 8-wave 128x64 tiles, 3 stages, for single GEMMs with >= g16_thr12864 such tiles: +0.8 % on the 8x512 step, +2.2 % at 8x1024
// (interleaved A/B, tools/probes/g16w8.sh); KK_G16_W8=0 restores the 4-wave form, =2 two stages.
int g16_w8 = kk_tune_env("KK_G16_W8", 3);

// Several independent GEMMs of one operand layout in ONE launch (a layer's weight gradients: they have no consumer
// before the optimizer, so they wait until the layer's backward is through and then fill the chip together — ~1000
// 64x64 tiles with the full reduction length each, no split-K atomics, one launch instead of four to six).
template <bool TA, bool TB, int BM, int BN, int NS, int WAVES = 4, int WC = 2>
__global__ __launch_bounds__(64 * WAVES) void gemm16_group_kernel(G16Group g) {
    __shared__ __attribute__((aligned(16))) char smem[NS * (BM + BN) * BK * 2];
    int i = 0;
    while (i + 1 < g.n && (int)blockIdx.x >= g.start[i + 1]) ++i;
    gemm16_body<TA, TB, BM, BN, NS, 0, WAVES, WC>(g.p[i], (int)blockIdx.x - g.start[i], smem);
}
template __global__ void gemm16_group_kernel<true, true, 128, 128, 2, 16, 4>(G16Group);
template __global__ void gemm16_group_kernel<true, true, 128, 64, 2, 8>(G16Group);
template __global__ void gemm16_group_kernel<true, true, 128, 128, 2, 8>(G16Group);
template __global__ void gemm16_group_kernel<true, true, 64, 64, 2>(G16Group);
template __global__ void gemm16_group_kernel<true, true, 64, 64, 3>(G16Group);
template __global__ void gemm16_group_kernel<true, true, 128, 64, 2>(G16Group);
template __global__ void gemm16_group_kernel<true, true, 128, 128, 2>(G16Group);

// (explicit instantiations: the host stubs of kernels only named inside launch_tile's if/else chain were not emitted)

template __global__ void gemm16_kernel<false, false, 128, 128, 2>(G16Args);
template __global__ void gemm16_kernel<false, true, 128, 128, 2>(G16Args);
template __global__ void gemm16_kernel<true, false, 128, 128, 2>(G16Args);
template __global__ void gemm16_kernel<true, true, 128, 128, 2>(G16Args);
template __global__ void gemm16_kernel<false, false, 128, 64, 2>(G16Args);
template __global__ void gemm16_kernel<false, true, 128, 64, 2>(G16Args);
template __global__ void gemm16_kernel<true, false, 128, 64, 2>(G16Args);
template __global__ void gemm16_kernel<true, true, 128, 64, 2>(G16Args);
template __global__ void gemm16_kernel<false, false, 128, 64, 3>(G16Args);
template __global__ void gemm16_kernel<false, true, 128, 64, 3>(G16Args);
template __global__ void gemm16_kernel<true, false, 128, 64, 3>(G16Args);
template __global__ void gemm16_kernel<true, true, 128, 64, 3>(G16Args);
template __global__ void gemm16_kernel<false, false, 64, 64, 2>(G16Args);
template __global__ void gemm16_kernel<false, true, 64, 64, 2>(G16Args);
template __global__ void gemm16_kernel<true, false, 64, 64, 2>(G16Args);
template __global__ void gemm16_kernel<true, true, 64, 64, 2>(G16Args);
template __global__ void gemm16_kernel<false, false, 64, 64, 3>(G16Args);
template __global__ void gemm16_kernel<false, true, 64, 64, 3>(G16Args);
template __global__ void gemm16_kernel<true, false, 64, 64, 3>(G16Args);
template __global__ void gemm16_kernel<true, true, 64, 64, 3>(G16Args);
template __global__ void gemm16_kernel<false, false, 64, 64, 4>(G16Args);
template __global__ void gemm16_kernel<false, true, 64, 64, 4>(G16Args);
template __global__ void gemm16_kernel<true, false, 64, 64, 4>(G16Args);
template __global__ void gemm16_kernel<true, true, 64, 64, 4>(G16Args);

template __global__ void gemm16_kernel<false, false, 64, 64, 2, 2>(G16Args);
template __global__ void gemm16_kernel<false, false, 64, 64, 2, 3>(G16Args);
template __global__ void gemm16_kernel<false, false, 64, 64, 3, 3>(G16Args);
template __global__ void gemm16_kernel<false, true, 64, 64, 2, 1>(G16Args);
template __global__ void gemm16_kernel<false, true, 64, 64, 3, 1>(G16Args);

template <int BM, int BN, int NS>
void launch_tile(int ta, int tb, const G16Args &a, dim3 grid, hipStream_t s) {
    kk_note_kernelf("gemm16<%d,%d,%d,%d,%d>", ta, tb, BM, BN, NS);
    if (!ta && !tb) hipLaunchKernelGGL((gemm16_kernel<false, false, BM, BN, NS>), grid, dim3(256), 0, s, a);
    else if (!ta && tb) hipLaunchKernelGGL((gemm16_kernel<false, true, BM, BN, NS>), grid, dim3(256), 0, s, a);
    else if (ta && !tb) hipLaunchKernelGGL((gemm16_kernel<true, false, BM, BN, NS>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((gemm16_kernel<true, true, BM, BN, NS>), grid, dim3(256), 0, s, a);
}

// Tile choice (by tile count): with four waves per workgroup the 64x64 tile wins on every shape of this model measured
// inside the train step (the launches are latency-bound: more, smaller workgroups keep more DMAs in flight); with EIGHT
// waves the 128x64 tile is level or better from 128 tiles on (the decoder's 4096-row GEMMs), so those take it and the
// encoder's 512-row GEMMs stay on 64x64.
// Split-K target: a small-tile launch with fewer than target / 2 tiles splits its reduction until it has ~target workgroups.  384 (fill
// the chip 1.5 x) was chosen stand-alone in round 1; INSIDE the step the launches it applies to are the side branch's (the text
// encoder's 512-row GEMMs), where more workgroups + a zero-fill launch + fp32 atomics per launch cost the critical chain beside them more
// than the shorter launch returns: 128 is -0.9 % on the 8 x 512 step (3.722 -> 3.690 ms interleaved), level at 8 x 1024
// (profiles/r05_splitk_target_ab.txt).
int g16_thr128 = 4096, g16_thr12864 = 128, g16_split_target = 128, g16_stages = 3, g16_split_major = 0;
// Optional override of the tile policy, read ONCE when the library is loaded (no mutable policy behind the ABI):
// KK_GEMM16_TUNE="thr128,thr12864,code" with code = flags*100000 + stages*10000 + split target, as tools/ encode it.
struct G16EnvInit {
    G16EnvInit() {
        const char *e = getenv("KK_GEMM16_TUNE");
        int a = 0, b = 0, c = 0;
        if (e && sscanf(e, "%d,%d,%d", &a, &b, &c) == 3 && a > 0) {
            g16_thr128 = a;
            g16_thr12864 = b;
            g16_stages = (c / 10000) % 10 ? (c / 10000) % 10 : 3;
            g16_split_major = c / 100000 ? 1 : 0;
            g16_split_target = c % 10000;
        }
    }
} g16_env_init;
int g16_group_tile = kk_tune_env("KK_GROUP_TILE", 1);                                         // grouped launches: 0 = 64x64, 1 = 128x64 (default: +1 % on the step), 2 = 128x128 tiles
int g16_group_waves = kk_tune_env("KK_GROUP_WAVES", 8);   // 8-wave workgroups on the 128-row tiles (4: the old form)
int g16_group_mfast = kk_tune_env("KK_GROUP_MFAST", 1);  // grouped launches: sweep direction by operand size (0: always n fastest)
int g16_group_split = 0;                                        // grouped launches: 0 = by the split target, n = n k-slices

// ---- the large-tile family (kk_gemm16x.hip) ------------------------------------------------------------------------------------
// A CU's L2 -> LDS stream runs at ~20 B/clk whatever a kernel does, so a launch's floor is the bytes that pass
// through its BUSIEST CU: rounds of workgroups x (BM + BN) x K x 2.  Every candidate tile is priced by that number (in units of
// K x 2 bytes) and the cheapest one runs; ties go to the smaller tile (more workgroups in flight, shorter prologue / epilogue).
int g16x_on = kk_tune_env("KK_G16X", 15);              // tools: bit 0 plain, 1 head-norm, 2 GLU forward / backward, 3 grouped weight gradients
int g16x_min_k_group = kk_tune_env("KK_G16X_GROUP_MIN_K", 1024);
int g16x_min_k_plain = kk_tune_env("KK_G16X_PLAIN_MIN_K", 1024);
int g16x_group_chunks = kk_tune_env("KK_G16X_GROUP_CHUNKS", 1);      // grouped launches: an XCD takes a contiguous eighth of ALL tiles (0: of every problem)
int g16x_min_n_hn = kk_tune_env("KK_G16X_HN_MIN_N", 1024);
int g16x_dbg = kk_tune_env("KK_G16X_DBG", 0);         // tools: probe bits of g16x_body (1 no epilogue, 4 no MFMAs, 8 DMA + barriers only)
int g16x_force = kk_tune_env("KK_G16X_FORCE", -1);
#ifdef KK_TUNING_HOOKS
void *g16x_trace = nullptr;                            // tools: destination of probe bit 32
#endif
int g16_cus() {
    static const int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) return v;
        return 256;
    }();
    return n;
}
long g16_cost(int64_t tiles, int bm, int bn) { return (long)((tiles + g16_cus() - 1) / g16_cus()) * (bm + bn); }

}  // namespace

// Tuning hook used by tools/ (not part of the C ABI).
void kk_gemm16_tune(int thr128, int thr12864, int split_target) {
    g16_thr128 = thr128;
    g16_thr12864 = thr12864;
    g16_stages = (split_target / 10000) % 10 ? (split_target / 10000) % 10 : 3;   // tools encode flags*100000 + stages*10000 + split target
    g16_split_major = split_target / 100000 ? 1 : 0;                     // 1xxxxx: split-major split-K (A/B comparison)
    g16_split_target = split_target % 10000;
}
void kk_gemm16_tune_group(int split) { g16_group_split = split % 100; g16_group_tile = split / 100; }
#ifdef KK_TUNING_HOOKS
// tools: large-tile family on/off bits, forced tile (-1 = by cost), probe bits
extern int g16x_lw;
extern "C" int kk_gemm_tune16x(int on, int force, int dbg) { g16x_on = on & 255; g16x_lw = (on >> 8) & 1 ? 0 : ((on >> 9) & 1 ? 2 : 1); g16x_group_chunks = (on >> 10) & 1 ? 0 : 1; g16x_force = force; g16x_dbg = dbg; return 0; }
void kk_g16x_probe(int bits, void *buf);
extern "C" int kk_gemm_trace16x(void *buf) { g16x_trace = buf; kk_g16x_probe(g16x_dbg, buf); return 0; }      // probe bit 32: 8 waves x 64 stamps (uint64) of workgroup 0
#endif

// True when this core can run the problem (both operands bf16 assumed by the caller).
bool kk_gemm16_eligible(int ta, int tb, int64_t M, int64_t N, int64_t K, const void *A, int64_t lda, const void *B, int64_t ldb) {
    if (((uintptr_t)A & 15) || ((uintptr_t)B & 15) || lda % 8 || ldb % 8) return false;
    if ((!ta || !tb) && K % BK) return false;                 // a k-contiguous operand has no zero-filled K tail
    const int64_t a_el = ta ? (K - 1) * lda + M : (M - 1) * lda + K, b_el = tb ? (K - 1) * ldb + N : (N - 1) * ldb + K;
    return a_el * 2 < (1ll << 31) && b_el * 2 < (1ll << 31);
}

int kk_gemm16_launch(int ta, int tb, int64_t M, int64_t N, int64_t K, float alpha, const void *A, int64_t lda, const void *B,
                     int64_t ldb, float beta, void *C, int64_t ldc, int c_bf16, const float *bias, const float *residual,
                     int64_t ldr, int64_t res_mod, int split_k, int xcd_swizzle, hipStream_t s) {
    auto cd = [](int64_t x, int64_t y) { return (int)((x + y - 1) / y); };
    int BM = 64, BN = 64;
    if (cd(M, 128) * cd(N, 128) >= g16_thr128) { BM = 128; BN = 128; }
    else if (cd(M, 128) * cd(N, 64) >= g16_thr12864) { BM = 128; BN = 64; }
    const int tiles = cd(M, BM) * cd(N, BN), ktiles = cd(K, BK);
    int splits = split_k;
    if (splits <= 0) {
        splits = 1;
        if (tiles * 2 <= g16_split_target) {
            splits = cd(g16_split_target, tiles);
            const int cap = ktiles / 2 > 0 ? ktiles / 2 : 1;
            if (splits > cap) splits = cap;
        }
    }
    // beta == 0 with k-slices needs C zero-filled first: one more node in front of a GEMM that is latency-bound anyway
    // (the encoder's 1000-row projections), so short reductions are not sliced then
    // (+1 % on the train step)
    if (split_k <= 0 && beta == 0.f && ktiles < 32) splits = 1;
    if (splits > ktiles) splits = ktiles;
    if (splits > 1 && (c_bf16 || !(beta == 1.f || (beta == 0.f && ldc == N)))) splits = 1;
    if (split_k <= 0 && splits >= 6 && g16_split_major) {       // a multiple of 8 slices: one XCD per slice residue class
        int s8 = ((splits + 4) / 8) * 8;
        while (s8 > 8 && ktiles / s8 < 2) s8 -= 8;
        if (ktiles / s8 >= 1) splits = s8;
    }
    int k_per_split = cd(ktiles, splits) * BK;
    splits = cd(K, k_per_split);
    const int split_major = (g16_split_major && splits > 1 && splits % 8 == 0) ? 1 : 0;
    G16Args a = {};
    a.M = (int)M; a.N = (int)N; a.K = (int)K;
    a.wt = kk_write_through(ta ? K : M);                       // (ta: a weight gradient — its launch length is the reduction)
    a.alpha = alpha; a.beta = beta;
    a.A = A; a.B = B; a.bias = bias; a.residual = residual; a.C = C; a.c_bf16 = c_bf16;
    a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.ldr = ldr; a.res_mod = res_mod;
    a.k_per_split = k_per_split;
    a.atomic = splits > 1 ? 1 : 0;
    a.splits = splits; a.split_major = split_major;
    a.tiles_m = cd(M, BM); a.tiles_n = cd(N, BN);
    a.xcd_swizzle = xcd_swizzle;
    a.a_bytes = (uint32_t)(((ta ? (K - 1) * lda + M : (M - 1) * lda + K)) * 2);
    a.b_bytes = (uint32_t)(((tb ? (K - 1) * ldb + N : (N - 1) * ldb + K)) * 2);
    // (long reductions only: at K = 512 a one-workgroup-per-CU launch shows its whole prologue and epilogue, two 128x64 workgroups per CU
    // hide each other's — measured 9.5 against 10.3 us at 8192 x 512 x 512, 46 against 37 us at K = 3072; weight-gradient layouts: grouped launches only)
    if ((g16x_on & 1) && splits == 1 && !ta && K >= g16x_min_k_plain) {
        const long cost_old = g16_cost(tiles, BM, BN);
        int best = -1;
        long best_cost = cost_old;
        for (int cfg : {G16X_128x128, G16X_256x128}) {
            int bm, bn;
            kk_g16x_tile(cfg, &bm, &bn);
            const long c = g16_cost((int64_t)cd(M, bm) * cd(N, bn), bm, bn);
            if (c < best_cost) { best = cfg; best_cost = c; }
        }
        if (g16x_force == G16X_128x128 || g16x_force == G16X_256x128) best = g16x_force;
        if (best >= 0) {
            int bm, bn;
            kk_g16x_tile(best, &bm, &bn);
            a.tiles_m = cd(M, bm); a.tiles_n = cd(N, bn); a.dbg = g16x_dbg;
            a.k_per_split = ktiles * BK; a.atomic = 0; a.splits = 1; a.split_major = 0;
            return kk_g16x_plain(best, ta, tb, a, s);
        }
    }
    if (splits > 1 && beta == 0.f) {
        const int e = kk_zero_async(C, (size_t)M * N * sizeof(float), s);
        if (e != 0) return e;
    }
    dim3 grid(a.tiles_m * a.tiles_n * splits);
    const int ns = ktiles / splits < 3 ? 2 : g16_stages;        // (a short reduction gains nothing from depth)
    if (BM == 128 && BN == 128) launch_tile<128, 128, 2>(ta, tb, a, grid, s);
    else if (BM == 128 && g16_w8) { if (ns >= 3 && g16_w8 >= 3) launch_w8<3>(ta, tb, a, grid, s); else launch_w8<2>(ta, tb, a, grid, s); }
    else if (BM == 128) { if (ns >= 3) launch_tile<128, 64, 3>(ta, tb, a, grid, s); else launch_tile<128, 64, 2>(ta, tb, a, grid, s); }
    else if (ns >= 4) launch_tile<64, 64, 4>(ta, tb, a, grid, s);
    else if (ns == 3) launch_tile<64, 64, 3>(ta, tb, a, grid, s);
    else launch_tile<64, 64, 2>(ta, tb, a, grid, s);
    KK_LAUNCH_CHECK("kk_gemm");
    return 0;
}

// dX[M, N] = dY[M, K] . W[K, N] (bf16 everywhere) with Delta[b, head, q] = sum_d dX * O as the epilogue: the dgrad of an attention
// output projection on the eight-wave 128x64 tile (a tile's 64 columns = one head).  `supported`: the shapes the epilogue is written
// for.  The decoder's 4096-row launches take that tile anyway; the text encoder's 512-row ones (32 tiles, a 64x64 launch otherwise)
// take it FOR the epilogue: Delta is what lets the attention backward run as one launch (kk_attn_bwd).
bool kk_gemm16_dgrad_delta_supported(int64_t M, int64_t N, int64_t K) {
    auto cd = [](int64_t x, int64_t y) { return (int)((x + y - 1) / y); };
    return g16_w8 != 0 && M >= 1 && N % 64 == 0 && K % BK == 0 && cd(M, 128) * cd(N, 128) < g16_thr128;
}
int kk_gemm16_dgrad_delta(int64_t M, int64_t N, int64_t K, const void *dy, int64_t lddy, const void *W, int64_t ldw, void *dx,
                          int64_t lddx, const void *O, int64_t ldo, float *delta, int S, int heads, int xcd_swizzle, hipStream_t s) {
    auto cd = [](int64_t x, int64_t y) { return (int)((x + y - 1) / y); };
    G16Args a = {};
    a.M = (int)M; a.N = (int)N; a.K = (int)K;
    a.wt = kk_write_through(M);
    a.alpha = 1.f; a.A = dy; a.B = W; a.C = dx; a.c_bf16 = 1;
    a.lda = lddy; a.ldb = ldw; a.ldc = lddx;
    a.k_per_split = cd(K, BK) * BK; a.splits = 1;
    a.tiles_m = cd(M, 128); a.tiles_n = cd(N, 64); a.xcd_swizzle = xcd_swizzle;
    a.a_bytes = (uint32_t)(((M - 1) * lddy + K) * 2);
    a.b_bytes = (uint32_t)(((K - 1) * ldw + N) * 2);
    a.dl_o = static_cast<const __bf16 *>(O); a.dl_out = delta; a.dl_ldo = ldo; a.dl_S = S; a.dl_heads = heads;
    if ((g16x_on & 1) && K >= g16x_min_k_plain) {
        const long c_old = g16_cost((int64_t)a.tiles_m * a.tiles_n, 128, 64), c_new = g16_cost((int64_t)cd(M, 128) * cd(N, 128), 128, 128);
        if (c_new < c_old) {
            a.tiles_m = cd(M, 128); a.tiles_n = cd(N, 128);
            return kk_g16x_plain(G16X_128x128, 0, 1, a, s);
        }
    }
    dim3 grid(a.tiles_m * a.tiles_n);
    if (cd(K, BK) >= 3 && g16_stages >= 3 && g16_w8 >= 3) launch_w8<3>(0, 1, a, grid, s);
    else launch_w8<2>(0, 1, a, grid, s);
    KK_LAUNCH_CHECK("kk_gemm_dgrad_delta");
    return 0;
}

int kk_gemm16_dgrad_glu(int64_t T, int64_t F, int64_t H, const void *dy, int64_t lddy, const void *W, const void *h1, void *dh1,
                        float *partials, const uint32_t *seed, uint32_t site, float p, int xcd_swizzle, hipStream_t s) {
    auto cd = [](int64_t x, int64_t y) { return (int)((x + y - 1) / y); };
    G16Args a = {};
    a.M = (int)T; a.N = (int)F; a.K = (int)H;
    a.wt = kk_write_through(T);
    a.alpha = 1.f; a.A = dy; a.B = W; a.lda = lddy; a.ldb = F;
    a.k_per_split = cd(H, BK) * BK; a.splits = 1;
    a.tiles_m = cd(T, 64); a.tiles_n = cd(F, 64); a.xcd_swizzle = xcd_swizzle;
    a.a_bytes = (uint32_t)(((T - 1) * lddy + H) * 2);
    a.b_bytes = (uint32_t)(((H - 1) * F + F) * 2);
    a.glu_h = static_cast<const __bf16 *>(h1); a.glu_dh = static_cast<__bf16 *>(dh1); a.glu_partials = partials;
    a.glu_seed = p > 0.f ? seed : nullptr; a.glu_site = site; a.glu_p = p;
    if ((g16x_on & 4) && cd(H, BK) >= 3) {
        const long c_old = g16_cost((int64_t)cd(T, 128) * cd(F, 64), 128, 64), c_new = g16_cost((int64_t)cd(T, 128) * cd(F, 192), 128, 192);
        if (c_new < c_old) {
            a.tiles_m = cd(T, 128); a.tiles_n = cd(F, 192);
            return kk_g16x_glu_bwd(a, s);
        }
    }
    if ((g16_w8_glu & 1) && cd(H, BK) >= 3 && cd(T, 128) * cd(F, 64) >= g16_thr12864) {       // eight waves on 128x64 tiles, like the plain GEMMs
        a.tiles_m = cd(T, 128);
        kk_note_kernel("gemm16_w8_glu<1,3,1>");
        hipLaunchKernelGGL((gemm16_kernel_w8_glu<true, 3, 1>), dim3(a.tiles_m * a.tiles_n), dim3(512), 0, s, a);
        KK_LAUNCH_CHECK("kk_gemm_dgrad_glu");
        return 0;
    }
    dim3 grid(a.tiles_m * a.tiles_n);
    kk_note_kernel("gemm16<0,1,64,64,*,1>");
    if (cd(H, BK) < 3) hipLaunchKernelGGL((gemm16_kernel<false, true, 64, 64, 2, 1>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((gemm16_kernel<false, true, 64, 64, 3, 1>), grid, dim3(256), 0, s, a);
    KK_LAUNCH_CHECK("kk_gemm_dgrad_glu");
    return 0;
}

int kk_gemm16_linear_glu(int64_t T, int64_t F, int64_t K, const void *x, int64_t ldx, const void *W, const float *bias, void *h1,
                         void *g, int64_t ldg, const uint32_t *seed, uint32_t site, float p, int xcd_swizzle, hipStream_t s) {
    auto cd = [](int64_t a_, int64_t b_) { return (int)((a_ + b_ - 1) / b_); };
    G16Args a = {};
    a.M = (int)T; a.N = (int)F; a.K = (int)K;          // N = F: a workgroup covers columns n and F + n of the [T, 2F] product
    a.wt = kk_write_through(T);
    a.alpha = 1.f; a.A = x; a.B = W; a.lda = ldx; a.ldb = K; a.bias = bias; a.C = g; a.ldc = ldg; a.c_bf16 = 1;
    a.k_per_split = cd(K, BK) * BK; a.splits = 1;
    a.tiles_m = cd(T, 64); a.tiles_n = cd(F, 64); a.xcd_swizzle = xcd_swizzle;
    a.a_bytes = (uint32_t)(((T - 1) * ldx + K) * 2);
    a.b_bytes = (uint32_t)(((2 * F - 1) * K + K) * 2);
    a.glu_dh = static_cast<__bf16 *>(h1);
    a.glu_seed = p > 0.f ? seed : nullptr; a.glu_site = site; a.glu_p = p;
    if ((g16x_on & 4) && cd(K, BK) >= 3) {               // 256 rows x (96 + 96) columns per workgroup against 64 x (64 + 64)
        const long c_old = g16_cost((int64_t)cd(T, 64) * cd(F, 64), 64, 128), c_new = g16_cost((int64_t)cd(T, 256) * cd(F, 96), 256, 192);
        if (c_new < c_old) {
            a.tiles_m = cd(T, 256); a.tiles_n = cd(F, 96);
            return kk_g16x_glu_fwd(a, s);
        }
    }
    if ((g16_w8_glu & 2) && cd(K, BK) >= 3 && cd(T, 128) * cd(F, 64) >= g16_thr12864) {
        a.tiles_m = cd(T, 128);
        kk_note_kernel("gemm16_w8_glu<0,3,2>");
        hipLaunchKernelGGL((gemm16_kernel_w8_glu<false, 3, 2>), dim3(a.tiles_m * a.tiles_n), dim3(512), 0, s, a);
        KK_LAUNCH_CHECK("kk_gemm_linear_glu");
        return 0;
    }
    kk_note_kernel("gemm16<0,0,64,64,2,2>");
    hipLaunchKernelGGL((gemm16_kernel<false, false, 64, 64, 2, 2>), dim3(a.tiles_m * a.tiles_n), dim3(256), 0, s, a);
    KK_LAUNCH_CHECK("kk_gemm_linear_glu");
    return 0;
}

// dW_i[M_i, N_i] += dY_i[T_i, M_i]^T . X_i[T_i, N_i] for i < n, one launch (see gemm16_group_kernel).
int kk_gemm16_wgrad_group(const KkWgradDesc *d, int n, int split_k, int overwrite, int xcd_swizzle, hipStream_t s, void *ss_rec,
                          const int32_t *ss_seg, int32_t *ss_count) {
    // ss_rec / ss_seg: when every problem of the launch is written exactly once per element by one workgroup (no k-slices, the
    // write-through fp32 epilogue), its tiles leave the sums of squares of what they stored as records [*ss_count ..) of ss_rec and
    // *ss_count is advanced; otherwise *ss_count stays and the caller's norm pass reads those tensors itself
    KkSegRec *rec = static_cast<KkSegRec *>(ss_rec);
    int rec_used = 0;
    auto cd = [](int64_t x, int64_t y) { return (int)((x + y - 1) / y); };
    if (n < 1 || n > GROUP_MAX) return kk_fail(KK_EINVAL, "kk_gemm_wgrad_group: 1..%d problems per launch, got %d", GROUP_MAX, n);
    int total = 0;
    const int BM = g16_group_tile >= 1 ? 128 : 64, BN = g16_group_tile >= 2 ? 128 : 64;
    for (int i = 0; i < n; ++i) {
        if (d[i].M <= 0 || d[i].N <= 0 || d[i].T <= 0 || !d[i].dy || !d[i].x || !d[i].dw)
            return kk_fail(KK_EINVAL, "kk_gemm_wgrad_group: bad problem %d", i);
        if (!kk_gemm16_eligible(1, 1, d[i].M, d[i].N, d[i].T, d[i].dy, d[i].lddy, d[i].x, d[i].ldx))
            return kk_fail(KK_EINVAL, "kk_gemm_wgrad_group: problem %d needs 16-byte aligned bf16 operands with row strides %% 8 == 0", i);
        total += cd(d[i].M, BM) * cd(d[i].N, BN);
    }
    int splits = 1;
    {   // 128x128 tiles of the large-tile family: full reductions only, long ones (short launches sit on the side branch, where
        // a 96 KB workgroup keeps the main chain's workgroups waiting)
        int total_x = 0;
        int64_t kmin = 1ll << 40;
        for (int i = 0; i < n; ++i) { total_x += cd(d[i].M, 128) * cd(d[i].N, 128); kmin = std::min<int64_t>(kmin, d[i].T); }
        const bool old_splits = split_k > 1 || g16_group_split > 1 || (split_k <= 0 && !overwrite && total * 2 <= g16_split_target);
        if ((g16x_on & 8) && !old_splits && kmin >= g16x_min_k_group && g16_cost(total_x, 128, 128) < g16_cost(total, BM, BN)) {
            G16Group g = {};
            g.n = n;
            g.xcd_chunks = (xcd_swizzle && g16x_group_chunks) ? 1 : 0;
            for (int i = 0; i < n; ++i) {
                const int64_t M = d[i].M, N = d[i].N, K = d[i].T;
                G16Args &a = g.p[i];
                a.M = (int)M; a.N = (int)N; a.K = (int)K;
                a.wt = kk_write_through(K);
                a.alpha = 1.f; a.beta = overwrite ? 0.f : 1.f;
                a.A = d[i].dy; a.B = d[i].x; a.C = d[i].dw;
                a.lda = d[i].lddy; a.ldb = d[i].ldx; a.ldc = d[i].lddw;
                a.k_per_split = cd(K, BK) * BK; a.splits = 1;
                a.tiles_m = cd(M, 128); a.tiles_n = cd(N, 128); a.xcd_swizzle = g.xcd_chunks ? 0 : xcd_swizzle; a.dbg = g16x_dbg;
                a.m_fast = (g16_group_mfast && xcd_swizzle && M < N) ? 1 : 0;
                a.a_bytes = (uint32_t)(((K - 1) * a.lda + M) * 2);
                a.b_bytes = (uint32_t)(((K - 1) * a.ldb + N) * 2);
                g.start[i + 1] = g.start[i] + a.tiles_m * a.tiles_n;
            }
            bool all_wt = rec != nullptr && ss_seg != nullptr && ss_count != nullptr;
            for (int i = 0; i < n && all_wt; ++i)               // (the conditions of g16x_body's write-through fp32 epilogue; a tile inside ONE segment)
                all_wt = (ss_seg[2 * i + 1] == 0 || ss_seg[2 * i + 1] % 128 == 0) && g.p[i].wt && (g.p[i].ldc & 3) == 0 && (g.p[i].N & 7) == 0 && (reinterpret_cast<uintptr_t>(g.p[i].C) & 15) == 0;
            if (all_wt) {
                for (int i = 0; i < n; ++i) {
                    g.p[i].ss_rec = rec + *ss_count + g.start[i];
                    g.p[i].ss_seg = ss_seg[2 * i];
                    g.p[i].ss_rows = ss_seg[2 * i + 1];
                }
                *ss_count += g.start[n];
            }
            return kk_g16x_group(g, g.start[n], s);
        }
    }
    if (split_k > 0) splits = split_k;                            // the caller's k-slice count (0 = by the split target)
    else if (g16_group_split > 0) splits = g16_group_split;
    else if (total * 2 <= g16_split_target) splits = cd(g16_split_target, total);
    if (overwrite) splits = 1;                                    // (k-slices accumulate with atomics: they need the old value)
    G16Group g = {};
    g.n = n;
    int min_per = 1 << 30;
    for (int i = 0; i < n; ++i) {
        const int64_t M = d[i].M, N = d[i].N, K = d[i].T;
        const int ktiles = cd(K, BK);
        int sp = std::min(splits, std::max(ktiles / 2, 1));
        const int kps = cd(ktiles, sp) * BK;
        sp = cd(K, kps);
        min_per = std::min(min_per, kps / BK);
        G16Args &a = g.p[i];
        a.M = (int)M; a.N = (int)N; a.K = (int)K;
        a.wt = kk_write_through(K);                                // (a weight gradient's launch length is its reduction: the tokens)
        a.alpha = 1.f; a.beta = overwrite ? 0.f : 1.f;
        a.A = d[i].dy; a.B = d[i].x; a.C = d[i].dw;
        a.lda = d[i].lddy; a.ldb = d[i].ldx; a.ldc = d[i].lddw;
        a.k_per_split = kps; a.splits = sp; a.atomic = sp > 1 ? 1 : 0;
        a.tiles_m = cd(M, BM); a.tiles_n = cd(N, BN); a.xcd_swizzle = xcd_swizzle;
        a.m_fast = (g16_group_mfast && xcd_swizzle && M < N) ? 1 : 0;
        a.a_bytes = (uint32_t)(((K - 1) * a.lda + M) * 2);
        a.b_bytes = (uint32_t)(((K - 1) * a.ldb + N) * 2);
        g.start[i + 1] = g.start[i] + a.tiles_m * a.tiles_n * sp;
    }
    {
        bool all_wt = rec != nullptr && ss_seg != nullptr && ss_count != nullptr && BM == 128 && BN == 64 && g16_group_waves == 8;
        for (int i = 0; i < n && all_wt; ++i)                   // (the conditions of gemm16_body's write-through fp32 epilogue, one k-slice)
            all_wt = (ss_seg[2 * i + 1] == 0 || ss_seg[2 * i + 1] % 128 == 0) && g.p[i].splits == 1 && g.p[i].wt && (g.p[i].ldc & 3) == 0 && (g.p[i].N & 7) == 0 && (reinterpret_cast<uintptr_t>(g.p[i].C) & 15) == 0;
        if (all_wt) {
            for (int i = 0; i < n; ++i) {
                g.p[i].ss_rec = rec + *ss_count + g.start[i];
                g.p[i].ss_seg = ss_seg[2 * i];
                g.p[i].ss_rows = ss_seg[2 * i + 1];
            }
            *ss_count += g.start[n];
        }
    }
    (void)rec_used;
    dim3 grid(g.start[n]);
    kk_note_kernelf("gemm16_group<%d,%d,w%d>", BM, BN, BM == 128 ? g16_group_waves : 4);
    if (BN == 128 && g16_group_waves == 16) hipLaunchKernelGGL((gemm16_group_kernel<true, true, 128, 128, 2, 16, 4>), grid, dim3(1024), 0, s, g);
    else if (BN == 128 && g16_group_waves == 8) hipLaunchKernelGGL((gemm16_group_kernel<true, true, 128, 128, 2, 8>), grid, dim3(512), 0, s, g);
    else if (BN == 128) hipLaunchKernelGGL((gemm16_group_kernel<true, true, 128, 128, 2>), grid, dim3(256), 0, s, g);
    else if (BM == 128 && g16_group_waves == 8) hipLaunchKernelGGL((gemm16_group_kernel<true, true, 128, 64, 2, 8>), grid, dim3(512), 0, s, g);
    else if (BM == 128) hipLaunchKernelGGL((gemm16_group_kernel<true, true, 128, 64, 2>), grid, dim3(256), 0, s, g);
    else if (min_per < 3 || g16_stages < 3) hipLaunchKernelGGL((gemm16_group_kernel<true, true, 64, 64, 2>), grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL((gemm16_group_kernel<true, true, 64, 64, 3>), grid, dim3(256), 0, s, g);
    KK_LAUNCH_CHECK("kk_gemm_wgrad_group");
    return 0;
}

int kk_gemm16_qkv_headnorm(int64_t T, int parts, int heads, int64_t K, const void *x, int64_t ldx, const void *W, const float *bias,
                           void *raw, int64_t ldraw, void *y, int64_t ldy, int S, const float *const *gains, int rope_mask,
                           const float *cos_t, const float *sin_t, int xcd_swizzle, hipStream_t s) {
    auto cd = [](int64_t a_, int64_t b_) { return (int)((a_ + b_ - 1) / b_); };
    const int64_t N = (int64_t)parts * heads * 64;
    G16Args a = {};
    a.M = (int)T; a.N = (int)N; a.K = (int)K;
    a.wt = kk_write_through(T);
    a.alpha = 1.f; a.A = x; a.B = W; a.lda = ldx; a.ldb = K; a.bias = bias; a.C = raw; a.ldc = ldraw; a.c_bf16 = 1;
    a.k_per_split = cd(K, BK) * BK; a.splits = 1;
    a.tiles_m = cd(T, 64); a.tiles_n = cd(N, 64); a.xcd_swizzle = xcd_swizzle;
    a.a_bytes = (uint32_t)(((T - 1) * ldx + K) * 2);
    a.b_bytes = (uint32_t)(((N - 1) * K + K) * 2);
    for (int i = 0; i < parts; ++i) a.hn_gain[i] = gains[i];
    a.hn_cos = cos_t; a.hn_sin = sin_t; a.hn_y = static_cast<__bf16 *>(y); a.hn_ldy = ldy;
    a.hn_S = S; a.hn_H = heads * 64; a.hn_rope_mask = rope_mask;
    if ((g16x_on & 2) && cd(K, BK) >= 3 && N >= g16x_min_n_hn && ldraw % 8 == 0 && ldy % 8 == 0 && (((uintptr_t)raw | (uintptr_t)y) & 15) == 0) {      // (16-byte stores)
        const bool w8 = g16_w8_hn && cd(T, 128) * cd(N, 64) >= g16_thr12864;
        int best = -1;
        long best_cost = w8 ? g16_cost((int64_t)cd(T, 128) * cd(N, 64), 128, 64) : g16_cost((int64_t)cd(T, 64) * cd(N, 64), 64, 64);
        for (int cfg : {G16X_128x128, G16X_128x192, G16X_256x128, G16X_256x192}) {
            int bm, bn;
            kk_g16x_tile(cfg, &bm, &bn);
            const long c = g16_cost((int64_t)cd(T, bm) * cd(N, bn), bm, bn);
            if (c < best_cost) { best = cfg; best_cost = c; }
        }
        if (g16x_force >= 0) best = g16x_force;
        if (best >= 0) {
            int bm, bn;
            kk_g16x_tile(best, &bm, &bn);
            a.tiles_m = cd(T, bm); a.tiles_n = cd(N, bn); a.dbg = g16x_dbg;
            return kk_g16x_headnorm(best, a, s);
        }
    }
    if (g16_w8_hn && cd(K, BK) >= 3 && cd(T, 128) * cd(N, 64) >= g16_thr12864) {       // eight waves on 128x64 tiles, like the plain GEMMs
        a.tiles_m = cd(T, 128);
        kk_note_kernel("gemm16_w8_hn<3>");
        hipLaunchKernelGGL((gemm16_kernel_w8_hn<3>), dim3(a.tiles_m * a.tiles_n), dim3(512), 0, s, a);
        KK_LAUNCH_CHECK("kk_gemm_qkv_headnorm");
        return 0;
    }
    dim3 grid(a.tiles_m * a.tiles_n);
    kk_note_kernel("gemm16<0,0,64,64,*,3>");
    if (cd(K, BK) < 3) hipLaunchKernelGGL((gemm16_kernel<false, false, 64, 64, 2, 3>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((gemm16_kernel<false, false, 64, 64, 3, 3>), grid, dim3(256), 0, s, a);
    KK_LAUNCH_CHECK("kk_gemm_qkv_headnorm");
    return 0;
}
#endif  // KK_BODIES_ONLY
