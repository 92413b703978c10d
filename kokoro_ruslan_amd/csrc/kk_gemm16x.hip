// Large-tile family of the bf16-storage MFMA GEMM core (gfx950).
//
// Why it exists.  A compute unit pulls ~20 B/clk from L2 into LDS whatever the staging depth, so a GEMM
// launch cannot finish before   max over CUs of  sum over its workgroups of (BM + BN) * K * 2 bytes  have gone through a CU.
// kk_gemm16.hip's tiles (64x64, 128x64: ONE 32x32 accumulator per wave) sit on that floor for the model's big launches.
// The tiles here hold 2 - 6 accumulators per wave on 128x128 .. 256x192 workgroup tiles: 1.5 - 1.8x fewer bytes through the
// busiest CU for the q|k|v projections, linear1 + GLU, the linear2 dgrad + GLU', and the grouped weight gradients, and half
// the LDS fragment reads per MFMA.  One workgroup (eight waves, two per SIMD) per CU; tile chosen per launch by that byte count
// (kk_gemm16.hip: g16x_pick).
//
// Same operand images as kk_gemm16.hip (see its header): LDS filled by buffer-load-to-LDS DMA, 16 bytes per lane;
//   k-contiguous operand: [rows][64] bf16, 16-byte chunk index XOR (row >> 1) & 7 applied on the global address;
//   k-strided operand: [64 k][rows] as it lies in memory, 32-byte blocks XOR-ed by a function of k, fragments by
//     ds_read_b64_tr_b16 (the XOR class follows the row pitch mod 256 bytes: 192- and 64-row images share one, 128 / 256 the other).
// Counted vmcnt + raw s_barrier keep NS - 1 k-tiles in flight across the barrier; LDS reads of the next 16-k slab(s) are in flight
// under the MFMAs of the current one (inline asm, counted lgkmcnt).
#include "kk_gemm16.h"

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
#define LDS_PTR(p) ((__attribute__((address_space(3))) void *)(p))

// One operand tile of ROWS x 64.  Rows [0, SPLIT) map to global rows r0.., rows [SPLIT, ROWS) to r1.. (the GLU forward's two
// panels of W1; SPLIT == ROWS otherwise).
template <int ROWS, bool KS, int NT, int SPLIT = ROWS, int AUX = 0> struct OperandX {      // AUX: cache policy of the DMA loads (kk_gemm16.h: KK_A_AUX)
    static constexpr int BYTES = ROWS * BK * 2;
    static constexpr int NP = ROWS * 8 / NT;                    // 16-byte pieces per thread per tile
    static_assert(ROWS * 8 % NT == 0 && SPLIT % 8 == 0, "whole pieces per thread");
    static constexpr int PITCH = KS ? ROWS * 2 : BK * 2;
    static constexpr bool S4 = (PITCH % 256) == 0;              // k-strided image: four k-rows alias mod 256 bytes (else two)
    uint32_t voff[NP];
    uint32_t kstep;
    __amdgpu_buffer_rsrc_t rsrc;

    __device__ __forceinline__ void init(const void *base, uint32_t bytes, int64_t ld, int r0, int r1, int t) {      // t: index among the NT issuing threads
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int p = t + NT * j;
            if constexpr (!KS) {
                const int row = p >> 3, c = (p & 7) ^ ((row >> 1) & 7);
                const int grow = row < SPLIT ? r0 + row : r1 + row - SPLIT;
                voff[j] = (uint32_t)(((int64_t)grow * ld + c * 8) * 2);
            } else {
                constexpr int PPR = ROWS / 8;
                const int k = p / PPR, q = p % PPR;
                const int s = S4 ? 2 * (k & 3) : 2 * ((k >> 1) & 1);
                const int col = ((((q >> 1) ^ s) << 1) | (q & 1)) * 8;
                const int gcol = col < SPLIT ? r0 + col : r1 + col - SPLIT;
                voff[j] = (uint32_t)(((int64_t)k * ld + gcol) * 2);
            }
        }
        kstep = KS ? (uint32_t)(ld * BK * 2) : (uint32_t)(BK * 2);
    }
    __device__ __forceinline__ void issue(char *dst, int kt, int wave) const {
        const uint32_t so = (uint32_t)kt * kstep;
#pragma unroll
        for (int j = 0; j < NP; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(dst + (wave * 64 + NT * j) * 16), 16, voff[j], so, 0, AUX);
    }
};

struct FragX {
    bf16x8 v;            // k-contiguous operand
    s16x4 lo, hi;        // k-strided operand
};

// Fragment reads of NF 32-row blocks of an operand image; blk[f] = index of block f inside the image (wave-uniform).
template <int ROWS, bool KS, int NF> struct FragAddrX {
    static constexpr int PITCH = KS ? ROWS * 2 : BK * 2;
    static constexpr bool S4 = (PITCH % 256) == 0;
    static constexpr int READS = KS ? 2 : 1;
    uint32_t base;
    uint32_t x[KS ? NF : 4];
    uint32_t boff[KS ? 1 : NF];
    __device__ __forceinline__ void init(int lane, const int (&blk)[NF]) {
        const int l31 = lane & 31, half = lane >> 5;
        if constexpr (!KS) {
            const int swz = (l31 >> 1) & 7;
            base = (uint32_t)(l31 * (BK * 2));
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) x[ks] = (uint32_t)(((2 * ks + half) ^ swz) * 16);
#pragma unroll
            for (int f = 0; f < NF; ++f) boff[f] = (uint32_t)(blk[f] * 32 * (BK * 2));
        } else {
            const int L = lane & 15, gi = (lane >> 4) & 1, kq = L >> 2;
            const int s = S4 ? 2 * (kq & 3) : 2 * ((kq >> 1) & 1);
            base = (uint32_t)((8 * half + kq) * PITCH + 8 * (L & 3));
#pragma unroll
            for (int f = 0; f < NF; ++f) x[f] = (uint32_t)(((2 * blk[f] + gi) ^ s) * 32);
        }
    }
    __device__ __forceinline__ void load(FragX &fr, const char *img, int f, int ks) const {
        if constexpr (!KS) {
            const uint32_t addr = (uint32_t)(uintptr_t)LDS_PTR(img) + base + x[ks] + boff[f];
            asm volatile("ds_read_b128 %0, %1" : "=v"(fr.v) : "v"(addr));
        } else {
            const uint32_t addr = (uint32_t)(uintptr_t)LDS_PTR(img) + base + x[f] + ks * 16 * PITCH;
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(fr.lo) : "v"(addr));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(fr.hi) : "v"(addr), "n"(4 * PITCH));
        }
    }
};

__device__ __forceinline__ bf16x8 fragx_value(const FragX &f, bool ks) {
    if (!ks) return f.v;
    s16x8 v;
    v[0] = f.lo[0]; v[1] = f.lo[1]; v[2] = f.lo[2]; v[3] = f.lo[3]; v[4] = f.hi[0]; v[5] = f.hi[1]; v[6] = f.hi[2]; v[7] = f.hi[3];
    return __builtin_bit_cast(bf16x8, v);
}
template <int PENDING> __device__ __forceinline__ void wait_readsx() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(PENDING) : "memory"); }
__device__ __forceinline__ void pin_fragx(FragX &f, bool ks) {
    if (ks) asm volatile("" : "+v"(f.lo), "+v"(f.hi));
    else asm volatile("" : "+v"(f.v));
}

// EPI: 0 plain (bias, bf16 / fp32 C, optional Delta rows), 1 GLU backward on the linear2 dgrad, 2 GLU forward on linear1 (BN = the
// a-panel rows + the b-panel rows), 3 per-head RMSNorm (+ RoPE) on a q / k / v projection.  Same arithmetic, same bits as the
// epilogues of kk_gemm16.hip (which document them).
template <bool TA, bool TB, int BM, int BN, int NS, int EPI, int WR, int WC, int LW>
__device__ __forceinline__ void g16x_body(const G16Args &a, const int wg, char *smem) {
    // WR x WC COMPUTE waves own the accumulators; LW LOADER waves (LW > 0) do nothing but issue the buffer-load-to-LDS DMA.
    // Why: a DMA instruction (1 KB per wave) costs its wave 70 - 140 clocks of issue when every wave of the CU issues its share
    // at the same point of the k-step (one address unit per CU), 350 - 700 clocks per k-step in which the wave's MFMAs wait behind
    // it (shader-clock stamps, tools/probes/g16x_trace.py: k-step 2100 clocks for 768 clocks of MFMA per SIMD; the DMA data had
    // always landed already).  With loader waves a SIMD holds one compute wave that never touches the address unit and one
    // loader wave that blocks there harmlessly.
    constexpr int CWV = WR * WC, WAVES = CWV + LW, NT = 64 * WAVES, LT = LW > 0 ? 64 * LW : NT;
    constexpr int BNH = EPI == 2 ? BN / 2 : BN;                 // output columns per workgroup (of ONE panel when EPI == 2)
    constexpr int MI = BM / (32 * WR);
    constexpr int NW = BNH / WC, NJ = NW / 32;                  // columns / 32-column blocks per wave (per panel)
    constexpr int NI = EPI == 2 ? 2 * NJ : NJ;
    static_assert(BM % (32 * WR) == 0 && BNH % (32 * WC) == 0, "whole 32x32 accumulators per wave");
    using OA = OperandX<BM, TA, LT, BM, KK_A_AUX>;
    using OB = OperandX<BN, TB, LT, BNH>;
    constexpr int STAGE = OA::BYTES + OB::BYTES;
    constexpr int NPT = OA::NP + OB::NP;

    // workgroup -> tile (same XCD-aware order as kk_gemm16.hip: an XCD sweeps a contiguous run of tiles)
    int tid_lin = wg;
    {
        const int ntiles = a.tiles_m * a.tiles_n;
        if (a.xcd_swizzle) {
            const int q = ntiles >> 3, r = ntiles & 7, xcd = tid_lin & 7, in = tid_lin >> 3;
            tid_lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + in;
        }
    }
    const int m0 = (a.m_fast ? tid_lin % a.tiles_m : tid_lin / a.tiles_n) * BM;
    const int n0 = (a.m_fast ? tid_lin / a.tiles_m : tid_lin % a.tiles_n) * BNH;
    const int nk = (a.K + BK - 1) / BK;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool loader = LW > 0 && wave >= CWV;
    const int wr = wave / WC, wc = wave % WC, half = lane >> 5, l31 = lane & 31;

    // probe (tools builds): shader-clock stamps of workgroup 0's waves, 4 per k-step, into the buffer at a.dl_out
    unsigned long long *trace = (KK_DBG(a, 32) && wg == 0 && lane == 0) ? reinterpret_cast<unsigned long long *>(a.dl_out) + wave * 64 : nullptr;
    auto stamp = [&](int kt, int which) {
        if (KK_DBG(a, 32) && trace != nullptr && kt < 16) trace[kt * 4 + which] = __builtin_amdgcn_s_memtime();
    };

    // the tile's bias row, read by phase A of the epilogues (24 dependent global loads there cost 3 us)
    float *bias_lds = reinterpret_cast<float *>(smem + NS * STAGE);
    if (EPI != 1 && !loader && threadIdx.x < BN) {
        const int tc = threadIdx.x, gc = (EPI == 2 && tc >= BNH) ? a.N + n0 + tc - BNH : n0 + tc;
        const int lim = EPI == 2 ? 2 * a.N : a.N;
        bias_lds[tc] = (a.bias != nullptr && gc < lim && (EPI != 2 || (tc < BNH ? gc < a.N : true))) ? a.bias[gc] : 0.f;
    }

    f32x16 acc[MI][NI];
    if (LW > 0 && loader) {
        // ---- loader waves: tile kt + NS - 1 goes out right behind the barrier that frees its stage
        OA oa;
        OB ob;
        const int lt = threadIdx.x - 64 * CWV, lw = wave - CWV;
        oa.init(a.A, a.a_bytes, a.lda, m0, m0, lt);
        ob.init(a.B, a.b_bytes, a.ldb, n0, EPI == 2 ? a.N + n0 : n0, lt);
#pragma unroll
        for (int p = 0; p < NS - 1; ++p)
            if (p < nk) {
                oa.issue(smem + p * STAGE, p, lw);
                ob.issue(smem + p * STAGE + OA::BYTES, p, lw);
            }
        int sn = NS - 1;
        for (int kt = 0; kt < nk; ++kt) {
            stamp(kt, 0);
            const int younger = min(nk - 1 - kt, NS - 2);
            if (NS >= 4 && younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NPT) : "memory");
            else if (NS >= 3 && younger >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPT) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            stamp(kt, 1);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            stamp(kt, 2);
            if (kt + NS - 1 < nk) {
                oa.issue(smem + sn * STAGE, kt + NS - 1, lw);
                ob.issue(smem + sn * STAGE + OA::BYTES, kt + NS - 1, lw);
            }
            stamp(kt, 3);
            sn = sn + 1 == NS ? 0 : sn + 1;
        }
        stamp(min(nk, 15), 0);
    } else {
    OA oa;
    OB ob;
    if constexpr (LW == 0) {
        oa.init(a.A, a.a_bytes, a.lda, m0, m0, threadIdx.x);
        ob.init(a.B, a.b_bytes, a.ldb, n0, EPI == 2 ? a.N + n0 : n0, threadIdx.x);
    }
    int ablk[MI], bblk[NI];
#pragma unroll
    for (int i = 0; i < MI; ++i) ablk[i] = wr * MI + i;
#pragma unroll
    for (int j = 0; j < NI; ++j) bblk[j] = (EPI == 2 && j >= NJ) ? BNH / 32 + wc * NJ + (j - NJ) : wc * NJ + j;
    FragAddrX<BM, TA, MI> fa;
    FragAddrX<BN, TB, NI> fb;
    fa.init(lane, ablk);
    fb.init(lane, bblk);

#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if constexpr (LW == 0) {
#pragma unroll
        for (int p = 0; p < NS - 1; ++p)
            if (p < nk) {
                oa.issue(smem + p * STAGE, p, wave);
                ob.issue(smem + p * STAGE + OA::BYTES, p, wave);
            }
    }
    int sc = 0, sn = NS - 1;
    for (int kt = 0; kt < nk; ++kt) {
        stamp(kt, 0);
        if constexpr (LW == 0) {
            const int younger = min(nk - 1 - kt, NS - 2);
            if (NS >= 4 && younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NPT) : "memory");
            else if (NS >= 3 && younger >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPT) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        stamp(kt, 1);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        stamp(kt, 2);
        if constexpr (LW == 0) {
            if (kt + NS - 1 < nk) {
                oa.issue(smem + sn * STAGE, kt + NS - 1, wave);
                ob.issue(smem + sn * STAGE + OA::BYTES, kt + NS - 1, wave);
            }
        }
        stamp(kt, 3);
        const char *Ai = smem + sc * STAGE, *Bi = Ai + OA::BYTES;
        sn = sc;
        sc = sc + 1 == NS ? 0 : sc + 1;
        constexpr int RPS = MI * FragAddrX<BM, TA, MI>::READS + NI * FragAddrX<BN, TB, NI>::READS;
        static_assert(RPS <= 15, "a slab's reads must fit the lgkmcnt counter");
        constexpr int AHEAD = 3 * RPS <= 15 ? 2 : 1;            // slabs whose reads are in flight under the MFMAs of the current one
        FragX af[4][MI], bf[4][NI];
        auto read_slab = [&](int ks) {
#pragma unroll
            for (int i = 0; i < MI; ++i) fa.load(af[ks][i], Ai, i, ks);
#pragma unroll
            for (int j = 0; j < NI; ++j) fb.load(bf[ks][j], Bi, j, ks);
        };
        if (KK_DBG(a, 8)) continue;                             // probe: DMA + barriers only
#pragma unroll
        for (int ks = 0; ks < AHEAD; ++ks) read_slab(ks);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks + AHEAD < 4) read_slab(ks + AHEAD);
            if (ks + AHEAD < 4) wait_readsx<AHEAD * RPS>();
            else if (AHEAD == 2 && ks == 2) wait_readsx<RPS>();
            else wait_readsx<0>();
#pragma unroll
            for (int i = 0; i < MI; ++i) pin_fragx(af[ks][i], TA);
#pragma unroll
            for (int j = 0; j < NI; ++j) pin_fragx(bf[ks][j], TB);
            if (!KK_DBG(a, 4)) {                                // (probe: no MFMAs)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fragx_value(bf[ks][j], TB), fragx_value(af[ks][i], TA), acc[i][j], 0, 0, 0);      // (operands swapped: the block's transpose)
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    stamp(min(nk, 15), 0);
    }
    if (nk <= 0) return;
    if (KK_DBG(a, 1)) {                                         // probe: no epilogue (one store keeps the accumulators alive)
        float t = 0.f;
        if (!loader) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) t += acc[i][j][r];
        }
        if (t == 1.2345f) static_cast<float *>(a.C)[0] = t;
        return;
    }

    // ---- epilogues.  The accumulators are TRANSPOSED blocks (the MFMAs run with their operands swapped): lane (l31, half) holds,
    // for row l31 of block (i, j), the columns 4 * half + 8 * g + e (g, e < 4) — four consecutive columns per register quad.  Phase A
    // (compute waves): the workgroup's tile goes to LDS with 8- / 16-byte stores.  Phase B (EVERY wave, loaders included): 32 x 32
    // blocks (32 x 64 for the plain epilogue: a head per block pair) are dealt out round robin; a lane works on 8 consecutive
    // columns of 2 rows with 16-byte global accesses, as the epilogues of kk_gemm16.hip do: same arithmetic, same bits.
    constexpr int TPF = BN + 4;                                 // floats per row of the fp32 tile (16-byte rows, 4-bank skew)
    constexpr int TPH = BN + 8;                                 // bf16 per row of the bf16 tile
    auto blockcol = [&](int j) { return (EPI == 2 && j >= NJ) ? BNH + wc * NW + (j - NJ) * 32 : wc * NW + j * 32; };      // tile-local first column of block j
    const int c8 = (lane & 3) * 8;
    __builtin_amdgcn_s_barrier();                               // every wave is done with the last stage
    stamp(9, 0);

    if constexpr (EPI == 1) {
        static_assert(EPI != 1 || NS * STAGE >= BM * TPF * 4, "the staging area holds the fp32 tile");
        const int F = a.N;
        const uint32_t thr = a.glu_seed ? kk_drop_threshold(a.glu_p) : 0u, seed = thr ? *a.glu_seed : 0u;
        const float ik = thr ? 1.f / (1.f - a.glu_p) : 1.f;
        float *tile = reinterpret_cast<float *>(smem);
        if (!loader) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        st4(tile + ((wr * MI + i) * 32 + l31) * TPF + blockcol(j) + 4 * half + 8 * g,
                            make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]));
        }
        stamp(9, 1);
        __syncthreads();
        stamp(9, 2);
        constexpr int NBR = BM / 32, NBC = BN / 32;
#pragma unroll 1
        for (int blk = wave; blk < NBR * NBC; blk += WAVES) {
            const int bi = blk / NBC, bj = blk % NBC;
            const int col = n0 + bj * 32 + c8;
            float sa[8], sb[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) sa[e] = sb[e] = 0.f;
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int rl = it * 16 + (lane >> 2), row = m0 + bi * 32 + rl;
                if (row < a.M && col < F) {
                    const float *tp = tile + (bi * 32 + rl) * TPF + bj * 32 + c8;
                    const float4 d0 = ld4(tp), d1 = ld4(tp + 4);
                    const float d[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
                    const int64_t o = (int64_t)row * 2 * F + col;
                    const bf16x8 av = *reinterpret_cast<const bf16x8 *>(a.glu_h + o), bv = *reinterpret_cast<const bf16x8 *>(a.glu_h + o + F);
                    float mk[8];
                    kk_drop_mul4(seed, a.glu_site, (uint64_t)row * F + col, thr, ik, *reinterpret_cast<float(*)[4]>(mk));
                    kk_drop_mul4(seed, a.glu_site, (uint64_t)row * F + col + 4, thr, ik, *reinterpret_cast<float(*)[4]>(mk + 4));
                    bf16x8 oa_, ob_;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float gv, gd;
                        kk_gelu_pair_fast((float)av[e], gv, gd);
                        const float dd = d[e] * mk[e];
                        const float da = dd * (float)bv[e] * gd, db = dd * gv;
                        oa_[e] = (__bf16)da;
                        ob_[e] = (__bf16)db;
                        sa[e] += da;
                        sb[e] += db;
                    }
                    kk_store16(a.glu_dh + o, __builtin_bit_cast(kk_u32x4, oa_), a.wt);
                    kk_store16(a.glu_dh + o + F, __builtin_bit_cast(kk_u32x4, ob_), a.wt);
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                {   // lanes 4 and 8 away inside the 16-lane row by DPP rotations (only lanes 0..3 are read below: for them the same additions as
                // the xor butterfly), the rows 16 and 32 away by ds_bpermute
                sa[e] += kk_dpp<0x124>(sa[e]); sb[e] += kk_dpp<0x124>(sb[e]);
                sa[e] += kk_dpp<0x128>(sa[e]); sb[e] += kk_dpp<0x128>(sb[e]);
                sa[e] += __shfl_xor(sa[e], 16, 64); sb[e] += __shfl_xor(sb[e], 16, 64);
                sa[e] += __shfl_xor(sa[e], 32, 64); sb[e] += __shfl_xor(sb[e], 32, 64);
            }
            }
            const int prow = m0 / 32 + bi;                      // one partial row per 32 rows of dY (kk_gemm_dgrad_glu_blocks)
            if (lane < 4 && col < F && prow < 2 * ((a.M + 63) / 64)) {
                float *pr = a.glu_partials + (int64_t)prow * 2 * F;
                st4(pr + col, make_float4(sa[0], sa[1], sa[2], sa[3]));
                st4(pr + col + 4, make_float4(sa[4], sa[5], sa[6], sa[7]));
                st4(pr + F + col, make_float4(sb[0], sb[1], sb[2], sb[3]));
                st4(pr + F + col + 4, make_float4(sb[4], sb[5], sb[6], sb[7]));
            }
        }
        stamp(14, 0);
        if (KK_DBG(a, 32)) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp(14, 1); }
        return;
    }
    if constexpr (EPI == 2) {
        // (a | b) + bias, ROUNDED to bf16 — what h1 stores and the gate reads — is what the tile holds: [BM][a columns | b columns]
        static_assert(EPI != 2 || NS * STAGE >= BM * TPH * 2, "the staging area holds the bf16 tile");
        const int F = a.N;
        const uint32_t thr = a.glu_seed ? kk_drop_threshold(a.glu_p) : 0u, seed = thr ? *a.glu_seed : 0u;
        const float ik = thr ? 1.f / (1.f - a.glu_p) : 1.f;
        __bf16 *tile = reinterpret_cast<__bf16 *>(smem);
        if (!loader) {
#pragma unroll
            for (int j = 0; j < NI; ++j) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 bq = ld4(bias_lds + blockcol(j) + 4 * half + 8 * g);
#pragma unroll
                    for (int i = 0; i < MI; ++i) {
                        bf16x4 o;
                        o[0] = (__bf16)(acc[i][j][4 * g] + bq.x); o[1] = (__bf16)(acc[i][j][4 * g + 1] + bq.y);
                        o[2] = (__bf16)(acc[i][j][4 * g + 2] + bq.z); o[3] = (__bf16)(acc[i][j][4 * g + 3] + bq.w);
                        *reinterpret_cast<bf16x4 *>(tile + ((wr * MI + i) * 32 + l31) * TPH + blockcol(j) + 4 * half + 8 * g) = o;
                    }
                }
            }
        }
        stamp(9, 1);
        __syncthreads();
        stamp(9, 2);
        __bf16 *h = a.glu_dh, *gout = static_cast<__bf16 *>(a.C);
        constexpr int NBR = BM / 32, NBC = BNH / 32;
#pragma unroll 1
        for (int blk = wave; blk < NBR * NBC; blk += WAVES) {
            const int bi = blk / NBC, bj = blk % NBC;
            const int col = n0 + bj * 32 + c8;
            if (col >= F) continue;
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int rl = it * 16 + (lane >> 2), row = m0 + bi * 32 + rl;
                if (row >= a.M) continue;
                const __bf16 *tp = tile + (bi * 32 + rl) * TPH + bj * 32 + c8;
                const bf16x8 oa_ = *reinterpret_cast<const bf16x8 *>(tp), ob_ = *reinterpret_cast<const bf16x8 *>(tp + BNH);
                float mk[8];
                kk_drop_mul4(seed, a.glu_site, (uint64_t)row * F + col, thr, ik, *reinterpret_cast<float(*)[4]>(mk));
                kk_drop_mul4(seed, a.glu_site, (uint64_t)row * F + col + 4, thr, ik, *reinterpret_cast<float(*)[4]>(mk + 4));
                bf16x8 og;
#pragma unroll
                for (int e = 0; e < 8; ++e) og[e] = (__bf16)(kk_gelu_fast((float)oa_[e]) * (float)ob_[e] * mk[e]);
                const int64_t o = (int64_t)row * 2 * F + col;
                kk_store16(h + o, __builtin_bit_cast(kk_u32x4, oa_), a.wt);
                kk_store16(h + o + F, __builtin_bit_cast(kk_u32x4, ob_), a.wt);
                kk_store16(gout + (int64_t)row * a.ldc + col, __builtin_bit_cast(kk_u32x4, og), a.wt);
            }
        }
        stamp(14, 0);
        if (KK_DBG(a, 32)) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp(14, 1); }
        return;
    }
    if constexpr (EPI == 3) {
        static_assert(EPI != 3 || BN % 64 == 0, "whole heads per workgroup tile");
        static_assert(EPI != 3 || NS * STAGE >= BM * TPH * 2, "the staging area holds the bf16 tile");
        constexpr int PITCH = TPH;
        __bf16 *tile = reinterpret_cast<__bf16 *>(smem);
        if (!loader) {
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int tc = blockcol(j) + 4 * half + 8 * g;
                    const float4 bq = ld4(bias_lds + tc);
#pragma unroll
                    for (int i = 0; i < MI; ++i) {
                        bf16x4 o;
                        o[0] = (__bf16)(acc[i][j][4 * g] + bq.x); o[1] = (__bf16)(acc[i][j][4 * g + 1] + bq.y);
                        o[2] = (__bf16)(acc[i][j][4 * g + 2] + bq.z); o[3] = (__bf16)(acc[i][j][4 * g + 3] + bq.w);
                        *reinterpret_cast<bf16x4 *>(tile + ((wr * MI + i) * 32 + l31) * PITCH + tc) = o;
                    }
                }
        }
        stamp(9, 1);
        __syncthreads();
        stamp(9, 2);
        // Eight lanes per (row, head) vector, 8 consecutive columns (16 bytes) each: 16-byte LDS reads and 16-byte global stores (8-byte
        // write-through stores cost 2.7x per byte).  A head's column block is walked row by row, so the part, its gain and the RoPE
        // flag are uniform over a pass; rotate-half's partner columns (d +- 32) are a second 16-byte read of the tile, normalised
        // with the same rs: no cross-lane traffic but the three steps of the sum of squares.  Same arithmetic in the same order
        // as kk_headnorm_rope (16 lanes x 4 columns): the two give the same bits.
        __bf16 *raw = static_cast<__bf16 *>(a.C);
        constexpr int ET = BM % (NT / 8) == 0 ? NT : 64 * CWV;   // threads of the second phase (all of them when that gives whole passes)
        constexpr int HPR = BN / 64, GROUPS = ET / 8, PPH = BM / GROUPS;
        static_assert(BM % GROUPS == 0, "whole passes per head column block");
        if (ET < NT && threadIdx.x >= ET) return;
        const int u = threadIdx.x & 7, gidx = threadIdx.x >> 3, pu = u ^ 4;
        const int pos0 = m0 % a.hn_S;
#pragma unroll 1
        for (int hl = 0; hl < HPR; ++hl) {
            const int hc = n0 + hl * 64;
            if (hc >= a.N) break;
            stamp(10 + hl, 0);
            const int part = hc / a.hn_H;
            const bool rope = (a.hn_rope_mask >> part) & 1;
            const float *gp = a.hn_gain[part];
            const float4 g0 = ld4(gp + 8 * u), g1 = ld4(gp + 8 * u + 4);
            float4 q0 = g0, q1 = g1;
            if (rope) { q0 = ld4(gp + 8 * pu); q1 = ld4(gp + 8 * pu + 4); }
#pragma unroll
            for (int ps = 0; ps < PPH; ++ps) {
                const int rl = ps * GROUPS + gidx, row = m0 + rl;
                const bf16x8 r8 = *reinterpret_cast<const bf16x8 *>(tile + rl * PITCH + hl * 64 + 8 * u);
                const float4 va = make_float4((float)r8[0], (float)r8[1], (float)r8[2], (float)r8[3]);
                const float4 vb = make_float4((float)r8[4], (float)r8[5], (float)r8[6], (float)r8[7]);
                float ss = (va.x * va.x + va.y * va.y + va.z * va.z + va.w * va.w) + (vb.x * vb.x + vb.y * vb.y + vb.z * vb.z + vb.w * vb.w);
                ss += kk_dpp<0xB1>(ss); ss += kk_dpp<0x4E>(ss); ss += kk_dpp<0x141>(ss);       // (lanes 1, 2, 4 away: DPP, kk_common.h)
                const float rs = 1.f / sqrtf(ss * (1.f / 64.f) + 1.1920928955078125e-7f);
                float4 na = make_float4(va.x * rs * g0.x, va.y * rs * g0.y, va.z * rs * g0.z, va.w * rs * g0.w);
                float4 nb = make_float4(vb.x * rs * g1.x, vb.y * rs * g1.y, vb.z * rs * g1.z, vb.w * rs * g1.w);
                if (rope) {
                    int pos = pos0 + rl;
                    if (row >= a.M) pos = (a.M - 1) % a.hn_S;
                    else if (pos >= a.hn_S) pos %= a.hn_S;
                    const bf16x8 p8 = *reinterpret_cast<const bf16x8 *>(tile + rl * PITCH + hl * 64 + 8 * pu);
                    const float4 oa = make_float4((float)p8[0] * rs * q0.x, (float)p8[1] * rs * q0.y, (float)p8[2] * rs * q0.z, (float)p8[3] * rs * q0.w);
                    const float4 ob_ = make_float4((float)p8[4] * rs * q1.x, (float)p8[5] * rs * q1.y, (float)p8[6] * rs * q1.z, (float)p8[7] * rs * q1.w);
                    const float *cr = a.hn_cos + pos * 64 + 8 * u, *sr = a.hn_sin + pos * 64 + 8 * u;
                    const float4 c0 = ld4(cr), c1 = ld4(cr + 4), s0 = ld4(sr), s1 = ld4(sr + 4);
                    const float sg = u < 4 ? -1.f : 1.f;
                    na = make_float4(na.x * c0.x + sg * oa.x * s0.x, na.y * c0.y + sg * oa.y * s0.y, na.z * c0.z + sg * oa.z * s0.z, na.w * c0.w + sg * oa.w * s0.w);
                    nb = make_float4(nb.x * c1.x + sg * ob_.x * s1.x, nb.y * c1.y + sg * ob_.y * s1.y, nb.z * c1.z + sg * ob_.z * s1.z, nb.w * c1.w + sg * ob_.w * s1.w);
                }
                if (row < a.M) {
                    if (!KK_DBG(a, 64 | 128)) kk_store16(raw + (int64_t)row * a.ldc + hc + 8 * u, __builtin_bit_cast(kk_u32x4, r8), a.wt);      // (probe 128: the raw projection alone is not stored — the price of writing two tensors)
                    bf16x8 n8;
                    n8[0] = (__bf16)na.x; n8[1] = (__bf16)na.y; n8[2] = (__bf16)na.z; n8[3] = (__bf16)na.w;
                    n8[4] = (__bf16)nb.x; n8[5] = (__bf16)nb.y; n8[6] = (__bf16)nb.z; n8[7] = (__bf16)nb.w;
                    if (!KK_DBG(a, 64)) kk_store16(a.hn_y + (int64_t)row * a.hn_ldy + hc + 8 * u, __builtin_bit_cast(kk_u32x4, n8), a.wt);
                }
            }
        }
        stamp(14, 0);
        if (KK_DBG(a, 32)) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp(14, 1); }
        return;
    }
    // ---- EPI == 0
    constexpr int NBR0 = BM / 32, NBP = BN / 64;                // phase-B items: 32 rows x 64 columns (a head of the Delta epilogue)
    if (a.c_bf16 && a.residual == nullptr && (a.ldc & 7) == 0 && (a.N & 7) == 0 && (reinterpret_cast<uintptr_t>(a.C) & 15) == 0) {
        // bf16 C: the tile holds alpha * acc + bias, rounded
        __bf16 *tile = reinterpret_cast<__bf16 *>(smem);
        __bf16 *C = static_cast<__bf16 *>(a.C);
        if (!loader) {
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int tc = blockcol(j) + 4 * half + 8 * g;
                    const float4 bq = ld4(bias_lds + tc);
#pragma unroll
                    for (int i = 0; i < MI; ++i) {
                        bf16x4 o;
                        o[0] = (__bf16)(a.alpha * acc[i][j][4 * g] + bq.x); o[1] = (__bf16)(a.alpha * acc[i][j][4 * g + 1] + bq.y);
                        o[2] = (__bf16)(a.alpha * acc[i][j][4 * g + 2] + bq.z); o[3] = (__bf16)(a.alpha * acc[i][j][4 * g + 3] + bq.w);
                        *reinterpret_cast<bf16x4 *>(tile + ((wr * MI + i) * 32 + l31) * TPH + tc) = o;
                    }
                }
        }
        __syncthreads();
#pragma unroll 1
        for (int blk = wave; blk < NBR0 * NBP; blk += WAVES) {
            const int bi = blk / NBP, bp = blk % NBP;
            float dsum[2] = {0.f, 0.f};
#pragma unroll
            for (int hj = 0; hj < 2; ++hj) {
                const int col = n0 + bp * 64 + hj * 32 + c8;
                if (col >= a.N) continue;
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int rl = it * 16 + (lane >> 2), row = m0 + bi * 32 + rl;
                    if (row >= a.M) continue;
                    const bf16x8 o = *reinterpret_cast<const bf16x8 *>(tile + (bi * 32 + rl) * TPH + bp * 64 + hj * 32 + c8);
                    kk_store16(C + (int64_t)row * a.ldc + col, __builtin_bit_cast(kk_u32x4, o), a.wt);
                    if (a.dl_out != nullptr) {                  // Delta rows (from the ROUNDED dO: what the attention kernels will read)
                        const bf16x8 ov = *reinterpret_cast<const bf16x8 *>(a.dl_o + (int64_t)row * a.dl_ldo + col);
#pragma unroll
                        for (int e = 0; e < 8; ++e) dsum[it] += (float)o[e] * (float)ov[e];
                    }
                }
            }
            if (a.dl_out != nullptr) {                          // the 4 lanes sharing lane >> 2 hold a row's 64 columns of this head
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    float t = dsum[it];
                    t += __shfl_xor(t, 1, 64);
                    t += __shfl_xor(t, 2, 64);
                    const int rl = it * 16 + (lane >> 2), row = m0 + bi * 32 + rl;
                    if ((lane & 3) == 0 && row < a.M && n0 + bp * 64 < a.N) {
                        const int bb = row / a.dl_S, q = row - bb * a.dl_S;
                        a.dl_out[((int64_t)bb * a.dl_heads + (n0 + bp * 64) / 64) * a.dl_S + q] = t;
                    }
                }
            }
        }
        return;
    }
    if (a.wt && !a.c_bf16 && !a.atomic && a.residual == nullptr && (a.ldc & 3) == 0 && (a.N & 7) == 0 &&
        (reinterpret_cast<uintptr_t>(a.C) & 15) == 0 && NS * STAGE >= BM * TPF * 4) {
        // fp32 C written (or accumulated into) exactly once per element — the weight gradients — as 16-byte write-through stores
        float *tile = reinterpret_cast<float *>(smem);
        float *C = static_cast<float *>(a.C);
        if (!loader) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        st4(tile + ((wr * MI + i) * 32 + l31) * TPF + blockcol(j) + 4 * half + 8 * g,
                            make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]));
        }
        __syncthreads();
        float ssq = 0.f;                                         // (ss_rec: this lane's share of the tile's sum of squares)
#pragma unroll 1
        for (int blk = wave; blk < NBR0 * (BN / 32); blk += WAVES) {
            const int bi = blk / (BN / 32), bj = blk % (BN / 32);
            const int col = n0 + bj * 32 + c8;
            if (col >= a.N) continue;
            float bv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) bv[e] = a.bias != nullptr ? a.bias[col + e] : 0.f;
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int rl = it * 16 + (lane >> 2), row = m0 + bi * 32 + rl;
                if (row >= a.M) continue;
                float *dst = C + (int64_t)row * a.ldc + col;
                const float *tp = tile + (bi * 32 + rl) * TPF + bj * 32 + c8;
                const float4 v0 = ld4(tp), v1 = ld4(tp + 4);
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
        if constexpr (NS * STAGE >= BM * TPF * 4 + 8 * (WR * WC + LW)) {      // (room for the wave sums behind the fp32 tile)
        if (a.ss_rec != nullptr) {                              // (workgroup-uniform) wave sums in wave order: the same bits whatever the schedule
            double *wsum = reinterpret_cast<double *>(smem + BM * TPF * 4);
            const double wv = wave_sum_d((double)ssq);
            if (lane == 0) wsum[wave] = wv;
            __syncthreads();
            if (threadIdx.x == 0) {
                double t = 0.0;
#pragma unroll
                for (int w = 0; w < WAVES; ++w) t += wsum[w];
                a.ss_rec[wg] = KkSegRec{t, a.ss_seg + (a.ss_rows > 0 ? m0 / a.ss_rows : 0), 0};
            }
        }
        }
        return;
    }
    // general form (residual, unaligned C, accumulation without write-through): straight from the registers, 4 columns at a time
    if (loader) return;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int row = m0 + (wr * MI + i) * 32 + l31;
            if (row >= a.M) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int col = n0 + blockcol(j) + 4 * half + 8 * (r >> 2) + (r & 3);
                if (col >= a.N) continue;
                float v = a.alpha * acc[i][j][r] + (a.bias != nullptr ? a.bias[col] : 0.f);
                if (a.residual != nullptr) {
                    const int64_t rr = a.res_mod > 0 ? (int64_t)row % a.res_mod : (int64_t)row;
                    v += a.residual[rr * a.ldr + col];
                }
                if (a.c_bf16) {
                    static_cast<__bf16 *>(a.C)[(int64_t)row * a.ldc + col] = (__bf16)v;
                    continue;
                }
                float *dst = static_cast<float *>(a.C) + (int64_t)row * a.ldc + col;
                if (a.beta != 0.f) v += a.beta * (*dst);
                *dst = v;
            }
        }
}

#ifdef KK_BODIES_ONLY
}  // namespace   (kk_chain.hip includes this file for g16x_body only)
#else
template <bool TA, bool TB, int BM, int BN, int NS, int EPI, int WR, int WC, int LW>
__global__ __launch_bounds__(64 * (WR * WC + LW)) void g16x_kernel(G16Args a) {
    __shared__ __attribute__((aligned(16))) char smem[NS * (BM + BN) * BK * 2 + BN * 4];      // (static: up to 145 KB, one workgroup per CU; the tail: the tile's bias row)
    g16x_body<TA, TB, BM, BN, NS, EPI, WR, WC, LW>(a, blockIdx.x, smem);
}
template <bool TA, bool TB, int BM, int BN, int NS, int WR, int WC, int LW>
__global__ __launch_bounds__(64 * (WR * WC + LW)) void g16x_group_kernel(G16Group g) {
    __shared__ __attribute__((aligned(16))) char smem[NS * (BM + BN) * BK * 2 + BN * 4];
    // Which tiles share an XCD's L2.  The dispatcher places workgroup w on XCD w % 8; with xcd_chunks set, XCD x works on the x-th
    // contiguous eighth of the concatenation of ALL problems' tile lists (each in its own sweep order), not on an eighth of every
    // problem: the 30 tiles it runs side by side then come from one or two problems and share operand panels — 115 instead of 188
    // panel fetches per decoder-layer launch from the Infinity Cache into the eight L2s (these launches are bound by the latency of
    // the L2 misses: ~64 requests in flight per CU).
    int w = (int)blockIdx.x;
    if (g.xcd_chunks) {
        const int total = g.start[g.n], q = total >> 3, r = total & 7, xcd = w & 7, in = w >> 3;
        w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + in;
    }
    int i = 0;
    while (i + 1 < g.n && w >= g.start[i + 1]) ++i;
    // (hipcc's host pass rejects a g16x_body specialization named by two kernels when TA = TB = true: the weight-gradient layout is
    // instantiated here only; single weight-gradient GEMMs are groups of one)
    g16x_body<TA, TB, BM, BN, NS, 0, WR, WC, LW>(g.p[i], w - g.start[i], smem);
}

#ifdef KK_TUNING_HOOKS
int g16x_probe_bits = 0;                                        // tools: probe bits for every launch of the family, and the stamp buffer of bit 32
void *g16x_probe_buf = nullptr;
#endif
template <bool TA, bool TB, int BM, int BN, int NS, int EPI, int WR, int WC, int LW>
int launch_x(const G16Args &a0, const char *name, hipStream_t s) {
#ifdef KK_TUNING_HOOKS
    G16Args a = a0;
    a.dbg |= g16x_probe_bits;
    if ((a.dbg & 32) && EPI != 0) a.dl_out = static_cast<float *>(g16x_probe_buf);
#else
    const G16Args &a = a0;
#endif
    kk_note_kernelf("g16x<%d,%d,%d,%d,%d,%d,%d,%d,%d>", (int)TA, (int)TB, BM, BN, NS, EPI, WR, WC, LW);
    if (kk_capture(kk_last_kernel(), a, dim3(a.tiles_m * a.tiles_n), 64 * (WR * WC + LW), 0)) return 0;
    hipLaunchKernelGGL((g16x_kernel<TA, TB, BM, BN, NS, EPI, WR, WC, LW>), dim3(a.tiles_m * a.tiles_n), dim3(64 * (WR * WC + LW)), 0, s, a);
    KK_LAUNCH_CHECK(name);
    return 0;
}

}  // namespace

#ifdef KK_TUNING_HOOKS
void kk_g16x_probe(int bits, void *buf) { g16x_probe_bits = bits; g16x_probe_buf = buf; }
#endif
// Loader-wave form of every launch: 1 = four compute + four loader waves (the product), 0 = every wave loads and computes (the form
// of kk_gemm16.hip), 2 = eight compute + four loader waves.  0 and 2 are A/B arms: they are instantiated in the TOOLS flavour only
// (KK_TUNING_HOOKS, KK_G16X_LW); the product build folds the choice to 1 and carries one kernel per tile and layout.
#ifdef KK_TUNING_HOOKS
int g16x_lw = kk_tune_env("KK_G16X_LW", 1);
int g16x_ns4 = kk_tune_env("KK_G16X_NS4", 0);          // tools: four stages (three k-tiles in flight) on the 128 x 128 tile
#define G16X_LW(lw1, lw0) (g16x_lw ? (lw1) : (lw0))
#else
#define G16X_LW(lw1, lw0) (lw1)
#endif

void kk_g16x_tile(int cfg, int *bm, int *bn) {
    static const int t[G16X_NCFG][2] = {{128, 128}, {256, 128}, {128, 192}, {256, 192}};
    *bm = t[cfg][0];
    *bn = t[cfg][1];
}

int kk_g16x_plain(int cfg, int ta, int tb, const G16Args &a, hipStream_t s) {
    const int lay = (ta ? 2 : 0) | (tb ? 1 : 0);
    if (cfg == G16X_128x128) {
#ifdef KK_TUNING_HOOKS
        if (g16x_ns4 && lay == 0) return launch_x<false, false, 128, 128, 4, 0, 2, 2, 4>(a, "kk_gemm", s);
        if (g16x_ns4 && lay == 1) return launch_x<false, true, 128, 128, 4, 0, 2, 2, 4>(a, "kk_gemm", s);
#endif
        if (lay == 0) return G16X_LW((launch_x<false, false, 128, 128, 3, 0, 2, 2, 4>(a, "kk_gemm", s)), (launch_x<false, false, 128, 128, 3, 0, 4, 2, 0>(a, "kk_gemm", s)));
        if (lay == 1) return G16X_LW((launch_x<false, true, 128, 128, 3, 0, 2, 2, 4>(a, "kk_gemm", s)), (launch_x<false, true, 128, 128, 3, 0, 4, 2, 0>(a, "kk_gemm", s)));
    } else if (cfg == G16X_256x128) {
        if (lay == 0) return G16X_LW((launch_x<false, false, 256, 128, 3, 0, 2, 2, 4>(a, "kk_gemm", s)), (launch_x<false, false, 256, 128, 3, 0, 4, 2, 0>(a, "kk_gemm", s)));
        if (lay == 1) return G16X_LW((launch_x<false, true, 256, 128, 3, 0, 2, 2, 4>(a, "kk_gemm", s)), (launch_x<false, true, 256, 128, 3, 0, 4, 2, 0>(a, "kk_gemm", s)));
    }
    return kk_fail(KK_EINVAL, "kk_g16x_plain: no kernel for tile %d, layout %d", cfg, lay);
}
int kk_g16x_headnorm(int cfg, const G16Args &a, hipStream_t s) {
#ifdef KK_TUNING_HOOKS
    if (cfg == G16X_128x192 && g16x_lw == 2) return launch_x<false, false, 128, 192, 3, 3, 4, 2, 4>(a, "kk_gemm_qkv_headnorm", s);
#endif
    if (cfg == G16X_128x192) return G16X_LW((launch_x<false, false, 128, 192, 3, 3, 2, 2, 4>(a, "kk_gemm_qkv_headnorm", s)), (launch_x<false, false, 128, 192, 3, 3, 4, 2, 0>(a, "kk_gemm_qkv_headnorm", s)));
    if (cfg == G16X_256x192) return launch_x<false, false, 256, 192, 2, 3, 4, 2, 0>(a, "kk_gemm_qkv_headnorm", s);
    if (cfg == G16X_128x128) return G16X_LW((launch_x<false, false, 128, 128, 3, 3, 2, 2, 4>(a, "kk_gemm_qkv_headnorm", s)), (launch_x<false, false, 128, 128, 3, 3, 4, 2, 0>(a, "kk_gemm_qkv_headnorm", s)));
    if (cfg == G16X_256x128) return G16X_LW((launch_x<false, false, 256, 128, 3, 3, 2, 2, 4>(a, "kk_gemm_qkv_headnorm", s)), (launch_x<false, false, 256, 128, 3, 3, 4, 2, 0>(a, "kk_gemm_qkv_headnorm", s)));
    return kk_fail(KK_EINVAL, "kk_g16x_headnorm: no kernel for tile %d", cfg);
}
int kk_g16x_glu_fwd(const G16Args &a0, hipStream_t s) {
    G16Args a = a0;
    a.tiles_n = kk_cdiv(a.N, 96);
#ifdef KK_TUNING_HOOKS
    if (g16x_lw == 2) {                                         // (tools: 128 rows x (96 + 96) columns, four compute waves of 32 x 192 + four loaders: 2 rounds at 4096 rows, slower)
        a.tiles_m = kk_cdiv(a.M, 128);
        return launch_x<false, false, 128, 192, 3, 2, 4, 1, 4>(a, "kk_gemm_linear_glu", s);
    }
#endif
    a.tiles_m = kk_cdiv(a.M, 256);
    // (eight compute waves of 32 x 192 + four loaders, 3 waves per SIMD at 166 registers, measured level: 24.2 against 25.3 us at 4096
    // rows, 51.9 against 50.6 at 8192 — the epilogue's 38 MB of stores is most of this launch; not instantiated)
    return launch_x<false, false, 256, 192, 2, 2, 8, 1, 0>(a, "kk_gemm_linear_glu", s);
}
int kk_g16x_glu_bwd(const G16Args &a, hipStream_t s) {
    return G16X_LW((launch_x<false, true, 128, 192, 3, 1, 2, 2, 4>(a, "kk_gemm_dgrad_glu", s)), (launch_x<false, true, 128, 192, 3, 1, 4, 2, 0>(a, "kk_gemm_dgrad_glu", s)));
}
int kk_g16x_group(const G16Group &g, int grid, hipStream_t s) {
#ifdef KK_TUNING_HOOKS
    kk_note_kernelf("g16x_group<1,1,128,128,3,lw%d>", g16x_lw);
    if (g16x_ns4) hipLaunchKernelGGL((g16x_group_kernel<true, true, 128, 128, 4, 2, 2, 4>), dim3(grid), dim3(512), 0, s, g);
    else if (g16x_lw == 2) hipLaunchKernelGGL((g16x_group_kernel<true, true, 128, 128, 3, 4, 2, 4>), dim3(grid), dim3(768), 0, s, g);
    else if (g16x_lw) hipLaunchKernelGGL((g16x_group_kernel<true, true, 128, 128, 3, 2, 2, 4>), dim3(grid), dim3(512), 0, s, g);
    else hipLaunchKernelGGL((g16x_group_kernel<true, true, 128, 128, 3, 4, 2, 0>), dim3(grid), dim3(512), 0, s, g);
#else
    kk_note_kernel("g16x_group<1,1,128,128,3,lw1>");
    hipLaunchKernelGGL((g16x_group_kernel<true, true, 128, 128, 3, 2, 2, 4>), dim3(grid), dim3(512), 0, s, g);
#endif
    KK_LAUNCH_CHECK("kk_gemm_wgrad_group");
    return 0;
}
#endif  // KK_BODIES_ONLY
