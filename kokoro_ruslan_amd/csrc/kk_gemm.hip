// MFMA GEMM for the Linear layers of the Kokoro train step (forward, dgrad, wgrad).
//
// Replaces the nn.Linear / F.linear call sites of the reference: attention projections
// (model/transformers.py:131-136,228,258-259,434), GLU feed-forward (transformers.py:90-91,106-108),
// mel projections and heads (model/model.py:173,190,523,561), conv-as-GEMM of the variance predictors
// (model/variance_predictor.py:46-51 via kk_im2col3_*), and their autograd backward.
//
// Design (gfx950): 128x128 output tile per 256-thread workgroup, 4 waves in a 2x2 grid, each wave owns
// 64x64 = 2x2 MFMA tiles of 32x32 (4 x 16 accumulator registers).  Operands are fp32 in HBM; they are
// staged global -> registers -> LDS with 16-byte loads along whichever dimension is contiguous in memory,
// so the three layouts a Linear needs (X.W^T, dY.W, dY^T.X) share one kernel: a k-strided operand is
// transposed for free while it is being converted/written to LDS.  LDS rows are padded (80 B for bf16,
// 68 B for fp32) so fragment reads spread over the banks.  Two LDS buffers, the next tile's global loads
// are in flight while the current tile is multiplied.  KK_MATH_BF16 rounds operands to bf16 at staging
// and uses v_mfma_f32_32x32x16_bf16; KK_MATH_F32 keeps fp32 and uses v_mfma_f32_32x32x2_f32 (bit-exact
// fp32 FMA chain) — the parity mode.  Short-M/N problems (weight gradients) split K over blockIdx.z and
// combine with fp32 atomics.
#include "kk_common.h"

namespace {

// Tile configurations: TM x TM output tile, BK reduction slab.  Both stage 4096 (bf16) / 2048 (fp32) operand
// elements per tile, i.e. the same 16 / 8 floats per thread.  The 64-tile (BK twice as deep) is used when the
// 128-tile grid would leave most of the 256 CUs idle — the Linear layers here have M = B*T = 4096..8192 rows and
// N = 512..3072, i.e. only 128..768 128x128 tiles.
template <bool BF16, int TM> struct GemmCfg;
template <> struct GemmCfg<true, 128> { static constexpr int BK = 32, LR = 40, NV = 4; typedef __bf16 elem; };   // 80-byte rows
template <> struct GemmCfg<true, 64> { static constexpr int BK = 64, LR = 72, NV = 4; typedef __bf16 elem; };    // 144-byte rows
template <> struct GemmCfg<false, 128> { static constexpr int BK = 16, LR = 17, NV = 2; typedef float elem; };
template <> struct GemmCfg<false, 64> { static constexpr int BK = 32, LR = 33, NV = 2; typedef float elem; };

struct GemmArgs {
    int M, N, K;
    float alpha, beta;
    const void *A, *B;          // fp32, or bf16 when the kernel is instantiated with A16 / B16
    const float *bias, *residual;
    void *C;                    // fp32, or bf16 when c_bf16 (runtime flag; plain stores only)
    int c_bf16;
    int64_t lda, ldb, ldc, ldr, res_mod;
    int k_per_split, atomic;
    int tiles_m, tiles_n, xcd_swizzle;
};

__device__ __forceinline__ float f4c(const float4 &v, int c) { return reinterpret_cast<const float *>(&v)[c]; }

// Global -> registers for one 128 x BK operand tile.  KS=false: element (row,k) at X[row*ld + k];
// KS=true: element (row,k) at X[k*ld + row].
template <bool BF16, int TM, bool KS>
__device__ __forceinline__ void g2r(const float *__restrict__ X, int64_t ld, int rows_total, int r0, int k0,
                                    int kend, float4 (&reg)[GemmCfg<BF16, TM>::NV]) {
    constexpr int NV = GemmCfg<BF16, TM>::NV;
    constexpr int CPR = GemmCfg<BF16, TM>::BK / 4;          // 16-byte chunks per tile row (k-contiguous operand)
    constexpr int RPI = 256 / CPR;                           // tile rows covered by one load instruction
    constexpr int RG = TM / 4;                               // 4-row groups per tile (k-strided operand)
    const int t = threadIdx.x;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!KS) {
        // every load instruction reads whole contiguous rows (CPR lanes x 16 B): a wave touches 64/CPR full row
        // segments instead of 64 scattered 16-byte pieces (measured: the scattered form ran at ~11 B/clk/CU)
        const int k = k0 + (t % CPR) * 4;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int row = r0 + t / CPR + i * RPI;
            reg[i] = (row < rows_total && k < kend) ? ld4(X + (int64_t)row * ld + k) : z;
        }
    } else {
        const int row = r0 + (t % RG) * 4;
        const int kb = k0 + (t / RG) * NV;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int k = kb + i;
            reg[i] = (row < rows_total && k < kend) ? ld4(X + (int64_t)k * ld + row) : z;
        }
    }
}

// Registers -> LDS tile S[128][LR] (k contiguous), converting to bf16 when BF16.
template <bool BF16, int TM, bool KS>
__device__ __forceinline__ void r2s(typename GemmCfg<BF16, TM>::elem *S, const float4 (&reg)[GemmCfg<BF16, TM>::NV]) {
    constexpr int LR = GemmCfg<BF16, TM>::LR, NV = GemmCfg<BF16, TM>::NV;
    constexpr int CPR = GemmCfg<BF16, TM>::BK / 4, RPI = 256 / CPR, RG = TM / 4;
    const int t = threadIdx.x;
    if constexpr (BF16) {
        if (!KS) {
            const int kofs = (t % CPR) * 4;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                bf16x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (__bf16)f4c(reg[i], e);
                *reinterpret_cast<bf16x4 *>(&S[(t / CPR + i * RPI) * LR + kofs]) = v;
            }
        } else {
            const int rowb = (t % RG) * 4, kofs = (t / RG) * 4;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                bf16x4 v;
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = (__bf16)f4c(reg[i], c);
                *reinterpret_cast<bf16x4 *>(&S[(rowb + c) * LR + kofs]) = v;
            }
        }
    } else {
        if (!KS) {
            const int kofs = (t % CPR) * 4;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) S[(t / CPR + i * RPI) * LR + kofs + e] = f4c(reg[i], e);
        } else {
            const int rowb = (t % RG) * 4, kofs = (t / RG) * 2;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int c = 0; c < 4; ++c) S[(rowb + c) * LR + kofs + i] = f4c(reg[i], c);
        }
    }
}

// ---- operands already stored as bf16 in HBM (bf16 mode: weights' shadow copy, bf16 activations) ----------------
// No conversion and half the bytes: 16-byte loads carry 8 elements.  Two loads per thread per tile for both tile sizes.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));   // native vector: stays in registers (HIP's uint4 struct did not)

template <int TM, int BK, bool KS>
__device__ __forceinline__ void g2r16(const __bf16 *__restrict__ X, int64_t ld, int rows_total, int r0, int k0, int kend,
                                      u32x4 (&reg)[2]) {
    const int t = threadIdx.x;
    const u32x4 z = {0u, 0u, 0u, 0u};
    if (!KS) {
        constexpr int CPR = BK / 8, RPI = 256 / CPR;          // 16-byte chunks per row; rows per instruction
        const int k = k0 + (t % CPR) * 8;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = r0 + t / CPR + i * RPI;
            reg[i] = (row < rows_total && k < kend) ? *reinterpret_cast<const u32x4 *>(X + (int64_t)row * ld + k) : z;
        }
    } else {
        constexpr int RG = TM / 8;                             // 8-row groups; each thread takes 2 consecutive k
        const int row = r0 + (t % RG) * 8;
        const int kb = k0 + (t / RG) * 2;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int k = kb + i;
            reg[i] = (row < rows_total && k < kend) ? *reinterpret_cast<const u32x4 *>(X + (int64_t)k * ld + row) : z;
        }
    }
}

template <int TM, int BK, int LR, bool KS>
__device__ __forceinline__ void r2s16(__bf16 *S, const u32x4 (&reg)[2]) {
    const int t = threadIdx.x;
    if (!KS) {
        constexpr int CPR = BK / 8, RPI = 256 / CPR;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            *reinterpret_cast<u32x4 *>(&S[(t / CPR + i * RPI) * LR + (t % CPR) * 8]) = reg[i];
    } else {
        constexpr int RG = TM / 8;
        const int rowb = (t % RG) * 8, kofs = (t / RG) * 2;
#pragma unroll
        for (int c = 0; c < 8; ++c) {    // (k, k+1) of row c packed into one 4-byte LDS store
            const unsigned int lo = (reg[0][c >> 1] >> (16 * (c & 1))) & 0xFFFFu;
            const unsigned int hi = (reg[1][c >> 1] >> (16 * (c & 1))) & 0xFFFFu;
            *reinterpret_cast<unsigned int *>(&S[(rowb + c) * LR + kofs]) = lo | (hi << 16);
        }
    }
}

template <bool TA, bool TB, bool BF16, int TM, bool A16, bool B16>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs a) {
    using Cfg = GemmCfg<BF16, TM>;
    using elem = typename Cfg::elem;
    constexpr int BK = Cfg::BK, LR = Cfg::LR, NV = Cfg::NV;
    constexpr int BM = TM, BN = TM, WT = TM / 2, MI = WT / 32;   // wave tile WT x WT = MI x MI MFMA tiles
    constexpr int TILE = BM * LR;
    __shared__ __attribute__((aligned(16))) elem smem[4 * TILE];   // A0 A1 B0 B1
    elem *As = smem, *Bs = smem + 2 * TILE;

    // XCD-aware tile order: the dispatcher places workgroup i on XCD i % 8 (each XCD has a private 4 MiB L2).  Remap
    // so every XCD sweeps a CONTIGUOUS run of tiles (n fastest): the A row panel and the whole weight matrix then
    // stay in that XCD's L2 instead of being fetched by all eight.  Bijective for any tile count.
    int tid_lin = blockIdx.x;
    if (a.xcd_swizzle) {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = tid_lin & 7, in = tid_lin >> 3;
        tid_lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + in;
    }
    const int m0 = (tid_lin / a.tiles_n) * BM, n0 = (tid_lin % a.tiles_n) * BN;
    const int kbeg = blockIdx.y * a.k_per_split;
    const int kend = min(a.K, kbeg + a.k_per_split);
    const int nk = (kend - kbeg + BK - 1) / BK;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1, half = lane >> 5, l31 = lane & 31;

    f32x16 acc[MI][MI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    static_assert(BF16 || (!A16 && !B16), "bf16 storage only with bf16 arithmetic");
    float4 ra[NV], rb[NV];
    u32x4 ra16[2], rb16[2];
    auto loadA = [&](int k0) {
        if constexpr (A16) g2r16<TM, BK, TA>(static_cast<const __bf16 *>(a.A), a.lda, a.M, m0, k0, kend, ra16);
        else g2r<BF16, TM, TA>(static_cast<const float *>(a.A), a.lda, a.M, m0, k0, kend, ra);
    };
    auto loadB = [&](int k0) {
        if constexpr (B16) g2r16<TM, BK, TB>(static_cast<const __bf16 *>(a.B), a.ldb, a.N, n0, k0, kend, rb16);
        else g2r<BF16, TM, TB>(static_cast<const float *>(a.B), a.ldb, a.N, n0, k0, kend, rb);
    };
    auto storeA = [&](elem *S) {
        if constexpr (A16) r2s16<TM, BK, LR, TA>(reinterpret_cast<__bf16 *>(S), ra16);
        else r2s<BF16, TM, TA>(S, ra);
    };
    auto storeB = [&](elem *S) {
        if constexpr (B16) r2s16<TM, BK, LR, TB>(reinterpret_cast<__bf16 *>(S), rb16);
        else r2s<BF16, TM, TB>(S, rb);
    };
    if (nk > 0) {
        loadA(kbeg);
        loadB(kbeg);
        storeA(As);
        storeB(Bs);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool more = kt + 1 < nk;
        if (more) {
            loadA(kbeg + (kt + 1) * BK);
            loadB(kbeg + (kt + 1) * BK);
        }
        const elem *Ac = As + cur * TILE + (wr * WT + l31) * LR;
        const elem *Bc = Bs + cur * TILE + (wc * WT + l31) * LR;
        if constexpr (BF16) {
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                bf16x8 af[MI], bf[MI];
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    af[i] = *reinterpret_cast<const bf16x8 *>(Ac + i * 32 * LR + ks * 16 + half * 8);
                    bf[i] = *reinterpret_cast<const bf16x8 *>(Bc + i * 32 * LR + ks * 16 + half * 8);
                }
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < MI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < BK / 2; ++ks) {
                float af[MI], bf[MI];
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    af[i] = Ac[i * 32 * LR + ks * 2 + half];
                    bf[i] = Bc[i * 32 * LR + ks * 2 + half];
                }
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < MI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
        }
        if (more) {
            storeA(As + (cur ^ 1) * TILE);
            storeB(Bs + (cur ^ 1) * TILE);
        }
        __syncthreads();
    }
    if (nk <= 0) return;

    const bool lead = (blockIdx.y == 0);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j) {
            const int col = n0 + wc * WT + j * 32 + l31;
            if (col >= a.N) continue;
            const float bv = (a.bias != nullptr && lead) ? a.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wr * WT + i * 32 + frag_row(r, half);
                if (row >= a.M) continue;
                float v = a.alpha * acc[i][j][r] + bv;
                if (a.residual != nullptr && lead) {
                    const int64_t rr = a.res_mod > 0 ? (int64_t)row % a.res_mod : (int64_t)row;
                    v += a.residual[rr * a.ldr + col];
                }
                if (a.c_bf16) {          // bf16 activation output (never atomic, never accumulated)
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

template <bool BF16, int TM, bool A16, bool B16>
int launch2(int ta, int tb, const GemmArgs &a, dim3 grid, hipStream_t s) {
    if (!ta && !tb) hipLaunchKernelGGL((gemm_kernel<false, false, BF16, TM, A16, B16>), grid, dim3(256), 0, s, a);
    else if (!ta && tb) hipLaunchKernelGGL((gemm_kernel<false, true, BF16, TM, A16, B16>), grid, dim3(256), 0, s, a);
    else if (ta && !tb) hipLaunchKernelGGL((gemm_kernel<true, false, BF16, TM, A16, B16>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((gemm_kernel<true, true, BF16, TM, A16, B16>), grid, dim3(256), 0, s, a);
    KK_LAUNCH_CHECK("kk_gemm");
    return 0;
}

template <bool BF16, int TM>
int launch(int ta, int tb, int a16, int b16, const GemmArgs &a, dim3 grid, hipStream_t s) {
    if constexpr (BF16) {
        if (a16 && b16) return launch2<true, TM, true, true>(ta, tb, a, grid, s);
        if (a16) return launch2<true, TM, true, false>(ta, tb, a, grid, s);
        if (b16) return launch2<true, TM, false, true>(ta, tb, a, grid, s);
    }
    return launch2<BF16, TM, false, false>(ta, tb, a, grid, s);
}

// ---- column sums (bias gradients): out[n] += sum_m X[m,n] -------------------------------------------
template <typename TX>
__global__ __launch_bounds__(256) void colsum_kernel(const TX *__restrict__ X, int64_t ldx, int64_t M, int N,
                                                     float *__restrict__ out, int rows_per_block) {
    // block = 64 columns x 4 row-lanes; grid.x = column groups, grid.y = row slabs
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rl = threadIdx.x >> 6;
    const int64_t rbeg = (int64_t)blockIdx.y * rows_per_block;
    const int64_t rend = rbeg + rows_per_block < M ? rbeg + rows_per_block : M;
    float s = 0.f;
    if (c < N)
#pragma unroll 8
        for (int64_t r = rbeg + rl; r < rend; r += 4) s += (float)X[r * ldx + c];
    red[rl][threadIdx.x & 63] = s;   // (loads above are independent: the compiler keeps several in flight)
    __syncthreads();
    if (rl == 0 && c < N) atomicAdd(&out[c], red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

int g_tm_threshold = 512;   // use 128x128 tiles when they number at least this many
int g_xcd_swizzle = 1;
int g_use_gemm16 = 1;       // route bf16 x bf16 problems to the DMA-staged core (kk_gemm16.hip)

}  // namespace

// Tuning hooks for tools/ (tile-selection thresholds, XCD swizzle, pipeline depth): NOT part of the product ABI — compiled
// only into a library built with KK_TUNING_HOOKS (python -m kokoro_ruslan_amd.build --tuning; include/kokoro_hip_tuning.h).
// The product library's tile policy is fixed at load time (optionally from KK_GEMM16_TUNE, read once in kk_gemm16.hip).
#ifdef KK_TUNING_HOOKS
extern "C" int kk_gemm_tune(int tm_threshold, int xcd_swizzle) {
    g_tm_threshold = tm_threshold;
    g_xcd_swizzle = xcd_swizzle;
    return 0;
}
extern "C" int kk_gemm_tune16(int enable, int thr128, int thr12864, int split_target) {
    g_use_gemm16 = enable;
    if (thr128 > 0) kk_gemm16_tune(thr128, thr12864, split_target);
    return 0;
}
extern "C" int kk_gemm_tune_group(int split) { kk_gemm16_tune_group(split); return 0; }
#endif

extern "C" int kk_gemm(int ta, int tb, int64_t M, int64_t N, int64_t K, float alpha, const float *A, int64_t lda,
                       const float *B, int64_t ldb, float beta, float *C, int64_t ldc, const float *bias,
                       const float *residual, int64_t ldr, int64_t res_mod, int split_k, int math, int dtypes,
                       void *stream) {
    const int a16 = dtypes & 1, b16 = (dtypes >> 1) & 1, c16 = (dtypes >> 2) & 1;
    KK_REQUIRE(dtypes == 0 || math == KK_MATH_BF16, "kk_gemm: bf16 storage needs KK_MATH_BF16");
    KK_REQUIRE(!c16 || beta == 0.f, "kk_gemm: a bf16 C cannot be accumulated into");
    if (a16) KK_REQUIRE(lda % 8 == 0 && (ta ? M % 8 == 0 : K % 8 == 0), "kk_gemm: bf16 A needs lda and its contiguous extent to be multiples of 8");
    if (b16) KK_REQUIRE(ldb % 8 == 0 && (tb ? N % 8 == 0 : K % 8 == 0), "kk_gemm: bf16 B needs ldb and its contiguous extent to be multiples of 8");
    KK_REQUIRE(M > 0 && N > 0 && K > 0, "kk_gemm: empty problem M=%ld N=%ld K=%ld", (long)M, (long)N, (long)K);
    KK_REQUIRE(M < (1ll << 31) && N < (1ll << 31) && K < (1ll << 31), "kk_gemm: dimension overflow");
    KK_REQUIRE(A && B && C, "kk_gemm: null operand");
    KK_REQUIRE(lda % 4 == 0 && ldb % 4 == 0, "kk_gemm: lda/ldb must be multiples of 4 (16-byte loads)");
    KK_REQUIRE(((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0, "kk_gemm: A/B must be 16-byte aligned");
    if (!ta || !tb) KK_REQUIRE(K % 4 == 0, "kk_gemm: K=%ld must be a multiple of 4 for a k-contiguous operand", (long)K);
    if (ta) KK_REQUIRE(M % 4 == 0, "kk_gemm: M=%ld must be a multiple of 4 when A is stored [K,M]", (long)M);
    if (tb) KK_REQUIRE(N % 4 == 0, "kk_gemm: N=%ld must be a multiple of 4 when B is stored [K,N]", (long)N);
    KK_REQUIRE(math == KK_MATH_F32 || math == KK_MATH_BF16, "kk_gemm: bad math mode %d", math);
    hipStream_t s = (hipStream_t)stream;
    if (a16 && b16 && g_use_gemm16 && kk_gemm16_eligible(ta, tb, M, N, K, A, lda, B, ldb))
        return kk_gemm16_launch(ta, tb, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, c16, bias, residual, ldr, res_mod, split_k,
                                g_xcd_swizzle, s);
    const int TM = (int64_t)kk_cdiv(M, 128) * kk_cdiv(N, 128) >= g_tm_threshold ? 128 : 64;
    const int BK = (math == KK_MATH_BF16 ? 32 : 16) * (TM == 128 ? 1 : 2);
    const int tiles = kk_cdiv(M, TM) * kk_cdiv(N, TM);
    const int ktiles = kk_cdiv(K, BK);
    int splits = split_k;
    if (splits <= 0) {   // auto: aim at ~2 workgroups per CU, keep >= 2 k-tiles per slice
        splits = 1;
        if (tiles < 384) {
            splits = kk_cdiv(512, tiles);
            const int cap = ktiles / 2 > 0 ? ktiles / 2 : 1;
            if (splits > cap) splits = cap;
        }
    }
    if (splits > ktiles) splits = ktiles;
    if (splits > 1 && (c16 || !(beta == 1.f || (beta == 0.f && ldc == N)))) splits = 1;
    int k_per_split = kk_cdiv(ktiles, splits) * BK;
    splits = kk_cdiv(K, k_per_split);
    GemmArgs a;
    a.M = (int)M; a.N = (int)N; a.K = (int)K;
    a.alpha = alpha; a.beta = beta;
    a.A = A; a.B = B; a.bias = bias; a.residual = residual; a.C = C; a.c_bf16 = c16;
    a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.ldr = ldr; a.res_mod = res_mod;
    a.k_per_split = k_per_split;
    a.atomic = splits > 1 ? 1 : 0;
    if (splits > 1 && beta == 0.f) {
        const int e = kk_zero_async(C, (size_t)M * N * sizeof(float), s);
        if (e != 0) return e;
    }
    a.tiles_m = kk_cdiv(M, TM);
    a.tiles_n = kk_cdiv(N, TM);
    a.xcd_swizzle = g_xcd_swizzle;
    dim3 grid(a.tiles_m * a.tiles_n, splits);
    if (TM == 128) return math == KK_MATH_BF16 ? launch<true, 128>(ta, tb, a16, b16, a, grid, s) : launch<false, 128>(ta, tb, 0, 0, a, grid, s);
    return math == KK_MATH_BF16 ? launch<true, 64>(ta, tb, a16, b16, a, grid, s) : launch<false, 64>(ta, tb, 0, 0, a, grid, s);
}

extern "C" int kk_colsum_acc(const float *X, int64_t ldx, int64_t M, int64_t N, float *out, int x_bf16, void *stream) {
    KK_REQUIRE(M > 0 && N > 0 && X && out, "kk_colsum_acc: bad args");
    int slabs = kk_cdiv(M, 64);
    if (slabs > 512) slabs = 512;
    const int rows_per_block = kk_cdiv(M, slabs);
    slabs = kk_cdiv(M, rows_per_block);
    if (x_bf16)
        hipLaunchKernelGGL(colsum_kernel<__bf16>, dim3(kk_cdiv(N, 64), slabs), dim3(256), 0, (hipStream_t)stream,
                           reinterpret_cast<const __bf16 *>(X), ldx, M, (int)N, out, rows_per_block);
    else
        hipLaunchKernelGGL(colsum_kernel<float>, dim3(kk_cdiv(N, 64), slabs), dim3(256), 0, (hipStream_t)stream, X, ldx, M,
                           (int)N, out, rows_per_block);
    KK_LAUNCH_CHECK("kk_colsum_acc");
    return 0;
}

// dG = dY.W2 with the GLU gate's backward as the epilogue (bf16 operands; see gemm16_kernel, EPI = 1).
extern "C" int kk_gemm_dgrad_glu_blocks(int64_t T) { return 2 * kk_cdiv(T, 64); }
extern "C" int kk_gemm_dgrad_glu(int64_t T, int64_t F, int64_t H, const void *dy, int64_t lddy, const void *W, const void *h1,
                                 void *dh1, float *partials, const uint32_t *seed, uint32_t site, float p, void *stream) {
    KK_REQUIRE(T > 0 && F > 0 && H > 0 && dy && W && h1 && dh1 && partials, "kk_gemm_dgrad_glu: bad args");
    KK_REQUIRE(p >= 0.f && p < 1.f, "kk_gemm_dgrad_glu: dropout probability must be in [0,1)");
    KK_REQUIRE(F % 8 == 0, "kk_gemm_dgrad_glu: F must be a multiple of 8 (16-byte epilogue accesses)");
    KK_REQUIRE(kk_gemm16_eligible(0, 1, T, F, H, dy, lddy, W, F), "kk_gemm_dgrad_glu: needs 16-byte aligned bf16 operands, H %% 64 == 0, F %% 8 == 0");
    return kk_gemm16_dgrad_glu(T, F, H, dy, lddy, W, h1, dh1, partials, seed, site, p, g_xcd_swizzle, (hipStream_t)stream);
}

// dX = dY.W of an attention output projection (bf16 operands and result) with the attention backward's row term
// Delta[b, head, q] = sum_d dX[b*S+q, 64 head + d] * O[b*S+q, 64 head + d] as the epilogue (see gemm16_body, DELTA_OK).
extern "C" int kk_gemm_dgrad_delta_supported(int64_t M, int64_t N, int64_t K) {
    return (g_use_gemm16 && kk_gemm16_dgrad_delta_supported(M, N, K)) ? 1 : 0;
}
extern "C" int kk_gemm_dgrad_delta(int64_t M, int64_t N, int64_t K, const void *dy, int64_t lddy, const void *W, int64_t ldw,
                                   void *dx, int64_t lddx, const void *O, int64_t ldo, float *delta, int S, int heads, void *stream) {
    KK_REQUIRE(M > 0 && N > 0 && K > 0 && dy && W && dx && O && delta, "kk_gemm_dgrad_delta: bad args");
    KK_REQUIRE(S > 0 && M % S == 0 && heads * 64 == N, "kk_gemm_dgrad_delta: rows must be whole sequences of S and N = heads x 64");
    KK_REQUIRE(lddx % 8 == 0 && ldo % 8 == 0 && ((uintptr_t)dx & 15) == 0 && ((uintptr_t)O & 15) == 0,
               "kk_gemm_dgrad_delta: dX and O must be 16-byte aligned with row strides %% 8 == 0");
    KK_REQUIRE(kk_gemm16_eligible(0, 1, M, N, K, dy, lddy, W, ldw), "kk_gemm_dgrad_delta: needs 16-byte aligned bf16 operands and K %% 64 == 0");
    KK_REQUIRE(kk_gemm_dgrad_delta_supported(M, N, K), "kk_gemm_dgrad_delta: shape %ldx%ldx%ld does not take the eight-wave 128x64 tile "
               "(ask kk_gemm_dgrad_delta_supported first)", (long)M, (long)N, (long)K);
    return kk_gemm16_dgrad_delta(M, N, K, dy, lddy, W, ldw, dx, lddx, O, ldo, delta, S, heads, g_xcd_swizzle, (hipStream_t)stream);
}

// h1 = x.W1^T + b1 with the GLU gate as the epilogue (bf16 operands; see gemm16_kernel, EPI = 2).
extern "C" int kk_gemm_linear_glu(int64_t T, int64_t F, int64_t K, const void *x, int64_t ldx, const void *W, const float *bias,
                                  void *h1, void *g, int64_t ldg, const uint32_t *seed, uint32_t site, float p, void *stream) {
    KK_REQUIRE(T > 0 && F > 0 && K > 0 && x && W && h1 && g, "kk_gemm_linear_glu: bad args");
    KK_REQUIRE(p >= 0.f && p < 1.f, "kk_gemm_linear_glu: dropout probability must be in [0,1)");
    KK_REQUIRE(kk_gemm16_eligible(0, 0, T, 2 * F, K, x, ldx, W, K), "kk_gemm_linear_glu: needs 16-byte aligned bf16 operands and K %% 64 == 0");
    KK_REQUIRE(F % 8 == 0 && ldg % 8 == 0, "kk_gemm_linear_glu: F and the row stride of g must be multiples of 8 (16-byte epilogue stores)");
    return kk_gemm16_linear_glu(T, F, K, x, ldx, W, bias, h1, g, ldg, seed, site, p, g_xcd_swizzle, (hipStream_t)stream);
}

// A layer's weight gradients as one grouped launch (bf16 operands, fp32 accumulate into dW).
extern "C" int kk_gemm_wgrad_group(const KkWgradDesc *descs, int n, int split_k, int overwrite, void *ss_rec, const int32_t *ss_seg,
                                   int32_t *ss_count, void *stream) {
    KK_REQUIRE(descs != nullptr, "kk_gemm_wgrad_group: null descriptor table");
    KK_REQUIRE(split_k >= 0 && split_k < 100, "kk_gemm_wgrad_group: split_k out of range");
    KK_REQUIRE(!overwrite || split_k <= 1, "kk_gemm_wgrad_group: overwrite cannot be combined with k-slices (they accumulate with atomics)");
    KK_REQUIRE(ss_rec == nullptr || (ss_seg != nullptr && ss_count != nullptr && *ss_count >= 0), "kk_gemm_wgrad_group: ss_rec needs ss_seg and ss_count");
    return kk_gemm16_wgrad_group(descs, n, split_k, overwrite, g_xcd_swizzle, (hipStream_t)stream, ss_rec, ss_seg, ss_count);
}

// q / k / v projection with the per-head RMSNorm (+ RoPE) as the epilogue (bf16 operands; see gemm16_kernel, EPI = 3).
extern "C" int kk_gemm_qkv_headnorm(int64_t T, int parts, int heads, int64_t K, const void *x, int64_t ldx, const void *W,
                                    const float *bias, void *raw, int64_t ldraw, void *y, int64_t ldy, int S,
                                    const float *const *gains, int rope_mask, const float *cos_t, const float *sin_t, void *stream) {
    KK_REQUIRE(T > 0 && parts >= 1 && parts <= 12 && heads > 0 && K > 0 && S > 0 && x && W && raw && y && gains,
               "kk_gemm_qkv_headnorm: bad args (1..12 parts)");
    for (int i = 0; i < parts; ++i) KK_REQUIRE(gains[i] != nullptr, "kk_gemm_qkv_headnorm: null gain vector");
    KK_REQUIRE(rope_mask == 0 || (cos_t && sin_t), "kk_gemm_qkv_headnorm: RoPE needs cos/sin tables");
    KK_REQUIRE(ldraw % 4 == 0 && ldy % 4 == 0 && ((uintptr_t)raw & 7) == 0 && ((uintptr_t)y & 7) == 0,
               "kk_gemm_qkv_headnorm: outputs must be 8-byte aligned with row strides %% 4 == 0");
    KK_REQUIRE(kk_gemm16_eligible(0, 0, T, (int64_t)parts * heads * 64, K, x, ldx, W, K),
               "kk_gemm_qkv_headnorm: needs 16-byte aligned bf16 operands and K %% 64 == 0");
    return kk_gemm16_qkv_headnorm(T, parts, heads, K, x, ldx, W, bias, raw, ldraw, y, ldy, S, gains, rope_mask, cos_t, sin_t,
                                  g_xcd_swizzle, (hipStream_t)stream);
}
