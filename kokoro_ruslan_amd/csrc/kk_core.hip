// Error plumbing, ABI version and an MFMA fragment-layout probe.
#include "kk_common.h"
#include <mutex>
#include <string>
#include <unordered_set>

static thread_local char g_err[512] = "";

int kk_fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char *kk_last_error(void) { return g_err; }

// route record (kk_last_kernel): a pointer to a string with static storage duration, set by the entry points' dispatch code
static thread_local const char *g_last_kernel = "";
void kk_note_kernel(const char *name) { g_last_kernel = name; }
void kk_note_kernelf(const char *fmt, ...) {                       // formatted names are interned: the record outlives the call
    static std::mutex mu;
    static std::unordered_set<std::string> names;
    char buf[96];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    std::lock_guard<std::mutex> lock(mu);
    g_last_kernel = names.emplace(buf).first->c_str();
}
extern "C" const char *kk_last_kernel(void) { return g_last_kernel; }
extern "C" int kk_abi_version(void) { return KK_ABI_VERSION; }

// launch capture (kk_common.h): per thread, so a trainer's prefetch thread is never affected
static thread_local KkLaunchCapture g_capture;
static thread_local bool g_capturing = false;
KkLaunchCapture *kk_capture_target() { return g_capturing ? &g_capture : nullptr; }
KkLaunchCapture *kk_capture_begin() {
    g_capture.n = 0;
    g_capture.overflow = 0;
    g_capturing = true;
    return &g_capture;
}
KkLaunchCapture *kk_capture_end() {
    g_capturing = false;
    return &g_capture;
}

namespace {
// One wave computes C[32x32] = A[32x16] * B[16x32] with A[i][k] = i + 0.25*k - 3, B[k][j] = 0.5*k - 0.125*j + 1
// (asymmetric, exactly representable in bf16 products' fp32 sums), through both MFMA flavours, using the
// operand/result lane maps every kernel in this library assumes.
__global__ void probe_kernel(float *out_f32, float *out_bf16) {
    const int lane = threadIdx.x & 63, half = lane >> 5, l31 = lane & 31;
    f32x16 c0, c1;
    for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; }
    for (int ks = 0; ks < 8; ++ks) {
        const int k = ks * 2 + half;
        const float a = (float)l31 + 0.25f * k - 3.f;      // A[row=l31][k]
        const float b = 0.5f * k - 0.125f * l31 + 1.f;     // B[k][col=l31]
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
    }
    bf16x8 av, bv;
    for (int j = 0; j < 8; ++j) {
        const int k = half * 8 + j;
        av[j] = (__bf16)((float)l31 + 0.25f * k - 3.f);
        bv[j] = (__bf16)(0.5f * k - 0.125f * l31 + 1.f);
    }
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, c1, 0, 0, 0);
    for (int r = 0; r < 16; ++r) {
        const int row = frag_row(r, half);
        out_f32[row * 32 + l31] = c0[r];
        out_bf16[row * 32 + l31] = c1[r];
    }
}

__global__ void axpby_kernel(float a, const float *__restrict__ x, float b, float *__restrict__ y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = a * x[i] + (b == 0.f ? 0.f : b * y[i]);
}
}  // namespace

extern "C" int kk_mfma_probe(float *out_f32, float *out_bf16, void *stream) {
    hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out_f32, out_bf16);
    KK_LAUNCH_CHECK("kk_mfma_probe");
    return 0;
}

extern "C" int kk_axpby(float a, const float *x, float b, float *y, int64_t n, void *stream) {
    KK_REQUIRE(n >= 0 && x && y, "kk_axpby: bad args");
    if (n == 0) return 0;
    int blocks = kk_cdiv(n, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(axpby_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, x, b, y, n);
    KK_LAUNCH_CHECK("kk_axpby");
    return 0;
}

// Device-side time stamp: *slot = the GPU's constant-rate wall clock (100 MHz ticks) when this one-thread kernel runs.
// A captured step dotted with these is the only timeline that shows how the graph's branches really overlap
// (rocprofv3 serialises them); see KokoroEngine.timeline().
namespace {
__global__ void timestamp_kernel(uint64_t *slot) { *slot = wall_clock64(); }
}  // namespace
extern "C" int kk_timestamp(uint64_t *slot, void *stream) {
    KK_REQUIRE(slot != nullptr, "kk_timestamp: null slot");
    hipLaunchKernelGGL(timestamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, slot);
    KK_LAUNCH_CHECK("kk_timestamp");
    return 0;
}

namespace {
__global__ __launch_bounds__(256) void zero_words_kernel(uint32_t *p, size_t words) {
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
    const size_t i0 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4, stride = (size_t)gridDim.x * 256 * 4;
    const bool aligned = ((uintptr_t)p & 15) == 0;
    for (size_t i = i0; i < words; i += stride) {
        if (aligned && i + 4 <= words) {
            const u32x4_t z = {0u, 0u, 0u, 0u};
            *reinterpret_cast<u32x4_t *>(p + i) = z;
        } else {
            for (size_t j = i; j < i + 4 && j < words; ++j) p[j] = 0u;
        }
    }
}
}  // namespace
// Up to ZERO_MAX ranges zero-filled by ONE launch (the gradient arena minus what a step's grouped weight-gradient launches
// overwrite anyway: ~100 small ranges between the weight matrices).  blockIdx.x walks 16 KiB chunks of the concatenation.
namespace {
constexpr int ZERO_MAX = 160, ZERO_CHUNK = 16384;
struct ZeroMany {
    int n;
    int start[ZERO_MAX + 1];
    char *dst[ZERO_MAX];
    int64_t bytes[ZERO_MAX];
};
__global__ __launch_bounds__(256) void zero_many_kernel(ZeroMany z) {
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
    int lo = 0, hi = z.n - 1;                                   // the range this chunk belongs to (start[] is ascending)
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (z.start[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const int64_t off = (int64_t)((int)blockIdx.x - z.start[lo]) * ZERO_CHUNK;
    const int64_t left = z.bytes[lo] - off;
    char *d = z.dst[lo] + off;
    const u32x4_t zero = {0u, 0u, 0u, 0u};
    for (int64_t i = (int64_t)threadIdx.x * 16; i < ZERO_CHUNK && i < left; i += 256 * 16) *reinterpret_cast<u32x4_t *>(d + i) = zero;
}
}  // namespace
extern "C" int kk_zero_many(void *const *dst, const int64_t *bytes, int n, void *stream) {
    KK_REQUIRE(dst && bytes && n >= 1 && n <= ZERO_MAX, "kk_zero_many: 1..%d ranges per call", ZERO_MAX);
    ZeroMany z = {};
    z.n = n;
    for (int i = 0; i < n; ++i) {
        KK_REQUIRE(dst[i] && bytes[i] > 0 && bytes[i] % 16 == 0 && ((uintptr_t)dst[i] & 15) == 0,
                   "kk_zero_many: ranges must be 16-byte aligned multiples of 16 bytes");
        z.dst[i] = static_cast<char *>(dst[i]);
        z.bytes[i] = bytes[i];
        z.start[i + 1] = z.start[i] + (int)((bytes[i] + ZERO_CHUNK - 1) / ZERO_CHUNK);
    }
    hipLaunchKernelGGL(zero_many_kernel, dim3(z.start[n]), dim3(256), 0, (hipStream_t)stream, z);
    KK_LAUNCH_CHECK("kk_zero_many");
    return 0;
}

int kk_zero_async(void *p, size_t bytes, hipStream_t s) {
    if (bytes == 0) return 0;
    if (!p || ((uintptr_t)p & 3) || (bytes & 3)) return kk_fail(KK_EINVAL, "kk_zero_async: needs a 4-byte aligned range");
    const size_t words = bytes / 4;
    size_t blocks = (words + 1023) / 1024;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)blocks), dim3(256), 0, s, static_cast<uint32_t *>(p), words);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return kk_fail((int)e, "kk_zero_async: launch failed: %s", hipGetErrorString(e));
    return 0;
}
