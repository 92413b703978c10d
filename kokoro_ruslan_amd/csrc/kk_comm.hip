// Data-parallel gradient exchange over RCCL (xGMI), behind the C ABI.
//
// The reference has no distributed code; the contract is "the same maths as one process seeing the global batch".
// One communicator per process (one process per GPU).  RCCL is bound at run time (dlopen) so that the library loads —
// and every other entry point works — on a host without RCCL, and so that the process uses the ONE RCCL instance that
// is already loaded (PyTorch-ROCm ships its own librccl.so.1 next to its own HIP runtime; a second copy would bring a
// second runtime).  Every collective is enqueued on the caller's stream and never synchronises, so the calls can be
// captured into the step's hipGraph: the exchange of a bucket then is a branch of the graph beside the rest of the
// backward, not a separate graph launch.
#include "kk_common.h"
#include <dlfcn.h>
#include <string.h>

namespace {

typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
enum { ncclSum = 0, ncclMax = 2, ncclInt64 = 4, ncclFloat32 = 7, ncclFloat64 = 8, ncclBfloat16 = 9 };

struct Rccl {
    void *handle = nullptr;
    int (*GetUniqueId)(ncclUniqueId *) = nullptr;
    int (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*ReduceScatter)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;
ncclComm_t g_comm = nullptr;
int g_rank = 0, g_world = 1;

int load_rccl(const char *path) {
    if (g_rccl.handle) return 0;
    void *h = nullptr;
    if (path && *path) h = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
    const char *names[] = {"librccl.so.1", "librccl.so"};
    for (int pass = 0; pass < 2 && !h; ++pass)            // first an instance that is already in the process, then the search path
        for (const char *n : names)
            if (!h) h = dlopen(n, pass == 0 ? (RTLD_NOW | RTLD_NOLOAD) : (RTLD_NOW | RTLD_GLOBAL));
    if (!h) return kk_fail(KK_ENOTSUP, "kk_comm: librccl.so.1 not found (%s)", dlerror());
    Rccl r;
    r.handle = h;
#define KK_SYM(field, name)                                                                      \
    *(void **)(&r.field) = dlsym(h, name);                                                       \
    if (!r.field) return kk_fail(KK_ENOTSUP, "kk_comm: %s missing from the RCCL library", name);
    KK_SYM(GetUniqueId, "ncclGetUniqueId")
    KK_SYM(CommInitRank, "ncclCommInitRank")
    KK_SYM(CommDestroy, "ncclCommDestroy")
    KK_SYM(AllReduce, "ncclAllReduce")
    KK_SYM(ReduceScatter, "ncclReduceScatter")
    KK_SYM(AllGather, "ncclAllGather")
    KK_SYM(GroupStart, "ncclGroupStart")
    KK_SYM(GroupEnd, "ncclGroupEnd")
    KK_SYM(GetErrorString, "ncclGetErrorString")
#undef KK_SYM
    g_rccl = r;
    return 0;
}

int check(int rc, const char *what) {
    if (rc == 0) return 0;
    return kk_fail(KK_EINVAL, "%s: RCCL error %d (%s)", what, rc, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?");
}
int dtype_of(int dtype, const char *what, int *out) {
    if (dtype == 0) { *out = ncclFloat32; return 0; }
    if (dtype == 1) { *out = ncclBfloat16; return 0; }
    return kk_fail(KK_EINVAL, "%s: dtype must be 0 (fp32) or 1 (bf16)", what);
}

__global__ __launch_bounds__(256) void cast_bf16_f32_kernel(const __bf16 *__restrict__ x, float *__restrict__ y, int64_t n, float scale) {
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * 1024) {
        if (i + 4 <= n) {
            const float4 v = ldv4<__bf16>(x + i);
            st4(y + i, make_float4(v.x * scale, v.y * scale, v.z * scale, v.w * scale));
        } else {
            for (int64_t j = i; j < n; ++j) y[j] = (float)x[j] * scale;
        }
    }
}

// bf16 gradient payload: all ranges of a bucket narrowed (or widened back) by ONE launch of at most CAST_WGS (128) workgroups.  The
// exchange runs on the communication stream beside the backward's one-workgroup-per-CU launches; the per-range casts of round 2
// (two launches per range, up to 8192 workgroups each) made the one-GPU step 42 % slower (profiles/r04_dp_exchange_one_gpu_ab.txt) —
// a thin grid costs the chain next to nothing and still moves a 16 MB bucket in ~30 us, hidden beside the backward.
constexpr int CAST_MAX = 48, CAST_WGS = 128, CAST_CHUNK = 8192;
struct CastRanges {
    int n, to_bf16;
    float scale;
    int64_t begin[CAST_MAX];
    int64_t chunk0[CAST_MAX + 1];                                // first chunk of each range in the concatenation
    int64_t len[CAST_MAX];
};
__global__ __launch_bounds__(256) void cast_ranges_kernel(const void *__restrict__ src, void *__restrict__ dst, CastRanges r) {
    const int64_t total = r.chunk0[r.n];
    int i = 0;
    for (int64_t c = blockIdx.x; c < total; c += gridDim.x) {
        while (c >= r.chunk0[i + 1]) ++i;                        // (chunks ascend per workgroup)
        const int64_t off = (c - r.chunk0[i]) * CAST_CHUNK, left = r.len[i] - off, e0 = r.begin[i] + off;
        if (left >= CAST_CHUNK) {
            // whole chunk: the four 16-byte loads of a thread are in flight together (a thin grid lives on bytes in flight per thread:
            // one load per round trip made a 16 MB bucket take ~90 us on 64 workgroups, and the 26 casts of a step longer than its backward)
            if (r.to_bf16) {
                const float *sp = static_cast<const float *>(src) + e0 + threadIdx.x * 4;
                float4 v[CAST_CHUNK / 1024];
#pragma unroll
                for (int k = 0; k < CAST_CHUNK / 1024; ++k) v[k] = ld4(sp + k * 1024);
#pragma unroll
                for (int k = 0; k < CAST_CHUNK / 1024; ++k) stv4<__bf16>(static_cast<__bf16 *>(dst) + e0 + k * 1024 + threadIdx.x * 4, v[k]);
            } else {
                const __bf16 *sp = static_cast<const __bf16 *>(src) + e0 + threadIdx.x * 4;
                float4 v[CAST_CHUNK / 1024];
#pragma unroll
                for (int k = 0; k < CAST_CHUNK / 1024; ++k) v[k] = ldv4<__bf16>(sp + k * 1024);
#pragma unroll
                for (int k = 0; k < CAST_CHUNK / 1024; ++k)
                    st4(static_cast<float *>(dst) + e0 + k * 1024 + threadIdx.x * 4, make_float4(v[k].x * r.scale, v[k].y * r.scale, v[k].z * r.scale, v[k].w * r.scale));
            }
            continue;
        }
        for (int64_t q = threadIdx.x; q < left; q += 256) {      // ragged tail of a range
            if (r.to_bf16) static_cast<__bf16 *>(dst)[e0 + q] = (__bf16) static_cast<const float *>(src)[e0 + q];
            else static_cast<float *>(dst)[e0 + q] = (float)static_cast<const __bf16 *>(src)[e0 + q] * r.scale;
        }
    }
}

}  // namespace

// dst[begin_i .. end_i) = cast(src[begin_i .. end_i)) for i < n (element offsets into two arrays of the SAME layout: the fp32 gradient
// arena and its bf16 payload twin), to_bf16 = 1: fp32 -> bf16, 0: bf16 -> fp32 times `scale`.  begin[i] must be a multiple of 4.
extern "C" int kk_cast_ranges(const void *src, void *dst, const int64_t *begin, const int64_t *end, int n, int to_bf16, float scale,
                              void *stream) {
    KK_REQUIRE(src && dst && begin && end && n >= 1, "kk_cast_ranges: bad arguments");
    for (int i0 = 0; i0 < n; i0 += CAST_MAX) {
        CastRanges r = {};
        r.n = n - i0 < CAST_MAX ? n - i0 : CAST_MAX;
        r.to_bf16 = to_bf16 ? 1 : 0;
        r.scale = scale;
        for (int i = 0; i < r.n; ++i) {
            KK_REQUIRE(end[i0 + i] > begin[i0 + i] && begin[i0 + i] % 4 == 0, "kk_cast_ranges: empty or unaligned range %d", i0 + i);
            r.begin[i] = begin[i0 + i];
            r.len[i] = end[i0 + i] - begin[i0 + i];
            r.chunk0[i + 1] = r.chunk0[i] + (r.len[i] + CAST_CHUNK - 1) / CAST_CHUNK;
        }
        const int64_t total = r.chunk0[r.n];
        hipLaunchKernelGGL(cast_ranges_kernel, dim3((unsigned)(total < CAST_WGS ? total : CAST_WGS)), dim3(256), 0, (hipStream_t)stream, src, dst, r);
        KK_LAUNCH_CHECK("kk_cast_ranges");
    }
    return 0;
}

extern "C" int kk_comm_load(const char *rccl_path) { return load_rccl(rccl_path); }

extern "C" int kk_comm_unique_id(void *id128) {
    KK_REQUIRE(id128 != nullptr, "kk_comm_unique_id: null buffer");
    if (int rc = load_rccl(nullptr)) return rc;
    ncclUniqueId id;
    if (int rc = check(g_rccl.GetUniqueId(&id), "kk_comm_unique_id")) return rc;
    memcpy(id128, &id, sizeof(id));
    return 0;
}

extern "C" int kk_comm_init(int rank, int world, const void *nccl_unique_id) {
    KK_REQUIRE(world >= 1 && rank >= 0 && rank < world && nccl_unique_id, "kk_comm_init: bad rank %d / world %d", rank, world);
    KK_REQUIRE(g_comm == nullptr, "kk_comm_init: a communicator already exists (kk_comm_destroy first)");
    if (int rc = load_rccl(nullptr)) return rc;
    ncclUniqueId id;
    memcpy(&id, nccl_unique_id, sizeof(id));
    if (int rc = check(g_rccl.CommInitRank(&g_comm, world, id, rank), "kk_comm_init")) { g_comm = nullptr; return rc; }
    g_rank = rank;
    g_world = world;
    return 0;
}

extern "C" int kk_comm_world(void) { return g_comm ? g_world : 0; }

extern "C" int kk_comm_destroy(void) {
    if (!g_comm) return 0;
    const int rc = g_rccl.CommDestroy(g_comm);
    g_comm = nullptr;
    g_world = 1;
    return check(rc, "kk_comm_destroy");
}

// In-place SUM all-reduce of one gradient bucket.
extern "C" int kk_comm_reduce_bucket(void *ptr, int64_t count, int dtype, void *comm_stream) {
    KK_REQUIRE(g_comm != nullptr, "kk_comm_reduce_bucket: no communicator (kk_comm_init)");
    KK_REQUIRE(ptr && count > 0, "kk_comm_reduce_bucket: empty bucket");
    int dt;
    if (int rc = dtype_of(dtype, "kk_comm_reduce_bucket", &dt)) return rc;
    return check(g_rccl.AllReduce(ptr, ptr, (size_t)count, dt, ncclSum, g_comm, (hipStream_t)comm_stream), "kk_comm_reduce_bucket");
}

// n in-place SUM all-reduces as ONE RCCL group (the ranges of a layer's weight matrices): one fused launch.
extern "C" int kk_comm_reduce_ranges(void *base, const int64_t *begin, const int64_t *end, int n, int dtype, void *comm_stream) {
    KK_REQUIRE(g_comm != nullptr, "kk_comm_reduce_ranges: no communicator (kk_comm_init)");
    KK_REQUIRE(base && begin && end && n >= 1, "kk_comm_reduce_ranges: bad arguments");
    int dt;
    if (int rc = dtype_of(dtype, "kk_comm_reduce_ranges", &dt)) return rc;
    const size_t esz = dtype == 0 ? 4 : 2;
    if (int rc = check(g_rccl.GroupStart(), "kk_comm_reduce_ranges")) return rc;
    int first = 0;
    for (int i = 0; i < n; ++i) {
        if (end[i] <= begin[i]) { first = first ? first : KK_EINVAL; continue; }
        char *p = static_cast<char *>(base) + (size_t)begin[i] * esz;
        const int rc = g_rccl.AllReduce(p, p, (size_t)(end[i] - begin[i]), dt, ncclSum, g_comm, (hipStream_t)comm_stream);
        if (rc && !first) first = rc;
    }
    const int rc_end = g_rccl.GroupEnd();
    if (first == KK_EINVAL) return kk_fail(KK_EINVAL, "kk_comm_reduce_ranges: empty range");
    return check(first ? first : rc_end, "kk_comm_reduce_ranges");
}

// The second collective of a data-parallel step with ragged shards: the loss normalisers.  The fp64 sums and
// valid-element counts of kk_losses_fwd (reference losses.py:40-46,82-105 normalise by counts of the batch the process sees)
// are SUM-reduced and the largest duration (the batch-shape heuristics of trainer.py:2218-2242) MAX-reduced, in place, as one
// RCCL group on the caller's stream — between kk_losses_fwd and kk_losses_finalize, inside the captured step.
extern "C" int kk_comm_loss_sync(double *acc, int n_acc, int64_t *max_dur, void *stream) {
    KK_REQUIRE(g_comm != nullptr, "kk_comm_loss_sync: no communicator (kk_comm_init)");
    KK_REQUIRE(acc && n_acc > 0 && max_dur, "kk_comm_loss_sync: bad arguments");
    if (int rc = check(g_rccl.GroupStart(), "kk_comm_loss_sync")) return rc;
    const int rc1 = g_rccl.AllReduce(acc, acc, (size_t)n_acc, ncclFloat64, ncclSum, g_comm, (hipStream_t)stream);
    const int rc2 = g_rccl.AllReduce(max_dur, max_dur, 1, ncclInt64, ncclMax, g_comm, (hipStream_t)stream);
    const int rc3 = g_rccl.GroupEnd();
    return check(rc1 ? rc1 : (rc2 ? rc2 : rc3), "kk_comm_loss_sync");
}

// The two halves of the ring all-reduce as separate calls (reduce-scatter -> [optimizer on the shard] -> all-gather).
extern "C" int kk_comm_reduce_scatter(const void *send, void *recv, int64_t recv_count, int dtype, void *comm_stream) {
    KK_REQUIRE(g_comm != nullptr, "kk_comm_reduce_scatter: no communicator (kk_comm_init)");
    KK_REQUIRE(send && recv && recv_count > 0, "kk_comm_reduce_scatter: bad arguments");
    int dt;
    if (int rc = dtype_of(dtype, "kk_comm_reduce_scatter", &dt)) return rc;
    return check(g_rccl.ReduceScatter(send, recv, (size_t)recv_count, dt, ncclSum, g_comm, (hipStream_t)comm_stream), "kk_comm_reduce_scatter");
}
extern "C" int kk_comm_all_gather(const void *send, void *recv, int64_t send_count, int dtype, void *comm_stream) {
    KK_REQUIRE(g_comm != nullptr, "kk_comm_all_gather: no communicator (kk_comm_init)");
    KK_REQUIRE(send && recv && send_count > 0, "kk_comm_all_gather: bad arguments");
    int dt;
    if (int rc = dtype_of(dtype, "kk_comm_all_gather", &dt)) return rc;
    return check(g_rccl.AllGather(send, recv, (size_t)send_count, dt, g_comm, (hipStream_t)comm_stream), "kk_comm_all_gather");
}

// y = scale * float(x): widens a bf16 gradient bucket back after the exchange (the narrowing is kk_cast_f32_bf16).
extern "C" int kk_cast_bf16_f32(const void *x, float *y, int64_t n, float scale, void *stream) {
    KK_REQUIRE(x && y && n > 0, "kk_cast_bf16_f32: bad arguments");
    int blocks = kk_cdiv(n, 1024);
    blocks = blocks > 8192 ? 8192 : blocks;
    hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, static_cast<const __bf16 *>(x), y, n, scale);
    KK_LAUNCH_CHECK("kk_cast_bf16_f32");
    return 0;
}
