// How fast does a CU fill LDS from L2 / Infinity Cache / HBM by (a) the buffer-load-to-LDS DMA and (b) vector loads into registers +
// ds_write_b128, as a function of the loads in flight?  256 workgroups x 8 waves, W of them loading; every workgroup streams its own
// chunk `passes` times (chunk 64 KB: L2-resident; 512 KB: Infinity Cache; 4 MB: HBM).
//     hipcc --offload-arch=gfx950 -O3 -o stream_rate stream_rate.hip && ./stream_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define LDS_PTR(p) ((__attribute__((address_space(3))) void *)(p))
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int D>      // D DMA instructions (1 KB each) in flight per loading wave
__global__ __launch_bounds__(512) void dma_kernel(const char *src, size_t chunk, int passes, int W, unsigned *sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave >= W) return;
    const char *base = src + (size_t)blockIdx.x * chunk;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(base), 0, (int)chunk, 0x00020000);
    const uint32_t per_wave = (uint32_t)(chunk / W), n = per_wave / 1024;          // 1 KB per instruction
    char *dst = lds + wave * 16384;
    for (int p = 0; p < passes; ++p) {
        for (uint32_t i = 0; i < n; i += D) {
#pragma unroll
            for (int j = 0; j < D; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDS_PTR(dst + ((i + j) & 15) * 1024), 16, wave * per_wave + (i + j) * 1024 + lane * 16, 0, 0, 0);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D / 2) : "memory");          // half of them may stay in flight
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (sink && lane == 0) sink[blockIdx.x * 8 + wave] = *reinterpret_cast<unsigned *>(dst);
}

template <int U>      // 2 x U vector loads of 16 B per lane in flight per loading wave
__global__ __launch_bounds__(512) void reg_kernel(const char *src, size_t chunk, int passes, int W, unsigned *sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave >= W) return;
    const u32x4 *base = reinterpret_cast<const u32x4 *>(src + (size_t)blockIdx.x * chunk + (size_t)wave * (chunk / W)) + lane;
    const uint32_t n = (uint32_t)(chunk / W / 1024);
    u32x4 *dst = reinterpret_cast<u32x4 *>(lds + wave * 16384) + lane;
    u32x4 a[U], b[U];
    for (int p = 0; p < passes; ++p) {
#pragma unroll
        for (int j = 0; j < U; ++j) a[j] = __builtin_nontemporal_load(base + (size_t)j * 64);
        for (uint32_t i = 0; i < n; i += 2 * U) {
#pragma unroll
            for (int j = 0; j < U; ++j) b[j] = __builtin_nontemporal_load(base + (size_t)((i + U + j) % n) * 64);
#pragma unroll
            for (int j = 0; j < U; ++j) dst[((i + j) & 15) * 64] = a[j];
#pragma unroll
            for (int j = 0; j < U; ++j) a[j] = __builtin_nontemporal_load(base + (size_t)((i + 2 * U + j) % n) * 64);
#pragma unroll
            for (int j = 0; j < U; ++j) dst[((i + U + j) & 15) * 64] = b[j];
        }
    }
    __syncthreads();
    if (sink && lane == 0) sink[blockIdx.x * 8 + wave] = *reinterpret_cast<unsigned *>(lds + wave * 16384);
}

template <typename K>
static void run(const char *tag, K kern, const char *buf, size_t chunk, int W, unsigned *sink) {
    const int passes = (int)((size_t)(64u << 20) / chunk) > 0 ? (int)((size_t)(16u << 20) / chunk) + 1 : 1;       // ~16 MB per workgroup
    hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 16384);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(256), dim3(512), 8 * 16384, 0, buf, chunk, 1, W, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(512), 8 * 16384, 0, buf, chunk, passes, W, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)chunk * passes * 256, clk = ms * 1e-3 * 2.4e9;
    printf("%-44s chunk %5zu KB  W=%d  %7.2f TB/s chip  %6.1f B/clk/CU\n", tag, chunk >> 10, W, bytes / (ms * 1e-3) / 1e12, bytes / 256 / clk);
}

int main() {
    char *buf; unsigned *sink;
    const size_t total = (size_t)256 * (4u << 20);
    hipMalloc(&buf, total); hipMemset(buf, 1, total); hipMalloc(&sink, 256 * 8 * 4);
    for (size_t chunk : {(size_t)64 << 10, (size_t)512 << 10, (size_t)4 << 20}) {
        for (int W : {4, 8}) {
            run("LDS-DMA, 4 in flight per wave", dma_kernel<4>, buf, chunk, W, sink);
            run("LDS-DMA, 8 in flight per wave", dma_kernel<8>, buf, chunk, W, sink);
            run("LDS-DMA, 16 in flight per wave", dma_kernel<16>, buf, chunk, W, sink);
            run("registers + ds_write, 2 x 4 loads per wave", reg_kernel<4>, buf, chunk, W, sink);
            run("registers + ds_write, 2 x 8 loads per wave", reg_kernel<8>, buf, chunk, W, sink);
        }
    }
    return 0;
}
