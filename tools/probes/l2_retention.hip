// Does an XCD's L2 keep READ-ONLY lines across a kernel boundary?  (round 6: could a kernel's epilogue warm the next GEMM's weights?)
// k_read streams a buffer with every workgroup (all XCDs); run it twice back to back as dependent launches and count, per launch,
// clock cycles per pass with s_memtime-free wall clock stamps: first launch = cold (buffer never touched), second = after a boundary.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/l2_retention tools/probes/l2_retention.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__global__ __launch_bounds__(256) void k_read(const uint4 *__restrict__ p, size_t n16, unsigned long long *stamps, uint4 *sink) {
    const unsigned long long t0 = wall_clock64();
    uint4 acc = {0, 0, 0, 0};
    // every workgroup reads the WHOLE buffer (as every XCD reads a whole weight matrix), 16 B per lane, coalesced
    for (size_t i = threadIdx.x; i < n16; i += 256) {
        const uint4 v = p[i];
        acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
    if (acc.x == 0x12345678u) sink[0] = acc;
    __syncthreads();
    if (threadIdx.x == 0) stamps[blockIdx.x] = wall_clock64() - t0;
}
__global__ void k_write(uint4 *p, size_t n16, unsigned v) {
    for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = uint4{v, v, v, v};
}
__global__ void k_flush(float *p, size_t n) {
    for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 1.f;
}

int main() {
    const size_t bytes = 1 << 20, n16 = bytes / 16;           // 1 MB: a w_q|w_k|w_v slice; fits every L2 (4 MB)
    uint4 *buf, *sink;
    unsigned long long *st;
    float *big;
    hipMalloc(&buf, bytes); hipMalloc(&sink, 64); hipMalloc(&st, 256 * 8 * 8);
    hipMalloc(&big, (size_t)1 << 30);
    std::vector<unsigned long long> h(256);
    auto report = [&](const char *what) {
        hipDeviceSynchronize();
        hipMemcpy(h.data(), st, 256 * 8, hipMemcpyDeviceToHost);
        double s = 0, mx = 0;
        for (auto v : h) { s += v; mx = v > mx ? v : mx; }
        printf("%-64s mean %7.2f us  max %7.2f us per workgroup (1 MB read by each of 256 workgroups)\n", what, s / 256 / 100.0, mx / 100.0);
    };
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_write, dim3(256), dim3(256), 0, 0, buf, n16, 7u + rep);      // written by other CUs (as the optimizer writes the weights)
        hipLaunchKernelGGL(k_flush, dim3(2048), dim3(256), 0, 0, big, (size_t)1 << 28);   // 1 GB through the Infinity Cache
        hipLaunchKernelGGL(k_read, dim3(256), dim3(256), 0, 0, buf, n16, st, sink);
        report("1st read after the buffer was written and the caches flushed:");
        hipLaunchKernelGGL(k_read, dim3(256), dim3(256), 0, 0, buf, n16, st, sink);
        report("2nd read, a dependent launch right behind the first:");
        hipLaunchKernelGGL(k_read, dim3(256), dim3(256), 0, 0, buf, n16, st, sink);
        report("3rd read:");
        hipLaunchKernelGGL(k_flush, dim3(2048), dim3(256), 0, 0, big, (size_t)(96 << 20) / 4);   // 96 MB: through every L2 (32 MB in all), well inside the 256 MB Infinity Cache
        hipLaunchKernelGGL(k_read, dim3(256), dim3(256), 0, 0, buf, n16, st, sink);
        report("read after 96 MB went through the L2s (Infinity Cache keeps the buffer):");
        hipLaunchKernelGGL(k_read, dim3(256), dim3(256), 0, 0, buf, n16, st, sink);
        report("and once more:");
    }
    return 0;
}
