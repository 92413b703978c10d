// How fast are no-return fp32 atomic adds to HBM-resident memory when 4 (or 8) workgroups on different CUs add into the same
// 128 KB region — the dQ accumulation pattern of a fused (5-matmul) attention backward?     hipcc --offload-arch=gfx950 -O3 -o atomic_rate atomic_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__global__ __launch_bounds__(512) void add_kernel(float *dst, int sharers, int region_floats, int reps, int mode) {
    // workgroup g adds into region g / sharers; every lane adds `reps` passes over the region, 64 consecutive floats per wave instruction
    float *r = dst + (size_t)(blockIdx.x / sharers) * region_floats;
    const int phase = (blockIdx.x % sharers) * (region_floats / sharers);      // sharers start at different offsets
    for (int p = 0; p < reps; ++p)
        for (int i = threadIdx.x; i < region_floats; i += 512) {
            const int j = (i + phase) % region_floats;
            if (mode == 0) atomicAdd(r + j, 1.0f);                               // (-munsafe-fp-atomics: global_atomic_add_f32, no return)
            else if (mode == 1) __hip_atomic_fetch_add(r + j, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else r[j] += 1.0f;                                                   // plain read-modify-write (wrong when shared): the bandwidth yardstick
        }
}

int main() {
    const int region = 32768;                       // 128 KB of floats: one (batch, head)'s dQ at S = 512
    for (int sharers : {1, 4, 8}) {
        const int wgs = 256 * (sharers == 8 ? 2 : 1);
        const int regions = wgs / sharers;
        float *d;
        hipMalloc(&d, (size_t)regions * region * 4);
        hipMemset(d, 0, (size_t)regions * region * 4);
        for (int mode : {0, 1, 2}) {
            hipEvent_t a, b;
            hipEventCreate(&a); hipEventCreate(&b);
            add_kernel<<<wgs, 512>>>(d, sharers, region, 1, mode);
            hipDeviceSynchronize();
            hipEventRecord(a);
            for (int it = 0; it < 20; ++it) add_kernel<<<wgs, 512>>>(d, sharers, region, 1, mode);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            const double bytes = (double)wgs * region * 4;
            printf("sharers %d mode %d (%s): %.2f us per launch, %.1f MB of adds -> %.2f TB/s of payload\n", sharers, mode,
                   mode == 0 ? "agent-scope atomic" : mode == 1 ? "workgroup-scope atomic" : "plain rmw", ms / 20 * 1e3, bytes / 1e6, bytes / (ms / 20 * 1e-3) / 1e12);
        }
        std::vector<float> h(region);
        hipMemcpy(h.data(), d, region * 4, hipMemcpyDeviceToHost);
        printf("  check: region 0 element 5 = %.0f\n", h[5]);
        hipFree(d);
    }
    return 0;
}
