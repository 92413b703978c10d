// What shader clock does the chip run at under (a) back-to-back tiny launches (the latency-bound regime of the train step),
// (b) one wave spinning alone, (c) every CU busy?  A wave counts shader cycles (s_memtime) against the 100 MHz wall clock
// (s_memrealtime).     hipcc --offload-arch=gfx950 -O3 -o sclk_probe sclk_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__global__ void spin_kernel(unsigned long long *out, long long cycles, int record) {
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    while ((long long)(clock64() - c0) < cycles) { __builtin_amdgcn_s_sleep(1); }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    if (record && threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; }
}
__global__ void burn_kernel(float *x, int iters) {       // keeps the vector ALUs of every CU busy
    float v = x[threadIdx.x];
    for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
    x[threadIdx.x] = v;
}

int main() {
    unsigned long long *d, h[2];
    float *x;
    hipMalloc(&d, 16);
    hipMalloc(&x, 4096);
    hipMemset(x, 0, 4096);
    auto report = [&](const char *tag) {
        hipDeviceSynchronize();
        hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("%-62s %8.0f MHz  (%llu shader cycles in %.1f us)\n", tag, (double)h[0] / ((double)h[1] / 100.0), h[0], (double)h[1] / 100.0);
    };
    // (b) one wave alone, after idling
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, 0, d, 2000000LL, 1);
    report("one wave, 2M cycles, chip otherwise idle (cold)");
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, 0, d, 20000000LL, 1);
    report("one wave, 20M cycles, chip otherwise idle");
    // (a) 20000 tiny dependent launches, then measure inside the last one
    for (int i = 0; i < 20000; ++i) hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, 0, d, 2000LL, 0);
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, 0, d, 200000LL, 1);
    report("after 20000 back-to-back one-wave launches of ~1 us");
    // (c) all CUs busy beside it
    hipStream_t s2;
    hipStreamCreate(&s2);
    hipLaunchKernelGGL(burn_kernel, dim3(256 * 8), dim3(256), 0, s2, x, 4000000);
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, 0, d, 20000000LL, 1);
    report("one wave, 20M cycles, 2048 workgroups of FMA chains beside it");
    hipDeviceSynchronize();
    return 0;
}
