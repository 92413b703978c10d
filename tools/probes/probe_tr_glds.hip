// Development probe (not part of the library): prints what ds_read_b64_tr_b16 and global_load_lds_dwordx4 do on gfx950.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/probe_tr_glds.hip -o tools/probes/probe_tr_glds && tools/probes/probe_tr_glds
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;

__global__ void probe_tr(const uint16_t *src, uint16_t *out, int pitch_elems, int mode) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = src[i];
    __syncthreads();
    const int lane = threadIdx.x;
    const int g = lane / 16, L = lane % 16;
    int row, col;
    if (mode == 0) { row = g * 4 + L / 4; col = 4 * (L % 4); }        // group reads a [4][16] block, lane L -> (L/4, 4*(L%4))
    else { row = g * 16 + L; col = 0; }                                // group reads a [16][4] block, lane L -> row L
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(lds + row * pitch_elems + col));
    for (int e = 0; e < 4; ++e) out[lane * 4 + e] = (uint16_t)v[e];
}

__global__ void probe_glds(const uint32_t *src, uint32_t *out) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[512];
    const int lane = threadIdx.x;
    const int p = (lane * 7) % 64;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + p * 4), (__attribute__((address_space(3))) void *)lds, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + 256 + p * 4), (__attribute__((address_space(3))) void *)(lds + 256), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int e = 0; e < 8; ++e) out[lane * 8 + e] = lds[lane * 8 + e];
}

int main() {
    uint16_t h[64 * 64], *d, *o, ho[256];
    for (int i = 0; i < 64 * 64; ++i) h[i] = (uint16_t)(((i / 64) << 8) | (i % 64));      // value = (row << 8) | col
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(ho));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(probe_tr, dim3(1), dim3(64), 0, 0, d, o, 64, mode);
        hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
        printf("ds_read_b64_tr_b16 mode %d (value = row<<8|col of the source element):\n", mode);
        for (int l = 0; l < 64; ++l) {
            if (l % 16 < 6 || l % 16 == 15) printf("  lane %2d: %04x %04x %04x %04x\n", l, ho[l * 4], ho[l * 4 + 1], ho[l * 4 + 2], ho[l * 4 + 3]);
        }
    }
    uint32_t hs[512], *ds, *dout, hout[512];
    for (int i = 0; i < 512; ++i) hs[i] = i;
    hipMalloc(&ds, sizeof(hs)); hipMalloc(&dout, sizeof(hout));
    hipMemcpy(ds, hs, sizeof(hs), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe_glds, dim3(1), dim3(64), 0, 0, ds, dout);
    hipMemcpy(hout, dout, sizeof(hout), hipMemcpyDeviceToHost);
    printf("global_load_lds x4: lds dword index -> source dword (lane l loaded source chunk (7l)%%64):\n");
    for (int l = 0; l < 6; ++l) printf("  lds[%3d..] = %u %u %u %u | lds[%3d..] = %u %u %u %u\n", l * 4, hout[l * 4], hout[l * 4 + 1], hout[l * 4 + 2], hout[l * 4 + 3],
                                       256 + l * 4, hout[256 + l * 4], hout[256 + l * 4 + 1], hout[256 + l * 4 + 2], hout[256 + l * 4 + 3]);
    return 0;
}
