#include <hip/hip_runtime.h>
#include <stdint.h>
__global__ void k(const uint32_t *src, uint32_t *out, uint32_t bytes) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[512];
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(src), 0, bytes, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)lds, 16, threadIdx.x * 16, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    out[threadIdx.x] = lds[threadIdx.x * 4];
}
