// Does an out-of-range lane of `buffer_load_dwordx4 ... offen lds` write ZEROS to LDS (or leave the old bytes)?  The buffer form of
// the convolution kernels' operand path relies on it (masked taps get an out-of-range offset instead of a zero-page address).
// Also: is the SGPR offset part of the range check?  build: hipcc -O3 --offload-arch=gfx950 tools/probes/buffer_lds_oob.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void* lds_p;
__global__ void k(const unsigned char* p, unsigned nbytes, int soff, unsigned* out) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ __attribute__((aligned(16))) unsigned smem[512];
    for (int i = threadIdx.x; i < 512; i += 64) smem[i] = 0xdeadbeefu;
    __syncthreads();
    auto r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, nbytes, 0x00020000);
    unsigned vo = threadIdx.x * 16;
    if (threadIdx.x & 1) vo = 0x80000000u;          // odd lanes: out of range
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_p)smem, 16, vo, soff, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = smem[i];
#endif
}
int main() {
    unsigned char* d; unsigned* o; unsigned h[256]; unsigned src[1024];
    for (int i = 0; i < 1024; ++i) src[i] = 0x1000 + i;
    hipMalloc(&d, 4096); hipMalloc(&o, 1024); hipMemcpy(d, src, 4096, hipMemcpyHostToDevice);
    for (int pass = 0; pass < 3; ++pass) {
        // pass 0: everything in range but the odd lanes; pass 1: records = 1024 and soff = 512: are lanes with vo + soff >= 1024 dropped?
        // pass 2: records = 1024, soff = 2048 (> records): is the SGPR offset range-checked at all?
        const unsigned rec = pass == 0 ? 4096 : 1024; const int soff = pass == 0 ? 0 : pass == 1 ? 512 : 2048;
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, rec, soff, o);
        hipMemcpy(h, o, 1024, hipMemcpyDeviceToHost);
        printf("pass %d (records %u, soffset %d): lane dword0 =", pass, rec, soff);
        for (int l = 0; l < 64; l += 1) if (l < 6 || l > 58 || (l >= 30 && l <= 34)) printf(" [%d]%x", l, h[l * 4]);
        printf("\n");
    }
    return 0;
}
