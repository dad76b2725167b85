// How fast can a CU move operand rows L2 -> LDS, by method?  (every MFMA kernel of this library is bound by
// this rate.)  Each workgroup re-reads its own 64 KiB region (L2-resident: 256 workgroups x 64 KiB = 2 MiB per XCD) as
// "stages" of 1 KiB-per-wave-instruction row groups, the pattern of igemm_ring_kernel.h, and does nothing else.
//   A  LDS-DMA (global_load_lds_dwordx4), NI instructions per wave and stage, s_waitcnt vmcnt(0) + s_barrier per stage
//   A2 the same with one stage kept in flight (vmcnt(NI))
//   B  global_load_dwordx4 into registers + ds_write_b128, next stage's loads issued before this stage's writes
// usage: ingest_bench [iterations = 2000]      build: hipcc -O3 --offload-arch=gfx950 tools/ingest_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_p;
typedef const __attribute__((address_space(1))) void* gbl_p;

template <int NWAVE, int NI, int INFLIGHT>
__global__ __launch_bounds__(64 * NWAVE) void dma_kernel(const unsigned char* __restrict__ src, unsigned* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned char* base = src + (long)blockIdx.x * 65536;
    constexpr int STAGE = NWAVE * NI * 1024;
    constexpr int NSTAGE = 65536 / STAGE;          // stages in the region
    int slot = 0;
    for (int it = 0; it < iters; ++it) {
        const int st = it % NSTAGE;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const unsigned char* g = base + st * STAGE + (NWAVE * j + wave) * 1024 + lane * 16;
            __builtin_amdgcn_global_load_lds((gbl_p)g, (lds_p)(smem + slot * STAGE + (NWAVE * j + wave) * 1024), 16, 0, 0);
        }
        if (INFLIGHT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        slot ^= 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = *reinterpret_cast<unsigned*>(smem + 64);
}

template <int NWAVE, int NI>
__global__ __launch_bounds__(64 * NWAVE) void reg_kernel(const unsigned char* __restrict__ src, unsigned* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned char* base = src + (long)blockIdx.x * 65536;
    constexpr int STAGE = NWAVE * NI * 1024;
    constexpr int NSTAGE = 65536 / STAGE;
    u32x4 r[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) r[j] = *reinterpret_cast<const u32x4*>(base + (NWAVE * j + wave) * 1024 + lane * 16);
    int slot = 0;
    for (int it = 0; it < iters; ++it) {
        const int st = (it + 1) % NSTAGE;
        u32x4 n[NI];
#pragma unroll
        for (int j = 0; j < NI; ++j) n[j] = *reinterpret_cast<const u32x4*>(base + st * STAGE + (NWAVE * j + wave) * 1024 + lane * 16);
#pragma unroll
        for (int j = 0; j < NI; ++j) *reinterpret_cast<u32x4*>(smem + slot * STAGE + (NWAVE * j + wave) * 1024 + lane * 16) = r[j];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int j = 0; j < NI; ++j) r[j] = n[j];
        slot ^= 1;
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = *reinterpret_cast<unsigned*>(smem + 64) + r[0][0];
}

// C: LDS-DMA of 128-byte row slices at a ROW PITCH of S bytes (what a convolution's pixel rows look like: pitch = 2 C bytes, a
// stage takes bytes [k0, k0 + 128) of 128 rows; the K loop walks k0 over the row).  8 regions of 128 rows shared by the
// workgroups of an XCD (L2-resident up to S = 2 KiB).  4 waves, 4 instructions per wave and stage (16 KiB stages).
template <int INFLIGHT>
__global__ __launch_bounds__(256) void dma_pitch_kernel(const unsigned char* __restrict__ src, unsigned* out, int iters, int pitch) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned char* base = src + (long)((blockIdx.x / 8) % 8) * 128 * pitch;
    const int slices = pitch / 128;
    int slot = 0, k0 = (blockIdx.x * 3) % slices;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = (4 * j + wave) * 8 + (lane >> 3);
            const unsigned char* g = base + (long)row * pitch + k0 * 128 + (lane & 7) * 16;
            __builtin_amdgcn_global_load_lds((gbl_p)g, (lds_p)(smem + slot * 16384 + (4 * j + wave) * 1024), 16, 0, 0);
        }
        if (INFLIGHT) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        slot ^= 1;
        if (++k0 == slices) k0 = 0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = *reinterpret_cast<unsigned*>(smem + 64);
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    unsigned char* src; unsigned* out;
    const int maxwg = 1024;
    CK(hipMalloc(&src, (size_t)maxwg * 65536)); CK(hipMemset(src, 1, (size_t)maxwg * 65536));
    CK(hipMalloc(&out, maxwg * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto launch, int wgs, int stage_bytes) {
        launch(wgs, 50);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        launch(wgs, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        const double bytes = (double)wgs * iters * stage_bytes;
        printf("%-58s %4d wg  %7.1f us  %6.2f TB/s chip  %6.1f GB/s per CU\n", name, wgs, ms * 1e3, bytes / ms / 1e9, bytes / ms / 1e6 / 256);
    };
#define RUN_DMA(NW, NI, INF, WGS) run("LDS-DMA " #NW " waves, " #NI " instr/wave/stage, in flight " #INF, [&](int g, int it) { \
        hipFuncSetAttribute(reinterpret_cast<const void*>(&dma_kernel<NW, NI, INF>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * NW * NI * 1024); \
        hipLaunchKernelGGL((dma_kernel<NW, NI, INF>), dim3(g), dim3(64 * NW), 2 * NW * NI * 1024, 0, src, out, it); }, WGS, NW * NI * 1024);
#define RUN_REG(NW, NI, WGS) run("registers + ds_write " #NW " waves, " #NI " loads/wave/stage", [&](int g, int it) { \
        hipFuncSetAttribute(reinterpret_cast<const void*>(&reg_kernel<NW, NI>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * NW * NI * 1024); \
        hipLaunchKernelGGL((reg_kernel<NW, NI>), dim3(g), dim3(64 * NW), 2 * NW * NI * 1024, 0, src, out, it); }, WGS, NW * NI * 1024);
    RUN_DMA(4, 8, 0, 256) RUN_DMA(4, 8, 1, 256) RUN_DMA(4, 8, 1, 512) RUN_DMA(4, 4, 1, 512) RUN_DMA(4, 4, 1, 1024)
    RUN_DMA(8, 8, 1, 256) RUN_DMA(8, 4, 1, 256) RUN_DMA(8, 4, 1, 512) RUN_DMA(4, 16, 1, 256)
    RUN_REG(4, 8, 256) RUN_REG(4, 8, 512) RUN_REG(4, 4, 512) RUN_REG(4, 4, 1024) RUN_REG(8, 8, 256) RUN_REG(8, 4, 256) RUN_REG(8, 4, 512) RUN_REG(4, 16, 256)
    for (int pitch : {128, 256, 384, 512, 640, 1024, 1152, 2048, 2176, 4096, 4224}) {
        char nm[96];
        snprintf(nm, sizeof nm, "LDS-DMA 128-byte slices, row pitch %d bytes", pitch);
        for (int wgs : {256, 512})
            run(nm, [&](int g, int it) { hipLaunchKernelGGL((dma_pitch_kernel<1>), dim3(g), dim3(256), 32768, 0, src, out, it, pitch); }, wgs, 16384);
    }
    return 0;
}
