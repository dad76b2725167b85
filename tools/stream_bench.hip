// What a three-stream elementwise pass (read g, read x, write dx: the shape of the BatchNorm-backward apply pass) can reach
// on this GPU, as a function of how the loop is written.  Standalone: hipcc -O3 --offload-arch=gfx950 tools/stream_bench.hip
// usage: stream_bench [MB per tensor = 128] [iterations = 30]
// Buffers cycle through NSET sets so that nothing is served from the 256 MB Infinity Cache.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u32x4 combine(u32x4 g, u32x4 x) {      // a few VALU ops per element pair, like A*g + B*x + C on packed bf16
    u32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float g0 = __uint_as_float(g[i] << 16), g1 = __uint_as_float(g[i] & 0xffff0000u);
        const float x0 = __uint_as_float(x[i] << 16), x1 = __uint_as_float(x[i] & 0xffff0000u);
        const float r0 = 0.5f * g0 + 0.25f * x0 + 1.f, r1 = 0.5f * g1 + 0.25f * x1 + 1.f;
        r[i] = (__float_as_uint(r0) >> 16) | (__float_as_uint(r1) & 0xffff0000u);
    }
    return r;
}

template <int UN, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void pass3(const u32x4* __restrict__ g, const u32x4* __restrict__ x, u32x4* __restrict__ d, long total) {
    const long stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UN - 1) * stride < total; i += UN * stride) {
        u32x4 a[UN], b[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            a[u] = NTL ? __builtin_nontemporal_load(g + i + u * stride) : g[i + u * stride];
            b[u] = NTL ? __builtin_nontemporal_load(x + i + u * stride) : x[i + u * stride];
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const u32x4 r = combine(a[u], b[u]);
            if (NTS) __builtin_nontemporal_store(r, d + i + u * stride); else d[i + u * stride] = r;
        }
    }
    for (; i < total; i += stride) d[i] = combine(g[i], x[i]);
}

// contiguous strips: a workgroup owns one contiguous range (what the reduce pass does)
template <int UN>
__global__ __launch_bounds__(256) void pass3_strip(const u32x4* __restrict__ g, const u32x4* __restrict__ x, u32x4* __restrict__ d, long total) {
    const long per = (total + gridDim.x - 1) / gridDim.x;
    const long lo = (long)blockIdx.x * per, hi = lo + per < total ? lo + per : total;
    long i = lo + threadIdx.x;
    for (; i + (UN - 1) * 256 < hi; i += UN * 256) {
        u32x4 a[UN], b[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) { a[u] = g[i + u * 256]; b[u] = x[i + u * 256]; }
#pragma unroll
        for (int u = 0; u < UN; ++u) d[i + u * 256] = combine(a[u], b[u]);
    }
    for (; i < hi; i += 256) d[i] = combine(g[i], x[i]);
}

template <int UN>
__global__ __launch_bounds__(256) void read2(const u32x4* __restrict__ g, const u32x4* __restrict__ x, u32x4* __restrict__ d, long total) {
    const long stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    u32x4 acc = {0, 0, 0, 0};
    for (; i + (UN - 1) * stride < total; i += UN * stride) {
        u32x4 a[UN], b[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) { a[u] = g[i + u * stride]; b[u] = x[i + u * stride]; }
#pragma unroll
        for (int u = 0; u < UN; ++u) acc ^= combine(a[u], b[u]);
    }
    if (acc[0] == 0x12345678u) d[0] = acc;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char** argv) {
    const long mb = argc > 1 ? atol(argv[1]) : 128;
    const int iters = argc > 2 ? atoi(argv[2]) : 30;
    const long bytes = mb << 20, total = bytes / 16;
    constexpr int NSET = 4;
    u32x4 *g[NSET], *x[NSET], *d[NSET];
    for (int s = 0; s < NSET; ++s) {
        CK(hipMalloc(&g[s], bytes)); CK(hipMalloc(&x[s], bytes)); CK(hipMalloc(&d[s], bytes));
        CK(hipMemset(g[s], 0x3c, bytes)); CK(hipMemset(x[s], 0x3d, bytes));
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto launch, int streams) {
        for (int it = 0; it < 3; ++it) launch(it % NSET);
        hipEventRecord(e0, 0);
        for (int it = 0; it < iters; ++it) launch(it % NSET);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / iters;
        printf("%-44s %8.1f us  %6.2f TB/s\n", name, us, streams * (double)bytes / us / 1e6);
        return 0;
    };
    char nm[128];
#define RUN3(UN, NTL, NTS, GRID) snprintf(nm, sizeof nm, "grid-stride un=%d ntl=%d nts=%d grid=%d", UN, NTL, NTS, GRID); \
    run(nm, [&](int s) { hipLaunchKernelGGL((pass3<UN, NTL, NTS>), dim3(GRID), dim3(256), 0, 0, g[s], x[s], d[s], total); }, 3);
    RUN3(1, false, false, 2048) RUN3(1, false, false, 4096) RUN3(1, false, false, 8192) RUN3(1, false, false, 1024)
    RUN3(2, false, false, 2048) RUN3(4, false, false, 2048) RUN3(4, false, false, 1024) RUN3(4, false, false, 4096) RUN3(8, false, false, 1024)
    RUN3(1, true, false, 2048) RUN3(1, false, true, 2048) RUN3(1, true, true, 2048) RUN3(4, true, true, 2048) RUN3(4, false, true, 2048)
    {
        const int full = (int)((total + 255) / 256);
        snprintf(nm, sizeof nm, "one chunk per thread, grid=%d", full);
        run(nm, [&](int s) { hipLaunchKernelGGL((pass3<1, false, false>), dim3(full), dim3(256), 0, 0, g[s], x[s], d[s], total); }, 3);
    }
#define RUNS(UN, GRID) snprintf(nm, sizeof nm, "contiguous strips un=%d grid=%d", UN, GRID); \
    run(nm, [&](int s) { hipLaunchKernelGGL((pass3_strip<UN>), dim3(GRID), dim3(256), 0, 0, g[s], x[s], d[s], total); }, 3);
    RUNS(1, 2048) RUNS(4, 2048) RUNS(4, 8192) RUNS(4, 32768)
#define RUNR(UN, GRID) snprintf(nm, sizeof nm, "two reads only un=%d grid=%d", UN, GRID); \
    run(nm, [&](int s) { hipLaunchKernelGGL((read2<UN>), dim3(GRID), dim3(256), 0, 0, g[s], x[s], d[s], total); }, 2);
    RUNR(1, 2048) RUNR(4, 2048) RUNR(4, 4096) RUNR(8, 2048)
    run("hipMemcpyAsync d2d (1 read + 1 write)", [&](int s) { hipMemcpyAsync(d[s], g[s], bytes, hipMemcpyDeviceToDevice, 0); }, 2);
    return 0;
}
