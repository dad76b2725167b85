// What do per-channel statistics accumulated by ATOMICS cost?  (the BatchNorm finalize / coefficient-fold launches
// could go if the convolution epilogue added its per-tile sums straight into [2][C] accumulators.)  G workgroups of 256 threads, each
// adds `per` values (one per thread, threads 0..per-1) to acc[rep][i], i = (slice * per + thread) % (2 C) where slice = blockIdx /
// tiles-per-slice models the output-channel tile a workgroup owns: all workgroups of one slice hit the same `per` words.
// Variants: device-scope fp64 add, device-scope int64 add, R replicas by blockIdx % R; body = a dependent FMA chain of `work` steps
// so the atomics have something to hide under.  Prints the kernel time with and without the atomics.
// build: hipcc -O3 --offload-arch=gfx950 tools/probes/stat_atomics.hip -o /tmp/stat_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int MODE>     // 0 none, 1 fp64, 2 int64
__global__ void k(double* accd, long long* acci, int words, int per, int slices, int reps, int work, float* sink) {
    float v = threadIdx.x * 1e-3f + blockIdx.x;
    for (int i = 0; i < work; ++i) v = fmaf(v, 1.0000001f, 1e-7f);
    const int slice = blockIdx.x % slices, rep = (blockIdx.x / slices) % reps;
    if ((int)threadIdx.x < per) {
        const int idx = rep * words + (slice * per + threadIdx.x) % words;
        if (MODE == 1) __hip_atomic_fetch_add(accd + idx, (double)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (MODE == 2) __hip_atomic_fetch_add(acci + idx, (long long)(v * 1024.f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (v == -1.f) sink[0] = v;
}

int main(int argc, char** argv) {
    double* accd; long long* acci; float* sink;
    hipMalloc(&accd, 1 << 22); hipMalloc(&acci, 1 << 22); hipMalloc(&sink, 64);
    hipMemset(accd, 0, 1 << 22); hipMemset(acci, 0, 1 << 22);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int work = argc > 1 ? atoi(argv[1]) : 2000;
    printf("# work %d FMA steps per thread; times in us per launch (mean of 50)\n", work);
    printf("# %6s %6s %6s %5s | %8s %8s %8s\n", "G", "C", "per", "reps", "none", "fp64", "int64");
    const int Gs[] = {256, 1024, 4096}, Cs[] = {64, 256, 1024}, Rs[] = {1, 8};
    for (int G : Gs) for (int C : Cs) for (int R : Rs) {
        const int words = 2 * C, per = words < 256 ? words : 256, slices = words / per;
        float t[3];
        for (int mode = 0; mode < 3; ++mode) {
            auto launch = [&]() {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(G), dim3(256), 0, 0, accd, acci, words, per, slices, R, work, sink);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(G), dim3(256), 0, 0, accd, acci, words, per, slices, R, work, sink);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(G), dim3(256), 0, 0, accd, acci, words, per, slices, R, work, sink);
            };
            for (int i = 0; i < 5; ++i) launch();
            hipEventRecord(e0, 0);
            for (int i = 0; i < 50; ++i) launch();
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            hipEventElapsedTime(&t[mode], e0, e1);
            t[mode] *= 1000.f / 50;
        }
        printf("  %6d %6d %6d %5d | %8.1f %8.1f %8.1f\n", G, C, per, R, t[0], t[1], t[2]);
    }
    return 0;
}
