// How well do LDS fragment reads and MFMAs overlap on one CU, and what does the per-stage barrier cost?  The stage loop of the tiled
// convolution kernel measures 900 cycles per stage (reads + MFMAs, no DMA) against an MFMA floor of 512 (profiles/r05_ingest_ladder.txt);
// this probe isolates the loop body: ONE workgroup per CU, NWAVE waves, every wave per "stage" = 2 K slices x [NRD ds_read_b128 + NMF
// v_mfma_f32_16x16x32_bf16]; wave tile 64 x 32 (NRD 6, NMF 8: the 8-wave form) or 64 x 64 (NRD 8, NMF 16: the 4-wave form).
//   V0 MFMAs only                      V1 reads in a burst, lgkmcnt ladder, then MFMAs (the kernel's form)
//   V2 reads of slice k + 1 spread between the MFMAs of slice k (two register sets)
//   V3 = V1 + s_barrier per stage      V4 = V2 + s_barrier per stage
//   V5 = V1 + barrier, waves 4..7 delayed by half a stage at the start (out of phase)
// Prints shader cycles per stage (s_memtime) and the ratio to the MFMA floor NMF x 2 slices x 16 cycles x waves per SIMD.
// build: hipcc -O3 --offload-arch=gfx950 tools/probes/lds_mfma_overlap.hip -o tools/probes/bin/lds_mfma_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NWAVE, int CT, int PT, int V>
__global__ __launch_bounds__(64 * NWAVE, 1) void k(unsigned long long* out, float* sink, int stages) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NR = CT + PT, KB = 128, GB = 16 * KB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 65536 / 4; i += 64 * NWAVE) ((unsigned*)smem)[i] = 0x3c003c00u + (i & 0xff);
    __syncthreads();
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    unsigned off[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int c = 4 * kk + (lane >> 4), r = lane & 15;
        off[kk] = lds_base + (wave % 4) * GB + r * KB + ((c ^ ((r / 2) & 7)) << 4);      // the kernel's swizzled fragment address
    }
    f32x4 acc[CT][PT];
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
        for (int j = 0; j < PT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    uint4 FA[NR], FB[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) FA[r] = FB[r] = uint4{0x3c003c00u + lane, 0x3c003c01u, 0x3c003c02u, 0x3c003c03u + r};
    auto rd = [&](uint4& dst, unsigned base, int r) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(base), "n"(0)); (void)r; };
    auto rdall = [&](uint4 (&F)[NR], int kk) {
#pragma unroll
        for (int r = 0; r < NR; ++r) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(F[r]) : "v"(off[kk]), "n"(0));
    };
    (void)rd;
    auto mm = [&](uint4 (&F)[NR]) {
#pragma unroll
        for (int i = 0; i < CT; ++i)
#pragma unroll
            for (int j = 0; j < PT; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, F[PT + i]), __builtin_bit_cast(bf16x8, F[j]), acc[i][j], 0, 0, 0);
    };
    auto mm_ladder = [&](uint4 (&F)[NR]) {             // MFMA group i starts when B fragments + A[i] have landed
#pragma unroll
        for (int i = 0; i < CT; ++i) {
            __builtin_amdgcn_sched_barrier(0);
            if (i == 0) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(CT - 1) : "memory");
            else if (i == 1) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(CT - 2) : "memory");
            else if (i == 2) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(CT > 2 ? CT - 3 : 0) : "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < PT; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, F[PT + i]), __builtin_bit_cast(bf16x8, F[j]), acc[i][j], 0, 0, 0);
        }
    };
    // MFMAs on F, the reads into G spread behind the MFMA groups
    auto mm_rd = [&](uint4 (&F)[NR], uint4 (&G)[NR], int kk) {
        constexpr int PERG = (NR + CT - 1) / CT;
#pragma unroll
        for (int i = 0; i < CT; ++i) {
#pragma unroll
            for (int j = 0; j < PT; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, F[PT + i]), __builtin_bit_cast(bf16x8, F[j]), acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = i * PERG; r < (i + 1) * PERG && r < NR; ++r) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(G[r]) : "v"(off[kk]), "n"(0));
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    constexpr bool BAR = V == 3 || V == 4 || V == 5;
    __syncthreads();
    if (V == 5 && wave >= NWAVE / 2) {                 // half a stage of head start for waves 0..3
        for (int i = 0; i < 64; ++i) asm volatile("s_nop 7");
    }
    const unsigned long long c0 = clock64();
    if (V == 0) {
        for (int s = 0; s < stages; ++s) { mm(FA); mm(FB); }
    } else if (V == 1 || V == 3 || V == 5) {
        for (int s = 0; s < stages; ++s) {
            if (BAR) __builtin_amdgcn_s_barrier();
            rdall(FA, 0); mm_ladder(FA);
            rdall(FB, 1); mm_ladder(FB);
        }
    } else {
        rdall(FA, 0);
        for (int s = 0; s < stages; ++s) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            mm_rd(FA, FB, 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (BAR) __builtin_amdgcn_s_barrier();
            mm_rd(FB, FA, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const unsigned long long c1 = clock64();
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
        for (int j = 0; j < PT; ++j) v += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (v == 12345.678f) sink[tid] = v;
    if (lane == 0) out[blockIdx.x * NWAVE + wave] = c1 - c0;
#endif
}

template <int NWAVE, int CT, int PT, int V>
static void run(const char* name, unsigned long long* d_out, float* sink) {
    const int stages = 400, nwg = 256;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<NWAVE, CT, PT, V>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    std::vector<unsigned long long> h(nwg * NWAVE);
    double best = 1e30;
    for (int rep = 0; rep < 4; ++rep) {
        hipLaunchKernelGGL((k<NWAVE, CT, PT, V>), dim3(nwg), dim3(64 * NWAVE), 65536, 0, d_out, sink, stages);
        hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
        std::vector<double> v(h.begin(), h.end());
        std::sort(v.begin(), v.end());
        best = std::min(best, v[v.size() / 2] / stages);
    }
    const double floor = CT * PT * 2 * 16.0 * (NWAVE / 4);
    printf("%-64s %2d waves, tile %2dx%2d: %7.1f cycles per stage  (MFMA floor %4.0f, x %.2f)\n", name, NWAVE, CT * 16, PT * 16, best, floor, best / floor);
}

int main() {
    unsigned long long* d_out; float* sink;
    hipMalloc(&d_out, 256 * 8 * 8); hipMalloc(&sink, 4096);
#define ALL(NW, CT, PT) \
    run<NW, CT, PT, 0>("V0 MFMAs only", d_out, sink); \
    run<NW, CT, PT, 1>("V1 reads in a burst, lgkmcnt ladder, MFMAs", d_out, sink); \
    run<NW, CT, PT, 2>("V2 reads of the next slice spread between this slice's MFMAs", d_out, sink); \
    run<NW, CT, PT, 3>("V3 = V1 + s_barrier per stage", d_out, sink); \
    run<NW, CT, PT, 4>("V4 = V2 + s_barrier per stage", d_out, sink); \
    run<NW, CT, PT, 5>("V5 = V3, waves 4..7 start half a stage late", d_out, sink);
    ALL(8, 4, 2)
    ALL(4, 4, 4)
    ALL(8, 4, 4)
    ALL(4, 4, 2)
    return 0;
}
