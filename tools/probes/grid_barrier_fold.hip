// What would a BatchNorm finalize INSIDE the convolution launch cost?  (conv epilogue -> per-tile sums -> grid barrier -> every
// workgroup folds the slab rows of its channel tile -> scale / shift -> BN + ReLU applied to the tile still in registers.)
// G workgroups of 512 threads (one or two per CU), CB channel tiles of 128 channels; each workgroup (pixel tile pb, channel tile cb):
//   1. dirties a 32 KB "output tile" with plain stores (what the epilogue has just done),
//   2. writes its slab row (2 x 128 floats) with sc1 stores,  s_waitcnt vmcnt(0), workgroup barrier,
//   3. one lane: agent-scope atomic add on the arrival counter, then polls it (sc1 loads) until the whole grid has arrived,
//   4. workgroup barrier; folds rows x 2 x 128 floats of its channel tile with sc1 loads (16 row lanes per channel, fp64),
//   5. checks the totals against the closed form.
// Prints per-phase times (s_memrealtime, 100 MHz): median / max over workgroups, and the number of wrong totals (must be 0).
// build: hipcc -O3 --offload-arch=gfx950 tools/probes/grid_barrier_fold.hip -o tools/probes/bin/grid_barrier_fold
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

__global__ __launch_bounds__(512, 2) void k(float* slab, unsigned* counter, uint4* tile, unsigned long long* stamps, int* bad, int rows, int cb_n, int work) {
    __shared__ double red[2][16][129];
    const int tid = threadIdx.x, bid = blockIdx.x, nblk = gridDim.x;
    const int pb = bid / cb_n, cb = bid % cb_n, c = cb_n * 128;
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    float v = tid * 1e-3f;
    for (int i = 0; i < work; ++i) v = fmaf(v, 1.0000001f, 1e-7f);          // the K loop
    if (tid == 0) t0 = wall_clock64();
    for (int i = tid; i < 2048; i += 512) tile[(long)bid * 2048 + i] = uint4{(unsigned)i, (unsigned)bid, 0u, (unsigned)(v == -1.f)};
    if (tid < 256) {
        const int which = tid / 128, col = tid % 128;
        const float val = (float)(pb + 1) * (which ? 0.5f : 1.0f) + (float)(cb * 128 + col) * 0.001f;
        __hip_atomic_store(slab + ((long)pb * 2 + which) * c + cb * 128 + col, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        t1 = wall_clock64();
        const unsigned old = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = (old / (unsigned)nblk + 1u) * (unsigned)nblk;
        int polls = 0;
        while ((int)(__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
            __builtin_amdgcn_s_sleep(2);
            if (++polls > (1 << 22)) { atomicAdd(bad, 1000000); break; }          // a grid that is not co-resident must not hang the box
        }
        t2 = wall_clock64();
    }
    __syncthreads();
    // fold: 128 channels x 16 row lanes; thread = (rl = tid >> 5, 4 channels (tid & 31) + 32 k)
    const int rl = tid >> 5;
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) {
        const int col = (tid & 31) + 32 * kq, ch = cb * 128 + col;
        double a = 0.0, b = 0.0;
        for (int r = rl; r < rows; r += 16) {
            a += (double)__hip_atomic_load(slab + ((long)r * 2) * c + ch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            b += (double)__hip_atomic_load(slab + ((long)r * 2 + 1) * c + ch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        red[0][rl][col] = a; red[1][rl][col] = b;
    }
    __syncthreads();
    if (tid < 128) {
        double s0 = 0.0, s1 = 0.0;
        for (int i = 0; i < 16; ++i) { s0 += red[0][i][tid]; s1 += red[1][i][tid]; }
        const double e0 = (double)rows * (rows + 1) / 2 + (double)rows * (double)((float)(cb * 128 + tid) * 0.001f);
        const double e1 = (double)rows * (rows + 1) / 4 + (double)rows * (double)((float)(cb * 128 + tid) * 0.001f);
        if (fabs(s0 - e0) > 1e-2 * rows || fabs(s1 - e1) > 1e-2 * rows) atomicAdd(bad, 1);
    }
    __syncthreads();
    if (tid == 0) {
        t3 = wall_clock64();
        stamps[bid * 4] = t0; stamps[bid * 4 + 1] = t1; stamps[bid * 4 + 2] = t2; stamps[bid * 4 + 3] = t3;
    }
}

__global__ void clear_slab(float* slab, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) slab[i] = -1e30f;
}

int main() {
    float* slab; unsigned* counter; uint4* tile; unsigned long long* stamps; int* bad;
    hipMalloc(&slab, 1 << 24); hipMalloc(&counter, 256); hipMalloc(&tile, (size_t)1024 * 32768); hipMalloc(&stamps, 1024 * 32); hipMalloc(&bad, 4);
    hipMemset(counter, 0, 256); hipMemset(bad, 0, 4);
    printf("# grid  rows x channel tiles | us: store->arrive   barrier (arrive->released)   fold   total epilogue extra   | wall us | wrong totals\n");
    const int cfgs[][2] = {{128, 2}, {128, 1}, {32, 8}, {32, 4}, {256, 2}, {512, 1}, {64, 4}};
    for (auto& cf : cfgs) {
        const int rows = cf[0], cbn = cf[1], G = rows * cbn;
        std::vector<unsigned long long> h(G * 4);
        std::vector<double> a, b, c, tot, wall;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 12; ++rep) {
            hipLaunchKernelGGL(clear_slab, dim3(256), dim3(256), 0, 0, slab, (long)rows * 2 * cbn * 128);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k, dim3(G), dim3(512), 0, 0, slab, counter, tile, stamps, bad, rows, cbn, 3000);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h.data(), stamps, G * 32, hipMemcpyDeviceToHost);
            if (rep < 2) continue;
            std::vector<double> x, y, z, w;
            for (int i = 0; i < G; ++i) {
                x.push_back((h[i * 4 + 1] - h[i * 4]) * 0.01); y.push_back((h[i * 4 + 2] - h[i * 4 + 1]) * 0.01);
                z.push_back((h[i * 4 + 3] - h[i * 4 + 2]) * 0.01); w.push_back((h[i * 4 + 3] - h[i * 4]) * 0.01);
            }
            auto med = [](std::vector<double> q) { std::sort(q.begin(), q.end()); return q[q.size() / 2]; };
            a.push_back(med(x)); b.push_back(med(y)); c.push_back(med(z)); tot.push_back(*std::max_element(w.begin(), w.end())); wall.push_back(ms * 1e3);
        }
        auto med = [](std::vector<double> q) { std::sort(q.begin(), q.end()); return q[q.size() / 2]; };
        int hb; hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
        printf("  %4d  %4d x %d              | %8.2f %20.2f %18.2f %12.2f (max over WGs) | %7.1f | %d\n", G, rows, cbn, med(a), med(b), med(c), med(tot), med(wall), hb);
    }
    return 0;
}
