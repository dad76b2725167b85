// Mixed multi-problem convolution launch: up to LH_MULTI_MAX independent 3x3 / 1x1 convolutions as ONE grid in which every
// problem runs the kernel BODY that suits it -- the direct 3x3 body (conv3x3_direct_kernel.h: C_in = 32 or 64 per tap, input
// patch and weights resident in LDS, 512 threads) or the tiled LDS-DMA ring body (igemm_ring_kernel.h, 64 x 128 tile, 256
// threads: waves 4-7 of such a workgroup retire at once; s_barrier only counts the waves that are left).
//
// Why: the same layer position of HRNet's 2-4 parallel branches (pose_hrnet.py:139-185) is one launch.  With one tiled
// configuration for all of them the 32-channel branch (K run = 64 bytes) forces 64-byte ring stages on the 256-channel branch,
// whose tile then walks 36 dependent stages, and the two high-resolution branches gather every input row nine times on a
// half-empty tile.  Here the high-resolution branches take the direct body and the deep ones a ring with 128-byte stages.
// Every body keeps the K order of the tiled kernel: results are bit-identical to the problems launched one by one.
#pragma once
#include "conv3x3_direct_kernel.h"

struct MixedKinds { int k[LH_MULTI_MAX]; };       // per problem: 32 / 64 = direct body with that many channels per tap, 0 = ring body

template <typename T, int D, int KB, bool STATS>
__global__ __launch_bounds__(512, 2) void igemm_mixed_multi_kernel(const LhMulti<IgemmArgs> m, const MixedKinds kt) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int bid, nblk;
    const int i = lh_multi_pick(m, bid, nblk);
    const int kind = kt.k[i];                     // wave-uniform
    if (kind == 64) { conv3x3_direct_body<T, 64, STATS>(m.a[i], smem, bid); return; }
    if (kind == 32) { conv3x3_direct_body<T, 32, STATS>(m.a[i], smem, bid); return; }
    if (threadIdx.x >= 256) return;
    igemm_ring_body<T, 64, 128, 1, 4, D, KB>(m.a[i], smem, bid, nblk);
}

template <typename T, int D, int KB>
static int launch_mixed(const LhMulti<IgemmArgs>& m, const MixedKinds& kt, bool stats, hipStream_t s) {
    constexpr int ring = D * (64 + 128) * KB, epi = lh_epi_lds_bytes<T, 64, 128, false>();
    int lds = ring > epi ? ring : epi;
    for (int i = 0; i < m.n; ++i)
        if (kt.k[i]) { const int l = lh_d3_lds_bytes(kt.k[i]); lds = l > lds ? l : lds; }
    const void* fn = stats ? reinterpret_cast<const void*>(&igemm_mixed_multi_kernel<T, D, KB, true>)
                           : reinterpret_cast<const void*>(&igemm_mixed_multi_kernel<T, D, KB, false>);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) {
            lh_set_error("igemm_mixed_multi: cannot raise dynamic LDS to %d bytes: %s", lds, hipGetErrorString(e));
            return LH_ERR_HIP;
        }
    }
    if (stats) hipLaunchKernelGGL((igemm_mixed_multi_kernel<T, D, KB, true>), dim3(m.first[m.n]), dim3(512), lds, s, m, kt);
    else hipLaunchKernelGGL((igemm_mixed_multi_kernel<T, D, KB, false>), dim3(m.first[m.n]), dim3(512), lds, s, m, kt);
    LH_LAUNCH_CHECK("igemm_mixed_multi launch");
    return LH_OK;
}
