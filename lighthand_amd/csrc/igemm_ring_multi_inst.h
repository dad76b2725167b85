// Instantiates the multi-problem form of the LDS-DMA convolution kernel (igemm_ring_multi_kernel) for one element type:
// the 4-wave tiles (parts "mid" and "small" of igemm_ring_cfgs.h) -- the layers that are batched are the small ones.
// The including .hip file defines LH_T (element type) and LH_FN (function name).  Returns 1 for an unknown configuration.
#include "igemm_ring_cfgs.h"
#include "igemm_ring_kernel.h"

int LH_FN(const LhMulti<IgemmArgs>& m, const RingCfg& c, hipStream_t s) {
#define X(BM, BP, WC, WP, D, KB) \
    if (c.bm == BM && c.bp == BP && c.depth == D && c.kb == KB) return launch_ring_multi<LH_T, BM, BP, WC, WP, D, KB>(m, s);
    LH_RING_CFGS_MID(X)
    LH_RING_CFGS_SMALL(X)
#undef X
    return 1;
}
