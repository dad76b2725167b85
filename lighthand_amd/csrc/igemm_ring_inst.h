// Instantiates one part of the LDS-DMA convolution kernel's configuration table for one element type.
// The including .hip file defines LH_T (element type), LH_FN (function name) and LH_LIST (X-macro list of
// igemm_ring_cfgs.h).  Returns 1 when the configuration is not in this part (the dispatcher tries the next one).
#include "igemm_ring_cfgs.h"
#include "igemm_ring_kernel.h"

#ifndef LH_DCODE
#define LH_DCODE 0          // offset of the ring depth in RingCfg.depth (LH_DENSE_DEPTH / LH_KSPLIT_DEPTH for the dense-wave / K-split configurations)
#endif

#ifndef LH_LAUNCH
#define LH_LAUNCH launch_ring   // launch_ring_ksplit for the K-split wave-pair forms
#endif

int LH_FN(const IgemmArgs& a, const RingCfg& c, hipStream_t s) {
#define X(BM, BP, WC, WP, D, KB) \
    if (c.bm == BM && c.bp == BP && c.depth == D + LH_DCODE && c.kb == KB) return LH_LAUNCH<LH_T, BM, BP, WC, WP, D, KB>(a, s);
    LH_LIST(X)
#undef X
    return 1;
}
