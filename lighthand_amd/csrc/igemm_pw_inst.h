// Instantiates the persistent pointwise convolution kernel's configuration table for one element type.
// The including .hip file defines LH_T (element type), LH_FN and LH_OCC_FN (function names).  Returns 1 for an unknown configuration.
#include "igemm_pw_cfgs.h"
#include "igemm_pw_kernel.h"

int LH_FN(const IgemmArgs& a, const RingCfg& c, hipStream_t s) {
#define X(BM, KC, PT) \
    if (c.bm == BM && c.kb == KC && c.bp == 16 * PT) return launch_pw<LH_T, BM, KC, PT>(a, s);
    LH_PW_CFGS(X)
#undef X
    return 1;
}

int LH_OCC_FN(const RingCfg& c, int mode) {
#define X(BM, KC, PT) \
    if (c.bm == BM && c.kb == KC && c.bp == 16 * PT)  \
        return mode == 3 ? pw_occupancy<LH_T, BM, KC, PT, true, 2>() : mode == 2 ? pw_occupancy<LH_T, BM, KC, PT, true, 1>()  \
             : mode == 1 ? pw_occupancy<LH_T, BM, KC, PT, true>() : pw_occupancy<LH_T, BM, KC, PT, false>();
    LH_PW_CFGS(X)
#undef X
    return 2;
}
