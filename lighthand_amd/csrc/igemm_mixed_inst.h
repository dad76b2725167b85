// Instantiates the mixed multi-problem convolution launch (igemm_mixed_kernel.h) for one element type: the ring depths / stage
// sizes of the 64 x 128 tile.  The including .hip file defines LH_T and LH_FN.  Returns 1 for an unknown configuration.
#include "igemm_mixed_kernel.h"

int LH_FN(const LhMulti<IgemmArgs>& m, const MixedKinds& kt, const RingCfg& c, bool stats, hipStream_t s) {
#define X(D, KB) if (c.depth == D && c.kb == KB) return launch_mixed<LH_T, D, KB>(m, kt, stats, s);
    X(2, 64) X(4, 64) X(2, 128) X(3, 128) X(4, 128)
#undef X
    return 1;
}
