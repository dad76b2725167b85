// bf16 instantiations of the direct 3x3 convolution kernel (conv3x3_direct_kernel.h): C_in = 32 and 64 per tap.
#include "conv3x3_direct_kernel.h"

// returns 1 for an unknown configuration
int lh_d3_launch_bf16(const IgemmArgs& a, const RingCfg& c, hipStream_t s) {
    if (c.kb == 64) return launch_d3<bf16, 64>(a, s);
    if (c.kb == 32) return launch_d3<bf16, 32>(a, s);
    return 1;
}
