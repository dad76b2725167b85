// bf16 instantiations of the LDS-DMA convolution kernel, configuration part "wide" (4-wave 256 x 256 tile; igemm_ring_inst.h).
#include "igemm_ring_cfgs.h"
#define LH_T bf16
#define LH_FN lh_ring_launch_bf16_wide
#define LH_LIST LH_RING_CFGS_WIDE
#define LH_DCODE LH_WIDE_DEPTH
#include "igemm_ring_inst.h"
