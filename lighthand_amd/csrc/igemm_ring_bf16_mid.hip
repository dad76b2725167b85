// bf16 instantiations of the LDS-DMA convolution kernel, configuration part "mid" (see igemm_ring_inst.h).
#define LH_T bf16
#define LH_FN lh_ring_launch_bf16_mid
#define LH_LIST LH_RING_CFGS_MID
#include "igemm_ring_inst.h"
