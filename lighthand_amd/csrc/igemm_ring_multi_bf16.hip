// bf16 instantiations of the multi-problem LDS-DMA convolution kernel (see igemm_ring_multi_inst.h).
#define LH_T bf16
#define LH_FN lh_ring_multi_launch_bf16
#include "igemm_ring_multi_inst.h"
