// f16 instantiations of the multi-problem LDS-DMA convolution kernel (see igemm_ring_multi_inst.h).
#define LH_T f16
#define LH_FN lh_ring_multi_launch_f16
#include "igemm_ring_multi_inst.h"
