// f16 instantiations of the LDS-DMA convolution kernel, configuration part "dense" (8-wave forms of the 4-wave tiles; igemm_ring_inst.h).
#include "igemm_ring_cfgs.h"
#define LH_T f16
#define LH_FN lh_ring_launch_f16_dense
#define LH_LIST LH_RING_CFGS_DENSE
#define LH_DCODE LH_DENSE_DEPTH
#include "igemm_ring_inst.h"
