// bf16 instantiations of the LDS-DMA convolution kernel, configuration part "ksplit" (K-split wave pairs; igemm_ring_inst.h).
#include "igemm_ring_cfgs.h"
#define LH_T bf16
#define LH_FN lh_ring_launch_bf16_ksplit
#define LH_LIST LH_RING_CFGS_KSPLIT
#define LH_DCODE LH_KSPLIT_DEPTH
#define LH_LAUNCH launch_ring_ksplit
#include "igemm_ring_inst.h"
