// fp32 instantiations of the LDS-DMA convolution kernel (parity precision: v_mfma_f32_16x16x4_f32, exact fp32).
#define LH_T float
#define LH_FN lh_ring_launch_f32
#define LH_LIST LH_RING_CFGS_F32
#include "igemm_ring_inst.h"
