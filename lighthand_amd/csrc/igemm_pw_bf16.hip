// bf16 instantiations of the persistent pointwise convolution kernel (see igemm_pw_inst.h).
#define LH_T bf16
#define LH_FN lh_pw_launch_bf16
#define LH_OCC_FN lh_pw_occ_bf16
#include "igemm_pw_inst.h"
