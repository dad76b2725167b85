// fp16 instantiations of the persistent pointwise convolution kernel (see igemm_pw_inst.h).
#define LH_T f16
#define LH_FN lh_pw_launch_f16
#define LH_OCC_FN lh_pw_occ_f16
#include "igemm_pw_inst.h"
