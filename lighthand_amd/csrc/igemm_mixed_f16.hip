// f16 instantiations of the mixed multi-problem convolution launch (see igemm_mixed_inst.h).
#define LH_T f16
#define LH_FN lh_mixed_multi_launch_f16
#include "igemm_mixed_inst.h"
