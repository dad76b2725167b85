// fp16 instantiations of the table form of the LDS-DMA weight-gradient kernel (wgrad_table_inst.h).
#include "wgrad_table_inst.h"
LH_WGRAD_TABLE_LAUNCHER(lh_wgrad_ring_table_launch_f16, f16)
