// Instantiations of the table form of the LDS-DMA weight-gradient kernel (wgrad_ring_kernel.h: wgrad_ring_table_kernel) for one
// element type: every compiled-in configuration of wgrad_cfgs.h, the 8-wave 256 x 256 tile included (the late stages' and the
// head's gradients are the ones that run split-free in a shared grid).
#pragma once
#include "wgrad_ring_kernel.h"
#include "wgrad_cfgs.h"

template <typename T, int BO, int BI, int WO, int WI, int D, int KPS>
static int launch_wgrad_ring_table(const WgradArgs* tab, const int2* items, int n_items, hipStream_t s) {
    constexpr int lds = D * KPS * (BO * 2 + BI * 2);
    static_assert(lds <= 160 * 1024, "LDS budget");
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_ring_table_kernel<T, BO, BI, WO, WI, D, KPS>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) {
            lh_set_error("wgrad_ring_table: cannot raise dynamic LDS to %d bytes: %s", lds, hipGetErrorString(e));
            return LH_ERR_HIP;
        }
    }
    hipLaunchKernelGGL((wgrad_ring_table_kernel<T, BO, BI, WO, WI, D, KPS>), dim3(n_items), dim3(64 * WO * WI), lds, s, tab, items);
    LH_LAUNCH_CHECK("wgrad_ring_table launch");
    return LH_OK;
}

// returns 1 when the configuration is not compiled in
#define LH_WGRAD_TABLE_LAUNCHER(NAME, T)                                                                                         \
    int NAME(const WgradArgs* tab, const int2* items, int n_items, int bo, int bi, int kps, int depth, hipStream_t s) {            \
        LH_WGRAD_CFGS(LH_WGRAD_TABLE_CASE_##T)                                                                                     \
        return 1;                                                                                                                  \
    }
#define LH_WGRAD_TABLE_CASE_bf16(BO, BI, WO, WI, D, KPS) \
    if (bo == BO && bi == BI && depth == D && kps == KPS) return launch_wgrad_ring_table<bf16, BO, BI, WO, WI, D, KPS>(tab, items, n_items, s);
#define LH_WGRAD_TABLE_CASE_f16(BO, BI, WO, WI, D, KPS) \
    if (bo == BO && bi == BI && depth == D && kps == KPS) return launch_wgrad_ring_table<f16, BO, BI, WO, WI, D, KPS>(tab, items, n_items, s);
