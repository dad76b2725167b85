// f16 instantiations of the LDS-DMA weight-gradient kernel (wgrad_ring_kernel.h, wgrad_cfgs.h).
#include "wgrad_ring_kernel.h"
#include "wgrad_cfgs.h"

struct WgradPlan {
    int bo, bi, kps, depth, nsplit, sps;
};

template <typename T, int BO, int BI, int WO, int WI, int D, int KPS>
static int launch_wgrad_ring(const WgradArgs& a, hipStream_t s) {
    constexpr int lds = D * KPS * (BO * 2 + BI * 2);
    static_assert(lds <= 160 * 1024, "LDS budget");
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_ring_kernel<T, BO, BI, WO, WI, D, KPS>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) {
            lh_set_error("wgrad_ring: cannot raise dynamic LDS to %d bytes: %s", lds, hipGetErrorString(e));
            return LH_ERR_HIP;
        }
    }
    dim3 grid(a.tiles * a.ntaps * a.nsplit);
    hipLaunchKernelGGL((wgrad_ring_kernel<T, BO, BI, WO, WI, D, KPS>), grid, dim3(64 * WO * WI), lds, s, a);
    LH_LAUNCH_CHECK("wgrad_ring launch");
    return LH_OK;
}

// returns 1 when the configuration is not compiled in
int lh_wgrad_ring_launch_f16(const WgradArgs& a, const WgradPlan& c, hipStream_t s) {
#define X(BO, BI, WO, WI, D, KPS) \
    if (c.bo == BO && c.bi == BI && c.depth == D && c.kps == KPS) return launch_wgrad_ring<f16, BO, BI, WO, WI, D, KPS>(a, s);
    LH_WGRAD_CFGS(X)
#undef X
    return 1;
}

// multi-problem form: the 4-wave tiles (the batched layers are the small ones); returns 1 for anything else
template <typename T, int BO, int BI, int WO, int WI, int D, int KPS>
static int launch_wgrad_ring_multi(const LhMulti<WgradArgs>& m, hipStream_t s) {
    if constexpr (WO * WI != 4) {
        return 1;
    } else {
        constexpr int lds = D * KPS * (BO * 2 + BI * 2);
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_ring_multi_kernel<T, BO, BI, WO, WI, D, KPS>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (e != hipSuccess) {
                lh_set_error("wgrad_ring_multi: cannot raise dynamic LDS to %d bytes: %s", lds, hipGetErrorString(e));
                return LH_ERR_HIP;
            }
        }
        hipLaunchKernelGGL((wgrad_ring_multi_kernel<T, BO, BI, WO, WI, D, KPS>), dim3(m.first[m.n]), dim3(64 * WO * WI), lds, s, m);
        LH_LAUNCH_CHECK("wgrad_ring_multi launch");
        return LH_OK;
    }
}

int lh_wgrad_ring_multi_launch_f16(const LhMulti<WgradArgs>& m, const WgradPlan& c, hipStream_t s) {
#define X(BO, BI, WO, WI, D, KPS) \
    if (c.bo == BO && c.bi == BI && c.depth == D && c.kps == KPS) return launch_wgrad_ring_multi<f16, BO, BI, WO, WI, D, KPS>(m, s);
    LH_WGRAD_CFGS(X)
#undef X
    return 1;
}
