// lh_bottleneck_infer: the eval-mode stride-1 ResNet bottleneck (conv1 1x1 -> conv2 3x3 -> conv3 1x1 + residual + ReLU, BatchNorm
// folded) as ONE persistent launch -- bottleneck_infer_kernel.h.  Host side: argument checks, the zero / dump pages, the grid.
#include "common.h"
#include "bottleneck_infer_kernel.h"
#include <mutex>
#include <cstring>
#include <cstdlib>

const unsigned char* lh_ring_zero_page();      // igemm_ring.hip: 16 zero bytes / 1 KiB nobody reads, per device
unsigned char* lh_ring_dump_page();

extern "C" int lh_bottleneck_infer(const lh_bottleneck_desc* d, const void* x, const void* w1, const void* w2, const void* w3,
                                   const float* s1, const float* b1, const float* s2, const float* b2, const float* s3, const float* b3,
                                   const void* residual, void* out, int dtype, void* stream) {
    LH_REQUIRE(d && x && w1 && w2 && w3 && s1 && b1 && s2 && b2 && s3 && b3 && residual && out, "lh_bottleneck_infer: null argument");
    LH_REQUIRE(dtype == LH_BF16 || dtype == LH_F16, "lh_bottleneck_infer: 16-bit types only (dtype %d)", dtype);
    LH_REQUIRE(d->n > 0 && d->h > 0 && d->w > 0, "lh_bottleneck_infer: empty problem %d x %d x %d", d->n, d->h, d->w);
    LH_REQUIRE(d->mid == 64 && d->cout == 256, "lh_bottleneck_infer: mid %d / cout %d (this kernel: 64 / 256, the first ResNet stage)", d->mid, d->cout);
    LH_REQUIRE(d->cin >= 64 && d->cin % 32 == 0 && d->cin <= 1024, "lh_bottleneck_infer: cin %d (a multiple of 32 in 64 .. 1024)", d->cin);
    LH_REQUIRE(out != x && out != residual, "lh_bottleneck_infer: the output may not alias the input or the residual (neighbouring tiles read the halo)");
    const long nt = (long)d->n * ceil_div(d->h, 16) * ceil_div(d->w, 16);
    LH_REQUIRE(nt < (1L << 30) && (long)d->n * d->h * d->w * d->cout * 2 < (1L << 40), "lh_bottleneck_infer: problem too large");
    BottleneckArgs a;
    memset((void*)&a, 0, sizeof a);
    a.p3.out = (unsigned char*)out; a.p3.addend = (const unsigned char*)residual; a.p3.out_pix_stride = d->cout; a.p3.cout = d->cout;
    a.p3.relu = 1;
    a.p3.zero = lh_ring_zero_page(); a.p3.dump = lh_ring_dump_page();
    if (!a.p3.zero || !a.p3.dump) {
        lh_set_error("lh_bottleneck_infer: cannot resolve the zero page on this device");
        return LH_ERR_HIP;
    }
    a.x = (const unsigned char*)x; a.w1 = (const unsigned char*)w1; a.w2 = (const unsigned char*)w2; a.w3 = (const unsigned char*)w3;
    a.s1 = s1; a.b1 = b1; a.s2 = s2; a.b2 = b2; a.s3 = s3; a.b3 = b3;
    a.n = d->n; a.h = d->h; a.w = d->w; a.cin = d->cin; a.kpad1 = (d->cin + 63) / 64 * 64;
    // 16 x 16 tiles, one 8-wave workgroup per CU.  (An 8 x 16 / 4-wave form with two workgroups per CU was equal to 3 % slower -- the launch moves
    // 4.4 GB through the fabric either way: x with its halo, x AGAIN as the residual, the output -- and was removed in round 6.)
    a.grid = (int)(nt < 256 ? nt : 256);
    const int lds = lh_bottleneck_lds_bytes(16, 4);
    hipStream_t s = (hipStream_t)stream;
    const void* fn = dtype == LH_BF16 ? reinterpret_cast<const void*>(&bottleneck_infer_kernel<bf16, 16, 8, 4>)
                                      : reinterpret_cast<const void*>(&bottleneck_infer_kernel<f16, 16, 8, 4>);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) {
        lh_set_error("lh_bottleneck_infer: cannot raise dynamic LDS to %d bytes: %s", lds, hipGetErrorString(e));
        return LH_ERR_HIP;
    }
    if (dtype == LH_BF16) hipLaunchKernelGGL((bottleneck_infer_kernel<bf16, 16, 8, 4>), dim3(a.grid), dim3(512), lds, s, a);
    else hipLaunchKernelGGL((bottleneck_infer_kernel<f16, 16, 8, 4>), dim3(a.grid), dim3(512), lds, s, a);
    LH_LAUNCH_CHECK("bottleneck_infer launch");
    return LH_OK;
}
