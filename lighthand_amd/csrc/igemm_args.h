// Kernel-argument block shared by the two implicit-GEMM kernels (igemm.hip, igemm_ring.hip).
#pragma once
#include "common.h"

struct IgemmArgs {
    const unsigned char* in;
    const unsigned char* w;
    unsigned char* out;
    const unsigned char* addend;
    const unsigned char* addend_mask; // optional: ReLU mask bits (lh_fuse_fwd relu_mask) gating the addend element-wise
    const unsigned char* zero;       // 16 zero bytes in device memory (LDS-DMA kernel: what masked lanes fetch)
    unsigned char* dump;             // 1 KiB nobody reads: where the persistent kernels' masked lanes STORE (fixed store count per tile)
    const float* bias;
    const float* scale;      // optional per-output-channel affine applied to the fp32 accumulator (eval-mode BN fold)
    const float* shift;
    float* stats;
    // BatchNorm-backward gate (lh_igemm_gated): the launch's output is the gradient of relu(BN(gx)); the epilogue stores the
    // gated gradient and `stats` receives { sum g, sum g * xhat } instead of { sum v, sum v^2 }
    const unsigned char* gx;
    const float* gmean;
    const float* ginv;
    const float* gscale;
    const float* gshift;
    const unsigned char* gmask;      // optional: the activation is a residual tail relu(BN(gx) + r) -- its sign comes from the stored mask bits
    const unsigned char* gx2;        // optional (with gmask, pointwise kernel): r = BN2(gx2), a projection shortcut -- `stats2` takes { sum g, sum g * xhat2 }
    const float* gmean2;
    const float* ginv2;
    float* stats2;
    // Fused 1x1 head (lh_igemm_phases_head, 256 x 256 tile only): the tile -- after affine + ReLU -- is multiplied by
    // head_w [>= 32 rows][cout] (K-major pack rows of the 1x1 convolution, rows >= head_j zero) inside the epilogue and
    // only head_out[n][j][OH][OW] (fp32) is written; `out` is not touched
    const unsigned char* head_w;
    const float* head_bias;
    float* head_out;
    int head_j, head_wstride;       // valid head channels (<= 32), bytes between head_w rows
    int n, hi, wi, in_pix_stride, k_run, kspt, kpad;   // kpad: elements per (row, tap) of the weight pack
    int ho, wo, M, sh, sw, cout;
    int OH, OW, osh, osw, ooh, oow, out_pix_stride;
    int ntaps, relu;
    // Phase batching (lh_igemm_phases): up to 4 launches that differ only in weight pack, tap grid, output placement and
    // stats rows (the sub-pixel phases of stride-2 transposed forms) run as ONE grid of nphase * phase_blocks work items.
    int nphase, phase_blocks;
    const unsigned char* ph_w[4];
    int ph_ntaps[4], ph_tw[4], ph_dh0[4], ph_dhs[4], ph_dw0[4], ph_dws[4], ph_ooh[4], ph_oow[4], ph_row0[4];
    int xcd;                         // ring kernel: XCD-aware work-item order (always on: speed only)
    int pw_g, pw_cb;                 // pointwise kernel (igemm_pw_kernel.h): workgroups per channel block, channel blocks
    int tw, dh0, dhs, dw0, dws;      // regular tap grid (ring kernel): tap t = (t / tw, t % tw)
    signed char dh[64];
    signed char dw[64];
};

// One configuration of the LDS-DMA kernel: tile (output channels x pixels), ring depth, K bytes per stage
// (igemm_ring_kernel.h).  depth == 100 selects the direct 3x3 kernel (conv3x3_direct_kernel.h): bm = 64, bp = 256 (a 16 x 16
// output tile), kb = input channels per tap (32 | 64);  depth == 1 selects the persistent pointwise kernel (igemm_pw_kernel.h): bm = channels of the
// resident weight panel, bp = pixels a wave takes per step (16 * PT), kb = padded K of the panel in elements.
struct RingCfg {
    int bm, bp, depth, kb;
};

// workgroups per channel block of a pointwise launch: every CU holds `occ` workgroups for the whole launch (occ = what the
// occupancy query reports for the instantiation, 1..4: lh_pw_occupancy)
static inline int lh_pw_lds_bytes(int bm, int kc, int pt, int gate = 0) {      // panel, staging, per-channel constants (2 vectors; 4 / 6 with the gate of one / two BatchNorm terms)
    return bm * kc * 2 + 4 * pt * 16 * (64 * 2 + 8) + (2 + 2 * gate) * bm * 4;
}

static inline void lh_pw_grid(int bm, int kc, int pt, long M, int cout, int occ, int* G, int* CB) {
    const int cb = (cout + bm - 1) / bm;
    const long ntile = (M + pt * 16 - 1) / (pt * 16);
    long g = 256L * occ / cb / 8 * 8;
    const long need = ((ntile + 3) / 4 + 7) / 8 * 8;           // no more workgroups than there are wave tiles
    if (g > need) g = need;
    if (g < 8) g = 8;
    *G = (int)g;
    *CB = cb;
}


// igemm_ring.hip
bool lh_ring_supported(const lh_igemm_desc* d, int dtype);
bool lh_tap_grid(const lh_igemm_desc* d, int* tw, int* dh0, int* dhs, int* dw0, int* dws);
int lh_ring_resolve(const lh_igemm_desc* d, int dtype, RingCfg* out);
void lh_ring_default_cfg(const lh_igemm_desc* d, int dtype, RingCfg* out);
int lh_ring_candidates(const lh_igemm_desc* d, int dtype, int* out, int max);
int lh_igemm_ring_launch(const IgemmArgs& a, const RingCfg& c, int dtype, hipStream_t s);
template <typename A> struct LhMulti;
int lh_igemm_ring_multi_launch(LhMulti<IgemmArgs>& m, const RingCfg& c, int dtype, hipStream_t s);
bool lh_pw_supported(const lh_igemm_desc* d, int dtype);
bool lh_d3_supported(const lh_igemm_desc* d, int dtype);
int lh_d3_rows(const lh_igemm_desc* d);
int lh_pw_rows(const lh_igemm_desc* d, const RingCfg& c, int dtype, int gate = 0);
int lh_pw_occupancy(const RingCfg& c, int dtype, int mode);      // mode: 0 plain, 1 statistics, 2 / 3 BatchNorm-backward gate of one / two terms
