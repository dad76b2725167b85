// Host side of the LDS-DMA convolution kernel (igemm_ring_kernel.h): which kernel configurations exist, which of them
// fit a launch, the static default choice, and the dispatch into the per-type instantiation files.
#include "common.h"

#include "igemm_args.h"
#include "multi.h"
#include "igemm_ring_cfgs.h"
#include "igemm_pw_cfgs.h"
#include <stdlib.h>
#include <cstdlib>
#include <mutex>

int lh_ring_launch_bf16_big(const IgemmArgs& a, const RingCfg& c, hipStream_t s);
int lh_ring_launch_bf16_mid(const IgemmArgs& a, const RingCfg& c, hipStream_t s);
int lh_ring_launch_bf16_small(const IgemmArgs& a, const RingCfg& c, hipStream_t s);
int lh_ring_launch_f16_big(const IgemmArgs& a, const RingCfg& c, hipStream_t s);
int lh_ring_launch_f16_mid(const IgemmArgs& a, const RingCfg& c, hipStream_t s);
int lh_ring_launch_f16_small(const IgemmArgs& a, const RingCfg& c, hipStream_t s);
int lh_ring_launch_bf16_dense(const IgemmArgs& a, const RingCfg& c, hipStream_t s);
int lh_ring_launch_f16_dense(const IgemmArgs& a, const RingCfg& c, hipStream_t s);
int lh_ring_launch_bf16_ksplit(const IgemmArgs& a, const RingCfg& c, hipStream_t s);
int lh_ring_launch_f16_ksplit(const IgemmArgs& a, const RingCfg& c, hipStream_t s);
int lh_ring_launch_f32(const IgemmArgs& a, const RingCfg& c, hipStream_t s);
int lh_ring_multi_launch_bf16(const LhMulti<IgemmArgs>& m, const RingCfg& c, hipStream_t s);
int lh_ring_multi_launch_f16(const LhMulti<IgemmArgs>& m, const RingCfg& c, hipStream_t s);
int lh_pw_launch_bf16(const IgemmArgs& a, const RingCfg& c, hipStream_t s);
int lh_pw_launch_f16(const IgemmArgs& a, const RingCfg& c, hipStream_t s);
int lh_d3_launch_bf16(const IgemmArgs& a, const RingCfg& c, hipStream_t s);
int lh_d3_launch_f16(const IgemmArgs& a, const RingCfg& c, hipStream_t s);
int lh_pw_occ_bf16(const RingCfg& c, int mode);
int lh_pw_occ_f16(const RingCfg& c, int mode);

static const RingCfg kCfg16[] = {
#define X(BM, BP, WC, WP, D, KB) {BM, BP, D, KB},
    LH_RING_CFGS_16BIT(X)
#undef X
#define X(BM, BP, WC, WP, D, KB) {BM, BP, D + LH_DENSE_DEPTH, KB},
    LH_RING_CFGS_DENSE(X)
#undef X
#define X(BM, BP, WC, WP, D, KB) {BM, BP, D + LH_KSPLIT_DEPTH, KB},
    LH_RING_CFGS_KSPLIT(X)
#undef X
};
static const RingCfg kCfg32[] = {
#define X(BM, BP, WC, WP, D, KB) {BM, BP, D, KB},
    LH_RING_CFGS_F32(X)
#undef X
};

struct PwCfg { int bm, kc, pt; };
static const PwCfg kCfgPw[] = {
#define X(BM, KC, PT) {BM, KC, PT},
    LH_PW_CFGS(X)
#undef X
};

static void cfg_table(int dtype, const RingCfg** t, int* n) {
    if (dtype == LH_F32) { *t = kCfg32; *n = (int)(sizeof(kCfg32) / sizeof(RingCfg)); }
    else { *t = kCfg16; *n = (int)(sizeof(kCfg16) / sizeof(RingCfg)); }
}

// ring depth of a tiled configuration (the dense-wave forms carry it as depth + LH_DENSE_DEPTH, the K-split forms as depth + LH_KSPLIT_DEPTH)
static inline int ring_depth(const RingCfg& c) { return c.depth >= 100 ? c.depth : c.depth % 10 == 0 ? 10 : c.depth % 10; }
static inline bool ring_dense(const RingCfg& c) { return c.depth >= LH_DENSE_DEPTH && c.depth < LH_DENSE_DEPTH + 10; }
static inline bool ring_ksplit(const RingCfg& c) { return c.depth >= LH_KSPLIT_DEPTH && c.depth < LH_KSPLIT_DEPTH + 10; }

static bool cfg_exists(int dtype, const RingCfg& c) {
    const RingCfg* t; int n;
    cfg_table(dtype, &t, &n);
    for (int i = 0; i < n; ++i)
        if (t[i].bm == c.bm && t[i].bp == c.bp && t[i].depth == c.depth && t[i].kb == c.kb) return true;
    return false;
}

// Taps of every convolution form on this path are a regular grid: tap t = (t / tw, t % tw) with
// dh = dh0 + (t / tw) * dhs, dw = dw0 + (t % tw) * dws.  Returns false for an irregular list.
bool lh_tap_grid(const lh_igemm_desc* d, int* tw, int* dh0, int* dhs, int* dw0, int* dws) {
    const int n = d->ntaps;
    if (n <= 0) return false;
    int w = 1;
    while (w < n && d->dh[w] == d->dh[0]) ++w;
    if (n % w) return false;
    *tw = w; *dh0 = d->dh[0]; *dw0 = d->dw[0];
    *dws = w > 1 ? d->dw[1] - d->dw[0] : 0;
    *dhs = n > w ? d->dh[w] - d->dh[0] : 0;
    for (int t = 0; t < n; ++t)
        if (d->dh[t] != *dh0 + (t / w) * *dhs || d->dw[t] != *dw0 + (t % w) * *dws) return false;
    return true;
}

// LDS-DMA moves 16 bytes per lane, so every (pixel, tap) row start must be 16-byte aligned: either the pixel rows
// themselves are (C_in * elt % 16 == 0), or -- the NHWC4 stem, 8-byte pixels -- the horizontal stride, the image row
// pitch and every tap's column offset are each a multiple of 16 bytes (7x7/s2 with row taps: 2 pixels per step).
bool lh_ring_supported(const lh_igemm_desc* d, int dtype) {
    const int es = lh_dtype_size(dtype);
    int tw, dh0, dhs, dw0, dws;
    if (d->ntaps <= 0 || d->ntaps > 32 || !lh_tap_grid(d, &tw, &dh0, &dhs, &dw0, &dws)) return false;     // per-lane tap mask: 32 bits
    const long ps = (long)d->in_pix_stride * es;
    if (ps % 16 == 0) return true;
    return (ps * d->sw) % 16 == 0 && (ps * d->wi) % 16 == 0 && (ps * dw0) % 16 == 0 && (ps * dws) % 16 == 0;
}

// A configuration fits a launch when its tile is not wider than the problem rounded up to the smallest tile
// (a 128-channel tile on <= 64 output channels only multiplies zeros) and its ring is not deeper than the K loop.
// Hard rules (an explicit lh_igemm_desc.cfg is refused when it breaks one): what the kernel's loads assume about the pack.
static bool cfg_safe(const lh_igemm_desc* d, int dtype, const RingCfg& c) {
    const int es = lh_dtype_size(dtype);
    if (c.bm == 256 && ((d->cout + 127) / 128) % 2 != 0) return false;   // weight packs are padded to 128 rows, not 256
    if (c.kb == 128 && d->k_run * es <= 64) return false;            // the second K slice would be all padding
    return true;
}

static bool cfg_fits(const lh_igemm_desc* d, int dtype, const RingCfg& c) {
    const int es = lh_dtype_size(dtype);
    const long M = (long)d->n * d->ho * d->wo;
    const int stages = d->ntaps * ((d->k_run * es + c.kb - 1) / c.kb);
    if (!cfg_safe(d, dtype, c)) return false;
    if (c.bm > 64 && d->cout <= c.bm / 2) return false;               // (these three only multiply zeros / repeat a shallower ring)
    if (c.bp > 64 && M <= c.bp / 2) return false;
    if (ring_depth(c) > 2 && ring_depth(c) - 1 > stages) return false;
    // dense-wave forms (LH_DENSE_TILES=0: not offered).  Built for launches that leave a CU one workgroup; measured 2-9 % ahead
    // on larger launches as well (two co-resident workgroups = four waves per SIMD), so every launch is offered them
    if (ring_dense(c)) {
        const char* sw = getenv("LH_DENSE_TILES");
        if (sw && atoi(sw) == 0) return false;
    }
    // the K-split wave-pair forms: measured EQUAL to the dense-wave forms on every layer of the benchmark networks (round 6: stage-3 3x3 30.6-30.9
    // vs 30.4-30.5 us, stage-4 3x3 48.8 vs 48.7, the step 8.94 vs 8.95 ms with fresh measurements) -- a third fewer LDS fragment reads buy
    // nothing because the LDS is not what binds this loop (profiles/r06_ksplit_ablation.txt).  They change the accumulation order, so they are
    // offered to the tuner only with LH_KSPLIT_TILES=1 (every default configuration then stays bit-equal to the others); an explicit cfg runs them.
    if (ring_ksplit(c)) {
        const char* sw = getenv("LH_KSPLIT_TILES");
        if (!(sw && atoi(sw) != 0) || stages < 2) return false;
    }
    return true;
}

// ---- persistent pointwise kernel (igemm_pw_kernel.h): one tap at (0, 0), dense output, K <= 512, 16-bit types
bool lh_pw_supported(const lh_igemm_desc* d, int dtype) {
    if (lh_dtype_size(dtype) != 2) return false;
    if (d->ntaps != 1 || d->dh[0] != 0 || d->dw[0] != 0) return false;
    if (d->osh != 1 || d->osw != 1 || d->OH != d->ho || d->OW != d->wo || d->ooh != 0 || d->oow != 0) return false;
    if (d->k_run > 512 || d->k_run % 8 != 0 || (d->in_pix_stride * 2) % 16 != 0) return false;
    return true;
}

static int pw_kc(const lh_igemm_desc* d) { return d->k_run <= 64 ? 64 : d->k_run <= 128 ? 128 : d->k_run <= 256 ? 256 : 512; }

static bool pw_fits(const lh_igemm_desc* d, int dtype, const RingCfg& c) {
    if (!lh_pw_supported(d, dtype) || c.kb != pw_kc(d)) return false;
    if (c.bm > 64 && d->cout <= c.bm / 2) return false;
    for (const PwCfg& k : kCfgPw)
        if (k.bm == c.bm && k.kc == c.kb && 16 * k.pt == c.bp) return true;
    return false;
}

int lh_pw_occupancy(const RingCfg& c, int dtype, int mode) {
    return dtype == LH_BF16 ? lh_pw_occ_bf16(c, mode) : lh_pw_occ_f16(c, mode);
}

// rows of the statistics slab a pointwise launch writes: one per workgroup of a channel block
int lh_pw_rows(const lh_igemm_desc* d, const RingCfg& c, int dtype, int gate) {
    int g, cb;
    lh_pw_grid(c.bm, c.kb, c.bp / 16, (long)d->n * d->ho * d->wo, d->cout, lh_pw_occupancy(c, dtype, 1 + gate), &g, &cb);
    return g;
}

// ---- direct 3x3 kernel (conv3x3_direct_kernel.h): nine taps (-1..1)^2 in either order, stride 1, same-size dense output,
//      32 or 64 input channels per tap, 16-bit types
bool lh_d3_supported(const lh_igemm_desc* d, int dtype) {
    if (lh_dtype_size(dtype) != 2 || d->ntaps != 9) return false;
    if (d->k_run != 32 && d->k_run != 64) return false;
    if (d->sh != 1 || d->sw != 1 || d->hi != d->ho || d->wi != d->wo) return false;
    if (d->osh != 1 || d->osw != 1 || d->OH != d->ho || d->OW != d->wo || d->ooh != 0 || d->oow != 0) return false;
    if ((d->in_pix_stride * 2) % 16 != 0) return false;
    unsigned seen = 0;
    for (int t = 0; t < 9; ++t) {
        if (d->dh[t] < -1 || d->dh[t] > 1 || d->dw[t] < -1 || d->dw[t] > 1) return false;
        seen |= 1u << ((d->dh[t] + 1) * 3 + d->dw[t] + 1);
    }
    return seen == 0x1ffu;
}

static bool d3_fits(const lh_igemm_desc* d, int dtype, const RingCfg& c) {
    return lh_d3_supported(d, dtype) && c.bm == 64 && c.bp == 256 && c.kb == d->k_run;
}

int lh_d3_rows(const lh_igemm_desc* d) {
    const int cb = (d->cout + 63) / 64;
    const int patch = 18 * 18 * d->k_run * 2, stg = 8 * 2 * 16 * 136;
    const int lds = 9 * 64 * d->k_run * 2 + 2 * patch + (stg <= patch ? 0 : stg) + 512;
    const int occ = lds <= 80 * 1024 ? 2 : 1;
    const long ntile = (long)d->n * ((d->ho + 15) / 16) * ((d->wo + 15) / 16);
    long g = 256L * occ / cb / 8 * 8;
    const long need = (ntile + 7) / 8 * 8;
    if (g > need) g = need;
    if (g < 8) g = 8;
    return (int)g;
}

// Static default (cfg all zero): the largest tile that still gives >= 2 workgroups per CU, a 2-stage ring for K loops
// of <= 4 steps (store-bound 1x1 convolutions: 4 workgroups share a CU) else 4 stages, no look-ahead.  The plan's
// autotuner replaces this by a measured choice (lh_igemm_candidates).
void lh_ring_default_cfg(const lh_igemm_desc* d, int dtype, RingCfg* out) {
    const long M = (long)d->n * d->ho * d->wo;
    const int es = lh_dtype_size(dtype);
    const int steps = d->ntaps * ((d->k_run * es + 63) / 64);
    const bool f32 = dtype == LH_F32;
    out->kb = 64;
    if (!f32 && d->cout % 256 == 0 && steps > 4 && ((M + 255) / 256) * (d->cout / 256) >= 256) {
        out->bm = 256; out->bp = 256; out->depth = 3;
        return;
    }
    const int cands[5][2] = {{128, 256}, {128, 128}, {128, 64}, {64, 128}, {64, 64}};
    int bm = 64, bp = 64;
    for (int i = 0; i < 5; ++i) {
        const int BM = cands[i][0], BP = cands[i][1];
        if (BM == 128 && d->cout <= 64) continue;
        if (BP == 256 && (f32 || steps <= 4)) continue;                 // 16-bit, deep K only
        if (f32 && BM == 128 && BP == 128) continue;                    // fp32 epilogue tile would not fit 64 KiB well
        const long blocks = ((M + BP - 1) / BP) * ((d->cout + BM - 1) / BM);
        if (blocks >= (BP == 256 ? 1024 : 512) || i == 4) { bm = BM; bp = BP; break; }
    }
    out->bm = bm; out->bp = bp;
    out->depth = bp == 256 ? 3 : (steps <= 4 ? 2 : 4);
}

// Resolve the configuration of a launch: the descriptor's explicit choice when given (validated), else the default.
int lh_ring_resolve(const lh_igemm_desc* d, int dtype, RingCfg* out) {
    if (d->cfg[0] == 0) {
        lh_ring_default_cfg(d, dtype, out);
        return LH_OK;
    }
    const RingCfg c = {d->cfg[0], d->cfg[1], d->cfg[2], d->cfg[3]};
    if (c.depth == 100) {                                // direct 3x3 kernel
        if (!d3_fits(d, dtype, c)) {
            lh_set_error("igemm: the direct 3x3 configuration (64, 256, 100, %d) does not fit this launch", c.kb);
            return LH_ERR_ARG;
        }
        *out = c;
        return LH_OK;
    }
    if (c.depth == 1) {                                  // persistent pointwise kernel
        if (!pw_fits(d, dtype, c)) {
            lh_set_error("igemm: pointwise configuration panel %d x K %d, %d pixels per wave does not exist or does not fit this launch",
                         c.bm, c.kb, c.bp);
            return LH_ERR_ARG;
        }
        *out = c;
        return LH_OK;
    }
    if (!cfg_exists(dtype, c)) {
        lh_set_error("igemm: configuration tile %dx%d depth %d kb %d is not compiled in for dtype %d", c.bm, c.bp, c.depth, c.kb, dtype);
        return LH_ERR_UNSUPPORTED;
    }
    // an explicit choice must pass the hard rules of the candidate list: weight packs are padded to 128 rows, so e.g. a
    // 256-row tile on a pack with an odd number of 128-row blocks would fetch past its end (the phases of a batched
    // launch share the lead's choice, so the soft rules -- tile wider than the problem, ring deeper than the loop -- stay legal)
    if (!cfg_safe(d, dtype, c)) {
        lh_set_error("igemm: configuration tile %dx%d depth %d kb %d does not fit this launch (cout %d, k_run %d, %d taps)",
                     c.bm, c.bp, c.depth, c.kb, d->cout, d->k_run, d->ntaps);
        return LH_ERR_ARG;
    }
    *out = c;
    return LH_OK;
}

int lh_ring_candidates(const lh_igemm_desc* d, int dtype, int* out, int max) {
    const RingCfg* t; int n, k = 0;
    cfg_table(dtype, &t, &n);
    for (int i = 0; i < n && k < max; ++i) {
        if (!cfg_fits(d, dtype, t[i])) continue;
        out[5 * k] = t[i].bm; out[5 * k + 1] = t[i].bp; out[5 * k + 2] = t[i].depth; out[5 * k + 3] = t[i].kb; out[5 * k + 4] = 0;
        ++k;
    }
    if (k < max && lh_d3_supported(d, dtype)) {          // direct 3x3 kernel: depth = 100
        out[5 * k] = 64; out[5 * k + 1] = 256; out[5 * k + 2] = 100; out[5 * k + 3] = d->k_run; out[5 * k + 4] = 0;
        ++k;
    }
    for (const PwCfg& c : kCfgPw) {                      // pointwise configurations: depth = 1
        const RingCfg r = {c.bm, 16 * c.pt, 1, c.kc};
        if (k >= max || !pw_fits(d, dtype, r)) continue;
        out[5 * k] = r.bm; out[5 * k + 1] = r.bp; out[5 * k + 2] = 1; out[5 * k + 3] = r.kb; out[5 * k + 4] = 0;
        ++k;
    }
    return k;
}

// 16 zero bytes in device memory: the source of every LDS-DMA lane that falls outside the image or the K run.
// One copy per device (module memory); its address is looked up once per device.
__device__ __attribute__((aligned(16))) unsigned int lh_zero_page[4] = {0u, 0u, 0u, 0u};

// 1 KiB that is only ever written: the persistent kernels store the lanes that fall outside the problem here instead of
// predicating the store away, so every tile issues the same number of store instructions and the counted waits
// (s_waitcnt vmcnt) that keep a tile's stores in flight across the next tile stay exact.
__device__ __attribute__((aligned(16))) unsigned int lh_dump_page[256];

static unsigned char* dump_page() {
    static std::mutex mu;
    static unsigned char* ptr[64] = {nullptr};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (!ptr[dev]) {
        void* q = nullptr;
        if (hipGetSymbolAddress(&q, HIP_SYMBOL(lh_dump_page)) != hipSuccess) return nullptr;
        ptr[dev] = (unsigned char*)q;
    }
    return ptr[dev];
}

static const unsigned char* zero_page() {
    static std::mutex mu;
    static const unsigned char* ptr[64] = {nullptr};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (!ptr[dev]) {
        void* q = nullptr;
        if (hipGetSymbolAddress(&q, HIP_SYMBOL(lh_zero_page)) != hipSuccess) return nullptr;
        ptr[dev] = (const unsigned char*)q;
    }
    return ptr[dev];
}

// The buffer form of the tiled kernel's operand path (igemm_ring_kernel.h) addresses a tile's pixels with 32-bit offsets
// relative to the first image the tile touches, and marks masked lanes with offset 2^31: everything a tile can reach --
// the images a tile of `bp` pixels spans, the tap window in front of them, a weight block -- must stay below 2^31 bytes.
static int lh_ring_offsets_fit(const IgemmArgs& a, int bm, int bp, int es) {
    const long ipix = (long)a.in_pix_stride * es, img = (long)a.hi * a.wi * ipix;
    const long hw = (long)a.ho * a.wo > 0 ? (long)a.ho * a.wo : 1;
    long reach = (bp / hw + 2) * img;
    for (int ph = 0; ph < (a.nphase > 1 ? a.nphase : 1); ++ph) {
        const int ntaps = a.nphase > 1 ? a.ph_ntaps[ph] : a.ntaps, tw = a.nphase > 1 ? a.ph_tw[ph] : a.tw;
        const int dh0 = a.nphase > 1 ? a.ph_dh0[ph] : a.dh0, dhs = a.nphase > 1 ? a.ph_dhs[ph] : a.dhs;
        const int dw0 = a.nphase > 1 ? a.ph_dw0[ph] : a.dw0, dws = a.nphase > 1 ? a.ph_dws[ph] : a.dws;
        const int th = tw > 0 ? (ntaps + tw - 1) / tw : 1;
        const long shift = ((long)(std::abs(dh0) + std::abs(dhs) * th) * a.wi + std::abs(dw0) + std::abs(dws) * tw) * ipix;
        LH_REQUIRE(reach + 2 * shift + (long)a.k_run * es + 256 < (1L << 31) && (long)bm * ntaps * a.kpad * es < (1L << 31),
                   "igemm_ring: a %d-pixel tile of this launch reaches %ld bytes of input (%d x %d pixels of %ld bytes per image, tap window %ld) "
                   "or %ld bytes of weights: more than the 2 GiB a workgroup's buffer descriptor addresses",
                   bp, reach + 2 * shift, a.hi, a.wi, ipix, shift, (long)bm * ntaps * a.kpad * es);
    }
    return LH_OK;
}

const unsigned char* lh_ring_zero_page() { return zero_page(); }      // bottleneck_infer.hip
unsigned char* lh_ring_dump_page() { return dump_page(); }

int lh_igemm_ring_launch(const IgemmArgs& a0, const RingCfg& c, int dtype, hipStream_t s) {
    IgemmArgs a = a0;
    a.zero = zero_page();
    a.dump = dump_page();
    if (!a.zero || !a.dump) {
        lh_set_error("igemm_ring: cannot resolve the zero page on this device");
        return LH_ERR_HIP;
    }
    int rc = 1;
    if (c.depth == 100) {
        if (dtype == LH_BF16) rc = lh_d3_launch_bf16(a, c, s);
        else if (dtype == LH_F16) rc = lh_d3_launch_f16(a, c, s);
        if (rc == 1) {
            lh_set_error("conv3x3_direct: no kernel for %d input channels, dtype %d", c.kb, dtype);
            return LH_ERR_UNSUPPORTED;
        }
        return rc;
    }
    if (c.depth == 1) {
        if (dtype == LH_BF16) rc = lh_pw_launch_bf16(a, c, s);
        else if (dtype == LH_F16) rc = lh_pw_launch_f16(a, c, s);
        if (rc == 1) {
            lh_set_error("igemm_pw: no kernel for panel %d x K %d, %d pixels per wave, dtype %d", c.bm, c.kb, c.bp, dtype);
            return LH_ERR_UNSUPPORTED;
        }
        return rc;
    }
    if (int e = lh_ring_offsets_fit(a, c.bm, c.bp, dtype == LH_F32 ? 4 : 2)) return e;
    switch (dtype) {
        case LH_BF16:
            rc = lh_ring_launch_bf16_big(a, c, s);
            if (rc == 1) rc = lh_ring_launch_bf16_mid(a, c, s);
            if (rc == 1) rc = lh_ring_launch_bf16_small(a, c, s);
            if (rc == 1) rc = lh_ring_launch_bf16_dense(a, c, s);
            if (rc == 1) rc = lh_ring_launch_bf16_ksplit(a, c, s);
            break;
        case LH_F16:
            rc = lh_ring_launch_f16_big(a, c, s);
            if (rc == 1) rc = lh_ring_launch_f16_mid(a, c, s);
            if (rc == 1) rc = lh_ring_launch_f16_small(a, c, s);
            if (rc == 1) rc = lh_ring_launch_f16_dense(a, c, s);
            if (rc == 1) rc = lh_ring_launch_f16_ksplit(a, c, s);
            break;
        case LH_F32:
            rc = lh_ring_launch_f32(a, c, s);
            break;
        default:
            lh_set_error("igemm_ring: unsupported dtype %d", dtype);
            return LH_ERR_ARG;
    }
    if (rc == 1) {
        lh_set_error("igemm_ring: no kernel for tile %dx%d depth %d kb %d dtype %d", c.bm, c.bp, c.depth, c.kb, dtype);
        return LH_ERR_UNSUPPORTED;
    }
    return rc;
}

int lh_igemm_ring_multi_launch(LhMulti<IgemmArgs>& m, const RingCfg& c, int dtype, hipStream_t s) {
    const unsigned char* z = zero_page();
    if (!z) {
        lh_set_error("igemm_ring_multi: cannot resolve the zero page on this device");
        return LH_ERR_HIP;
    }
    for (int i = 0; i < m.n; ++i) { m.a[i].zero = z; m.a[i].dump = dump_page(); }
    for (int i = 0; i < m.n; ++i)
        if (int e = lh_ring_offsets_fit(m.a[i], c.bm, c.bp, 2)) return e;
    int rc = 1;
    if (dtype == LH_BF16) rc = lh_ring_multi_launch_bf16(m, c, s);
    else if (dtype == LH_F16) rc = lh_ring_multi_launch_f16(m, c, s);
    if (rc == 1) {
        lh_set_error("igemm_ring_multi: no multi-problem kernel for tile %dx%d depth %d kb %d dtype %d (4-wave tiles, 16-bit types)",
                     c.bm, c.bp, c.depth, c.kb, dtype);
        return LH_ERR_UNSUPPORTED;
    }
    return rc;
}
