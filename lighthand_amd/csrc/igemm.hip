// Implicit-GEMM gather convolution for gfx950 (MI355X): forward conv, data-gradient,
// and the sub-pixel phases of stride-2 transposed convolutions, all through one kernel.
//
//   out[pixel][co] = sum_{tap} sum_{k < k_run} in[pix(pixel, tap)][k] * wpack[co][tap][k]
//
// GEMM view: D[co][pixel] (MFMA "A" = weights, "B" = gathered pixels, both K-major in LDS),
// so each lane ends up with 4 consecutive output channels of one pixel (NHWC-friendly).
//  * 256 threads = 4 waves, tile BM (channels) x BP (pixels), K step = 64 bytes per row
//    (32 bf16 / 16 fp32): one v_mfma_f32_16x16x32_bf16 (or 4 v_mfma_f32_16x16x4_f32) per
//    16x16 sub-tile and step.
//  * global -> registers -> LDS staging, double buffered, ONE barrier per K step; next
//    step's global loads are in flight while the MFMAs of the current step run.
//  * LDS image: [16-row group][16-B chunk c][row ^ 2c] x 16 B  -- lane-linear for the
//    fragment reads (conflict free) and conflict free for the staging writes.
//  * epilogue through LDS: bias, then full-line NHWC stores with optional addend / ReLU,
//    and per-block column sums / sums of squares of the STORED values for BatchNorm.
#include "common.h"
#include <cstring>
#include <algorithm>
#include <stdlib.h>

#include "igemm_args.h"
#include "multi.h"
#include "igemm_ring_cfgs.h"

template <typename T> struct Mma;
template <> struct Mma<bf16> {
    static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x4& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a),
                                                   __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct Mma<f16> {
    static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x4& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a),
                                                  __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    // lane group g = lane>>4 holds k = 4g..4g+3 of the 16-float step; instruction kk consumes
    // element kk of every group, so A and B agree on the k each slot stands for.
    static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x4& c) {
        const f32x4 fa = __builtin_bit_cast(f32x4, a), fb = __builtin_bit_cast(f32x4, b);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[0], fb[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[1], fb[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[2], fb[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[3], fb[3], c, 0, 0, 0);
    }
};

__device__ __forceinline__ int stage_off(int row, int c) {
    return ((row >> 4) << 10) + (c << 8) + ((((row & 15) ^ (c << 1)) & 15) << 4);
}

template <typename T, int BM, int BP, int WC, int WP>
__global__ __launch_bounds__(256) void igemm_kernel(const IgemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int ES = sizeof(T);
    constexpr int EPC = 16 / ES;          // elements per 16-byte chunk
    constexpr int KSTEP = 64 / ES;        // elements per K step
    constexpr int TC = BM / WC, TP = BP / WP;
    constexpr int CT = TC / 16, PT = TP / 16;
    constexpr int RW = BM / 64, RX = BP / 64;
    constexpr int STAGE = (BM + BP) * 64;
    static_assert(WC * WP == 4 && CT >= 1 && PT >= 1, "bad tile");

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wc = wave / WP, wp = wave % WP;
    const int pblk = blockIdx.x, cblk = blockIdx.y;
    const int c4 = tid & 3, r4 = tid >> 2;

    // ---- per-thread gather state for the pixel rows it stages
    int xbase[RX], xh[RX], xw[RX];
    bool xok[RX];
    const int hw = p.ho * p.wo;
#pragma unroll
    for (int i = 0; i < RX; ++i) {
        const int m = pblk * BP + r4 + 64 * i;
        xok[i] = m < p.M;
        const int mm = xok[i] ? m : 0;
        const int n = mm / hw, rem = mm - n * hw;
        const int a = rem / p.wo, b = rem - a * p.wo;
        xbase[i] = n * p.hi * p.wi;
        xh[i] = a * p.sh;
        xw[i] = b * p.sw;
    }
    const long kpad = p.kpad;
    const unsigned char* wrow[RW];
#pragma unroll
    for (int i = 0; i < RW; ++i)
        wrow[i] = p.w + ((long)(cblk * BM + r4 + 64 * i) * p.ntaps * kpad + c4 * EPC) * ES;

    f32x4 acc[CT][PT];
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
        for (int j = 0; j < PT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    uint4 xr[RX], wr[RW];
    auto gload = [&](int tap, int kc) {
        const int koff = kc * KSTEP + c4 * EPC;
        const int dh = p.dh[tap], dw = p.dw[tap];
#pragma unroll
        for (int i = 0; i < RX; ++i) {
            const int ih = xh[i] + dh, iw = xw[i] + dw;
            const bool ok = xok[i] && (unsigned)ih < (unsigned)p.hi && (unsigned)iw < (unsigned)p.wi &&
                            koff < p.k_run;
            if (ok) {
                const long e = (long)(xbase[i] + ih * p.wi + iw) * p.in_pix_stride + koff;
                xr[i] = *reinterpret_cast<const uint4*>(p.in + e * ES);
            } else {
                xr[i] = uint4{0u, 0u, 0u, 0u};
            }
        }
        const long wo = ((long)tap * kpad + (long)kc * KSTEP) * ES;
#pragma unroll
        for (int i = 0; i < RW; ++i) wr[i] = *reinterpret_cast<const uint4*>(wrow[i] + wo);
    };

    const int S = p.ntaps * p.kspt;
    int tap = 0, kc = 0;
    if (S > 0) gload(0, 0);
    const int lane_off = ((lane >> 4) << 8) + ((((lane & 15) ^ ((lane >> 4) << 1)) & 15) << 4);

    for (int s = 0; s < S; ++s) {
        unsigned char* st = smem + (s & 1) * STAGE;
#pragma unroll
        for (int i = 0; i < RW; ++i)
            *reinterpret_cast<uint4*>(st + stage_off(r4 + 64 * i, c4)) = wr[i];
#pragma unroll
        for (int i = 0; i < RX; ++i)
            *reinterpret_cast<uint4*>(st + BM * 64 + stage_off(r4 + 64 * i, c4)) = xr[i];
        __syncthreads();
        if (++kc == p.kspt) { kc = 0; ++tap; }
        if (s + 1 < S) gload(tap, kc);

        uint4 fa[CT], fb[PT];
#pragma unroll
        for (int i = 0; i < CT; ++i)
            fa[i] = *reinterpret_cast<const uint4*>(st + ((wc * CT + i) << 10) + lane_off);
#pragma unroll
        for (int j = 0; j < PT; ++j)
            fb[j] = *reinterpret_cast<const uint4*>(st + BM * 64 + ((wp * PT + j) << 10) + lane_off);
#pragma unroll
        for (int i = 0; i < CT; ++i)
#pragma unroll
            for (int j = 0; j < PT; ++j) Mma<T>::run(fa[i], fb[j], acc[i][j]);
    }

    // ---- epilogue: D -> LDS tile [BP][BM] (row stride RS), then full-line stores
    constexpr int RS = BM * ES + 8;
    __syncthreads();
    {
        const int q = lane >> 4, pl = lane & 15;
#pragma unroll
        for (int i = 0; i < CT; ++i) {
            const int col = wc * TC + i * 16 + q * 4;
            float bv[4] = {0.f, 0.f, 0.f, 0.f}, sv[4] = {1.f, 1.f, 1.f, 1.f};
            if (p.bias) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int gc = cblk * BM + col + r;
                    bv[r] = gc < p.cout ? p.bias[gc] : 0.f;
                }
            }
            if (p.scale) {                      // out = acc * scale + shift (+ bias * scale folded by the host if both are given)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int gc = cblk * BM + col + r;
                    sv[r] = gc < p.cout ? p.scale[gc] : 1.f;
                    bv[r] = bv[r] * sv[r] + (gc < p.cout ? p.shift[gc] : 0.f);
                }
            }
#pragma unroll
            for (int j = 0; j < PT; ++j) {
                const int pr = wp * TP + j * 16 + pl;
                T* dst = reinterpret_cast<T*>(smem + pr * RS + col * ES);
                if constexpr (ES == 4) {
                    float2 lo = {acc[i][j][0] * sv[0] + bv[0], acc[i][j][1] * sv[1] + bv[1]};
                    float2 hi = {acc[i][j][2] * sv[2] + bv[2], acc[i][j][3] * sv[3] + bv[3]};
                    reinterpret_cast<float2*>(dst)[0] = lo;
                    reinterpret_cast<float2*>(dst)[1] = hi;
                } else {
                    union { uint2 u; T e[4]; } pk;
#pragma unroll
                    for (int r = 0; r < 4; ++r) pk.e[r] = from_f<T>(acc[i][j][r] * sv[r] + bv[r]);
                    *reinterpret_cast<uint2*>(dst) = pk.u;
                }
            }
        }
    }
    __syncthreads();

    constexpr int CH = BM * ES / 16;        // 16-byte chunks per pixel row
    constexpr int RPP = 256 / CH;           // pixel rows per pass
    const int chunk = tid % CH, r0 = tid / CH;
    const int col0 = cblk * BM + chunk * EPC;
    const bool col_ok = col0 < p.cout;      // cout % EPC == 0 is required by the host
    float s1[EPC], s2[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) s1[e] = s2[e] = 0.f;

    for (int pr = r0; pr < BP; pr += RPP) {
        const int m = pblk * BP + pr;
        if (m >= p.M || !col_ok) continue;
        const int n = m / hw, rem = m - n * hw;
        const int a = rem / p.wo, b = rem - a * p.wo;
        const long opix = ((long)n * p.OH + a * p.osh + p.ooh) * p.OW + b * p.osw + p.oow;
        const long eoff = opix * p.out_pix_stride + col0;
        const unsigned char* src = smem + pr * RS + chunk * 16;
        uint4 u;
        {
            const uint2 lo = *reinterpret_cast<const uint2*>(src);
            const uint2 hi = *reinterpret_cast<const uint2*>(src + 8);
            u = uint4{lo.x, lo.y, hi.x, hi.y};
        }
        float v[EPC];
        unpack16<T>(u, v);
        if (p.addend) {
            float av[EPC];
            unpack16<T>(*reinterpret_cast<const uint4*>(p.addend + eoff * ES), av);
            if (p.addend_mask) {
                const unsigned mb = p.addend_mask[eoff / EPC];
#pragma unroll
                for (int e = 0; e < EPC; ++e) av[e] = ((mb >> e) & 1u) ? av[e] : 0.f;
            }
#pragma unroll
            for (int e = 0; e < EPC; ++e) v[e] += av[e];
        }
        if (p.relu) {
#pragma unroll
            for (int e = 0; e < EPC; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (p.addend || p.relu) u = pack16<T>(v);
        if (p.stats) {
            float sv[EPC];
            unpack16<T>(u, sv);
#pragma unroll
            for (int e = 0; e < EPC; ++e) { s1[e] += sv[e]; s2[e] += sv[e] * sv[e]; }
        }
        *reinterpret_cast<uint4*>(p.out + eoff * ES) = u;
    }

    if (p.stats) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);      // [RPP][2][BM]
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            red[(r0 * 2 + 0) * BM + chunk * EPC + e] = s1[e];
            red[(r0 * 2 + 1) * BM + chunk * EPC + e] = s2[e];
        }
        __syncthreads();
        for (int t = tid; t < 2 * BM; t += 256) {
            const int which = t / BM, col = t - which * BM;
            float a = 0.f;
#pragma unroll 4
            for (int r = 0; r < RPP; ++r) a += red[(r * 2 + which) * BM + col];
            const int gc = cblk * BM + col;
            if (gc < p.cout) p.stats[((long)pblk * 2 + which) * p.cout + gc] = a;
        }
    }
}

// ------------------------------------------------------------------------------------------------
template <typename T, int BM, int BP, int WC, int WP>
static int launch_tile(const IgemmArgs& a, hipStream_t s) {
    constexpr int ES = sizeof(T);
    constexpr int stage = 2 * (BM + BP) * 64;
    constexpr int epi = BP * (BM * ES + 8);
    constexpr int red = (256 / (BM * ES / 16)) * 2 * BM * 4;
    constexpr int lds = stage > epi ? (stage > red ? stage : red) : (epi > red ? epi : red);
    static_assert(lds <= 65536, "LDS budget");
    dim3 grid(ceil_div(a.M, BP), ceil_div(a.cout, BM));
    hipLaunchKernelGGL((igemm_kernel<T, BM, BP, WC, WP>), grid, dim3(256), lds, s, a);
    LH_LAUNCH_CHECK("igemm launch");
    return LH_OK;
}

static void pick_tile(const lh_igemm_desc* d, int dtype, int* bm, int* bp) {
    if (lh_ring_supported(d, dtype)) {
        RingCfg c;
        if (lh_ring_resolve(d, dtype, &c) != LH_OK) lh_ring_default_cfg(d, dtype, &c);
        *bm = c.bm; *bp = c.bp;
        return;
    }
    const long M = (long)d->n * d->ho * d->wo;
    int BM = d->cout > 64 ? 128 : 64;
    int BP = 128;
    if (dtype == LH_F32 && BM == 128) BP = 64;          // LDS budget of the fp32 epilogue tile
    const long blocks = ((M + BP - 1) / BP) * ((d->cout + BM - 1) / BM);
    if (BP == 128 && blocks < 384) BP = 64;
    *bm = BM;
    *bp = BP;
}

extern "C" int lh_igemm_tile(const lh_igemm_desc* d, int dtype, int* bm, int* bp, int* ring) {
    LH_REQUIRE(d && bm && bp && ring, "lh_igemm_tile: null pointer");
    int cfg[5];
    const int rc = lh_igemm_config(d, dtype, cfg);
    if (rc) return rc;
    *bm = cfg[0]; *bp = cfg[1];
    *ring = cfg[2] ? cfg[3] * 10 + cfg[2] : 0;          // K bytes per stage * 10 + ring depth; 0 = register-staged kernel
    return LH_OK;
}

extern "C" int lh_igemm_config(const lh_igemm_desc* d, int dtype, int* cfg5) {
    LH_REQUIRE(d && cfg5, "lh_igemm_config: null pointer");
    if (lh_ring_supported(d, dtype)) {
        RingCfg c;
        const int rc = lh_ring_resolve(d, dtype, &c);
        if (rc) return rc;
        cfg5[0] = c.bm; cfg5[1] = c.bp; cfg5[2] = c.depth; cfg5[3] = c.kb; cfg5[4] = 0;
        return LH_OK;
    }
    pick_tile(d, dtype, &cfg5[0], &cfg5[1]);
    cfg5[2] = cfg5[3] = cfg5[4] = 0;
    return LH_OK;
}

extern "C" int lh_igemm_candidates(const lh_igemm_desc* d, int dtype, int* cfgs, int max) {
    if (!d || !cfgs || max <= 0 || !lh_ring_supported(d, dtype)) return 0;
    return lh_ring_candidates(d, dtype, cfgs, max);
}

extern "C" int lh_igemm_stats_rows(const lh_igemm_desc* d, int dtype) {
    int bm, bp;
    if (d->cfg[2] == 100 && lh_ring_supported(d, dtype)) {        // direct 3x3 kernel: one row per workgroup
        RingCfg c;
        if (lh_ring_resolve(d, dtype, &c) == LH_OK && c.depth == 100) return lh_d3_rows(d);
    }
    if (d->cfg[2] == 1 && lh_ring_supported(d, dtype)) {          // persistent pointwise kernel: one row per workgroup
        RingCfg c;
        if (lh_ring_resolve(d, dtype, &c) == LH_OK && c.depth == 1) return lh_pw_rows(d, c, dtype);
    }
    pick_tile(d, dtype, &bm, &bp);
    const long M = (long)d->n * d->ho * d->wo;
    return (int)((M + bp - 1) / bp);
}

// rows of the partial-sum slab a gated launch (lh_igemm_gated) writes: those of the statistics slab, except on the pointwise kernel, whose
// gate instantiation may hold a different number of workgroups per CU
extern "C" int lh_igemm_gated_rows(const lh_igemm_desc* d, int dtype, int nterms) {
    if (d && d->cfg[2] == 1 && lh_ring_supported(d, dtype)) {
        RingCfg c;
        if (lh_ring_resolve(d, dtype, &c) == LH_OK && c.depth == 1) return lh_pw_rows(d, c, dtype, nterms >= 2 ? 2 : 1);
    }
    return lh_igemm_stats_rows(d, dtype);
}

struct PhaseSet {                 // lh_igemm_phases: the other descriptors / packs of the batch (lead = descs[lead])
    const lh_igemm_desc* const* descs;
    const void* const* wpacks;
    int n;
};

static int igemm_impl(const lh_igemm_desc* d, const void* in, const void* wpack, void* out,
                      const void* addend, const void* addend_mask, const float* bias, const float* scale, const float* shift, float* stats,
                      int dtype, void* stream, const PhaseSet* phases = nullptr, const lh_head* head = nullptr,
                      IgemmArgs* prep_args = nullptr, RingCfg* prep_cfg = nullptr, const lh_bn_bwd_gate* gate = nullptr) {
    LH_REQUIRE(d && in && wpack && (out || head), "lh_igemm: null pointer");
    const int es = lh_dtype_size(dtype);
    LH_REQUIRE(es > 0, "lh_igemm: bad dtype %d", dtype);
    const int epc = 16 / es;
    LH_REQUIRE(d->ntaps >= 0 && d->ntaps <= 64, "lh_igemm: ntaps %d out of range", d->ntaps);
    LH_REQUIRE(d->k_run > 0 && d->k_run % epc == 0, "lh_igemm: k_run %d must be a multiple of %d", d->k_run, epc);
    LH_REQUIRE(d->cout > 0 && d->cout % epc == 0, "lh_igemm: cout %d must be a multiple of %d", d->cout, epc);
    LH_REQUIRE(d->out_pix_stride % epc == 0 && d->out_pix_stride >= d->cout, "lh_igemm: bad out_pix_stride %d", d->out_pix_stride);
    LH_REQUIRE(d->in_pix_stride > 0 && (d->in_pix_stride * es) % 4 == 0, "lh_igemm: bad in_pix_stride");
    LH_REQUIRE(d->n > 0 && d->ho > 0 && d->wo > 0 && d->hi > 0 && d->wi > 0, "lh_igemm: bad sizes");
    LH_REQUIRE((d->ho - 1) * d->osh + d->ooh < d->OH && (d->wo - 1) * d->osw + d->oow < d->OW && d->ooh >= 0 && d->oow >= 0,
               "lh_igemm: output placement exceeds the %dx%d image", d->OH, d->OW);
    IgemmArgs a;
    a.in = (const unsigned char*)in; a.w = (const unsigned char*)wpack; a.out = (unsigned char*)out;
    a.addend = (const unsigned char*)addend; a.bias = bias; a.stats = stats; a.zero = nullptr; a.dump = nullptr;
    a.addend_mask = (const unsigned char*)addend_mask;
    LH_REQUIRE(!addend_mask || (addend && d->out_pix_stride == d->cout), "lh_igemm: addend_mask needs an addend and a dense output (mask bits index 16-byte chunks)");
    a.scale = scale; a.shift = shift;
    LH_REQUIRE((scale == nullptr) == (shift == nullptr), "lh_igemm: scale and shift must come together");
    a.head_w = nullptr; a.head_bias = nullptr; a.head_out = nullptr; a.head_j = 0; a.head_wstride = 0;
    a.gx = nullptr; a.gmean = a.ginv = a.gscale = a.gshift = nullptr; a.gmask = nullptr;
    a.gx2 = nullptr; a.gmean2 = a.ginv2 = nullptr; a.stats2 = nullptr;
    if (gate) {
        LH_REQUIRE(gate->x && gate->mean && gate->invstd && gate->partial && (gate->mask || (gate->scale && gate->shift)),
                   "lh_igemm_gated: null pointer in the gate (the sign of the activation comes from scale + shift or from mask)");
        LH_REQUIRE(!bias && !scale && !d->relu && !phases && !head, "lh_igemm_gated: a data gradient carries no bias / affine / ReLU / phases");
        LH_REQUIRE(!gate->mask || d->out_pix_stride == d->cout, "lh_igemm_gated: gate.mask needs a dense output (mask bits index 16-byte chunks)");
        a.gx = (const unsigned char*)gate->x; a.gmean = gate->mean; a.ginv = gate->invstd; a.gscale = gate->scale; a.gshift = gate->shift;
        a.gmask = (const unsigned char*)gate->mask;
        a.stats = gate->partial;
        if (gate->x2) {
            LH_REQUIRE(gate->mask && gate->mean2 && gate->invstd2 && gate->partial2, "lh_igemm_gated: a second BatchNorm term (x2) needs mask, mean2, invstd2, partial2");
            a.gx2 = (const unsigned char*)gate->x2; a.gmean2 = gate->mean2; a.ginv2 = gate->invstd2; a.stats2 = gate->partial2;
        }
    }
    a.n = d->n; a.hi = d->hi; a.wi = d->wi; a.in_pix_stride = d->in_pix_stride; a.k_run = d->k_run;
    a.kspt = (d->k_run * es + 63) / 64;
    a.kpad = (d->k_run * es + 127) / 128 * (128 / es);
    a.ho = d->ho; a.wo = d->wo; a.M = d->n * d->ho * d->wo; a.sh = d->sh; a.sw = d->sw; a.cout = d->cout;
    a.OH = d->OH; a.OW = d->OW; a.osh = d->osh; a.osw = d->osw; a.ooh = d->ooh; a.oow = d->oow;
    a.out_pix_stride = d->out_pix_stride; a.ntaps = d->ntaps; a.relu = d->relu;
    for (int i = 0; i < 64; ++i) { a.dh[i] = d->dh[i]; a.dw[i] = d->dw[i]; }
    int bm, bp;
    RingCfg rc_;
    const bool ring = lh_ring_supported(d, dtype);
    if (ring) {
        const int rc = lh_ring_resolve(d, dtype, &rc_);
        if (rc) return rc;
        bm = rc_.bm; bp = rc_.bp;
    } else {
        pick_tile(d, dtype, &bm, &bp);
    }
    hipStream_t s = (hipStream_t)stream;
    a.tw = 1; a.dh0 = a.dhs = a.dw0 = a.dws = 0;
    a.xcd = 1;
    a.nphase = 1; a.phase_blocks = 0;
    LH_REQUIRE(!(ring && (rc_.depth == 1 || rc_.depth == 100) && (phases || head)), "lh_igemm_phases: the persistent kernels take single launches only");
    if (phases) {
        LH_REQUIRE(ring, "lh_igemm_phases: the form is not supported by the LDS-DMA kernel");
        a.nphase = phases->n;
        a.phase_blocks = ceil_div(a.M, bp) * ceil_div(a.cout, bm);
        for (int i = 0; i < phases->n; ++i) {
            const lh_igemm_desc* q = phases->descs[i];
            a.ph_w[i] = (const unsigned char*)phases->wpacks[i];
            a.ph_ntaps[i] = q->ntaps; a.ph_tw[i] = 1; a.ph_dh0[i] = a.ph_dhs[i] = a.ph_dw0[i] = a.ph_dws[i] = 0;
            if (q->ntaps > 0)
                LH_REQUIRE(lh_tap_grid(q, &a.ph_tw[i], &a.ph_dh0[i], &a.ph_dhs[i], &a.ph_dw0[i], &a.ph_dws[i]) && a.ph_w[i],
                           "lh_igemm_phases: phase %d has an irregular tap list or no weights", i);
            a.ph_ooh[i] = q->ooh; a.ph_oow[i] = q->oow;
            a.ph_row0[i] = i * ceil_div(a.M, bp);
        }
    }
    if (head) {
        LH_REQUIRE(head->w && head->out && head->n_out >= 1 && head->n_out <= 32 && head->w_row_bytes >= d->cout * es,
                   "lh_igemm_phases_head: bad head (1..32 output channels, K-major weight rows)");
        LH_REQUIRE(ring && es == 2 && bm == 256 && bp == 256 && d->cout <= 256 && !addend && !stats,
                   "lh_igemm_phases_head: needs the 256 x 256 tile of the LDS-DMA kernel (cfg), a 16-bit type, <= 256 channels, no addend / statistics");
        a.head_w = (const unsigned char*)head->w; a.head_bias = head->bias; a.head_out = head->out;
        a.head_j = head->n_out; a.head_wstride = (int)head->w_row_bytes;
    }
    if (prep_args) {            // lh_igemm_multi: hand the argument block back instead of launching
        LH_REQUIRE(ring && ((rc_.depth >= 2 && rc_.depth < LH_WIDE_DEPTH) || rc_.depth == 100),
                   "lh_igemm_multi: problem does not run on the LDS-DMA ring kernel or the direct 3x3 kernel (its cfg must name a tiled or the direct configuration)");
        if (rc_.depth == 100) {                     // a lone direct 3x3 problem handed to lh_igemm_multi: its persistent grid
            a.pw_cb = (d->cout + 63) / 64;
            a.pw_g = lh_d3_rows(d);
        } else {
            lh_tap_grid(d, &a.tw, &a.dh0, &a.dhs, &a.dw0, &a.dws);
            a.kspt = (d->k_run * es + rc_.kb - 1) / rc_.kb;
        }
        *prep_args = a;
        *prep_cfg = rc_;
        return LH_OK;
    }
    if (gate && !(ring && (rc_.depth == 1 || rc_.depth == 100 || (rc_.depth >= 2 && rc_.depth < 10) || (rc_.depth >= LH_DENSE_DEPTH && rc_.depth < LH_KSPLIT_DEPTH + 10)) && es == 2)) {
        lh_set_error("lh_igemm_gated: the launch runs neither on a tiled LDS-DMA configuration nor on a persistent kernel (ring depth %d)", ring ? rc_.depth : 0);
        return LH_ERR_UNSUPPORTED;
    }
    if (gate && gate->x2 && !(ring && rc_.depth == 1)) {
        lh_set_error("lh_igemm_gated: two BatchNorm terms are gated by the pointwise kernel only (ring depth %d)", ring ? rc_.depth : 0);
        return LH_ERR_UNSUPPORTED;
    }
    if (ring) {
        lh_tap_grid(d, &a.tw, &a.dh0, &a.dhs, &a.dw0, &a.dws);
        a.kspt = (d->k_run * es + rc_.kb - 1) / rc_.kb;
        return lh_igemm_ring_launch(a, rc_, dtype, s);
    }
#define LH_TILE(T)                                                               \
    if (bm == 128 && bp == 128) return launch_tile<T, 128, 128, 2, 2>(a, s);     \
    if (bm == 128 && bp == 64) return launch_tile<T, 128, 64, 4, 1>(a, s);       \
    if (bm == 64 && bp == 128) return launch_tile<T, 64, 128, 1, 4>(a, s);       \
    return launch_tile<T, 64, 64, 2, 2>(a, s);
    switch (dtype) {
        case LH_BF16: { LH_TILE(bf16) }
        case LH_F16: { LH_TILE(f16) }
        case LH_F32: {
            if (bm == 128) return launch_tile<float, 128, 64, 4, 1>(a, s);
            if (bp == 128) return launch_tile<float, 64, 128, 1, 4>(a, s);
            return launch_tile<float, 64, 64, 2, 2>(a, s);
        }
    }
#undef LH_TILE
    lh_set_error("lh_igemm: unsupported dtype %d", dtype);
    return LH_ERR_ARG;
}

extern "C" int lh_igemm(const lh_igemm_desc* d, const void* in, const void* wpack, void* out,
                        const void* addend, const void* addend_mask, const float* bias, const float* scale, const float* shift,
                        float* stats, int dtype, void* stream) {
    return igemm_impl(d, in, wpack, out, addend, addend_mask, bias, scale, shift, stats, dtype, stream);
}

extern "C" int lh_igemm_gated(const lh_igemm_desc* d, const void* in, const void* wpack, void* out, const void* addend, const void* addend_mask,
                              const lh_bn_bwd_gate* gate, int dtype, void* stream) {
    LH_REQUIRE(gate, "lh_igemm_gated: null gate");
    return igemm_impl(d, in, wpack, out, addend, addend_mask, nullptr, nullptr, nullptr, nullptr, dtype, stream, nullptr, nullptr, nullptr, nullptr, gate);
}

// n independent convolutions that share ONE kernel configuration (every descriptor's cfg names the same tiled
// configuration of the LDS-DMA kernel) as one grid: the same layer position of HRNet's parallel branches
// (pose_hrnet.py:139-185).  Groups of LH_MULTI_MAX problems per launch.
extern "C" int lh_igemm_multi(const lh_igemm_call* calls, int n, int dtype, void* stream) {
    LH_REQUIRE(calls && n >= 1, "lh_igemm_multi: bad arguments");
    for (int i0 = 0; i0 < n; i0 += LH_MULTI_MAX) {
        const int cnt = n - i0 < LH_MULTI_MAX ? n - i0 : LH_MULTI_MAX;
        LhMulti<IgemmArgs> m;
        IgemmArgs tmp[LH_MULTI_MAX];
        RingCfg cfgs[LH_MULTI_MAX];
        RingCfg cfg0 = {0, 0, 0, 0};                 // the tiled configuration of the call (its ring members share it)
        int ndirect = 0;
        m.n = cnt; m.first[0] = 0;
        for (int i = 0; i < cnt; ++i) {
            const lh_igemm_call& q = calls[i0 + i];
            const int rc = igemm_impl(q.d, q.in, q.wpack, q.out, q.addend, q.addend_mask, q.bias, q.scale, q.shift, q.stats, dtype, stream,
                                      nullptr, nullptr, &tmp[i], &cfgs[i]);
            if (rc) return rc;
            const RingCfg& c = cfgs[i];
            if (c.depth == 100) { ++ndirect; continue; }
            if (cfg0.bm == 0) cfg0 = c;
            LH_REQUIRE(c.bm == cfg0.bm && c.bp == cfg0.bp && c.depth == cfg0.depth && c.kb == cfg0.kb,
                       "lh_igemm_multi: problem %d runs tile %dx%d depth %d kb %d, another %dx%d depth %d kb %d -- one tiled configuration per call",
                       i0 + i, c.bm, c.bp, c.depth, c.kb, cfg0.bm, cfg0.bp, cfg0.depth, cfg0.kb);
        }
        // longest K loop first: workgroups are dispatched in grid order, so the problem whose tiles take longest must not
        // be the one that starts last (the launch ends with its last tile); the direct members' workgroups (a few tiles each,
        // no dependent stages) come last
        int order[LH_MULTI_MAX];
        for (int i = 0; i < cnt; ++i) order[i] = i;
        std::stable_sort(order, order + cnt, [&](int x, int y) {
            const int kx = cfgs[x].depth == 100 ? -1 : tmp[x].ntaps * tmp[x].kspt, ky = cfgs[y].depth == 100 ? -1 : tmp[y].ntaps * tmp[y].kspt;
            return kx > ky;
        });
        if (ndirect == 0) {
            for (int i = 0; i < cnt; ++i) {
                m.a[i] = tmp[order[i]];
                m.first[i + 1] = m.first[i] + ceil_div(m.a[i].M, cfg0.bp) * ceil_div(m.a[i].cout, cfg0.bm);
            }
            const int rc = cnt == 1 ? lh_igemm_ring_launch(m.a[0], cfg0, dtype, (hipStream_t)stream)
                                    : lh_igemm_ring_multi_launch(m, cfg0, dtype, (hipStream_t)stream);
            if (rc) return rc;
            continue;
        }
        if (cnt == 1) {                              // a lone direct problem: its own kernel
            const int rc = lh_igemm_ring_launch(tmp[0], cfgs[0], dtype, (hipStream_t)stream);
            if (rc) return rc;
            continue;
        }
        // (round 4's mixed launch -- direct 3x3 bodies beside ring tiles in one grid -- was measured slower and removed in round 6)
        lh_set_error("lh_igemm_multi: direct 3x3 problems do not share a launch with tiled ones (launch them one by one)");
        return LH_ERR_UNSUPPORTED;
    }
    return LH_OK;
}

// ---- phase batching ------------------------------------------------------------------------------------------------
static int phase_lead(const lh_igemm_desc* const* descs, int nphase) {
    int lead = 0;
    for (int i = 1; i < nphase; ++i)
        if (descs[i]->ntaps > descs[lead]->ntaps) lead = i;
    return lead;
}

static bool phases_ok(const lh_igemm_desc* const* descs, int nphase) {
    if (!descs || nphase < 2 || nphase > 4) return false;
    const lh_igemm_desc* a = descs[0];
    for (int i = 0; i < nphase; ++i) {
        const lh_igemm_desc* b = descs[i];
        if (!b || b->n != a->n || b->hi != a->hi || b->wi != a->wi || b->in_pix_stride != a->in_pix_stride || b->k_run != a->k_run ||
            b->ho != a->ho || b->wo != a->wo || b->sh != a->sh || b->sw != a->sw || b->cout != a->cout || b->OH != a->OH ||
            b->OW != a->OW || b->osh != a->osh || b->osw != a->osw || b->out_pix_stride != a->out_pix_stride || b->relu != a->relu)
            return false;
    }
    return true;
}

extern "C" int lh_igemm_phases_rows(const lh_igemm_desc* const* descs, int nphase, int dtype) {
    if (!phases_ok(descs, nphase)) return -1;
    const lh_igemm_desc* d = descs[phase_lead(descs, nphase)];
    if (!lh_ring_supported(d, dtype)) return -1;
    return lh_igemm_stats_rows(d, dtype);
}

extern "C" int lh_igemm_phases(const lh_igemm_desc* const* descs, int nphase, const void* in, const void* const* wpacks,
                               void* out, const void* addend, const void* addend_mask, const float* bias, const float* scale,
                               const float* shift, float* stats, int dtype, void* stream) {
    LH_REQUIRE(phases_ok(descs, nphase) && wpacks, "lh_igemm_phases: 2..4 descriptors that differ only in taps / placement are required");
    const int lead = phase_lead(descs, nphase);
    LH_REQUIRE(descs[lead]->ntaps > 0 && wpacks[lead], "lh_igemm_phases: no phase has taps");
    PhaseSet ps = {descs, wpacks, nphase};
    return igemm_impl(descs[lead], in, wpacks[lead], out, addend, addend_mask, bias, scale, shift, stats, dtype, stream, &ps);
}

// The phases of a transposed convolution + per-channel affine + ReLU + a 1x1 head in ONE launch (include/lighthand_hip.h).
extern "C" int lh_igemm_phases_head(const lh_igemm_desc* const* descs, int nphase, const void* in, const void* const* wpacks,
                                    const float* scale, const float* shift, const lh_head* head, int dtype, void* stream) {
    LH_REQUIRE(phases_ok(descs, nphase) && wpacks && head, "lh_igemm_phases_head: 2..4 descriptors that differ only in taps / placement are required");
    const int lead = phase_lead(descs, nphase);
    LH_REQUIRE(descs[lead]->ntaps > 0 && wpacks[lead], "lh_igemm_phases_head: no phase has taps");
    PhaseSet ps = {descs, wpacks, nphase};
    return igemm_impl(descs[lead], in, wpacks[lead], nullptr, nullptr, nullptr, nullptr, scale, shift, nullptr, dtype, stream, &ps, head);
}

