// Implicit-GEMM gather convolution, LDS-DMA ring version (the production path on gfx950).
//
// Same contract as igemm.hip (see there for the GEMM view, LDS image and epilogue) but the K loop
// is fed by direct-to-LDS loads (global_load_lds_dwordx4): no staging registers, a ring of D
// stages of (BM + BP) rows x 128 bytes of K, ONE raw s_barrier per K step and a COUNTED
// s_waitcnt vmcnt(N) that leaves D-2 later stages in flight across the barrier, so HBM/L2 latency
// is covered by the ring and not by occupancy (one or two workgroups per CU).
//  * the LDS destination of an LDS-DMA is lane-linear (wave base + lane*16) and the texture-address unit handles
//    four lanes per cycle, so CONSECUTIVE LANES FETCH CONSECUTIVE 16-BYTE CHUNKS OF ONE ROW (one cache-line tag per
//    cycle; lanes in 16 different rows cost 4 tag look-ups per cycle and a quarter of the L1 rate).  The LDS image of
//    a 16-row group is therefore row-major, [row][KB/16 slots], and the bank swizzle is a permutation of the slots
//    INSIDE a row applied to the per-lane source: slot s of row r holds chunk s ^ f(r), f(r) = r >> 2 (KB = 64) or
//    r >> 1 (KB = 128), which makes the 16 rows x one chunk of a ds_read_b128 quarter-wave hit all 64 banks once;
//  * rows that fall into the zero padding (or past k_run / past the last pixel) read a 16-byte
//    zero page instead, so every lane issues every load and the vmcnt bookkeeping is exact;
//  * the pixel operand needs 16-byte aligned pixel rows (in_pix_stride * sizeof(T) % 16 == 0); the
//    C_in = 3 stem keeps the register-staged kernel of igemm.hip.
#include "common.h"

#include "igemm_args.h"
#include "igemm_epilogue.h"
#include <stdlib.h>

__device__ __attribute__((aligned(16))) unsigned int lh_zero_page[4] = {0u, 0u, 0u, 0u};

template <typename T> struct MmaR;
template <> struct MmaR<bf16> {
    static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x4& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct MmaR<f16> {
    static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x4& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};
template <> struct MmaR<float> {
    static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x4& c) {
        const f32x4 fa = __builtin_bit_cast(f32x4, a), fb = __builtin_bit_cast(f32x4, b);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[0], fb[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[1], fb[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[2], fb[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[3], fb[3], c, 0, 0, 0);
    }
};

// Debug-only ablation builds (tools/ablate.sh): -DLH_ABL=<bits>  1 = drop the MFMAs, 2 = drop the fragment reads,
// 4 = drop the LDS-DMA loads, 8 = drop the epilogue.  Results are garbage; only the timing is of interest.  Never set in the product build.
#ifndef LH_ABL
#define LH_ABL 0
#endif

typedef __attribute__((address_space(3))) void* lds_void_p;
typedef const __attribute__((address_space(1))) void* gbl_void_p;

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <typename T, int BM, int BP, int WC, int WP, int D, int KB>
__global__ __launch_bounds__(64 * WC * WP) void igemm_ring_kernel(const IgemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int ES = sizeof(T);
    constexpr int EPC = 16 / ES;
    constexpr int KSTEP = KB / ES;                    // elements per K step (KB bytes per row)
    constexpr int TC = BM / WC, TP = BP / WP;
    constexpr int CT = TC / 16, PT = TP / 16;
    constexpr int STAGE = (BM + BP) * KB;
    constexpr int H = KB / 64;                        // LDS-DMA instructions per 16-row group (1 KiB each)
    constexpr int SL = KB / 16;                       // 16-byte slots per row
    constexpr int RPI = 64 / SL;                      // rows one LDS-DMA instruction covers
    constexpr int GB = 16 * KB;                       // bytes of one 16-row group
    constexpr int NWAVE = WC * WP;                    // 4 waves, or 8 for the 256 x 256 tile
    constexpr int NW = BM / 16 * H / NWAVE, NX = BP / 16 * H / NWAVE;   // instructions per wave and stage
    constexpr int L = NW + NX;
    constexpr int KSUB = KB / 64;                     // MFMA K sub-steps per stage
    static_assert((NWAVE == 4 || NWAVE == 8) && D >= 2 && D <= 5 && (KB == 64 || KB == 128) && NW >= 1 && NX >= 1, "bad configuration");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave / WP, wp = wave % WP;
    // 1-D grid; work item w = (pixel tile, channel tile) with the channel tile fastest: the channel tiles of one pixel
    // tile and neighbouring pixel tiles (3x3 halos) run on one XCD at about the same time and share its L2.
    const int CB = (p.cout + BM - 1) / BM;
    int w = p.xcd ? lh_xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    // per-phase quantities (scalars; the kernel argument block itself is never copied)
    const unsigned char* wgt = p.w;
    int ntaps = p.ntaps, tw = p.tw, dh0 = p.dh0, dhs = p.dhs, dw0 = p.dw0, dws = p.dws, ooh = p.ooh, oow = p.oow;
    float* stats = p.stats;
    if (p.nphase > 1) {                               // phase fastest: the phases of one tile read the same input rows
        const int ph = w % p.nphase;
        w /= p.nphase;
        wgt = p.ph_w[ph]; ntaps = p.ph_ntaps[ph]; tw = p.ph_tw[ph];
        dh0 = p.ph_dh0[ph]; dhs = p.ph_dhs[ph]; dw0 = p.ph_dw0[ph]; dws = p.ph_dws[ph];
        ooh = p.ph_ooh[ph]; oow = p.ph_oow[ph];
        if (stats) stats += (long)p.ph_row0[ph] * 2 * p.cout;
    }
    const int pblk = w / CB, cblk = w - pblk * CB;
    const int hw = p.ho * p.wo;

    // ---- per-lane source bookkeeping.  Instruction q = 4*j + wave of a stage fills (16-row group,
    //      half) = (q / H, q % H); this lane supplies chunk c = 4*half + (lane>>4) of row (lane&15)^2c.
    //      Everything that depends on the lane is folded ONCE into a 64-bit byte offset (tap (0,0), k = 0) and a
    //      bit mask of the taps that fall inside the image; per K step only a wave-uniform offset is added.
    long pbase[NX];
    unsigned hmask[NX], wmask[NX];          // bit ti / tj set when tap row ti / column tj stays inside the image
    int xc[NX];
    const int th = ntaps / tw;
#pragma unroll
    for (int j = 0; j < NX; ++j) {
        const int q = NWAVE * j + wave;
        const int g = q / H, lrow = (q % H) * RPI + lane / SL;
        const int c = (lane % SL) ^ ((lrow / (16 / SL)) & (SL - 1));
        const int row = g * 16 + lrow;
        const int m = pblk * BP + row;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int n = mm / hw, rem = mm - n * hw;
        const int a = rem / p.wo, b = rem - a * p.wo;
        const int ih0 = a * p.sh, iw0 = b * p.sw;
        xc[j] = c * EPC;
        pbase[j] = ((long)(n * p.hi * p.wi + ih0 * p.wi + iw0) * p.in_pix_stride + xc[j]) * ES;
        unsigned hm = 0, wm = 0;
        for (int ti = 0, dh = dh0; ti < th; ++ti, dh += dhs)
            if (ok && (unsigned)(ih0 + dh) < (unsigned)p.hi) hm |= 1u << ti;
        for (int tjj = 0, dw = dw0; tjj < tw; ++tjj, dw += dws)
            if ((unsigned)(iw0 + dw) < (unsigned)p.wi) wm |= 1u << tjj;
        hmask[j] = hm;
        wmask[j] = wm;
    }
    const long kpad = p.kpad;
    const unsigned char* wsrc[NW];
#pragma unroll
    for (int j = 0; j < NW; ++j) {
        const int q = NWAVE * j + wave;
        const int g = q / H, lrow = (q % H) * RPI + lane / SL;
        const int c = (lane % SL) ^ ((lrow / (16 / SL)) & (SL - 1));
        const int row = g * 16 + lrow;
        wsrc[j] = wgt + ((long)(cblk * BM + row) * ntaps * kpad + c * EPC) * ES;
    }
    const unsigned char* zero = reinterpret_cast<const unsigned char*>(lh_zero_page);
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const bool ktail = (p.k_run % KSTEP) != 0;           // only then a chunk can lie past k_run

    // stage index -> (tap, kc) is tracked incrementally; `woff` is the byte offset of the stage inside a weight
    // row (stages are contiguous there), `toff` the activation byte offset of the tap + K step.
    int itap = 0, ikc = 0, tj = 0, ti = 0, cdh = dh0, cdw = dw0;
    unsigned issued = 0;
    long woff = 0;
    auto issue = [&]() {
        unsigned char* st = smem + (issued % D) * STAGE;
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            const int q = NWAVE * j + wave;
            if (!(LH_ABL & 4))
                __builtin_amdgcn_global_load_lds((gbl_void_p)(wsrc[j] + woff), (lds_void_p)(st + (q / H) * GB + (q % H) * 1024), 16, 0, 0);
        }
        const long toff = ((long)(cdh * p.wi + cdw) * p.in_pix_stride + ikc * KSTEP) * ES;
        const int kbase = ikc * KSTEP;
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            const int q = NWAVE * j + wave;
            bool ok = (((hmask[j] >> ti) & (wmask[j] >> tj)) & 1u) != 0;
            if (ktail) ok = ok && (kbase + xc[j] < p.k_run);
            const unsigned char* src = ok ? p.in + pbase[j] + toff : zero;     // select: every lane issues the load
            if (!(LH_ABL & 4))
                __builtin_amdgcn_global_load_lds((gbl_void_p)src, (lds_void_p)(st + BM * KB + (q / H) * GB + (q % H) * 1024), 16, 0, 0);
        }
        ++issued;
        woff += KB;
        if (++ikc == p.kspt) {
            ikc = 0; ++itap;
            woff = (long)itap * kpad * ES;
            cdw += dws;
            if (++tj == tw) { tj = 0; ++ti; cdw = dw0; cdh += dhs; }
        }
    };

    f32x4 acc[CT][PT];
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
        for (int j = 0; j < PT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int S = ntaps * p.kspt;
#pragma unroll
    for (int s = 0; s < D - 1; ++s)
        if ((int)issued < S) issue();
    // fragment read offsets: chunk c = 4*kk + (lane>>4), row = lane&15
    int foff[KSUB];
#pragma unroll
    for (int kk = 0; kk < KSUB; ++kk) {
        const int c = 4 * kk + (lane >> 4), r = lane & 15;
        foff[kk] = r * KB + ((c ^ ((r / (16 / SL)) & (SL - 1))) << 4);
    }

    for (int s = 0; s < S; ++s) {
        // stage s must have landed; stages s+1 .. issued-1 may stay in flight.  In the steady state (a stage is
        // issued every iteration) that is always D-2 stages: one constant wait, no branches.
        const bool steady = (int)issued < S;
        if (steady) {
            wait_vmcnt<(D - 2) * L>();
        } else {
            const int ahead = S - 1 - s;          // <= D - 2 here
            if (D > 4 && ahead >= 3) wait_vmcnt<3 * L>();
            else if (D > 3 && ahead >= 2) wait_vmcnt<2 * L>();
            else if (D > 2 && ahead == 1) wait_vmcnt<L>();
            else wait_vmcnt<0>();
        }
        __builtin_amdgcn_s_barrier();
        if (steady) issue();
        // Fragment reads are inline asm: the compiler cannot tell LDS-DMA writes from these reads
        // and would otherwise drain the whole ring (s_waitcnt vmcnt(0)) in front of every ds_read.
        const unsigned st = lds_base + (s % D) * STAGE;
        uint4 fa[KSUB][CT], fb[KSUB][PT];
#pragma unroll
        for (int kk = 0; kk < KSUB; ++kk) {
#pragma unroll
            for (int i = 0; i < CT; ++i) {
                if (LH_ABL & 2) fa[kk][i] = uint4{st, st, st, st};
                else asm volatile("ds_read_b128 %0, %1" : "=v"(fa[kk][i]) : "v"(st + (wc * CT + i) * GB + foff[kk]));
            }
#pragma unroll
            for (int j = 0; j < PT; ++j) {
                if (LH_ABL & 2) fb[kk][j] = uint4{st, st, st, st};
                else asm volatile("ds_read_b128 %0, %1" : "=v"(fb[kk][j]) : "v"(st + BM * KB + (wp * PT + j) * GB + foff[kk]));
            }
        }
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"((KSUB - 1) * (CT + PT)) : "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < CT; ++i)
#pragma unroll
            for (int j = 0; j < PT; ++j) {
                if (!(LH_ABL & 1)) MmaR<T>::run(fa[0][i], fb[0][j], acc[i][j]);     // (the volatile reads above stay)
            }
        if constexpr (KSUB == 2) {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < CT; ++i)
#pragma unroll
                for (int j = 0; j < PT; ++j) MmaR<T>::run(fa[KSUB - 1][i], fb[KSUB - 1][j], acc[i][j]);
        }
    }

    igemm_epilogue<T, BM, BP, WC, WP>(p, smem, acc, pblk, cblk, tid, lane, wc, wp, hw, ooh, oow, stats);
}

// ------------------------------------------------------------------------------------------------
template <typename T, int BM, int BP, int WC, int WP, int D, int KB>
static int launch_ring(const IgemmArgs& a, hipStream_t s) {
    constexpr int ES = sizeof(T);
    constexpr int ring = D * (BM + BP) * KB;
    constexpr int epi = BP * (BM * ES + 8);
    constexpr int lds = ring > epi ? ring : epi;
    static_assert(lds <= 160 * 1024, "LDS budget");
    static bool attr_done = false;
    if (!attr_done) {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_ring_kernel<T, BM, BP, WC, WP, D, KB>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (e != hipSuccess) {
                lh_set_error("igemm_ring: cannot raise dynamic LDS to %d bytes: %s", lds, hipGetErrorString(e));
                return LH_ERR_HIP;
            }
        }
        attr_done = true;
    }
    dim3 grid(ceil_div(a.M, BP) * ceil_div(a.cout, BM) * (a.nphase > 1 ? a.nphase : 1));
    hipLaunchKernelGGL((igemm_ring_kernel<T, BM, BP, WC, WP, D, KB>), grid, dim3(64 * WC * WP), lds, s, a);
    LH_LAUNCH_CHECK("igemm_ring launch");
    return LH_OK;
}

// Tile choice: the largest tile that still gives >= 2 workgroups per CU, else the smallest.
void lh_ring_pick_tile(const lh_igemm_desc* d, int dtype, int* bm, int* bp) {
    const long M = (long)d->n * d->ho * d->wo;
    // 256 x 256 tile: for grids of at least LH_TILE_MIN_256 workgroups (default 256 = one round of one workgroup per CU,
    // 0 = never) and K loops longer than LH_TILE_256_STEPS steps.  Read per call: the parity test flips it in one process.
    const char* e256 = getenv("LH_TILE_MIN_256");
    const char* s256 = getenv("LH_TILE_256_STEPS");
    const int min_256 = e256 ? atoi(e256) : 256, steps_256 = s256 ? atoi(s256) : 4;
    if (min_256 > 0 && dtype != LH_F32 && lh_ring_kb() == 64 && d->cout % 256 == 0 && d->ntaps * ((d->k_run * 2 + 63) / 64) > steps_256 &&
        ((M + 255) / 256) * (d->cout / 256) >= min_256) {
        *bm = 256; *bp = 256;
        return;
    }
    const int cands[5][2] = {{128, 256}, {128, 128}, {128, 64}, {64, 128}, {64, 64}};
    const bool f32 = dtype == LH_F32;
    static int big = -1;
    if (big < 0) big = getenv("LH_NO_BIG_TILE") ? 0 : 1;
    static int min_blocks = 0, min_big = 0;     // smallest grid for which a tile shape is preferred (tuning knobs)
    if (!min_blocks) {
        const char* e1 = getenv("LH_TILE_MIN"); const char* e2 = getenv("LH_TILE_MIN_BIG");
        min_blocks = e1 ? atoi(e1) : 512;
        min_big = e2 ? atoi(e2) : 1024;
    }
    for (int i = 0; i < 5; ++i) {
        const int BM = cands[i][0], BP = cands[i][1];
        if (BM == 128 && d->cout <= 64) continue;
        if (BP == 256 && (f32 || !big || lh_ring_kb() != 64 || d->ntaps * ((d->k_run * 2 + 63) / 64) <= 4)) continue;   // 16-bit, deep K only
        if (f32 && BM == 128 && BP == 128) continue;            // fp32 epilogue tile would not fit 64 KiB well
        const long blocks = ((M + BP - 1) / BP) * ((d->cout + BM - 1) / BM);
        if (blocks >= (BP == 256 ? min_big : min_blocks) || i == 4) { *bm = BM; *bp = BP; return; }
    }
    *bm = 64; *bp = 64;
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
    if (d->ntaps <= 0 || !lh_tap_grid(d, &tw, &dh0, &dhs, &dw0, &dws)) return false;
    const long ps = (long)d->in_pix_stride * es;
    if (ps % 16 == 0) return true;
    if (getenv("LH_NO_STEM_RING")) return false;
    return (ps * d->sw) % 16 == 0 && (ps * d->wi) % 16 == 0 && (ps * dw0) % 16 == 0 && (ps * dws) % 16 == 0;
}

int lh_ring_kb() {
    static int kb = 0;
    if (!kb) {
        const char* e = getenv("LH_RING_KB");
        kb = (e && atoi(e) == 128) ? 128 : 64;
    }
    return kb;
}

static bool lh_ring_five(const IgemmArgs& a) {
    static int thr = -1;
    if (thr < 0) { const char* e = getenv("LH_RING_T5"); thr = e ? atoi(e) : 0; }      // 0 = never
    return thr > 0 && a.ntaps * a.kspt >= thr;
}

template <typename T, int KB, int D>
static int ring_dispatch(const IgemmArgs& a, int bm, int bp, hipStream_t s) {
    if (bm == 256 && bp == 256) {
        // 256 x 256 tile, 8 waves of 128 x 64 (32 MFMAs per K step each), one workgroup per CU: half the operand bytes per
        // FLOP of the 128 x 128 tile.  3-stage ring = 96 KiB; the epilogue tile (133 KiB) sets the LDS size.
        if constexpr (sizeof(T) == 2 && KB == 64) return launch_ring<T, 256, 256, 2, 4, 3, KB>(a, s);
        else { lh_set_error("igemm_ring: 256x256 tile is 16-bit / 64-byte-step only"); return LH_ERR_UNSUPPORTED; }
    }
    if (bm == 128 && bp == 256) {
        // 128 x 256 tile: each wave owns 64 x 128 (32 MFMAs per K step), 3-stage ring = 72 KiB -> two workgroups per CU.
        // The tile choice (and the stats-slab row count derived from it) must be honoured whatever depth was asked for.
        if constexpr (sizeof(T) == 2 && KB == 64) return launch_ring<T, 128, 256, 2, 2, 3, KB>(a, s);
        else { lh_set_error("igemm_ring: 128x256 tile is 16-bit / 64-byte-step only"); return LH_ERR_UNSUPPORTED; }
    }
    if (bm == 128 && bp == 128) {
        if constexpr (sizeof(T) == 4) return launch_ring<T, 128, 64, 4, 1, D, KB>(a, s);
        else {
            // deep K: a 5-stage ring (2 x 80 KiB = the whole LDS of a CU) keeps 128 KiB of loads in flight per CU
            if constexpr (D == 4 && KB == 64) { if (lh_ring_five(a)) return launch_ring<T, 128, 128, 2, 2, 5, KB>(a, s); }
            return launch_ring<T, 128, 128, 2, 2, D, KB>(a, s);
        }
    }
    if (bm == 128 && bp == 64) return launch_ring<T, 128, 64, 4, 1, D, KB>(a, s);
    if (bm == 64 && bp == 128) return launch_ring<T, 64, 128, 1, 4, D, KB>(a, s);
    return launch_ring<T, 64, 64, 2, 2, D, KB>(a, s);
}

// K loops of <= 4 steps (1x1 convolutions on 64..128 channels) are store-bound: a 2-stage ring keeps the
// LDS footprint at the epilogue tile's size so 4 workgroups share a CU instead of 2.
int lh_ring_depth(const IgemmArgs& a) {
    static int forced = -1;
    if (forced < 0) { const char* e = getenv("LH_RING_D"); forced = e ? atoi(e) : 0; }
    if (forced >= 2 && forced <= 4) return forced;
    static int t2 = -1, t3 = -1;
    if (t2 < 0) { const char* e = getenv("LH_RING_T2"); t2 = e ? atoi(e) : 4; e = getenv("LH_RING_T3"); t3 = e ? atoi(e) : 0; }
    const int steps = a.ntaps * a.kspt;
    return steps <= t2 ? 2 : steps <= t3 ? 3 : 4;
}

int lh_igemm_ring_launch(const IgemmArgs& a, int bm, int bp, int dtype, hipStream_t s) {
    const int kb = lh_ring_kb();
    switch (dtype) {
        case LH_BF16: {
            const int dep = lh_ring_depth(a);
            if (kb == 128) return dep == 2 ? ring_dispatch<bf16, 128, 2>(a, bm, bp, s) : dep == 3 ? ring_dispatch<bf16, 128, 3>(a, bm, bp, s) : ring_dispatch<bf16, 128, 4>(a, bm, bp, s);
            return dep == 2 ? ring_dispatch<bf16, 64, 2>(a, bm, bp, s) : dep == 3 ? ring_dispatch<bf16, 64, 3>(a, bm, bp, s) : ring_dispatch<bf16, 64, 4>(a, bm, bp, s);
        }
        case LH_F16:
            if (kb == 128) return ring_dispatch<f16, 128, 4>(a, bm, bp, s);
            return lh_ring_depth(a) == 2 ? ring_dispatch<f16, 64, 2>(a, bm, bp, s) : ring_dispatch<f16, 64, 4>(a, bm, bp, s);
        case LH_F32:
            if (kb == 128) return ring_dispatch<float, 128, 4>(a, bm, bp, s);
            return lh_ring_depth(a) == 2 ? ring_dispatch<float, 64, 2>(a, bm, bp, s) : ring_dispatch<float, 64, 4>(a, bm, bp, s);
    }
    lh_set_error("igemm_ring: unsupported dtype %d", dtype);
    return LH_ERR_ARG;
}
