// Weight-gradient GEMM, LDS-DMA ring kernel (16-bit types; the production path on gfx950):
//
//   dW[tap][o][i] = sum_pixels dy[pixel][o] * x[pix(pixel, tap)][i]
//
// The contraction index (pixels) is the SLOW index of both NHWC operands, so the MFMA fragments (8 consecutive k per
// lane) are formed with the CDNA4 transposing LDS read ds_read_b64_tr_b16 from row-major [pixel][channel] LDS images.
//  * tile BO x BI of one tap, a stage = KPS pixel rows of both operands (KPS / 32 logical steps of one
//    v_mfma_f32_16x16x32 K slice each), ring of D stages filled by global_load_lds_dwordx4, ONE raw s_barrier per
//    stage and a counted s_waitcnt vmcnt that leaves D - 2 stages in flight (protocol of igemm_ring_kernel.h);
//  * LDS image per operand: [KPS pixel rows][B * 2 bytes], unpadded (an LDS-DMA writes 1 KiB lane-linear), 32-byte
//    granules XOR-swizzled by the row so the 8 pixel rows a half-wave touches per transposed read hit distinct banks;
//    the swizzle is applied to the per-lane SOURCE address and to the read address;
//  * the pixel a lane fetches advances by KPS per stage: its (image, row, column) is carried in registers and advanced
//    by a mixed-radix add (no division in the loop); rows past the split's end or outside the image read a zero page;
//  * fragment reads are retired by a ladder of counted s_waitcnt lgkmcnt: the MFMAs of output-channel tile i start as
//    soon as its two reads are back;
//  * split-K over pixels -> fp32 slabs [split][tap][o][i] (16-byte stores: the MFMA operands are swapped so that a lane
//    holds four consecutive input channels), folded deterministically, in split order, into the reference-layout
//    gradient by wgrad_reduce_kernel (wgrad.hip).  A fold INSIDE this launch by the last-arriving workgroup of every
//    (tile, tap) -- write-through partial tiles, agent-scope arrival counter, acquire, sum -- was built and measured:
//    the serial tail of one workgroup reading nsplit tiles costs more than the 5-8 us reduce launch it replaces at
//    every split count this network uses, and its registers slowed the large tiles; removed.
#pragma once
// Debug builds only: cache-policy A/B of the operand streams (aux = 2 selects the non-temporal form of the LDS-DMA load)
#ifndef LH_NT_WGRAD_X
#define LH_NT_WGRAD_X 0
#endif
#ifndef LH_NT_WGRAD_DY
#define LH_NT_WGRAD_DY 0
#endif
#include "common.h"
#include "multi.h"

#include <type_traits>

#ifndef LH_ABL      // debug-only ablation builds, see igemm_ring_kernel.h / tools/ablate.sh
#define LH_ABL 0
#endif

struct WgradArgs {
    const unsigned char* x;
    const unsigned char* dy;
    const unsigned char* zero;   // 16 zero bytes in device memory (what masked lanes fetch)
    float* slab;
    int n, hi, wi, in_pix_stride, k_run;
    int ho, wo, M, sh, sw;
    int dy_pix_stride, n_out, n_in;
    int ntaps, nsplit, steps_per_split, i_tiles;   // steps_per_split: ring stages (ring kernel) / K steps (wgrad_kernel)
    int tiles, xcd;              // ring kernel: 1-D grid of tiles * ntaps * nsplit work items, XCD-aware order
    int fold_k;                  // ring kernel, row fold (lh_wgrad_rowfold): input index i = row * fold_k + k, 0 = off
    int adv_n, adv_a, adv_b;     // ring kernel: one stage of KPS pixels = adv_n images + adv_a rows + adv_b columns
    signed char dh[64];
    signed char dw[64];
};

template <int I> using wic = std::integral_constant<int, I>;
template <int B, int E, typename F> __device__ __forceinline__ void wstatic_for(F&& f) {
    if constexpr (B < E) {
        f(wic<B>{});
        wstatic_for<B + 1, E>(f);
    }
}

template <int L, int MAXS> __device__ __forceinline__ void wwait_stages(int stages) {
    static_assert(MAXS * L <= 63, "vmcnt is a 6-bit counter");
    if constexpr (MAXS == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        if (stages >= MAXS) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MAXS * L) : "memory");
        else wwait_stages<L, MAXS - 1>(stages);
    }
}

template <int ROWB> __device__ __forceinline__ constexpr int wswz(int row) {
    return ROWB >= 256 ? (row & 7) : ((row >> 1) & 3);
}

template <typename T> struct WMma;
template <> struct WMma<bf16> {
    static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x4& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct WMma<f16> {
    static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x4& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};

// The kernel proper for workgroup `bid` of `nblk` of ONE problem (plain launch: the block index; multi-problem launch,
// multi.h: the index inside the problem the workgroup belongs to).
// tap_dh / tap_dw: the tap offset tables (the argument block's own arrays, or -- table launches, whose block is a register copy of
// a table entry -- the entry's arrays in device memory: a dynamically indexed array inside a local copy would live in scratch).
template <typename T, int BO, int BI, int WO, int WI, int D, int KPS>
__device__ __forceinline__ void wgrad_ring_body(const WgradArgs& p, const signed char* tap_dh, const signed char* tap_dw, unsigned char* smem,
                                                const int bid, const int nblk) {
    static_assert(sizeof(T) == 2, "16-bit types only");
    constexpr int KP = 32;                                    // pixels per logical step (one MFMA K slice)
    constexpr int KSUB = KPS / KP;
    constexpr int RBO = BO * 2, RBI = BI * 2;                 // bytes per pixel row
    constexpr int RPO = 1024 / RBO, RPI = 1024 / RBI;         // pixel rows per LDS-DMA instruction
    constexpr int NWAVE = WO * WI;                            // 4 waves, or 8 for the 256 x 256 tile
    constexpr int NO = KPS / RPO / NWAVE, NI = KPS / RPI / NWAVE;   // instructions per wave and stage
    constexpr int L = NO + NI;
    constexpr int STAGE = KPS * (RBO + RBI);
    constexpr int TO = BO / WO, TI = BI / WI, OT = TO / 16, IT = TI / 16;
    static_assert((NWAVE == 4 || NWAVE == 8) && NO >= 1 && NI >= 1 && D >= 2 && D <= 8 && (KPS == 32 || KPS == 64), "bad tile");
    typedef __attribute__((address_space(3))) void* lds_p;
    typedef const __attribute__((address_space(1))) void* gbl_p;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wo_ = wave / WI, wi_ = wave % WI;
    // work item w = (split, tap, tile), tile fastest: every XCD gets a contiguous range of pixel splits with all their
    // taps and tiles, which re-read the same dy / x rows from that XCD's L2 instead of the Infinity Cache
    const int w = p.xcd ? lh_xcd_remap(bid, nblk) : bid;
    const int tile = w % p.tiles, tap = __builtin_amdgcn_readfirstlane((w / p.tiles) % p.ntaps), split = w / (p.tiles * p.ntaps);
    const int otile = tile / p.i_tiles, itile = tile % p.i_tiles;
    // the tap offsets come out of the argument block through a VECTOR load (dynamic index): consume them here -- left to the
    // compiler their first use lands behind the first LDS-DMA instructions, and the s_waitcnt vmcnt(0) in front of it waits
    // for those too (one memory latency per workgroup, in launches of 10-30 us per workgroup)
    const int dh = __builtin_amdgcn_readfirstlane((int)tap_dh[tap]), dw = __builtin_amdgcn_readfirstlane((int)tap_dw[tap]);
    const int hw = p.ho * p.wo;
    const long m_begin = (long)split * p.steps_per_split * KPS;
    long m_end = m_begin + (long)p.steps_per_split * KPS;
    if (m_end > p.M) m_end = p.M;
    const int S = m_begin < m_end ? (int)((m_end - m_begin + KPS - 1) / KPS) : 0;
    const int span = (int)(m_end - m_begin);                  // pixels of this split (<= 0: nothing to do)
    const unsigned char* zero = p.zero;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;

    // ---- per-lane source bookkeeping: instruction q = NWAVE*j + wave covers rows [q*RP, (q+1)*RP) of the stage.
    // Instruction j of a wave lies j * NWAVE * RP rows below instruction 0, a multiple of 8 rows: same swizzle, same
    // channel chunk, so one address / one limit per operand serves all j (the row offset of j is wave-uniform).
    static_assert((NWAVE * RPO) % 8 == 0 && (NWAVE * RPI) % 8 == 0, "instructions of a wave must share the swizzle phase");
    const int ro0 = wave * RPO + lane / (RBO / 16), oc16 = lane % (RBO / 16);
    const int ocol = otile * BO + ((((oc16 >> 1) ^ wswz<RBO>(ro0)) << 1) | (oc16 & 1)) * 8;
    const unsigned char* osrc = p.dy + ((m_begin + ro0) * p.dy_pix_stride + ocol) * 2;   // dy address of this lane's chunk, instruction 0, stage 0
    const int oleft = ocol < p.n_out ? span - ro0 : 0;       // chunk j is live while (pixels done) + j * NWAVE * RPO < oleft
    const long ostep = (long)NWAVE * RPO * p.dy_pix_stride * 2;
    const int ri0 = wave * RPI + lane / (RBI / 16), ic16 = lane % (RBI / 16);
    const int icol = itile * BI + ((((ic16 >> 1) ^ wswz<RBI>(ri0)) << 1) | (ic16 & 1)) * 8;
    // row fold: the gradient's input index covers `rows` kernel rows of fold_k contiguous elements each; this lane's
    // chunk belongs to kernel row icol / fold_k (an extra input-row offset) and element icol % fold_k of the run
    const int xdh = dh + (p.fold_k ? icol / p.fold_k : 0);   // input row offset of this lane's chunk
    const unsigned char* xsrc = p.x + (long)(p.fold_k ? icol % p.fold_k : icol) * 2;     // (image 0, row 0, column 0) + channel chunk
    const int xleft = icol < p.k_run ? span - ri0 : 0;
    // output pixel (image, row, column) instruction 0 fetches for in the next stage; instruction j's lies j * NWAVE * RPI
    // pixels further on: (jn, ja, jb) is that distance as a mixed-radix number (wave-uniform)
    int xn, xa, xb;
    const int xpix = p.in_pix_stride * 2;   // bytes per input pixel
    {
        const long m = m_begin + ri0;
        const int mi = m < p.M ? (int)m : 0;
        xn = mi / hw;
        const int rem = mi - xn * hw;
        xa = rem / p.wo;
        xb = rem - xa * p.wo;
    }
    const int jq = (NWAVE * RPI) / p.wo, jb = NWAVE * RPI - jq * p.wo, jn = jq / p.ho, ja = jq - jn * p.ho;

    int issued = 0, islot = 0, done = 0;    // done = pixels of the split covered by the stages issued so far
    long oadv = 0;
    auto issue = [&]() __attribute__((always_inline)) {
        unsigned char* st = smem + islot * STAGE;
#pragma unroll
        for (int j = 0; j < NO; ++j) {
            const unsigned char* src = done + j * NWAVE * RPO < oleft ? osrc + (oadv + j * ostep) : zero;
            if (!(LH_ABL & 4)) __builtin_amdgcn_global_load_lds((gbl_p)src, (lds_p)(st + (NWAVE * j + wave) * 1024), 16, 0, LH_NT_WGRAD_DY ? 2 : 0);
        }
        int cn = xn, ca = xa, cb = xb;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int ih = ca * p.sh + xdh, iw = cb * p.sw + dw;
            const bool ok = (int)(done + j * NWAVE * RPI < xleft) & (int)((unsigned)ih < (unsigned)p.hi) & (int)((unsigned)iw < (unsigned)p.wi);
            const int pix = (cn * p.hi + ih) * p.wi + iw;               // input pixel index (any value when !ok)
            const unsigned char* src = ok ? xsrc + (long)pix * xpix : zero;
            if (!(LH_ABL & 4)) __builtin_amdgcn_global_load_lds((gbl_p)src, (lds_p)(st + KPS * RBO + (NWAVE * j + wave) * 1024), 16, 0, LH_NT_WGRAD_X ? 2 : 0);
            if (j + 1 < NI) {                                           // on to instruction j + 1 (mixed-radix add: no division)
                cb += jb; ca += ja; cn += jn;
                if (cb >= p.wo) { cb -= p.wo; ++ca; }
                if (ca >= p.ho) { ca -= p.ho; ++cn; }
            }
        }
        {                                                               // this lane's pixel one stage on
            int b = xb + p.adv_b, a = xa + p.adv_a, n = xn + p.adv_n;
            if (b >= p.wo) { b -= p.wo; ++a; }
            if (a >= p.ho) { a -= p.ho; ++n; }
            xb = b; xa = a; xn = n;
        }
        ++issued;
        if (++islot == D) islot = 0;
        done += KPS;
        oadv += (long)KPS * p.dy_pix_stride * 2;
    };

    f32x4 acc[OT][IT];
#pragma unroll
    for (int i = 0; i < OT; ++i)
#pragma unroll
        for (int j = 0; j < IT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int s = 0; s < D - 1; ++s)
        if (issued < S) issue();

    // transposed-read addresses: lane (group g, q, pp) reads pixel row 4g+q (and +16), 4 channels at 4*pp of a 16-channel tile
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int row0 = 4 * g + q;
    // (the second read of a fragment, pixel rows +16, has the same swizzle: it is the first address + 16 rows, an immediate)
    unsigned ao[OT], ai[IT];
#pragma unroll
    for (int i = 0; i < OT; ++i) {
        const int ct = wo_ * OT + i;                              // 32-byte granule index inside the row
        ao[i] = lds_base + row0 * RBO + ((ct ^ wswz<RBO>(row0)) << 5) + pp * 8;
    }
#pragma unroll
    for (int j = 0; j < IT; ++j) {
        const int ct = wi_ * IT + j;
        ai[j] = lds_base + KPS * RBO + row0 * RBI + ((ct ^ wswz<RBI>(row0)) << 5) + pp * 8;
    }
    static_assert(wswz<RBO>(3) == wswz<RBO>(19) && wswz<RBI>(5) == wswz<RBI>(21), "the swizzle must repeat every 16 rows");

    auto rdtr = [](auto OFFc, uint2& dst, unsigned addr) __attribute__((always_inline)) {
        if constexpr ((LH_ABL & 2) != 0) { dst = uint2{addr, addr}; return; }
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(decltype(OFFc)::value));
    };
    // one logical step: pixel rows [32*kk, 32*kk + 32) of ring slot SLOT (the swizzle repeats every 8 rows).  Slot and
    // step offsets are compile-time: they ride in the reads' immediate offset fields, the address registers never change
    // (the 8-wave tile's second slot lies beyond the 16-bit immediate range and takes one add per read instead).
    auto step = [&](auto KKc, auto SLOTc) __attribute__((always_inline)) {
        constexpr int kk = decltype(KKc)::value, slot = decltype(SLOTc)::value;
        constexpr bool imm = (D - 1) * STAGE + KPS * RBO + KP * RBI * (KSUB - 1) + 16 * (RBO > RBI ? RBO : RBI) < 65536;
        constexpr int so_imm = imm ? slot * STAGE : 0;
        const unsigned so_reg = imm ? 0u : (unsigned)(slot * STAGE);
        // Output-channel fragments live in OW register slots and ROLL: once the MFMAs of fragment i are issued, its slot
        // takes the read of fragment i + OW (the large tiles would not fit their 8 fragments beside 128 accumulators).
        constexpr int OW = OT > 4 ? 4 : OT;
        uint2 fi[IT][2], fo[OW][2];
        auto rd_o = [&](auto Ic) __attribute__((always_inline)) {
            constexpr int i = decltype(Ic)::value;
            rdtr(wic<so_imm + kk * KP * RBO>{}, fo[i % OW][0], ao[i] + so_reg);
            rdtr(wic<so_imm + kk * KP * RBO + 16 * RBO>{}, fo[i % OW][1], ao[i] + so_reg);
        };
        wstatic_for<0, IT>([&](auto Jc) {
            rdtr(wic<so_imm + kk * KP * RBI>{}, fi[decltype(Jc)::value][0], ai[decltype(Jc)::value] + so_reg);
            rdtr(wic<so_imm + kk * KP * RBI + 16 * RBI>{}, fi[decltype(Jc)::value][1], ai[decltype(Jc)::value] + so_reg);
        });
        wstatic_for<0, OW>(rd_o);
        wstatic_for<0, OT>([&](auto Ic) {
            constexpr int i = decltype(Ic)::value;
            // reads younger than fragment i's may stay in flight (LDS returns in order): the rest of the first OW, plus
            // the rolled reads issued since (one per earlier fragment k with k + OW < OT, issued after k's MFMAs)
            constexpr int first = i < OW ? OW - 1 - i : 0;
            constexpr int lo = i < OW ? 0 : i - OW + 1;
            constexpr int hi = i < OT - OW ? i : OT - OW;                 // rolled reads come from k in [lo, hi)
            constexpr int younger = first + (hi > lo ? hi - lo : 0);
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * younger) : "memory");
            __builtin_amdgcn_sched_barrier(0);
            const uint4 a = uint4{fo[i % OW][0].x, fo[i % OW][0].y, fo[i % OW][1].x, fo[i % OW][1].y};
#pragma unroll
            for (int j = 0; j < IT; ++j) {
                const uint4 b = uint4{fi[j][0].x, fi[j][0].y, fi[j][1].x, fi[j][1].y};
                // D[input channel][output channel]: a lane ends up with 4 CONSECUTIVE input channels of one output
                // channel, i.e. one 16-byte store into the [o][i] slab
                if (!(LH_ABL & 1)) WMma<T>::run(b, a, acc[i][j]);
            }
            if constexpr (i + OW < OT) {
                __builtin_amdgcn_sched_barrier(0);
                rd_o(wic<i + OW>{});
            }
        });
    };

    // waves 4-7 of the 8-wave tile share their SIMDs with waves 0-3: they refill the ring half a stage later
    const bool late = NWAVE == 8 && KSUB == 2 && wave >= NWAVE / 2;
    auto stage = [&](auto SLOTc, int s) __attribute__((always_inline)) {
        wwait_stages<L, D - 2>(issued - 1 - s);          // stage s has landed; later stages stay in flight
        __builtin_amdgcn_s_barrier();
        if (!late && issued < S) issue();                // into the slot of stage s - 1, retired by every wave
        step(wic<0>{}, SLOTc);
        if constexpr (KSUB == 2) {
            if (late && issued < S) issue();
            step(wic<1>{}, SLOTc);
        }
    };
    // D stages per trip, so that the slot of every stage is a compile-time constant
    int s = 0;
    for (; s + D <= S; s += D) wstatic_for<0, D>([&](auto Kc) { stage(Kc, s + decltype(Kc)::value); });
    wstatic_for<0, D - 1>([&](auto Kc) {
        if (s + decltype(Kc)::value < S) stage(Kc, s + decltype(Kc)::value);
    });

    if (LH_ABL & 8) { if (acc[0][0][0] == 123.456f) p.slab[0] = 1.f; return; }
    float* slab = p.slab + ((long)split * p.ntaps + tap) * p.n_out * p.n_in;
    const int qq = lane >> 4, cc = lane & 15;
#pragma unroll
    for (int i = 0; i < OT; ++i) {
        const int o = otile * BO + wo_ * TO + i * 16 + cc;
#pragma unroll
        for (int j = 0; j < IT; ++j) {
            const int ci = itile * BI + wi_ * TI + j * 16 + qq * 4;            // n_in % 8 == 0: whole 16-byte groups
            if (o < p.n_out && ci < p.n_in)
                *reinterpret_cast<float4*>(slab + (long)o * p.n_in + ci) = float4{acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
        }
    }
}

template <typename T, int BO, int BI, int WO, int WI, int D, int KPS>
__global__ __launch_bounds__(64 * WO * WI, 2) void wgrad_ring_kernel(const WgradArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    wgrad_ring_body<T, BO, BI, WO, WI, D, KPS>(p, p.dh, p.dw, smem, blockIdx.x, gridDim.x);
}

// Up to LH_MULTI_MAX independent weight gradients that share the tile / stage / ring depth as ONE grid (lh_wgrad_fused_multi).
template <typename T, int BO, int BI, int WO, int WI, int D, int KPS>
__global__ __launch_bounds__(64 * WO * WI, 2) void wgrad_ring_multi_kernel(const LhMulti<WgradArgs> m) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int bid, nblk;
    const int i = lh_multi_pick(m, bid, nblk);
    wgrad_ring_body<T, BO, BI, WO, WI, D, KPS>(m.a[i], m.a[i].dh, m.a[i].dw, smem, bid, nblk);
}

// Table launch (lh_wgrad_table_run): ANY number of independent weight gradients of one tile / stage / ring depth as ONE grid -- the
// deferred weight gradients of a whole stage (pose_resnet.py:61-99 x the blocks of a layer, 207-232).  The argument blocks live in
// DEVICE memory (built once per plan: every pointer of a plan is static); workgroup b runs work item items[b] = (problem, block index
// inside the problem).  Every problem carries its OWN pixel-split count, so the deep-K layers of the late stages run split-free (their
// tiles alone fill the machine once they share a grid) while the few-tile layers of the early stages keep their splits; items are
// ordered longest first.  The scalar fields of the entry are copied into registers (the stage loop's counted waits are asm statements
// with memory clobbers: fields read through a pointer would be re-fetched behind every one of them).
template <typename T, int BO, int BI, int WO, int WI, int D, int KPS>
__global__ __launch_bounds__(64 * WO * WI, 2) void wgrad_ring_table_kernel(const WgradArgs* __restrict__ tab, const int2* __restrict__ items) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int2 it = items[blockIdx.x];
    const int prob = __builtin_amdgcn_readfirstlane(it.x), bid = __builtin_amdgcn_readfirstlane(it.y);
    if (prob < 0) return;                 // padding of a round of the XCD-aware item order (lh_wgrad_table_build)
    const WgradArgs* g = tab + prob;
    WgradArgs p;
    p.x = g->x; p.dy = g->dy; p.zero = g->zero; p.slab = g->slab;
    p.n = g->n; p.hi = g->hi; p.wi = g->wi; p.in_pix_stride = g->in_pix_stride; p.k_run = g->k_run;
    p.ho = g->ho; p.wo = g->wo; p.M = g->M; p.sh = g->sh; p.sw = g->sw;
    p.dy_pix_stride = g->dy_pix_stride; p.n_out = g->n_out; p.n_in = g->n_in;
    p.ntaps = g->ntaps; p.nsplit = g->nsplit; p.steps_per_split = g->steps_per_split; p.i_tiles = g->i_tiles;
    p.tiles = g->tiles; p.xcd = g->xcd; p.fold_k = g->fold_k; p.adv_n = g->adv_n; p.adv_a = g->adv_a; p.adv_b = g->adv_b;
    wgrad_ring_body<T, BO, BI, WO, WI, D, KPS>(p, g->dh, g->dw, smem, bid, p.tiles * p.ntaps * p.nsplit);
}
