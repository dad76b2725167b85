// Implicit-GEMM gather convolution, LDS-DMA ring kernel (the production path on gfx950): forward convolution, data
// gradient, every sub-pixel phase of the stride-2 transposed forms and the C_in = 3 stem.
//
//   out[pixel][co] = sum_{tap} sum_{k < k_run} in[pix(pixel, tap)][k] * wpack[co][tap][k]
//
// GEMM view D[co][pixel]: MFMA "A" = weight rows, "B" = gathered pixel rows, both K-major.  The K loop is fed by
// direct-to-LDS loads (global_load_lds_dwordx4): no staging registers, a ring of D stages of (BM + BP) rows x KB bytes
// of K, ONE raw s_barrier per stage and a COUNTED s_waitcnt vmcnt(N) that leaves the later stages in flight across the
// barrier, so L2 / HBM latency is covered by the ring and not by occupancy.
//  * the LDS destination of an LDS-DMA is lane-linear (wave base + lane*16) and the texture-address unit handles four
//    lanes per cycle, so CONSECUTIVE LANES FETCH CONSECUTIVE 16-BYTE CHUNKS OF ONE ROW (one cache-line tag per cycle).
//    The LDS image of a 16-row group is row-major, [row][KB/16 slots], and the bank swizzle is a permutation of the
//    slots INSIDE a row applied to the per-lane source: slot s of row r holds chunk s ^ f(r), f(r) = r >> 2 (KB = 64)
//    or r >> 1 (KB = 128), which makes the 16 rows x one chunk of a ds_read_b128 quarter-wave hit all 64 banks once;
//  * rows that fall into the zero padding (or past k_run / past the last pixel) fetch zeros (an out-of-range buffer
//    offset, below), so every lane issues every load and the vmcnt bookkeeping is exact;
//  * fragment pipeline: a logical step (one MFMA K slice of 64 bytes) reads PT pixel fragments then CT weight fragments
//    (ds_read_b128, inline asm: the compiler cannot tell LDS-DMA writes from these reads and would drain the ring in
//    front of every one) and retires them with a LADDER of counted s_waitcnt lgkmcnt: the MFMAs of weight-row tile i
//    start as soon as fragment i is back, the rest of the reads land under them.  (Fetching fragments of the next step
//    ahead across the barrier was built and measured: <= 2 % on any layer, not kept.)
//  * a stage of KB = 128 bytes carries two logical steps per barrier; in the 8-wave tile the two waves of a SIMD issue
//    their LDS-DMA at DIFFERENT points of the stage (waves 0-3 before the first step, waves 4-7 between the two), so
//    one wave's address arithmetic runs under its partner's MFMAs instead of both stalling the matrix pipe together;
//  * BUFFER FORM of the operand path (round 5; tools/ingest_ladder.hip: the per-load vector address work -- tap-mask bit
//    test, K-limit compare, 64-bit add, select against a zero page -- cost 1.3-3.4 us of this loop's 21 us on the stage-3
//    3x3 layer): the loads are `buffer_load_dwordx4 ... offen lds` through ONE descriptor per operand and workgroup.  A
//    lane's 32-bit offset is fixed for the whole launch; everything that changes per stage (tap, K step) is wave-uniform
//    and travels in the SGPR offset, so a load costs NO vector instruction.  A lane whose tap falls outside the image (or
//    whose chunk lies past the K run) carries an out-of-range offset instead: the descriptor's range check makes the
//    load write ZEROS to LDS (probed: tools/probes/buffer_lds_oob.hip; the SGPR offset is part of the check).  The
//    choice in-range / out-of-range is remade once per TAP (per stage only when the K run is not a whole number of
//    stages).  Every lane still issues every load, so the vmcnt bookkeeping stays exact.
#pragma once
#include "common.h"
#include "igemm_args.h"
#include "igemm_epilogue.h"
#include "multi.h"

#include <type_traits>

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
// 4 = drop the LDS-DMA loads, 8 = drop the epilogue (the K loop is pruned with it), 16 = keep the epilogue but drop its
// global stores, 32 = drop the epilogue but keep every accumulator live.  Results are garbage; only the timing is of interest.  Never set in the product build.
#ifndef LH_PREREAD
#define LH_PREREAD 0   // debug builds only: both K slices' fragment reads of a 128-byte stage issued up front (measured neutral)
#endif
#ifndef LH_PRIO
#define LH_PRIO 0      // debug builds only (s_setprio around the MFMA block of a K slice; measured)
#endif
#ifndef LH_ABL
#define LH_ABL 0
#endif

typedef __attribute__((address_space(3))) void* lds_void_p;
typedef const __attribute__((address_space(1))) void* gbl_void_p;

template <int I> using ic = std::integral_constant<int, I>;
template <int B, int E, typename F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(ic<B>{});
        static_for<B + 1, E>(f);
    }
}

// s_waitcnt vmcnt(stages * L): `stages` ring stages (of L loads per wave each) may stay in flight.
template <int L, int MAXS> __device__ __forceinline__ void wait_stages(int stages) {
    static_assert(MAXS * L <= 63, "vmcnt is a 6-bit counter");
    if constexpr (MAXS == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        if (stages >= MAXS) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MAXS * L) : "memory");
        else wait_stages<L, MAXS - 1>(stages);
    }
}

// The kernel proper, for workgroup `bid` of `nblk` of ONE problem: the plain kernel passes its block index, the
// multi-problem kernel (multi.h) the index inside the problem the workgroup belongs to.
// KZ = 2, the K-SPLIT WAVE PAIR (round 6): WC x WP pairs of waves, each pair owning a (BM / WC) x (BP / WP) sub-tile -- 64 x 64 on the
// 128 x 128 tile where the 8-wave form gives a wave 64 x 32 -- and the two waves of a pair taking ALTERNATE K slices of every 128-byte
// stage.  Per stage a wave issues PT + CT = 8 fragment reads for 16 MFMAs instead of 2 x 6 reads for 2 x 8: two thirds of the LDS
// fragment bytes per MFMA at the same occupancy (two waves per SIMD, one workgroup per CU for a 256-tile launch).  The partial sums of
// a pair meet in the epilogue (one fp32 addition per element through LDS), so the accumulation ORDER differs from the KZ = 1 kernels:
// results agree with them to fp32 rounding, not bit for bit.  RingCfg depth = depth + LH_KSPLIT_DEPTH.
template <typename T, int BM, int BP, int WC, int WP, int D, int KB, int KZ = 1>
__device__ __forceinline__ void igemm_ring_body(const IgemmArgs& p, unsigned char* smem, const int bid, const int nblk) {
#if defined(__HIP_DEVICE_COMPILE__)      // the buffer builtins exist in the device pass only
    constexpr int ES = sizeof(T);
    constexpr int EPC = 16 / ES;
    constexpr int KSTEP = KB / ES;                    // elements per ring stage (KB bytes per row)
    constexpr int TC = BM / WC, TP = BP / WP;
    constexpr int CT = TC / 16, PT = TP / 16;
    constexpr int STAGE = (BM + BP) * KB;
    constexpr int H = KB / 64;                        // LDS-DMA instructions per 16-row group (1 KiB each)
    constexpr int SL = KB / 16;                       // 16-byte slots per row
    constexpr int RPI = 64 / SL;                      // rows one LDS-DMA instruction covers
    constexpr int GB = 16 * KB;                       // bytes of one 16-row group
    constexpr int NWAVE = WC * WP * KZ;               // 4 waves, or 8 (the 256 x 256 tile, the dense-wave forms, the K-split pairs)
    static_assert(KZ == 1 || (KZ == 2 && KB == 128 && sizeof(T) == 2), "K-split pairs: two K slices per stage, 16-bit types");
    constexpr int NW = BM / 16 * H / NWAVE, NX = BP / 16 * H / NWAVE;   // instructions per wave and stage
    constexpr int L = NW + NX;
    constexpr int KSUB = KB / 64;                     // logical steps (MFMA K slices) per stage
    constexpr int NR = PT + CT;                       // fragment reads per logical step: B_0 .. B_PT-1, A_0 .. A_CT-1
    static_assert((NWAVE == 4 || NWAVE == 8) && D >= 2 && D <= 10 && (KB == 64 || KB == 128) && NW >= 1 && NX >= 1, "bad configuration");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kz = KZ == 2 ? wave / (WC * WP) : 0;   // K-split pairs: which K slice of every stage this wave multiplies
    const int wq = KZ == 2 ? wave % (WC * WP) : wave;
    const int wc = wq / WP, wp = wq % WP;
    // 1-D grid; work item w = (pixel tile, channel tile) with the channel tile fastest: the channel tiles of one pixel
    // tile and neighbouring pixel tiles (3x3 halos) run on one XCD at about the same time and share its L2.
    const int CB = (p.cout + BM - 1) / BM;
    int w = p.xcd ? lh_xcd_remap(bid, nblk) : bid;
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

    // ---- per-lane source bookkeeping (buffer form).  Instruction q = NWAVE*j + wave of a stage fills (16-row group, part) =
    //      (q / H, q % H); this lane supplies slot lane % SL of row (q % H) * RPI + lane / SL, i.e. chunk slot ^ f(row).
    //      The activation descriptor starts `shift` bytes in front of the first image this tile touches (shift >= the most
    //      negative tap offset, so the SGPR offset shift + tap offset + K offset is never negative); a lane's offset is its
    //      pixel (tap (0,0), k = 0) relative to that image.
    constexpr unsigned OOR = 0x80000000u;   // an offset no descriptor of this kernel covers (records < 2^31: lh_ring_offsets_fit)
    const int th = (ntaps + tw - 1) / tw;
    const long ipix = (long)p.in_pix_stride * ES;
    const long shift = ((long)((dh0 < 0 ? -dh0 : dh0) + (dhs < 0 ? -dhs : dhs) * th) * p.wi + (dw0 < 0 ? -dw0 : dw0) + (dws < 0 ? -dws : dws) * tw) * ipix;
    const long img_bytes = (long)p.hi * p.wi * ipix;
    const int m_first = pblk * BP < p.M ? pblk * BP : p.M - 1;
    const int n_first = m_first / hw;
    const long in_off = (long)n_first * img_bytes - shift;
    const long in_rec = (long)p.n * img_bytes - in_off + 16;   // + 16: a chunk that straddles the end of the K run may straddle the tensor's end (those elements meet zero weights)
    const auto rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)(p.in + in_off), 0, (int)(in_rec < 0x7fffffffL ? in_rec : 0x7fffffffL), 0x00020000);
    const long kpad = p.kpad;
    const long w_off = (long)cblk * BM * ntaps * kpad * ES;
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)(wgt + w_off), 0, (int)((long)BM * ntaps * kpad * ES), 0x00020000);
    unsigned pvoff[NX], cvoff[NX];          // the lane's offset; what it issues for the current tap (pvoff or OOR)
    unsigned tmask[NX];                     // bit t set when tap t of this lane's pixel lies inside the image
    int klim[NX];                           // chunk is inside the K run while (stage K base) < klim
    const bool kfull = p.k_run % KSTEP == 0;   // every chunk of every stage lies inside the K run: klim never bites
#pragma unroll
    for (int j = 0; j < NX; ++j) {
        const int q = NWAVE * j + wave;
        const int g = q / H, lrow = (q % H) * RPI + lane / SL;
        const int c = (lane % SL) ^ ((lrow / (16 / SL)) & (SL - 1));
        const int row = g * 16 + lrow;
        const int m = pblk * BP + row;
        const bool ok = m < p.M;
        const int mm = ok ? m : m_first;
        const int n = mm / hw, rem = mm - n * hw;
        const int a = rem / p.wo, b = rem - a * p.wo;
        const int ih0 = a * p.sh, iw0 = b * p.sw;
        pvoff[j] = (unsigned)(((long)((n - n_first) * p.hi * p.wi + ih0 * p.wi + iw0) * p.in_pix_stride + c * EPC) * ES);
        klim[j] = p.k_run - c * EPC;        // <= 0: the chunk lies past the K run in every stage
        unsigned tm = 0;
        int t = 0;
        for (int ti = 0, dh = dh0; ti * tw < ntaps; ++ti, dh += dhs)
            for (int tjj = 0, dw = dw0; tjj < tw; ++tjj, dw += dws, ++t)
                if (ok && (unsigned)(ih0 + dh) < (unsigned)p.hi && (unsigned)(iw0 + dw) < (unsigned)p.wi) tm |= 1u << t;
        tmask[j] = tm;
        cvoff[j] = ((tm & 1u) != 0 && klim[j] > 0) ? pvoff[j] : OOR;     // (a K run shorter than one stage: the chunks past it are masked from stage 0 on)
    }
    unsigned wvoff[NW];
#pragma unroll
    for (int j = 0; j < NW; ++j) {
        const int q = NWAVE * j + wave;
        const int g = q / H, lrow = (q % H) * RPI + lane / SL;
        const int c = (lane % SL) ^ ((lrow / (16 / SL)) & (SL - 1));
        const int row = g * 16 + lrow;
        wvoff[j] = (unsigned)(((long)row * ntaps * kpad + c * EPC) * ES);
    }
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;

    // stage index -> (tap, kc) is tracked incrementally; `woff` is the byte offset of the stage inside a weight
    // row (stages are contiguous there), the activation's SGPR offset = shift + tap offset + K offset.
    int itap = 0, ikc = 0, tj = 0, cdh = dh0, cdw = dw0;
    int issued = 0, islot = 0;
    int woff = 0;
    auto issue = [&]() {
        unsigned char* st = smem + islot * STAGE;
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            const int q = NWAVE * j + wave;
            if (!(LH_ABL & 4))
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void_p)(st + (q / H) * GB + (q % H) * 1024), 16, wvoff[j], woff, 0, 0);
        }
        const int kbase = ikc * KSTEP;
        const int soff = (int)(shift + ((long)(cdh * p.wi + cdw) * p.in_pix_stride + kbase) * ES);
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            const int q = NWAVE * j + wave;
            if (!(LH_ABL & 4))
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_void_p)(st + BM * KB + (q / H) * GB + (q % H) * 1024), 16, cvoff[j], soff, 0, 0);
        }
        ++issued;
        if (++islot == D) islot = 0;
        woff += KB;
        bool retap = false;
        if (++ikc == p.kspt) {
            ikc = 0; ++itap;
            woff = (int)((long)itap * kpad * ES);
            cdw += dws;
            if (++tj == tw) { tj = 0; cdw = dw0; cdh += dhs; }
            retap = true;
        }
        if (retap || !kfull) {                // the next stage's in-range / out-of-range choice (wave-uniform branch)
            const int kb2 = ikc * KSTEP;
#pragma unroll
            for (int j = 0; j < NX; ++j) cvoff[j] = (((tmask[j] >> itap) & 1u) != 0 && kb2 < klim[j]) ? pvoff[j] : OOR;
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
        if (issued < S) issue();

    // fragment read addresses: lane (chunk c = 4*kk + (lane>>4), row = lane&15) of 16-row group g reads g*GB + foff[kk];
    // the group index goes into the instruction's immediate offset
    unsigned offA[KSUB], offB[KSUB];
#pragma unroll
    for (int kk = 0; kk < KSUB; ++kk) {
        const int c = 4 * kk + (lane >> 4), r = lane & 15;
        const unsigned foff = r * KB + ((c ^ ((r / (16 / SL)) & (SL - 1))) << 4);
        offA[kk] = lds_base + wc * CT * GB + foff;
        offB[kk] = lds_base + BM * KB + wp * PT * GB + foff;
    }
    // read r (r < PT: pixel fragment r, else weight fragment r - PT) given the lane's two base addresses of the K slice
    auto rd = [](auto Rc, uint4& dst, unsigned base_a, unsigned base_b) {
        constexpr int r = decltype(Rc)::value;
        if constexpr ((LH_ABL & 2) != 0) { dst = uint4{base_a, base_b, base_a, base_b}; return; }
        if constexpr (r < PT) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(base_b), "n"(r * GB));
        else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(base_a), "n"((r - PT) * GB));
    };
    // one logical step: the K slice whose fragment base addresses are ca / cb
    auto step_at = [&](const unsigned ca, const unsigned cb) {
        uint4 F[NR];
        static_for<0, NR>([&](auto r) { rd(r, F[decltype(r)::value], ca, cb); });
        if (LH_PRIO) __builtin_amdgcn_s_setprio(LH_PRIO);      // experiment: the wave in its MFMA phase issues ahead of its SIMD partner
        static_for<0, CT>([&](auto Ic) {
            constexpr int i = decltype(Ic)::value;
            // the reads younger than weight fragment i may stay in flight (LDS returns in order); the scheduling barrier in
            // front keeps the wait BEHIND the MFMAs of fragment i - 1 (the compiler hoisted it in front of all but one of them)
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(CT - 1 - i) : "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < PT; ++j)
                if (!(LH_ABL & 1)) MmaR<T>::run(F[PT + i], F[j], acc[i][j]);
        });
        if (LH_PRIO) __builtin_amdgcn_s_setprio(0);
    };
    // K slice kk of the stage at byte offset `so`
    auto step = [&](auto KKc, unsigned so) { step_at(offA[decltype(KKc)::value] + so, offB[decltype(KKc)::value] + so); };

    auto mfma_group = [&](auto Ic, uint4 (&F)[NR]) {
        constexpr int i = decltype(Ic)::value;
#pragma unroll
        for (int j = 0; j < PT; ++j)
            if (!(LH_ABL & 1)) MmaR<T>::run(F[PT + i], F[j], acc[i][j]);
    };
    // waves 4-7 of the 8-wave tile share their SIMDs with waves 0-3: they refill the ring half a stage later
    const bool late = NWAVE == 8 && KSUB == 2 && wave >= NWAVE / 2;
    int cslot = 0;                                   // ring slot of the stage being consumed
    for (int s = 0; s < S; ++s) {
        // stage s must have landed; every stage issued after it may stay in flight across the barrier
        wait_stages<L, D - 2>(issued - 1 - s);
        __builtin_amdgcn_s_barrier();
        const unsigned so = cslot * STAGE;
        if (++cslot == D) cslot = 0;
        // the refill goes into the slot of stage s - 1, whose reads every wave retired before the barrier
        if (!late && issued < S) issue();
        if constexpr (KZ == 2) {                     // this wave's K slice of the stage only (its pair partner takes the other)
            step_at(offA[KSUB - 1 < kz ? KSUB - 1 : kz] + so, offB[KSUB - 1 < kz ? KSUB - 1 : kz] + so);
            if (late && issued < S) issue();
            continue;
        }
        if constexpr (LH_PREREAD && KSUB == 2 && NR <= 8) {
            // experiment: the fragment reads of BOTH K slices of the stage up front (a second fragment register set), the
            // second slice's read latency under the first slice's MFMAs
            uint4 F0[NR], F1[NR];
            static_for<0, NR>([&](auto r) { rd(r, F0[decltype(r)::value], offA[0] + so, offB[0] + so); });
            static_for<0, NR>([&](auto r) { rd(r, F1[decltype(r)::value], offA[1] + so, offB[1] + so); });
            static_for<0, CT>([&](auto Ic) {
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NR + CT - 1 - decltype(Ic)::value) : "memory");
                __builtin_amdgcn_sched_barrier(0);
                mfma_group(Ic, F0);
            });
            if (late && issued < S) issue();
            static_for<0, CT>([&](auto Ic) {
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(CT - 1 - decltype(Ic)::value) : "memory");
                __builtin_amdgcn_sched_barrier(0);
                mfma_group(Ic, F1);
            });
            continue;
        }
        step(ic<0>{}, so);
        if constexpr (KSUB == 2) {
            if (late && issued < S) issue();
            step(ic<1>{}, so);
        }
    }
    if constexpr (BM == 256 && BP == 256 && sizeof(T) == 2) {
        if (p.head_w) {
            igemm_epilogue_head<T, BM, BP, WC, WP, MmaR<T>>(p, smem, acc, pblk, tid, lane, wave, wc, wp, hw, ooh, oow);
            return;
        }
    }
    igemm_epilogue<T, BM, BP, WC, WP, KZ>(p, smem, acc, pblk, cblk, tid, lane, wc, wp, hw, ooh, oow, stats, kz);
#endif
}

// waves per SIMD the register budget is sized for
template <int BM, int BP, int WC, int WP> constexpr int ring_waves_per_simd() { return 2; }

template <typename T, int BM, int BP, int WC, int WP, int D, int KB>
__global__ __launch_bounds__(64 * WC * WP, (ring_waves_per_simd<BM, BP, WC, WP>())) void igemm_ring_kernel(const IgemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    igemm_ring_body<T, BM, BP, WC, WP, D, KB>(p, smem, blockIdx.x, gridDim.x);
}
// the K-split wave-pair form (KZ = 2 above): 2 x WC x WP waves
template <typename T, int BM, int BP, int WC, int WP, int D, int KB>
__global__ __launch_bounds__(128 * WC * WP, 1) void igemm_ring_ksplit_kernel(const IgemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    igemm_ring_body<T, BM, BP, WC, WP, D, KB, 2>(p, smem, blockIdx.x, gridDim.x);
}

// Up to LH_MULTI_MAX independent convolutions that share the kernel configuration as ONE grid (lh_igemm_multi).
template <typename T, int BM, int BP, int WC, int WP, int D, int KB>
__global__ __launch_bounds__(64 * WC * WP, 2) void igemm_ring_multi_kernel(const LhMulti<IgemmArgs> m) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int bid, nblk;
    const int i = lh_multi_pick(m, bid, nblk);
    igemm_ring_body<T, BM, BP, WC, WP, D, KB>(m.a[i], smem, bid, nblk);
}

template <typename T, int BM, int BP, int WC, int WP, int D, int KB>
static int launch_ring(const IgemmArgs& a, hipStream_t s) {
    constexpr int ES = sizeof(T);
    constexpr int ring = D * (BM + BP) * KB;
    constexpr int epi = lh_epi_lds_bytes<T, BM, BP, (BM == 256 && BP == 256 && ES == 2)>();   // tile + the fused head's weights + per-channel constants
    constexpr int lds = ring > epi ? ring : epi;
    static_assert(lds <= 160 * 1024, "LDS budget");
    if (lds > 64 * 1024) {
        // per-device function attribute; cheap enough to set on every launch (no process-wide "done" flag:
        // a process may drive several devices)
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_ring_kernel<T, BM, BP, WC, WP, D, KB>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) {
            lh_set_error("igemm_ring: cannot raise dynamic LDS to %d bytes: %s", lds, hipGetErrorString(e));
            return LH_ERR_HIP;
        }
    }
    dim3 grid(ceil_div(a.M, BP) * ceil_div(a.cout, BM) * (a.nphase > 1 ? a.nphase : 1));
    hipLaunchKernelGGL((igemm_ring_kernel<T, BM, BP, WC, WP, D, KB>), grid, dim3(64 * WC * WP), lds, s, a);
    LH_LAUNCH_CHECK("igemm_ring launch");
    return LH_OK;
}


template <typename T, int BM, int BP, int WC, int WP, int D, int KB>
static int launch_ring_ksplit(const IgemmArgs& a, hipStream_t s) {
    constexpr int ring = D * (BM + BP) * KB;
    constexpr int epi = lh_epi_lds_bytes<T, BM, BP, false>() + lh_epi_ksplit_bytes<BM, BP, WC, WP>();
    constexpr int lds = ring > epi ? ring : epi;
    static_assert(lds <= 160 * 1024, "LDS budget");
    if (a.head_w) {
        lh_set_error("igemm_ring: the K-split form carries no fused head");
        return LH_ERR_UNSUPPORTED;
    }
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_ring_ksplit_kernel<T, BM, BP, WC, WP, D, KB>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) {
            lh_set_error("igemm_ring (K-split): cannot raise dynamic LDS to %d bytes: %s", lds, hipGetErrorString(e));
            return LH_ERR_HIP;
        }
    }
    dim3 grid(ceil_div(a.M, BP) * ceil_div(a.cout, BM) * (a.nphase > 1 ? a.nphase : 1));
    hipLaunchKernelGGL((igemm_ring_ksplit_kernel<T, BM, BP, WC, WP, D, KB>), grid, dim3(128 * WC * WP), lds, s, a);
    LH_LAUNCH_CHECK("igemm_ring (K-split) launch");
    return LH_OK;
}

template <typename T, int BM, int BP, int WC, int WP, int D, int KB>
static int launch_ring_multi(const LhMulti<IgemmArgs>& m, hipStream_t s) {
    constexpr int ring = D * (BM + BP) * KB;
    constexpr int epi = lh_epi_lds_bytes<T, BM, BP, false>();
    constexpr int lds = ring > epi ? ring : epi;
    static_assert(lds <= 160 * 1024, "LDS budget");
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_ring_multi_kernel<T, BM, BP, WC, WP, D, KB>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) {
            lh_set_error("igemm_ring_multi: cannot raise dynamic LDS to %d bytes: %s", lds, hipGetErrorString(e));
            return LH_ERR_HIP;
        }
    }
    hipLaunchKernelGGL((igemm_ring_multi_kernel<T, BM, BP, WC, WP, D, KB>), dim3(m.first[m.n]), dim3(64 * WC * WP), lds, s, m);
    LH_LAUNCH_CHECK("igemm_ring_multi launch");
    return LH_OK;
}
