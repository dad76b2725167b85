// Where does the operand-ingest rate of the tiled convolution kernel fall?  (round-4 verdict item 1a.)
//
// tools/ingest_bench.hip measured 131-145 GB/s per CU for LDS-DMA from an L2-resident private region; the library's
// kernels ingest 42-66.  This program walks from the one to the other ONE STEP AT A TIME on the launch the verdict names:
// the stage-3 3x3 convolution of R50 at batch 64 (256 -> 256 channels, 16 x 16 pixels, 64 images, bf16;
// reference: src/modeling/simplebaseline/pose_resnet.py:61-99) as igemm_ring_kernel<bf16,128,128,...,128> runs it --
// 256 workgroups (128 pixel tiles x 2 channel tiles, one per CU), 36 ring stages of (128 + 128) rows x 128 bytes = 32 KiB.
//
//   rung   source addresses of the LDS-DMA                                   add-ons
//   R0     linear, a private 64 KiB region per workgroup (L2-resident)       -
//   R1     the real footprint, centre tap only: 128 weight rows (pitch 4608 B) + 128 pixel rows (pitch 512 B),
//          K walk of 4 stages x 9                                            -
//   R2     + the nine shifted taps (halo re-reads), unmasked (guard band)    -
//   R3     + the kernel's per-lane address path: tap-mask bit test, K-limit compare, 64-bit add, select against the zero page
//   R4     R3                                                                + fragment ds_read_b128 (no MFMA)
//   R5     R3                                                                + fragment reads + MFMAs  (= the kernel's K loop)
// each with the ring D = 2 / 3 / 4 stages deep, 4 or 8 waves, with and without the per-stage s_barrier, in three cache states:
//   warm      the launch repeated back to back (its 9.6 MB footprint stays in L2 / Infinity Cache)
//   cold      a 512 MiB fill before every launch (operands come from HBM)
//   producer  the fill, then the activation rewritten by a copy kernel (what a launch meets INSIDE a training step)
// Times: span = last workgroup's end - first workgroup's start (s_memrealtime), loop = mean over workgroups of the stage
// loop alone; GB/s per CU = 36 x 32 KiB / loop.
// build: hipcc -O3 --offload-arch=gfx950 tools/ingest_ladder.hip -o /tmp/ingest_ladder      usage: ingest_ladder [reps = 12]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_p;
typedef const __attribute__((address_space(1))) void* gbl_p;

struct Args {
    const unsigned char* in;     // [N*H*W][C] bf16, guard band on both sides
    const unsigned char* w;      // [COUT][9][C] bf16
    const unsigned char* zero;   // 16 zero bytes
    const unsigned char* lin;    // 256 x 64 KiB
    float* sink;
    unsigned long long* stamps;  // [nwg][5]: realtime t0 t1 t2, shader clock c1 c2
    int n, h, wd, c, cout;
};

constexpr int BM = 128, BP = 128, KB = 128, SL = 8, RPI = 8, GB = 16 * KB, STAGE = (BM + BP) * KB, HH = 2;

__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}

template <int I> struct ic { static constexpr int value = I; };
template <int B, int E, typename F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) { f(ic<B>{}); static_for<B + 1, E>(f); }
}
template <int L, int MAXS> __device__ __forceinline__ void wait_stages(int stages) {
    if constexpr (MAXS == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else {
        if (stages >= MAXS) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MAXS * L) : "memory");
        else wait_stages<L, MAXS - 1>(stages);
    }
}

// SRC 0..3 = rungs R0..R3's address form; ADD bit 0 = fragment reads, bit 1 = MFMAs; SYNC 0 = s_barrier per stage, 1 = none
// DW > 0 (buffer form, plain loop only): the WEIGHT halves of the stages live in a ring of their own, DW stages deep, filled DW - D
// stages further ahead than the pixel halves -- the weights are what every workgroup first-touches in lockstep (sitting 6)
// KS = 2 (sitting 8): SPLIT-K emulation -- twice the workgroups, each walks HALF the stages of a tile (workgroup 2 t + h: tile t,
// stages 18 h .. 18 h + 17), two of them per CU (D = 2: 64 KB of LDS each): does a CU with two independent half-loops finish a tile's
// 36 stages sooner than one workgroup walking them in a row?  (No reduction of the two partial tiles is done: timing only.)
template <int NWAVE, int D, int SRC, int ADD, int SYNC, int DW = 0, int KS = 1>
__global__ __launch_bounds__(64 * NWAVE, 2) void ladder_kernel(const Args p) {
#if defined(__HIP_DEVICE_COMPILE__)        // the buffer builtins do not exist in the host pass
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int WC = 2, WP = NWAVE / 2, TC = BM / WC, TP = BP / WP, CT = TC / 16, PT = TP / 16, NR = CT + PT;
    constexpr int NW = BM / 16 * HH / NWAVE, NX = BP / 16 * HH / NWAVE, L = NW + NX;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave / WP, wp = wave % WP;
    const int bid = blockIdx.x, nblk = gridDim.x;
    unsigned long long t0 = 0, t1 = 0, t2 = 0, c1 = 0, c2 = 0;
    if (tid == 0) t0 = wall_clock64();
    const int w0 = xcd_remap(bid, nblk);
    const int khalf = KS > 1 ? w0 % KS : 0;
    const int w = KS > 1 ? w0 / KS : w0;
    const int pblk = w / 2, cblk = w % 2;
    const int hw = p.h * p.wd, ntaps = 9, kspt = p.c * 2 / KB;
    const long kpad = p.c;

    long pbase[NX];
    unsigned tmask[NX];
    int klim[NX];
#pragma unroll
    for (int j = 0; j < NX; ++j) {
        const int q = NWAVE * j + wave;
        const int g = q / HH, lrow = (q % HH) * RPI + lane / SL;
        const int c = (lane % SL) ^ ((lrow / (16 / SL)) & (SL - 1));
        const int row = g * 16 + lrow;
        const int m = pblk * BP + row;
        const int n = m / hw, rem = m - n * hw;
        const int a = rem / p.wd, b = rem - a * p.wd;
        pbase[j] = ((long)(n * hw + a * p.wd + b) * p.c + c * 8) * 2;
        klim[j] = p.c - c * 8;
        unsigned tm = 0;
        int t = 0;
        for (int dh = -1; dh <= 1; ++dh)
            for (int dw = -1; dw <= 1; ++dw, ++t)
                if ((unsigned)(a + dh) < (unsigned)p.h && (unsigned)(b + dw) < (unsigned)p.wd) tm |= 1u << t;
        tmask[j] = tm;
        if (SRC == 0) pbase[j] = (long)bid * 65536 + (NW * NWAVE + q) * 1024 + lane * 16;   // second half of the 32 KiB stage image
    }
    const unsigned char* wsrc[NW];
#pragma unroll
    for (int j = 0; j < NW; ++j) {
        const int q = NWAVE * j + wave;
        const int g = q / HH, lrow = (q % HH) * RPI + lane / SL;
        const int c = (lane % SL) ^ ((lrow / (16 / SL)) & (SL - 1));
        const int row = g * 16 + lrow;
        wsrc[j] = p.w + ((long)(cblk * BM + row) * ntaps * kpad + c * 8) * 2;
        if (SRC == 0) wsrc[j] = p.lin + (long)bid * 65536 + q * 1024 + lane * 16;
    }
    // SRC 4: the buffer form -- one descriptor per operand, per-lane 32-bit offsets fixed for the whole launch, everything that
    // changes per stage in the SGPR offset; a lane whose tap falls outside the image gets an out-of-range offset (the range check
    // returns zeros), chosen ONCE PER TAP, so a load costs no vector instruction at all
    const long shift = (long)(p.wd + 1) * p.c * 2;
    const auto rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)(p.in - shift), 0, (unsigned)((long)p.n * hw * p.c * 2 + shift), 0x00020000);
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)(p.w + (long)cblk * BM * ntaps * kpad * 2), 0, (unsigned)((long)BM * ntaps * kpad * 2), 0x00020000);
    unsigned pvoff[NX], cvoff[NX], wvoff[NW];
#pragma unroll
    for (int j = 0; j < NX; ++j) { pvoff[j] = (unsigned)pbase[j]; cvoff[j] = (tmask[j] & 1u) ? pvoff[j] : 0x80000000u; }
#pragma unroll
    for (int j = 0; j < NW; ++j) wvoff[j] = (unsigned)(wsrc[j] - (p.w + (long)cblk * BM * ntaps * kpad * 2));
    const unsigned char* zero = p.zero;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;

    int itap = 0, ikc = 0, tj = 0, cdh = -1, cdw = -1, issued = 0, islot = 0;
    long woff = 0;
    constexpr int WRING = DW * BM * KB;                           // bytes of the separate weight ring (0: weights share the stage slots)
    int w_issued = 0, w_slot = 0, w_tap = 0, w_kc = 0;
    long w_off = 0;
    // piece i of the stage being issued: i < NW a weight piece, else a pixel piece; `advance` closes the stage
    auto piece = [&](int i) {
        unsigned char* st = smem + islot * STAGE;
        const long lin_off = (issued & 1) * 32768;
        const int kbase = ikc * (KB / 2);
        if (SRC == 5) return;
        if (i < NW) {
            const int j = i, q = NWAVE * j + wave;
            if (DW > 0) return;                                   // weights: issue_w()
            if (SRC == 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_p)(st + (q / HH) * GB + (q % HH) * 1024), 16, wvoff[j], (int)woff, 0, 0);
            else {
                const unsigned char* src = SRC == 0 ? wsrc[j] + lin_off : wsrc[j] + woff;
                __builtin_amdgcn_global_load_lds((gbl_p)src, (lds_p)(st + (q / HH) * GB + (q % HH) * 1024), 16, 0, 0);
            }
        } else {
            const int j = i - NW, q = NWAVE * j + wave;
            lds_p dst = DW > 0 ? (lds_p)(smem + WRING + islot * (BP * KB) + (q / HH) * GB + (q % HH) * 1024)
                               : (lds_p)(st + BM * KB + (q / HH) * GB + (q % HH) * 1024);
            if (SRC == 4) {
                const int soff = ((cdh + 1) * p.wd + (cdw + 1)) * p.c * 2 + kbase * 2;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, dst, 16, cvoff[j], soff, 0, 0);
                return;
            }
            const unsigned char* tsrc = SRC >= 2 ? p.in + ((long)(cdh * p.wd + cdw) * p.c + kbase) * 2 : p.in + (long)kbase * 2;
            const unsigned char* src;
            if (SRC == 0) src = p.lin + pbase[j] + lin_off;
            else if (SRC <= 2) src = tsrc + pbase[j];
            else {
                const bool ok = (int)((tmask[j] >> itap) & 1u) & (int)(kbase < klim[j]);
                src = ok ? tsrc + pbase[j] : zero;
            }
            __builtin_amdgcn_global_load_lds((gbl_p)src, dst, 16, 0, 0);
        }
    };
    auto advance = [&]() {
        ++issued;
        if (++islot == D) islot = 0;
        woff += KB;
        if (++ikc == kspt) {
            ikc = 0; ++itap;
            woff = (long)itap * kpad * 2;
            cdw += 1;
            if (++tj == 3) { tj = 0; cdw = -1; cdh += 1; }
            if (SRC == 4) {
#pragma unroll
                for (int j = 0; j < NX; ++j) cvoff[j] = ((tmask[j] >> itap) & 1u) ? pvoff[j] : 0x80000000u;
            }
        }
    };
    auto issue_w = [&]() {                                        // the weight half of stage w_issued into the weight ring
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            const int q = NWAVE * j + wave;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_p)(smem + w_slot * (BM * KB) + (q / HH) * GB + (q % HH) * 1024), 16, wvoff[j], (int)w_off, 0, 0);
        }
        ++w_issued;
        if (++w_slot == (DW > 0 ? DW : 1)) w_slot = 0;
        w_off += KB;
        if (++w_kc == kspt) { w_kc = 0; ++w_tap; w_off = (long)w_tap * kpad * 2; }
    };
    auto issue = [&]() {
        if (DW > 0 && w_issued < ntaps * kspt) issue_w();
#pragma unroll
        for (int i = 0; i < L; ++i) piece(i);
        advance();
    };

    f32x4 acc[CT][PT];
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
        for (int j = 0; j < PT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int S = ntaps * kspt / KS;
    if (KS > 1 && khalf > 0) {                                    // fast-forward the stage walk to this half's first stage
        for (int s = 0; s < S * khalf; ++s) advance();
        issued = 0; islot = 0;
    }
    if (tid == 0) { t1 = wall_clock64(); c1 = clock64(); }
    if (DW > 0) {
#pragma unroll
        for (int s = 0; s < DW - D; ++s)
            if (w_issued < S) issue_w();                          // the weight ring's head start
    }
#pragma unroll
    for (int s = 0; s < D - 1; ++s)
        if (issued < S) issue();

    unsigned offA[2], offB[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int c = 4 * kk + (lane >> 4), r = lane & 15;
        const unsigned foff = r * KB + ((c ^ ((r / (16 / SL)) & (SL - 1))) << 4);
        offA[kk] = lds_base + wc * CT * GB + foff;
        offB[kk] = lds_base + (DW > 0 ? WRING : BM * KB) + wp * PT * GB + foff;
    }
    auto rd = [](auto Rc, uint4& dst, unsigned base_a, unsigned base_b) {
        constexpr int r = decltype(Rc)::value;
        if constexpr ((ADD & 16) != 0) { dst = uint4{base_a | 0x3c003c00u, base_b | 0x3c003c00u, 0x3c013c02u + r, 0x3c033c04u}; return; }
        if constexpr (r < PT) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(base_b), "n"(r * GB));
        else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(base_a), "n"((r - PT) * GB));
    };
    constexpr bool ILV = (ADD & 4) != 0;          // the stage's DMA pieces issued BETWEEN the MFMA groups instead of up front
    constexpr int SLOTS = 2 * CT, PER = (L + SLOTS - 1) / SLOTS, EVERY = SLOTS / (L < SLOTS ? L : SLOTS);
    int cw_slot = 0;                                              // weight-ring slot of the stage being consumed (DW > 0)
    auto step = [&](auto KKc, unsigned so, bool dma) {
        constexpr int kk = decltype(KKc)::value;
        const unsigned ca = offA[kk] + (DW > 0 ? (unsigned)(cw_slot * (BM * KB)) : so);
        const unsigned cb = offB[kk] + (DW > 0 ? so / (unsigned)STAGE * (unsigned)(BP * KB) : so);
        uint4 F[NR];
        static_for<0, NR>([&](auto r) { rd(r, F[decltype(r)::value], ca, cb); });
        if constexpr ((ADD & 2) != 0) {
            static_for<0, CT>([&](auto Ic) {
                constexpr int i = decltype(Ic)::value;
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(CT - 1 - i) : "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < PT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, F[PT + i]), __builtin_bit_cast(bf16x8, F[j]), acc[i][j], 0, 0, 0);
                if constexpr (ILV) {
                    constexpr int slot = kk * CT + i;
                    if constexpr (slot % EVERY == 0 && slot / EVERY * PER < L) {
                        if (dma) {
#pragma unroll
                            for (int q = 0; q < PER; ++q) piece(slot / EVERY * PER + q);
                            if constexpr (slot / EVERY * PER + PER >= L) advance();
                        }
                    }
                }
            });
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int r = 0; r < NR; ++r) asm volatile("" ::"v"(F[r].x), "v"(F[r].y), "v"(F[r].z), "v"(F[r].w));
        }
    };

    const bool late = NWAVE == 8 && wave >= NWAVE / 2;
    int cslot = 0;
    if constexpr ((ADD & 8) != 0) {
        // software-pipelined form: the fragment reads run ONE K SLICE AHEAD of the MFMAs (two fragment register sets), the
        // stage's barrier sits in the MIDDLE of the stage (behind the reads of its second slice), the refill of the slot it
        // frees and the first reads of the next stage follow it and land under the second slice's MFMAs
        uint4 FA[NR], FB[NR];
        auto rdall = [&](uint4 (&F)[NR], auto KKc, unsigned so) {
            constexpr int kk = decltype(KKc)::value;
            const unsigned ca = offA[kk] + so, cb = offB[kk] + so;
            static_for<0, NR>([&](auto r) { rd(r, F[decltype(r)::value], ca, cb); });
        };
        auto mm = [&](uint4 (&F)[NR]) {
#pragma unroll
            for (int i = 0; i < CT; ++i)
#pragma unroll
                for (int j = 0; j < PT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, F[PT + i]), __builtin_bit_cast(bf16x8, F[j]), acc[i][j], 0, 0, 0);
        };
        // ADD bit 5 (32): the same loop with the next slice's reads SPREAD between this slice's MFMA groups (NR / CT reads behind
        // every group but the last) instead of in one burst in front of them
        auto mm_rd = [&](uint4 (&F)[NR], uint4 (&G)[NR], auto KKc, unsigned so) {
            constexpr int kk = decltype(KKc)::value;
            const unsigned ca = offA[kk] + so, cb = offB[kk] + so;
            constexpr int PERG = (NR + CT - 2) / (CT - 1);
            static_for<0, CT>([&](auto Ic) {
                constexpr int i = decltype(Ic)::value;
#pragma unroll
                for (int j = 0; j < PT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, F[PT + i]), __builtin_bit_cast(bf16x8, F[j]), acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                static_for<i * PERG, ((i + 1) * PERG < NR ? (i + 1) * PERG : NR)>([&](auto r) { rd(r, G[decltype(r)::value], ca, cb); });
                __builtin_amdgcn_sched_barrier(0);
            });
        };
        if constexpr ((ADD & 32) != 0) {
            wait_stages<L, D - 2>(issued - 1);
            if (SYNC == 0) __builtin_amdgcn_s_barrier();
            if (issued < S) issue();
            rdall(FA, ic<0>{}, 0u);
            for (int s = 0; s < S; ++s) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                mm_rd(FA, FB, ic<1>{}, (unsigned)(cslot * STAGE));          // slice 0's MFMAs, slice 1's reads between them
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                if (s + 1 < S) {
                    wait_stages<L, D - 2>(issued - 2 - s);
                    if (SYNC == 0) __builtin_amdgcn_s_barrier();
                    if (issued < S) issue();
                    if (++cslot == D) cslot = 0;
                    mm_rd(FB, FA, ic<0>{}, (unsigned)(cslot * STAGE));      // slice 1's MFMAs, the next stage's first reads between them
                } else {
                    mm(FB);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
        wait_stages<L, D - 2>(issued - 1);
        if (SYNC == 0) __builtin_amdgcn_s_barrier();
        if (issued < S) issue();
        rdall(FA, ic<0>{}, 0u);
        for (int s = 0; s < S; ++s) {
            rdall(FB, ic<1>{}, (unsigned)(cslot * STAGE));
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NR) : "memory");
            __builtin_amdgcn_sched_barrier(0);
            mm(FA);
            __builtin_amdgcn_sched_barrier(0);
            if (s + 1 < S) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                wait_stages<L, D - 2>(issued - 2 - s);
                if (SYNC == 0) __builtin_amdgcn_s_barrier();
                if (issued < S) issue();
                if (++cslot == D) cslot = 0;
                rdall(FA, ic<0>{}, (unsigned)(cslot * STAGE));
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NR) : "memory");
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
            mm(FB);
            __builtin_amdgcn_sched_barrier(0);
        }
        }
    } else
    for (int s = 0; s < S; ++s) {
        wait_stages<L, D - 2>(issued - 1 - s);
        if (SYNC == 0) __builtin_amdgcn_s_barrier();
        const unsigned so = cslot * STAGE;
        if (++cslot == D) cslot = 0;
        if constexpr (ILV) {
            const bool dma = issued < S;
            step(ic<0>{}, so, dma);
            step(ic<1>{}, so, dma);
        } else {
            if (!late && issued < S) issue();
            if constexpr ((ADD & 1) != 0) step(ic<0>{}, so, false);
            if (late && issued < S) issue();
            if constexpr ((ADD & 1) != 0) step(ic<1>{}, so, false);
            if (DW > 0 && ++cw_slot == DW) cw_slot = 0;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        t2 = wall_clock64(); c2 = clock64();
        p.stamps[bid * 5] = t0; p.stamps[bid * 5 + 1] = t1; p.stamps[bid * 5 + 2] = t2; p.stamps[bid * 5 + 3] = c1; p.stamps[bid * 5 + 4] = c2;
    }
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
        for (int j = 0; j < PT; ++j) v += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    v += (float)smem[(tid * 16) % (D * STAGE)];
    if (v == 12345.678f) p.sink[bid * 64 * NWAVE + tid] = v;      // never true in practice; keeps every value live
#endif
}

__global__ void copy_kernel(const uint4* __restrict__ a, uint4* __restrict__ b, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) b[i] = a[i];
}

// L2 warm-up kernels for the cache-state experiment: every XCD (blocks b, b + 8, ... share one) reads the WHOLE weight pack / its
// own eighth of the pixel rows (the conv's work-item order: XCD x owns pixel tiles [16 x, 16 x + 16) of the 128)
__global__ void touch_kernel(const uint4* __restrict__ w, long nw, const uint4* __restrict__ in, long nin_per_xcd, int what, unsigned* sink) {
    const int xcd = blockIdx.x & 7, sub = blockIdx.x >> 3, nsub = gridDim.x >> 3;
    unsigned acc = 0;
    if (what & 1)
        for (long i = (long)sub * blockDim.x + threadIdx.x; i < nw; i += (long)nsub * blockDim.x) acc ^= w[i].x;
    if (what & 2)
        for (long i = (long)sub * blockDim.x + threadIdx.x; i < nin_per_xcd; i += (long)nsub * blockDim.x) acc ^= in[(long)xcd * nin_per_xcd + i].x;
    if (acc == 0x12345u) sink[0] = acc;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

static bool g_more_states = false;     // + producer state followed by an L2 warm-up of the weights (L2w), the pixel rows (L2x), both (L2wx)

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 12;
    const int N = 64, H = 16, W = 16, C = 256, COUT = 256, NWG = 256;
    const size_t in_bytes = (size_t)N * H * W * C * 2, guard = 65536, w_bytes = (size_t)COUT * 9 * C * 2;
    unsigned char *in_raw, *twin, *wgt, *zero, *lin, *flush;
    float* sink; unsigned long long* stamps;
    CK(hipMalloc(&in_raw, in_bytes + 2 * guard)); CK(hipMalloc(&twin, in_bytes)); CK(hipMalloc(&wgt, w_bytes));
    CK(hipMalloc(&zero, 256)); CK(hipMalloc(&lin, (size_t)NWG * 65536)); CK(hipMalloc(&flush, 512u << 20));
    CK(hipMalloc(&sink, 2 * NWG * 512 * 4)); CK(hipMalloc(&stamps, 2 * NWG * 5 * 8));
    {   // random bit patterns of ordinary bf16 magnitude (zero operands clock higher)
        std::vector<unsigned short> hbuf((in_bytes + 2 * guard) / 2);
        unsigned s = 12345u;
        for (auto& v : hbuf) { s = s * 1664525u + 1013904223u; v = (unsigned short)(0x3c00u + ((s >> 9) & 0x3ffu) + ((s >> 31) << 15)); }
        CK(hipMemcpy(in_raw, hbuf.data(), in_bytes + 2 * guard, hipMemcpyHostToDevice));
        CK(hipMemcpy(twin, hbuf.data() + guard / 2, in_bytes, hipMemcpyHostToDevice));
        CK(hipMemcpy(wgt, hbuf.data(), w_bytes, hipMemcpyHostToDevice));
        CK(hipMemcpy(lin, hbuf.data(), std::min((size_t)NWG * 65536, in_bytes), hipMemcpyHostToDevice));
        CK(hipMemset(zero, 0, 256));
    }
    Args a{in_raw + guard, wgt, zero, lin, sink, stamps, N, H, W, C, COUT};
    std::vector<unsigned long long> hs(2 * NWG * 5);
    int nrun = NWG;                                               // workgroups of the launch being measured (split-K sitting: 2 NWG)
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

    printf("# rung / ring / waves / sync                    state     span us   loop us  prologue us  GB/s per CU (loop)   wall us (events)\n");
    auto measure = [&](const char* name, auto launch) {
        for (int state = 0; state < (g_more_states ? 6 : 3); ++state) {
            std::vector<double> span, loop, pro, wall, ghz;
            for (int r = 0; r < reps + 2; ++r) {
                if (state >= 1) CK(hipMemsetAsync(flush, r, 512u << 20, 0));
                if (state >= 2) hipLaunchKernelGGL(copy_kernel, dim3(2048), dim3(256), 0, 0, (const uint4*)twin, (uint4*)(in_raw + guard), (long)(in_bytes / 16));
                if (state >= 3) hipLaunchKernelGGL(touch_kernel, dim3(256), dim3(256), 0, 0, (const uint4*)wgt, (long)(w_bytes / 16), (const uint4*)(in_raw + guard),
                                                   (long)(in_bytes / 16 / 8), state - 2, (unsigned*)sink);
                CK(hipEventRecord(e0, 0));
                launch();
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
                CK(hipMemcpy(hs.data(), stamps, (size_t)nrun * 5 * 8, hipMemcpyDeviceToHost));
                if (r < 2) continue;
                unsigned long long mn = ~0ull, mx = 0; double lp = 0, pr = 0, cy = 0;
                for (int i = 0; i < nrun; ++i) {
                    mn = std::min(mn, hs[i * 5]); mx = std::max(mx, hs[i * 5 + 2]);
                    lp += (double)(hs[i * 5 + 2] - hs[i * 5 + 1]); pr += (double)(hs[i * 5 + 1] - hs[i * 5]);
                    cy += (double)(hs[i * 5 + 4] - hs[i * 5 + 3]);
                }
                ghz.push_back(cy / (lp * 10.0));            // shader cycles per ns over the loop (s_memtime / s_memrealtime at 100 MHz)
                span.push_back((mx - mn) * 0.01); loop.push_back(lp / nrun * 0.01); pro.push_back(pr / nrun * 0.01); wall.push_back(ms * 1e3);
            }
            auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
            const double l = med(loop);
            printf("%-48s %-9s %7.2f  %7.2f  %7.2f      %7.1f              %7.2f   %5.2f GHz  %5.0f cyc/stage\n", name, state == 0 ? "warm" : state == 1 ? "cold" : state == 2 ? "producer" : state == 3 ? "prod+L2w" : state == 4 ? "prod+L2x" : "prod+L2wx",
                   med(span), l, med(pro), 36.0 * 32768 / l / 1e3, med(wall), med(ghz), l * 1e3 * med(ghz) / 36.0);
        }
        fflush(stdout);
    };
#define RUN(NWAVE, D, SRC, ADD, SYNC, NAME) { \
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ladder_kernel<NWAVE, D, SRC, ADD, SYNC>), hipFuncAttributeMaxDynamicSharedMemorySize, D * STAGE)); \
        measure(NAME, [&]() { hipLaunchKernelGGL((ladder_kernel<NWAVE, D, SRC, ADD, SYNC>), dim3(NWG), dim3(64 * NWAVE), D * STAGE, 0, a); }); }
#define RUNK(NWAVE, D, ADD, NAME) { \
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ladder_kernel<NWAVE, D, 4, ADD, 0, 0, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, D * STAGE)); \
        nrun = 2 * NWG; \
        measure(NAME, [&]() { hipLaunchKernelGGL((ladder_kernel<NWAVE, D, 4, ADD, 0, 0, 2>), dim3(2 * NWG), dim3(64 * NWAVE), D * STAGE, 0, a); }); \
        nrun = NWG; }
#define RUNW(NWAVE, D, DW, ADD, NAME) { \
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ladder_kernel<NWAVE, D, 4, ADD, 0, DW>), hipFuncAttributeMaxDynamicSharedMemorySize, DW * BM * KB + D * BP * KB)); \
        measure(NAME, [&]() { hipLaunchKernelGGL((ladder_kernel<NWAVE, D, 4, ADD, 0, DW>), dim3(NWG), dim3(64 * NWAVE), DW * BM * KB + D * BP * KB, 0, a); }); }

    const bool second = argc > 2 && atoi(argv[2]) >= 2;
    g_more_states = argc > 2 && atoi(argv[2]) == 3;
    if (!second) {
    RUN(8, 3, 0, 0, 0, "R0 linear private region      D3 8w barrier")
    RUN(8, 3, 1, 0, 0, "R1 real rows, centre tap      D3 8w barrier")
    RUN(8, 3, 2, 0, 0, "R2 + nine shifted taps        D3 8w barrier")
    RUN(8, 3, 3, 0, 0, "R3 + mask / limit / select    D3 8w barrier")
    RUN(8, 3, 3, 1, 0, "R4 + fragment reads           D3 8w barrier")
    RUN(8, 3, 3, 3, 0, "R5 + MFMAs (the K loop)       D3 8w barrier")
    RUN(8, 3, 0, 3, 0, "R0 + reads + MFMAs            D3 8w barrier")
    RUN(8, 3, 3, 0, 1, "R3                            D3 8w no barrier")
    RUN(8, 3, 3, 3, 1, "R5 (results invalid)          D3 8w no barrier")
    RUN(8, 2, 3, 0, 0, "R3                            D2 8w barrier")
    RUN(8, 4, 3, 0, 0, "R3                            D4 8w barrier")
    RUN(8, 2, 3, 3, 0, "R5                            D2 8w barrier")
    RUN(8, 4, 3, 3, 0, "R5                            D4 8w barrier")
    RUN(4, 3, 3, 0, 0, "R3                            D3 4w barrier")
    RUN(4, 3, 3, 3, 0, "R5                            D3 4w barrier")
    RUN(4, 4, 3, 3, 0, "R5                            D4 4w barrier")
    RUN(4, 3, 0, 3, 0, "R0 + reads + MFMAs            D3 4w barrier")
    } else {
    // later sittings: what the complete loop is made of, the buffer form, pipelined stage loops
    if (argc > 2 && atoi(argv[2]) == 5) {                       // sitting 8: split-K emulation (span = the whole launch; 'loop' = ONE half)
        RUN(8, 3, 4, 3, 0, "B5 buffer form + reads + MFMA D3 8w (256 WGs x 36 stages)")
        RUN(8, 2, 4, 3, 0, "B5 buffer form + reads + MFMA D2 8w (256 WGs x 36 stages)")
        RUNK(8, 2, 3, "K2 split-K 2: 512 WGs x 18 stages, D2 8w, two per CU")
        RUNK(4, 2, 3, "K2 split-K 2: 512 WGs x 18 stages, D2 4w, two per CU")
        RUNK(4, 3, 3, "K2 split-K 2: 512 WGs x 18 stages, D3 4w (one per CU: 96 KB)")
        RUN(8, 3, 4, 0, 0, "B3 DMA only D3 8w (256 WGs x 36 stages)")
        RUNK(8, 2, 0, "K2 DMA only: 512 WGs x 18 stages, D2 8w")
        return 0;
    }
    if (argc > 2 && atoi(argv[2]) == 4) {                       // sitting 7: a deeper ring for the weights only
        RUN(8, 3, 4, 0, 0, "B3 buffer form, DMA only      D3 8w")
        RUNW(8, 3, 5, 0, "B3 DMA only, weight ring 5 deep, pixels 3")
        RUNW(8, 3, 7, 0, "B3 DMA only, weight ring 7 deep, pixels 3")
        RUN(8, 3, 4, 3, 0, "B5 buffer form + reads + MFMA D3 8w")
        RUNW(8, 3, 4, 3, "B5 weight ring 4 deep, pixels 3      8w")
        RUNW(8, 3, 5, 3, "B5 weight ring 5 deep, pixels 3      8w")
        RUNW(8, 3, 7, 3, "B5 weight ring 7 deep, pixels 3      8w")
        RUNW(8, 2, 6, 3, "B5 weight ring 6 deep, pixels 2      8w")
        RUN(4, 3, 4, 3, 0, "B5 buffer form + reads + MFMA D3 4w")
        RUNW(4, 3, 5, 3, "B5 weight ring 5 deep, pixels 3      4w")
        RUNW(4, 3, 7, 3, "B5 weight ring 7 deep, pixels 3      4w")
        return 0;
    }
    if (g_more_states) {
        RUN(8, 3, 4, 0, 0, "B3 buffer form, DMA only      D3 8w barrier")
        RUN(8, 3, 4, 3, 0, "B5 buffer form + reads + MFMA D3 8w barrier")
        RUN(8, 3, 4, 43, 0, "I5 buffer form, reads spread            D3 8w")
        RUN(4, 3, 4, 43, 0, "I5 buffer form, reads spread            D3 4w")
        return 0;
    }
    RUN(8, 3, 5, 3, 0, "no DMA: reads + MFMAs only    D3 8w barrier")
    RUN(8, 3, 5, 3, 1, "no DMA: reads + MFMAs only    D3 8w no barrier")
    RUN(8, 3, 3, 3, 0, "R5 global_load_lds, masked    D3 8w barrier")
    RUN(8, 3, 4, 0, 0, "B3 buffer form, DMA only      D3 8w barrier")
    RUN(8, 3, 4, 3, 0, "B5 buffer form + reads + MFMA D3 8w barrier")
    RUN(4, 3, 5, 3, 0, "no DMA: reads + MFMAs only    D3 4w barrier")
    RUN(4, 3, 4, 3, 0, "B5 buffer form + reads + MFMA D3 4w barrier")
    RUN(8, 3, 5, 11, 0, "P  no DMA, reads a slice ahead (burst)  D3 8w")
    RUN(8, 3, 4, 11, 0, "P5 buffer form, reads a slice ahead     D3 8w")
    RUN(8, 3, 5, 43, 0, "I  no DMA, reads ahead, SPREAD between MFMA groups D3 8w")
    RUN(8, 3, 5, 43, 1, "I  the same, no barrier                 D3 8w")
    RUN(8, 3, 4, 43, 0, "I5 buffer form, reads spread            D3 8w")
    RUN(8, 4, 4, 43, 0, "I5 buffer form, reads spread            D4 8w")
    RUN(4, 3, 5, 43, 0, "I  no DMA, reads spread                 D3 4w")
    RUN(4, 3, 4, 43, 0, "I5 buffer form, reads spread            D3 4w")
    RUN(4, 4, 4, 43, 0, "I5 buffer form, reads spread            D4 4w")
    }
    return 0;
}
