// Persistent pointwise (1x1) convolution kernel for the short-K layers (K <= 512, 16-bit types): forward and data
// gradient of the bottleneck 1x1 convolutions (pose_resnet.py:66-72, 180-184), the projection shortcuts and the head.
//
//   out[pixel][co] = sum_k in[pix(pixel)][k] * wpack[co][k]                       (one tap at offset (0, 0), dense output)
//
// These layers are bandwidth-bound streams with 1-4 K slices per tile: a tiled launch pays a weight fetch, a ring fill
// and an epilogue drain per 64-256 pixels.  Here a workgroup is resident for the whole launch:
//  * its weight panel [BM output channels][KC] is fetched ONCE by LDS-DMA (same swizzled 128-byte row image as a stage
//    of igemm_ring_kernel, so the fragment reads are the same conflict-free ds_read_b128) and stays in LDS;
//  * its four waves are independent streams (no barrier inside the loop): a wave takes 16*PT pixels at a time, loads
//    their K run STRAIGHT INTO MFMA "B" fragments (global_load_dwordx4, no LDS round trip), multiplies against the
//    panel, and transposes its [pixels][BM] tile through a wave-private LDS patch 64 channels at a time so that every
//    global store is a full 128-byte line of an NHWC row;
//  * the operand rows of the NEXT tile are requested before the epilogue of this one, so the loads fly under the
//    stores (the cross-tile overlap a one-tile-per-workgroup launch cannot have);
//  * BatchNorm partial sums are kept in registers across all tiles of the wave and leave the workgroup as ONE slab row
//    (rows = workgroups per channel block, a few hundred, instead of pixels / tile): the finalize launch shrinks with it.
// The K order inside the MFMA chain and the epilogue arithmetic are those of igemm_ring_kernel / igemm_epilogue, so
// results are bit-identical to every other configuration (tests/test_gpu_ops.py::test_every_conv_kernel_configuration).
#pragma once
#include "igemm_ring_kernel.h"
#include "igemm_wave_epilogue.h"
#include <stdlib.h>

#ifndef LH_PW_NT
#define LH_PW_NT 0          // debug builds only (-DLH_PW_NT=1): A/B of the cache policy of the operand-row loads
#endif

// GATE (with STATS; lh_igemm_gated): the output is the gradient of an activation relu(BN(gx)) or of a residual tail relu(BN(gx) + r): the
// epilogue stores the ReLU-gated gradient and the statistics row holds { sum g, sum g * xhat } (igemm_wave_epilogue.h).  GATE = 2: r is a
// projection shortcut BN2(gx2), a second statistics row takes { sum g, sum g * xhat2 }.
template <typename T, int BM, int KC, int PT, bool STATS, int GATE = 0>
__global__ __launch_bounds__(256, 2) void igemm_pw_kernel(const IgemmArgs p) {
    static_assert(!GATE || STATS, "the gate leaves its sums in the statistics row");
    // two operand register sets (the rows of tile n + 1 are requested before tile n is multiplied) where the register
    // file holds them beside the accumulators and the statistics
    constexpr bool DB = KC <= 256 && !(BM == 256 && KC >= 128) && (2 * PT * (KC / 32) * 4 + (BM / 16) * PT * 4 + (STATS ? (BM / 64) * 16 : 0)) <= 176;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int ES = sizeof(T);
    static_assert(ES == 2, "16-bit element types only");
    constexpr int EPC = 8;
    constexpr int CT = BM / 16;                       // output-channel tiles per wave (a wave owns all BM channels)
    constexpr int KS = KC / 32;                       // MFMA K slices
    constexpr int NSP = KC / 64;                      // 128-byte sub-panels of the weight panel
    constexpr int SPB = BM * 128;                     // bytes of one sub-panel: [BM rows][128 bytes], slots swizzled in-row
    constexpr int PANEL = NSP * SPB;
    constexpr int SUBW = 64;                          // channels per epilogue sub-block (one 128-byte line per pixel)
    constexpr int NSB = BM / SUBW;
    constexpr int RS = SUBW * ES + 8;                 // staging row pitch: 34 dwords -> the ds_write_b64 of 16 pixels hit 32 banks once
    constexpr int STG = PT * 16 * RS;                 // staging bytes per wave
    constexpr int CST = PANEL + 4 * STG;              // per-channel constants: float sv[BM], bv[BM] (GATE: mean, invstd, scale, shift of the gated BatchNorm)
    constexpr int NI = PANEL / 1024;                  // LDS-DMA instructions that fill the panel
    static_assert(NI % 4 == 0 && NI / 4 <= 60, "panel fill: instructions per wave");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, pl = lane & 15;
    // grid = G * CB workgroups, G a multiple of 8: the channel blocks of one pixel stream sit on one XCD (ids b, b + 8 share an L2)
    const int CB = p.pw_cb, G = p.pw_g;
    const int b = blockIdx.x;
    const int cblk = (b >> 3) % CB;
    const int g = (b & 7) + 8 * ((b >> 3) / CB);

    // ---- weight panel: fetched once
    {
        const int prow_lim = (p.cout + 127) / 128 * 128;          // rows the pack holds (padding rows are zero)
#pragma unroll
        for (int j = 0; j < NI / 4; ++j) {
            const int inst = 4 * j + wave;
            const int s = inst / (BM / 8), r8 = inst % (BM / 8);
            const int row = r8 * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((row >> 1) & 7);
            const int grow = cblk * BM + row;
            const bool ok = grow < prow_lim && s * 64 < p.kpad;
            const unsigned char* src = ok ? p.w + ((long)grow * p.kpad + s * 64 + c * EPC) * ES : p.zero;
            if (!(LH_ABL & 4)) __builtin_amdgcn_global_load_lds((gbl_void_p)src, (lds_void_p)(smem + s * SPB + r8 * 1024), 16, 0, 0);
        }
        float* cst = reinterpret_cast<float*>(smem + CST);
        for (int c = tid; c < BM; c += 256) {
            const int gc = cblk * BM + c;
            if constexpr (GATE) {                         // a data gradient carries no bias / affine: the slots hold the gate's constants
                const int gk = gc < p.cout ? gc : p.cout - 1;
                cst[c] = p.gmean[gk];
                cst[BM + c] = p.ginv[gk];
                cst[2 * BM + c] = p.gmask ? 0.f : p.gscale[gk];
                cst[3 * BM + c] = p.gmask ? 0.f : p.gshift[gk];
                if constexpr (GATE == 2) { cst[4 * BM + c] = p.gmean2[gk]; cst[5 * BM + c] = p.ginv2[gk]; }
            } else {
                float sv = 1.f, bv = 0.f;
                if (gc < p.cout) {
                    if (p.bias) bv = p.bias[gc];
                    if (p.scale) { sv = p.scale[gc]; bv = bv * sv + p.shift[gc]; }
                }
                cst[c] = sv;
                cst[BM + c] = bv;
            }
        }
    }

    // ---- operand rows -> MFMA B fragments.  Lane (q, pl) holds chunk 4*kk + q (8 elements) of pixel pl of sub-tile j:
    //      the K assignment of igemm_ring_kernel's fragment reads, so the accumulation order is the same.
    const int M = p.M;
    const int ntile = (M + PT * 16 - 1) / (PT * 16);
    const int tstride = G * 4;
    const bool lin = p.sh == 1 && p.sw == 1 && p.hi == p.ho && p.wi == p.wo;
    const int hw = p.ho * p.wo;
    // Every lane issues every load (rows past the last pixel / chunks past the K run / tiles past the last one fetch the
    // 16-byte zero page instead): straight-line code, so the compiler's vmcnt bookkeeping is exact and the loads of the
    // next tile stay in flight across this tile's MFMAs and stores.
    auto loadB = [&](int tile, uint4 (&B)[PT][KS]) {
#pragma unroll
        for (int j = 0; j < PT; ++j) {
            const int m = (tile * PT + j) * 16 + pl;
            const bool ok = m < M;                              // tile >= ntile gives m >= M for every lane
            long pix = m;
            if (!lin) {
                const int mm = ok ? m : 0;
                const int n = mm / hw, rem = mm - n * hw;
                const int a = rem / p.wo, bb = rem - a * p.wo;
                pix = (long)(n * p.hi + a * p.sh) * p.wi + bb * p.sw;
            }
            const unsigned char* base = p.in + (pix * p.in_pix_stride + q * EPC) * ES;
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                const unsigned char* src = (ok && kk * 32 + q * EPC < p.k_run) ? base + kk * 64 : p.zero;
                if (LH_ABL & 4) B[j][kk] = uint4{(unsigned)(size_t)src, 1u, 2u, 3u};
                else if (LH_PW_NT) {                        // experiment: operand rows as non-temporal (read-once) loads
                    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
                    const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(src));
                    B[j][kk] = uint4{v[0], v[1], v[2], v[3]};
                } else B[j][kk] = *reinterpret_cast<const uint4*>(src);
            }
        }
    };
    int t = g * 4 + wave;
    uint4 B0[PT][KS], B1[DB ? PT : 1][DB ? KS : 1];
    loadB(t, B0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // A fragment of channel tile i, K slice kk: sub-panel kk / 2, 16-row group i, row pl, logical chunk 4 * (kk & 1) + q
    const unsigned aoff0 = pl * 128 + (((0 + q) ^ ((pl >> 1) & 7)) << 4);
    const unsigned aoff1 = pl * 128 + (((4 + q) ^ ((pl >> 1) & 7)) << 4);
    unsigned char* stg = smem + PANEL + wave * STG;
    const float* cst = reinterpret_cast<const float*>(smem + CST);

    float s1[STATS ? NSB : 1][EPC], s2[STATS ? NSB : 1][EPC], s3[GATE == 2 ? NSB : 1][EPC];
#pragma unroll
    for (int i = 0; i < (STATS ? NSB : 1); ++i)
#pragma unroll
        for (int e = 0; e < EPC; ++e) s1[i][e] = s2[i][e] = 0.f;
#pragma unroll
    for (int i = 0; i < (GATE == 2 ? NSB : 1); ++i)
#pragma unroll
        for (int e = 0; e < EPC; ++e) s3[i][e] = 0.f;

    auto tilework = [&](const int t, uint4 (&Bf)[PT][KS], auto&& prefetch) {
        f32x4 acc[CT][PT];
#pragma unroll
        for (int i = 0; i < CT; ++i)
#pragma unroll
            for (int j = 0; j < PT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // MFMA chain, channel tiles in groups of four: the A fragments of group n + 1 are requested before the MFMAs of
        // group n issue (two register sets); the scheduling fences keep the compiler from hoisting all CT * KS reads
        // to the top (that needs more registers than the file has)
        constexpr int NG = KS * (CT / 4);
        uint4 A[2][4];
        auto readA = [&](auto Gc, uint4 (&dst)[4]) {
            constexpr int grp = decltype(Gc)::value;
            constexpr int kk = grp / (CT / 4), i0 = (grp % (CT / 4)) * 4;
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (LH_ABL & 2) dst[u] = uint4{aoff0 + u, aoff1, 5u, 7u};
                else dst[u] = *reinterpret_cast<const uint4*>(smem + (kk >> 1) * SPB + (i0 + u) * 2048 + ((kk & 1) ? aoff1 : aoff0));
        };
        readA(ic<0>{}, A[0]);
        static_for<0, NG>([&](auto Gc) {
            constexpr int grp = decltype(Gc)::value;
            constexpr int kk = grp / (CT / 4), i0 = (grp % (CT / 4)) * 4;
            if constexpr (grp + 1 < NG) readA(ic<grp + 1>{}, A[(grp + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < PT; ++j) {
                    if (LH_ABL & 1) acc[i0 + u][j][0] += __builtin_bit_cast(float, A[grp & 1][u].x ^ Bf[j][kk].x);
                    else MmaR<T>::run(A[grp & 1][u], Bf[j][kk], acc[i0 + u][j]);
                }
            __builtin_amdgcn_sched_barrier(0);
        });
        prefetch();                                  // single register set: the next tile's rows fly under this epilogue

        if (LH_ABL & 8) {                             // no epilogue, every accumulator stays live
            float z = 0.f;
#pragma unroll
            for (int i = 0; i < CT; ++i)
#pragma unroll
                for (int j = 0; j < PT; ++j) z += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
            if (z == 123.456f) p.out[0] = 1;
            return;
        }
        const int m0 = t * PT * 16;
        wave_epilogue<T, BM, PT, STATS, GATE>(p, acc, stg, cst, cblk, lane, [&](int row) { const int m = m0 + row; return m < M ? (long)m : -1L; }, s1, s2, s3);
    };
    if constexpr (DB) {
        while (t < ntile) {
            loadB(t + tstride, B1);
            tilework(t, B0, [] {});
            t += tstride;
            if (t >= ntile) break;
            loadB(t + tstride, B0);
            tilework(t, B1, [] {});
            t += tstride;
        }
    } else {
        for (; t < ntile; t += tstride) tilework(t, B0, [&] { loadB(t + tstride, B0); });
    }

    if constexpr (STATS) {
        // one slab row per workgroup (rows = workgroups per channel block)
        __syncthreads();                                 // every wave is done with the panel
        wave_stats_row<BM, 4>(s1, s2, reinterpret_cast<float*>(smem), p.stats ? p.stats + (long)g * 2 * p.cout : nullptr, cblk, p.cout, tid);
        if constexpr (GATE == 2) {
            __syncthreads();                             // the first row has been read out of the scratch
            wave_stats_row<BM, 4>(s1, s3, reinterpret_cast<float*>(smem), p.stats2 + (long)g * 2 * p.cout, cblk, p.cout, tid);
        }
    }
}

// Workgroups of this instantiation a CU holds at once (registers, LDS), 1..4; 2 when no device can be asked.
template <typename T, int BM, int KC, int PT, bool STATS, int GATE = 0>
static int pw_occupancy() {
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 2;
    if (cached[dev]) return cached[dev];
    const int lds = lh_pw_lds_bytes(BM, KC, PT, GATE);
    const void* fn = reinterpret_cast<const void*>(&igemm_pw_kernel<T, BM, KC, PT, STATS, GATE>);
    if (lds > 64 * 1024 && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return 1;
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, 256, lds) != hipSuccess || n < 1) return 2;
    static const int cap = getenv("LH_PW_OCC") ? atoi(getenv("LH_PW_OCC")) : 4;
    cached[dev] = n > cap ? cap : n;
    return cached[dev];
}

template <typename T, int BM, int KC, int PT>
static int launch_pw(const IgemmArgs& a0, hipStream_t s) {
    IgemmArgs a = a0;
    const int gate = a.gx2 ? 2 : a.gx ? 1 : 0;
    const int occ = gate == 2 ? pw_occupancy<T, BM, KC, PT, true, 2>() : gate ? pw_occupancy<T, BM, KC, PT, true, 1>()
                  : a.stats ? pw_occupancy<T, BM, KC, PT, true>() : pw_occupancy<T, BM, KC, PT, false>();
    lh_pw_grid(BM, KC, PT, a.M, a.cout, occ, &a.pw_g, &a.pw_cb);
    const int lds = lh_pw_lds_bytes(BM, KC, PT, gate);
    dim3 grid(a.pw_g * a.pw_cb);
    if (gate == 2) hipLaunchKernelGGL((igemm_pw_kernel<T, BM, KC, PT, true, 2>), grid, dim3(256), lds, s, a);
    else if (gate) hipLaunchKernelGGL((igemm_pw_kernel<T, BM, KC, PT, true, 1>), grid, dim3(256), lds, s, a);
    else if (a.stats) hipLaunchKernelGGL((igemm_pw_kernel<T, BM, KC, PT, true>), grid, dim3(256), lds, s, a);
    else hipLaunchKernelGGL((igemm_pw_kernel<T, BM, KC, PT, false>), grid, dim3(256), lds, s, a);
    LH_LAUNCH_CHECK("igemm_pw launch");
    return LH_OK;
}
