// Direct 3x3 convolution (stride 1, padding 1) for the small-channel layers: C_in = 32 or 64 per tap (the 3x3 of the first
// bottleneck stage, pose_resnet.py:66-72; HRNet's high-resolution branches, pose_hrnet.py:139-185), forward and data
// gradient (the data gradient of a 3x3 / s1 / p1 convolution is one too, with its own pack and tap order).
//
//   out[n][y][x][co] = sum_{tap = (dy, dx)} sum_k in[n][y + dy][x + dx][k] * wpack[co][tap][k]
//
// The tiled kernel (igemm_ring_kernel.h) gathers every input pixel row nine times -- once per tap -- and re-fetches the
// 9 * C * 64 weights for every 64-256 pixels: at C <= 64 it is bound by what a CU can ingest, at 4-5x its MFMA time.
// Here a workgroup is resident for the whole launch:
//  * the weights of its 64 output channels, [9 taps][64 rows][C], are fetched ONCE into LDS (74 KiB at C = 64);
//  * it walks over 16 x 16 output tiles; the (16 + 2) x (16 + 2) x C input patch of a tile is fetched ONCE by LDS-DMA
//    (zero padding = the zero page), double buffered: the patch of tile n + 1 is requested piece by piece between the
//    MFMA steps of tile n and lands under them -- 7x less ingest than nine gathered taps;
//  * the nine taps are nine OFFSETS into the patch: the B fragments of tap (dy, dx) are ds_read_b128 at pixel
//    (y + dy, x + dx); patch rows are 16-byte-slot swizzled by the patch pixel index, so the 16 consecutive pixels of a
//    fragment hit all 64 banks whatever the tap offset;
//  * eight waves (two per SIMD) share the panel and the patch; a wave owns two image rows of 16 pixels x 64 channels, and
//    writes them with the epilogue of igemm_wave_epilogue.h (full-line NHWC stores, BN statistics in registers, one slab row per
//    workgroup); its staging patch reuses the input patch the tile has finished with.
// Accumulation order = tap-major, K ascending in 32-element MFMA slices: the order of the tiled kernel, so the results
// are bit-identical to it (tests/test_gpu_ops.py::test_direct3x3_kernel_configurations).
#pragma once
#include "igemm_ring_kernel.h"
#include "igemm_wave_epilogue.h"
#include <stdlib.h>

// One LDS-DMA instruction as inline asm: 64 lanes x 16 bytes from `src` (per lane) to LDS address `lds` (wave-uniform) +
// 16 * lane.  Why not __builtin_amdgcn_global_load_lds: a ds_read the COMPILER sees after an LDS-DMA it cannot disambiguate
// from makes it wait for that DMA (s_waitcnt vmcnt(0)) -- in this kernel after every patch piece requested between the
// MFMA steps, the memory latency exposed six times per tile, and at the top of every tile for the previous tile's output
// stores.  With the DMA invisible to the compiler the kernel's own counted waits are the only ones (the compiler's counts
// for its own loads then under-count the operations in flight, which only makes its waits stronger).
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"      // m0 is a reserved register: it is exactly what the instruction reads, and nothing else in these kernels uses it
__device__ __forceinline__ void d3_lds_dma16(const unsigned char* src, unsigned lds) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds) : "memory", "m0");
}
#pragma clang diagnostic pop

// The kernel proper for workgroup `b` of its launch (plain launch: the block index; the mixed multi-problem launch of
// igemm_mixed_kernel.h: the index inside the problem the workgroup belongs to).  512 threads.
// GATE (with STATS; lh_igemm_gated): see igemm_wave_epilogue.h -- the constants area holds mean, invstd, scale, shift of the gated BatchNorm.
template <typename T, int C, bool STATS, bool GATE = false>
__device__ __forceinline__ void conv3x3_direct_body(const IgemmArgs& p, unsigned char* smem, const int b) {
    constexpr int ES = sizeof(T);
    static_assert(ES == 2 && (C == 32 || C == 64), "16-bit element types, 32 or 64 input channels per tap");
    constexpr int BM = 64, TH = 16, TW = 16, PH = TH + 2, PW = TW + 2;
    constexpr int NWAVE = 8;                          // two waves per SIMD: one wave's fragment reads / patch requests / epilogue run under its partner's MFMAs
    constexpr int ROWB = C * ES;                      // bytes of one pixel / one weight row of a tap: 64 or 128
    constexpr int SL = ROWB / 16;                     // 16-byte slots per row: 4 or 8
    constexpr int PPI = 1024 / ROWB;                  // rows (pixels) one LDS-DMA instruction covers: 16 or 8
    constexpr int KS = C / 32;                        // MFMA K slices per tap
    constexpr int WBYTES = 9 * BM * ROWB;             // weight panel
    constexpr int NPIX = PH * PW;                     // 324 patch pixels
    constexpr int NPI = (NPIX + PPI - 1) / PPI;       // LDS-DMA instructions per patch
    constexpr int NPW = (NPI + NWAVE - 1) / NWAVE;    // ... per wave
    constexpr int PATCH = NPIX * ROWB;                // bytes of one patch buffer (the lanes of the last instruction past it are masked off)
    constexpr int PT = TH / NWAVE;                    // 16-pixel tiles per wave: two image rows
    constexpr int RS = 64 * ES + 8;
    constexpr int STG = PT * 16 * RS;
    constexpr bool ALIAS = NWAVE * STG <= PATCH;          // C = 64: the waves' staging patches live in the input patch buffer the tile is done with
    constexpr int OFF_PATCH = WBYTES, OFF_STG = WBYTES + 2 * PATCH, OFF_CST = OFF_STG + (ALIAS ? 0 : NWAVE * STG);
    constexpr int NWT = WBYTES / 1024;                // weight-panel LDS-DMA instructions of the workgroup

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, pl = lane & 15;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int CB = p.pw_cb, G = p.pw_g;
    const int cblk = (b >> 3) % CB;
    const int g = (b & 7) + 8 * ((b >> 3) / CB);
    // swizzle: slot s of row r holds chunk s ^ f(r); f repeats every 16 rows and makes 16 consecutive rows x one chunk hit
    // all 64 banks (the scheme of igemm_ring_kernel.h: 64-byte rows f = r >> 2, 128-byte rows f = r >> 1)
    auto fswz = [](int r) { return SL == 8 ? ((r >> 1) & 7) : ((r >> 2) & 3); };

    // ---- weights [tap][64 rows][C]: once
    {
        const int prow_lim = (p.cout + 127) / 128 * 128;
#pragma unroll
        for (int j = 0; j < (NWT + NWAVE - 1) / NWAVE; ++j) {
            const int inst = NWAVE * j + wave;                    // covers PPI rows of one tap
            if (inst >= NWT) break;
            const int tap = inst / (BM / PPI), r0 = (inst % (BM / PPI)) * PPI;
            const int row = r0 + lane / SL;
            const int c = (lane % SL) ^ fswz(row);
            const int grow = cblk * BM + row;
            const unsigned char* src = grow < prow_lim ? p.w + (((long)grow * 9 + tap) * p.kpad + c * 8) * ES : p.zero;
            if (!(LH_ABL & 4)) d3_lds_dma16(src, lds_base + tap * (BM * ROWB) + r0 * ROWB);
        }
        float* cst = reinterpret_cast<float*>(smem + OFF_CST);
        for (int c = tid; c < BM; c += 64 * NWAVE) {
            const int gc = cblk * BM + c;
            if constexpr (GATE) {
                const int gk = gc < p.cout ? gc : p.cout - 1;
                cst[c] = p.gmean[gk];
                cst[BM + c] = p.ginv[gk];
                cst[2 * BM + c] = p.gmask ? 0.f : p.gscale[gk];
                cst[3 * BM + c] = p.gmask ? 0.f : p.gshift[gk];
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

    // ---- tiles: (image, tile row, tile column), tile column fastest
    const int H = p.ho, W = p.wo;
    const int tx_n = (W + TW - 1) / TW, ty_n = (H + TH - 1) / TH;
    const int ntile = p.n * ty_n * tx_n;
    // the patch pixels this lane fetches (instruction j of this wave: pixels [PPI * (NWAVE j + wave), + PPI)): position inside the
    // patch and chunk, both independent of the tile
    int ppy[NPW], ppx[NPW], pch[NPW];
#pragma unroll
    for (int j = 0; j < NPW; ++j) {
        const int pp = (NWAVE * j + wave) * PPI + lane / SL;
        ppy[j] = pp < NPIX ? pp / PW : -1;                        // past the patch: the lane takes no part
        ppx[j] = pp % PW;
        pch[j] = ((lane % SL) ^ fswz(pp)) * 16;
    }
    const long img_stride = (long)p.hi * p.wi * p.in_pix_stride * ES, row_stride = (long)p.wi * p.in_pix_stride * ES;
    const int pix_stride = p.in_pix_stride * ES;
    // piece j of the patch of `tile` (this wave's instruction j): (tile origin, image base) are computed by tile_origin
    const unsigned char* fbase = p.in;
    int fy0 = 0, fx0 = 0;
    bool flive = false;
    auto tile_origin = [&](int tile) {
        flive = tile < ntile;
        const int tt = flive ? tile : 0;
        const int n = tt / (ty_n * tx_n), rem = tt - n * (ty_n * tx_n);
        fy0 = (rem / tx_n) * TH - 1;
        fx0 = (rem % tx_n) * TW - 1;
        fbase = p.in + n * img_stride;
    };
    auto fetch_piece = [&](int j, int buf) {
        const int iy = fy0 + ppy[j], ix = fx0 + ppx[j];
        const bool ok = flive & ((unsigned)iy < (unsigned)p.hi) & ((unsigned)ix < (unsigned)p.wi);
        const unsigned char* src = ok ? fbase + iy * row_stride + (long)ix * pix_stride + pch[j] : p.zero;
        const unsigned dst = lds_base + OFF_PATCH + buf * PATCH + (NWAVE * j + wave) * 1024;
        if (!(LH_ABL & 4) && ppy[j] >= 0) d3_lds_dma16(src, dst);   // lanes past the patch: masked off
    };
    // fragment addresses.  A (weights): row = 16 i + pl of tap t: t * BM * ROWB + i * 16 * ROWB + pl * ROWB + ((c ^ f(pl)) << 4).
    // B (pixels): the wave's two image rows are patch rows 2 wave + j + dy (dy = 0..2), pixel column pl + dx.
    unsigned aoff[KS];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) aoff[kk] = pl * ROWB + ((((SL == 8 ? 4 * kk : 0) + q) ^ fswz(pl)) << 4);

    float s1[STATS ? 1 : 1][8], s2[STATS ? 1 : 1][8], s3[1][8];          // s3: the two-term gate of the pointwise kernel, unused here
#pragma unroll
    for (int e = 0; e < 8; ++e) s1[0][e] = s2[0][e] = 0.f;
    const float* cst = reinterpret_cast<const float*>(smem + OFF_CST);

    int t = g;
    tile_origin(t);
#pragma unroll
    for (int j = 0; j < NPW; ++j) fetch_piece(j, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // this wave's share of the weights and of the first patch has landed
    int buf = 0;
    for (; t < ntile; t += G, buf ^= 1) {
        // patch t: everyone's share has landed (counted vmcnt below / before the loop), and everyone is done with the other
        // buffer.  A RAW barrier: __syncthreads() waits for vmcnt(0), i.e. for the previous tile's output stores to retire.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        tile_origin(t + G);
        const unsigned char* patch = smem + OFF_PATCH + buf * PATCH;
        f32x4 acc[4][PT];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < PT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // 9 taps x KS slices; the fragments of step n + 1 are requested before the MFMAs of step n issue, and one piece of
        // the NEXT tile's patch is requested per step (an LDS-DMA costs ~100 issue cycles: spread, they hide under the MFMAs)
        constexpr int NSTEP = 9 * KS;
        static_assert(NPW <= NSTEP, "one patch piece per MFMA step");
        uint4 A[2][4], B[2][PT];
        auto rd = [&](int step, uint4 (&a)[4], uint4 (&bb)[PT]) {
            const int tap = step / KS, kk = step % KS;
            const int dy = p.dh[tap] + 1, dx = p.dw[tap] + 1;       // tap offset inside the patch: 0 .. 2
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (LH_ABL & 2) a[i] = uint4{aoff[kk] + i, 3u, 5u, 7u};
                else a[i] = *reinterpret_cast<const uint4*>(smem + tap * (BM * ROWB) + i * 16 * ROWB + aoff[kk]);
            }
#pragma unroll
            for (int j = 0; j < PT; ++j) {
                const int pp = (PT * wave + j + dy) * PW + dx + pl;
                const unsigned off = pp * ROWB + ((((SL == 8 ? 4 * kk : 0) + q) ^ fswz(pp)) << 4);
                if (LH_ABL & 2) bb[j] = uint4{off, 1u, 2u, 3u};
                else bb[j] = *reinterpret_cast<const uint4*>(patch + off);
            }
        };
        rd(0, A[0], B[0]);
#pragma unroll
        for (int step = 0; step < NSTEP; ++step) {
            if (step + 1 < NSTEP) rd(step + 1, A[(step + 1) & 1], B[(step + 1) & 1]);
            if (step < NPW) fetch_piece(step, buf ^ 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < PT; ++j) {
                    if (LH_ABL & 1) acc[i][j][0] += __builtin_bit_cast(float, A[step & 1][i].x ^ B[step & 1][j].x);
                    else MmaR<T>::run(A[step & 1][i], B[step & 1][j], acc[i][j]);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (LH_ABL & 8) {
            float z = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < PT; ++j) z += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
            if (z == 123.456f) p.out[0] = 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            continue;
        }
        if (ALIAS) {                                              // every wave is done reading patch t: its buffer becomes the staging area
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        unsigned char* stg = (ALIAS ? smem + OFF_PATCH + buf * PATCH : smem + OFF_STG) + wave * STG;
        const int n = t / (ty_n * tx_n), rem = t - n * (ty_n * tx_n);
        const int oy0 = (rem / tx_n) * TH + PT * wave, ox0 = (rem % tx_n) * TW;
        wave_epilogue<T, BM, PT, STATS, (GATE ? 1 : 0)>(p, acc, stg, cst, cblk, lane, [&](int row) {
            const int y = oy0 + (row >> 4), x = ox0 + (row & 15);
            return (y < H && x < W) ? ((long)n * H + y) * W + x : -1L;
        }, s1, s2, s3);
        // the next patch (requested between this tile's MFMAs) must have landed; this tile's 2 * PT stores -- the youngest
        // vector-memory operations of the wave, their count fixed by the dump-page rule of wave_epilogue -- stay in flight
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PT) : "memory");
    }
    if constexpr (STATS) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        wave_stats_row<BM, NWAVE>(s1, s2, reinterpret_cast<float*>(smem), p.stats ? p.stats + (long)g * 2 * p.cout : nullptr, cblk, p.cout, tid);
    }
}

template <typename T, int C, bool STATS, bool GATE = false>
__global__ __launch_bounds__(512, 2) void conv3x3_direct_kernel(const IgemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    conv3x3_direct_body<T, C, STATS, GATE>(p, smem, blockIdx.x);
}

static inline int lh_d3_lds_bytes(int c) {
    const int patch = 18 * 18 * c * 2, stg = 8 * 2 * 16 * 136;
    return 9 * 64 * c * 2 + 2 * patch + (stg <= patch ? 0 : stg) + 2 * 64 * 4;
}

// workgroups per channel block: every CU holds `occ` workgroups (LDS: one at C = 64, two at C = 32) for the whole launch
static inline void lh_d3_grid(int c, int n, int h, int w, int cout, int* G, int* CB) {
    const int cb = (cout + 63) / 64;
    const int occ = lh_d3_lds_bytes(c) <= 80 * 1024 ? 2 : 1;
    const long ntile = (long)n * ((h + 15) / 16) * ((w + 15) / 16);
    long g = 256L * occ / cb / 8 * 8;
    const long need = (ntile + 7) / 8 * 8;
    if (g > need) g = need;
    if (g < 8) g = 8;
    *G = (int)g;
    *CB = cb;
}

template <typename T, int C>
static int launch_d3(const IgemmArgs& a0, hipStream_t s) {
    IgemmArgs a = a0;
    lh_d3_grid(C, a.n, a.ho, a.wo, a.cout, &a.pw_g, &a.pw_cb);
    const bool gate = a.gx != nullptr;
    const int lds = lh_d3_lds_bytes(C) + (gate ? 2 * 64 * 4 : 0);            // the gate's two extra constant vectors
    const void* fn = gate ? reinterpret_cast<const void*>(&conv3x3_direct_kernel<T, C, true, true>)
                   : a.stats ? reinterpret_cast<const void*>(&conv3x3_direct_kernel<T, C, true>)
                             : reinterpret_cast<const void*>(&conv3x3_direct_kernel<T, C, false>);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) {
            lh_set_error("conv3x3_direct: cannot raise dynamic LDS to %d bytes: %s", lds, hipGetErrorString(e));
            return LH_ERR_HIP;
        }
    }
    dim3 grid(a.pw_g * a.pw_cb);
    if (gate) hipLaunchKernelGGL((conv3x3_direct_kernel<T, C, true, true>), grid, dim3(512), lds, s, a);
    else if (a.stats) hipLaunchKernelGGL((conv3x3_direct_kernel<T, C, true>), grid, dim3(512), lds, s, a);
    else hipLaunchKernelGGL((conv3x3_direct_kernel<T, C, false>), grid, dim3(512), lds, s, a);
    LH_LAUNCH_CHECK("conv3x3_direct launch");
    return LH_OK;
}
