// Inference bottleneck as ONE launch (round 5): the stride-1 residual block of the first ResNet stage in eval mode,
//
//   out = relu( bn3(conv3( relu(bn2(conv2( relu(bn1(conv1(x))) ))) )) + residual )        (reference: src/modeling/simplebaseline/
//                                                                                           pose_resnet.py:61-99, Bottleneck.forward)
//
// with conv1 = 1x1 (CIN -> 64), conv2 = 3x3 / s1 / p1 (64 -> 64), conv3 = 1x1 (64 -> 256), every BatchNorm folded into a
// per-channel scale / shift (running statistics).  Launched one by one these three convolutions move 4.8 GB per block at
// BASELINE.json configs[4] (R50, 384 x 384, batch 256, fp16: the block's 1.2 GB input is read by conv1 and again as the
// residual, the 0.3 GB intermediates are written and read back); here the 64-channel intermediates never leave LDS.
// MEASURED (round 5, PMC): the launch still moves 4.4 GB through the fabric -- the input with its tile halo (1.27 x 1.2 GB), the
// input AGAIN as the residual ten microseconds later (by then evicted from the 4 MB L2: 55 % hit rate) and the 1.2 GB output --
// i.e. 8 % less than the three launches, not half; it runs 1 130 us per identity block against ~1 040 us for the three in the
// step, and +2 % on the C5 step comes from the two launch boundaries and the projection block.  Halving the traffic needs the
// tile's 128 KB of input held ON CHIP between conv1 and the residual add (64 registers per lane as conv1's B fragments).
//
//  * A persistent workgroup (8 waves, one per CU: 148 KB of LDS) walks over 16 x 16 output tiles.  Per tile:
//      P1  conv1 on the (16+2) x (16+2) haloed patch -> LDS patch [324 pixels][64 ch] (positions outside the image = 0: they are
//          conv2's zero padding, not relu(shift));
//      P2  conv2, the nine taps as nine offsets into that patch (conv3x3_direct_kernel.h's layout and swizzle); its output (bn2 +
//          ReLU) goes through a wave-private 4 KB of LDS: a wave writes the two image rows it owns and is their only reader;
//      P3  conv3 (+ bn3 + residual + ReLU) through the per-wave epilogue of igemm_wave_epilogue.h (full-line NHWC stores).
//  * EVERYTHING that comes from memory -- the input patch in 32-channel chunks, and the three weight tensors -- arrives through
//    ONE ring of four 25 KB stages filled by LDS-DMA, as a continuous stream of stages that runs ahead of the phases and
//    across tile boundaries:   per tile  CIN/32 x { input chunk [324][32] + W1 chunk [64][32] },  3 x { W2 taps 3g..3g+2 },
//    2 x { W3 rows 128h..128h+127 }.   One raw s_barrier per stage; counted vmcnt (every wave issues the same number of DMA
//    instructions per stage kind: surplus ones copy the zero page into a dump kilobyte).
//  * K order of every convolution = that of the tiled kernels (tap-major, K ascending in 32-element MFMA slices), the
//    epilogue arithmetic that of igemm_wave_epilogue.h: the block output is BIT-IDENTICAL to the three launches
//    (tests/test_gpu_ops.py::test_fused_inference_bottleneck_is_bit_identical).
#pragma once
#include "igemm_ring_kernel.h"
#include "igemm_wave_epilogue.h"
#include "conv3x3_direct_kernel.h"

// Debug-only ablation builds (timing experiments; results are garbage): -DLH_BNK_ABL=<bits>  1 / 2 / 4 = no MFMAs in conv1 / conv2 /
// conv3, 8 = no LDS-DMA, 16 = no output epilogue, 32 = no fragment reads in conv2.  Never set in the product build.
#ifndef LH_BNK_ABL
#define LH_BNK_ABL 0
#endif

// wave_epilogue (igemm_wave_epilogue.h) with the addend rows ALREADY IN REGISTERS: the residual of a tile is requested ahead of
// conv3's MFMAs, instead of being loaded -- and waited for, four times per tile -- inside the epilogue.
// Same arithmetic, same store pattern (every lane stores in every pass); ad[k] = the 16 bytes wave_epilogue would have loaded for
// staging row k * 8 + (lane >> 3), columns cblk * 128 + sb * 64 + (lane & 7) * 8.
template <typename T, typename PixFn>
__device__ __forceinline__ void bneck_epilogue_half(const IgemmArgs& p, f32x4 (&acc)[8][2], unsigned char* stg, const float* cst, const int cblk,
                                                    const int lane, PixFn&& pix, const uint4 (&adh)[2][4]) {
    constexpr int ES = sizeof(T), EPC = 8, SUBW = 64, RS = SUBW * ES + 8, PT = 2, BM = 128;
    const int q = lane >> 4, pl = lane & 15;
    const int rrow = lane >> 3, rch = lane & 7;
#pragma unroll
    for (int sb = 0; sb < 2; ++sb) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int i = sb * 4 + it;
            const int col = i * 16 + q * 4;
            const float4 sv = *reinterpret_cast<const float4*>(cst + col);
            const float4 bv = *reinterpret_cast<const float4*>(cst + BM + col);
#pragma unroll
            for (int j = 0; j < PT; ++j) {
                union { uint2 u; T e[4]; } pk;
                pk.e[0] = from_f<T>(acc[i][j][0] * sv.x + bv.x);
                pk.e[1] = from_f<T>(acc[i][j][1] * sv.y + bv.y);
                pk.e[2] = from_f<T>(acc[i][j][2] * sv.z + bv.z);
                pk.e[3] = from_f<T>(acc[i][j][3] * sv.w + bv.w);
                *reinterpret_cast<uint2*>(stg + (j * 16 + pl) * RS + (it * 16 + q * 4) * ES) = pk.u;
            }
        }
        const int col0 = cblk * BM + sb * SUBW + rch * EPC;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int row = k * 8 + rrow;
            const long opix = pix(row);
            const unsigned char* src = stg + row * RS + rch * 16;
            const uint2 lo = *reinterpret_cast<const uint2*>(src);
            const uint2 hi = *reinterpret_cast<const uint2*>(src + 8);
            uint4 u = uint4{lo.x, lo.y, hi.x, hi.y};
            float v[EPC], av[EPC];
            unpack16<T>(u, v);
            unpack16<T>(adh[sb][k], av);
#pragma unroll
            for (int e = 0; e < EPC; ++e) v[e] = fmaxf(v[e] + av[e], 0.f);
            u = pack16<T>(v);
            unsigned char* dst = opix >= 0 ? p.out + (opix * p.out_pix_stride + col0) * ES : p.dump + lane * 16;
            *reinterpret_cast<uint4*>(dst) = u;
        }
    }
}

struct BottleneckArgs {
    IgemmArgs p3;                 // what the shared wave epilogue reads: out, out_pix_stride, cout, relu, addend (= residual), zero, dump
    const unsigned char* x;       // block input  [n][h][w][cin]
    const unsigned char* w1;      // packs (lh_pack_weight layout): [64 pad 128][1][kpad1], [64 pad 128][9][64], [256][1][64]
    const unsigned char* w2;
    const unsigned char* w3;
    const float* s1; const float* b1; const float* s2; const float* b2; const float* s3; const float* b3;
    int n, h, w, cin, kpad1, grid;
};

// TH x 16 output tiles, NWAVE waves, NSLOT ring slots:  <16, 8, 4> = one workgroup per CU (148 KB of LDS);  <8, 4, 3> = 75 KB, TWO
// workgroups per CU, so that one's epilogue and DMA waits run under the other's MFMAs (at 1.5x the weight traffic per pixel).
template <typename T, int TH, int NWAVE, int NSLOT>
__device__ __forceinline__ void bottleneck_infer_body(const BottleneckArgs& a, unsigned char* smem) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int ES = sizeof(T);
    static_assert(ES == 2, "16-bit element types");
    static_assert(TH == 2 * NWAVE && (NWAVE == 4 || NWAVE == 8) && (NSLOT == 3 || NSLOT == 4), "a wave owns two image rows of the tile");
    constexpr int MID = 64, COUT = 256, TW = 16, PH = TH + 2, PW = 18, NPIX = PH * PW;
    constexpr int NG1 = (NPIX + 15) / 16;                         // pixel groups of conv1: 21 (16 x 16 tile) / 12 (8 x 16)
    constexpr int NGW = (NG1 + NWAVE - 1) / NWAVE;                // ... per wave: 3
    constexpr int PATCH = NPIX * MID * ES;                        // 41,472
    constexpr int IN_CH = NPIX * 64;                              // input chunk: 324 rows of 64 bytes (32 channels)
    constexpr int IN_INST = (IN_CH + 1023) / 1024;                // 21 LDS-DMA instructions (16 rows each; the last one 4 rows)
    constexpr int IN_PAD = IN_INST * 1024;                        // the last instruction's rows past the patch land in [IN_CH, IN_PAD): W1 sits behind them
    constexpr int SLOT = IN_PAD + 4096;                           // 25,600 (16 x 16) / 16,384 (8 x 16): >= TPS taps of W2, >= half of W3
    constexpr int TPS = SLOT / 8192;                              // W2 taps per stage: 3 / 2
    constexpr int NS2 = (9 + TPS - 1) / TPS;                      // W2 stages per tile: 3 / 5 (the last one of the 8 x 16 form holds one tap)
    constexpr int OFF_PATCH = 0, OFF_RING = PATCH, OFF_DUMP = OFF_RING + NSLOT * SLOT, OFF_CST = OFF_DUMP + 1024;
    // DMA instructions per wave and stage kind (every wave issues the same number: surplus ones copy the zero page to the dump KB)
    constexpr int NI1 = (IN_INST + 4 + NWAVE - 1) / NWAVE, NI2 = (TPS * 8 + NWAVE - 1) / NWAVE, NI3 = 16 / NWAVE;
    constexpr int RS = 64 * ES + 8, STG = 2 * 16 * RS;            // staging patch of the per-wave epilogue (32 rows x 64 channels)
    static_assert(NWAVE * STG <= PATCH && 32 * 128 <= STG, "the epilogue's staging and the wave's conv2 output rows live in the conv1 patch");

    const int tid = threadIdx.x, lane0 = tid & 63, lane = lane0;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int NK1 = a.cin / 32;                                   // conv1 K chunks = S1 stages per tile
    const int SPT = NK1 + NS2 + 2;                                // stages per tile
    const int H = a.h, W = a.w;
    const int tx_n = (W + TW - 1) / TW, ty_n = (H + TH - 1) / TH;
    const int ntile = a.n * ty_n * tx_n;
    const int G = a.grid, b = blockIdx.x;
    const int my_tiles = b < ntile ? (ntile - b + G - 1) / G : 0;
    const long total = (long)my_tiles * SPT;                      // stages of this workgroup's stream
    const long pixb = (long)a.cin * ES, rowb = (long)W * pixb, imgb = (long)H * rowb;

    // per-channel constants -> LDS: [s1 b1 | s2 b2 | s3 b3 of channels 0..127 | s3 b3 of channels 128..255]
    float* cst = reinterpret_cast<float*>(smem + OFF_CST);
    for (int c = tid; c < 64; c += 64 * NWAVE) {
        cst[c] = a.s1[c]; cst[64 + c] = a.b1[c]; cst[128 + c] = a.s2[c]; cst[192 + c] = a.b2[c];
    }
    for (int c = tid; c < COUT; c += 64 * NWAVE) {                // conv3: per half of 128 channels [scale | shift] (wave_epilogue<.., 128, ..>)
        cst[256 + (c >> 7) * 256 + (c & 127)] = a.s3[c];
        cst[256 + (c >> 7) * 256 + 128 + (c & 127)] = a.b3[c];
    }

    // ---- the stage stream: issue cursor
    long issued = 0;
    int i_tile = b, i_k = 0, i_slot = 0;                          // tile of the stage being issued, its index inside the tile
    int iy0 = 0, ix0 = 0;
    const unsigned char* ibase = a.x;
    auto tile_origin = [&](int tile, int& y0, int& x0, int& n) {
        n = tile / (ty_n * tx_n);
        const int rem = tile - n * (ty_n * tx_n);
        y0 = (rem / tx_n) * TH;
        x0 = (rem % tx_n) * TW;
    };
    {
        int n0;
        tile_origin(i_tile < ntile ? i_tile : 0, iy0, ix0, n0);
        ibase = a.x + n0 * imgb;
    }
    auto dma = [&](const unsigned char* src, unsigned dst) { if (!(LH_BNK_ABL & 8)) d3_lds_dma16(src, dst); };
    auto dump_dma = [&]() { dma(a.p3.zero, lds_base + OFF_DUMP); };
    auto issue = [&]() {
        const unsigned slot = lds_base + OFF_RING + i_slot * SLOT;
        int lane = lane0;
        asm volatile("" : "+v"(lane));                            // (the same for the DMA source addresses: not hoisted, not spilled)
        if (i_k < NK1) {                                          // S1: input chunk (rows of 64 B, 16 rows per instruction) + W1 chunk
#pragma unroll
            for (int j = 0; j < NI1; ++j) {
                const int inst = NWAVE * j + wave;
                if (inst < IN_INST) {
                    const int pp = inst * 16 + (lane >> 2);
                    const int c = (lane & 3) ^ ((pp >> 2) & 3);
                    const int py = pp / PW, px = pp - py * PW;
                    const int iy = iy0 - 1 + py, ix = ix0 - 1 + px;
                    const bool ok = (pp < NPIX) & ((unsigned)iy < (unsigned)H) & ((unsigned)ix < (unsigned)W);
                    const unsigned char* src = ok ? ibase + iy * rowb + ix * pixb + i_k * 64 + c * 16 : a.p3.zero;
                    dma(src, slot + inst * 1024);
                } else if (inst < IN_INST + 4) {
                    const int r = (inst - IN_INST) * 16 + (lane >> 2);
                    const int c = (lane & 3) ^ ((r >> 2) & 3);
                    dma(a.w1 + ((long)r * a.kpad1 + i_k * 32 + c * 8) * ES, slot + IN_PAD + (inst - IN_INST) * 1024);
                } else dump_dma();
            }
        } else if (i_k < NK1 + NS2) {                             // S2: TPS taps of W2, [tap][64 rows][128 B]
            const int g = i_k - NK1;
#pragma unroll
            for (int j = 0; j < NI2; ++j) {
                const int inst = NWAVE * j + wave;                // tap tt = inst / 8 of the stage, rows 8 (inst % 8) ..
                const int tt = inst >> 3, r = (inst & 7) * 8 + (lane >> 3);
                const int c = (lane & 7) ^ ((r >> 1) & 7);
                if (inst < TPS * 8 && TPS * g + tt < 9) dma(a.w2 + (((long)r * 9 + TPS * g + tt) * 64 + c * 8) * ES, slot + inst * 1024);
                else dump_dma();
            }
        } else {                                                  // S3: half of W3, [128 rows][128 B]
            const int hh = i_k - NK1 - NS2;
#pragma unroll
            for (int j = 0; j < NI3; ++j) {
                const int inst = NWAVE * j + wave;                // 0 .. 15: rows 8 inst ..
                const int r = inst * 8 + (lane >> 3);
                const int c = (lane & 7) ^ ((r >> 1) & 7);
                dma(a.w3 + ((long)(128 * hh + r) * 64 + c * 8) * ES, slot + inst * 1024);
            }
        }
        ++issued;
        if (++i_slot == NSLOT) i_slot = 0;
        if (++i_k == SPT) {
            i_k = 0;
            i_tile += G;
            int n0;
            tile_origin(i_tile < ntile ? i_tile : 0, iy0, ix0, n0);
            ibase = a.x + n0 * imgb;
        }
    };
    // DMA instructions per wave of the stage `k` positions behind the issue cursor (what may stay in flight)
    auto ni_of = [&](int k_in_tile) { return k_in_tile < NK1 ? NI1 : k_in_tile < NK1 + NS2 ? NI2 : NI3; };

    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int s0 = 0; s0 < NSLOT - 1; ++s0)
        if (issued < total) issue();

    // ---- consume cursor
    long done = 0;                                                // stages consumed so far
    int c_slot = 0;
    // stage `done` has landed for this wave: the stages issued after it may stay in flight (at most two, their instruction
    // counts depend on their kind); then the barrier: everyone's share has landed AND everyone is done with stage done - 1
    // The output stores of the per-wave epilogue (8 per half tile and wave: wave_epilogue stores from every lane in every pass)
    // are YOUNGER than the stages that were in flight when they were issued: the waits that follow leave them in flight too
    // (`stores` = how many of them may still be behind the stage waited for).
    auto stage_wait = [&](int k_in_tile, bool first_tile) {
        // stores of the two half-tile epilogues still behind the stage waited for (see above): the second W3 stage has the first
        // half's 8 behind it; the next tile's stages 0 / 1 / 2 have 16 / 16 / 8
        // ... and so are the 8 residual loads each half requests behind the barrier of its W3 stage (consumed by the half's epilogue,
        // but still counted among the operations younger than the stage waited for): 8 + 8 per half
        // (what is still behind stage k of the NEXT tile depends on how far the ring runs ahead: NSLOT - 1 stages)
        const int stores = k_in_tile == SPT - 1 ? 16 : first_tile ? 0
                           : NSLOT == 4 ? (k_in_tile < 2 ? 32 : k_in_tile == 2 ? 16 : 0) : (k_in_tile == 0 ? 32 : k_in_tile == 1 ? 16 : 0);
        const long ahead = issued - 1 - done;                     // NSLOT - 2 in the steady state, fewer at the end of the stream
        int allow = stores;
        if (ahead >= 1) allow += ni_of((k_in_tile + 1) % SPT);
        if (ahead >= 2) allow += ni_of((k_in_tile + 2) % SPT);
        if (allow > 32) allow = 32;                               // fewer than are really behind it: stricter, never wrong
        // vmcnt(allow): allow <= 8 + 16; a binary ladder of immediates (the count is wave-uniform)
#define LH_BN_WAIT(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
        switch (allow) {
            LH_BN_WAIT(0) LH_BN_WAIT(1) LH_BN_WAIT(2) LH_BN_WAIT(3) LH_BN_WAIT(4) LH_BN_WAIT(5) LH_BN_WAIT(6) LH_BN_WAIT(7) LH_BN_WAIT(8)
            LH_BN_WAIT(9) LH_BN_WAIT(10) LH_BN_WAIT(11) LH_BN_WAIT(12) LH_BN_WAIT(13) LH_BN_WAIT(14) LH_BN_WAIT(15) LH_BN_WAIT(16)
            LH_BN_WAIT(17) LH_BN_WAIT(18) LH_BN_WAIT(19) LH_BN_WAIT(20) LH_BN_WAIT(21) LH_BN_WAIT(22) LH_BN_WAIT(23) LH_BN_WAIT(24)
            LH_BN_WAIT(25) LH_BN_WAIT(26) LH_BN_WAIT(27) LH_BN_WAIT(28) LH_BN_WAIT(29) LH_BN_WAIT(30) LH_BN_WAIT(31) LH_BN_WAIT(32)
            default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        }
#undef LH_BN_WAIT
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (issued < total) issue();                              // into the slot of stage done - 1
    };
    auto stage_done = [&]() {
        ++done;
        if (++c_slot == NSLOT) c_slot = 0;
    };

    const int ng1 = NGW - (wave + NWAVE * (NGW - 1) >= NG1 ? 1 : 0);   // conv1 pixel groups of this wave: gq = wave, wave + NWAVE (, wave + 2 NWAVE)
    static_assert(NGW == 3, "three conv1 pixel groups per wave at most");
    const int q0 = lane >> 4, pl0 = lane & 15;
    for (int t = b; t < ntile; t += G) {
        int y0, x0, n;
        tile_origin(t, y0, x0, n);
        // the lane's fragment coordinates, made opaque once per tile: every LDS address below is tile-invariant, and hoisted out of
        // this loop (a hundred address registers) they would spill -- recomputing them per tile costs a few vector instructions
        int q = q0, pl = pl0;
        asm volatile("" : "+v"(q), "+v"(pl));
        // ================= P1: conv1 over the haloed patch, K in 32-channel stages
        f32x4 acc1[4][3];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) acc1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < NK1; ++k) {
            stage_wait(k, t == b);
            const unsigned char* st = smem + OFF_RING + c_slot * SLOT;
            uint4 A[4], B[3];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 16 * i + pl;
                A[i] = *reinterpret_cast<const uint4*>(st + IN_PAD + r * 64 + ((q ^ ((r >> 2) & 3)) << 4));
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int pp = 16 * (wave + NWAVE * j) + pl;
                if (j < ng1) B[j] = *reinterpret_cast<const uint4*>(st + pp * 64 + ((q ^ ((pp >> 2) & 3)) << 4));
                else B[j] = uint4{0u, 0u, 0u, 0u};
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    if ((j < 2 || ng1 == 3) && !(LH_BNK_ABL & 1)) MmaR<T>::run(A[i], B[j], acc1[i][j]);
            stage_done();
        }
        // conv1 epilogue: bn1 + ReLU -> patch [pp][64 ch] (swizzled by pp); positions outside the image are conv2's zero padding
        {
            unsigned char* patch = smem + OFF_PATCH;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                if (j >= ng1) continue;
                const int pp = 16 * (wave + NWAVE * j) + pl;
                const int py = pp / PW, px = pp - py * PW;
                const int iy = y0 - 1 + py, ix = x0 - 1 + px;
                const bool inside = ((unsigned)iy < (unsigned)H) & ((unsigned)ix < (unsigned)W);
                if (pp < NPIX) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int col = 16 * i + 4 * q;
                        const float4 sv = *reinterpret_cast<const float4*>(cst + col);
                        const float4 bv = *reinterpret_cast<const float4*>(cst + 64 + col);
                        union { uint2 u; T e[4]; } pk;
                        pk.e[0] = from_f<T>(fmaxf(to_f<T>(from_f<T>(acc1[i][j][0] * sv.x + bv.x)), 0.f));
                        pk.e[1] = from_f<T>(fmaxf(to_f<T>(from_f<T>(acc1[i][j][1] * sv.y + bv.y)), 0.f));
                        pk.e[2] = from_f<T>(fmaxf(to_f<T>(from_f<T>(acc1[i][j][2] * sv.z + bv.z)), 0.f));
                        pk.e[3] = from_f<T>(fmaxf(to_f<T>(from_f<T>(acc1[i][j][3] * sv.w + bv.w)), 0.f));
                        if (!inside) pk.u = uint2{0u, 0u};
                        const int chunk = col >> 3;                           // 16-byte slot of these four channels
                        *reinterpret_cast<uint2*>(patch + pp * 128 + ((chunk ^ ((pp >> 1) & 7)) << 4) + (col & 7) * ES) = pk.u;
                    }
                }
            }
        }
        // ================= P2: conv2, three taps per stage; the wave owns image rows 2 wave, 2 wave + 1 of the tile
        f32x4 acc2[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int oy0 = y0 + 2 * wave;
        auto out_pix = [&](int row) {
            const int y = oy0 + (row >> 4), x = x0 + (row & 15);
            return (y < H && x < W) ? ((long)n * H + y) * W + x : -1L;
        };
        for (int g = 0; g < NS2; ++g) {
            stage_wait(NK1 + g, t == b);                          // (g == 0: also "every wave's part of the patch is written")
            const unsigned char* st = smem + OFF_RING + c_slot * SLOT;
            const unsigned char* patch = smem + OFF_PATCH;
            // 2 TPS steps (tap, K slice) per stage; the fragments of step n + 1 are requested before the MFMAs of step n issue
            // (two fragment register sets: without it every step exposed its LDS round trip -- 38 % of the wave cycles parked)
            const int nstep = 2 * (9 - TPS * g < TPS ? 9 - TPS * g : TPS);
            uint4 A[2][4], B[2][2];
            auto rd2 = [&](int step, uint4 (&av)[4], uint4 (&bv)[2]) {
                const int tt = step >> 1, kk = step & 1;
                const int tap = TPS * g + tt;                     // (dy, dx) of the 3 x 3 window, row-major
                const int dy = tap / 3, dx = tap - 3 * dy;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = 16 * i + pl;
                    av[i] = *reinterpret_cast<const uint4*>(st + tt * 8192 + r * 128 + (((4 * kk + q) ^ ((r >> 1) & 7)) << 4));
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int pp = (2 * wave + j + dy) * PW + dx + pl;
                    bv[j] = *reinterpret_cast<const uint4*>(patch + pp * 128 + (((4 * kk + q) ^ ((pp >> 1) & 7)) << 4));
                }
            };
            rd2(0, A[0], B[0]);
#pragma unroll
            for (int step = 0; step < 2 * TPS; ++step) {
                if (step < nstep) {
                    if (step + 1 < nstep) rd2(step + 1, A[(step + 1) & 1], B[(step + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            if (!(LH_BNK_ABL & 2)) MmaR<T>::run(A[step & 1][i], B[step & 1][j], acc2[i][j]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            stage_done();
        }
        // ================= P3: conv3 in two halves of 128 output channels (one W3 stage each); B fragments = this wave's mid2 rows.
        // Each half goes straight through bn3 + residual + ReLU and out (64 accumulator registers instead of 128); the staging
        // patch of the epilogue = the conv1 patch, which every wave left before the barrier of the first W3 stage.
        uint4 Bm[2][2];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            stage_wait(NK1 + NS2 + hh, t == b);
            const unsigned char* st = smem + OFF_RING + c_slot * SLOT;
            if (hh == 0) {
                // conv2 epilogue, behind the barrier that ends conv2 for EVERY wave (the patch is dead now): bn2 + ReLU -> this
                // wave's 32 pixels x 64 channels in its own slice of the patch (the slice its output epilogue stages in later),
                // read straight back as conv3's B fragments -- wave-private, no further barrier
                unsigned char* mid = smem + OFF_PATCH + wave * STG;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int col = 16 * i + 4 * q, r = 16 * j + pl;
                        const float4 sv = *reinterpret_cast<const float4*>(cst + 128 + col);
                        const float4 bv = *reinterpret_cast<const float4*>(cst + 192 + col);
                        union { uint2 u; T e[4]; } pk;
                        pk.e[0] = from_f<T>(fmaxf(to_f<T>(from_f<T>(acc2[i][j][0] * sv.x + bv.x)), 0.f));
                        pk.e[1] = from_f<T>(fmaxf(to_f<T>(from_f<T>(acc2[i][j][1] * sv.y + bv.y)), 0.f));
                        pk.e[2] = from_f<T>(fmaxf(to_f<T>(from_f<T>(acc2[i][j][2] * sv.z + bv.z)), 0.f));
                        pk.e[3] = from_f<T>(fmaxf(to_f<T>(from_f<T>(acc2[i][j][3] * sv.w + bv.w)), 0.f));
                        *reinterpret_cast<uint2*>(mid + r * 128 + (((col >> 3) ^ ((r >> 1) & 7)) << 4) + (col & 7) * ES) = pk.u;
                    }
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int r = 16 * j + pl;
                        Bm[kk][j] = *reinterpret_cast<const uint4*>(mid + r * 128 + (((4 * kk + q) ^ ((r >> 1) & 7)) << 4));
                    }
            }
            // the residual rows of this half (this wave's 32 pixels x 128 channels): 8 loads requested here, ahead of conv3's MFMAs,
            // instead of inside the epilogue (where each of its passes waited for its own) -- every lane loads (zero page outside)
            uint4 resid[2][4];
#pragma unroll
            for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const long opix = out_pix(k * 8 + (lane >> 3));
                    const long eoff = opix * a.p3.out_pix_stride + hh * 128 + sb * 64 + (lane & 7) * 8;
                    resid[sb][k] = *reinterpret_cast<const uint4*>(opix >= 0 ? a.p3.addend + eoff * ES : a.p3.zero);
                }
            f32x4 acc3[8][2];
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc3[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            // four groups of four weight fragments (K slice kk, rows 64 (g4 & 1) ..): the next group is requested before this group's MFMAs
            uint4 A3[2][4];
            auto rd3 = [&](int g4, uint4 (&av)[4]) {
                const int kk = g4 >> 1, i0 = (g4 & 1) * 4;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = 16 * (i0 + i) + pl;
                    av[i] = *reinterpret_cast<const uint4*>(st + r * 128 + (((4 * kk + q) ^ ((r >> 1) & 7)) << 4));
                }
            };
            rd3(0, A3[0]);
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                if (g4 + 1 < 4) rd3(g4 + 1, A3[(g4 + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        if (!(LH_BNK_ABL & 4)) MmaR<T>::run(A3[g4 & 1][i], Bm[g4 >> 1][j], acc3[(g4 & 1) * 4 + i][j]);
                __builtin_amdgcn_sched_barrier(0);
            }
            stage_done();
            unsigned char* stg = smem + OFF_PATCH + wave * STG;
            bneck_epilogue_half<T>(a.p3, acc3, stg, cst + 256 + hh * 256, hh, lane, out_pix, resid);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}

template <typename T, int TH, int NWAVE, int NSLOT>
__global__ __launch_bounds__(64 * NWAVE, 2) void bottleneck_infer_kernel(const BottleneckArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bottleneck_infer_body<T, TH, NWAVE, NSLOT>(a, smem);
}

static inline int lh_bottleneck_lds_bytes(int th, int nslot) {
    const int npix = (th + 2) * 18, in_inst = (npix * 64 + 1023) / 1024;
    return npix * 128 + nslot * (in_inst * 1024 + 4096) + 1024 + (128 + 128 + 512) * 4;
}
