// Kernel configurations of the LDS-DMA convolution kernel that are compiled in: X(BM, BP, WC, WP, D, KB) =
// tile (output channels x pixels), wave grid, ring depth, K bytes per stage.
// The per-launch choice among them is the caller's (lh_igemm_desc.cfg, filled by the plan's autotuner) or, with
// cfg = 0, the static heuristic of lh_ring_default_cfg.
#pragma once
#define LH_RING_CFGS_BIG(X) \
    X(256,256,2,4,3,64) X(256,256,2,4,4,64) X(256,256,2,4,2,128) X(128,256,2,2,3,64) \
    X(128,256,2,2,4,64) X(128,256,2,2,2,128) X(128,256,2,2,3,128)
#define LH_RING_CFGS_MID(X) \
    X(128,128,2,2,2,64) X(128,128,2,2,3,64) X(128,128,2,2,4,64) X(128,128,2,2,2,128) \
    X(128,128,2,2,3,128) X(128,128,2,2,4,128) X(128,64,4,1,2,64) X(128,64,4,1,4,64) \
    X(128,64,4,1,2,128) X(128,64,4,1,3,128) X(128,64,4,1,4,128) X(64,128,1,4,2,64) \
    X(64,128,1,4,4,64) X(64,128,1,4,2,128) X(64,128,1,4,3,128) X(64,128,1,4,4,128)
#define LH_RING_CFGS_SMALL(X) \
    X(64,64,2,2,2,64) X(64,64,2,2,4,64) X(64,64,2,2,2,128) X(64,64,2,2,3,128) \
    X(64,64,2,2,4,128)
#define LH_RING_CFGS_16BIT(X) LH_RING_CFGS_BIG(X) LH_RING_CFGS_MID(X) LH_RING_CFGS_SMALL(X)
// (Ring depth codes 10..19 belonged to the "wide-wave" form of the 256 x 256 tile -- four waves of 128 x 128 -- which was 20-25 % slower than the
//  8-wave tile on every launch, profiles/r04_c5_deconv_what_holds_the_pipe.txt, and was removed in round 6.)
#define LH_WIDE_DEPTH 10
// The "dense-wave" forms: EIGHT waves (two per SIMD) on the tiles the 4-wave forms run with one wave per SIMD.  A launch
// of <= 256 workgroups (stages 3-4 of the ResNets at batch 64) leaves every CU with ONE workgroup, and with one wave per
// SIMD the wave's own LDS-DMA issue (address arithmetic included), fragment reads and MFMAs run one after the other
// (ablation, 256-channel 3x3 at 16 x 16: LDS-DMA alone 19 us, complete 34 us, MFMA floor 9 us); with two waves per SIMD one
// wave's ingest runs under the other's MFMAs.  Half the per-wave tile: more fragment reads per MFMA, which the LDS has room
// for at these sizes.  Same K order, same epilogue: bit-identical results.  RingCfg depth = depth + LH_DENSE_DEPTH.
// (The list holds the forms the measured database chose at least three times; four more were tried and dropped in round 4, and in
//  round 5 the 8-wave 128 x 256 / 256 x 128 tiles -- 64 x 64 per wave, two thirds of the LDS reads per MFMA: chosen for 11 of 371 launches
//  by a fresh measurement, the step, C4 and C5 within 0.5 % of the shipped choices.)
#define LH_DENSE_DEPTH 20
#define LH_RING_CFGS_DENSE(X) \
    X(128,128,2,4,2,128) X(128,128,2,4,3,128) X(128,64,4,2,3,128) X(64,128,2,4,3,128) \
    X(64,64,2,4,2,128) X(64,64,2,4,4,128)
// The "K-split" forms (round 6): EIGHT waves as WC x WP pairs; the two waves of a pair multiply alternate 64-byte K slices of every 128-byte
// stage into the same (BM / WC) x (BP / WP) sub-tile and add their partial sums in the epilogue: two thirds of the LDS fragment reads per
// MFMA of the dense-wave forms at the same occupancy.  The accumulation order differs from every other form (results agree to fp32
// rounding, not bit for bit).  X(BM, BP, WC, WP, D, KB) with WC x WP = the PAIR grid; RingCfg depth = depth + LH_KSPLIT_DEPTH.
#define LH_KSPLIT_DEPTH 30
#define LH_RING_CFGS_KSPLIT(X) \
    X(128,128,2,2,2,128) X(128,128,2,2,3,128) X(128,128,2,2,4,128)
#define LH_RING_CFGS_F32(X) \
    X(128,64,4,1,2,64) X(128,64,4,1,4,64) X(64,128,1,4,2,64) X(64,128,1,4,4,64) \
    X(64,64,2,2,2,64) X(64,64,2,2,4,64)
