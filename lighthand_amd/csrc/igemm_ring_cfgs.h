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
// The "wide-wave" form of the 256 x 256 tile: FOUR waves (one per SIMD), 128 x 128 per wave, up to 512 registers per
// lane (256 of them accumulators): half the LDS fragment reads per MFMA of the 8-wave form.  Same K order, same
// epilogue: bit-identical results.  In RingCfg / lh_igemm_desc.cfg its ring depth is written depth + LH_WIDE_DEPTH.
#define LH_WIDE_DEPTH 10
#define LH_RING_CFGS_WIDE(X) \
    X(256,256,2,2,3,64) X(256,256,2,2,4,64) X(256,256,2,2,2,128)
#define LH_RING_CFGS_F32(X) \
    X(128,64,4,1,2,64) X(128,64,4,1,4,64) X(64,128,1,4,2,64) X(64,128,1,4,4,64) \
    X(64,64,2,2,2,64) X(64,64,2,2,4,64)
