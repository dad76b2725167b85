// Epilogue of the LDS-DMA convolution kernel (igemm_ring.hip), kept apart from its K loop: accumulators -> LDS tile
// [BP pixels][BM channels] (bias / per-channel affine applied on the fp32 accumulator), then full-line NHWC stores with
// optional addend / ReLU and the BN partial sums of the STORED values.
#pragma once
#include "common.h"
#include "igemm_args.h"

#ifndef LH_ABL
#define LH_ABL 0
#endif

// LDS bytes of the epilogue: the [BP][BM] tile (row pitch BM * ES + 8), the fused head's 32 weight rows where it exists,
// and the per-channel bias / scale / shift of the tile's BM channels (3 * BM floats).
template <typename T, int BM, int BP, bool HEAD> constexpr int lh_epi_lds_bytes() {
    return BP * (BM * (int)sizeof(T) + 8) + (HEAD ? 32 * (BM * (int)sizeof(T) + 16) : 0) + 3 * BM * 4;
}

// The tile's per-channel constants, fetched by the workgroup in ONE round trip into LDS at `cst` ([3][BM] floats: value to
// add, factor, -- both already combined as the epilogue applies them): every thread loads (index clamped, dropped by a
// select).  Per-lane loads under `channel < cout` conditions are branched around and waited for one by one: 2 * CT dependent
// L2 round trips per wave, 8 us of the 256 x 256 tile's epilogue in eval-mode plans (round 4).
template <int BM, int NT>
__device__ __forceinline__ void igemm_epilogue_consts(const IgemmArgs& p, float* cst, int c0, int tid) {
    const int cmax = p.cout - 1;
    for (int c = tid; c < BM; c += NT) {
        const int gc = c0 + c, gk = gc < cmax ? gc : cmax;
        const bool ok = gc < p.cout;
        const float b = p.bias ? p.bias[gk] : 0.f, sc = p.scale ? p.scale[gk] : 1.f, sh = p.scale ? p.shift[gk] : 0.f;   // wave-uniform conditions
        float bv = (p.bias && ok) ? b : 0.f, sv = 1.f;
        if (p.scale) {                      // out = acc * scale + shift (+ bias * scale folded by the host if both are given)
            sv = ok ? sc : 1.f;
            bv = bv * sv + (ok ? sh : 0.f);
        }
        cst[c] = sv;
        cst[BM + c] = bv;
    }
}

// LDS bytes the K-split wave pairs (KZ = 2, igemm_ring_kernel.h) need BEHIND the tile and its constants: one fp32 partial tile per pair
template <int BM, int BP, int WC, int WP> constexpr int lh_epi_ksplit_bytes() { return (BM / WC) * (BP / WP) * 4 * WC * WP; }

// KZ = 2: the workgroup holds WC x WP PAIRS of waves; both waves of a pair accumulated the same (BM / WC) x (BP / WP) sub-tile over
// alternate K slices.  Wave kz = 1 of a pair hands its partial sums to wave kz = 0 through LDS (fp32, one addition per element: the
// sum order differs from the KZ = 1 kernels' by exactly that), wave 0 writes the tile, all 64 * WC * WP * KZ threads store its rows.
template <typename T, int BM, int BP, int WC, int WP, int KZ = 1>
__device__ __forceinline__ void igemm_epilogue(const IgemmArgs& p, unsigned char* smem, f32x4 (&acc)[BM / WC / 16][BP / WP / 16],
                                               int pblk, int cblk, int tid, int lane, int wc, int wp, int hw,
                                               int ooh, int oow, float* stats, int kz = 0) {
    constexpr int ES = sizeof(T);
    constexpr int EPC = 16 / ES;
    constexpr int TC = BM / WC, TP = BP / WP;
    constexpr int CT = TC / 16, PT = TP / 16;
    constexpr int RS = BM * ES + 8;
    if (LH_ABL & 8) { if (acc[0][0][0] == 123.456f) p.out[0] = 1; return; }
    if (LH_ABL & 32) {                  // no epilogue, but every accumulator stays live (the K loop is not pruned)
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < BM / WC / 16; ++i)
#pragma unroll
            for (int j = 0; j < BP / WP / 16; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (t == 123.456f) p.out[0] = 1;
        return;
    }
    __syncthreads();
    float* cst = reinterpret_cast<float*>(smem + BP * RS);
    const bool affine = p.bias || p.scale;              // wave-uniform: training-mode forward / gradient launches carry neither
    if constexpr (KZ == 2) {
        f32x4* part = reinterpret_cast<f32x4*>(smem + BP * RS + 3 * BM * 4) + (wc * WP + wp) * (CT * PT * 64) + lane;
        if (kz == 1) {
#pragma unroll
            for (int i = 0; i < CT; ++i)
#pragma unroll
                for (int j = 0; j < PT; ++j) part[(i * PT + j) * 64] = acc[i][j];
        }
        if (affine) igemm_epilogue_consts<BM, 64 * WC * WP * KZ>(p, cst, cblk * BM, tid);
        __syncthreads();
        if (kz == 0) {
#pragma unroll
            for (int i = 0; i < CT; ++i)
#pragma unroll
                for (int j = 0; j < PT; ++j) acc[i][j] += part[(i * PT + j) * 64];
        }
    } else if (affine) {
        igemm_epilogue_consts<BM, 64 * WC * WP>(p, cst, cblk * BM, tid);
        __syncthreads();
    }
    if (KZ == 1 || kz == 0) {
        const int q = lane >> 4, pl = lane & 15;
#pragma unroll
        for (int i = 0; i < CT; ++i) {
            const int col = wc * TC + i * 16 + q * 4;
            float4 s4 = float4{1.f, 1.f, 1.f, 1.f}, b4 = float4{0.f, 0.f, 0.f, 0.f};
            if (affine) {
                s4 = *reinterpret_cast<const float4*>(cst + col);
                b4 = *reinterpret_cast<const float4*>(cst + BM + col);
            }
            const float sv[4] = {s4.x, s4.y, s4.z, s4.w}, bv[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int j = 0; j < PT; ++j) {
                const int pr = wp * TP + j * 16 + pl;
                T* dst = reinterpret_cast<T*>(smem + pr * RS + col * ES);
                if constexpr (ES == 4) {
                    reinterpret_cast<float2*>(dst)[0] = float2{acc[i][j][0] * sv[0] + bv[0], acc[i][j][1] * sv[1] + bv[1]};
                    reinterpret_cast<float2*>(dst)[1] = float2{acc[i][j][2] * sv[2] + bv[2], acc[i][j][3] * sv[3] + bv[3]};
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

    constexpr int CH = BM * ES / 16;
    constexpr int NT = 64 * WC * WP * KZ;             // threads of the workgroup
    constexpr int RPP = NT / CH;
    const int chunk = tid % CH, r0 = tid / CH;
    const int col0 = cblk * BM + chunk * EPC;
    const bool col_ok = col0 < p.cout;
    float s1[EPC], s2[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) s1[e] = s2[e] = 0.f;

    // Output pixel of row pr: m = pblk * BP + pr -> (image, row, column) -> placement in the (possibly strided / phased)
    // output.  A dense output (the common case) has opix == m; otherwise the decomposition is done ONCE per thread and
    // advanced by RPP rows per iteration with two conditional subtracts (integer divisions per row were most of the
    // epilogue's instruction count).
    const bool dense = p.osh == 1 && p.osw == 1 && p.OH == p.ho && p.OW == p.wo && ooh == 0 && oow == 0;
    int m = pblk * BP + r0;
    int pn = 0, pa = 0, pb = 0;
    const int qa = RPP / p.wo, step_b = RPP - qa * p.wo, step_n = qa / p.ho, step_a = qa - step_n * p.ho;   // wave-uniform
    if (!dense) {
        pn = m / hw;
        const int rem = m - pn * hw;
        pa = rem / p.wo;
        pb = rem - pa * p.wo;
    }
    // Pass 1: where every row of this thread goes, and -- when an addend rides in -- ALL its addend chunks and mask bytes
    // requested at once (one row's load at a time left the epilogue latency-bound at ~3.6 TB/s on the data gradients
    // that carry the identity-shortcut gradient); the accumulators are dead by now, the registers are free.
    constexpr int NR = BP / RPP;
    static_assert(BP % RPP == 0, "rows per thread");
    int opx[NR];                                      // output pixel of row k, -1: nothing to store
    uint4 ad[NR];
    uint4 xd[NR];                                     // BatchNorm-backward gate (lh_igemm_gated): the BN input at the output position
    unsigned mbits[NR];
    unsigned gbits[NR];
    // gate constants of this thread's EPC channels (every thread loads, index clamped)
    float gmean[EPC], ginv[EPC], gsc[EPC], gsh[EPC];
    if (p.gx) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            const int gk = col0 + e < p.cout ? col0 + e : p.cout - 1;
            gmean[e] = p.gmean[gk]; ginv[e] = p.ginv[gk];
            gsc[e] = p.gmask ? 0.f : p.gscale[gk]; gsh[e] = p.gmask ? 0.f : p.gshift[gk];
        }
    }
#pragma unroll
    for (int k = 0; k < NR; ++k, m += RPP) {
        int opix = m;
        if (!dense) {
            opix = (pn * p.OH + pa * p.osh + ooh) * p.OW + pb * p.osw + oow;
            pb += step_b; pa += step_a; pn += step_n;
            if (pb >= p.wo) { pb -= p.wo; ++pa; }
            if (pa >= p.ho) { pa -= p.ho; ++pn; }
        }
        opx[k] = (m < p.M && col_ok) ? opix : -1;
        ad[k] = uint4{0u, 0u, 0u, 0u};
        mbits[k] = 0xffu;
        if (p.addend) {                                  // wave-uniform; rows with nothing to store read the zero page (every lane loads)
            const bool ok = opx[k] >= 0;
            const long eoff = (long)((unsigned long)(unsigned)(ok ? opx[k] : 0) * (unsigned)p.out_pix_stride) + col0;
            ad[k] = *reinterpret_cast<const uint4*>(ok ? p.addend + eoff * ES : p.zero);
            if (p.addend_mask) mbits[k] = *(ok ? p.addend_mask + eoff / EPC : p.zero);    // addend = upstream gradient, gated by the activation's ReLU mask
        }
        xd[k] = uint4{0u, 0u, 0u, 0u};
        gbits[k] = 0xffu;
        if (p.gx) {
            const bool ok = opx[k] >= 0;
            const long eoff = (long)((unsigned long)(unsigned)(ok ? opx[k] : 0) * (unsigned)p.out_pix_stride) + col0;
            xd[k] = *reinterpret_cast<const uint4*>(ok ? p.gx + eoff * ES : p.zero);
            if (p.gmask) gbits[k] = *(ok ? p.gmask + eoff / EPC : p.zero);      // a residual tail: its sign was stored as mask bits
        }
    }
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        if (opx[k] < 0) continue;
        const int pr = r0 + k * RPP;
        const long eoff = (long)((unsigned long)(unsigned)opx[k] * (unsigned)p.out_pix_stride) + col0;
        const unsigned char* src = smem + pr * RS + chunk * 16;
        const uint2 lo = *reinterpret_cast<const uint2*>(src);
        const uint2 hi = *reinterpret_cast<const uint2*>(src + 8);
        uint4 u = uint4{lo.x, lo.y, hi.x, hi.y};
        if (p.addend || p.relu) {
            float v[EPC];
            unpack16<T>(u, v);
            if (p.addend) {
                float av[EPC];
                unpack16<T>(ad[k], av);
                if (p.addend_mask) {
#pragma unroll
                    for (int e = 0; e < EPC; ++e) av[e] = ((mbits[k] >> e) & 1u) ? av[e] : 0.f;
                }
#pragma unroll
                for (int e = 0; e < EPC; ++e) v[e] += av[e];
            }
            if (p.relu) {
#pragma unroll
                for (int e = 0; e < EPC; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            u = pack16<T>(v);
        }
        if (p.gx) {
            // the stored value is the gradient of relu(BN(x)): gate it with the activation's sign (recomputed from x as the
            // BN-backward kernels do, bn.hip fuse_bwd_reduce_flat_body<MASK_X>) and take its share of the BatchNorm-backward sums
            float g[EPC], xv[EPC];
            unpack16<T>(u, g);
            unpack16<T>(xd[k], xv);
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                g[e] = (p.gmask ? ((gbits[k] >> e) & 1u) != 0u : (xv[e] * gsc[e] + gsh[e]) > 0.f) ? g[e] : 0.f;
                s1[e] += g[e];
                s2[e] += g[e] * (xv[e] - gmean[e]) * ginv[e];
            }
            u = pack16<T>(g);
        } else if (stats) {
            float sv[EPC];
            unpack16<T>(u, sv);
#pragma unroll
            for (int e = 0; e < EPC; ++e) { s1[e] += sv[e]; s2[e] += sv[e] * sv[e]; }
        }
        if (!(LH_ABL & 16) || u.x == 0x12345678u) *reinterpret_cast<uint4*>(p.out + eoff * ES) = u;
    }

    if (stats) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            red[(r0 * 2 + 0) * BM + chunk * EPC + e] = s1[e];
            red[(r0 * 2 + 1) * BM + chunk * EPC + e] = s2[e];
        }
        __syncthreads();
        for (int t = tid; t < 2 * BM; t += NT) {
            const int which = t / BM, col = t - which * BM;
            float a = 0.f;
#pragma unroll 4
            for (int r = 0; r < RPP; ++r) a += red[(r * 2 + which) * BM + col];
            const int gc = cblk * BM + col;
            if (gc < p.cout) {
                float* dst = stats + ((long)pblk * 2 + which) * p.cout + gc;
                *dst = a;
            }
        }
    }
}

// Epilogue with the 1x1 head fused (C5: deconv + BN + ReLU + final_layer, pose_resnet.py:245-246): the [BP][BM] tile never
// leaves the CU.  Phase 1 as above plus the ReLU; then D[j][pixel] = sum_c head_w[j][c] * tile[pixel][c] on the MFMA
// (32 head rows x 32 pixels per wave, K = BM), bias, and fp32 stores straight into the NCHW heat-map.
template <typename T, int BM, int BP, int WC, int WP, typename MMA>
__device__ __forceinline__ void igemm_epilogue_head(const IgemmArgs& p, unsigned char* smem, f32x4 (&acc)[BM / WC / 16][BP / WP / 16],
                                                    int pblk, int tid, int lane, int wave, int wc, int wp, int hw, int ooh, int oow) {
    constexpr int ES = sizeof(T);
    constexpr int TC = BM / WC, TP = BP / WP;
    constexpr int CT = TC / 16, PT = TP / 16;
    constexpr int RS = BM * ES + 8;
    constexpr int HS = BM * ES + 16;                  // head-weight row pitch in LDS (16-byte aligned, banks shifted per row)
    constexpr int NT = 64 * WC * WP, NW = WC * WP;
    static_assert(ES == 2 && BP % (16 * NW) == 0 && BM % 32 == 0, "head epilogue: 16-bit types, whole pixel tiles per wave");
    unsigned char* hw_lds = smem + BP * RS;
    __syncthreads();
    float* cst = reinterpret_cast<float*>(hw_lds + 32 * HS);
    igemm_epilogue_consts<BM, NT>(p, cst, 0, tid);      // one round trip for the tile's per-channel constants (see above)
    __syncthreads();
    {
        const int q = lane >> 4, pl = lane & 15;
#pragma unroll
        for (int i = 0; i < CT; ++i) {
            const int col = wc * TC + i * 16 + q * 4;
            const float4 s4 = *reinterpret_cast<const float4*>(cst + col), b4 = *reinterpret_cast<const float4*>(cst + BM + col);
            const float sv[4] = {s4.x, s4.y, s4.z, s4.w}, bv[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int j = 0; j < PT; ++j) {
                const int pr = wp * TP + j * 16 + pl;
                union { uint2 u; T e[4]; } pk;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = acc[i][j][r] * sv[r] + bv[r];
                    if (p.relu) v = fmaxf(v, 0.f);
                    pk.e[r] = from_f<T>(col + r < p.cout ? v : 0.f);
                }
                *reinterpret_cast<uint2*>(smem + pr * RS + col * ES) = pk.u;
            }
        }
        // head weights: 32 rows of BM elements
        for (int c = tid; c < 32 * (BM * ES / 16); c += NT) {
            const int row = c / (BM * ES / 16), ch = c % (BM * ES / 16);
            uint4 v = uint4{0u, 0u, 0u, 0u};
            if (ch * (16 / ES) < p.cout) v = *reinterpret_cast<const uint4*>(p.head_w + (long)row * p.head_wstride + ch * 16);
            *reinterpret_cast<uint4*>(hw_lds + row * HS + ch * 16) = v;
        }
    }
    __syncthreads();
    constexpr int PW = BP / NW;                       // pixels of the tile per wave
    constexpr int NPT = PW / 16;
    f32x4 hacc[2][NPT];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NPT; ++b) hacc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int l15 = lane & 15, kg = lane >> 4;
#pragma unroll 2
    for (int ks = 0; ks < BM / 32; ++ks) {
        uint4 fa[2], fb[NPT];
#pragma unroll
        for (int a = 0; a < 2; ++a) fa[a] = *reinterpret_cast<const uint4*>(hw_lds + (a * 16 + l15) * HS + (ks * 32 + kg * 8) * ES);
#pragma unroll
        for (int b = 0; b < NPT; ++b) {
            const unsigned char* src = smem + (wave * PW + b * 16 + l15) * RS + (ks * 32 + kg * 8) * ES;
            const uint2 lo = *reinterpret_cast<const uint2*>(src), hi = *reinterpret_cast<const uint2*>(src + 8);
            fb[b] = uint4{lo.x, lo.y, hi.x, hi.y};
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < NPT; ++b) MMA::run(fa[a], fb[b], hacc[a][b]);
    }
    // lane (q = lane >> 4, pl = lane & 15) holds head rows a * 16 + q * 4 + r of pixel b * 16 + pl
    const long plane = (long)p.OH * p.OW;
#pragma unroll
    for (int b = 0; b < NPT; ++b) {
        const int m = pblk * BP + wave * PW + b * 16 + l15;
        if (m >= p.M) continue;
        const int n = m / hw, rem = m - n * hw;
        const int ya = rem / p.wo, xb = rem - ya * p.wo;
        const long opix = (long)(ya * p.osh + ooh) * p.OW + xb * p.osw + oow;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = a * 16 + kg * 4 + r;
                if (j < p.head_j) p.head_out[((long)n * p.head_j + j) * plane + opix] = hacc[a][b][r] + (p.head_bias ? p.head_bias[j] : 0.f);
            }
    }
}
