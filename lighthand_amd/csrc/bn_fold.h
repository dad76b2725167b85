// The BatchNorm statistics fold and finalize arithmetic of the finalize kernels (bn.hip).
#pragma once
#include "common.h"

struct FinalizeArgs {
    const void* slab;            // [rows][2][c] floats (or doubles: the second level of a two-launch fold)
    int rows, count, c;
    const float* gamma;
    const float* beta;
    float* rmean;
    float* rvar;
    long long* nbt;
    float momentum, eps;
    float* scale;
    float* shift;
    float* smean;
    float* sinv;
};

// Row lane rl (of 16) of channel ch: the partial totals a (sums) and b (sums of squares, or g * xhat) over rows rl, rl + 16, ... of a
// [rows][2][c] slab, in the fold's fixed order (eight / four independent row groups in flight: the fold is a latency chain).
// ld(pointer) -> TI loads one slab word (plain, or sc1 where the slab was written by other workgroups of the SAME launch).
template <typename TI, typename LD>
__device__ __forceinline__ void slab_lane16(const TI* slab, int rows, int c, int ch, int rl, LD&& ld, double& a, double& b) {
    a = 0.0; b = 0.0;
    int r = rl;
    for (; r + 112 < rows; r += 128) {
        TI av[8], bv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { av[u] = ld(slab + ((long)(r + 16 * u) * 2) * c + ch); bv[u] = ld(slab + ((long)(r + 16 * u) * 2 + 1) * c + ch); }
        a += (((double)av[0] + (double)av[1]) + ((double)av[2] + (double)av[3])) + (((double)av[4] + (double)av[5]) + ((double)av[6] + (double)av[7]));
        b += (((double)bv[0] + (double)bv[1]) + ((double)bv[2] + (double)bv[3])) + (((double)bv[4] + (double)bv[5]) + ((double)bv[6] + (double)bv[7]));
    }
    for (; r + 48 < rows; r += 64) {
        const TI a0 = ld(slab + ((long)r * 2) * c + ch), b0 = ld(slab + ((long)r * 2 + 1) * c + ch);
        const TI a1 = ld(slab + ((long)(r + 16) * 2) * c + ch), b1 = ld(slab + ((long)(r + 16) * 2 + 1) * c + ch);
        const TI a2 = ld(slab + ((long)(r + 32) * 2) * c + ch), b2 = ld(slab + ((long)(r + 32) * 2 + 1) * c + ch);
        const TI a3 = ld(slab + ((long)(r + 48) * 2) * c + ch), b3 = ld(slab + ((long)(r + 48) * 2 + 1) * c + ch);
        a += ((double)a0 + (double)a1) + ((double)a2 + (double)a3);
        b += ((double)b0 + (double)b1) + ((double)b2 + (double)b3);
    }
    for (; r < rows; r += 16) {
        a += (double)ld(slab + ((long)r * 2) * c + ch);
        b += (double)ld(slab + ((long)r * 2 + 1) * c + ch);
    }
}

// Batch totals of channel ch -> scale = gamma * rsqrt(var + eps), shift = beta - mean * scale (returned), and -- when `write` --
// the stored results: scale / shift / saved mean / invstd, running statistics with momentum and the unbiased variance.
__device__ __forceinline__ void bn_finalize_channel(const FinalizeArgs& p, int ch, double s0, double s1, bool write, float& sc_out, float& sh_out) {
    const int count = p.count;
    const double mean = s0 / count;
    double var = s1 / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)p.eps));
    const float g = p.gamma ? p.gamma[ch] : 1.f, b = p.beta ? p.beta[ch] : 0.f;
    const float sc = g * invstd;
    const float sh = b - (float)mean * sc;
    sc_out = sc; sh_out = sh;
    if (!write) return;
    p.scale[ch] = sc;
    p.shift[ch] = sh;
    if (p.smean) p.smean[ch] = (float)mean;
    if (p.sinv) p.sinv[ch] = invstd;
    if (p.rmean) p.rmean[ch] = (1.f - p.momentum) * p.rmean[ch] + p.momentum * (float)mean;
    if (p.rvar) {
        const double unb = count > 1 ? var * ((double)count / (count - 1)) : var;
        p.rvar[ch] = (1.f - p.momentum) * p.rvar[ch] + p.momentum * (float)unb;
    }
}
