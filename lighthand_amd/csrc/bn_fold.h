// The BatchNorm statistics fold and finalize arithmetic, shared by the finalize kernels (bn.hip) and the convolution launch that
// carries its own finalize (lh_igemm_bn_relu, igemm_epilogue.h): ONE definition, so both produce the same bits.
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

// The same partial totals for FOUR consecutive channels ch0 .. ch0 + 3 (ch0 % 4 == 0, c % 4 == 0) of a slab other workgroups of THIS launch
// wrote with sc1 stores: 16-byte sc1 loads (L1 bypassed), all loads of a row group in flight before the first is used.  Per channel the
// additions are slab_lane16's, in its order: the totals are bit-identical.
__device__ __forceinline__ void slab_lane16_x4_sc1(const float* slab, int rows, int c, int ch0, int rl, double (&a)[4], double (&b)[4]) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef float f4 __attribute__((ext_vector_type(4)));
    auto ld = [](const float* q) { f4 v; asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(q) : "memory"); return v; };
    auto landed = []() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); };
#pragma unroll
    for (int e = 0; e < 4; ++e) { a[e] = 0.0; b[e] = 0.0; }
    int r = rl;
    for (; r + 112 < rows; r += 128) {
        f4 av[8], bv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { av[u] = ld(slab + ((long)(r + 16 * u) * 2) * c + ch0); bv[u] = ld(slab + ((long)(r + 16 * u) * 2 + 1) * c + ch0); }
        landed();
#pragma unroll
        for (int u = 0; u < 8; ++u) { asm volatile("" : "+v"(av[u])); asm volatile("" : "+v"(bv[u])); }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            a[e] += (((double)av[0][e] + (double)av[1][e]) + ((double)av[2][e] + (double)av[3][e])) + (((double)av[4][e] + (double)av[5][e]) + ((double)av[6][e] + (double)av[7][e]));
            b[e] += (((double)bv[0][e] + (double)bv[1][e]) + ((double)bv[2][e] + (double)bv[3][e])) + (((double)bv[4][e] + (double)bv[5][e]) + ((double)bv[6][e] + (double)bv[7][e]));
        }
    }
    for (; r + 48 < rows; r += 64) {
        f4 av[4], bv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { av[u] = ld(slab + ((long)(r + 16 * u) * 2) * c + ch0); bv[u] = ld(slab + ((long)(r + 16 * u) * 2 + 1) * c + ch0); }
        landed();
#pragma unroll
        for (int u = 0; u < 4; ++u) { asm volatile("" : "+v"(av[u])); asm volatile("" : "+v"(bv[u])); }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            a[e] += ((double)av[0][e] + (double)av[1][e]) + ((double)av[2][e] + (double)av[3][e]);
            b[e] += ((double)bv[0][e] + (double)bv[1][e]) + ((double)bv[2][e] + (double)bv[3][e]);
        }
    }
    {   // the last (at most three) rows of this lane: requested together, added one by one
        f4 av[3], bv[3];
        int nrest = 0;
#pragma unroll
        for (int u = 0; u < 3; ++u)
            if (r + 16 * u < rows) { av[u] = ld(slab + ((long)(r + 16 * u) * 2) * c + ch0); bv[u] = ld(slab + ((long)(r + 16 * u) * 2 + 1) * c + ch0); nrest = u + 1; }
        landed();
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            if (u < nrest) {
                asm volatile("" : "+v"(av[u])); asm volatile("" : "+v"(bv[u]));
#pragma unroll
                for (int e = 0; e < 4; ++e) { a[e] += (double)av[u][e]; b[e] += (double)bv[u][e]; }
            }
        }
    }
#endif
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
