// BatchNorm statistics / finalize, the fused "sum of affine terms (+nearest upsample) + ReLU"
// elementwise op with its backward, and the 3x3/s2 max-pool -- all HBM-bound NHWC kernels:
// one 16-byte chunk (8 x 16-bit or 4 x fp32 channels) per lane, per-channel vectors in fp32.
#include "common.h"
#ifndef LH_BN_EXP_DEFAULT
#define LH_BN_EXP_DEFAULT 4      // measured (round 4): non-temporal loads of the BN inputs in the forward pass, -0.11 ms per R50 step
#endif
#include "multi.h"
#include "bn_fold.h"
#include <type_traits>
#include <vector>
#include <algorithm>
#include <stdlib.h>

// ------------------------------------------------------------------------------------------------
// Column sums of a [rows][ncol] slab in double: block = 16 columns x 16 row lanes.
template <typename TI>
__global__ __launch_bounds__(256) void colsum_kernel(const TI* src, int rows, int ncol, int rows_per_block,
                                                     double* dst) {
    __shared__ double red[16][17];
    const int col = blockIdx.x * 16 + (threadIdx.x & 15), rl = threadIdx.x >> 4;
    const int r0 = blockIdx.y * rows_per_block;
    int r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
    double a = 0.0;
    if (col < ncol)
        for (int r = r0 + rl; r < r1; r += 16) a += (double)src[(long)r * ncol + col];
    red[rl][threadIdx.x & 15] = a;
    __syncthreads();
    if (threadIdx.x < 16) {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) s += red[i][threadIdx.x];
        if (col < ncol) dst[(long)blockIdx.y * ncol + col] = s;
    }
}

// One workgroup = 16 channels x 16 row lanes: totals of the sum / sum-of-squares (or g / g*xhat) columns of a
// [rows][2][c] slab, handed to a per-channel functor by the first 16 threads (no second launch).
template <typename TI, typename F>
__device__ __forceinline__ void slab_totals_then(const TI* slab, int rows, int c, int bid, F&& fin) {
    __shared__ double red[2][16][17];
    const int ch = bid * 16 + (threadIdx.x & 15), rl = threadIdx.x >> 4;
    double a = 0.0, b = 0.0;
    if (ch < c) slab_lane16(slab, rows, c, ch, rl, [](const TI* q) { return *q; }, a, b);
    red[0][rl][threadIdx.x & 15] = a;
    red[1][rl][threadIdx.x & 15] = b;
    __syncthreads();
    if (threadIdx.x < 16 && ch < c) {
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) { s0 += red[0][i][threadIdx.x]; s1 += red[1][i][threadIdx.x]; }
        fin(ch, s0, s1);
    }
}

// Sum `rows` rows of `ncol` floats into totals[ncol] (double).  scratch: >= ceil(rows/256)*ncol doubles.
// The same totals with 4 channels x 64 row lanes per workgroup, for slabs of >= 256 rows: four times the workgroups and a
// quarter of the rows per lane (the fold kernels sit on the dependency chain of every BatchNorm: their length is a
// latency chain of row loads, 16 deep at 1 024 rows with 16 lanes, 4 deep with 64).  Lanes of a wave that share a channel
// fold by shuffles, the four waves through LDS, in a fixed order.
template <typename TI, typename F>
__device__ __forceinline__ void slab_totals_then64(const TI* slab, int rows, int c, int bid, F&& fin) {
    __shared__ double red64[2][4][4];
    const int ch = bid * 4 + (threadIdx.x & 3), rl = threadIdx.x >> 2;
    double a = 0.0, b = 0.0;
    if (ch < c) {
        int r = rl;
        for (; r + 448 < rows; r += 512) {        // eight independent row groups in flight (the fold is a latency chain)
            TI av[8], bv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { av[u] = slab[((long)(r + 64 * u) * 2) * c + ch]; bv[u] = slab[((long)(r + 64 * u) * 2 + 1) * c + ch]; }
            a += (((double)av[0] + (double)av[1]) + ((double)av[2] + (double)av[3])) + (((double)av[4] + (double)av[5]) + ((double)av[6] + (double)av[7]));
            b += (((double)bv[0] + (double)bv[1]) + ((double)bv[2] + (double)bv[3])) + (((double)bv[4] + (double)bv[5]) + ((double)bv[6] + (double)bv[7]));
        }
        for (; r + 192 < rows; r += 256) {        // four independent row groups in flight
            const TI a0 = slab[((long)r * 2) * c + ch], b0 = slab[((long)r * 2 + 1) * c + ch];
            const TI a1 = slab[((long)(r + 64) * 2) * c + ch], b1 = slab[((long)(r + 64) * 2 + 1) * c + ch];
            const TI a2 = slab[((long)(r + 128) * 2) * c + ch], b2 = slab[((long)(r + 128) * 2 + 1) * c + ch];
            const TI a3 = slab[((long)(r + 192) * 2) * c + ch], b3 = slab[((long)(r + 192) * 2 + 1) * c + ch];
            a += ((double)a0 + (double)a1) + ((double)a2 + (double)a3);
            b += ((double)b0 + (double)b1) + ((double)b2 + (double)b3);
        }
        for (; r < rows; r += 64) {
            a += (double)slab[((long)r * 2) * c + ch];
            b += (double)slab[((long)r * 2 + 1) * c + ch];
        }
    }
#pragma unroll
    for (int o = 4; o < 64; o <<= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane < 4) { red64[0][wave][lane] = a; red64[1][wave][lane] = b; }
    __syncthreads();
    if (threadIdx.x < 4 && ch < c) {
        const double s0 = ((red64[0][0][threadIdx.x] + red64[0][1][threadIdx.x]) + red64[0][2][threadIdx.x]) + red64[0][3][threadIdx.x];
        const double s1 = ((red64[1][0][threadIdx.x] + red64[1][1][threadIdx.x]) + red64[1][2][threadIdx.x]) + red64[1][3][threadIdx.x];
        fin(ch, s0, s1);
    }
}

constexpr int LH_FOLD_WIDE_ROWS = 256;        // slabs with at least this many rows use the 64-lane fold (grid = c / 4)
static inline int fold_grid(int rows, int c) { return rows >= LH_FOLD_WIDE_ROWS ? ceil_div(c, 4) : ceil_div(c, 16); }

static int column_totals(const float* slab, int rows, int ncol, double* scratch, double* totals, hipStream_t s) {
    const int gx = ceil_div(ncol, 16);
    if (rows <= 512) {
        hipLaunchKernelGGL((colsum_kernel<float>), dim3(gx, 1), dim3(256), 0, s, slab, rows, ncol, rows, totals);
    } else {
        const int gy = ceil_div(rows, 256);
        hipLaunchKernelGGL((colsum_kernel<float>), dim3(gx, gy), dim3(256), 0, s, slab, rows, ncol, 256, scratch);
        hipLaunchKernelGGL((colsum_kernel<double>), dim3(gx, 1), dim3(256), 0, s, (const double*)scratch, gy, ncol, gy, totals);
    }
    LH_LAUNCH_CHECK("colsum launch");
    return LH_OK;
}

// ------------------------------------------------------------------------------------------------
// Generic stats of a dense [m][c] activation (the conv kernels normally produce these slabs
// in their epilogue; this kernel serves tensors that did not come out of lh_igemm).
constexpr int STAT_ROWS = 128;

template <typename T>
__global__ __launch_bounds__(256) void bn_stats_kernel(const T* y, int m, int c, float* stats) {
    constexpr int EPC = 16 / sizeof(T);
    const int nchunk = c / EPC;
    const long r0 = (long)blockIdx.x * STAT_ROWS;
    float* out = stats + (long)blockIdx.x * 2 * c;
    for (int ch = threadIdx.x; ch < nchunk; ch += 256) {
        float s1[EPC], s2[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) s1[e] = s2[e] = 0.f;
        for (int r = 0; r < STAT_ROWS && r0 + r < m; ++r) {
            float v[EPC];
            unpack16<T>(*reinterpret_cast<const uint4*>(y + (r0 + r) * c + ch * EPC), v);
#pragma unroll
            for (int e = 0; e < EPC; ++e) { s1[e] += v[e]; s2[e] += v[e] * v[e]; }
        }
#pragma unroll
        for (int e = 0; e < EPC; ++e) { out[ch * EPC + e] = s1[e]; out[c + ch * EPC + e] = s2[e]; }
    }
}

extern "C" int lh_bn_stats_rows(int m, int c) { (void)c; return ceil_div(m, STAT_ROWS); }

extern "C" int lh_bn_stats(const void* y, int m, int c, float* stats, int* rows_out, int dtype, void* stream) {
    LH_REQUIRE(y && stats && m > 0 && c > 0, "lh_bn_stats: bad arguments");
    const int es = lh_dtype_size(dtype);
    LH_REQUIRE(es > 0 && c % (16 / es) == 0, "lh_bn_stats: c %d not a multiple of %d", c, 16 / (es > 0 ? es : 1));
    const int rows = ceil_div(m, STAT_ROWS);
    if (rows_out) *rows_out = rows;
    LH_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((bn_stats_kernel<T>), dim3(rows), dim3(256), 0, (hipStream_t)stream,
                                                   (const T*)y, m, c, stats));
    LH_LAUNCH_CHECK("bn_stats launch");
    return LH_OK;
}

__global__ void bn_finalize_kernel(const double* tot, int count, int c, const float* gamma, const float* beta,
                                   float* rmean, float* rvar, long long* nbt, float momentum, float eps,
                                   float* scale, float* shift, float* smean, float* sinv) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch == 0 && nbt) *nbt += 1;
    if (ch >= c) return;
    const double mean = tot[ch] / count;
    double var = tot[c + ch] / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float g = gamma ? gamma[ch] : 1.f, b = beta ? beta[ch] : 0.f;
    const float sc = g * invstd;
    scale[ch] = sc;
    shift[ch] = b - (float)mean * sc;
    if (smean) smean[ch] = (float)mean;
    if (sinv) sinv[ch] = invstd;
    if (rmean) rmean[ch] = (1.f - momentum) * rmean[ch] + momentum * (float)mean;
    if (rvar) {
        const double unb = count > 1 ? var * ((double)count / (count - 1)) : var;
        rvar[ch] = (1.f - momentum) * rvar[ch] + momentum * (float)unb;
    }
}

extern "C" size_t lh_bn_stats_slab_bytes(int rows, int c) {
    return ((size_t)rows * 2 * c + 2) * 4 + (size_t)(ceil_div(rows, 256) + 1) * 2 * c * 8;
}

template <typename TI>
__device__ __forceinline__ void bn_finalize_fused_body(const FinalizeArgs& p, const int bid, const int nblk) {
    if (bid == 0 && threadIdx.x == 0 && p.nbt) *p.nbt += 1;
    auto fin = [&](int ch, double s0, double s1) { float sc, sh; bn_finalize_channel(p, ch, s0, s1, true, sc, sh); };
    if (p.rows >= LH_FOLD_WIDE_ROWS) slab_totals_then64((const TI*)p.slab, p.rows, p.c, bid, fin);
    else slab_totals_then((const TI*)p.slab, p.rows, p.c, bid, fin);
}
template <typename TI>
__global__ __launch_bounds__(256) void bn_finalize_fused_kernel(const FinalizeArgs p) { bn_finalize_fused_body<TI>(p, blockIdx.x, gridDim.x); }
template <typename TI>
__global__ __launch_bounds__(256) void bn_finalize_fused_multi_kernel(const LhMulti<FinalizeArgs> m) {
    int bid, nblk;
    const int i = lh_multi_pick(m, bid, nblk);
    bn_finalize_fused_body<TI>(m.a[i], bid, nblk);
}

// stats: [rows][2][c] floats followed by scratch for (ceil(rows/256) + 1) * 2c doubles.
static FinalizeArgs finalize_args(const void* slab, int rows, int count, int c, const float* gamma, const float* beta, float* running_mean,
                                  float* running_var, long long* nbt, float momentum, float eps, float* scale, float* shift,
                                  float* save_mean, float* save_invstd) {
    FinalizeArgs a;
    a.slab = slab; a.rows = rows; a.count = count; a.c = c; a.gamma = gamma; a.beta = beta; a.rmean = running_mean; a.rvar = running_var;
    a.nbt = nbt; a.momentum = momentum; a.eps = eps; a.scale = scale; a.shift = shift; a.smean = save_mean; a.sinv = save_invstd;
    return a;
}

extern "C" int lh_bn_finalize(const float* stats, int rows, int count, int c, const float* gamma,
                              const float* beta, float* running_mean, float* running_var,
                              long long* num_batches_tracked, float momentum, float eps, float* scale,
                              float* shift, float* save_mean, float* save_invstd, void* stream) {
    LH_REQUIRE(stats && scale && shift && rows > 0 && count > 0 && c > 0, "lh_bn_finalize: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const long slab_floats = (long)rows * 2 * c;
    double* scratch = (double*)(stats + ((slab_floats + 1) & ~1L));
    if (rows <= 1024) {      // one launch: fold the slab and finalize
        const FinalizeArgs a = finalize_args(stats, rows, count, c, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps,
                                             scale, shift, save_mean, save_invstd);
        hipLaunchKernelGGL((bn_finalize_fused_kernel<float>), dim3(fold_grid(rows, c)), dim3(256), 0, s, a);
        LH_LAUNCH_CHECK("bn_finalize launch");
        return LH_OK;
    }
    // two launches: 256-row partial folds (fp64), then fold + finalize
    const int gy = ceil_div(rows, 256);
    hipLaunchKernelGGL((colsum_kernel<float>), dim3(ceil_div(2 * c, 16), gy), dim3(256), 0, s, stats, rows, 2 * c, 256, scratch);
    const FinalizeArgs a = finalize_args(scratch, gy, count, c, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps,
                                         scale, shift, save_mean, save_invstd);
    hipLaunchKernelGGL((bn_finalize_fused_kernel<double>), dim3(fold_grid(gy, c)), dim3(256), 0, s, a);
    LH_LAUNCH_CHECK("bn_finalize launch");
    return LH_OK;
}

// The same folds for up to n independent BatchNorm layers as ONE launch (multi.h); layers whose slab has more than 1024
// rows take the two-launch path of lh_bn_finalize one by one.
extern "C" int lh_bn_finalize_multi(const lh_bn_finalize_call* calls, int n, void* stream) {
    LH_REQUIRE(calls && n >= 1, "lh_bn_finalize_multi: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    LhMulti<FinalizeArgs> m;
    m.n = 0; m.first[0] = 0;
    auto flush = [&]() -> int {
        if (m.n == 0) return LH_OK;
        if (m.n == 1) hipLaunchKernelGGL((bn_finalize_fused_kernel<float>), dim3(m.first[1]), dim3(256), 0, s, m.a[0]);
        else hipLaunchKernelGGL((bn_finalize_fused_multi_kernel<float>), dim3(m.first[m.n]), dim3(256), 0, s, m);
        LH_LAUNCH_CHECK("bn_finalize_multi launch");
        m.n = 0;
        return LH_OK;
    };
    for (int i = 0; i < n; ++i) {
        const lh_bn_finalize_call& q = calls[i];
        LH_REQUIRE(q.stats && q.scale && q.shift && q.rows > 0 && q.count > 0 && q.c > 0, "lh_bn_finalize_multi: bad arguments (layer %d)", i);
        if (q.rows > 1024) {
            const int rc = lh_bn_finalize(q.stats, q.rows, q.count, q.c, q.gamma, q.beta, q.running_mean, q.running_var, q.num_batches_tracked,
                                          q.momentum, q.eps, q.scale, q.shift, q.save_mean, q.save_invstd, stream);
            if (rc) return rc;
            continue;
        }
        m.a[m.n] = finalize_args(q.stats, q.rows, q.count, q.c, q.gamma, q.beta, q.running_mean, q.running_var, q.num_batches_tracked,
                                 q.momentum, q.eps, q.scale, q.shift, q.save_mean, q.save_invstd);
        m.first[m.n + 1] = m.first[m.n] + fold_grid(q.rows, q.c);
        if (++m.n == LH_MULTI_MAX) { const int rc = flush(); if (rc) return rc; }
    }
    return flush();
}

__global__ void bn_eval_affine_kernel(const float* gamma, const float* beta, const float* rm, const float* rv,
                                      float eps, int c, float* scale, float* shift) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= c) return;
    const float sc = gamma[ch] / sqrtf(rv[ch] + eps);
    scale[ch] = sc;
    shift[ch] = beta[ch] - rm[ch] * sc;
}

extern "C" int lh_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean,
                                 const float* running_var, float eps, int c, float* scale, float* shift,
                                 void* stream) {
    LH_REQUIRE(gamma && beta && running_mean && running_var && scale && shift && c > 0, "lh_bn_eval_affine: bad arguments");
    hipLaunchKernelGGL(bn_eval_affine_kernel, dim3(ceil_div(c, 128)), dim3(128), 0, (hipStream_t)stream, gamma, beta,
                       running_mean, running_var, eps, c, scale, shift);
    LH_LAUNCH_CHECK("bn_eval_affine launch");
    return LH_OK;
}

// ------------------------------------------------------------------------------------------------
// Cache-policy / traversal experiments of the streaming BN passes (LH_BN_EXP, read once per process; speed only, results
// do not depend on it): bit 0 = non-temporal loads for the LAST-USE reads of the backward apply passes (dout, x, out),
// bit 1 = the apply passes walk the tensor back to front (what the reduce pass read last is re-read first), bit 2 =
// non-temporal loads of the BN inputs in the forward pass, bit 3 = the forward pass walks back to front (the tail of the
// convolution's output, written last, is read first).
static int bn_exp_flags() {
    static const int v = [] { const char* e = getenv("LH_BN_EXP"); return e ? atoi(e) : LH_BN_EXP_DEFAULT; }();
    return v;
}
typedef unsigned int lh_u32x4 __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ uint4 ld16(const unsigned char* p) {
    if constexpr (NT) {
        const lh_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const lh_u32x4*>(p));
        return uint4{v[0], v[1], v[2], v[3]};
    } else {
        return *reinterpret_cast<const uint4*>(p);
    }
}
// index of round r of a grid-stride walk over `rounds` rounds, front to back or back to front
__device__ __forceinline__ long walk_round(long r, long rounds, bool rev) { return rev ? rounds - 1 - r : r; }

struct FuseArgs {
    int exp;                     // bn_exp_flags() >> 2 (forward bits)
    const unsigned char* x[4];
    const float* scale[4];
    const float* shift[4];
    int log2up[4];
    int nterms, relu;
    unsigned char* out;
    unsigned char* mask;         // optional: one byte per 16-byte chunk of `out`, bit e = (element e > 0)
    int n, h, w, c;
    long total;                  // 16-byte chunks of `out` (flat kernels)
    const unsigned char* touch;  // optional (lh_fuse_desc.l2_touch): bytes the NEXT launch on the stream will read first -- its weight pack
    unsigned touch_bytes;
};

// L2 warm-up at the tail of an elementwise pass (round 5, profiles/r05_ingest_ladder.txt sitting 6): the convolution that follows on
// the stream walks its weight pack stage by stage in every workgroup at once, so each stage waits for lines no XCD has seen yet (the
// complete K loop of the stage-3 3x3: 21.3 us, 19.1 us with the pack already in L2).  L2 contents survive the kernel boundary: the
// workgroups that share an XCD (ids b, b + 8, ...) read one 4-byte word of every 128-byte line of the pack between them, right
// before they end.  Speed only: nothing depends on the values.
__device__ __forceinline__ void lh_l2_touch(const unsigned char* p, unsigned bytes, int bid, int nblk) {
    if (!p) return;
    const int per_xcd = nblk >> 3, idx = bid >> 3;
    if (idx >= per_xcd) return;
    const unsigned lines = bytes >> 7;
    unsigned acc = 0;
    for (unsigned l = (unsigned)idx * 256u + threadIdx.x; l < lines; l += (unsigned)per_xcd * 256u)
        acc ^= *reinterpret_cast<const unsigned*>(p + ((unsigned long)l << 7));
    asm volatile("" ::"v"(acc));
}

// bit e of the result = (stored element e > 0): computed from the ROUNDED values so that it equals `out > 0`
template <typename T> __device__ __forceinline__ unsigned char positive_bits(const uint4& u) {
    constexpr int EPC = 16 / sizeof(T);
    float r[EPC];
    unpack16<T>(u, r);
    unsigned m = 0;
#pragma unroll
    for (int e = 0; e < EPC; ++e) m |= (r[e] > 0.f ? 1u : 0u) << e;
    return (unsigned char)m;
}
template <int EPC> __device__ __forceinline__ void mask_by_bits(unsigned m, float* g) {
#pragma unroll
    for (int e = 0; e < EPC; ++e) g[e] = ((m >> e) & 1u) ? g[e] : 0.f;
}

template <typename T>
__device__ __forceinline__ void fuse_fwd_body(const FuseArgs& p, const int bid, const int nblk) {
    constexpr int EPC = 16 / sizeof(T);
    const int nchunk = p.c / EPC;
    const long total = (long)p.n * p.h * p.w * nchunk;
    for (long idx = (long)bid * 256 + threadIdx.x; idx < total; idx += (long)nblk * 256) {
        // 32-bit index arithmetic (plan_fuse_fwd checks total < 2^31): 64-bit divisions cost more than the rest of the loop
        const unsigned iu = (unsigned)idx;
        const unsigned pix = iu / (unsigned)nchunk;
        const int ch = (int)(iu - pix * (unsigned)nchunk);
        const unsigned t2 = pix / (unsigned)p.w;
        const int x = (int)(pix - t2 * (unsigned)p.w);
        const int n = (int)(t2 / (unsigned)p.h), y = (int)(t2 - (unsigned)n * (unsigned)p.h);
        float acc[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
        for (int t = 0; t < p.nterms; ++t) {
            const int l = p.log2up[t];
            const long sp = ((long)n * (p.h >> l) + (y >> l)) * (p.w >> l) + (x >> l);
            float v[EPC];
            unpack16<T>(*reinterpret_cast<const uint4*>(p.x[t] + (sp * p.c + ch * EPC) * sizeof(T)), v);
            if (p.scale[t]) {
                const float* sc = p.scale[t] + ch * EPC;
                const float* sh = p.shift[t] + ch * EPC;
#pragma unroll
                for (int e = 0; e < EPC; ++e) v[e] = v[e] * sc[e] + sh[e];
            }
#pragma unroll
            for (int e = 0; e < EPC; ++e) acc[e] += v[e];
        }
        if (p.relu) {
#pragma unroll
            for (int e = 0; e < EPC; ++e) acc[e] = fmaxf(acc[e], 0.f);
        }
        const uint4 u = pack16<T>(acc);
        *reinterpret_cast<uint4*>(p.out + idx * 16) = u;
        if (p.mask) p.mask[idx] = positive_bits<T>(u);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void fuse_fwd_kernel(const FuseArgs p) { fuse_fwd_body<T>(p, blockIdx.x, gridDim.x); }
template <typename T>
__global__ __launch_bounds__(256) void fuse_fwd_multi_kernel(const LhMulti<FuseArgs> m) {
    int bid, nblk;
    const int i = lh_multi_pick(m, bid, nblk);
    fuse_fwd_body<T>(m.a[i], bid, nblk);
}

// EPC consecutive floats of a per-channel vector with 16-byte loads.
template <int EPC> __device__ __forceinline__ void load_vec(const float* p, float* dst) {
#pragma unroll
    for (int q = 0; q < EPC / 4; ++q) {
        const float4 v = reinterpret_cast<const float4*>(p)[q];
        dst[4 * q] = v.x; dst[4 * q + 1] = v.y; dst[4 * q + 2] = v.z; dst[4 * q + 3] = v.w;
    }
}
template <int EPC> __device__ __forceinline__ void fill_vec(float* dst, float v) {
#pragma unroll
    for (int e = 0; e < EPC; ++e) dst[e] = v;
}

// Fast path: no upsampled term, <= 2 terms, channel chunks a power of two <= 256.  The grid stride is a multiple of
// the chunk count, so a thread keeps ONE channel chunk: its scale/shift live in registers and the loop has no
// integer division -- the kernel is a pure 16-byte-per-lane stream.
template <typename T, int NT>
__device__ __forceinline__ void fuse_fwd_flat_body(const FuseArgs& p, const int bid, const int nblk) {
    const long total = p.total;
    constexpr int EPC = 16 / sizeof(T);
    const int nchunk = p.c / EPC;
    const int ch = threadIdx.x & (nchunk - 1);
    float sc[NT][EPC], sh[NT][EPC];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (p.scale[t]) { load_vec<EPC>(p.scale[t] + ch * EPC, sc[t]); load_vec<EPC>(p.shift[t] + ch * EPC, sh[t]); }
        else { fill_vec<EPC>(sc[t], 1.f); fill_vec<EPC>(sh[t], 0.f); }
    }
    const long stride = (long)nblk * 256;
    const long rounds = (total + stride - 1) / stride;
    const bool rev = p.exp & 2;
    auto body = [&](auto NTc) __attribute__((always_inline)) {
    constexpr bool LNT = decltype(NTc)::value;
    for (long rr = 0; rr < rounds; ++rr) {
        const long idx = walk_round(rr, rounds, rev) * stride + (long)bid * 256 + threadIdx.x;
        if (idx >= total) continue;
        float acc[EPC];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float v[EPC];
            unpack16<T>(ld16<LNT>(p.x[t] + idx * 16), v);
            if (p.scale[t]) {
#pragma unroll
                for (int e = 0; e < EPC; ++e) v[e] = v[e] * sc[t][e] + sh[t][e];
            }
#pragma unroll
            for (int e = 0; e < EPC; ++e) acc[e] = t == 0 ? v[e] : acc[e] + v[e];
        }
        if (p.relu) {
#pragma unroll
            for (int e = 0; e < EPC; ++e) acc[e] = fmaxf(acc[e], 0.f);
        }
        const uint4 u = pack16<T>(acc);
        *reinterpret_cast<uint4*>(p.out + idx * 16) = u;
        if (p.mask) p.mask[idx] = positive_bits<T>(u);
    }
    };
    if (p.exp & 1) body(std::true_type{}); else body(std::false_type{});
    lh_l2_touch(p.touch, p.touch_bytes, bid, nblk);
}
template <typename T, int NT>
__global__ __launch_bounds__(256) void fuse_fwd_flat_kernel(const FuseArgs p) { fuse_fwd_flat_body<T, NT>(p, blockIdx.x, gridDim.x); }
template <typename T, int NT>
__global__ __launch_bounds__(256) void fuse_fwd_flat_multi_kernel(const LhMulti<FuseArgs> m) {
    int bid, nblk;
    const int i = lh_multi_pick(m, bid, nblk);
    fuse_fwd_flat_body<T, NT>(m.a[i], bid, nblk);
}

// (Round 4's channel-slice form -- BatchNorm finalize + apply as ONE launch for small tensors -- was measured slower than the two launches
//  (the fold every workgroup repeats costs more than the 5.4 us finalize launch it removes) and removed in round 6.)

// ---- launch records: a C-ABI call is first PLANNED into the kernel launches it consists of (kind, grid, argument block),
// then run -- one record as a plain launch, the records of several independent calls that agree in kind as one
// multi-problem launch (multi.h).
enum BnKind { K_FF_GEN, K_FF_FLAT1, K_FF_FLAT2, K_FB_REDUCE_GEN, K_FB_REDUCE_FLAT, K_FB_REDUCE_FLAT_X, K_FB_COEF, K_FB_APPLY_GEN,
              K_FB_APPLY_FLAT, K_FB_APPLY_FLAT_X, K_FB_APPLY2 };
struct FuseBwdArgs;
struct FuseBwd2Args;
struct CoefArgs;
struct BnLaunch;
static int bn_run(const BnLaunch* const* L, int n, int dtype, hipStream_t s);

// streaming grid of the flat passes: one workgroup per 4 x 256 chunks, at most 2048 workgroups (re-measured in round 3:
// 1024 / 4096 workgroups, 2 / 8 chunks per thread: 9.60-9.66 ms against 9.61, 8 chunks 9.72)
static int flat_grid(long total) {
    const long g = (total + 1023) / 1024;
    return (int)(g > 2048 ? 2048 : (g < 1 ? 1 : g));
}

// pre_fin (out): bit t set when term t carries a BatchNorm finalize (lh_fuse_desc.fin) that the planned kernel does NOT run
// itself -- the caller launches it first.
static int plan_fuse_fwd(const lh_fuse_desc* d, void* out, int n, int h, int w, int c, int dtype, FuseArgs* ap, int* kind, int* grid_out,
                         int* pre_fin = nullptr) {
    LH_REQUIRE(d && out && d->nterms >= 1 && d->nterms <= 4, "lh_fuse_fwd: bad descriptor");
    const int es = lh_dtype_size(dtype);
    LH_REQUIRE(es > 0 && c % (16 / es) == 0, "lh_fuse_fwd: c %d not a multiple of the 16-byte chunk", c);
    FuseArgs& a = *ap;
    for (int t = 0; t < 4; ++t) {
        a.x[t] = t < d->nterms ? (const unsigned char*)d->x[t] : nullptr;
        a.scale[t] = t < d->nterms ? d->scale[t] : nullptr;
        a.shift[t] = t < d->nterms ? d->shift[t] : nullptr;
        a.log2up[t] = t < d->nterms ? d->log2up[t] : 0;
        if (t < d->nterms) {
            LH_REQUIRE(a.x[t], "lh_fuse_fwd: term %d has no input", t);
            LH_REQUIRE(a.log2up[t] >= 0 && (h >> a.log2up[t]) << a.log2up[t] == h && (w >> a.log2up[t]) << a.log2up[t] == w,
                       "lh_fuse_fwd: %dx%d not divisible by 2^%d", h, w, a.log2up[t]);
            LH_REQUIRE((a.scale[t] == nullptr) == (a.shift[t] == nullptr), "lh_fuse_fwd: scale/shift must come together");
        }
    }
    a.nterms = d->nterms; a.relu = d->relu; a.out = (unsigned char*)out; a.mask = (unsigned char*)d->relu_mask; a.n = n; a.h = h; a.w = w; a.c = c;
    const long total = (long)n * h * w * (c / (16 / es));
    LH_REQUIRE(total < (1L << 31), "lh_fuse_fwd: tensor too large for 32-bit chunk indices");
    a.total = total;
    a.exp = bn_exp_flags() >> 2;
    a.touch = (const unsigned char*)d->l2_touch;
    a.touch_bytes = d->l2_touch && d->l2_touch_bytes < (1UL << 31) ? (unsigned)d->l2_touch_bytes : 0u;
    if (!a.touch_bytes) a.touch = nullptr;
    const int nchunk = c / (16 / es);
    bool flat = d->nterms <= 2 && (nchunk & (nchunk - 1)) == 0 && nchunk <= 256;
    for (int t = 0; t < d->nterms; ++t) flat = flat && a.log2up[t] == 0;
    int fin_mask = 0;
    for (int t = 0; t < d->nterms; ++t) {
        const lh_bn_finalize_call* f = d->fin[t];
        if (!f) continue;
        LH_REQUIRE(pre_fin, "lh_fuse_fwd: this entry point does not take a descriptor with a pending finalize");
        LH_REQUIRE(f->stats && f->rows > 0 && f->count > 0 && f->c == c && f->scale == d->scale[t] && f->shift == d->shift[t] && f->scale && f->shift,
                   "lh_fuse_fwd: term %d: the pending finalize must produce this term's scale / shift (c %d vs %d)", t, f->c, c);
        fin_mask |= 1 << t;
    }
    if (pre_fin) *pre_fin = fin_mask;
    if (flat) {
        *grid_out = flat_grid(total);      // >= 4 chunks per thread
        *kind = d->nterms == 1 ? K_FF_FLAT1 : K_FF_FLAT2;
    } else {
        *grid_out = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
        *kind = K_FF_GEN;
    }
    return LH_OK;
}

// ------------------------------------------------------------------------------------------------
// Backward of one term.  g = dout * (out > 0) summed over the term's 2^l x 2^l upsampling cell.
struct FuseBwdArgs {
    const unsigned char* dout;
    const unsigned char* out;
    const unsigned char* mask;   // optional ReLU mask bits written by lh_fuse_fwd (replaces reading `out`)
    const unsigned char* x;      // raw BN input of this term (null: identity)
    const float* scale;
    const float* mean;
    const float* invstd;
    unsigned char* dx;
    float* partial;              // [strips][2][c]
    const double* totals;        // [2][c]
    float* coef;                 // [2][c]: mean(g), mean(g*xhat)
    float* dgamma;
    float* dbeta;
    int n, h, w, c;              // OUTPUT resolution
    int l, relu, accumulate;
    int mask_from_x;             // ReLU mask recomputed from x*scale+shift (single BN term): `out` is not read
    const float* shift;
    int rows_per_strip;
    const unsigned char* touch;  // optional (lh_fuse_bwd_desc.l2_touch): the last apply launch of the call warms it in L2 (lh_l2_touch)
    unsigned touch_bytes;
    long count;                  // n * (h>>l) * (w>>l)
    long total;                  // 16-byte chunks of dx (flat apply kernel)
    int fold_rows;               // > 0: the flat apply pass folds partial[fold_rows][2][c] itself (no coefficient launch)
    int exp;                     // bn_exp_flags() & 3 (backward bits)
};

template <typename T, int EPC>
__device__ __forceinline__ void cell_grad(const FuseBwdArgs& p, int n, int ys, int xs, int chunk, float* g) {
#pragma unroll
    for (int e = 0; e < EPC; ++e) g[e] = 0.f;
    const int f = 1 << p.l;
    for (int dy = 0; dy < f; ++dy)
        for (int dx = 0; dx < f; ++dx) {
            const long pix = ((long)n * p.h + (ys << p.l) + dy) * p.w + (xs << p.l) + dx;
            const long off = (pix * p.c + chunk * EPC) * sizeof(T);
            float d[EPC];
            unpack16<T>(*reinterpret_cast<const uint4*>(p.dout + off), d);
            if (p.relu && p.mask) {
                mask_by_bits<EPC>(p.mask[off >> 4], d);
            } else if (p.relu) {
                float o[EPC];
                unpack16<T>(*reinterpret_cast<const uint4*>(p.out + off), o);
#pragma unroll
                for (int e = 0; e < EPC; ++e) d[e] = o[e] > 0.f ? d[e] : 0.f;
            }
#pragma unroll
            for (int e = 0; e < EPC; ++e) g[e] += d[e];
        }
}

template <typename T>
__device__ __forceinline__ void fuse_bwd_reduce_body(const FuseBwdArgs& p, const int bid, const int nblk) {
    constexpr int EPC = 16 / sizeof(T);
    __shared__ float red[256 * EPC * 2];
    const int nchunk = p.c / EPC;
    const int hs = p.h >> p.l, ws = p.w >> p.l;
    const long r0 = (long)bid * p.rows_per_strip;
    long r1 = r0 + p.rows_per_strip;
    if (r1 > p.count) r1 = p.count;
    float* out = p.partial + (long)bid * 2 * p.c;
    // active threads: a whole number of row lanes over the chunks (chunk fixed per thread)
    for (int cb = 0; cb < nchunk; cb += 256) {
        const int nc = nchunk - cb < 256 ? nchunk - cb : 256;
        const int lanes = 256 / nc;
        const int chunk = cb + (int)(threadIdx.x % nc), rl = threadIdx.x / nc;
        float s1[EPC], s2[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) s1[e] = s2[e] = 0.f;
        if (rl < lanes) {
            float mean[EPC], inv[EPC];
#pragma unroll
            for (int e = 0; e < EPC; ++e) { mean[e] = p.mean[chunk * EPC + e]; inv[e] = p.invstd[chunk * EPC + e]; }
            for (long r = r0 + rl; r < r1; r += lanes) {
                const unsigned t2 = (unsigned)r / (unsigned)ws;          // 32-bit: count < 2^31 (plan_fuse_bwd)
                const int xs = (int)((unsigned)r - t2 * (unsigned)ws);
                const int n = (int)(t2 / (unsigned)hs), ys = (int)(t2 - (unsigned)n * (unsigned)hs);
                float g[EPC], xv[EPC];
                cell_grad<T, EPC>(p, n, ys, xs, chunk, g);
                unpack16<T>(*reinterpret_cast<const uint4*>(p.x + (r * p.c + chunk * EPC) * sizeof(T)), xv);
#pragma unroll
                for (int e = 0; e < EPC; ++e) { s1[e] += g[e]; s2[e] += g[e] * (xv[e] - mean[e]) * inv[e]; }
            }
        }
#pragma unroll
        for (int e = 0; e < EPC; ++e) { red[(threadIdx.x * EPC + e) * 2] = s1[e]; red[(threadIdx.x * EPC + e) * 2 + 1] = s2[e]; }
        __syncthreads();
        for (int t = threadIdx.x; t < nc * EPC; t += 256) {
            const int cl = t / EPC, e = t % EPC;
            float a = 0.f, b = 0.f;
            for (int k = 0; k < lanes; ++k) { a += red[((k * nc + cl) * EPC + e) * 2]; b += red[((k * nc + cl) * EPC + e) * 2 + 1]; }
            out[(cb + cl) * EPC + e] = a;
            out[p.c + (cb + cl) * EPC + e] = b;
        }
        __syncthreads();
    }
}
template <typename T>
__global__ __launch_bounds__(256) void fuse_bwd_reduce_kernel(const FuseBwdArgs p) { fuse_bwd_reduce_body<T>(p, blockIdx.x, gridDim.x); }
template <typename T>
__global__ __launch_bounds__(256) void fuse_bwd_reduce_multi_kernel(const LhMulti<FuseBwdArgs> m) {
    int bid, nblk;
    const int i = lh_multi_pick(m, bid, nblk);
    fuse_bwd_reduce_body<T>(m.a[i], bid, nblk);
}

template <typename T>
__device__ __forceinline__ void fuse_bwd_apply_body(const FuseBwdArgs& p, const int bid, const int nblk) {
    constexpr int EPC = 16 / sizeof(T);
    const int nchunk = p.c / EPC;
    const int hs = p.h >> p.l, ws = p.w >> p.l;
    const long total = p.count * nchunk;
    for (long idx = (long)bid * 256 + threadIdx.x; idx < total; idx += (long)nblk * 256) {
        const unsigned iu = (unsigned)idx;                               // 32-bit: total < 2^31 (plan_fuse_bwd)
        const unsigned r = iu / (unsigned)nchunk;
        const int chunk = (int)(iu - r * (unsigned)nchunk);
        const unsigned t2 = r / (unsigned)ws;
        const int xs = (int)(r - t2 * (unsigned)ws);
        const int n = (int)(t2 / (unsigned)hs), ys = (int)(t2 - (unsigned)n * (unsigned)hs);
        float g[EPC];
        cell_grad<T, EPC>(p, n, ys, xs, chunk, g);
        if (p.x) {
            float xv[EPC];
            unpack16<T>(*reinterpret_cast<const uint4*>(p.x + idx * 16), xv);
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                const int ch = chunk * EPC + e;
                const float xh = (xv[e] - p.mean[ch]) * p.invstd[ch];
                g[e] = p.scale[ch] * (g[e] - p.coef[ch] - xh * p.coef[p.c + ch]);
            }
        }
        uint4* dst = reinterpret_cast<uint4*>(p.dx + idx * 16);
        if (p.accumulate) {
            float o[EPC];
            unpack16<T>(*dst, o);
#pragma unroll
            for (int e = 0; e < EPC; ++e) g[e] += o[e];
        }
        *dst = pack16<T>(g);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void fuse_bwd_apply_kernel(const FuseBwdArgs p) { fuse_bwd_apply_body<T>(p, blockIdx.x, gridDim.x); }
template <typename T>
__global__ __launch_bounds__(256) void fuse_bwd_apply_multi_kernel(const LhMulti<FuseBwdArgs> m) {
    int bid, nblk;
    const int i = lh_multi_pick(m, bid, nblk);
    fuse_bwd_apply_body<T>(m.a[i], bid, nblk);
}

// Coefficient fold INSIDE the apply pass (small tensors only, plan_fuse_bwd): every workgroup folds the reduce pass's
// [rows][2][c] partial sums itself -- rows * 2c <= 16 Ki floats, all of its loads in flight at once, fp64 sums in a fixed
// order, so every workgroup gets bit-identical coefficients -- and the separate fold launch (5-6 us on the dependency chain
// of every BatchNorm of HRNet's branches) disappears.  coefL[0..c) = mean(g), coefL[c..2c) = mean(g * xhat); workgroup 0
// also stores the parameter gradients dbeta = sum(g), dgamma = sum(g * xhat).
constexpr int LH_FOLD_IN_APPLY_FLOATS = 16384;
__device__ __forceinline__ void fold_coef_block(const float* slab, int rows, int c, long count, float* dgamma, float* dbeta,
                                                bool writer, float* coefL) {
    __shared__ double fold_part[512];
    const int ncol = 2 * c;                              // power of two, <= 512
    const int t = threadIdx.x;
    if (ncol <= 256) {
        const int parts = 256 / ncol, col = t & (ncol - 1), part = t / ncol;
        double a = 0.0;
        int r = part;
        for (; r + 7 * parts < rows; r += 8 * parts) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = slab[(long)(r + u * parts) * ncol + col];
#pragma unroll
            for (int u = 0; u < 8; ++u) a += (double)v[u];
        }
        for (; r < rows; r += parts) a += (double)slab[(long)r * ncol + col];
        fold_part[t] = a;
        __syncthreads();
        if (t < ncol) {
            double tot = 0.0;
            for (int q = 0; q < parts; ++q) tot += fold_part[q * ncol + t];
            coefL[t] = (float)(tot / (double)count);
            if (writer) {
                if (t < c) { if (dbeta) dbeta[t] = (float)tot; }
                else if (dgamma) dgamma[t - c] = (float)tot;
            }
        }
    } else {                                             // 2c = 512: two columns per thread, every row
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int col = t + 256 * h;
            double a = 0.0;
            int r = 0;
            for (; r + 8 <= rows; r += 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = slab[(long)(r + u) * ncol + col];
#pragma unroll
                for (int u = 0; u < 8; ++u) a += (double)v[u];
            }
            for (; r < rows; ++r) a += (double)slab[(long)r * ncol + col];
            coefL[col] = (float)(a / (double)count);
            if (writer) {
                if (col < c) { if (dbeta) dbeta[col] = (float)a; }
                else if (dgamma) dgamma[col - c] = (float)a;
            }
        }
    }
    __syncthreads();
}

// ---- l == 0 fast paths: dout / out / x / dx share one flat element offset, the thread keeps one channel chunk.
template <typename T, bool MASK_X>
__device__ __forceinline__ void fuse_bwd_reduce_flat_body(const FuseBwdArgs& p, const int bid, const int nblk) {
    constexpr int EPC = 16 / sizeof(T);
    __shared__ float red[256 * EPC * 2];
    const int nchunk = p.c / EPC;                       // power of two <= 256
    const int lanes = 256 / nchunk;
    const int chunk = threadIdx.x & (nchunk - 1), rl = threadIdx.x / nchunk;
    const long r0 = (long)bid * p.rows_per_strip;
    long r1 = r0 + p.rows_per_strip;
    if (r1 > p.count) r1 = p.count;
    float mean[EPC], inv[EPC], sc[EPC], sh[EPC], s1[EPC], s2[EPC];
    load_vec<EPC>(p.mean + chunk * EPC, mean);
    load_vec<EPC>(p.invstd + chunk * EPC, inv);
    if (MASK_X) { load_vec<EPC>(p.scale + chunk * EPC, sc); load_vec<EPC>(p.shift + chunk * EPC, sh); }
    fill_vec<EPC>(s1, 0.f);
    fill_vec<EPC>(s2, 0.f);
    const long rowb = (long)p.c * sizeof(T);
    // one row of this thread's channel chunk: gate the gradient, accumulate (rows in ascending order: the sums do not
    // depend on how many rows are fetched ahead)
    auto row = [&](const uint4& gd, const uint4& xd, unsigned mbits, const uint4& od) __attribute__((always_inline)) {
        float g[EPC], xv[EPC];
        unpack16<T>(gd, g);
        unpack16<T>(xd, xv);
        if (MASK_X) {
#pragma unroll
            for (int e = 0; e < EPC; ++e) g[e] = (xv[e] * sc[e] + sh[e]) > 0.f ? g[e] : 0.f;
        } else if (p.relu && p.mask) {
            mask_by_bits<EPC>(mbits, g);
        } else if (p.relu) {
            float o[EPC];
            unpack16<T>(od, o);
#pragma unroll
            for (int e = 0; e < EPC; ++e) g[e] = o[e] > 0.f ? g[e] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < EPC; ++e) { s1[e] += g[e]; s2[e] += g[e] * (xv[e] - mean[e]) * inv[e]; }
    };
    const bool bits = !MASK_X && p.relu && p.mask, outs = !MASK_X && p.relu && !p.mask;
    constexpr int UN = MASK_X ? 1 : 4;      // rows fetched ahead per thread (measured: the mask-from-x form, with 16 more registers, is faster without)
    long r = r0 + rl;
    for (; r + (UN - 1) * (long)lanes < r1; r += UN * (long)lanes) {
        uint4 gd[UN], xd[UN], od[UN];
        unsigned mb[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const long off = (r + u * (long)lanes) * rowb + chunk * 16;
            gd[u] = *reinterpret_cast<const uint4*>(p.dout + off);
            xd[u] = *reinterpret_cast<const uint4*>(p.x + off);
            mb[u] = bits ? p.mask[off >> 4] : 0u;
            od[u] = outs ? *reinterpret_cast<const uint4*>(p.out + off) : uint4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) row(gd[u], xd[u], mb[u], od[u]);
    }
    for (; r < r1; r += lanes) {
        const long off = r * rowb + chunk * 16;
        row(*reinterpret_cast<const uint4*>(p.dout + off), *reinterpret_cast<const uint4*>(p.x + off), bits ? p.mask[off >> 4] : 0u,
            outs ? *reinterpret_cast<const uint4*>(p.out + off) : uint4{0u, 0u, 0u, 0u});
    }
#pragma unroll
    for (int e = 0; e < EPC; ++e) { red[(threadIdx.x * EPC + e) * 2] = s1[e]; red[(threadIdx.x * EPC + e) * 2 + 1] = s2[e]; }
    __syncthreads();
    float* out = p.partial + (long)bid * 2 * p.c;
    for (int t = threadIdx.x; t < nchunk * EPC; t += 256) {
        const int cl = t / EPC, e = t % EPC;
        float a = 0.f, b = 0.f;
        for (int k = 0; k < lanes; ++k) { a += red[((k * nchunk + cl) * EPC + e) * 2]; b += red[((k * nchunk + cl) * EPC + e) * 2 + 1]; }
        out[cl * EPC + e] = a;
        out[p.c + cl * EPC + e] = b;
    }
}
template <typename T, bool MASK_X>
__global__ __launch_bounds__(256) void fuse_bwd_reduce_flat_kernel(const FuseBwdArgs p) { fuse_bwd_reduce_flat_body<T, MASK_X>(p, blockIdx.x, gridDim.x); }
template <typename T, bool MASK_X>
__global__ __launch_bounds__(256) void fuse_bwd_reduce_flat_multi_kernel(const LhMulti<FuseBwdArgs> m) {
    int bid, nblk;
    const int i = lh_multi_pick(m, bid, nblk);
    fuse_bwd_reduce_flat_body<T, MASK_X>(m.a[i], bid, nblk);
}

template <typename T, bool MASK_X>
__device__ __forceinline__ void fuse_bwd_apply_flat_body(const FuseBwdArgs& p, const int bid, const int nblk) {
    const long total = p.total;
    constexpr int EPC = 16 / sizeof(T);
    const int nchunk = p.c / EPC;
    const int chunk = threadIdx.x & (nchunk - 1);
    // dx = scale*(g - c0 - xhat*c1) = A*g + B*x + C  with  xhat = (x - mean)*invstd
    float A[EPC], B[EPC], Cc[EPC], sc[EPC], sh[EPC];
    if (p.x) {
        float iv[EPC], c0[EPC], c1[EPC], mn[EPC];
        load_vec<EPC>(p.scale + chunk * EPC, A);
        load_vec<EPC>(p.invstd + chunk * EPC, iv);
        if (p.fold_rows > 0) {
            __shared__ float coefL[512];
            fold_coef_block(p.partial, p.fold_rows, p.c, p.count, p.dgamma, p.dbeta, bid == 0, coefL);
#pragma unroll
            for (int e = 0; e < EPC; ++e) { c0[e] = coefL[chunk * EPC + e]; c1[e] = coefL[p.c + chunk * EPC + e]; }
        } else {
            load_vec<EPC>(p.coef + chunk * EPC, c0);
            load_vec<EPC>(p.coef + p.c + chunk * EPC, c1);
        }
        load_vec<EPC>(p.mean + chunk * EPC, mn);
#pragma unroll
        for (int e = 0; e < EPC; ++e) { B[e] = -A[e] * iv[e] * c1[e]; Cc[e] = -A[e] * c0[e] - B[e] * mn[e]; }
        if (MASK_X) {
#pragma unroll
            for (int e = 0; e < EPC; ++e) sc[e] = A[e];
            load_vec<EPC>(p.shift + chunk * EPC, sh);
        }
    } else { fill_vec<EPC>(A, 1.f); fill_vec<EPC>(B, 0.f); fill_vec<EPC>(Cc, 0.f); }
    const long stride = (long)nblk * 256;
    const long rounds = (total + stride - 1) / stride;
    const bool rev = p.exp & 2;
    auto body = [&](auto NTc) __attribute__((always_inline)) {
    constexpr bool LNT = decltype(NTc)::value;
    for (long rr = 0; rr < rounds; ++rr) {
        const long idx = walk_round(rr, rounds, rev) * stride + (long)bid * 256 + threadIdx.x;
        if (idx >= total) continue;
        const long off = idx * 16;
        float g[EPC], xv[EPC];
        unpack16<T>(ld16<LNT>(p.dout + off), g);
        if (p.x) unpack16<T>(ld16<LNT>(p.x + off), xv);
        if (MASK_X) {
#pragma unroll
            for (int e = 0; e < EPC; ++e) g[e] = (xv[e] * sc[e] + sh[e]) > 0.f ? g[e] : 0.f;
        } else if (p.relu && p.mask) {
            mask_by_bits<EPC>(p.mask[off >> 4], g);
        } else if (p.relu) {
            float o[EPC];
            unpack16<T>(ld16<LNT>(p.out + off), o);
#pragma unroll
            for (int e = 0; e < EPC; ++e) g[e] = o[e] > 0.f ? g[e] : 0.f;
        }
        if (p.x) {
#pragma unroll
            for (int e = 0; e < EPC; ++e) g[e] = A[e] * g[e] + B[e] * xv[e] + Cc[e];
        }
        uint4* dst = reinterpret_cast<uint4*>(p.dx + off);
        if (p.accumulate) {
            float o[EPC];
            unpack16<T>(*dst, o);
#pragma unroll
            for (int e = 0; e < EPC; ++e) g[e] += o[e];
        }
        *dst = pack16<T>(g);
    }
    };
    if (p.exp & 1) body(std::true_type{}); else body(std::false_type{});
    lh_l2_touch(p.touch, p.touch_bytes, bid, nblk);
}
template <typename T, bool MASK_X>
__global__ __launch_bounds__(256) void fuse_bwd_apply_flat_kernel(const FuseBwdArgs p) { fuse_bwd_apply_flat_body<T, MASK_X>(p, blockIdx.x, gridDim.x); }
template <typename T, bool MASK_X>
__global__ __launch_bounds__(256) void fuse_bwd_apply_flat_multi_kernel(const LhMulti<FuseBwdArgs> m) {
    int bid, nblk;
    const int i = lh_multi_pick(m, bid, nblk);
    fuse_bwd_apply_flat_body<T, MASK_X>(m.a[i], bid, nblk);
}

// Two terms, no upsampling (the residual-unit tail: BN(main) + identity | BN(shortcut)): ONE pass reads dout / out once
// and writes both input gradients.  Term k: BN when x[k] != null (dx = A*g + B*x + C) else identity (dx = g).
struct FuseBwd2Args {
    const unsigned char* dout;
    const unsigned char* out;
    const unsigned char* mask;
    const unsigned char* x[2];
    const float* scale[2];
    const float* mean[2];
    const float* invstd[2];
    const float* coef[2];
    unsigned char* dx[2];
    int accumulate[2];
    int c, relu;
    long total;
    const float* fold_slab[2];   // term k folds fold_slab[k][fold_rows[k]][2][c] itself (see fold_coef_block); null: coef[k]
    int fold_rows[2];
    const unsigned char* touch;  // optional: lh_l2_touch at the tail
    unsigned touch_bytes;
    long count;
    float* dgamma[2];
    float* dbeta[2];
    int exp;                     // bn_exp_flags() & 3
};

template <typename T>
__device__ __forceinline__ void fuse_bwd_apply2_flat_body(const FuseBwd2Args& p, const int bid, const int nblk) {
    const long total = p.total;
    constexpr int EPC = 16 / sizeof(T);
    const int nchunk = p.c / EPC;
    const int chunk = threadIdx.x & (nchunk - 1);
    float A[2][EPC], B[2][EPC], Cc[2][EPC];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        if (p.x[k]) {
            float iv[EPC], c0[EPC], c1[EPC], mn[EPC];
            load_vec<EPC>(p.scale[k] + chunk * EPC, A[k]);
            load_vec<EPC>(p.invstd[k] + chunk * EPC, iv);
            if (p.fold_slab[k]) {
                __shared__ float coefL2[512];
                if (k) __syncthreads();                  // the other term's coefficients have been read by every thread
                fold_coef_block(p.fold_slab[k], p.fold_rows[k], p.c, p.count, p.dgamma[k], p.dbeta[k], bid == 0, coefL2);
#pragma unroll
                for (int e = 0; e < EPC; ++e) { c0[e] = coefL2[chunk * EPC + e]; c1[e] = coefL2[p.c + chunk * EPC + e]; }
            } else {
                load_vec<EPC>(p.coef[k] + chunk * EPC, c0);
                load_vec<EPC>(p.coef[k] + p.c + chunk * EPC, c1);
            }
            load_vec<EPC>(p.mean[k] + chunk * EPC, mn);
#pragma unroll
            for (int e = 0; e < EPC; ++e) { B[k][e] = -A[k][e] * iv[e] * c1[e]; Cc[k][e] = -A[k][e] * c0[e] - B[k][e] * mn[e]; }
        } else { fill_vec<EPC>(A[k], 1.f); fill_vec<EPC>(B[k], 0.f); fill_vec<EPC>(Cc[k], 0.f); }
    }
    const long stride = (long)nblk * 256;
    const long rounds = (total + stride - 1) / stride;
    const bool rev = p.exp & 2;
    auto body = [&](auto NTc) __attribute__((always_inline)) {
    constexpr bool LNT = decltype(NTc)::value;
    for (long rr = 0; rr < rounds; ++rr) {
        const long idx = walk_round(rr, rounds, rev) * stride + (long)bid * 256 + threadIdx.x;
        if (idx >= total) continue;
        const long off = idx * 16;
        float g[EPC];
        unpack16<T>(ld16<LNT>(p.dout + off), g);
        if (p.relu && p.mask) {
            mask_by_bits<EPC>(p.mask[idx], g);
        } else if (p.relu) {
            float o[EPC];
            unpack16<T>(ld16<LNT>(p.out + off), o);
#pragma unroll
            for (int e = 0; e < EPC; ++e) g[e] = o[e] > 0.f ? g[e] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (!p.dx[k]) continue;
            float r[EPC];
            if (p.x[k]) {
                float xv[EPC];
                unpack16<T>(ld16<LNT>(p.x[k] + off), xv);
#pragma unroll
                for (int e = 0; e < EPC; ++e) r[e] = A[k][e] * g[e] + B[k][e] * xv[e] + Cc[k][e];
            } else {
#pragma unroll
                for (int e = 0; e < EPC; ++e) r[e] = g[e];
            }
            uint4* dst = reinterpret_cast<uint4*>(p.dx[k] + off);
            if (p.accumulate[k]) {
                float o[EPC];
                unpack16<T>(*dst, o);
#pragma unroll
                for (int e = 0; e < EPC; ++e) r[e] += o[e];
            }
            *dst = pack16<T>(r);
        }
    }
    };
    if (p.exp & 1) body(std::true_type{}); else body(std::false_type{});
    lh_l2_touch(p.touch, p.touch_bytes, bid, nblk);
}
template <typename T>
__global__ __launch_bounds__(256) void fuse_bwd_apply2_flat_kernel(const FuseBwd2Args p) { fuse_bwd_apply2_flat_body<T>(p, blockIdx.x, gridDim.x); }
template <typename T>
__global__ __launch_bounds__(256) void fuse_bwd_apply2_flat_multi_kernel(const LhMulti<FuseBwd2Args> m) {
    int bid, nblk;
    const int i = lh_multi_pick(m, bid, nblk);
    fuse_bwd_apply2_flat_body<T>(m.a[i], bid, nblk);
}

__global__ void fuse_bwd_coef_kernel(const double* totals, long count, int c, float* coef, float* dgamma, float* dbeta) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= c) return;
    coef[ch] = (float)(totals[ch] / (double)count);
    coef[c + ch] = (float)(totals[c + ch] / (double)count);
    if (dbeta) dbeta[ch] = (float)totals[ch];
    if (dgamma) dgamma[ch] = (float)totals[c + ch];
}

struct CoefArgs {
    const float* slab;           // [rows][2][c]
    int rows, c;
    long count;
    float* coef;
    float* dgamma;
    float* dbeta;
};

__device__ __forceinline__ void fuse_bwd_coef_fused_body(const CoefArgs& p, const int bid, const int nblk) {
    const long count = p.count;
    const int c = p.c;
    auto fin = [&](int ch, double s0, double s1) {
        p.coef[ch] = (float)(s0 / (double)count);
        p.coef[c + ch] = (float)(s1 / (double)count);
        if (p.dbeta) p.dbeta[ch] = (float)s0;
        if (p.dgamma) p.dgamma[ch] = (float)s1;
    };
    if (p.rows >= LH_FOLD_WIDE_ROWS) slab_totals_then64(p.slab, p.rows, p.c, bid, fin);
    else slab_totals_then(p.slab, p.rows, p.c, bid, fin);
}
__global__ __launch_bounds__(256) void fuse_bwd_coef_fused_kernel(const CoefArgs p) { fuse_bwd_coef_fused_body(p, blockIdx.x, gridDim.x); }
__global__ __launch_bounds__(256) void fuse_bwd_coef_fused_multi_kernel(const LhMulti<CoefArgs> m) {
    int bid, nblk;
    const int i = lh_multi_pick(m, bid, nblk);
    fuse_bwd_coef_fused_body(m.a[i], bid, nblk);
}

// Strips of the streaming reduce pass = rows of the partial-sum slab the coefficient fold reads.  Measured on the R50 and
// HRNet-W32 steps: 512 strips for a node that runs alone (1 024: the fold is a longer latency chain, -0.08 ms per step
// for 512; 256: the reduce pass loses occupancy), 256 per node when several nodes share a launch (lh_fuse_bwd_multi;
// the caller says so in lh_fuse_bwd_desc.strips_cap, so that a node plans the same strips alone and in company).
static long fuse_bwd_strips(long count, int* rows_per_strip, int strips_cap = 0) {
    const long cap = strips_cap >= 16 && strips_cap <= 512 ? strips_cap : 512;
    long rps = (count + cap - 1) / cap;
    if (rps < 16) rps = 16;
    *rows_per_strip = (int)rps;
    return (count + rps - 1) / rps;
}

// bytes of ONE term's slice: partial slab + colsum scratch + totals (doubles) + coefficients, rounded to 256
static size_t fuse_bwd_term_bytes(int n, int h, int w, int c) {
    int rps;
    const long strips = fuse_bwd_strips((long)n * h * w, &rps);
    const size_t b = (size_t)strips * 2 * c * 4 + 16 + (size_t)(ceil_div(strips, 256) + 1) * 2 * c * 8 + (size_t)4 * c * 4;
    return (b + 255) & ~(size_t)255;
}

// every term of a node owns a slice of the workspace: the terms' passes are independent of each other and run as
// multi-problem launches (all reduce passes, then all coefficient folds, then all apply passes)
extern "C" size_t lh_fuse_bwd_workspace_bytes(int n, int h, int w, int c) { return 4 * fuse_bwd_term_bytes(n, h, w, c); }

struct BnLaunch {
    int kind, grid;
    int phase;                   // backward passes: 0 reduce, 1 coefficient fold, 2 apply (a pass only depends on the passes of lower phase)
    FuseArgs ff;
    FuseBwdArgs fb;
    FuseBwd2Args fb2;
    CoefArgs co;
};

// Run n <= LH_MULTI_MAX records of ONE kind: a plain launch for one, a multi-problem launch for several.
#define BN_RUN(KERNEL, MULTI, FIELD, ARGS_T)                                                                          \
    do {                                                                                                               \
        if (n == 1) {                                                                                                  \
            hipLaunchKernelGGL(KERNEL, dim3(L[0]->grid), dim3(256), 0, s, L[0]->FIELD);                                \
        } else {                                                                                                       \
            LhMulti<ARGS_T> m;                                                                                         \
            m.n = n; m.first[0] = 0;                                                                                   \
            for (int i = 0; i < n; ++i) { m.a[i] = L[i]->FIELD; m.first[i + 1] = m.first[i] + L[i]->grid; }           \
            hipLaunchKernelGGL(MULTI, dim3(m.first[n]), dim3(256), 0, s, m);                                           \
        }                                                                                                              \
    } while (0)

static int bn_run(const BnLaunch* const* L, int n, int dtype, hipStream_t s) {
    LH_REQUIRE(n >= 1 && n <= LH_MULTI_MAX, "bn_run: %d records", n);
    switch (L[0]->kind) {
        case K_FF_GEN: LH_DISPATCH_DTYPE(dtype, T, BN_RUN((fuse_fwd_kernel<T>), (fuse_fwd_multi_kernel<T>), ff, FuseArgs)); break;
        case K_FF_FLAT1: LH_DISPATCH_DTYPE(dtype, T, BN_RUN((fuse_fwd_flat_kernel<T, 1>), (fuse_fwd_flat_multi_kernel<T, 1>), ff, FuseArgs)); break;
        case K_FF_FLAT2: LH_DISPATCH_DTYPE(dtype, T, BN_RUN((fuse_fwd_flat_kernel<T, 2>), (fuse_fwd_flat_multi_kernel<T, 2>), ff, FuseArgs)); break;
        case K_FB_REDUCE_GEN: LH_DISPATCH_DTYPE(dtype, T, BN_RUN((fuse_bwd_reduce_kernel<T>), (fuse_bwd_reduce_multi_kernel<T>), fb, FuseBwdArgs)); break;
        case K_FB_REDUCE_FLAT: LH_DISPATCH_DTYPE(dtype, T, BN_RUN((fuse_bwd_reduce_flat_kernel<T, false>), (fuse_bwd_reduce_flat_multi_kernel<T, false>), fb, FuseBwdArgs)); break;
        case K_FB_REDUCE_FLAT_X: LH_DISPATCH_DTYPE(dtype, T, BN_RUN((fuse_bwd_reduce_flat_kernel<T, true>), (fuse_bwd_reduce_flat_multi_kernel<T, true>), fb, FuseBwdArgs)); break;
        case K_FB_COEF: BN_RUN(fuse_bwd_coef_fused_kernel, fuse_bwd_coef_fused_multi_kernel, co, CoefArgs); break;
        case K_FB_APPLY_GEN: LH_DISPATCH_DTYPE(dtype, T, BN_RUN((fuse_bwd_apply_kernel<T>), (fuse_bwd_apply_multi_kernel<T>), fb, FuseBwdArgs)); break;
        case K_FB_APPLY_FLAT: LH_DISPATCH_DTYPE(dtype, T, BN_RUN((fuse_bwd_apply_flat_kernel<T, false>), (fuse_bwd_apply_flat_multi_kernel<T, false>), fb, FuseBwdArgs)); break;
        case K_FB_APPLY_FLAT_X: LH_DISPATCH_DTYPE(dtype, T, BN_RUN((fuse_bwd_apply_flat_kernel<T, true>), (fuse_bwd_apply_flat_multi_kernel<T, true>), fb, FuseBwdArgs)); break;
        case K_FB_APPLY2: LH_DISPATCH_DTYPE(dtype, T, BN_RUN((fuse_bwd_apply2_flat_kernel<T>), (fuse_bwd_apply2_flat_multi_kernel<T>), fb2, FuseBwd2Args)); break;
        default: lh_set_error("bn_run: unknown kind %d", L[0]->kind); return LH_ERR_ARG;
    }
    LH_LAUNCH_CHECK("BatchNorm / ReLU pass launch");
    return LH_OK;
}

// Reduce / apply passes of DIFFERENT kinds in one grid (the terms of an HRNet exchange sum: BN terms without and with
// upsampling, identity terms): the kind is a per-problem tag next to the argument blocks.
struct KindTags { int k[LH_MULTI_MAX]; };
template <typename T>
__global__ __launch_bounds__(256) void fuse_bwd_reduce_mixed_kernel(const LhMulti<FuseBwdArgs> m, const KindTags kt) {
    int bid, nblk;
    const int i = lh_multi_pick(m, bid, nblk);
    switch (kt.k[i]) {
        case K_FB_REDUCE_GEN: fuse_bwd_reduce_body<T>(m.a[i], bid, nblk); break;
        case K_FB_REDUCE_FLAT: fuse_bwd_reduce_flat_body<T, false>(m.a[i], bid, nblk); break;
        default: fuse_bwd_reduce_flat_body<T, true>(m.a[i], bid, nblk); break;
    }
}
template <typename T>
__global__ __launch_bounds__(256) void fuse_bwd_apply_mixed_kernel(const LhMulti<FuseBwdArgs> m, const KindTags kt) {
    int bid, nblk;
    const int i = lh_multi_pick(m, bid, nblk);
    switch (kt.k[i]) {
        case K_FB_APPLY_GEN: fuse_bwd_apply_body<T>(m.a[i], bid, nblk); break;
        case K_FB_APPLY_FLAT: fuse_bwd_apply_flat_body<T, false>(m.a[i], bid, nblk); break;
        default: fuse_bwd_apply_flat_body<T, true>(m.a[i], bid, nblk); break;
    }
}

static int bn_run_mixed(const BnLaunch* const* L, int n, int dtype, hipStream_t s) {
    LhMulti<FuseBwdArgs> m;
    KindTags kt;
    m.n = n; m.first[0] = 0;
    for (int i = 0; i < n; ++i) { m.a[i] = L[i]->fb; kt.k[i] = L[i]->kind; m.first[i + 1] = m.first[i] + L[i]->grid; }
    for (int i = n; i < LH_MULTI_MAX; ++i) kt.k[i] = 0;
    if (L[0]->phase == 0) { LH_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((fuse_bwd_reduce_mixed_kernel<T>), dim3(m.first[n]), dim3(256), 0, s, m, kt)); }
    else { LH_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((fuse_bwd_apply_mixed_kernel<T>), dim3(m.first[n]), dim3(256), 0, s, m, kt)); }
    LH_LAUNCH_CHECK("BatchNorm / ReLU mixed pass launch");
    return LH_OK;
}

// The backward passes of n planned calls, phase by phase (all reduce passes, all coefficient folds, all apply passes):
// inside a phase the passes are independent -- every term has its own workspace slice and its own gradient -- so they
// run LH_MULTI_MAX per launch, passes of one kind through that kind's kernel, the rest through the mixed kernels.
static int bn_run_phases(const std::vector<std::vector<BnLaunch>>& plans, int dtype, hipStream_t s) {
    // two passes that write ONE gradient buffer (an activation that enters a node twice) must keep their recorded order
    std::vector<const void*> dsts;
    for (const auto& pl : plans)
        for (const BnLaunch& r : pl) {
            if (r.phase != 2) continue;
            if (r.kind == K_FB_APPLY2) { if (r.fb2.dx[0]) dsts.push_back(r.fb2.dx[0]); if (r.fb2.dx[1]) dsts.push_back(r.fb2.dx[1]); }
            else dsts.push_back(r.fb.dx);
        }
    std::sort(dsts.begin(), dsts.end());
    if (std::adjacent_find(dsts.begin(), dsts.end()) != dsts.end()) {
        const BnLaunch* L[1];
        for (const auto& pl : plans)
            for (const BnLaunch& r : pl) {
                L[0] = &r;
                const int rc = bn_run(L, 1, dtype, s);
                if (rc) return rc;
            }
        return LH_OK;
    }
    for (int phase = 0; phase < 3; ++phase) {
        std::vector<const BnLaunch*> fbk, other;        // FuseBwdArgs reduce / apply passes | coefficient folds, two-term applies
        for (const auto& pl : plans)
            for (const BnLaunch& r : pl) {
                if (r.phase != phase) continue;
                (r.kind == K_FB_COEF || r.kind == K_FB_APPLY2 ? other : fbk).push_back(&r);
            }
        // passes of one kind first (their own kernels, no kind switch), what is left over goes to the mixed kernel
        std::vector<const BnLaunch*> rest;
        for (int kind = K_FF_GEN; kind <= K_FB_APPLY2; ++kind) {
            std::vector<const BnLaunch*> same;
            for (const BnLaunch* r : fbk) if (r->kind == kind) same.push_back(r);
            const size_t whole = same.size() / LH_MULTI_MAX * LH_MULTI_MAX;
            for (size_t i = 0; i < whole; i += LH_MULTI_MAX) {
                const int rc = bn_run(&same[i], LH_MULTI_MAX, dtype, s);
                if (rc) return rc;
            }
            rest.insert(rest.end(), same.begin() + whole, same.end());
        }
        for (size_t i = 0; i < rest.size(); i += LH_MULTI_MAX) {
            const int m = (int)(rest.size() - i < (size_t)LH_MULTI_MAX ? rest.size() - i : LH_MULTI_MAX);
            bool one = true;
            for (int k = 1; k < m; ++k) one = one && rest[i + k]->kind == rest[i]->kind;
            const int rc = one ? bn_run(&rest[i], m, dtype, s) : bn_run_mixed(&rest[i], m, dtype, s);
            if (rc) return rc;
        }
        for (int kind : {(int)K_FB_COEF, (int)K_FB_APPLY2}) {
            std::vector<const BnLaunch*> same;
            for (const BnLaunch* r : other) if (r->kind == kind) same.push_back(r);
            for (size_t i = 0; i < same.size(); i += LH_MULTI_MAX) {
                const int m = (int)(same.size() - i < (size_t)LH_MULTI_MAX ? same.size() - i : LH_MULTI_MAX);
                const int rc = bn_run(&same[i], m, dtype, s);
                if (rc) return rc;
            }
        }
    }
    return LH_OK;
}

// Runs the records of n planned calls: position by position as multi-problem launches when every call planned the same
// sequence of kinds (the parallel branches of an HRNet module do), else call by call.
static int bn_run_calls(const std::vector<std::vector<BnLaunch>>& plans, int dtype, hipStream_t s) {
    const int n = (int)plans.size();
    bool same = n > 1;
    for (int i = 1; i < n && same; ++i) {
        same = plans[i].size() == plans[0].size();
        for (size_t k = 0; same && k < plans[0].size(); ++k) same = plans[i][k].kind == plans[0][k].kind;
    }
    const BnLaunch* L[LH_MULTI_MAX];
    if (!same) {
        for (int i = 0; i < n; ++i)
            for (const BnLaunch& r : plans[i]) {
                L[0] = &r;
                const int rc = bn_run(L, 1, dtype, s);
                if (rc) return rc;
            }
        return LH_OK;
    }
    for (size_t k = 0; k < plans[0].size(); ++k)
        for (int i0 = 0; i0 < n; i0 += LH_MULTI_MAX) {
            const int m = n - i0 < LH_MULTI_MAX ? n - i0 : LH_MULTI_MAX;
            for (int i = 0; i < m; ++i) L[i] = &plans[i0 + i][k];
            const int rc = bn_run(L, m, dtype, s);
            if (rc) return rc;
        }
    return LH_OK;
}

// the pending finalizes of a descriptor that the planned kernel does not run itself: launched first (one multi-problem launch)
static int run_pending_finalizes(const std::vector<lh_bn_finalize_call>& fins, void* stream) {
    return fins.empty() ? LH_OK : lh_bn_finalize_multi(fins.data(), (int)fins.size(), stream);
}

extern "C" int lh_fuse_fwd(const lh_fuse_desc* d, void* out, int n, int h, int w, int c, int dtype, void* stream) {
    BnLaunch r;
    int pre = 0;
    int rc = plan_fuse_fwd(d, out, n, h, w, c, dtype, &r.ff, &r.kind, &r.grid, &pre);
    if (rc) return rc;
    lh_bn_finalize_call fins[4];                  // at most one pending finalize per term: no heap allocation on the launch path
    int nf = 0;
    for (int t = 0; t < d->nterms && t < 4; ++t)
        if ((pre >> t) & 1) fins[nf++] = *d->fin[t];
    if (nf) rc = lh_bn_finalize_multi(fins, nf, stream);
    if (rc) return rc;
    const BnLaunch* L[1] = {&r};
    return bn_run(L, 1, dtype, (hipStream_t)stream);
}

extern "C" int lh_fuse_fwd_multi(const lh_fuse_fwd_call* calls, int n, int dtype, void* stream) {
    LH_REQUIRE(calls && n >= 1, "lh_fuse_fwd_multi: bad arguments");
    std::vector<std::vector<BnLaunch>> plans(n, std::vector<BnLaunch>(1));
    std::vector<lh_bn_finalize_call> fins;
    // the calls merge into one launch only when they plan the SAME kernel: the in-launch finalize is used when every call
    // takes it, else none does (their finalizes then run first, as one multi-problem launch)
    for (int pass = 0; pass < 2; ++pass) {
        fins.clear();
        bool same = true;
        for (int i = 0; i < n; ++i) {
            BnLaunch& r = plans[i][0];
            int pre = 0;
            const int rc = plan_fuse_fwd(calls[i].d, calls[i].out, calls[i].n, calls[i].h, calls[i].w, calls[i].c, dtype, &r.ff, &r.kind, &r.grid,
                                         &pre);
            if (rc) return rc;
            for (int t = 0; t < calls[i].d->nterms; ++t)
                if ((pre >> t) & 1) fins.push_back(*calls[i].d->fin[t]);
            same = same && r.kind == plans[0][0].kind;
        }
        if (same || pass == 1) break;
    }
    const int rc = run_pending_finalizes(fins, stream);
    if (rc) return rc;
    return bn_run_calls(plans, dtype, (hipStream_t)stream);
}

static int plan_fuse_bwd(const lh_fuse_bwd_desc* d, int n, int h, int w, int c, void* workspace, int dtype, std::vector<BnLaunch>& v) {
    LH_REQUIRE(d && d->dout && d->nterms >= 1 && d->nterms <= 4, "lh_fuse_bwd: bad descriptor");
    LH_REQUIRE(!d->relu || d->out || d->relu_mask, "lh_fuse_bwd: relu needs the forward output or its mask bits");
    const int es = lh_dtype_size(dtype);
    LH_REQUIRE(es > 0 && c % (16 / es) == 0, "lh_fuse_bwd: c %d not a multiple of the 16-byte chunk", c);
    const int nchunk0 = c / (16 / es);
    const bool merge2 = d->nterms == 2 && d->log2up[0] == 0 && d->log2up[1] == 0 && (nchunk0 & (nchunk0 - 1)) == 0 &&
                        nchunk0 <= 256 && (d->dx[0] || d->dx[1]);
    const size_t term_bytes = fuse_bwd_term_bytes(n, h, w, c);
    FuseBwd2Args m2;
    m2.touch = nullptr; m2.touch_bytes = 0;
    if (merge2) {
        m2.dout = (const unsigned char*)d->dout; m2.out = (const unsigned char*)d->out; m2.c = c; m2.relu = d->relu;
        m2.mask = (const unsigned char*)d->relu_mask;
        for (int k = 0; k < 2; ++k) {
            m2.x[k] = (const unsigned char*)d->x[k]; m2.scale[k] = d->scale[k]; m2.mean[k] = d->save_mean[k];
            m2.invstd[k] = d->save_invstd[k]; m2.coef[k] = nullptr; m2.dx[k] = (unsigned char*)d->dx[k];
            m2.accumulate[k] = d->accumulate[k];
        }
    }
    if (merge2 && d->pre_partial) { m2.relu = 0; m2.out = nullptr; m2.mask = nullptr; }      // dout is the gated gradient already
    if (merge2) {
        m2.count = (long)n * h * w;
        for (int k = 0; k < 2; ++k) { m2.fold_slab[k] = nullptr; m2.fold_rows[k] = 0; m2.dgamma[k] = nullptr; m2.dbeta[k] = nullptr; }
    }
    for (int t = 0; t < d->nterms; ++t) {
        if (!d->dx[t]) continue;
        FuseBwdArgs a;
        a.fold_rows = 0;
        a.touch = nullptr; a.touch_bytes = 0;
        a.dout = (const unsigned char*)d->dout; a.out = (const unsigned char*)d->out;
        a.mask = (const unsigned char*)d->relu_mask;
        a.x = (const unsigned char*)d->x[t]; a.scale = d->scale[t]; a.mean = d->save_mean[t]; a.invstd = d->save_invstd[t];
        a.dx = (unsigned char*)d->dx[t]; a.dgamma = d->dgamma[t]; a.dbeta = d->dbeta[t];
        a.n = n; a.h = h; a.w = w; a.c = c; a.l = d->log2up[t]; a.relu = d->relu; a.accumulate = d->accumulate[t];
        LH_REQUIRE(a.l >= 0 && (h >> a.l) << a.l == h && (w >> a.l) << a.l == w, "lh_fuse_bwd: bad upsampling factor");
        a.count = (long)n * (h >> a.l) * (w >> a.l);
        a.partial = nullptr; a.totals = nullptr; a.coef = nullptr; a.rows_per_strip = 0;
        a.shift = nullptr;
        const int nchunk = c / (16 / es);
        const bool flat = a.l == 0 && (nchunk & (nchunk - 1)) == 0 && nchunk <= 256;
        // dout written by lh_igemm_gated: already the gated gradient, its partial sums come with it
        const bool two_bn = d->nterms == 2 && d->x[0] && d->x[1];
        const float* pre_slab = !(d->pre_partial && a.x) ? nullptr : (two_bn && t == 1) ? d->pre_partial2 : d->pre_partial;
        const bool pre = pre_slab != nullptr;
        if (d->pre_partial) {
            // one BatchNorm term under the ReLU, alone or with an identity term beside it (a residual tail, merged apply pass); or a tail
            // with a projection shortcut: two BatchNorm terms, each with its slab
            LH_REQUIRE((d->nterms == 1 || merge2) && d->relu && d->pre_rows >= 1 && a.l == 0 && (a.x || merge2) && (!two_bn || d->pre_partial2),
                       "lh_fuse_bwd: pre_partial takes a ReLU node with one BatchNorm term (alone, or beside one identity term) or a two-term tail with pre_partial2");
            a.relu = 0;
            a.out = nullptr; a.mask = nullptr;
        }
        // single BN term under the ReLU: the mask is sign(x*scale+shift), no need to read the stored activation
        a.mask_from_x = (!pre && flat && d->relu && d->nterms == 1 && a.x && d->shift[t]) ? 1 : 0;
        if (a.mask_from_x) a.shift = d->shift[t];
        a.total = a.count * (c / (16 / es));
        a.exp = bn_exp_flags() & 3;
        LH_REQUIRE((long)n * h * w * (c / (16 / es)) < (1L << 31), "lh_fuse_bwd: tensor too large for 32-bit chunk indices");
        if (a.x) {
            LH_REQUIRE(workspace && a.scale && a.mean && a.invstd, "lh_fuse_bwd: BN term %d lacks workspace/statistics", t);
            long strips = fuse_bwd_strips(a.count, &a.rows_per_strip, d->strips_cap);
            // small tensors (HRNet's branches): few enough strips that the apply pass folds them itself; a strip must stay
            // short (<= 64 KiB of the operand), else the reduce pass would lose the workgroups it streams with
            bool fold_in_apply = false;
            if (pre) {
                strips = d->pre_rows;
                fold_in_apply = flat && c <= 256 && getenv("LH_FOLD_IN_APPLY") == nullptr && strips * 2 * c <= LH_FOLD_IN_APPLY_FLOATS;
            } else if (flat && c <= 256 && getenv("LH_FOLD_IN_APPLY") == nullptr) {
                int rps2;
                const long s2 = fuse_bwd_strips(a.count, &rps2, (int)std::min<long>(d->strips_cap >= 16 ? d->strips_cap : 512, LH_FOLD_IN_APPLY_FLOATS / (2 * c)));
                if (s2 * 2 * c <= LH_FOLD_IN_APPLY_FLOATS && (long)rps2 * c * es <= 65536) {
                    fold_in_apply = true;
                    strips = s2;
                    a.rows_per_strip = rps2;
                }
            }
            a.partial = (float*)((unsigned char*)workspace + (size_t)t * term_bytes);
            const long slab_floats = pre ? 0 : strips * 2 * c;      // pre: the slab is the caller's, the workspace holds the scratch only
            double* scratch = (double*)(a.partial + ((slab_floats + 3) & ~3L));
            if (pre) a.partial = const_cast<float*>(pre_slab);
            double* totals = scratch + (long)ceil_div(strips, 256) * 2 * c;
            a.totals = totals;
            a.coef = (float*)(totals + 2 * c) + (size_t)(merge2 ? t : 0) * 2 * c;   // merging keeps one coefficient block per term
            if (!pre) {
                BnLaunch r;
                r.kind = flat ? (a.mask_from_x ? K_FB_REDUCE_FLAT_X : K_FB_REDUCE_FLAT) : K_FB_REDUCE_GEN;
                r.grid = (int)strips;
                r.phase = 0;
                r.fb = a;
                v.push_back(r);
            }
            if (fold_in_apply) {
                if (merge2) { m2.fold_slab[t] = a.partial; m2.fold_rows[t] = (int)strips; m2.dgamma[t] = a.dgamma; m2.dbeta[t] = a.dbeta; }
                else a.fold_rows = (int)strips;
            } else {
                BnLaunch q;
                q.phase = 1;
                q.kind = K_FB_COEF; q.grid = fold_grid((int)strips, c);
                q.co.slab = a.partial; q.co.rows = (int)strips; q.co.c = c; q.co.count = a.count; q.co.coef = a.coef; q.co.dgamma = a.dgamma; q.co.dbeta = a.dbeta;
                v.push_back(q);
            }
            if (merge2) m2.coef[t] = a.coef;
        }
        if (merge2) continue;                 // both gradients are written by ONE pass below
        BnLaunch r;
        if (flat) {
            r.kind = a.mask_from_x ? K_FB_APPLY_FLAT_X : K_FB_APPLY_FLAT;
            r.grid = flat_grid(a.total);          // >= 4 chunks per thread
        } else {
            r.kind = K_FB_APPLY_GEN;
            r.grid = (int)((a.total + 255) / 256 > 4096 ? 4096 : (a.total + 255) / 256);
        }
        r.phase = 2;
        r.fb = a;
        v.push_back(r);
    }
    if (merge2) {
        BnLaunch r;
        r.phase = 2;
        m2.total = (long)n * h * w * nchunk0;
        m2.exp = bn_exp_flags() & 3;
        r.kind = K_FB_APPLY2;
        r.grid = flat_grid(m2.total);
        r.fb2 = m2;
        v.push_back(r);
    }
    // the LAST launch of the call (an apply pass) warms what the next launch on the stream reads first (lh_fuse_bwd_desc.l2_touch)
    if (d->l2_touch && d->l2_touch_bytes > 0 && d->l2_touch_bytes < (1UL << 31) && !v.empty()) {
        BnLaunch& last = v.back();
        if (last.kind == K_FB_APPLY2) { last.fb2.touch = (const unsigned char*)d->l2_touch; last.fb2.touch_bytes = (unsigned)d->l2_touch_bytes; }
        else if (last.kind == K_FB_APPLY_FLAT || last.kind == K_FB_APPLY_FLAT_X) {
            last.fb.touch = (const unsigned char*)d->l2_touch; last.fb.touch_bytes = (unsigned)d->l2_touch_bytes;
        }
    }
    return LH_OK;
}

extern "C" int lh_fuse_bwd(const lh_fuse_bwd_desc* d, int n, int h, int w, int c, void* workspace, int dtype,
                           void* stream) {
    std::vector<std::vector<BnLaunch>> plans(1);
    const int rc = plan_fuse_bwd(d, n, h, w, c, workspace, dtype, plans[0]);
    if (rc) return rc;
    return bn_run_phases(plans, dtype, (hipStream_t)stream);
}

// n independent nodes (each with its OWN workspace): reduce / coefficient fold / apply of all of them as three launches.
extern "C" int lh_fuse_bwd_multi(const lh_fuse_bwd_call* calls, int n, int dtype, void* stream) {
    LH_REQUIRE(calls && n >= 1, "lh_fuse_bwd_multi: bad arguments");
    std::vector<std::vector<BnLaunch>> plans(n);
    for (int i = 0; i < n; ++i) {
        const int rc = plan_fuse_bwd(calls[i].d, calls[i].n, calls[i].h, calls[i].w, calls[i].c, calls[i].workspace, dtype, plans[i]);
        if (rc) return rc;
    }
    return bn_run_phases(plans, dtype, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------
// BN = true: x is the RAW BatchNorm input and every tap becomes relu(x * scale + shift), rounded to T as lh_fuse_fwd would
// have stored it, before it enters the maximum (lh_bn_relu_maxpool3x3s2_fwd: the activation between the BatchNorm and the
// pool is never written).
template <typename T, bool BN>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* x, T* out, unsigned char* idx, int n, int h, int w,
                                                          int c, int ho, int wo, const float* scale, const float* shift) {
    constexpr int EPC = 16 / sizeof(T);
    const int nchunk = c / EPC;
    const long total = (long)n * ho * wo * nchunk;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        // 32-bit index arithmetic (the launcher checks total < 2^31): 64-bit divisions were most of this kernel's instructions
        const unsigned iu = (unsigned)i;
        const unsigned pix = iu / (unsigned)nchunk;
        const int ch = (int)(iu - pix * (unsigned)nchunk);
        const unsigned t2 = pix / (unsigned)wo;
        const int ow = (int)(pix - t2 * (unsigned)wo);
        const int b = (int)(t2 / (unsigned)ho), oh = (int)(t2 - (unsigned)b * (unsigned)ho);
        // all nine taps are requested before the first compare (taps outside the image re-read the centre tap, which is
        // always inside, and are skipped below): the loads of a window overlap instead of alternating with the compares
        uint4 raw[9];
        bool ok[9];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int ih = oh * 2 - 1 + r, iw = ow * 2 - 1 + s;
                ok[r * 3 + s] = (unsigned)ih < (unsigned)h && (unsigned)iw < (unsigned)w;
                const int jh = ok[r * 3 + s] ? ih : oh * 2, jw = ok[r * 3 + s] ? iw : ow * 2;
                raw[r * 3 + s] = *reinterpret_cast<const uint4*>(x + (((long)b * h + jh) * w + jw) * c + ch * EPC);
            }
        float best[EPC];
        unsigned char bi[EPC];
        bool any = false;
        float sc[EPC], sh[EPC];
        if constexpr (BN) { load_vec<EPC>(scale + ch * EPC, sc); load_vec<EPC>(shift + ch * EPC, sh); }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            if (!ok[t]) continue;
            float v[EPC];
            unpack16<T>(raw[t], v);
            if constexpr (BN) {
#pragma unroll
                for (int e = 0; e < EPC; ++e) v[e] = fmaxf(v[e] * sc[e] + sh[e], 0.f);
                unpack16<T>(pack16<T>(v), v);        // the rounding of the stored activation
            }
            if (!any) {                              // first tap inside the image (scan order)
#pragma unroll
                for (int e = 0; e < EPC; ++e) { best[e] = v[e]; bi[e] = (unsigned char)t; }
                any = true;
            } else {
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    // same rule as the framework's CPU kernel: strictly greater or NaN replaces
                    if (v[e] > best[e] || v[e] != v[e]) { best[e] = v[e]; bi[e] = (unsigned char)t; }
                }
            }
        }
        *reinterpret_cast<uint4*>(out + i * EPC) = pack16<T>(best);
        // the EPC window positions of the chunk as ONE store (i * EPC bytes in: 4- or 8-byte aligned)
        unsigned pk[EPC / 4];
#pragma unroll
        for (int e = 0; e < EPC / 4; ++e)
            pk[e] = (unsigned)bi[4 * e] | ((unsigned)bi[4 * e + 1] << 8) | ((unsigned)bi[4 * e + 2] << 16) | ((unsigned)bi[4 * e + 3] << 24);
        if (idx) {
            if constexpr (EPC == 8) *reinterpret_cast<uint2*>(idx + i * EPC) = uint2{pk[0], pk[1]};
            else *reinterpret_cast<unsigned*>(idx + i * EPC) = pk[0];
        }
    }
}

// The BN = true pass as a COLUMN STRIP kernel (round 5): a thread owns one 16-byte channel chunk of R vertically adjacent pooled
// pixels, so the 2 R + 1 input rows its windows touch are loaded and run through relu(x * scale + shift) once per strip instead of once
// per window (9 -> (6 R + 3) / R chunk transforms per pooled chunk), and the maxima are taken on INTEGER keys: after the ReLU every tap
// is >= 0, so the 16-bit patterns of the rounded values order like the values; key = magnitude << 16 | (3 - row) << 3 | (3 - col) << 1 |
// sign bit picks the largest value and, among equals (-0 = +0 as for floats), the first position in scan order -- the old kernel's
// `strictly greater replaces` rule -- and carries the winner's sign bit back out.  Same pooled values, same window positions, bit for bit.
template <typename T, int R>
__global__ __launch_bounds__(256) void bn_relu_pool_strip_kernel(const T* x, T* out, unsigned char* idx, int n, int h, int w, int c, int ho, int wo,
                                                                const float* scale, const float* shift) {
    static_assert(sizeof(T) == 2, "16-bit element types");
    constexpr int EPC = 8, NRW = 2 * R + 1;
    const int nchunk = c / EPC, strips = (ho + R - 1) / R;
    const unsigned total = (unsigned)n * strips * wo * nchunk;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const unsigned pix = i / (unsigned)nchunk;
        const int ch = (int)(i - pix * (unsigned)nchunk);
        const unsigned t2 = pix / (unsigned)wo;
        const int ow = (int)(pix - t2 * (unsigned)wo);
        const int b = (int)(t2 / (unsigned)strips), oh0 = (int)(t2 - (unsigned)b * (unsigned)strips) * R;
        // every tap is requested before the first is used (taps outside the image re-read a tap that is inside; their keys are 0)
        uint4 raw[NRW][3];
        bool rok[NRW], cok[3];
#pragma unroll
        for (int s = 0; s < 3; ++s) cok[s] = (unsigned)(ow * 2 - 1 + s) < (unsigned)w;
#pragma unroll
        for (int r = 0; r < NRW; ++r) {
            const int ih = oh0 * 2 - 1 + r;
            rok[r] = (unsigned)ih < (unsigned)h;
            const int jh = rok[r] ? ih : oh0 * 2;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int jw = cok[s] ? ow * 2 - 1 + s : ow * 2;
                raw[r][s] = *reinterpret_cast<const uint4*>(x + (((long)b * h + jh) * w + jw) * c + ch * EPC);
            }
        }
        float sc[EPC], sh[EPC];
        load_vec<EPC>(scale + ch * EPC, sc);
        load_vec<EPC>(shift + ch * EPC, sh);
        unsigned hkey[NRW][EPC];                       // per input row: the best of its three columns (0: no tap of the row is inside)
#pragma unroll
        for (int r = 0; r < NRW; ++r) {
#pragma unroll
            for (int e = 0; e < EPC; ++e) hkey[r][e] = 0u;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                float v[EPC];
                unpack16<T>(raw[r][s], v);
#pragma unroll
                for (int e = 0; e < EPC; ++e) v[e] = fmaxf(v[e] * sc[e] + sh[e], 0.f);
                const uint4 u = pack16<T>(v);         // the rounding of the stored activation
                const unsigned wd[4] = {u.x, u.y, u.z, u.w};
                const bool ok = rok[r] && cok[s];
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    const unsigned bits = (e & 1) ? wd[e >> 1] >> 16 : wd[e >> 1] & 0xffffu;
                    const unsigned key = ((bits & 0x7fffu) << 16) | ((unsigned)(3 - s) << 1) | (bits >> 15);
                    const unsigned k2 = ok ? key : 0u;
                    hkey[r][e] = k2 > hkey[r][e] ? k2 : hkey[r][e];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int oh = oh0 + j;
            if (oh >= ho) break;
            unsigned best[EPC];
#pragma unroll
            for (int e = 0; e < EPC; ++e) best[e] = 0u;
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    const unsigned hk = hkey[2 * j + r][e];
                    const unsigned k = hk ? (hk | ((unsigned)(3 - r) << 3)) : 0u;
                    best[e] = k > best[e] ? k : best[e];
                }
            unsigned ov[4], pk[2];
#pragma unroll
            for (int e = 0; e < EPC; e += 2) {
                const unsigned lo = (best[e] >> 16) | ((best[e] & 1u) << 15), hi = (best[e + 1] >> 16) | ((best[e + 1] & 1u) << 15);
                ov[e >> 1] = lo | (hi << 16);
            }
            unsigned tpos[EPC];
#pragma unroll
            for (int e = 0; e < EPC; ++e) tpos[e] = (3u - ((best[e] >> 3) & 3u)) * 3u + (3u - ((best[e] >> 1) & 3u));
            pk[0] = tpos[0] | (tpos[1] << 8) | (tpos[2] << 16) | (tpos[3] << 24);
            pk[1] = tpos[4] | (tpos[5] << 8) | (tpos[6] << 16) | (tpos[7] << 24);
            const long o = ((((long)b * ho + oh) * wo + ow) * nchunk + ch) * EPC;
            *reinterpret_cast<uint4*>(out + o) = uint4{ov[0], ov[1], ov[2], ov[3]};
            if (idx) *reinterpret_cast<uint2*>(idx + o) = uint2{pk[0], pk[1]};
        }
    }
}

// GATE = true (lh_maxpool3x3s2_bwd_gated): dx is the gradient of a = relu(BN(gx)); the pass stores the ReLU-gated gradient
// and writes the BatchNorm-backward partial sums { sum g, sum g * xhat } of its elements, one row per workgroup (what
// lh_igemm_gated does for a data gradient): lh_fuse_bwd then runs without its reduce pass.  Needs a power-of-two number of
// 16-byte chunks per pixel <= 256 (a thread keeps its chunk over the grid-stride loop).
struct PoolGate {
    const unsigned char* x;
    const float* mean;
    const float* invstd;
    const float* scale;
    const float* shift;
    float* partial;
};
template <typename T, bool GATE>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* dout, const unsigned char* idx, T* dx, int n, int h,
                                                          int w, int c, int ho, int wo, const PoolGate gt) {
    constexpr int EPC = 16 / sizeof(T);
    const int nchunk = c / EPC;
    const long total = (long)n * h * w * nchunk;
    float gmean[EPC], ginv[EPC], gsc[EPC], gsh[EPC], s1[EPC], s2[EPC];
    if constexpr (GATE) {
        const int chunk = threadIdx.x & (nchunk - 1);
        load_vec<EPC>(gt.mean + chunk * EPC, gmean); load_vec<EPC>(gt.invstd + chunk * EPC, ginv);
        load_vec<EPC>(gt.scale + chunk * EPC, gsc); load_vec<EPC>(gt.shift + chunk * EPC, gsh);
        fill_vec<EPC>(s1, 0.f); fill_vec<EPC>(s2, 0.f);
    }
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const unsigned iu = (unsigned)i;             // 32-bit index arithmetic (total < 2^31, checked by the launcher)
        const unsigned pix = iu / (unsigned)nchunk;
        const int ch = (int)(iu - pix * (unsigned)nchunk);
        const unsigned t2 = pix / (unsigned)w;
        const int iw = (int)(pix - t2 * (unsigned)w);
        const int b = (int)(t2 / (unsigned)h), ih = (int)(t2 - (unsigned)b * (unsigned)h);
        // an input pixel lies in at most 2 x 2 windows: all four (gradient chunk, position bytes) pairs are requested
        // up front (windows that do not exist re-read the first one and are skipped), then added in window order
        float g[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) g[e] = 0.f;
        uint4 dv[4];
        unsigned pk[4][2];
        int code[4];                                 // window position this pixel has in window k, -1: no such window
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int oh = (ih >> 1) + (k >> 1), ow = (iw >> 1) + (k & 1);
            const int r = ih + 1 - 2 * oh, s2 = iw + 1 - 2 * ow;
            const bool ok = oh <= ((ih + 1) >> 1) && ow <= ((iw + 1) >> 1) && oh < ho && ow < wo && r >= 0 && r <= 2 && s2 >= 0 && s2 <= 2;
            code[k] = ok ? r * 3 + s2 : -1;
            const int jh = ok ? oh : (ih >> 1) < ho ? (ih >> 1) : ho - 1, jw = ok ? ow : (iw >> 1) < wo ? (iw >> 1) : wo - 1;
            const long o = (((long)b * ho + jh) * wo + jw) * c + ch * EPC;
            dv[k] = *reinterpret_cast<const uint4*>(dout + o);
            if constexpr (EPC == 8) { const uint2 v = *reinterpret_cast<const uint2*>(idx + o); pk[k][0] = v.x; pk[k][1] = v.y; }
            else { pk[k][0] = *reinterpret_cast<const unsigned*>(idx + o); pk[k][1] = 0u; }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (code[k] < 0) continue;
            float d[EPC];
            unpack16<T>(dv[k], d);
#pragma unroll
            for (int e = 0; e < EPC; ++e)
                if (((pk[k][e >> 2] >> (8 * (e & 3))) & 0xffu) == (unsigned)code[k]) g[e] += d[e];
        }
        if constexpr (GATE) {
            float xv[EPC];
            unpack16<T>(*reinterpret_cast<const uint4*>(gt.x + i * 16), xv);
            unpack16<T>(pack16<T>(g), g);            // the gradient as the ungated pass stores it
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                g[e] = (xv[e] * gsc[e] + gsh[e]) > 0.f ? g[e] : 0.f;
                s1[e] += g[e];
                s2[e] += g[e] * (xv[e] - gmean[e]) * ginv[e];
            }
        }
        *reinterpret_cast<uint4*>(dx + i * EPC) = pack16<T>(g);
    }
    if constexpr (GATE) {
        __shared__ float red[256 * EPC * 2];
        const int lanes = 256 / nchunk, chunk = threadIdx.x & (nchunk - 1), rl = threadIdx.x / nchunk;
#pragma unroll
        for (int e = 0; e < EPC; ++e) { red[((rl * nchunk + chunk) * EPC + e) * 2] = s1[e]; red[((rl * nchunk + chunk) * EPC + e) * 2 + 1] = s2[e]; }
        __syncthreads();
        float* row = gt.partial + (long)blockIdx.x * 2 * c;
        for (int t = threadIdx.x; t < c; t += 256) {
            float a = 0.f, b = 0.f;
            for (int k = 0; k < lanes; ++k) { a += red[((k * nchunk) * EPC + t) * 2]; b += red[((k * nchunk) * EPC + t) * 2 + 1]; }
            row[t] = a;
            row[c + t] = b;
        }
    }
}

// The gated pass as a 2 x 2 BLOCK kernel (round 5): a thread owns one channel chunk of the input pixels (2a, 2b), (2a, 2b+1), (2a+1, 2b),
// (2a+1, 2b+1).  Only the four windows (a, b), (a, b+1), (a+1, b), (a+1, b+1) reach them, at fixed positions: centre of (a, b) for the
// even-even pixel, two windows for the mixed ones, all four for the odd-odd one -- four (gradient, position) loads and nine compare-adds
// per element and block instead of sixteen of each behind per-lane `continue`s, added in the old kernel's window order (same dx bits).
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_block_kernel(const T* dout, const unsigned char* idx, T* dx, int n, int h, int w, int c, int ho, int wo,
                                                               const PoolGate gt) {
    static_assert(sizeof(T) == 2, "16-bit element types");
    constexpr int EPC = 8;
    const int nchunk = c / EPC, hb = (h + 1) >> 1, wb = (w + 1) >> 1;
    const unsigned total = (unsigned)n * hb * wb * nchunk;
    float gmean[EPC], ginv[EPC], gsc[EPC], gsh[EPC], s1[EPC], s2[EPC];
    {
        const int chunk = threadIdx.x & (nchunk - 1);
        load_vec<EPC>(gt.mean + chunk * EPC, gmean); load_vec<EPC>(gt.invstd + chunk * EPC, ginv);
        load_vec<EPC>(gt.scale + chunk * EPC, gsc); load_vec<EPC>(gt.shift + chunk * EPC, gsh);
        fill_vec<EPC>(s1, 0.f); fill_vec<EPC>(s2, 0.f);
    }
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const unsigned blk = i / (unsigned)nchunk;
        const int ch = (int)(i - blk * (unsigned)nchunk);
        const unsigned t2 = blk / (unsigned)wb;
        const int bb = (int)(blk - t2 * (unsigned)wb);
        const int b = (int)(t2 / (unsigned)hb), a = (int)(t2 - (unsigned)b * (unsigned)hb);
        // windows W[k] = (a + (k >> 1), bb + (k & 1)); the ones outside the pooled image re-read window 0 and are never used
        const bool wok[4] = {true, bb + 1 < wo, a + 1 < ho, a + 1 < ho && bb + 1 < wo};      // (a < ho and bb < wo always: h, w >= 1)
        uint4 dv[4];
        uint2 pk[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int oh = wok[k] ? a + (k >> 1) : a, ow = wok[k] ? bb + (k & 1) : bb;
            const long o = (((long)b * ho + oh) * wo + ow) * c + ch * EPC;
            dv[k] = *reinterpret_cast<const uint4*>(dout + o);
            pk[k] = *reinterpret_cast<const uint2*>(idx + o);
        }
        const int ih0 = 2 * a, iw0 = 2 * bb;
        const bool pok[4] = {true, iw0 + 1 < w, ih0 + 1 < h, ih0 + 1 < h && iw0 + 1 < w};     // pixel p = (ih0 + (p >> 1), iw0 + (p & 1))
        uint4 xr[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const long o = pok[p] ? (((long)b * h + ih0 + (p >> 1)) * w + iw0 + (p & 1)) * c + ch * EPC : (((long)b * h + ih0) * w + iw0) * c + ch * EPC;
            xr[p] = *reinterpret_cast<const uint4*>(gt.x + o * 2);
        }
        float d[4][EPC];
#pragma unroll
        for (int k = 0; k < 4; ++k) unpack16<T>(dv[k], d[k]);
        // pixel p takes window k's gradient where that window's stored position is code(p, k): position (r, s) = (ih + 1 - 2 oh, iw + 1 - 2 ow)
        //   p0 (even, even): k0 at (1,1)=4              p1 (even, odd): k0 at (1,2)=5, k1 at (1,0)=3
        //   p2 (odd, even):  k0 at (2,1)=7, k2 at (0,1)=1   p3 (odd, odd): k0 (2,2)=8, k1 (2,0)=6, k2 (0,2)=2, k3 (0,0)=0
        constexpr int code[4][4] = {{4, -1, -1, -1}, {5, 3, -1, -1}, {7, -1, 1, -1}, {8, 6, 2, 0}};
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            if (!pok[p]) continue;
            float g[EPC];
#pragma unroll
            for (int e = 0; e < EPC; ++e) g[e] = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (code[p][k] < 0) continue;
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    const unsigned pos = (((e < 4 ? pk[k].x : pk[k].y) >> (8 * (e & 3))) & 0xffu);
                    if (wok[k] && pos == (unsigned)code[p][k]) g[e] += d[k][e];
                }
            }
            float xv[EPC];
            unpack16<T>(xr[p], xv);
            unpack16<T>(pack16<T>(g), g);            // the gradient as the ungated pass stores it
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                g[e] = (xv[e] * gsc[e] + gsh[e]) > 0.f ? g[e] : 0.f;
                s1[e] += g[e];
                s2[e] += g[e] * (xv[e] - gmean[e]) * ginv[e];
            }
            const long o = (((long)b * h + ih0 + (p >> 1)) * w + iw0 + (p & 1)) * c + ch * EPC;
            *reinterpret_cast<uint4*>(dx + o) = pack16<T>(g);
        }
    }
    __shared__ float red[256 * EPC * 2];
    const int lanes = 256 / nchunk, chunk = threadIdx.x & (nchunk - 1), rl = threadIdx.x / nchunk;
#pragma unroll
    for (int e = 0; e < EPC; ++e) { red[((rl * nchunk + chunk) * EPC + e) * 2] = s1[e]; red[((rl * nchunk + chunk) * EPC + e) * 2 + 1] = s2[e]; }
    __syncthreads();
    float* row = gt.partial + (long)blockIdx.x * 2 * c;
    for (int t = threadIdx.x; t < c; t += 256) {
        float a2 = 0.f, b2 = 0.f;
        for (int k = 0; k < lanes; ++k) { a2 += red[((k * nchunk) * EPC + t) * 2]; b2 += red[((k * nchunk) * EPC + t) * 2 + 1]; }
        row[t] = a2;
        row[c + t] = b2;
    }
}

extern "C" int lh_maxpool3x3s2_fwd(const void* x, void* out, unsigned char* idx, int n, int h, int w, int c,
                                   int dtype, void* stream) {
    LH_REQUIRE(x && out && n > 0 && h > 0 && w > 0, "lh_maxpool3x3s2_fwd: bad arguments");
    const int es = lh_dtype_size(dtype);
    LH_REQUIRE(es > 0 && c % (16 / es) == 0, "lh_maxpool3x3s2_fwd: c %d not a multiple of the 16-byte chunk", c);
    const int ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
    const long total = (long)n * ho * wo * (c / (16 / es));
    LH_REQUIRE((long)n * h * w * (c / (16 / es)) < (1L << 31), "lh_maxpool3x3s2_fwd: tensor too large for 32-bit chunk indices");
    const int grid = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    LH_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((maxpool_fwd_kernel<T, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                                                   (const T*)x, (T*)out, idx, n, h, w, c, ho, wo, nullptr, nullptr));
    LH_LAUNCH_CHECK("maxpool_fwd launch");
    return LH_OK;
}

extern "C" int lh_bn_relu_maxpool3x3s2_fwd(const void* x, const float* scale, const float* shift, void* out, unsigned char* idx, int n,
                                           int h, int w, int c, int dtype, void* stream) {
    LH_REQUIRE(x && scale && shift && out && n > 0 && h > 0 && w > 0, "lh_bn_relu_maxpool3x3s2_fwd: bad arguments");
    const int es = lh_dtype_size(dtype);
    LH_REQUIRE(es > 0 && c % (16 / es) == 0, "lh_bn_relu_maxpool3x3s2_fwd: c %d not a multiple of the 16-byte chunk", c);
    const int ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
    const long total = (long)n * ho * wo * (c / (16 / es));
    LH_REQUIRE((long)n * h * w * (c / (16 / es)) < (1L << 31), "lh_bn_relu_maxpool3x3s2_fwd: tensor too large for 32-bit chunk indices");
    static const int strip = [] { const char* e = getenv("LH_POOL_STRIP"); return e ? atoi(e) : 4; }();      // 0: the window-per-thread kernel
    if (es == 2 && strip > 0) {
        const int R = strip >= 4 ? 4 : 2;
        const long items = (long)n * ((ho + R - 1) / R) * wo * (c / 8);
        const int g = (int)((items + 255) / 256 > 8192 ? 8192 : (items + 255) / 256);
        if (dtype == LH_BF16) {
            if (R == 4) hipLaunchKernelGGL((bn_relu_pool_strip_kernel<bf16, 4>), dim3(g), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, (bf16*)out, idx, n, h, w, c, ho, wo, scale, shift);
            else hipLaunchKernelGGL((bn_relu_pool_strip_kernel<bf16, 2>), dim3(g), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, (bf16*)out, idx, n, h, w, c, ho, wo, scale, shift);
        } else {
            if (R == 4) hipLaunchKernelGGL((bn_relu_pool_strip_kernel<f16, 4>), dim3(g), dim3(256), 0, (hipStream_t)stream, (const f16*)x, (f16*)out, idx, n, h, w, c, ho, wo, scale, shift);
            else hipLaunchKernelGGL((bn_relu_pool_strip_kernel<f16, 2>), dim3(g), dim3(256), 0, (hipStream_t)stream, (const f16*)x, (f16*)out, idx, n, h, w, c, ho, wo, scale, shift);
        }
        LH_LAUNCH_CHECK("bn_relu_maxpool_fwd (strip) launch");
        return LH_OK;
    }
    const int grid = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    LH_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((maxpool_fwd_kernel<T, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                                                   (const T*)x, (T*)out, idx, n, h, w, c, ho, wo, scale, shift));
    LH_LAUNCH_CHECK("bn_relu_maxpool_fwd launch");
    return LH_OK;
}

extern "C" int lh_maxpool3x3s2_bwd(const void* dout, const unsigned char* idx, void* dx, int n, int h, int w,
                                   int c, int dtype, void* stream) {
    LH_REQUIRE(dout && dx && idx && n > 0 && h > 0 && w > 0, "lh_maxpool3x3s2_bwd: bad arguments");
    const int es = lh_dtype_size(dtype);
    LH_REQUIRE(es > 0 && c % (16 / es) == 0, "lh_maxpool3x3s2_bwd: c %d not a multiple of the 16-byte chunk", c);
    const int ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
    const long total = (long)n * h * w * (c / (16 / es));
    LH_REQUIRE(total < (1L << 31), "lh_maxpool3x3s2_bwd: tensor too large for 32-bit chunk indices");
    const int grid = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    LH_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((maxpool_bwd_kernel<T, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                                                   (const T*)dout, idx, (T*)dx, n, h, w, c, ho, wo, PoolGate{}));
    LH_LAUNCH_CHECK("maxpool_bwd launch");
    return LH_OK;
}

// rows of the partial-sum slab lh_maxpool3x3s2_bwd_gated writes (= its grid)
extern "C" int lh_maxpool3x3s2_bwd_gated_rows(int n, int h, int w, int c, int dtype) {
    const int es = lh_dtype_size(dtype);
    if (es <= 0 || c % (16 / es)) return 0;
    const long total = (long)n * h * w * (c / (16 / es));
    return (int)((total + 255) / 256 > 1024 ? 1024 : (total + 255) / 256);       // (512 .. 8192 rows re-measured in round 5: 103 .. 130 us, 1024: 104)
}

extern "C" int lh_maxpool3x3s2_bwd_gated(const void* dout, const unsigned char* idx, void* dx, const lh_bn_bwd_gate* gate, int n, int h,
                                         int w, int c, int dtype, void* stream) {
    LH_REQUIRE(dout && dx && idx && gate && n > 0 && h > 0 && w > 0, "lh_maxpool3x3s2_bwd_gated: bad arguments");
    LH_REQUIRE(gate->x && gate->mean && gate->invstd && gate->scale && gate->shift && gate->partial, "lh_maxpool3x3s2_bwd_gated: null pointer in the gate");
    const int es = lh_dtype_size(dtype);
    LH_REQUIRE(es == 2 && c % 8 == 0, "lh_maxpool3x3s2_bwd_gated: 16-bit types, c %d a multiple of 8", c);
    const int nchunk = c / 8;
    LH_REQUIRE((nchunk & (nchunk - 1)) == 0 && nchunk <= 256, "lh_maxpool3x3s2_bwd_gated: c / 8 = %d must be a power of two <= 256", nchunk);
    const int ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
    const long total = (long)n * h * w * nchunk;
    LH_REQUIRE(total < (1L << 31), "lh_maxpool3x3s2_bwd_gated: tensor too large for 32-bit chunk indices");
    const int grid = lh_maxpool3x3s2_bwd_gated_rows(n, h, w, c, dtype);
    PoolGate g;
    g.x = (const unsigned char*)gate->x; g.mean = gate->mean; g.invstd = gate->invstd; g.scale = gate->scale; g.shift = gate->shift; g.partial = gate->partial;
    static const bool block = [] { const char* e = getenv("LH_POOL_BLOCK"); return !e || atoi(e) != 0; }();      // 0: the pixel-per-thread kernel
    if (block) {
        if (dtype == LH_BF16) hipLaunchKernelGGL((maxpool_bwd_block_kernel<bf16>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16*)dout, idx, (bf16*)dx, n, h, w, c, ho, wo, g);
        else hipLaunchKernelGGL((maxpool_bwd_block_kernel<f16>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const f16*)dout, idx, (f16*)dx, n, h, w, c, ho, wo, g);
        LH_LAUNCH_CHECK("maxpool_bwd_gated (block) launch");
        return LH_OK;
    }
    if (dtype == LH_BF16) hipLaunchKernelGGL((maxpool_bwd_kernel<bf16, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16*)dout, idx, (bf16*)dx, n, h, w, c, ho, wo, g);
    else hipLaunchKernelGGL((maxpool_bwd_kernel<f16, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const f16*)dout, idx, (f16*)dx, n, h, w, c, ho, wo, g);
    LH_LAUNCH_CHECK("maxpool_bwd_gated launch");
    return LH_OK;
}
