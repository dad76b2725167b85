// Weight-gradient GEMM for gfx950:  dW[tap][o][i] = sum_pixels dy[pixel][o] * x[pix(pixel,tap)][i]
//
// The contraction index (pixels) is the SLOW index of both NHWC operands, so the MFMA
// fragments (8 consecutive k per lane) are formed with the CDNA4 transposing LDS read
// ds_read_b64_tr_b16 from row-major [pixel][channel] LDS images (16-bit types); the fp32
// path reads its one-k-per-lane fragments with plain ds_read_b32.
//  * tile BO x BI output, K step = 32 pixels (16-bit) / 16 pixels (fp32), 4 waves,
//    double-buffered register staging with one barrier per step (as igemm.hip);
//  * LDS rows are padded so that the 8 pixel rows a half-wave touches per transposed read
//    land on distinct bank groups; the k order inside a step is {4g..4g+3, 16+4g..16+4g+3}
//    for lane group g -- the same permutation for both operands, so the sum is unchanged;
//  * split-K over pixels -> fp32 slabs [split][tap][o][i], folded (deterministically, in
//    split order) into the reference-layout gradient by wgrad_reduce_kernel.
#include "common.h"
#ifndef LH_NT_WREDUCE
#define LH_NT_WREDUCE 0      // debug builds only: the fold reads the split-K slabs (their last use) with non-temporal loads
#endif
#include <algorithm>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "wgrad_ring_kernel.h"
#include "wgrad_cfgs.h"
#include <mutex>

template <typename T> struct WFrag;

// 16-bit types: two transposed reads give the lane 8 k-values of one channel.
template <typename T> struct WFrag {
    static constexpr int KP = 32;
    static constexpr int PAD = 32;
    static __device__ __forceinline__ uint4 load(const unsigned char* tile, int rs, int ctile, int lane) {
        const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
        const unsigned char* a0 = tile + (4 * g + q) * rs + (ctile * 16 + 4 * pp) * 2;
        const unsigned char* a1 = a0 + 16 * rs;
        typedef s16x4 __attribute__((address_space(3))) * lds_p;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a1));
        union { struct { s16x4 l, h; } s; uint4 u; } r;
        r.s.l = lo; r.s.h = hi;
        return r.u;
    }
    static __device__ __forceinline__ void mma(const uint4& a, const uint4& b, f32x4& c);
};
template <> __device__ __forceinline__ void WFrag<bf16>::mma(const uint4& a, const uint4& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ void WFrag<f16>::mma(const uint4& a, const uint4& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// fp32: 16 pixels per step, instruction kk uses pixel 4*kk + (lane>>4).
template <> struct WFrag<float> {
    static constexpr int KP = 16;
    static constexpr int PAD = 64;
    static __device__ __forceinline__ uint4 load(const unsigned char* tile, int rs, int ctile, int lane) {
        const int g = lane >> 4, c = lane & 15;
        const unsigned char* a = tile + g * rs + (ctile * 16 + c) * 4;
        uint4 r;
        r.x = *reinterpret_cast<const unsigned*>(a);
        r.y = *reinterpret_cast<const unsigned*>(a + 4 * rs);
        r.z = *reinterpret_cast<const unsigned*>(a + 8 * rs);
        r.w = *reinterpret_cast<const unsigned*>(a + 12 * rs);
        return r;
    }
    static __device__ __forceinline__ void mma(const uint4& a, const uint4& b, f32x4& c) {
        const f32x4 fa = __builtin_bit_cast(f32x4, a), fb = __builtin_bit_cast(f32x4, b);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[0], fb[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[1], fb[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[2], fb[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[3], fb[3], c, 0, 0, 0);
    }
};

template <typename T, int BO, int BI, int WO, int WI>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int ES = sizeof(T);
    constexpr int EPC = 16 / ES;
    constexpr int KP = WFrag<T>::KP;
    constexpr int RSO = BO * ES + WFrag<T>::PAD, RSI = BI * ES + WFrag<T>::PAD;
    constexpr int CPO = BO * ES / 16, CPI = BI * ES / 16;     // chunks per pixel row
    constexpr int NO = KP * CPO / 256, NI = KP * CPI / 256;   // chunks per thread and step
    constexpr int STAGE = KP * (RSO + RSI);
    constexpr int TO = BO / WO, TI = BI / WI, OT = TO / 16, IT = TI / 16;
    static_assert(WO * WI == 4 && NO >= 1 && NI >= 1, "bad tile");

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wo_ = wave / WI, wi_ = wave % WI;
    const int otile = blockIdx.x / p.i_tiles, itile = blockIdx.x % p.i_tiles;
    const int tap = blockIdx.y, split = blockIdx.z;
    const int dh = p.dh[tap], dw = p.dw[tap];
    const int hw = p.ho * p.wo;
    const long m_begin = (long)split * p.steps_per_split * KP;
    long m_end = m_begin + (long)p.steps_per_split * KP;
    if (m_end > p.M) m_end = p.M;
    const int S = m_begin < m_end ? (int)((m_end - m_begin + KP - 1) / KP) : 0;

    f32x4 acc[OT][IT];
#pragma unroll
    for (int i = 0; i < OT; ++i)
#pragma unroll
        for (int j = 0; j < IT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int orow = tid / CPO, ochunk = tid % CPO;
    const int irow = tid / CPI, ichunk = tid % CPI;
    const int ocol = otile * BO + ochunk * EPC, icol = itile * BI + ichunk * EPC;
    uint4 ro[NO], ri[NI];

    auto gload = [&](int s) {
        const long mb = m_begin + (long)s * KP;
#pragma unroll
        for (int i = 0; i < NO; ++i) {
            const long m = mb + orow + i * (256 / CPO);
            if (m < m_end && ocol < p.n_out)
                ro[i] = *reinterpret_cast<const uint4*>(p.dy + (m * p.dy_pix_stride + ocol) * ES);
            else
                ro[i] = uint4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const long m = mb + irow + i * (256 / CPI);
            bool ok = m < m_end && icol < p.k_run;
            long e = 0;
            if (ok) {
                const int mi = (int)m;
                const int n = mi / hw, rem = mi - n * hw;
                const int a = rem / p.wo, b = rem - a * p.wo;
                const int ih = a * p.sh + dh, iw = b * p.sw + dw;
                ok = (unsigned)ih < (unsigned)p.hi && (unsigned)iw < (unsigned)p.wi;
                e = ((long)(n * p.hi + ih) * p.wi + iw) * p.in_pix_stride + icol;
            }
            ri[i] = ok ? *reinterpret_cast<const uint4*>(p.x + e * ES) : uint4{0u, 0u, 0u, 0u};
        }
    };

    if (S > 0) gload(0);
    for (int s = 0; s < S; ++s) {
        unsigned char* so = smem + (s & 1) * STAGE;
        unsigned char* si = so + KP * RSO;
#pragma unroll
        for (int i = 0; i < NO; ++i)
            *reinterpret_cast<uint4*>(so + (orow + i * (256 / CPO)) * RSO + ochunk * 16) = ro[i];
#pragma unroll
        for (int i = 0; i < NI; ++i)
            *reinterpret_cast<uint4*>(si + (irow + i * (256 / CPI)) * RSI + ichunk * 16) = ri[i];
        __syncthreads();
        if (s + 1 < S) gload(s + 1);
        uint4 fo[OT], fi[IT];
#pragma unroll
        for (int i = 0; i < OT; ++i) fo[i] = WFrag<T>::load(so, RSO, wo_ * OT + i, lane);
#pragma unroll
        for (int j = 0; j < IT; ++j) fi[j] = WFrag<T>::load(si, RSI, wi_ * IT + j, lane);
#pragma unroll
        for (int i = 0; i < OT; ++i)
#pragma unroll
            for (int j = 0; j < IT; ++j) WFrag<T>::mma(fo[i], fi[j], acc[i][j]);
    }

    float* slab = p.slab + ((long)split * p.ntaps + tap) * p.n_out * p.n_in;
    const int q = lane >> 4, c = lane & 15;
#pragma unroll
    for (int i = 0; i < OT; ++i)
#pragma unroll
        for (int j = 0; j < IT; ++j) {
            const int ci = itile * BI + wi_ * TI + j * 16 + c;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = otile * BO + wo_ * TO + i * 16 + q * 4 + r;
                if (o < p.n_out && ci < p.n_in) slab[(long)o * p.n_in + ci] = acc[i][j][r];
            }
        }
}

struct WreduceArgs {
    const float* slab;
    float* grad;
    int n_out, n_in, ntaps, nsplit, accumulate;
    int contig;                  // table launches: this problem folds on the transposing path (wgrad_reduce_contig_body)
    long so, si, sr, ss;
    signed char r[64];
    signed char s[64];
};

// Fast path for the plain layout grad[(o*n_in + i)*ntaps + t]: a workgroup folds 256 consecutive (o, i)
// pairs (coalesced slab reads per tap), transposes through LDS and writes 256*ntaps contiguous floats.
__device__ __forceinline__ void wgrad_reduce_contig_body(const WreduceArgs& p, float* tile, const int bid) {
    const long per = (long)p.n_out * p.n_in;
    const long base = (long)bid * 256;
    const long idx = base + threadIdx.x;
    const int ld = p.ntaps + 1;
    if (idx < per) {
        for (int t = 0; t < p.ntaps; ++t) {
            float a = 0.f;
            for (int sp = 0; sp < p.nsplit; ++sp) a += p.slab[((long)sp * p.ntaps + t) * per + idx];
            tile[threadIdx.x * ld + t] = a;
        }
    }
    __syncthreads();
    const long n_here = per - base < 256 ? per - base : 256;
    const long total = n_here * p.ntaps;
    float* g = p.grad + base * p.ntaps;
    for (long e = threadIdx.x; e < total; e += 256) {
        const int pair = (int)(e / p.ntaps), t = (int)(e - (long)pair * p.ntaps);
        const float v = tile[pair * ld + t];
        g[e] = p.accumulate ? g[e] + v : v;
    }
}
__global__ __launch_bounds__(256) void wgrad_reduce_contig_kernel(const WreduceArgs p) {
    extern __shared__ float tile[];                       // [256][ntaps + 1]
    wgrad_reduce_contig_body(p, tile, blockIdx.x);
}

// Generic layout / small tensors: one thread per (tap, o, FOUR consecutive i) -- 16-byte coalesced slab reads, eight splits
// requested before the first is added (the fold is a latency chain over the splits), the sum itself strictly in split
// order; strided writes.  n_in not a multiple of 4: one element per thread.
static inline long wgrad_reduce_threads(long n_out, long n_in, long ntaps) {
    return (n_in & 3) == 0 ? n_out * n_in * ntaps / 4 : n_out * n_in * ntaps;
}
__device__ __forceinline__ void wgrad_reduce_body(const WreduceArgs& p, const int bid, const int nblk) {
    const long idx = (long)bid * blockDim.x + threadIdx.x;
    const long per = (long)p.n_out * p.n_in;
    if ((p.n_in & 3) == 0) {
        const long per4 = per >> 2;
        if (idx >= per4 * p.ntaps) return;
        // 32-bit index arithmetic: a weight tensor has far fewer than 2^31 elements (reduce_prepare checks)
        const unsigned iu = (unsigned)idx, p4 = (unsigned)per4;
        const int t = (int)(iu / p4);
        const unsigned pair4 = iu - (unsigned)t * p4;
        const unsigned n4 = (unsigned)p.n_in >> 2;
        const int o = (int)(pair4 / n4), i = (int)(pair4 - (unsigned)o * n4) * 4;
        const float4* src = reinterpret_cast<const float4*>(p.slab + (long)t * per + (long)o * p.n_in + i);
        const long stride = (long)p.ntaps * per4;          // float4 units between splits
        float4 a = float4{0.f, 0.f, 0.f, 0.f};
        int sp = 0;
        for (; sp + 8 <= p.nsplit; sp += 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = LH_NT_WREDUCE ? lh_ld_nt(src + (long)(sp + u) * stride) : src[(long)(sp + u) * stride];
#pragma unroll
            for (int u = 0; u < 8; ++u) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
        }
        for (; sp < p.nsplit; ++sp) {
            const float4 v = src[(long)sp * stride];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        float* g = p.grad + o * p.so + i * p.si + p.r[t] * p.sr + p.s[t] * p.ss;
        const float r[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) g[k * p.si] = p.accumulate ? g[k * p.si] + r[k] : r[k];
        return;
    }
    if (idx >= per * p.ntaps) return;
    const unsigned iu = (unsigned)idx, pu = (unsigned)per;
    const int t = (int)(iu / pu);
    const unsigned pair = iu - (unsigned)t * pu;
    const int o = (int)(pair / (unsigned)p.n_in), i = (int)(pair - (unsigned)o * (unsigned)p.n_in);
    float a = 0.f;
    const float* src = p.slab + (long)t * per + pair;
    const long stride = (long)p.ntaps * per;
    int sp = 0;
    for (; sp + 8 <= p.nsplit; sp += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(long)(sp + u) * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) a += v[u];
    }
    for (; sp < p.nsplit; ++sp) a += src[(long)sp * stride];
    float* g = p.grad + o * p.so + i * p.si + p.r[t] * p.sr + p.s[t] * p.ss;
    *g = p.accumulate ? *g + a : a;
}
__global__ void wgrad_reduce_kernel(const WreduceArgs p) { wgrad_reduce_body(p, blockIdx.x, gridDim.x); }
__global__ void wgrad_reduce_multi_kernel(const LhMulti<WreduceArgs> m) {
    int bid, nblk;
    const int i = lh_multi_pick(m, bid, nblk);
    wgrad_reduce_body(m.a[i], bid, nblk);
}
// Table launch: the folds of any number of weight gradients as one grid (items[b] = problem, block index inside it); a problem
// folds on the transposing path or the generic one (WreduceArgs.contig).  With ONE pixel split the fold is the transposition
// [tap][o][i] -> the reference layout.
__global__ __launch_bounds__(256) void wgrad_reduce_table_kernel(const WreduceArgs* __restrict__ tab, const int2* __restrict__ items) {
    extern __shared__ float tile[];
    const int2 it = items[blockIdx.x];
    const WreduceArgs& p = tab[__builtin_amdgcn_readfirstlane(it.x)];
    if (p.contig) wgrad_reduce_contig_body(p, tile, it.y);
    else wgrad_reduce_body(p, it.y, 0);
}

// ------------------------------------------------------------------------------------------------
// The LDS-DMA kernel needs every x row it fetches 16-byte aligned: 16-byte pixel rows, or (NHWC4 stem: 8-byte pixels)
// an even horizontal stride, an aligned image pitch and tap column offsets that are multiples of two pixels.
static bool wgrad_ring_ok(const lh_igemm_desc* d, int es) {
    if (es != 2) return false;
    const long ps = (long)d->in_pix_stride * es;
    if (ps % 16 == 0) return true;
    if ((ps * d->sw) % 16 != 0 || (ps * d->wi) % 16 != 0) return false;
    for (int t = 0; t < d->ntaps; ++t)
        if ((ps * d->dw[t]) % 16 != 0) return false;
    return true;
}

// One launch plan: tile, pixel rows per ring stage (0 = register-staged kernel), ring depth, pixel splits, stages per split.
struct WgradPlan {
    int bo, bi, kps, depth, nsplit, sps;
};

int lh_wgrad_ring_launch_bf16(const WgradArgs& a, const WgradPlan& c, hipStream_t s);
int lh_wgrad_ring_launch_f16(const WgradArgs& a, const WgradPlan& c, hipStream_t s);
int lh_wgrad_ring_multi_launch_bf16(const LhMulti<WgradArgs>& m, const WgradPlan& c, hipStream_t s);
int lh_wgrad_ring_multi_launch_f16(const LhMulti<WgradArgs>& m, const WgradPlan& c, hipStream_t s);

struct WgradCfg { int bo, bi, depth, kps; };
static const WgradCfg kWCfg[] = {
#define X(BO, BI, WO, WI, D, KPS) {BO, BI, D, KPS},
    LH_WGRAD_CFGS(X)
#undef X
};
static const int kNWCfg = (int)(sizeof(kWCfg) / sizeof(WgradCfg));

static bool wcfg_fits(const WgradCfg& c, int n_out, int n_in) {
    if (c.bo > 64 && n_out <= c.bo / 2) return false;
    if (c.bi > 64 && n_in <= c.bi / 2) return false;
    return true;
}

// Pixel splits for a tile shape: `target` workgroups in all, at least `min_stages` ring stages per workgroup, at most
// ~24 MiB of fp32 slab (every split writes, and the fold re-reads, a full copy of the weight tensor).
static void wgrad_splits(const lh_igemm_desc* d, int n_out, int n_in, int bo, int bi, int kps, long target, int* nsplit, int* sps) {
    const long M = (long)d->n * d->ho * d->wo;
    const long stages = (M + kps - 1) / kps;
    const long tiles = (long)((n_out + bo - 1) / bo) * ((n_in + bi - 1) / bi) * d->ntaps;
    const long min_stages = 256 / kps;               // >= 256 pixels per workgroup
    long want = (target + tiles - 1) / tiles;
    long max_split = (stages + min_stages - 1) / min_stages;
    if (max_split > 128) max_split = 128;
    if (want > max_split) want = max_split;
    const long per_split_bytes = (long)n_out * n_in * d->ntaps * 4;
    long by_bytes = (24L << 20) / (per_split_bytes > 0 ? per_split_bytes : 1);
    if (by_bytes < 1) by_bytes = 1;
    if (want > by_bytes && tiles * by_bytes >= 256) want = by_bytes;
    if (want < 1) want = 1;
    *sps = (int)((stages + want - 1) / want);
    *nsplit = (int)((stages + *sps - 1) / *sps);
}

// Resolve the plan of a launch: the descriptor's explicit choice (cfg[5..7], validated) or the static default
// (tile by channel counts, 32-pixel stages, 4-stage ring, ~1024 workgroups).
static int wgrad_plan(const lh_igemm_desc* d, int n_out, int n_in, int dtype, WgradPlan* out) {
    const int es = lh_dtype_size(dtype);
    const bool ring = wgrad_ring_ok(d, es);
    const long M = (long)d->n * d->ho * d->wo;
    if (d->cfg[5] != 0) {
        const int enc = d->cfg[7];
        WgradPlan c = {d->cfg[5], d->cfg[6], (enc >> 16) & 0xff, (enc >> 24) & 0xff, enc & 0xffff, 0};
        bool found = false;
        for (int i = 0; i < kNWCfg && !found; ++i)
            found = kWCfg[i].bo == c.bo && kWCfg[i].bi == c.bi && kWCfg[i].depth == c.depth && kWCfg[i].kps == c.kps;
        if (!ring || !found || c.nsplit < 1) {
            lh_set_error("lh_wgrad: configuration tile %dx%d stage %d depth %d splits %d is not available for this launch", c.bo, c.bi, c.kps, c.depth, c.nsplit);
            return LH_ERR_UNSUPPORTED;
        }
        const WgradCfg wc = {c.bo, c.bi, c.depth, c.kps};
        if (!wcfg_fits(wc, n_out, n_in)) {          // same rule as lh_wgrad_candidates: an explicit choice must fit the gradient
            lh_set_error("lh_wgrad: tile %dx%d does not fit a %d x %d gradient", c.bo, c.bi, n_out, n_in);
            return LH_ERR_ARG;
        }
        const long stages = (M + c.kps - 1) / c.kps;
        c.sps = (int)((stages + c.nsplit - 1) / c.nsplit);
        c.nsplit = (int)((stages + c.sps - 1) / c.sps);
        *out = c;
        return LH_OK;
    }
    WgradPlan c;
    c.bo = n_out > 64 ? 128 : 64;
    c.bi = n_in > 64 ? 128 : 64;
    c.kps = ring ? 32 : 0;
    c.depth = ring ? 4 : 0;
    wgrad_splits(d, n_out, n_in, c.bo, c.bi, ring ? 32 : (dtype == LH_F32 ? 16 : 32), 1024, &c.nsplit, &c.sps);
    *out = c;
    return LH_OK;
}

extern "C" int lh_wgrad_tile(const lh_igemm_desc* d, int n_out, int n_in, int dtype, int* bo, int* bi, int* nsplit, int* ring) {
    LH_REQUIRE(d && bo && bi && nsplit && ring, "lh_wgrad_tile: null pointer");
    WgradPlan c;
    const int rc = wgrad_plan(d, n_out, n_in, dtype, &c);
    if (rc) return rc;
    *bo = c.bo; *bi = c.bi; *nsplit = c.nsplit;
    *ring = c.kps ? c.kps * 10 + c.depth : 0;           // pixel rows per stage * 10 + ring depth; 0 = register-staged kernel
    return LH_OK;
}

// Candidate plans of the LDS-DMA kernel for this launch: every compiled-in (tile, stage, depth) that fits, each with the
// split counts that aim at 256 .. 2048 workgroups.  out: 5 ints per candidate = bo, bi, cfg[7] encoding, workgroups, slab MiB.
extern "C" int lh_wgrad_candidates(const lh_igemm_desc* d, int n_out, int n_in, int dtype, int* out, int max) {
    if (!d || !out || max <= 0 || !wgrad_ring_ok(d, lh_dtype_size(dtype))) return 0;
    int k = 0;
    for (int i = 0; i < kNWCfg; ++i) {
        const WgradCfg& c = kWCfg[i];
        if (!wcfg_fits(c, n_out, n_in)) continue;
        const long tiles = (long)((n_out + c.bo - 1) / c.bo) * ((n_in + c.bi - 1) / c.bi) * d->ntaps;
        const long stages = ((long)d->n * d->ho * d->wo + c.kps - 1) / c.kps;
        int seen[12], nseen = 0;
        const long targets[5] = {256, 512, 1024, 2048, 4096};
        const bool big = c.bo * c.bi >= 256 * 256;
        for (int t = 0; t < (big ? 8 : 11) && k < max; ++t) {
            int ns, sps;
            if (t < 5) {
                wgrad_splits(d, n_out, n_in, c.bo, c.bi, c.kps, targets[t], &ns, &sps);
            } else {
                // also offer the split counts that fill the 256 CUs exactly once, twice, ... (the 8-wave tile runs one workgroup
                // per CU: 1 - 3 rounds; the 4-wave tiles two to four per CU: 1 - 6 x 256 workgroups): the targets above round the
                // split count UP, and a launch of 576 workgroups on 512 slots runs a second round for its last 64
                const long fill = 256L * (t - 4) / tiles;
                if (fill < 1 || fill > 0xffff) continue;
                sps = (int)((stages + fill - 1) / fill);
                ns = (int)((stages + sps - 1) / sps);
                if (sps * c.kps < 256) continue;
            }
            bool dup = ns > 0xffff;
            for (int q = 0; q < nseen; ++q) dup = dup || seen[q] == ns;
            if (dup) continue;
            seen[nseen++] = ns;
            out[5 * k] = c.bo; out[5 * k + 1] = c.bi; out[5 * k + 2] = ns | (c.kps << 16) | (c.depth << 24);
            out[5 * k + 3] = (int)(tiles * ns);
            out[5 * k + 4] = (int)(((long)ns * d->ntaps * n_out * n_in * 4) >> 20);
            ++k;
        }
    }
    return k;
}

extern "C" size_t lh_wgrad_slab_bytes(const lh_igemm_desc* d, int n_out, int n_in, int dtype) {
    WgradPlan c;
    if (wgrad_plan(d, n_out, n_in, dtype, &c)) return 0;
    return (size_t)c.nsplit * d->ntaps * n_out * n_in * sizeof(float);
}

// 16 zero bytes in device memory (one copy per device), the source of every masked LDS-DMA lane.
__device__ __attribute__((aligned(16))) unsigned int lh_wzero_page[4] = {0u, 0u, 0u, 0u};

static const unsigned char* wzero_page() {
    static std::mutex mu;
    static const unsigned char* ptr[64] = {nullptr};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (!ptr[dev]) {
        void* q = nullptr;
        if (hipGetSymbolAddress(&q, HIP_SYMBOL(lh_wzero_page)) != hipSuccess) return nullptr;
        ptr[dev] = (const unsigned char*)q;
    }
    return ptr[dev];
}

template <typename T, int BO, int BI, int WO, int WI>
static int launch_wgrad(const WgradArgs& a, hipStream_t s) {
    constexpr int ES = sizeof(T);
    constexpr int lds = 2 * WFrag<T>::KP * (BO * ES + BI * ES + 2 * WFrag<T>::PAD);
    dim3 grid(ceil_div(a.n_out, BO) * a.i_tiles, a.ntaps, a.nsplit);
    hipLaunchKernelGGL((wgrad_kernel<T, BO, BI, WO, WI>), grid, dim3(256), lds, s, a);
    LH_LAUNCH_CHECK("wgrad launch");
    return LH_OK;
}

static int wgrad_impl(const lh_igemm_desc* d, const void* x, const void* dy, int dy_pix_stride,
                      int n_out, int n_in, float* slab, int dtype, void* stream, int fold_rows,
                      WgradArgs* prep_args = nullptr, WgradPlan* prep_plan = nullptr, const WgradPlan* forced = nullptr) {
    LH_REQUIRE(d && x && dy && slab, "lh_wgrad: null pointer");
    const int es = lh_dtype_size(dtype);
    LH_REQUIRE(es > 0, "lh_wgrad: bad dtype %d", dtype);
    const int epc = 16 / es;
    LH_REQUIRE(d->ntaps > 0 && d->ntaps <= 64, "lh_wgrad: ntaps %d out of range", d->ntaps);
    LH_REQUIRE(n_in == d->k_run && n_in % epc == 0, "lh_wgrad: n_in %d must equal k_run %d and be a multiple of %d", n_in, d->k_run, epc);
    LH_REQUIRE(n_out % epc == 0 && dy_pix_stride % epc == 0 && dy_pix_stride >= n_out, "lh_wgrad: n_out %d / stride %d", n_out, dy_pix_stride);
    WgradPlan c;
    if (forced) {                 // table launches: the table's tile / stage / depth and this problem's own split count
        LH_REQUIRE(wgrad_ring_ok(d, es), "lh_wgrad_table: problem does not run on the LDS-DMA weight-gradient kernel");
        c = *forced;
    } else {
        const int rc = wgrad_plan(d, n_out, n_in, dtype, &c);
        if (rc) return rc;
    }
    WgradArgs a;
    a.x = (const unsigned char*)x; a.dy = (const unsigned char*)dy; a.slab = slab; a.zero = nullptr;
    a.n = d->n; a.hi = d->hi; a.wi = d->wi; a.in_pix_stride = d->in_pix_stride; a.k_run = d->k_run;
    a.ho = d->ho; a.wo = d->wo; a.M = d->n * d->ho * d->wo; a.sh = d->sh; a.sw = d->sw;
    a.dy_pix_stride = dy_pix_stride; a.n_out = n_out; a.n_in = n_in; a.ntaps = d->ntaps;
    for (int i = 0; i < 64; ++i) { a.dh[i] = d->dh[i]; a.dw[i] = d->dw[i]; }
    a.fold_k = 0;
    a.nsplit = c.nsplit; a.steps_per_split = c.sps;
    a.i_tiles = ceil_div(n_in, c.bi);
    a.tiles = ceil_div(n_out, c.bo) * a.i_tiles;
    a.xcd = 1;
    a.adv_n = a.adv_a = a.adv_b = 0;
    hipStream_t s = (hipStream_t)stream;
    if (fold_rows > 1) {
        LH_REQUIRE(c.kps && d->ntaps == 1 && n_in % fold_rows == 0 && (n_in / fold_rows) % epc == 0,
                   "lh_wgrad_rowfold: needs the LDS-DMA kernel, one tap and a run of whole 16-byte chunks per row");
        a.fold_k = n_in / fold_rows;
    }
    if (c.kps) {                  // LDS-DMA ring kernel (16-bit types, 16-byte aligned pixel rows)
        a.zero = wzero_page();
        LH_REQUIRE(a.zero, "lh_wgrad: cannot resolve the zero page on this device");
        const int hw = d->ho * d->wo;
        a.adv_n = c.kps / hw;
        a.adv_a = (c.kps % hw) / d->wo;
        a.adv_b = c.kps % d->wo;
        if (prep_args) {          // lh_wgrad_fused_multi: hand the argument block back instead of launching
            *prep_args = a;
            *prep_plan = c;
            return LH_OK;
        }
        const int r = dtype == LH_BF16 ? lh_wgrad_ring_launch_bf16(a, c, s) : lh_wgrad_ring_launch_f16(a, c, s);
        if (r == 1) {
            lh_set_error("lh_wgrad: no kernel for tile %dx%d stage %d depth %d", c.bo, c.bi, c.kps, c.depth);
            return LH_ERR_UNSUPPORTED;
        }
        return r;
    }
    LH_REQUIRE(!prep_args, "lh_wgrad_fused_multi: problem does not run on the LDS-DMA weight-gradient kernel");
#define LH_WT(T)                                                                 \
    if (c.bo == 128 && c.bi == 128) return launch_wgrad<T, 128, 128, 2, 2>(a, s); \
    if (c.bo == 128 && c.bi == 64) return launch_wgrad<T, 128, 64, 4, 1>(a, s);   \
    if (c.bo == 64 && c.bi == 128) return launch_wgrad<T, 64, 128, 1, 4>(a, s);   \
    return launch_wgrad<T, 64, 64, 2, 2>(a, s);
    switch (dtype) {
        case LH_BF16: { LH_WT(bf16) }
        case LH_F16: { LH_WT(f16) }
        case LH_F32: { LH_WT(float) }
    }
#undef LH_WT
    lh_set_error("lh_wgrad: unsupported dtype %d", dtype);
    return LH_ERR_ARG;
}

extern "C" int lh_wgrad(const lh_igemm_desc* d, const void* x, const void* dy, int dy_pix_stride,
                        int n_out, int n_in, float* slab, int dtype, void* stream) {
    return wgrad_impl(d, x, dy, dy_pix_stride, n_out, n_in, slab, dtype, stream, 0);
}

extern "C" int lh_wgrad_rowfold(const lh_igemm_desc* d, int rows, const void* x, const void* dy, int dy_pix_stride,
                                int n_out, float* slab, int dtype, void* stream) {
    LH_REQUIRE(d && rows >= 1, "lh_wgrad_rowfold: bad arguments");
    return wgrad_impl(d, x, dy, dy_pix_stride, n_out, d->k_run, slab, dtype, stream, rows);
}

extern "C" size_t lh_wgrad_workspace_bytes(const lh_igemm_desc* d, int n_out, int n_in, int dtype) {
    return lh_wgrad_slab_bytes(d, n_out, n_in, dtype);
}

extern "C" int lh_wgrad_fused(const lh_igemm_desc* d, int rows, const void* x, const void* dy, int dy_pix_stride, int n_out,
                              int n_in, void* workspace, float* grad, long so, long si, long sr, long ss, const int* taps_rs,
                              int accumulate, int dtype, void* stream) {
    LH_REQUIRE(d && workspace && grad && taps_rs, "lh_wgrad_fused: null pointer");
    const int rc = wgrad_impl(d, x, dy, dy_pix_stride, n_out, n_in, (float*)workspace, dtype, stream, rows);
    if (rc) return rc;
    return lh_wgrad_reduce(d, (const float*)workspace, grad, n_out, n_in, so, si, sr, ss, taps_rs, accumulate, dtype, stream);
}

// Argument block + launch shape of the fold of one weight gradient: contig = the transposing fast path applies.
static int reduce_prepare(const lh_igemm_desc* d, const float* slab, float* grad, int n_out, int n_in, long so, long si, long sr, long ss,
                          const int* taps_rs, int accumulate, int dtype, WreduceArgs* ap, bool* contig_out) {
    LH_REQUIRE(d && slab && grad && taps_rs, "lh_wgrad_reduce: null pointer");
    LH_REQUIRE(d->ntaps > 0 && d->ntaps <= 64, "lh_wgrad_reduce: ntaps %d out of range", d->ntaps);
    LH_REQUIRE((long)n_out * n_in * d->ntaps < (1L << 31), "lh_wgrad_reduce: weight tensor too large for 32-bit element indices");
    WreduceArgs& a = *ap;
    WgradPlan c;
    const int rc = wgrad_plan(d, n_out, n_in, dtype, &c);
    if (rc) return rc;
    a.nsplit = c.nsplit;
    a.contig = 0;
    a.slab = slab; a.grad = grad; a.n_out = n_out; a.n_in = n_in; a.ntaps = d->ntaps;
    a.accumulate = accumulate; a.so = so; a.si = si; a.sr = sr; a.ss = ss;
    for (int t = 0; t < 64; ++t) {
        a.r[t] = t < d->ntaps ? (signed char)taps_rs[2 * t] : 0;
        a.s[t] = t < d->ntaps ? (signed char)taps_rs[2 * t + 1] : 0;
    }
    const long per = (long)n_out * n_in;
    bool contig = so == (long)n_in * d->ntaps && si == d->ntaps && ss == 1;
    for (int t = 0; t < d->ntaps && contig; ++t) contig = (a.r[t] * sr + a.s[t] * ss) == t;
    *contig_out = contig && per * d->ntaps > (2L << 20);
    return LH_OK;
}

extern "C" int lh_wgrad_reduce(const lh_igemm_desc* d, const float* slab, float* grad, int n_out,
                               int n_in, long so, long si, long sr, long ss, const int* taps_rs,
                               int accumulate, int dtype, void* stream) {
    WreduceArgs a;
    bool contig;
    const int rc = reduce_prepare(d, slab, grad, n_out, n_in, so, si, sr, ss, taps_rs, accumulate, dtype, &a, &contig);
    if (rc) return rc;
    const long per = (long)n_out * n_in;
    if (contig) {
        hipLaunchKernelGGL(wgrad_reduce_contig_kernel, dim3(ceil_div(per, 256)), dim3(256), 256 * (d->ntaps + 1) * sizeof(float),
                           (hipStream_t)stream, a);
        LH_LAUNCH_CHECK("wgrad_reduce launch");
        return LH_OK;
    }
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(ceil_div(wgrad_reduce_threads(n_out, n_in, d->ntaps), 256)), dim3(256), 0, (hipStream_t)stream, a);
    LH_LAUNCH_CHECK("wgrad_reduce launch");
    return LH_OK;
}

// n independent weight gradients (each with its OWN slab workspace) that share tile / stage / ring depth: ONE launch of
// the LDS-DMA kernel for all of them, then ONE launch that folds all their pixel-split slabs (the same layer position of
// HRNet's parallel branches, pose_hrnet.py:139-185).  Problems that need another kernel -- row folds, the register-staged
// kernel, the transposing fold of very large tensors -- run through lh_wgrad_fused one by one.
extern "C" int lh_wgrad_fused_multi(const lh_wgrad_call* calls, int n, int dtype, void* stream) {
    LH_REQUIRE(calls && n >= 1, "lh_wgrad_fused_multi: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    LhMulti<WgradArgs> m;
    LhMulti<WreduceArgs> r;
    WgradPlan plan0 = {0, 0, 0, 0, 0, 0};
    m.n = r.n = 0; m.first[0] = r.first[0] = 0;
    auto flush = [&]() -> int {
        if (m.n == 0) return LH_OK;
        int rc;
        if (m.n > 1) {
            // longest split first: workgroups are dispatched in grid order, and the launch ends with its last one
            int order[LH_MULTI_MAX];
            for (int i = 0; i < m.n; ++i) order[i] = i;
            std::stable_sort(order, order + m.n, [&](int x, int y) { return m.a[x].steps_per_split > m.a[y].steps_per_split; });
            LhMulti<WgradArgs> t = m;
            for (int i = 0; i < m.n; ++i) {
                m.a[i] = t.a[order[i]];
                m.first[i + 1] = m.first[i] + m.a[i].tiles * m.a[i].ntaps * m.a[i].nsplit;
            }
        }
        if (m.n == 1) rc = dtype == LH_BF16 ? lh_wgrad_ring_launch_bf16(m.a[0], plan0, s) : lh_wgrad_ring_launch_f16(m.a[0], plan0, s);
        else rc = dtype == LH_BF16 ? lh_wgrad_ring_multi_launch_bf16(m, plan0, s) : lh_wgrad_ring_multi_launch_f16(m, plan0, s);
        if (rc == 1) {
            lh_set_error("lh_wgrad_fused_multi: no multi-problem kernel for tile %dx%d stage %d depth %d", plan0.bo, plan0.bi, plan0.kps, plan0.depth);
            return LH_ERR_UNSUPPORTED;
        }
        if (rc) return rc;
        if (r.n == 1) hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(r.first[1]), dim3(256), 0, s, r.a[0]);
        else hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3(r.first[r.n]), dim3(256), 0, s, r);
        LH_LAUNCH_CHECK("wgrad_reduce_multi launch");
        m.n = r.n = 0;
        return LH_OK;
    };
    for (int i = 0; i < n; ++i) {
        const lh_wgrad_call& q = calls[i];
        LH_REQUIRE(q.d && q.workspace && q.grad && q.taps_rs, "lh_wgrad_fused_multi: null pointer (problem %d)", i);
        WgradPlan c;
        int rc = wgrad_plan(q.d, q.n_out, q.n_in, dtype, &c);
        if (rc) return rc;
        WreduceArgs ra;
        bool contig = false;
        rc = reduce_prepare(q.d, (const float*)q.workspace, q.grad, q.n_out, q.n_in, q.so, q.si, q.sr, q.ss, q.taps_rs, q.accumulate, dtype, &ra, &contig);
        if (rc) return rc;
        const bool batchable = lh_dtype_size(dtype) == 2 && c.kps && q.rows <= 1 && !contig &&
                               (c.bo <= 128 && c.bi <= 128);           // 4-wave tiles only
        const bool same = m.n == 0 || (c.bo == plan0.bo && c.bi == plan0.bi && c.kps == plan0.kps && c.depth == plan0.depth);
        if (!batchable || !same) {
            rc = flush();
            if (rc) return rc;
        }
        if (!batchable) {
            rc = lh_wgrad_fused(q.d, q.rows, q.x, q.dy, q.dy_pix_stride, q.n_out, q.n_in, q.workspace, q.grad, q.so, q.si, q.sr, q.ss, q.taps_rs,
                                q.accumulate, dtype, stream);
            if (rc) return rc;
            continue;
        }
        rc = wgrad_impl(q.d, q.x, q.dy, q.dy_pix_stride, q.n_out, q.n_in, (float*)q.workspace, dtype, stream, 0, &m.a[m.n], &c);
        if (rc) return rc;
        if (m.n == 0) plan0 = c;
        m.first[m.n + 1] = m.first[m.n] + m.a[m.n].tiles * m.a[m.n].ntaps * m.a[m.n].nsplit;
        r.a[r.n] = ra;
        r.first[r.n + 1] = r.first[r.n] + ceil_div(wgrad_reduce_threads(q.n_out, q.n_in, q.d->ntaps), 256);
        ++m.n; ++r.n;
        if (m.n == LH_MULTI_MAX) {
            rc = flush();
            if (rc) return rc;
        }
    }
    return flush();
}

// ------------------------------------------------------------------------------------------------
// Table launches: the deferred weight gradients of a whole stage as ONE grid + (at most) ONE fold grid.
int lh_wgrad_ring_table_launch_bf16(const WgradArgs* tab, const int2* items, int n_items, int bo, int bi, int kps, int depth, hipStream_t s);
int lh_wgrad_ring_table_launch_f16(const WgradArgs* tab, const int2* items, int n_items, int bo, int bi, int kps, int depth, hipStream_t s);

static inline size_t up256(size_t v) { return (v + 255) / 256 * 256; }

extern "C" int lh_wgrad_table_build(const lh_wgrad_call* calls, int n, int dtype, const int* cfg4, int target_stages, void* workspace,
                                    void* host_blob, size_t host_bytes, lh_wgrad_table_info* info) {
    LH_REQUIRE(calls && n >= 1 && cfg4 && info, "lh_wgrad_table_build: bad arguments");
    LH_REQUIRE(lh_dtype_size(dtype) == 2, "lh_wgrad_table_build: 16-bit types only");
    WgradPlan base = {cfg4[0], cfg4[1], cfg4[2], cfg4[3], 1, 0};
    bool found = false;
    for (int i = 0; i < kNWCfg && !found; ++i)
        found = kWCfg[i].bo == base.bo && kWCfg[i].bi == base.bi && kWCfg[i].depth == base.depth && kWCfg[i].kps == base.kps;
    if (!found) {
        lh_set_error("lh_wgrad_table_build: tile %dx%d stage %d depth %d is not compiled in", base.bo, base.bi, base.kps, base.depth);
        return LH_ERR_UNSUPPORTED;
    }
    // every problem's stage count; the automatic item length aims at ~4 rounds of the machine's workgroup slots
    const int lds = base.depth * base.kps * (base.bo + base.bi) * 2;
    const int per_cu = base.bo * base.bi >= 256 * 256 ? 1 : std::min(2, (160 * 1024) / lds);
    const long slots = 256L * per_cu;
    std::vector<long> stages(n), tiles(n);
    long work = 0, smax = 0;
    for (int i = 0; i < n; ++i) {
        const lh_wgrad_call& q = calls[i];
        LH_REQUIRE(q.d && q.x && q.dy && q.grad && q.taps_rs && q.rows <= 1, "lh_wgrad_table_build: bad problem %d", i);
        LH_REQUIRE(q.d->ntaps > 0 && q.d->ntaps <= 64, "lh_wgrad_table_build: ntaps %d out of range (problem %d)", q.d->ntaps, i);
        const long M = (long)q.d->n * q.d->ho * q.d->wo;
        stages[i] = (M + base.kps - 1) / base.kps;
        tiles[i] = (long)ceil_div(q.n_out, base.bo) * ceil_div(q.n_in, base.bi) * q.d->ntaps;
        work += stages[i] * tiles[i];
        smax = std::max(smax, stages[i]);
    }
    const long min_stages = std::max(1, 256 / base.kps);          // >= 256 pixels per work item
    long L = target_stages;
    if (L <= 0) {
        // split-free when the tiles alone make two rounds of the slots; else the item length that makes about four rounds
        long tsum = 0;
        for (int i = 0; i < n; ++i) tsum += tiles[i];
        L = tsum >= 2 * slots ? smax : std::max(min_stages, work / (4 * slots));
    }
    L = std::max(L, min_stages);
    // blob layout: [WgradArgs x n][int2 items][WreduceArgs x nfold][int2 fold items]
    std::vector<WgradArgs> args(n);
    std::vector<WreduceArgs> folds;
    std::vector<int> order(n);
    size_t ws = 0;
    int fold_lds = 0;
    for (int i = 0; i < n; ++i) {
        const lh_wgrad_call& q = calls[i];
        long ns = (stages[i] + L / 2) / L;                            // nearest split count for items of ~L stages
        const long max_split = std::min(128L, std::max(1L, stages[i] / min_stages));
        ns = std::max(1L, std::min(ns, max_split));
        WgradPlan c = base;
        c.sps = (int)((stages[i] + ns - 1) / ns);
        c.nsplit = (int)((stages[i] + c.sps - 1) / c.sps);
        const size_t per = (size_t)q.n_out * q.n_in;
        // one split, one tap, the gradient dense in [o][i]: the kernel's 16-byte stores ARE the gradient -- no slab, no fold
        const bool direct = c.nsplit == 1 && q.d->ntaps == 1 && !q.accumulate && q.so == q.n_in && q.si == 1 &&
                            ((uintptr_t)q.grad & 15) == 0;
        float* slab = direct ? q.grad : (workspace ? (float*)((char*)workspace + ws) : (float*)(uintptr_t)256);
        if (host_blob) {
            const int rc = wgrad_impl(q.d, q.x, q.dy, q.dy_pix_stride, q.n_out, q.n_in, slab, dtype, nullptr, 0, &args[i], &c, &c);
            if (rc) return rc;
        } else {                                                      // size query (no device needed): the launch shape only
            LH_REQUIRE(wgrad_ring_ok(q.d, 2), "lh_wgrad_table: problem %d does not run on the LDS-DMA weight-gradient kernel", i);
            args[i].i_tiles = ceil_div(q.n_in, c.bi);
            args[i].tiles = ceil_div(q.n_out, c.bo) * args[i].i_tiles;
            args[i].ntaps = q.d->ntaps; args[i].nsplit = c.nsplit; args[i].steps_per_split = c.sps;
        }
        if (!direct) {
            WreduceArgs ra;
            bool contig = false;
            const int rr = reduce_prepare(q.d, slab, q.grad, q.n_out, q.n_in, q.so, q.si, q.sr, q.ss, q.taps_rs, q.accumulate, dtype, &ra, &contig);
            if (rr) return rr;
            ra.nsplit = c.nsplit;
            ra.contig = contig ? 1 : 0;
            if (contig) fold_lds = std::max(fold_lds, (int)(256 * (q.d->ntaps + 1) * sizeof(float)));
            folds.push_back(ra);
            ws += up256((size_t)c.nsplit * q.d->ntaps * per * sizeof(float));
        }
        order[i] = i;
    }
    // Work-item order.  Workgroup ids are dealt round-robin over the 8 XCDs (block b runs on XCD b % 8, each with a private L2), and the
    // items that share operand rows are the taps x tiles of ONE pixel split of ONE layer (every tap of a 3x3 walks the same x / dy rows,
    // the channel tiles of a 1x1 share one of the two operands): such a GROUP is laid out on ONE XCD -- consecutive positions of that
    // XCD's sub-sequence b = 8 k + x -- so that its members run side by side and read their rows from that L2 once instead of once per
    // item from the fabric (PMC, R50's 32-layer table: 3.2 GB of L2 misses per launch for ~1 GB of operands with the per-layer order).
    // Groups (at most 16 items) are taken longest item first and handed to the XCD with the least work so far, so every XCD's queue
    // runs from long to short items and the queues are equal in cost; queues that hold fewer items end in empty items, which exit at once.
    // LH_WGRAD_TABLE_XCD=0: the plain longest-first order with the per-layer XCD remap.
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return args[x].steps_per_split > args[y].steps_per_split; });
    const char* xsw = getenv("LH_WGRAD_TABLE_XCD");
    const bool by_xcd = !(xsw && atoi(xsw) == 0);
    struct Group { int prob, first, count; };                         // `count` consecutive block indices of problem `prob` from `first`
    std::vector<Group> groups;
    std::vector<int2> item_list;
    if (by_xcd) {
        // groups of at most 16 items (half an XCD's CUs for the 8-wave tile): the taps x tiles of one split, cut where there are more
        for (int k = 0; k < n; ++k) {
            const int i = order[k], per = args[i].tiles * args[i].ntaps;
            args[i].xcd = 0;                                          // the table places the items; the body must not remap them
            const int parts = (per + 15) / 16, sz = (per + parts - 1) / parts;
            for (int sp = 0; sp < args[i].nsplit; ++sp)
                for (int o = 0; o < per; o += sz) groups.push_back({i, sp * per + o, std::min(sz, per - o)});
        }
        std::stable_sort(groups.begin(), groups.end(), [&](const Group& a, const Group& b) {
            return args[a.prob].steps_per_split > args[b.prob].steps_per_split;
        });
        // longest items first, each group to the XCD with the least work so far: every XCD's queue runs long -> short, equal in cost
        std::vector<int2> queue[8];
        long load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (const Group& g : groups) {
            int x = 0;
            for (int y = 1; y < 8; ++y)
                if (load[y] < load[x]) x = y;
            for (int b = 0; b < g.count; ++b) queue[x].push_back(int2{g.prob, g.first + b});
            load[x] += (long)g.count * args[g.prob].steps_per_split;
        }
        size_t len = 0;
        for (int x = 0; x < 8; ++x) len = std::max(len, queue[x].size());
        for (size_t pos = 0; pos < len; ++pos)
            for (int x = 0; x < 8; ++x) item_list.push_back(pos < queue[x].size() ? queue[x][pos] : int2{-1, 0});
        while (!item_list.empty() && item_list.back().x < 0) item_list.pop_back();
    } else {
        for (int k = 0; k < n; ++k) {
            const int i = order[k], cnt = args[i].tiles * args[i].ntaps * args[i].nsplit;
            for (int b = 0; b < cnt; ++b) item_list.push_back(int2{i, b});
        }
    }
    {   // every (problem, block) exactly once, whatever the order: a block left out would leave its slab slice unwritten
        std::vector<long> first_of(n + 1, 0);
        for (int i = 0; i < n; ++i) first_of[i + 1] = first_of[i] + (long)args[i].tiles * args[i].ntaps * args[i].nsplit;
        std::vector<unsigned char> seen((size_t)first_of[n], 0);
        long placed = 0;
        for (const int2& it : item_list) {
            if (it.x < 0) continue;
            const long cnt = first_of[it.x + 1] - first_of[it.x];
            LH_REQUIRE(it.x < n && it.y >= 0 && it.y < cnt && !seen[(size_t)(first_of[it.x] + it.y)], "lh_wgrad_table_build: internal error in the work-item order");
            seen[(size_t)(first_of[it.x] + it.y)] = 1;
            ++placed;
        }
        LH_REQUIRE(placed == first_of[n], "lh_wgrad_table_build: internal error: %ld of %ld work items placed", placed, first_of[n]);
    }
    long n_items = (long)item_list.size(), n_fold_items = 0;
    for (const WreduceArgs& r : folds)
        n_fold_items += r.contig ? ceil_div((long)r.n_out * r.n_in, 256) : ceil_div(wgrad_reduce_threads(r.n_out, r.n_in, r.ntaps), 256);
    LH_REQUIRE(n_items < (1L << 30) && n_fold_items < (1L << 30), "lh_wgrad_table_build: too many work items");
    lh_wgrad_table_info t;
    memset(&t, 0, sizeof(t));
    t.bo = base.bo; t.bi = base.bi; t.kps = base.kps; t.depth = base.depth;
    t.n_problems = n; t.n_fold = (int)folds.size();
    t.n_items = (int)n_items; t.n_fold_items = (int)n_fold_items; t.fold_lds = fold_lds;
    t.target_stages = (int)L;
    t.off_items = up256(sizeof(WgradArgs) * n);
    t.off_fold_args = t.off_items + up256(sizeof(int2) * n_items);
    t.off_fold_items = t.off_fold_args + up256(sizeof(WreduceArgs) * folds.size());
    t.table_bytes = t.off_fold_items + up256(sizeof(int2) * n_fold_items);
    t.workspace_bytes = std::max(ws, (size_t)256);
    for (int i = 0; i < n; ++i) t.nsplit_max = std::max(t.nsplit_max, args[i].nsplit);
    *info = t;
    if (!host_blob) return LH_OK;                                   // size query
    LH_REQUIRE(workspace && host_bytes >= t.table_bytes, "lh_wgrad_table_build: blob of %zu bytes needs %zu (and a workspace)", host_bytes, t.table_bytes);
    char* blob = (char*)host_blob;
    memset(blob, 0, t.table_bytes);
    memcpy(blob, args.data(), sizeof(WgradArgs) * n);
    memcpy(blob + t.off_items, item_list.data(), sizeof(int2) * item_list.size());
    if (!folds.empty()) memcpy(blob + t.off_fold_args, folds.data(), sizeof(WreduceArgs) * folds.size());
    int2* fi = (int2*)(blob + t.off_fold_items);
    for (int k = 0; k < (int)folds.size(); ++k) {
        const WreduceArgs& r = folds[k];
        const int cnt = r.contig ? ceil_div((long)r.n_out * r.n_in, 256) : ceil_div(wgrad_reduce_threads(r.n_out, r.n_in, r.ntaps), 256);
        for (int b = 0; b < cnt; ++b) *fi++ = int2{k, b};
    }
    return LH_OK;
}

extern "C" int lh_wgrad_table_run(const void* table, const lh_wgrad_table_info* info, int dtype, void* stream) {
    LH_REQUIRE(table && info && info->n_items > 0, "lh_wgrad_table_run: bad arguments");
    LH_REQUIRE(dtype == LH_BF16 || dtype == LH_F16, "lh_wgrad_table_run: 16-bit types only");
    hipStream_t s = (hipStream_t)stream;
    const char* blob = (const char*)table;
    const WgradArgs* tab = (const WgradArgs*)blob;
    const int2* items = (const int2*)(blob + info->off_items);
    int rc = LH_OK;
    if (info->run_parts != 2)
        rc = dtype == LH_BF16 ? lh_wgrad_ring_table_launch_bf16(tab, items, info->n_items, info->bo, info->bi, info->kps, info->depth, s)
                              : lh_wgrad_ring_table_launch_f16(tab, items, info->n_items, info->bo, info->bi, info->kps, info->depth, s);
    if (rc == 1) {
        lh_set_error("lh_wgrad_table_run: no kernel for tile %dx%d stage %d depth %d", info->bo, info->bi, info->kps, info->depth);
        return LH_ERR_UNSUPPORTED;
    }
    if (rc) return rc;
    if (info->n_fold_items > 0 && info->run_parts != 1) {
        hipLaunchKernelGGL(wgrad_reduce_table_kernel, dim3(info->n_fold_items), dim3(256), info->fold_lds, s,
                           (const WreduceArgs*)(blob + info->off_fold_args), (const int2*)(blob + info->off_fold_items));
        LH_LAUNCH_CHECK("wgrad_reduce_table launch");
    }
    return LH_OK;
}
