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
#include <stdlib.h>

struct WgradArgs {
    const unsigned char* x;
    const unsigned char* dy;
    float* slab;
    int n, hi, wi, in_pix_stride, k_run;
    int ho, wo, M, sh, sw;
    int dy_pix_stride, n_out, n_in;
    int ntaps, nsplit, steps_per_split, i_tiles;
    int tiles, xcd;              // ring kernel: 1-D grid of tiles * ntaps * nsplit work items, XCD-aware order
    int fold_k;                  // ring kernel, row fold (lh_wgrad_rowfold): input index i = row * fold_k + k, 0 = off
    signed char dh[64];
    signed char dw[64];
};

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

// ------------------------------------------------------------------------------------------------
// LDS-DMA ring version for the 16-bit types (production path).  Same GEMM as wgrad_kernel, but both
// operands arrive by global_load_lds_dwordx4 into a D-stage ring (see igemm_ring.hip for the protocol:
// one raw s_barrier per K step, counted vmcnt leaving D-2 stages in flight).  LDS image per operand:
// [32 pixel rows][B*2 bytes], unpadded (an LDS-DMA writes 1 KiB lane-linear), 32-byte granules XOR-swizzled by
// the row so the 8 pixel rows a half-wave touches per transposed read (ds_read_b64_tr_b16) hit distinct banks;
// the swizzle is applied to the per-lane SOURCE address and to the read address.
__device__ __attribute__((aligned(16))) unsigned int lh_wzero_page[4] = {0u, 0u, 0u, 0u};

#ifndef LH_ABL      // debug-only ablation builds, see igemm_ring.hip / tools/ablate.sh
#define LH_ABL 0
#endif

template <int ROWB> __device__ __forceinline__ int wswz(int row) {
    return ROWB >= 256 ? (row & 7) : ((row >> 1) & 3);
}

template <typename T, int BO, int BI, int WO, int WI, int D>
__global__ __launch_bounds__(64 * WO * WI) void wgrad_ring_kernel(const WgradArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    static_assert(sizeof(T) == 2, "16-bit types only");
    constexpr int KP = 32;
    constexpr int RBO = BO * 2, RBI = BI * 2;                 // bytes per pixel row
    constexpr int RPO = 1024 / RBO, RPI = 1024 / RBI;         // pixel rows per LDS-DMA instruction
    constexpr int NWAVE = WO * WI;                           // 4 waves, or 8 for the 256 x 256 tile
    constexpr int NO = KP / RPO / NWAVE, NI = KP / RPI / NWAVE;   // instructions per wave and stage
    constexpr int L = NO + NI;
    constexpr int STAGE = KP * (RBO + RBI);
    constexpr int TO = BO / WO, TI = BI / WI, OT = TO / 16, IT = TI / 16;
    static_assert((NWAVE == 4 || NWAVE == 8) && NO >= 1 && NI >= 1 && D >= 2 && D <= 4, "bad tile");
    typedef __attribute__((address_space(3))) void* lds_p;
    typedef const __attribute__((address_space(1))) void* gbl_p;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wo_ = wave / WI, wi_ = wave % WI;
    // work item w = (split, tap, tile), tile fastest: every XCD gets a contiguous range of pixel splits with all their
    // taps and tiles, which re-read the same dy / x rows from that XCD's L2 instead of the Infinity Cache
    const int w = p.xcd ? lh_xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    const int tile = w % p.tiles, tap = (w / p.tiles) % p.ntaps, split = w / (p.tiles * p.ntaps);
    const int otile = tile / p.i_tiles, itile = tile % p.i_tiles;
    const int dh = p.dh[tap], dw = p.dw[tap];
    const int hw = p.ho * p.wo;
    const long m_begin = (long)split * p.steps_per_split * KP;
    long m_end = m_begin + (long)p.steps_per_split * KP;
    if (m_end > p.M) m_end = p.M;
    const int S = m_begin < m_end ? (int)((m_end - m_begin + KP - 1) / KP) : 0;
    const unsigned char* zero = reinterpret_cast<const unsigned char*>(lh_wzero_page);
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;

    // per-lane source bookkeeping: instruction q = 4*j + wave covers rows [q*RP, (q+1)*RP)
    int orow[NO], ocol[NO], irow[NI], icol[NI], idh[NI], ick[NI];
#pragma unroll
    for (int j = 0; j < NO; ++j) {
        const int q = NWAVE * j + wave;
        const int r = q * RPO + lane / (RBO / 16), c16 = lane % (RBO / 16);
        orow[j] = r;
        ocol[j] = otile * BO + ((((c16 >> 1) ^ wswz<RBO>(r)) << 1) | (c16 & 1)) * 8;
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int q = NWAVE * j + wave;
        const int r = q * RPI + lane / (RBI / 16), c16 = lane % (RBI / 16);
        irow[j] = r;
        icol[j] = itile * BI + ((((c16 >> 1) ^ wswz<RBI>(r)) << 1) | (c16 & 1)) * 8;
        // row fold: the gradient's input index covers `rows` kernel rows of fold_k contiguous elements each; this lane's
        // chunk belongs to kernel row icol / fold_k (an extra input-row offset) and element icol % fold_k of the run
        idh[j] = p.fold_k ? icol[j] / p.fold_k : 0;
        ick[j] = p.fold_k ? icol[j] % p.fold_k : icol[j];
    }

    auto issue = [&](int s, int slot) {
        unsigned char* st = smem + slot * STAGE;
        const long mb = m_begin + (long)s * KP;
#pragma unroll
        for (int j = 0; j < NO; ++j) {
            const long m = mb + orow[j];
            const bool ok = (int)(m < m_end) & (int)(ocol[j] < p.n_out);
            const unsigned char* src = p.dy + (m * p.dy_pix_stride + ocol[j]) * 2;
            src = ok ? src : zero;
            if (!(LH_ABL & 4)) __builtin_amdgcn_global_load_lds((gbl_p)src, (lds_p)(st + (NWAVE * j + wave) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const long m = mb + irow[j];
            const int mi = m < m_end ? (int)m : 0;
            const int n = mi / hw, rem = mi - n * hw;
            const int a = rem / p.wo, b = rem - a * p.wo;
            const int ih = a * p.sh + dh + idh[j], iw = b * p.sw + dw;
            const bool ok = (int)(m < m_end) & (int)(icol[j] < p.k_run) & (int)((unsigned)ih < (unsigned)p.hi) &
                            (int)((unsigned)iw < (unsigned)p.wi);
            const unsigned char* src = p.x + (((long)(n * p.hi + ih) * p.wi + iw) * p.in_pix_stride + ick[j]) * 2;
            src = ok ? src : zero;
            if (!(LH_ABL & 4)) __builtin_amdgcn_global_load_lds((gbl_p)src, (lds_p)(st + KP * RBO + (NWAVE * j + wave) * 1024), 16, 0, 0);
        }
    };

    f32x4 acc[OT][IT];
#pragma unroll
    for (int i = 0; i < OT; ++i)
#pragma unroll
        for (int j = 0; j < IT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    int issued = 0;
#pragma unroll
    for (int s = 0; s < D - 1; ++s)
        if (issued < S) { issue(issued, issued % D); ++issued; }

    // transposed-read addresses: lane (group g, q, pp) reads pixel row 4g+q (and +16), 4 channels at 4*pp of a 16-channel tile
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int row0 = 4 * g + q, row1 = row0 + 16;
    unsigned ao[OT][2], ai[IT][2];
#pragma unroll
    for (int i = 0; i < OT; ++i) {
        const int ct = wo_ * OT + i;                              // 32-byte granule index inside the row
        ao[i][0] = row0 * RBO + ((ct ^ wswz<RBO>(row0)) << 5) + pp * 8;
        ao[i][1] = row1 * RBO + ((ct ^ wswz<RBO>(row1)) << 5) + pp * 8;
    }
#pragma unroll
    for (int j = 0; j < IT; ++j) {
        const int ct = wi_ * IT + j;
        ai[j][0] = KP * RBO + row0 * RBI + ((ct ^ wswz<RBI>(row0)) << 5) + pp * 8;
        ai[j][1] = KP * RBO + row1 * RBI + ((ct ^ wswz<RBI>(row1)) << 5) + pp * 8;
    }

    for (int s = 0; s < S; ++s) {
        const int ahead = issued - 1 - s;
        if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * L) : "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (issued < S) { issue(issued, issued % D); ++issued; }
        const unsigned st = lds_base + (s % D) * STAGE;
        uint2 fo[OT][2], fi[IT][2];
#pragma unroll
        for (int i = 0; i < OT; ++i) {
            if (LH_ABL & 2) { fo[i][0] = fo[i][1] = uint2{st, st}; continue; }
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(fo[i][0]) : "v"(st + ao[i][0]));
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(fo[i][1]) : "v"(st + ao[i][1]));
        }
#pragma unroll
        for (int j = 0; j < IT; ++j) {
            if (LH_ABL & 2) { fi[j][0] = fi[j][1] = uint2{st, st}; continue; }
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(fi[j][0]) : "v"(st + ai[j][0]));
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(fi[j][1]) : "v"(st + ai[j][1]));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < OT; ++i)
#pragma unroll
            for (int j = 0; j < IT; ++j) {
                const uint4 a = uint4{fo[i][0].x, fo[i][0].y, fo[i][1].x, fo[i][1].y};
                const uint4 b = uint4{fi[j][0].x, fi[j][0].y, fi[j][1].x, fi[j][1].y};
                if (!(LH_ABL & 1)) WFrag<T>::mma(a, b, acc[i][j]);
            }
    }

    if (LH_ABL & 8) { if (acc[0][0][0] == 123.456f) p.slab[0] = 1.f; return; }
    float* slab = p.slab + ((long)split * p.ntaps + tap) * p.n_out * p.n_in;
    const int qq = lane >> 4, cc = lane & 15;
#pragma unroll
    for (int i = 0; i < OT; ++i)
#pragma unroll
        for (int j = 0; j < IT; ++j) {
            const int ci = itile * BI + wi_ * TI + j * 16 + cc;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = otile * BO + wo_ * TO + i * 16 + qq * 4 + r;
                if (o < p.n_out && ci < p.n_in) slab[(long)o * p.n_in + ci] = acc[i][j][r];
            }
        }
}

struct WreduceArgs {
    const float* slab;
    float* grad;
    int n_out, n_in, ntaps, nsplit, accumulate;
    long so, si, sr, ss;
    signed char r[64];
    signed char s[64];
};

// Fast path for the plain layout grad[(o*n_in + i)*ntaps + t]: a workgroup folds 256 consecutive (o, i)
// pairs (coalesced slab reads per tap), transposes through LDS and writes 256*ntaps contiguous floats.
__global__ __launch_bounds__(256) void wgrad_reduce_contig_kernel(const WreduceArgs p) {
    extern __shared__ float tile[];                       // [256][ntaps + 1]
    const long per = (long)p.n_out * p.n_in;
    const long base = (long)blockIdx.x * 256;
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

// Generic layout / small tensors: one thread per (tap, o, i) -- coalesced slab reads, strided writes.
__global__ void wgrad_reduce_kernel(const WreduceArgs p) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long per = (long)p.n_out * p.n_in;
    if (idx >= per * p.ntaps) return;
    const int t = (int)(idx / per);
    const long pair = idx - (long)t * per;
    const int o = (int)(pair / p.n_in), i = (int)(pair - (long)o * p.n_in);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    const float* src = p.slab + (long)t * per + pair;
    const long stride = (long)p.ntaps * per;
    int sp = 0;
    for (; sp + 4 <= p.nsplit; sp += 4) {
        a0 += src[(long)sp * stride];
        a1 += src[(long)(sp + 1) * stride];
        a2 += src[(long)(sp + 2) * stride];
        a3 += src[(long)(sp + 3) * stride];
    }
    for (; sp < p.nsplit; ++sp) a0 += src[(long)sp * stride];
    const float a = (a0 + a1) + (a2 + a3);
    float* g = p.grad + o * p.so + i * p.si + p.r[t] * p.sr + p.s[t] * p.ss;
    *g = p.accumulate ? *g + a : a;
}

// ------------------------------------------------------------------------------------------------
// The LDS-DMA kernel needs every x row it fetches 16-byte aligned: 16-byte pixel rows, or (NHWC4 stem: 8-byte pixels)
// an even horizontal stride, an aligned image pitch and tap column offsets that are multiples of two pixels.
static bool wgrad_ring_ok(const lh_igemm_desc* d, int es) {
    if (es != 2 || getenv("LH_NO_WGRAD_RING")) return false;
    const long ps = (long)d->in_pix_stride * es;
    if (ps % 16 == 0) return true;
    if (getenv("LH_NO_STEM_RING")) return false;
    if ((ps * d->sw) % 16 != 0 || (ps * d->wi) % 16 != 0) return false;
    for (int t = 0; t < d->ntaps; ++t)
        if ((ps * d->dw[t]) % 16 != 0) return false;
    return true;
}

// Split-K plan of one tile shape: how many pixel splits, steps per split.
static void wgrad_splits(const lh_igemm_desc* d, int n_out, int n_in, int bo, int bi, int kp, long target, long* nsplit, long* sps) {
    const long M = (long)d->n * d->ho * d->wo;
    const long steps = (M + kp - 1) / kp;
    const long tiles = (long)((n_out + bo - 1) / bo) * ((n_in + bi - 1) / bi) * d->ntaps;
    static long min_steps = 0;                       // tuning knob: fewest K steps per workgroup
    if (!min_steps) { const char* e = getenv("LH_WGRAD_MINSTEPS"); min_steps = e ? atol(e) : 8; }
    long want = (target + tiles - 1) / tiles;        // aim at >= ~3 workgroups per CU
    long max_split = (steps + min_steps - 1) / min_steps;
    if (max_split > 128) max_split = 128;            // bound the slab traffic of the fold
    if (want > max_split) want = max_split;
    // every split writes (and the fold re-reads) a full fp32 copy of the weight tensor: keep the slab <= ~24 MiB
    const long per_split_bytes = (long)n_out * n_in * d->ntaps * 4;
    long by_bytes = (24L << 20) / (per_split_bytes > 0 ? per_split_bytes : 1);
    if (by_bytes < 1) by_bytes = 1;
    if (want > by_bytes && tiles * by_bytes >= 256) want = by_bytes;
    if (want < 1) want = 1;
    *sps = (steps + want - 1) / want;
    *nsplit = (steps + *sps - 1) / *sps;
}

static void wgrad_plan(const lh_igemm_desc* d, int n_out, int n_in, int dtype, int* bo, int* bi,
                       int* nsplit, int* steps_per_split) {
    const int kp = dtype == LH_F32 ? 16 : 32;
    *bo = n_out > 64 ? 128 : 64;
    *bi = n_in > 64 ? 128 : 64;
    static long target = 0;                          // tuning knob: workgroups aimed at
    if (!target) { const char* e = getenv("LH_WGRAD_WANT"); target = e ? atol(e) : 1024; }
    long ns, sps;
    wgrad_splits(d, n_out, n_in, *bo, *bi, kp, target, &ns, &sps);
    // 256 x 256 tile (8 waves, one workgroup per CU, LDS-DMA ring kernel only): half the operand bytes per FLOP of the
    // 128 x 128 tile, but a quarter of the tiles, so more pixel splits -- each of which writes a full fp32 copy of the
    // weight tensor that the fold re-reads.  Chosen where the estimated saving on operand traffic (served at the LDS-DMA
    // rate) exceeds the extra slab traffic (HBM rate).  LH_WGRAD_BIG=0 never, 2 always (where the shape allows).
    const char* be = getenv("LH_WGRAD_BIG");          // read per call: the parity test flips it inside one process
    const int big = be ? atoi(be) : 1;
    const bool ringable = wgrad_ring_ok(d, lh_dtype_size(dtype));
    if (big && ringable && *bo == 128 && *bi == 128 && n_out % 256 == 0 && n_in % 256 == 0) {
        long nb, sb;
        wgrad_splits(d, n_out, n_in, 256, 256, kp, target / 2, &nb, &sb);
        const double M = (double)d->n * d->ho * d->wo, wbytes = (double)n_out * n_in * d->ntaps * 4;
        const double tiles_s = (double)(n_out / 128) * (n_in / 128) * d->ntaps, tiles_b = tiles_s / 4;
        const double t_small = M * 512 * tiles_s / 10e12 + 2 * ns * wbytes / 5e12;
        const double t_big = M * 1024 * tiles_b / 10e12 + 2 * nb * wbytes / 5e12;
        static double margin = 0.0;
        if (margin == 0.0) { const char* e = getenv("LH_WGRAD_BIG_MARGIN"); margin = e ? atof(e) : 0.85; }
        if (big > 1 || (t_big < margin * t_small && tiles_b * nb >= 192)) { *bo = 256; *bi = 256; ns = nb; sps = sb; }
    }
    *nsplit = (int)ns;
    *steps_per_split = (int)sps;
}

extern "C" int lh_wgrad_tile(const lh_igemm_desc* d, int n_out, int n_in, int dtype, int* bo, int* bi, int* nsplit, int* ring) {
    LH_REQUIRE(d && bo && bi && nsplit && ring, "lh_wgrad_tile: null pointer");
    int sps;
    wgrad_plan(d, n_out, n_in, dtype, bo, bi, nsplit, &sps);
    const int es = lh_dtype_size(dtype);
    *ring = wgrad_ring_ok(d, es) ? 4 : 0;
    return LH_OK;
}

extern "C" size_t lh_wgrad_slab_bytes(const lh_igemm_desc* d, int n_out, int n_in, int dtype) {
    int bo, bi, ns, sps;
    wgrad_plan(d, n_out, n_in, dtype, &bo, &bi, &ns, &sps);
    return (size_t)ns * d->ntaps * n_out * n_in * sizeof(float);
}

template <typename T, int BO, int BI, int WO, int WI>
static int launch_wgrad_ring(const WgradArgs& a, hipStream_t s) {
    constexpr int D = 4;
    constexpr int lds = D * 32 * (BO * 2 + BI * 2);
    static bool attr_done = false;
    if (!attr_done && lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_ring_kernel<T, BO, BI, WO, WI, D>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) {
            lh_set_error("wgrad_ring: cannot raise dynamic LDS to %d bytes: %s", lds, hipGetErrorString(e));
            return LH_ERR_HIP;
        }
        attr_done = true;
    }
    WgradArgs b = a;
    static int xcd = -1;
    if (xcd < 0) xcd = getenv("LH_NO_XCD") ? 0 : 1;
    b.tiles = ceil_div(a.n_out, BO) * a.i_tiles;
    b.xcd = xcd;
    dim3 grid(b.tiles * a.ntaps * a.nsplit);
    hipLaunchKernelGGL((wgrad_ring_kernel<T, BO, BI, WO, WI, D>), grid, dim3(64 * WO * WI), lds, s, b);
    LH_LAUNCH_CHECK("wgrad_ring launch");
    return LH_OK;
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
                      int n_out, int n_in, float* slab, int dtype, void* stream, int fold_rows) {
    LH_REQUIRE(d && x && dy && slab, "lh_wgrad: null pointer");
    const int es = lh_dtype_size(dtype);
    LH_REQUIRE(es > 0, "lh_wgrad: bad dtype %d", dtype);
    const int epc = 16 / es;
    LH_REQUIRE(d->ntaps > 0 && d->ntaps <= 64, "lh_wgrad: ntaps %d out of range", d->ntaps);
    LH_REQUIRE(n_in == d->k_run && n_in % epc == 0, "lh_wgrad: n_in %d must equal k_run %d and be a multiple of %d", n_in, d->k_run, epc);
    LH_REQUIRE(n_out % epc == 0 && dy_pix_stride % epc == 0 && dy_pix_stride >= n_out, "lh_wgrad: n_out %d / stride %d", n_out, dy_pix_stride);
    WgradArgs a;
    a.x = (const unsigned char*)x; a.dy = (const unsigned char*)dy; a.slab = slab;
    a.n = d->n; a.hi = d->hi; a.wi = d->wi; a.in_pix_stride = d->in_pix_stride; a.k_run = d->k_run;
    a.ho = d->ho; a.wo = d->wo; a.M = d->n * d->ho * d->wo; a.sh = d->sh; a.sw = d->sw;
    a.dy_pix_stride = dy_pix_stride; a.n_out = n_out; a.n_in = n_in; a.ntaps = d->ntaps;
    for (int i = 0; i < 64; ++i) { a.dh[i] = d->dh[i]; a.dw[i] = d->dw[i]; }
    a.fold_k = 0;
    int bo, bi;
    wgrad_plan(d, n_out, n_in, dtype, &bo, &bi, &a.nsplit, &a.steps_per_split);
    a.i_tiles = ceil_div(n_in, bi);
    hipStream_t s = (hipStream_t)stream;
    // 16-bit types with 16-byte aligned pixel rows take the LDS-DMA ring kernel
    const bool ring = wgrad_ring_ok(d, es);
    if (fold_rows > 1) {
        LH_REQUIRE(ring && d->ntaps == 1 && n_in % fold_rows == 0 && (n_in / fold_rows) % epc == 0,
                   "lh_wgrad_rowfold: needs the LDS-DMA kernel, one tap and a run of whole 16-byte chunks per row");
        a.fold_k = n_in / fold_rows;
    }
#define LH_WR(T)                                                                  \
    if (bo == 256 && bi == 256) return launch_wgrad_ring<T, 256, 256, 2, 4>(a, s); \
    if (bo == 128 && bi == 128) return launch_wgrad_ring<T, 128, 128, 2, 2>(a, s); \
    if (bo == 128 && bi == 64) return launch_wgrad_ring<T, 128, 64, 4, 1>(a, s);   \
    if (bo == 64 && bi == 128) return launch_wgrad_ring<T, 64, 128, 1, 4>(a, s);   \
    return launch_wgrad_ring<T, 64, 64, 2, 2>(a, s);
    if (ring && dtype == LH_BF16) { LH_WR(bf16) }
    if (ring && dtype == LH_F16) { LH_WR(f16) }
#undef LH_WR
#define LH_WT(T)                                                             \
    if (bo == 128 && bi == 128) return launch_wgrad<T, 128, 128, 2, 2>(a, s); \
    if (bo == 128 && bi == 64) return launch_wgrad<T, 128, 64, 4, 1>(a, s);   \
    if (bo == 64 && bi == 128) return launch_wgrad<T, 64, 128, 1, 4>(a, s);   \
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

extern "C" int lh_wgrad_reduce(const lh_igemm_desc* d, const float* slab, float* grad, int n_out,
                               int n_in, long so, long si, long sr, long ss, const int* taps_rs,
                               int accumulate, int dtype, void* stream) {
    LH_REQUIRE(d && slab && grad && taps_rs, "lh_wgrad_reduce: null pointer");
    LH_REQUIRE(d->ntaps > 0 && d->ntaps <= 64, "lh_wgrad_reduce: ntaps %d out of range", d->ntaps);
    WreduceArgs a;
    int bo, bi, sps;
    wgrad_plan(d, n_out, n_in, dtype, &bo, &bi, &a.nsplit, &sps);
    a.slab = slab; a.grad = grad; a.n_out = n_out; a.n_in = n_in; a.ntaps = d->ntaps;
    a.accumulate = accumulate; a.so = so; a.si = si; a.sr = sr; a.ss = ss;
    for (int t = 0; t < 64; ++t) {
        a.r[t] = t < d->ntaps ? (signed char)taps_rs[2 * t] : 0;
        a.s[t] = t < d->ntaps ? (signed char)taps_rs[2 * t + 1] : 0;
    }
    const long per = (long)n_out * n_in;
    bool contig = so == (long)n_in * d->ntaps && si == d->ntaps && ss == 1;
    for (int t = 0; t < d->ntaps && contig; ++t) contig = (a.r[t] * sr + a.s[t] * ss) == t;
    if (contig && per * d->ntaps > (2L << 20)) {
        hipLaunchKernelGGL(wgrad_reduce_contig_kernel, dim3(ceil_div(per, 256)), dim3(256), 256 * (d->ntaps + 1) * sizeof(float),
                           (hipStream_t)stream, a);
        LH_LAUNCH_CHECK("wgrad_reduce launch");
        return LH_OK;
    }
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(ceil_div(per * d->ntaps, 256)), dim3(256), 0, (hipStream_t)stream, a);
    LH_LAUNCH_CHECK("wgrad_reduce launch");
    return LH_OK;
}
