// Layout transforms, weight packs, heatmap target / loss / arg-max decode, fused Adam.
#include "common.h"
#ifndef LH_NT_ADAM
#define LH_NT_ADAM 0         // debug builds only: the optimizer reads the gradient arena (its last use) with non-temporal loads
#endif
#include <stdarg.h>
#include <stdio.h>

// ------------------------------------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";
void lh_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* lh_last_error(void) { return g_err; }
extern "C" int lh_version(void) { return 100; }
extern "C" int lh_dtype_size(int dtype) {
    switch (dtype) {
        case LH_F32: return 4;
        case LH_BF16: return 2;
        case LH_F16: return 2;
        default: return 0;
    }
}

// ------------------------------------------------------------------------------------------------ transforms
template <typename T>
__global__ void image_to_nhwc4_kernel(const float* src, T* dst, int n, int h, int w, int pad, int hp, int wp) {
    // 32-bit index arithmetic (the launcher checks n * hp * wp < 2^31: the 64-bit divisions were most of this kernel's instructions)
    // and one store per pixel
    const unsigned total = (unsigned)n * hp * wp;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const unsigned t = i / (unsigned)wp;
        const int x = (int)(i - t * (unsigned)wp);
        const int b = (int)(t / (unsigned)hp), y = (int)(t - (unsigned)b * (unsigned)hp);
        const int sy = y - pad, sx = x - pad;
        float v[3] = {0.f, 0.f, 0.f};
        if ((unsigned)sy < (unsigned)h && (unsigned)sx < (unsigned)w) {
            const long base = ((long)b * 3 * h + sy) * w + sx;
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = src[base + (long)c * h * w];
        }
        if constexpr (sizeof(T) == 2) {
            union { uint2 u; T e[4]; } pk;
            pk.e[0] = from_f<T>(v[0]); pk.e[1] = from_f<T>(v[1]); pk.e[2] = from_f<T>(v[2]); pk.e[3] = from_f<T>(0.f);
            *reinterpret_cast<uint2*>(dst + (long)i * 4) = pk.u;
        } else {
            T* o = dst + (long)i * 4;
            o[0] = from_f<T>(v[0]); o[1] = from_f<T>(v[1]); o[2] = from_f<T>(v[2]); o[3] = from_f<T>(0.f);
        }
    }
}

extern "C" int lh_image_to_nhwc4(const float* nchw, void* out, int n, int h, int w, int pad, int wp, int dtype,
                                 void* stream) {
    LH_REQUIRE(nchw && out && n > 0 && h > 0 && w > 0 && pad >= 0 && wp >= w + 2 * pad, "lh_image_to_nhwc4: bad arguments");
    const int hp = h + 2 * pad;
    const long total = (long)n * hp * wp;
    LH_REQUIRE(total < (1L << 31), "lh_image_to_nhwc4: image batch too large for 32-bit pixel indices");
    const int grid = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    LH_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((image_to_nhwc4_kernel<T>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                                                   nchw, (T*)out, n, h, w, pad, hp, wp));
    LH_LAUNCH_CHECK("image_to_nhwc4 launch");
    return LH_OK;
}

// Fused input pipeline: uint8 HWC image -> ToTensor (/255) -> bilinear Resize(h, w)
// (half-pixel centres, no antialias: torchvision's tensor Resize when upsampling 224 -> 256) -> Normalize(mean, std)
// -> zero-padded NHWC4 in the run dtype.  Reference CPU path: src/tools/dataset.py:128-159.
struct U8Args {
    const unsigned char* src;
    void* dst;
    int n, hs, ws, h, w, pad, hp, wp;
    float mean[3], istd[3];
};

template <typename T>
__global__ void image_u8_to_nhwc4_kernel(const U8Args p) {
    const long total = (long)p.n * p.hp * p.wp;
    const float sy = (float)p.hs / p.h, sx = (float)p.ws / p.w;
    T* dst = (T*)p.dst;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % p.wp);
        const long t = i / p.wp;
        const int y = (int)(t % p.hp), b = (int)(t / p.hp);
        const int oy = y - p.pad, ox = x - p.pad;
        float v[3] = {0.f, 0.f, 0.f};
        if ((unsigned)oy < (unsigned)p.h && (unsigned)ox < (unsigned)p.w) {
            float fy = (oy + 0.5f) * sy - 0.5f, fx = (ox + 0.5f) * sx - 0.5f;
            fy = fy < 0.f ? 0.f : fy;
            fx = fx < 0.f ? 0.f : fx;
            const int y0 = (int)fy, x0 = (int)fx;
            const int y1 = y0 + 1 < p.hs ? y0 + 1 : p.hs - 1, x1 = x0 + 1 < p.ws ? x0 + 1 : p.ws - 1;
            const float wy = fy - y0, wx = fx - x0;
            const unsigned char* base = p.src + (long)b * p.hs * p.ws * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float a00 = base[((long)y0 * p.ws + x0) * 3 + c], a01 = base[((long)y0 * p.ws + x1) * 3 + c];
                const float a10 = base[((long)y1 * p.ws + x0) * 3 + c], a11 = base[((long)y1 * p.ws + x1) * 3 + c];
                const float top = a00 + (a01 - a00) * wx, bot = a10 + (a11 - a10) * wx;
                const float pix = (top + (bot - top) * wy) * (1.f / 255.f);
                v[c] = (pix - p.mean[c]) * p.istd[c];
            }
        }
        T* o = dst + i * 4;
        o[0] = from_f<T>(v[0]); o[1] = from_f<T>(v[1]); o[2] = from_f<T>(v[2]); o[3] = from_f<T>(0.f);
    }
}

extern "C" int lh_image_u8_to_nhwc4(const unsigned char* hwc, void* out, int n, int hs, int ws, int h, int w, int pad, int wp,
                                    const float* mean3, const float* std3, int dtype, void* stream) {
    LH_REQUIRE(hwc && out && mean3 && std3 && n > 0 && hs > 0 && ws > 0 && h > 0 && w > 0 && pad >= 0 && wp >= w + 2 * pad,
               "lh_image_u8_to_nhwc4: bad arguments");
    U8Args a;
    a.src = hwc; a.dst = out; a.n = n; a.hs = hs; a.ws = ws; a.h = h; a.w = w; a.pad = pad; a.hp = h + 2 * pad; a.wp = wp;
    for (int c = 0; c < 3; ++c) { a.mean[c] = mean3[c]; a.istd[c] = 1.f / std3[c]; }
    const long total = (long)n * a.hp * wp;
    const int grid = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    LH_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((image_u8_to_nhwc4_kernel<T>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a));
    LH_LAUNCH_CHECK("image_u8_to_nhwc4 launch");
    return LH_OK;
}

// ---- the same pipeline with torchvision's ColorJitter between Resize and Normalize (src/tools/dataset.py:134-146).
// The random draw stays on the host (ColorJitter.get_params): per image four factors (brightness, contrast,
// saturation, hue) and the op order (four op ids 0..3, negative = skip) arrive as device arrays.  Contrast blends
// with the mean grey level of the WHOLE image as it is when the op runs, so a first kernel reduces that mean (of the
// image after the ops that precede contrast) into fp64 strip sums, and the second kernel applies everything.
__device__ __forceinline__ float cj_clamp01(float v) { return fminf(fmaxf(v, 0.f), 1.f); }
__device__ __forceinline__ float cj_gray(const float* c) { return 0.2989f * c[0] + 0.587f * c[1] + 0.114f * c[2]; }
__device__ __forceinline__ void cj_blend(float* c, float o0, float o1, float o2, float r) {
    c[0] = cj_clamp01(r * c[0] + (1.f - r) * o0);
    c[1] = cj_clamp01(r * c[1] + (1.f - r) * o1);
    c[2] = cj_clamp01(r * c[2] + (1.f - r) * o2);
}
__device__ __forceinline__ void cj_hue(float* c, float f) {
    const float r = c[0], g = c[1], b = c[2];
    const float maxc = fmaxf(r, fmaxf(g, b)), minc = fminf(r, fminf(g, b));
    const bool eq = maxc == minc;
    const float cr = maxc - minc;
    const float s = cr / (eq ? 1.f : maxc);
    const float div = eq ? 1.f : cr;
    const float rc = (maxc - r) / div, gc = (maxc - g) / div, bc = (maxc - b) / div;
    float h = 0.f;
    if (maxc == r) h = bc - gc;
    else if (maxc == g) h = 2.f + rc - bc;
    else h = 4.f + gc - rc;
    h = fmodf(h / 6.f + 1.f, 1.f);
    h = fmodf(h + f, 1.f);
    if (h < 0.f) h += 1.f;
    const float h6 = h * 6.f;
    const float fl = floorf(h6);
    const float fr = h6 - fl;
    int i = (int)fl % 6;
    if (i < 0) i += 6;
    const float v = maxc;
    const float p = cj_clamp01(v * (1.f - s)), q = cj_clamp01(v * (1.f - s * fr)), t = cj_clamp01(v * (1.f - s * (1.f - fr)));
    switch (i) {
        case 0: c[0] = v; c[1] = t; c[2] = p; break;
        case 1: c[0] = q; c[1] = v; c[2] = p; break;
        case 2: c[0] = p; c[1] = v; c[2] = t; break;
        case 3: c[0] = p; c[1] = q; c[2] = v; break;
        case 4: c[0] = t; c[1] = p; c[2] = v; break;
        default: c[0] = v; c[1] = p; c[2] = q; break;
    }
}
// ops order[first .. last) on one pixel; `mean` = the image's grey mean for the contrast op
__device__ __forceinline__ void cj_apply(float* c, const float* f, const int* order, int first, int last, float mean) {
    for (int k = first; k < last; ++k) {
        const int op = order[k];
        if (op == 0) cj_blend(c, 0.f, 0.f, 0.f, f[0]);
        else if (op == 1) cj_blend(c, mean, mean, mean, f[1]);
        else if (op == 2) { const float g = cj_gray(c); cj_blend(c, g, g, g, f[2]); }
        else if (op == 3) cj_hue(c, f[3]);
    }
}
__device__ __forceinline__ void u8_bilinear(const U8Args& p, int b, int oy, int ox, float* c) {
    const float sy = (float)p.hs / p.h, sx = (float)p.ws / p.w;
    float fy = (oy + 0.5f) * sy - 0.5f, fx = (ox + 0.5f) * sx - 0.5f;
    fy = fy < 0.f ? 0.f : fy;
    fx = fx < 0.f ? 0.f : fx;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + 1 < p.hs ? y0 + 1 : p.hs - 1, x1 = x0 + 1 < p.ws ? x0 + 1 : p.ws - 1;
    const float wy = fy - y0, wx = fx - x0;
    const unsigned char* base = p.src + (long)b * p.hs * p.ws * 3;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        const float a00 = base[((long)y0 * p.ws + x0) * 3 + ch], a01 = base[((long)y0 * p.ws + x1) * 3 + ch];
        const float a10 = base[((long)y1 * p.ws + x0) * 3 + ch], a11 = base[((long)y1 * p.ws + x1) * 3 + ch];
        const float top = a00 + (a01 - a00) * wx, bot = a10 + (a11 - a10) * wx;
        c[ch] = (top + (bot - top) * wy) * (1.f / 255.f);
    }
}

constexpr int CJ_STRIPS = 32;

__global__ __launch_bounds__(256) void jitter_mean_kernel(const U8Args p, const float* factors, const int* order, double* partial) {
    __shared__ double red[256];
    const int b = blockIdx.y, strip = blockIdx.x;
    const float* f = factors + b * 4;
    const int* ord = order + b * 4;
    int kc = 4;                                         // position of the contrast op (4 = absent)
    for (int k = 3; k >= 0; --k)
        if (ord[k] == 1) kc = k;
    double acc = 0.0;
    const int rows = (p.h + CJ_STRIPS - 1) / CJ_STRIPS;
    const int y0 = strip * rows, y1 = min(p.h, y0 + rows);
    if (kc < 4)
        for (int i = threadIdx.x; i < (y1 - y0) * p.w; i += 256) {
            float c[3];
            u8_bilinear(p, b, y0 + i / p.w, i % p.w, c);
            cj_apply(c, f, ord, 0, kc, 0.f);
            acc += (double)cj_gray(c);
        }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[b * CJ_STRIPS + strip] = red[0];
}

template <typename T>
__global__ void image_u8_jitter_to_nhwc4_kernel(const U8Args p, const float* factors, const int* order, const double* partial) {
    const long total = (long)p.n * p.hp * p.wp;
    T* dst = (T*)p.dst;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % p.wp);
        const long t = i / p.wp;
        const int y = (int)(t % p.hp), b = (int)(t / p.hp);
        const int oy = y - p.pad, ox = x - p.pad;
        float v[3] = {0.f, 0.f, 0.f};
        if ((unsigned)oy < (unsigned)p.h && (unsigned)ox < (unsigned)p.w) {
            double m = 0.0;
            for (int k = 0; k < CJ_STRIPS; ++k) m += partial[b * CJ_STRIPS + k];
            const float mean = (float)(m / ((double)p.h * p.w));
            float c[3];
            u8_bilinear(p, b, oy, ox, c);
            cj_apply(c, factors + b * 4, order + b * 4, 0, 4, mean);
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) v[ch] = (c[ch] - p.mean[ch]) * p.istd[ch];
        }
        T* o = dst + i * 4;
        o[0] = from_f<T>(v[0]); o[1] = from_f<T>(v[1]); o[2] = from_f<T>(v[2]); o[3] = from_f<T>(0.f);
    }
}

extern "C" size_t lh_image_jitter_workspace_bytes(int n) { return (size_t)n * CJ_STRIPS * sizeof(double); }

extern "C" int lh_image_u8_jitter_to_nhwc4(const unsigned char* hwc, void* out, int n, int hs, int ws, int h, int w, int pad, int wp,
                                           const float* mean3, const float* std3, const float* factors_dev, const int* order_dev,
                                           void* workspace, int dtype, void* stream) {
    LH_REQUIRE(hwc && out && mean3 && std3 && factors_dev && order_dev && workspace && n > 0 && hs > 0 && ws > 0 && h > 0 && w > 0 &&
               pad >= 0 && wp >= w + 2 * pad, "lh_image_u8_jitter_to_nhwc4: bad arguments");
    U8Args a;
    a.src = hwc; a.dst = out; a.n = n; a.hs = hs; a.ws = ws; a.h = h; a.w = w; a.pad = pad; a.hp = h + 2 * pad; a.wp = wp;
    for (int c = 0; c < 3; ++c) { a.mean[c] = mean3[c]; a.istd[c] = 1.f / std3[c]; }
    hipLaunchKernelGGL(jitter_mean_kernel, dim3(CJ_STRIPS, n), dim3(256), 0, (hipStream_t)stream, a, factors_dev, order_dev, (double*)workspace);
    const long total = (long)n * a.hp * wp;
    const int grid = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    LH_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((image_u8_jitter_to_nhwc4_kernel<T>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a,
                                                   factors_dev, order_dev, (const double*)workspace));
    LH_LAUNCH_CHECK("image_u8_jitter_to_nhwc4 launch");
    return LH_OK;
}

template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* src, float* dst, int n, int hw, int c, int cs, int vec) {
    const long total = (long)n * hw;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int b = (int)(i / hw), p = (int)(i % hw);
        const T* s = src + i * cs;
        constexpr int EPC = 16 / sizeof(T);
        if (vec) {                               // whole, 16-byte ALIGNED chunks per pixel: one vector load per EPC channels (was one 2-byte load per channel)
            for (int c0 = 0; c0 < c; c0 += EPC) {
                float v[EPC];
                unpack16<T>(*reinterpret_cast<const uint4*>(s + c0), v);
#pragma unroll
                for (int e = 0; e < EPC; ++e)
                    if (c0 + e < c) dst[((long)b * c + c0 + e) * hw + p] = v[e];
            }
        } else {
            for (int ch = 0; ch < c; ++ch) dst[((long)b * c + ch) * hw + p] = to_f<T>(s[ch]);
        }
    }
}
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* src, T* dst, int n, int hw, int c, int cs, int vec) {
    const long total = (long)n * hw;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int b = (int)(i / hw), p = (int)(i % hw);
        T* d = dst + i * cs;
        constexpr int EPC = 16 / sizeof(T);
        if (vec) {                               // whole, 16-byte aligned chunks per pixel: gather EPC channels, one vector store
            for (int c0 = 0; c0 < cs; c0 += EPC) {
                float v[EPC];
#pragma unroll
                for (int e = 0; e < EPC; ++e) v[e] = c0 + e < c ? src[((long)b * c + c0 + e) * hw + p] : 0.f;
                *reinterpret_cast<uint4*>(d + c0) = pack16<T>(v);
            }
        } else {
            for (int ch = 0; ch < cs; ++ch) d[ch] = from_f<T>(ch < c ? src[((long)b * c + ch) * hw + p] : 0.f);
        }
    }
}

extern "C" int lh_nhwc_to_nchw_f32(const void* nhwc, float* nchw, int n, int h, int w, int c, int c_stride, int dtype,
                                   void* stream) {
    LH_REQUIRE(nhwc && nchw && n > 0 && h > 0 && w > 0 && c > 0 && c_stride >= c, "lh_nhwc_to_nchw_f32: bad arguments");
    const long total = (long)n * h * w;
    const int grid = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    // the vector path needs 16-byte aligned pixel rows: a channel-sliced base pointer (base + 4 channels, stride 64) takes the scalar loop
    const int es = lh_dtype_size(dtype);
    const int vec = es > 0 && c_stride % (16 / es) == 0 && ((uintptr_t)nhwc & 15) == 0;
    LH_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((nhwc_to_nchw_kernel<T>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                                                   (const T*)nhwc, nchw, n, h * w, c, c_stride, vec));
    LH_LAUNCH_CHECK("nhwc_to_nchw launch");
    return LH_OK;
}
extern "C" int lh_nchw_f32_to_nhwc(const float* nchw, void* nhwc, int n, int h, int w, int c, int c_stride, int dtype,
                                   void* stream) {
    LH_REQUIRE(nhwc && nchw && n > 0 && h > 0 && w > 0 && c > 0 && c_stride >= c, "lh_nchw_f32_to_nhwc: bad arguments");
    const long total = (long)n * h * w;
    const int grid = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    const int es = lh_dtype_size(dtype);
    const int vec = es > 0 && c_stride % (16 / es) == 0 && ((uintptr_t)nhwc & 15) == 0;
    LH_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((nchw_to_nhwc_kernel<T>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                                                   nchw, (T*)nhwc, n, h * w, c, c_stride, vec));
    LH_LAUNCH_CHECK("nchw_to_nhwc launch");
    return LH_OK;
}

// ------------------------------------------------------------------------------------------------ weight pack
struct PackArgs {
    const float* w;
    void* out;
    int n_out, n_in, ntaps, kpad, rows;
    long so, si, sr, ss;
    signed char r[64];
    signed char s[64];
};

template <typename T>
__global__ void pack_weight_kernel(const PackArgs p) {
    const long total = (long)p.rows * p.ntaps * p.kpad;
    T* out = (T*)p.out;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k = (int)(i % p.kpad);
        const long t2 = i / p.kpad;
        const int t = (int)(t2 % p.ntaps), o = (int)(t2 / p.ntaps);
        float v = 0.f;
        if (o < p.n_out && k < p.n_in) v = p.w[o * p.so + k * p.si + p.r[t] * p.sr + p.s[t] * p.ss];
        out[i] = from_f<T>(v);
    }
}

extern "C" int lh_pack_weight(const float* w, void* out, size_t* bytes, int n_out, int n_in, long so, long si,
                              long sr, long ss, int ntaps, const int* taps_rs, int dtype, void* stream) {
    const int es = lh_dtype_size(dtype);
    LH_REQUIRE(es > 0, "lh_pack_weight: bad dtype %d", dtype);
    LH_REQUIRE(n_out > 0 && n_in > 0 && ntaps >= 0 && ntaps <= 64, "lh_pack_weight: bad sizes");
    const int kstep = 128 / es;          // K is padded to the 128-byte step of the ring kernel
    const int kpad = (n_in + kstep - 1) / kstep * kstep;
    const int rows = (n_out + 127) / 128 * 128;
    const size_t need = (size_t)rows * (ntaps > 0 ? ntaps : 1) * kpad * es;
    if (bytes) *bytes = need;
    if (!out) return LH_OK;
    if (ntaps == 0) return LH_OK;
    LH_REQUIRE(w && taps_rs, "lh_pack_weight: null pointer");
    PackArgs a;
    a.w = w; a.out = out; a.n_out = n_out; a.n_in = n_in; a.ntaps = ntaps; a.kpad = kpad; a.rows = rows;
    a.so = so; a.si = si; a.sr = sr; a.ss = ss;
    for (int t = 0; t < 64; ++t) {
        a.r[t] = t < ntaps ? (signed char)taps_rs[2 * t] : 0;
        a.s[t] = t < ntaps ? (signed char)taps_rs[2 * t + 1] : 0;
    }
    const long total = (long)rows * ntaps * kpad;
    const int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    LH_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((pack_weight_kernel<T>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a));
    LH_LAUNCH_CHECK("pack_weight launch");
    return LH_OK;
}

// All packs of a model in ONE launch: the host cuts every pack into chunks of PACK_CHUNK output elements and
// blockIdx.x walks the chunk table (device arrays), so big and small packs are balanced over the grid.
constexpr int PACK_CHUNK = 2048;         // elements per workgroup: 8 dependent gathers per thread (the stem pack is 57k elements: 28 workgroups, not 2)

template <typename T>
__global__ __launch_bounds__(256) void pack_weight_multi_kernel(const lh_pack_item* items, const int* chunk_item,
                                                                const long* chunk_start) {
    const lh_pack_item& p = items[chunk_item[blockIdx.x]];
    const int es = sizeof(T);
    const int kstep = 128 / es;
    const int kpad = (p.n_in + kstep - 1) / kstep * kstep;
    const int rows = (p.n_out + 127) / 128 * 128;
    const long total = (long)rows * p.ntaps * kpad;
    const long begin = chunk_start[blockIdx.x];
    long end = begin + PACK_CHUNK;
    if (end > total) end = total;
    T* out = (T*)p.out;
    for (long i = begin + threadIdx.x; i < end; i += 256) {
        const int k = (int)(i % kpad);
        const long t2 = i / kpad;
        const int t = (int)(t2 % p.ntaps), o = (int)(t2 / p.ntaps);
        float v = 0.f;
        if (o < p.n_out && k < p.n_in) v = p.w[o * p.so + k * p.si + p.r[t] * p.sr + p.s[t] * p.ss];
        out[i] = from_f<T>(v);
    }
}

extern "C" int lh_pack_chunk_elems(void) { return PACK_CHUNK; }

extern "C" int lh_pack_weights_multi(const lh_pack_item* items_dev, const int* chunk_item_dev, const long* chunk_start_dev,
                                     int n_chunks, int dtype, void* stream) {
    LH_REQUIRE(items_dev && chunk_item_dev && chunk_start_dev && n_chunks > 0, "lh_pack_weights_multi: bad arguments");
    LH_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((pack_weight_multi_kernel<T>), dim3(n_chunks), dim3(256), 0, (hipStream_t)stream,
                                                   items_dev, chunk_item_dev, chunk_start_dev));
    LH_LAUNCH_CHECK("pack_weights_multi launch");
    return LH_OK;
}

// Transposing pack for regular weight tensors w[d0][d1][rs] (Conv2d: d0 = C_out, d1 = C_in; ConvTranspose2d:
// d0 = C_in, d1 = C_out): one workgroup reads a 32 x 32 x rs tile with fully coalesced loads (the strided
// per-element gather of pack_weight_multi_kernel over-fetches ~16x, profiles/r01_pmc_hbm_traffic.txt), keeps it in
// LDS and writes every pack that needs it -- "row = d0" packs [d0][tap][d1] and "row = d1" packs [d1][tap][d0] --
// in 64-byte runs.  Pack padding (rows >= n, K >= n_in) is zeroed once at allocation and never written.
template <typename T>
__global__ __launch_bounds__(256) void pack_tiled_kernel(const lh_pack_conv* convs, const int* chunk_conv, const int* chunk_t0,
                                                         const int* chunk_t1) {
    extern __shared__ __attribute__((aligned(16))) unsigned char psm[];
    T* tile = reinterpret_cast<T*>(psm);                         // [32 d0][32 d1][rs] (+4 pad per d0 row)
    const lh_pack_conv& c = convs[chunk_conv[blockIdx.x]];
    const int t0 = chunk_t0[blockIdx.x] * 32, t1 = chunk_t1[blockIdx.x] * 32;
    const int rs = c.rs;
    const int rowlen = 32 * rs;                                  // contiguous floats per d0 row of the tile
    const int ld = rowlen + 4;                                   // LDS row stride in elements (8-byte aligned rows)
    const bool full = t0 + 32 <= c.d0 && t1 + 32 <= c.d1 && ((long)c.d1 * rs) % 4 == 0 && sizeof(T) == 2;
    if (full) {                                                  // interior tile: 16-byte loads, 8-byte LDS stores
        const int vpr = rowlen / 4;                              // float4 per row
        for (int i = threadIdx.x; i < 32 * vpr; i += 256) {
            const int a = i / vpr, v4 = i - a * vpr;
            const float4 v = *reinterpret_cast<const float4*>(c.w + ((long)(t0 + a) * c.d1 + t1) * rs + v4 * 4);
            union { uint2 u; T e[4]; } pk;
            pk.e[0] = from_f<T>(v.x); pk.e[1] = from_f<T>(v.y); pk.e[2] = from_f<T>(v.z); pk.e[3] = from_f<T>(v.w);
            *reinterpret_cast<uint2*>(tile + a * ld + v4 * 4) = pk.u;
        }
    } else {
        for (int i = threadIdx.x; i < 32 * rowlen; i += 256) {
            const int a = i / rowlen, rem = i - a * rowlen;      // a = d0 offset, rem = d1_off * rs + tap
            const int d0 = t0 + a, d1 = t1 + rem / rs;
            float v = 0.f;
            if (d0 < c.d0 && d1 < c.d1) v = c.w[((long)d0 * c.d1 + t1) * rs + rem];
            tile[a * ld + rem] = from_f<T>(v);
        }
    }
    __syncthreads();
    for (int p = 0; p < c.npacks; ++p) {
        const lh_pack_out& o = c.packs[p];
        T* out = reinterpret_cast<T*>(o.out);
        const int nrow = o.row_is_d1 ? c.d1 : c.d0, nk = o.row_is_d1 ? c.d0 : c.d1;
        const int r0 = o.row_is_d1 ? t1 : t0, k0 = o.row_is_d1 ? t0 : t1;
        if (full && (o.kpad & 3) == 0) {                         // four K values per thread: one 8-byte store
            const int total = o.ntaps * 32 * 8;
            for (int i = threadIdx.x; i < total; i += 256) {
                const int k = (i & 7) * 4, rest = i >> 3;
                const int t = rest % o.ntaps, row = rest / o.ntaps;
                const int tap = o.taps[t];
                union { uint2 u; T e[4]; } pk;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    pk.e[e] = o.row_is_d1 ? tile[(k + e) * ld + row * rs + tap] : tile[row * ld + (k + e) * rs + tap];
                *reinterpret_cast<uint2*>(out + ((long)(r0 + row) * o.ntaps + t) * o.kpad + k0 + k) = pk.u;
            }
            continue;
        }
        const int total = o.ntaps * 32 * 32;
        for (int i = threadIdx.x; i < total; i += 256) {
            const int k = i & 31, rest = i >> 5;                 // k runs along the pack's K (fastest in memory)
            const int t = rest % o.ntaps, row = rest / o.ntaps;
            const int tap = o.taps[t];
            int a, b;                                            // a = d0 offset, b = d1 offset inside the tile
            if (o.row_is_d1) { b = row; a = k; } else { a = row; b = k; }
            const int grow = r0 + row, gk = k0 + k;
            if (grow < nrow && gk < nk) out[((long)grow * o.ntaps + t) * o.kpad + gk] = tile[a * ld + b * rs + tap];
        }
    }
}

extern "C" int lh_pack_weights_tiled(const lh_pack_conv* convs_dev, const int* chunk_conv_dev, const int* chunk_t0_dev,
                                     const int* chunk_t1_dev, int n_chunks, int max_rs, int dtype, void* stream) {
    LH_REQUIRE(convs_dev && chunk_conv_dev && chunk_t0_dev && chunk_t1_dev && n_chunks > 0 && max_rs > 0 && max_rs <= 49,
               "lh_pack_weights_tiled: bad arguments");
    const int es = lh_dtype_size(dtype);
    const size_t lds = (size_t)32 * (32 * max_rs + 4) * es;
    LH_REQUIRE(lds <= 64 * 1024, "lh_pack_weights_tiled: tile of %d taps does not fit LDS for this dtype", max_rs);
    LH_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((pack_tiled_kernel<T>), dim3(n_chunks), dim3(256), lds, (hipStream_t)stream,
                                                   convs_dev, chunk_conv_dev, chunk_t0_dev, chunk_t1_dev));
    LH_LAUNCH_CHECK("pack_weights_tiled launch");
    return LH_OK;
}

// ------------------------------------------------------------------------------------------------ Gaussian target
__global__ void gaussian_target_kernel(const float* joints, int jstride, const float* patch, int radius, float* target,
                                       int bj, int size) {
    const long total = (long)bj * size * size;
    const int pw = 2 * radius + 1;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % size);
        const long t = i / size;
        const int y = (int)(t % size);
        const long j = t / size;
        const float jx = joints[j * jstride], jy = joints[j * jstride + 1];
        // int(v / 4 + 0.5): truncation toward zero, like Python's int()
        const int mx = (int)(jx * 0.25f + 0.5f), my = (int)(jy * 0.25f + 0.5f);
        const int x0 = mx - radius, y0 = my - radius, x1 = mx + radius + 1, y1 = my + radius + 1;
        float v = 0.f;
        const bool skip = x0 >= size || y0 >= size || x1 < 0 || y1 < 0;
        if (!skip && x >= x0 && x < x1 && y >= y0 && y < y1) v = patch[(y - y0) * pw + (x - x0)];
        target[i] = v;
    }
}

// GenerateHeatmap (src/utils/dataset_loader.py:22-53), the alternate renderer: points are ALREADY in heat-map coordinates,
// sigma = res / 64 (an integer here), patch of (6 * sigma + 3)^2 centred on int(point), np.maximum blend with the zero
// map (= plain placement: every joint owns its plane); a joint is skipped when x <= 0 or int(point) lies outside the map.
__global__ void gaussian_target_alt_kernel(const float* points, int pstride, const float* patch, int sigma, float* target,
                                           int bj, int res) {
    const long total = (long)bj * res * res;
    const int pw = 6 * sigma + 3;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % res);
        const long t = i / res;
        const int y = (int)(t % res);
        const long j = t / res;
        const float fx = points[j * pstride], fy = points[j * pstride + 1];
        float v = 0.f;
        if (fx > 0.f) {
            const int px = (int)fx, py = (int)fy;                    // Python int(): truncation toward zero
            if (px >= 0 && py >= 0 && px < res && py < res) {
                const int ulx = px - 3 * sigma - 1, uly = py - 3 * sigma - 1;
                const int gx = x - ulx, gy = y - uly;                // hms[aa:bb, cc:dd] <- g[a:b, c:d]: same offset on both axes
                if (gx >= 0 && gx < pw && gy >= 0 && gy < pw) v = patch[gy * pw + gx];
            }
        }
        target[i] = v;
    }
}

extern "C" int lh_gaussian_target_alt(const float* points, int pstride, const float* patch, int sigma, float* target,
                                      int b, int j, int res, void* stream) {
    LH_REQUIRE(points && patch && target && pstride >= 2 && b > 0 && j > 0 && res > 0 && sigma >= 1 && res == 64 * sigma,
               "lh_gaussian_target_alt: bad arguments (res must be 64 * sigma, sigma a positive integer)");
    const long total = (long)b * j * res * res;
    const int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(gaussian_target_alt_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, points, pstride, patch, sigma,
                       target, b * j, res);
    LH_LAUNCH_CHECK("gaussian_target_alt launch");
    return LH_OK;
}

extern "C" int lh_gaussian_target(const float* joints, int jstride, const float* patch, int radius, float* target,
                                  int b, int j, int size, void* stream) {
    LH_REQUIRE(joints && patch && target && jstride >= 2 && b > 0 && j > 0 && size > 0 && radius >= 0,
               "lh_gaussian_target: bad arguments");
    const long total = (long)b * j * size * size;
    const int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(gaussian_target_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, joints, jstride, patch, radius,
                       target, b * j, size);
    LH_LAUNCH_CHECK("gaussian_target launch");
    return LH_OK;
}

// ------------------------------------------------------------------------------------------------ MSE loss
constexpr int MSE_BLOCKS = 512;

__global__ __launch_bounds__(256) void mse_partial_kernel(const float* pred, const float* target, long numel, float* grad,
                                                         const float* grad_scale, double* partial) {
    __shared__ double red[4];
    const float gs = (grad_scale ? *grad_scale : 1.f) / (float)numel;
    double acc = 0.0;
    const long nvec = numel / 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long)gridDim.x * 256) {
        const float4 p = reinterpret_cast<const float4*>(pred)[i];
        const float4 t = reinterpret_cast<const float4*>(target)[i];
        const float4 d = {p.x - t.x, p.y - t.y, p.z - t.z, p.w - t.w};
        acc += (double)(d.x * d.x) + (double)(d.y * d.y) + (double)(d.z * d.z) + (double)(d.w * d.w);
        if (grad) reinterpret_cast<float4*>(grad)[i] = float4{d.x * gs, d.y * gs, d.z * gs, d.w * gs};
    }
    if (blockIdx.x == 0)
        for (long i = nvec * 4 + threadIdx.x; i < numel; i += 256) {
            const float d = pred[i] - target[i];
            acc += (double)(d * d);
            if (grad) grad[i] = d * gs;
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ void mse_final_kernel(const double* partial, int nblocks, long numel, float* loss) {
    __shared__ double red[4];
    double acc = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) acc += partial[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) *loss = (float)(0.5 * (red[0] + red[1] + red[2] + red[3]) / (double)numel);
}

extern "C" size_t lh_mse_workspace_bytes(long numel) { (void)numel; return MSE_BLOCKS * sizeof(double); }

extern "C" int lh_mse_heatmap(const float* pred, const float* target, long numel, float* loss, float* grad,
                              const float* grad_scale, void* workspace, void* stream) {
    LH_REQUIRE(pred && target && loss && workspace && numel > 0, "lh_mse_heatmap: bad arguments");
    LH_REQUIRE(((uintptr_t)pred % 16 == 0) && ((uintptr_t)target % 16 == 0) && (!grad || (uintptr_t)grad % 16 == 0),
               "lh_mse_heatmap: buffers must be 16-byte aligned");
    int blocks = (int)((numel / 4 + 255) / 256);
    if (blocks > MSE_BLOCKS) blocks = MSE_BLOCKS;
    if (blocks < 1) blocks = 1;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(mse_partial_kernel, dim3(blocks), dim3(256), 0, s, pred, target, numel, grad, grad_scale,
                       (double*)workspace);
    hipLaunchKernelGGL(mse_final_kernel, dim3(1), dim3(256), 0, s, (const double*)workspace, blocks, numel, loss);
    LH_LAUNCH_CHECK("mse launch");
    return LH_OK;
}

// ------------------------------------------------------------------------------------------------ arg-max decode
struct Cand { float v; int i; };
// true when a precedes b under numpy.argmax's rule: NaN beats everything, then larger value,
// ties (and NaN vs NaN) broken by the lower flat index.
__device__ __forceinline__ bool cand_before(const Cand& a, const Cand& b) {
    const bool an = a.v != a.v, bn = b.v != b.v;
    if (an || bn) return an && (!bn || a.i < b.i);
    return a.v > b.v || (a.v == b.v && a.i < b.i);
}

__global__ __launch_bounds__(256) void heatmap_argmax_kernel(const float* hm, int hw, int w, float scale, float* preds,
                                                            float* maxvals, int* idx) {
    __shared__ Cand red[4];
    const float* m = hm + (long)blockIdx.x * hw;
    Cand best = {0.f, 0x7fffffff};
    bool have = false;
    for (int i = threadIdx.x; i < hw; i += 256) {
        const Cand c = {m[i], i};
        if (!have || cand_before(c, best)) { best = c; have = true; }
    }
    if (!have) best = Cand{-INFINITY, 0x7fffffff};
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        Cand other = {__shfl_xor(best.v, o), __shfl_xor(best.i, o)};
        if (other.i != 0x7fffffff && (best.i == 0x7fffffff || cand_before(other, best))) best = other;
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        Cand b = red[0];
        for (int k = 1; k < 4; ++k)
            if (red[k].i != 0x7fffffff && (b.i == 0x7fffffff || cand_before(red[k], b))) b = red[k];
        const float keep = b.v > 0.f ? 1.f : 0.f;
        preds[blockIdx.x * 2 + 0] = (float)(b.i % w) * keep * scale;
        preds[blockIdx.x * 2 + 1] = (float)(b.i / w) * keep * scale;
        maxvals[blockIdx.x] = b.v;
        if (idx) idx[blockIdx.x] = b.i;
    }
}

extern "C" int lh_heatmap_argmax(const float* heatmaps, int bj, int h, int w, float scale, float* preds, float* maxvals,
                                 int* idx, void* stream) {
    LH_REQUIRE(heatmaps && preds && maxvals && bj > 0 && h > 0 && w > 0, "lh_heatmap_argmax: bad arguments");
    hipLaunchKernelGGL(heatmap_argmax_kernel, dim3(bj), dim3(256), 0, (hipStream_t)stream, heatmaps, h * w, w, scale, preds,
                       maxvals, idx);
    LH_LAUNCH_CHECK("heatmap_argmax launch");
    return LH_OK;
}

// Opt-in quarter-pixel refinement of the hard arg-max (the reference carries the switch
// TEST.POST_PROCESS, src/modeling/simplebaseline/config.py:109, but never uses it): the published SimpleBaseline
// `get_final_preds` rule -- when the peak (px, py) is strictly inside the map (1 < px < W-1, 1 < py < H-1) move it a
// quarter pixel toward the higher neighbour on each axis: coord += 0.25 * sign(hm[..+1] - hm[..-1]).  Works on the
// UNSCALED peak; `scale` is the factor lh_heatmap_argmax already applied to `preds`.
__global__ void heatmap_refine_kernel(const float* hm, const int* idx, const float* maxvals, int bj, int h, int w,
                                      float scale, float* preds) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= bj) return;
    if (!(maxvals[t] > 0.f)) return;                      // get_max_preds zeroed the coordinate: px = py = 0, never interior
    const int k = idx[t];
    const int px = k % w, py = k / w;
    if (!(1 < px && px < w - 1 && 1 < py && py < h - 1)) return;
    const float* m = hm + (long)t * h * w;
    const float dx = m[py * w + px + 1] - m[py * w + px - 1];
    const float dy = m[(py + 1) * w + px] - m[(py - 1) * w + px];
    const float sx = dx > 0.f ? 1.f : (dx < 0.f ? -1.f : 0.f), sy = dy > 0.f ? 1.f : (dy < 0.f ? -1.f : 0.f);
    preds[t * 2 + 0] = ((float)px + 0.25f * sx) * scale;
    preds[t * 2 + 1] = ((float)py + 0.25f * sy) * scale;
}

extern "C" int lh_heatmap_refine(const float* heatmaps, const int* idx, const float* maxvals, int bj, int h, int w,
                                 float scale, float* preds, void* stream) {
    LH_REQUIRE(heatmaps && idx && maxvals && preds && bj > 0 && h > 0 && w > 0, "lh_heatmap_refine: bad arguments");
    hipLaunchKernelGGL(heatmap_refine_kernel, dim3((bj + 255) / 256), dim3(256), 0, (hipStream_t)stream, heatmaps, idx, maxvals,
                       bj, h, w, scale, preds);
    LH_LAUNCH_CHECK("heatmap_refine launch");
    return LH_OK;
}

// Opt-in soft-arg-max decode (named in the project's north star; NOT in the reference, which decodes with the hard arg-max
// of get_max_preds): preds = sum_p softmax(beta * hm)[p] * (x_p, y_p), computed per map with the usual max subtraction,
// fp32 exponentials and fp64 sums.  One workgroup per (sample, joint).
__global__ __launch_bounds__(256) void heatmap_soft_argmax_kernel(const float* hm, int hw, int w, float beta, float scale, float* preds) {
    __shared__ float rmax[4];
    __shared__ double rs[4][3];
    const float* m = hm + (long)blockIdx.x * hw;
    float mx = -INFINITY;
    for (int i = threadIdx.x; i < hw; i += 256) mx = fmaxf(mx, m[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((threadIdx.x & 63) == 0) rmax[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(rmax[0], rmax[1]), fmaxf(rmax[2], rmax[3]));
    double s0 = 0.0, sx = 0.0, sy = 0.0;
    for (int i = threadIdx.x; i < hw; i += 256) {
        const double e = (double)expf(beta * (m[i] - mx));
        s0 += e; sx += e * (double)(i % w); sy += e * (double)(i / w);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s0 += __shfl_xor(s0, o); sx += __shfl_xor(sx, o); sy += __shfl_xor(sy, o); }
    if ((threadIdx.x & 63) == 0) { rs[threadIdx.x >> 6][0] = s0; rs[threadIdx.x >> 6][1] = sx; rs[threadIdx.x >> 6][2] = sy; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double t0 = rs[0][0] + rs[1][0] + rs[2][0] + rs[3][0];
        const double tx = rs[0][1] + rs[1][1] + rs[2][1] + rs[3][1];
        const double ty = rs[0][2] + rs[1][2] + rs[2][2] + rs[3][2];
        preds[blockIdx.x * 2 + 0] = (float)(tx / t0) * scale;
        preds[blockIdx.x * 2 + 1] = (float)(ty / t0) * scale;
    }
}

extern "C" int lh_heatmap_soft_argmax(const float* heatmaps, int bj, int h, int w, float beta, float scale, float* preds,
                                      void* stream) {
    LH_REQUIRE(heatmaps && preds && bj > 0 && h > 0 && w > 0, "lh_heatmap_soft_argmax: bad arguments");
    hipLaunchKernelGGL(heatmap_soft_argmax_kernel, dim3(bj), dim3(256), 0, (hipStream_t)stream, heatmaps, h * w, w, beta, scale, preds);
    LH_LAUNCH_CHECK("heatmap_soft_argmax launch");
    return LH_OK;
}

// ------------------------------------------------------------------------------------------------ validation metrics
// PCK_2d_loss(T, 'proportion') + EPE_train on the device (src/utils/loss.py:50-67,116-148): one wave per
// sample.  wrong[b] = #joints whose error / bbox-diagonal(gt) > T; epe[b] = sum of errors of joints 1..J-2 (the
// reference's joint range quirk).  Sums over the batch are left to the caller (device tensors, no host sync).
__global__ __launch_bounds__(64) void keypoint_metrics_kernel(const float* pred, const float* gt, int gt_stride, int j, float T,
                                                              int* wrong, float* epe) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const float* g = gt + (long)b * j * gt_stride;
    const float* p = pred + (long)b * j * 2;
    float xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY;
    for (int k = lane; k < j; k += 64) {
        const float x = g[k * gt_stride], y = g[k * gt_stride + 1];
        xmin = fminf(xmin, x); xmax = fmaxf(xmax, x); ymin = fminf(ymin, y); ymax = fmaxf(ymax, y);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        xmin = fminf(xmin, __shfl_xor(xmin, o)); xmax = fmaxf(xmax, __shfl_xor(xmax, o));
        ymin = fminf(ymin, __shfl_xor(ymin, o)); ymax = fmaxf(ymax, __shfl_xor(ymax, o));
    }
    const float diag = sqrtf((xmax - xmin) * (xmax - xmin) + (ymax - ymin) * (ymax - ymin));
    int w = 0;
    float e = 0.f;
    for (int k = lane; k < j; k += 64) {
        const float dx = g[k * gt_stride] - p[k * 2], dy = g[k * gt_stride + 1] - p[k * 2 + 1];
        const float dist = sqrtf(dx * dx + dy * dy);
        if (dist / diag > T) ++w;
        if (k >= 1 && k <= j - 2) e += dist;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { w += __shfl_xor(w, o); e += __shfl_xor(e, o); }
    if (lane == 0) { wrong[b] = w; epe[b] = e; }
}

extern "C" int lh_keypoint_metrics(const float* pred, const float* gt, int gt_stride, int b, int j, float T, int* wrong,
                                   float* epe, void* stream) {
    LH_REQUIRE(pred && gt && wrong && epe && b > 0 && j > 2 && gt_stride >= 2, "lh_keypoint_metrics: bad arguments");
    hipLaunchKernelGGL(keypoint_metrics_kernel, dim3(b), dim3(64), 0, (hipStream_t)stream, pred, gt, gt_stride, j, T, wrong, epe);
    LH_LAUNCH_CHECK("keypoint_metrics launch");
    return LH_OK;
}

// ------------------------------------------------------------------------------------------------ Adam
__global__ void adam_tick_kernel(const double* hyper, int* step, float* derived) {
    const int t = *step + 1;
    *step = t;
    const double lr = hyper[0], b1 = hyper[1], b2 = hyper[2];
    derived[0] = (float)(lr / (1.0 - pow(b1, (double)t)));      // step size
    derived[1] = (float)sqrt(1.0 - pow(b2, (double)t));         // sqrt of bias correction 2
    derived[2] = (float)b1;
    derived[3] = (float)b2;
    derived[4] = (float)hyper[3];
}

__global__ __launch_bounds__(256) void adam_kernel(float* p, const float* g, float* m, float* v, long numel,
                                                   const float* derived, float gscale) {
    const float step_size = derived[0], bc2 = derived[1], b1 = derived[2], b2 = derived[3], eps = derived[4];
    const long nvec = numel / 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long)gridDim.x * 256) {
        float4 pp = reinterpret_cast<float4*>(p)[i];
        const float4 gg = LH_NT_ADAM ? lh_ld_nt(reinterpret_cast<const float4*>(g) + i) : reinterpret_cast<const float4*>(g)[i];
        float4 mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
        float* P = &pp.x; const float* G = &gg.x; float* M = &mm.x; float* V = &vv.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gk = G[k] * gscale;
            M[k] = M[k] * b1 + gk * (1.f - b1);
            V[k] = V[k] * b2 + gk * gk * (1.f - b2);
            P[k] -= step_size * (M[k] / (sqrtf(V[k]) / bc2 + eps));
        }
        reinterpret_cast<float4*>(p)[i] = pp;
        reinterpret_cast<float4*>(m)[i] = mm;
        reinterpret_cast<float4*>(v)[i] = vv;
    }
    if (blockIdx.x == 0)
        for (long i = nvec * 4 + threadIdx.x; i < numel; i += 256) {
            const float gk = g[i] * gscale;
            m[i] = m[i] * b1 + gk * (1.f - b1);
            v[i] = v[i] * b2 + gk * gk * (1.f - b2);
            p[i] -= step_size * (m[i] / (sqrtf(v[i]) / bc2 + eps));
        }
}

extern "C" int lh_adam_tick(const double* hyper, int* step, float* derived, void* stream) {
    LH_REQUIRE(hyper && step && derived, "lh_adam_tick: null pointer");
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, hyper, step, derived);
    LH_LAUNCH_CHECK("adam_tick launch");
    return LH_OK;
}

extern "C" int lh_adam_apply(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long numel, const float* derived,
                             float grad_scale, void* stream) {
    LH_REQUIRE(param && grad && exp_avg && exp_avg_sq && derived && numel > 0, "lh_adam_apply: bad arguments");
    LH_REQUIRE((((size_t)param | (size_t)grad | (size_t)exp_avg | (size_t)exp_avg_sq) & 15) == 0, "lh_adam_apply: slices must start on 16-byte boundaries");
    const long nvec = numel / 4;
    const int grid = (int)((nvec + 255) / 256 > 2048 ? 2048 : ((nvec + 255) / 256 < 1 ? 1 : (nvec + 255) / 256));
    hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, numel, derived, grad_scale);
    LH_LAUNCH_CHECK("adam launch");
    return LH_OK;
}

extern "C" int lh_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long numel,
                            const double* hyper, int* step, float* derived, float grad_scale, void* stream) {
    LH_REQUIRE(param && grad && exp_avg && exp_avg_sq && hyper && step && derived && numel > 0, "lh_adam_step: bad arguments");
    const int rc = lh_adam_tick(hyper, step, derived, stream);
    if (rc) return rc;
    return lh_adam_apply(param, grad, exp_avg, exp_avg_sq, numel, derived, grad_scale, stream);
}

// ------------------------------------------------------------------------------------------------ head bias gradient
// d(bias)[c] = sum over n, h, w of an NCHW fp32 gradient (the head's 1x1 convolution with bias, pose_resnet.py:169-175):
// one workgroup per (channel, slice of the batch), fp64 partials, fixed order -> deterministic.
__global__ __launch_bounds__(256) void channel_sum_nchw_kernel(const float* x, int n, int c, int hw, double* partial, int slices) {
    __shared__ double red[4];
    const int ch = blockIdx.x, sl = blockIdx.y;
    const int n0 = (int)((long)n * sl / slices), n1 = (int)((long)n * (sl + 1) / slices);
    double acc = 0.0;
    for (int b = n0; b < n1; ++b) {
        const float* p = x + ((long)b * c + ch) * hw;
        if ((hw & 3) == 0) {                     // planes are 16-byte aligned: four elements per load, same summation order per thread
            const float4* p4 = reinterpret_cast<const float4*>(p);
            for (int i = threadIdx.x; i < hw / 4; i += 256) {
                const float4 v = p4[i];
                acc += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
            }
        } else {
            for (int i = threadIdx.x; i < hw; i += 256) acc += (double)p[i];
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[(long)ch * slices + sl] = red[0] + red[1] + red[2] + red[3];
}

__global__ void channel_sum_final_kernel(const double* partial, int c, int slices, float* out) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= c) return;
    double a = 0.0;
    for (int s = 0; s < slices; ++s) a += partial[(long)ch * slices + s];
    out[ch] = (float)a;
}

extern "C" size_t lh_channel_sum_workspace_bytes(int c) { return (size_t)c * 64 * sizeof(double); }

extern "C" int lh_channel_sum_nchw(const float* x, int n, int c, int hw, float* out, void* workspace, void* stream) {
    LH_REQUIRE(x && out && workspace && n > 0 && c > 0 && hw > 0, "lh_channel_sum_nchw: bad arguments");
    const int slices = n < 64 ? n : 64;
    hipLaunchKernelGGL(channel_sum_nchw_kernel, dim3(c, slices), dim3(256), 0, (hipStream_t)stream, x, n, c, hw, (double*)workspace, slices);
    hipLaunchKernelGGL(channel_sum_final_kernel, dim3((c + 63) / 64), dim3(64), 0, (hipStream_t)stream, (const double*)workspace, c, slices, out);
    LH_LAUNCH_CHECK("channel_sum launch");
    return LH_OK;
}

// Bias gradient of a convolution / transposed convolution with bias inside the network (DECONV_WITH_BIAS,
// pose_resnet.py:149,227): out[ch] = sum over the pixels of an NHWC gradient of the run precision, channels [0, c) of rows of
// `stride` elements.  A workgroup takes a slice of the pixels: thread (row group r, 16-byte chunk q) sums its chunk's
// elements over rows r, r + R, ... in fp64, the row groups are folded through LDS in a fixed order, and the slices by
// channel_sum_final_kernel: deterministic.
template <typename T>
__global__ __launch_bounds__(256) void channel_sum_nhwc_kernel(const T* x, long pixels, int c, int stride, double* partial, int slices) {
    constexpr int EPC = 16 / sizeof(T);
    __shared__ double red[256 * EPC];
    const int nchunk = (c + EPC - 1) / EPC;              // 16-byte chunks per row that hold requested channels
    const int sl = blockIdx.x;
    const long p0 = pixels * sl / slices, p1 = pixels * (sl + 1) / slices;
    for (int q0 = 0; q0 < nchunk; q0 += 256) {           // <= 256 chunks at a time (2048 bf16 channels)
        const int cw = nchunk - q0 < 256 ? nchunk - q0 : 256;
        int R = 1;                                       // row groups: the largest power of two with R * cw <= 256
        while (R * 2 * cw <= 256) R *= 2;
        const int r = threadIdx.x / cw, q = threadIdx.x - r * cw;
        double acc[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[e] = 0.0;
        if (r < R) {
            for (long px = p0 + r; px < p1; px += R) {
                float v[EPC];
                unpack16<T>(*reinterpret_cast<const uint4*>(x + px * stride + (long)(q0 + q) * EPC), v);
#pragma unroll
                for (int e = 0; e < EPC; ++e) acc[e] += (double)v[e];
            }
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < EPC; ++e) red[threadIdx.x * EPC + e] = acc[e];
        __syncthreads();
        for (int t = threadIdx.x; t < cw * EPC; t += 256) {
            const int qq = t / EPC, e = t - qq * EPC, ch = (q0 + qq) * EPC + e;
            double a = 0.0;
            for (int rr = 0; rr < R; ++rr) a += red[(rr * cw + qq) * EPC + e];
            if (ch < c) partial[(long)ch * slices + sl] = a;
        }
    }
}

extern "C" int lh_channel_sum_nhwc(const void* x, long pixels, int c, int pix_stride, float* out, void* workspace, int dtype, void* stream) {
    LH_REQUIRE(x && out && workspace && pixels > 0 && c > 0 && pix_stride >= c, "lh_channel_sum_nhwc: bad arguments");
    const int es = lh_dtype_size(dtype);
    LH_REQUIRE(es > 0 && (pix_stride * es) % 16 == 0, "lh_channel_sum_nhwc: pixel rows must be 16-byte aligned (stride %d, dtype %d)", pix_stride, dtype);
    const int slices = pixels < 64 ? (int)pixels : 64;
    LH_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((channel_sum_nhwc_kernel<T>), dim3(slices), dim3(256), 0, (hipStream_t)stream,
                                                   (const T*)x, pixels, c, pix_stride, (double*)workspace, slices));
    hipLaunchKernelGGL(channel_sum_final_kernel, dim3((c + 63) / 64), dim3(64), 0, (hipStream_t)stream, (const double*)workspace, c, slices, out);
    LH_LAUNCH_CHECK("channel_sum_nhwc launch");
    return LH_OK;
}

// ------------------------------------------------------------------------------------------------ PCK curve / AUC
// pred_eval (src/utils/argparser.py:326-388) on the device: for every threshold, the number of VISIBLE
// joints (gt[..][2] == 1) whose error -- pixel distance, divided by the sample's bbox size when bb is given ('pckb') --
// is < thr[t].  float64 like the NumPy original, integer atomics (exact, order independent: ranks add their counts with one
// small all-reduce).  diff_row[s] = sum of the pixel errors of ALL joints of sample s (the EPE numerator).
__global__ void pck_curve_kernel(const float* pred, const float* gt, int gt_stride, const float* bb, int n, int j,
                                 const double* thr, int nthr, unsigned long long* counts, unsigned long long* nvis,
                                 double* diff_row) {
    const int sidx = blockIdx.x;
    if (sidx >= n) return;
    __shared__ double err[64];
    __shared__ int vis[64];
    for (int k = threadIdx.x; k < j; k += blockDim.x) {
        const float* g = gt + ((long)sidx * j + k) * gt_stride;
        const float* q = pred + ((long)sidx * j + k) * 2;
        const double dx = (double)g[0] - (double)q[0], dy = (double)g[1] - (double)q[1];
        err[k] = sqrt(dx * dx + dy * dy);
        vis[k] = g[2] == 1.f ? 1 : 0;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double sum = 0.0;
        int nv = 0;
        for (int k = 0; k < j; ++k) { sum += err[k]; nv += vis[k]; }
        diff_row[sidx] = sum;
        if (nv) atomicAdd(nvis, (unsigned long long)nv);
    }
    const double scale = bb ? (double)bb[sidx] : 1.0;
    for (int t = threadIdx.x; t < nthr; t += blockDim.x) {
        int c = 0;
        for (int k = 0; k < j; ++k)
            if (vis[k] && err[k] / scale < thr[t]) ++c;
        if (c) atomicAdd(counts + t, (unsigned long long)c);
    }
}

extern "C" int lh_pck_curve(const float* pred, const float* gt, int gt_stride, const float* bb, int n, int j, const double* thr,
                            int nthr, unsigned long long* counts, unsigned long long* nvis, double* diff_row, void* stream) {
    LH_REQUIRE(pred && gt && thr && counts && nvis && diff_row && n > 0 && j > 0 && j <= 64 && gt_stride >= 3 && nthr > 0,
               "lh_pck_curve: bad arguments (j <= 64, gt rows of >= 3 values: x, y, visibility)");
    hipLaunchKernelGGL(pck_curve_kernel, dim3(n), dim3(128), 0, (hipStream_t)stream, pred, gt, gt_stride, bb, n, j, thr, nthr,
                       counts, nvis, diff_row);
    LH_LAUNCH_CHECK("pck_curve launch");
    return LH_OK;
}

// ------------------------------------------------------------------------------------------------ strided fp32 copy
// dst[i0][i1][i2][i3] = src[i0][i1][i2][i3] with arbitrary element strides on both sides: the small layout shuffles of a
// step (stem weight [O][3][k][k] <-> [O][k][k'][4] staging, head-gradient crop, bias padding) without a framework op.
__global__ void copy_strided_f32_kernel(float* dst, const float* src, int n0, int n1, int n2, int n3, long d0, long d1, long d2,
                                        long d3, long s0, long s1, long s2, long s3) {
    const long total = (long)n0 * n1 * n2 * n3;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int i3 = (int)(i % n3);
        long t = i / n3;
        const int i2 = (int)(t % n2);
        t /= n2;
        const int i1 = (int)(t % n1), i0 = (int)(t / n1);
        dst[i0 * d0 + i1 * d1 + i2 * d2 + i3 * d3] = src[i0 * s0 + i1 * s1 + i2 * s2 + i3 * s3];
    }
}

extern "C" int lh_copy_strided_f32(float* dst, const float* src, const int* shape4, const long* dst_strides4, const long* src_strides4,
                                   void* stream) {
    LH_REQUIRE(dst && src && shape4 && dst_strides4 && src_strides4, "lh_copy_strided_f32: null pointer");
    const long total = (long)shape4[0] * shape4[1] * shape4[2] * shape4[3];
    LH_REQUIRE(shape4[0] > 0 && shape4[1] > 0 && shape4[2] > 0 && shape4[3] > 0, "lh_copy_strided_f32: empty shape");
    const int grid = (int)((total + 255) / 256 > 1024 ? 1024 : (total + 255) / 256);
    hipLaunchKernelGGL(copy_strided_f32_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dst, src, shape4[0], shape4[1], shape4[2],
                       shape4[3], dst_strides4[0], dst_strides4[1], dst_strides4[2], dst_strides4[3], src_strides4[0], src_strides4[1],
                       src_strides4[2], src_strides4[3]);
    LH_LAUNCH_CHECK("copy_strided_f32 launch");
    return LH_OK;
}

// ---- gradient bucket staging for bf16 collectives (parallel.GradSync(compress="bf16")): fp32 arena slice <-> bf16 buffer,
// 16 bytes of bf16 per lane (round to nearest even through the hardware convert; NaN stays NaN)
__global__ void cast_f32_bf16_kernel(const float* __restrict__ src, bf16* __restrict__ dst, long n) {
    const long i0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    const long stride = (long)gridDim.x * blockDim.x * 8;
    for (long i = i0; i < n; i += stride) {
        if (i + 8 <= n && (((size_t)(src + i) | (size_t)(dst + i)) & 15) == 0) {
            const float4 a = *reinterpret_cast<const float4*>(src + i), b = *reinterpret_cast<const float4*>(src + i + 4);
            Vec16<bf16> v;
            v.e[0] = (bf16)a.x; v.e[1] = (bf16)a.y; v.e[2] = (bf16)a.z; v.e[3] = (bf16)a.w;
            v.e[4] = (bf16)b.x; v.e[5] = (bf16)b.y; v.e[6] = (bf16)b.z; v.e[7] = (bf16)b.w;
            *reinterpret_cast<uint4*>(dst + i) = v.u;
        } else {
            for (long k = i; k < n && k < i + 8; ++k) dst[k] = (bf16)src[k];
        }
    }
}

__global__ void cast_bf16_f32_kernel(const bf16* __restrict__ src, float* __restrict__ dst, long n) {
    const long i0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    const long stride = (long)gridDim.x * blockDim.x * 8;
    for (long i = i0; i < n; i += stride) {
        if (i + 8 <= n && (((size_t)(src + i) | (size_t)(dst + i)) & 15) == 0) {
            Vec16<bf16> v;
            v.u = *reinterpret_cast<const uint4*>(src + i);
            *reinterpret_cast<float4*>(dst + i) = float4{(float)v.e[0], (float)v.e[1], (float)v.e[2], (float)v.e[3]};
            *reinterpret_cast<float4*>(dst + i + 4) = float4{(float)v.e[4], (float)v.e[5], (float)v.e[6], (float)v.e[7]};
        } else {
            for (long k = i; k < n && k < i + 8; ++k) dst[k] = (float)src[k];
        }
    }
}

extern "C" int lh_cast_f32_bf16(float* src, void* dst, long n, int to_f32, void* stream) {
    LH_REQUIRE(src && dst && n > 0, "lh_cast_f32_bf16: bad arguments");
    const long groups = (n + 7) / 8;
    const int grid = (int)((groups + 255) / 256 > 8192 ? 8192 : (groups + 255) / 256);
    if (to_f32) hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16*)dst, src, n);
    else hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, (bf16*)dst, n);
    LH_LAUNCH_CHECK("cast_f32_bf16 launch");
    return LH_OK;
}


// The local half of the DIRECT gradient exchange (parallel.GradSync(algo="direct"); SURVEY 8e: reduce-scatter + all-gather with all
// seven xGMI peers at once instead of a ring): after the all-to-all a rank holds `rows` chunks of `len` elements -- chunk r = rank r's
// contribution to the slice this rank owns -- and sums them IN RANK ORDER (fp32 accumulation; bf16 chunks are widened, the sum is
// rounded once), so every rank's owned slice, and after the all-gather every rank's whole bucket, holds bit-identical values.
template <typename T>
__global__ void sum_chunks_kernel(const T* __restrict__ in, T* __restrict__ out, int rows, long len) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += stride) {
        float acc = (float)in[i];
        for (int r = 1; r < rows; ++r) acc += (float)in[(long)r * len + i];
        out[i] = (T)acc;
    }
}

extern "C" int lh_sum_chunks(const void* in, void* out, int rows, long len, int dtype, void* stream) {
    LH_REQUIRE(in && out && rows >= 1 && len > 0, "lh_sum_chunks: bad arguments");
    LH_REQUIRE(dtype == LH_F32 || dtype == LH_BF16, "lh_sum_chunks: fp32 or bf16 chunks (dtype %d)", dtype);
    const int grid = (int)((len + 255) / 256 > 4096 ? 4096 : (len + 255) / 256);
    if (dtype == LH_F32) hipLaunchKernelGGL(sum_chunks_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)in, (float*)out, rows, len);
    else hipLaunchKernelGGL(sum_chunks_kernel<bf16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16*)in, (bf16*)out, rows, len);
    LH_LAUNCH_CHECK("sum_chunks launch");
    return LH_OK;
}
