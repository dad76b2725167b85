// Shared device/host helpers for liblighthand_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/lighthand_hip.h"

typedef __bf16 bf16;
typedef _Float16 f16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

void lh_set_error(const char* fmt, ...);

#define LH_REQUIRE(cond, ...)                 \
    do {                                      \
        if (!(cond)) {                        \
            lh_set_error(__VA_ARGS__);        \
            return LH_ERR_ARG;                \
        }                                     \
    } while (0)

#define LH_LAUNCH_CHECK(what)                                                       \
    do {                                                                            \
        hipError_t e__ = hipGetLastError();                                         \
        if (e__ != hipSuccess) {                                                    \
            lh_set_error("%s: %s", what, hipGetErrorString(e__));                   \
            return LH_ERR_HIP;                                                      \
        }                                                                           \
    } while (0)

// Dispatch a lambda-like macro body on the activation dtype.
#define LH_DISPATCH_DTYPE(dtype, T, ...)                        \
    switch (dtype) {                                            \
        case LH_F32: { typedef float T; __VA_ARGS__; } break;   \
        case LH_BF16: { typedef bf16 T; __VA_ARGS__; } break;   \
        case LH_F16: { typedef f16 T; __VA_ARGS__; } break;     \
        default: lh_set_error("unsupported dtype %d", dtype); return LH_ERR_ARG; \
    }

template <typename T> __device__ __forceinline__ float to_f(T v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f(float v) { return (T)v; }

// 16-byte vector of T  <-> floats
template <typename T> struct Vec16 {
    static constexpr int N = 16 / sizeof(T);
    union { uint4 u; T e[N]; };
};

template <typename T> __device__ __forceinline__ void unpack16(const uint4& u, float* f) {
    Vec16<T> v; v.u = u;
#pragma unroll
    for (int i = 0; i < Vec16<T>::N; ++i) f[i] = to_f<T>(v.e[i]);
}
template <typename T> __device__ __forceinline__ uint4 pack16(const float* f) {
    Vec16<T> v;
#pragma unroll
    for (int i = 0; i < Vec16<T>::N; ++i) v.e[i] = from_f<T>(f[i]);
    return v.u;
}

// 16-byte loads with the non-temporal cache policy (global_load_dwordx4 ... nt): last-use streams
__device__ __forceinline__ float4 lh_ld_nt(const float4* p) {
    const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
    return float4{v[0], v[1], v[2], v[3]};
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

// XCD-aware work-item index (speed only, never correctness).  Workgroup ids are dealt round-robin over the 8 XCDs,
// each with a private L2, so ids b and b+8 share an L2 while neighbours b, b+1 do not.  The remap gives every XCD a
// CONTIGUOUS range of work items (bijective for any nwg), so that tiles sharing an operand panel hit the same L2.
__device__ __forceinline__ int lh_xcd_remap(int orig, int nwg) {
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}
