// Multi-problem launches: up to LH_MULTI_MAX independent problems of ONE kernel instantiation run as ONE grid.
// The per-problem argument blocks travel BY VALUE in the kernel-argument segment (an array indexed with a wave-uniform
// index: scalar loads, no table in device memory, nothing to upload, capturable like any other launch); workgroup b
// belongs to problem i with first[i] <= b < first[i + 1] and runs that problem's body with its local block index.
// Used for the parallel branches of HRNet (pose_hrnet.py:139-185, 247-265: the same layer position of 2-4 branches),
// whose kernels are a few microseconds of work each: one launch instead of four.
#pragma once
#include "common.h"

constexpr int LH_MULTI_MAX = 4;

template <typename A> struct LhMulti {
    A a[LH_MULTI_MAX];
    int first[LH_MULTI_MAX + 1];
    int n;
};

// problem index of this workgroup; bid / nblk = its block index and block count inside that problem
template <typename M> __device__ __forceinline__ int lh_multi_pick(const M& m, int& bid, int& nblk) {
    const int b = blockIdx.x;
    int i = 0;
    while (i + 1 < m.n && b >= m.first[i + 1]) ++i;
    bid = b - m.first[i];
    nblk = m.first[i + 1] - m.first[i];
    return i;
}
