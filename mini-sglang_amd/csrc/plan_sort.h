// Launch-order planning shared by the attention kernels: a counting sort of work items by cost, most
// expensive first, done by ONE work group of 1024 threads.  The hardware hands work groups to compute units
// in launch order, so a launch sorted this way is longest-processing-time-first list scheduling, without the
// host ever reading the lengths the costs depend on.
#pragma once
#include "msgl_common.h"

namespace msgl {

constexpr int kPlanBins = 2048;
constexpr int kPlanThreads = 1024;

// cost_of(i) -> cost of item i (clamped to kPlanBins - 1), or a negative value for an item without work.
// plan[0 .. total) = the items with work, non-increasing cost (items of equal cost in arbitrary order: LDS
// atomics); plan[total .. n) = -1.  `bins` and `part` are LDS arrays of kPlanBins / kPlanThreads ints.
template <typename CostFn>
__device__ __forceinline__ void plan_sort_descending(int n, CostFn cost_of, int32_t* __restrict__ plan, int* bins, int* part) {
    static_assert(kPlanBins == 2 * kPlanThreads, "two bins per thread in the scan below");
    const int tid = threadIdx.x;
    for (int i = tid; i < kPlanBins; i += kPlanThreads) bins[i] = 0;
    __syncthreads();
    auto bin_of = [&](int i) -> int {  // ascending bin = descending cost
        const int c = cost_of(i);
        return c < 0 ? -1 : kPlanBins - 1 - min(c, kPlanBins - 1);
    };
    for (int i = tid; i < n; i += kPlanThreads) {
        const int r = bin_of(i);
        if (r >= 0) atomicAdd(&bins[r], 1);
    }
    __syncthreads();
    // exclusive scan of the bins: two per thread, then Hillis-Steele over the partial sums
    const int c0 = bins[2 * tid], c1 = bins[2 * tid + 1];
    part[tid] = c0 + c1;
    __syncthreads();
    for (int d = 1; d < kPlanThreads; d <<= 1) {
        const int add = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += add;
        __syncthreads();
    }
    const int before = part[tid] - (c0 + c1);
    const int total = part[kPlanThreads - 1];
    __syncthreads();
    bins[2 * tid] = before;
    bins[2 * tid + 1] = before + c0;
    __syncthreads();
    for (int i = tid; i < n; i += kPlanThreads) {
        const int r = bin_of(i);
        if (r >= 0) plan[atomicAdd(&bins[r], 1)] = i;
    }
    for (int i = total + tid; i < n; i += kPlanThreads) plan[i] = -1;
}

}  // namespace msgl
