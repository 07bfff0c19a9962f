// The photometric mean's sum of squares in ONE fixed order, shared by mse_kernel's two consumers of stored residuals
// (niw_mse_from_residuals in niw_sampling.hip, the closing kernel of niw_train_step in niw_step.hip).
#pragma once
#include "niw_common.h"

namespace niw {
// sum_i resid[i]^2 (each square rounded to fp32, accumulated in fp64) in exactly the order of mse_kernel (niw_sampling.hip) -- a
// workgroup of 1024 threads there: thread t takes the elements t, t + 1024, .. in ascending order, the 64 lanes of a wave fold by an xor
// butterfly, the 16 wave totals are added in wave order.  Here a workgroup of 256 threads stands in for the 1024: thread t plays the
// threads t, t + 256, t + 512, t + 768 (same lane, waves w, w + 4, w + 8, w + 12), so the result is bit-identical.  Every thread of the
// 256-thread workgroup must call it; `red` is 16 doubles of LDS; the total is returned to thread 0 (others: unspecified).
__device__ __forceinline__ double sq_sum_in_mse_order(const float* __restrict__ resid, long long total, double* red) {
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    // four rounds of 1024 elements per trip, all sixteen loads of the thread issued before the first use (the kernel that calls this
    // closes a chain of small dependent launches: a rolled loop paid one cold read per element and played thread); every accumulator
    // still takes its elements in ascending order, and an element beyond the end adds +0.0, which changes nothing
    for (long long base = 0; base < total; base += 4096) {
        float v[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const long long i = base + 1024 * u + threadIdx.x + 256 * j;
                v[u][j] = i < total ? resid[i] : 0.f;
            }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] += v[u][j] * v[u][j];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc[j] += __shfl_xor(acc[j], o);
        if ((threadIdx.x & 63) == 0) red[(threadIdx.x >> 6) + 4 * j] = acc[j];
    }
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0)
        for (int w = 0; w < 16; ++w) t += red[w];
    __syncthreads();                       // (red is free for the next call)
    return t;
}
}  // namespace niw
