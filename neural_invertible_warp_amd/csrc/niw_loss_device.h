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
#pragma unroll
    for (int j = 0; j < 4; ++j)
        for (long long i = threadIdx.x + 256 * j; i < total; i += 1024) {
            const float diff = resid[i];
            acc[j] += diff * diff;
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
