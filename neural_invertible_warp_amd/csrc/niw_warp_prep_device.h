// The warp's code projection code_b = lin_c(code) + code (reference model/nvp/nvp_ndr.py:381) as a device function shared by its own
// launch (niw_warp_prep.hip) and by the front kernel of the train iteration (niw_sampling.hip: one launch fewer per iteration).
#pragma once
#include "niw_common.h"

namespace niw_warp_prep {
constexpr int kHid = 128, kLat = 128, kEa = 26, kEb = 13;
constexpr int kKa = kEa + kLat, kKb = kEb + kLat;                       // 154, 141
constexpr int kBlkA = kHid + kHid * kKa + kHid + kHid + 1;              // 20097
constexpr int kBlkB = kHid + kHid * kKb + kHid + 3 * kHid + 3;          // 18691
constexpr int kBlkC = kLat * kLat + kLat;                               // 16512
constexpr int kOffB = 3 * kBlkA, kOffC = kOffB + 3 * kBlkB;
static_assert(kOffC + 3 * kBlkC == NIW_WARP_PARAM_FLOATS, "flat warp parameter count");

__device__ __forceinline__ float wave_sum(float x) {    // butterfly: every lane ends with the total, same order on every run
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    return x;
}

// code_b[b][v][j] = bc[j] + code[v][j] + sum_k Wc[j][k] code[v][k]: one workgroup of 256 threads per (b, v), one wave per 32 rows j
__device__ __forceinline__ void code_projection(int b, int v, const float* __restrict__ P, const float* __restrict__ code, int B,
                                                float* __restrict__ codeb) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* Wc = P + kOffC + b * kBlkC;
    const float* bc = Wc + kLat * kLat;
    const float c0 = code[v * kLat + lane], c1 = code[v * kLat + 64 + lane];
    // all 64 row loads of the wave are issued before the first butterfly: the kernel is one cold HBM round trip plus arithmetic
    // (rolled, every row waited for its own two loads: 21 us for a [B,128] x [128,128] product; eight at a time still 21)
    float w0[32], w1[32];
#pragma unroll
    for (int jj = 0; jj < 32; ++jj) {
        const int j = wave * 32 + jj;
        w0[jj] = Wc[j * kLat + lane];
        w1[jj] = Wc[j * kLat + 64 + lane];
    }
    const float extra = lane < 32 ? bc[wave * 32 + lane] + code[v * kLat + wave * 32 + lane] : 0.f;
    float mine = 0.f;
#pragma unroll
    for (int jj = 0; jj < 32; ++jj) {
        const float s = wave_sum(w0[jj] * c0 + w1[jj] * c1);
        mine = lane == jj ? s : mine;
    }
    if (lane < 32) codeb[((long long)b * B + v) * kLat + wave * 32 + lane] = mine + extra;     // one coalesced store per wave
}
}  // namespace niw_warp_prep
