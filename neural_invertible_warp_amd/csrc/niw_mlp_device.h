// Device helpers shared by the forward and backward field-MLP kernels.
#pragma once
#include "niw_common.h"

namespace niw {

// band frequency 2^k * fp32(pi)  (reference: 2**arange(L) * np.pi evaluated in fp32, nerf.py:478)
__device__ __forceinline__ float band_freq(int k) { return 3.14159274101257324f * (float)(1 << k); }

// Register-chained GEMM piece: acc[nb] += sum over KB k-blocks of A-fragments (packed weights,
// one coalesced 16 B/lane load per (k-block, row-block), prefetched one k-block ahead) times the
// B operand held in registers (b[4q+t] = slot 8q+4h+t of this lane's sample).
// STRIDE = row-blocks per k-block in the packed image (> NB when only a sub-range of the row
// blocks is computed; `wp` then points at the first block of the sub-range).
template <int KB, int NB, int STRIDE = NB>
__device__ __forceinline__ void gemm_regs(const f32x4* __restrict__ wp, int lane, const float (&b)[4 * KB], f32x16 (&acc)[NB]) {
    f32x4 cur[NB], nxt[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) cur[nb] = wp[nb * 64 + lane];
#pragma unroll
    for (int q = 0; q < KB; ++q) {
        if (q + 1 < KB) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) nxt[nb] = wp[((q + 1) * STRIDE + nb) * 64 + lane];
        }
        // Pin the software pipeline: left alone, hipcc sinks each load to 4 MFMAs before its use and
        // waits vmcnt(0) there, exposing the L2 latency every 256 cycles.  With the barriers all
        // NB loads of k-block q+1 are in flight across the 4*NB MFMAs (>= 1024 cycles) of k-block q.
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[nb] = mfma32(cur[nb][t], b[4 * q + t], acc[nb]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) cur[nb] = nxt[nb];
    }
}


}  // namespace niw
