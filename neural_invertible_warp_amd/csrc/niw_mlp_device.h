// Device helpers shared by the forward and backward field-MLP kernels.
#pragma once
#include "niw_common.h"

// weight fragments kept in flight per wave (each covers 4 MFMAs = 256 cycles)
#ifndef NIW_RING_DEPTH
#define NIW_RING_DEPTH 8
#endif

namespace niw {

// band frequency 2^k * fp32(pi)  (reference: 2**arange(L) * np.pi evaluated in fp32, nerf.py:478)
__device__ __forceinline__ float band_freq(int k) { return 3.14159274101257324f * (float)(1 << k); }

// ---------------------------------------------------------------------------------------------
// Streaming register-chained layer.
//
//   out[n][m] = sum_k A[n][k] * B[k][m]      A = packed weight fragments (global, L2 resident)
//                                            B = this lane's operand registers (b1 then b2)
//
// Row blocks (32 output rows) are the OUTER loop, k-blocks the inner one, so exactly one
// accumulator (16 registers) is being produced at a time.  While block nb accumulates, the
// epilogue of block nb-1 (accumulator -> activation/mask -> next-layer operand + store) is
// spread over the MFMA gaps of block nb: the wave is in-order and a dependent
// v_mfma_f32_32x32x2_f32 cannot issue for 64 cycles, so VALU/VMEM work placed between two MFMAs
// of one chain is free.  Only the last block's epilogue is exposed.
//
//   * weight fragments: one coalesced 16 B/lane load per (row block, k-block), an 8-deep ring
//     keeps 8 loads (>= 2048 MFMA cycles) in flight; the ring index is compile-time;
//   * per-block epilogue inputs (bias / saved activation for the ReLU mask) are fetched by
//     `pol.pre(nb, buf)` at the START of block nb and consumed during block nb+1;
//   * `pol.epi(nb, r, acc_r, pre_r)` handles accumulator register r of block nb.
//
// STRIDE = row blocks per k-block in the packed image ([k-block][row block][lane][4]).
// sched_barrier(0) after every k-block pins this order (hipcc otherwise sinks the loads next to
// their uses and gathers the epilogue at the end).
// ---------------------------------------------------------------------------------------------
template <int KB1, int KB2, int NB, int STRIDE, typename Policy>
__device__ __forceinline__ void stream_layer(const f32x4* __restrict__ wp, int lane, const float (&b1)[4 * KB1],
                                             const float (&b2)[4 * (KB2 > 0 ? KB2 : 1)], Policy& pol) {
    constexpr int KB = KB1 + KB2, N = NB * KB, D = NIW_RING_DEPTH, GAPS = 4 * KB;
    constexpr int G0 = GAPS >= 32 ? 8 : 0;                 // first gap used by the epilogue
    constexpr int GS = (GAPS - G0) / 16 > 0 ? (GAPS - G0) / 16 : 1;
    f32x4 ring[D];
#pragma unroll
    for (int i = 0; i < D; ++i)
        if (i < N) ring[i] = wp[((i % KB) * STRIDE + i / KB) * 64 + lane];
    f32x16 acc[2];
    float pre[2][16];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        f32x16& cur = acc[nb & 1];
#pragma unroll
        for (int r = 0; r < 16; ++r) cur[r] = 0.f;
        pol.pre(nb, pre[nb & 1]);
#pragma unroll
        for (int q = 0; q < KB; ++q) {
            const int i = nb * KB + q;
            const f32x4 a = ring[i % D];
            if (i + D < N) ring[i % D] = wp[(((i + D) % KB) * STRIDE + (i + D) / KB) * 64 + lane];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float bv = q < KB1 ? b1[4 * (q < KB1 ? q : 0) + t] : b2[4 * (q >= KB1 ? q - KB1 : 0) + t];
                cur = mfma32(a[t], bv, cur);
                const int gap = 4 * q + t;
                if (nb > 0 && gap >= G0 && (gap - G0) % GS == 0 && (gap - G0) / GS < 16) {
                    const int r = (gap - G0) / GS;
                    pol.epi(nb - 1, r, acc[(nb - 1) & 1][r], pre[(nb - 1) & 1][r]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (nb > 0 && G0 + 15 * GS >= GAPS) {              // short layers: epilogue registers that found no gap
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (G0 + r * GS >= GAPS) pol.epi(nb - 1, r, acc[(nb - 1) & 1][r], pre[(nb - 1) & 1][r]);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) pol.epi(NB - 1, r, acc[(NB - 1) & 1][r], pre[(NB - 1) & 1][r]);
}

}  // namespace niw
