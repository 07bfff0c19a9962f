// Device helpers shared by the forward and backward field-MLP kernels.
#pragma once
#include "niw_common.h"

// weight fragments kept in flight per wave (each covers 4 MFMAs = 256 cycles)
#ifndef NIW_RING_DEPTH
#define NIW_RING_DEPTH 8
#endif

// cache policy of the streaming workspace stores (aux of raw_buffer_store: 0 = default, 2 = nt, 16 = sc1).
// Measured on MI355X, cfg2 step: nt -2 % on the forward and the dX chain (their stores are read again only by the
// dW pass, gigabytes later); sc1 / sc1+nt no better.  (Also measured: a quad-row workspace layout with 16-byte
// stores, 4x fewer store instructions for the same bytes, gained 2.4 % on the forward -- the cost of the saves is
// their bytes, not their instruction count: 119-124 TFLOP/s with them, 140 without.)
#ifndef NIW_STORE_AUX
#define NIW_STORE_AUX 2
#endif

namespace niw {

// band frequency 2^k * fp32(pi)  (reference: 2**arange(L) * np.pi evaluated in fp32, nerf.py:478)
__device__ __forceinline__ float band_freq(int k) { return 3.14159274101257324f * (float)(1 << k); }

// ---------------------------------------------------------------------------------------------
// Buffer addressing.  Measured on MI355X (tools/mfma_ceiling.hip): a global_load/store whose 64
// lanes each carry a 64-bit address costs the issuing wave ~17 (load) / ~24 (store) cycles that a
// dependent MFMA chain cannot hide -- 6.5 % + 4.7 % of this kernel shape.  The same access through
// a buffer instruction (SGPR descriptor + SGPR byte offset + ONE 32-bit lane offset shared by all
// accesses) is free: 154 TFLOP/s (98 %) versus 136.  Every hot-loop access therefore goes through
// raw_buffer_load/store; byte offsets must stay below 2^31 (the host caps Mpad accordingly).
// ---------------------------------------------------------------------------------------------
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}
template <int AUX>
__device__ __forceinline__ f32x4 buf_load4_aux(rsrc_t r, int voff, int soff) {      // AUX: cache policy (2 = nt: touched once, whole lines)
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, AUX));
}
__device__ __forceinline__ f32x4 buf_load4(rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ float buf_load1(rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void buf_store1(float v, rsrc_t r, int voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, NIW_STORE_AUX);
}
__device__ __forceinline__ void buf_store4(float v0, float v1, float v2, float v3, rsrc_t r, int voff, int soff) {
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, f32x4{v0, v1, v2, v3}), r, voff, soff, NIW_STORE_AUX);
}

__device__ __forceinline__ void buf_store2(unsigned d0, unsigned d1, rsrc_t r, int voff, int soff) {
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{d0, d1}, r, voff, soff, NIW_STORE_AUX);
}

// The packed-weight image of one network (niw_mlp_pack_weights) as seen by a wave.
struct PackedWeights {
    rsrc_t rsrc;
    const float* base;
    int lane16;          // this lane's byte offset inside a 1 KiB fragment
};
__device__ __forceinline__ PackedWeights packed_weights(const float* packed, int lane) {
    return PackedWeights{make_rsrc(packed), packed, lane * 16};
}

// The activation / gradient workspaces are "quad-row" images: [row / 4][Mpad][4] floats, i.e. the four consecutive rows that one
// lane holds in accumulator registers 4q .. 4q+3 (rows 8q + 4h + {0,1,2,3} of a 32-row block) sit side by side for each sample.
// A lane stores them as ONE 16-byte access (a wave: 512 contiguous bytes per lane half), a quarter of the store instructions of
// the plain [row][Mpad] image for the same bytes (round 2: +1.6 % on the training forward; the dW loader, which splits the 16
// bytes over four LDS rows, is 1.3 % faster with it as well).  Row r of a window still starts r * Mpad floats after the window base
// whenever r is a multiple of 4, so window bases and 32-row descriptors are computed exactly as for the plain image:
// quad (row0 + 8q + 4h) / 4 of sample m = descriptor(row0) + 8q*pitch4 + voff4,  voff4 = (h*Mpad + m) * 16 bytes.
struct RowWindow {
    const float* base;   // first row of the window (a multiple of 4 rows into the workspace)
    int pitch4;          // Mpad * 4 bytes (= the bytes of one quad per 4 samples; 8 rows = 2 quads = 8*pitch4 bytes)
    int voff4;           // (h*Mpad + m) * 16 bytes
    // The window base is wave-uniform by construction; readfirstlane says so to the compiler.  (Where it had kept such a base in
    // VGPRs -- 66 stores of the training forward -- every use of the descriptor was wrapped in a readfirstlane / compare /
    // exec-mask "waterfall" loop.)
    __device__ __forceinline__ rsrc_t rsrc(int row) const {
        const unsigned long long p = reinterpret_cast<unsigned long long>(base + (long long)row * (pitch4 >> 2));
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)p), hi = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
        return make_rsrc(reinterpret_cast<const void*>(((unsigned long long)hi << 32) | lo));
    }
};
// row offset inside a 32-row block of accumulator register r (the lane half's +4 rows live in voff4)
__device__ __forceinline__ constexpr int reg_row(int r) { return (r & 3) + 8 * (r >> 2); }

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
// v_mfma_f32_32x32x2_f32 cannot issue for 64 cycles, so work placed between two MFMAs of one
// chain is free.  Only the last block's epilogue is exposed.
//
//   * weight fragments: one coalesced 16 B/lane buffer load per (row block, k-block), a ring keeps
//     NIW_RING_DEPTH loads (>= 2048 MFMA cycles) in flight; the ring index is compile-time;
//   * per-block epilogue inputs (parked gradients, head weights) are fetched by `pol.pre(nb, buf)` at the START of block nb
//     and consumed during block nb+1; a policy with kAccInit also supplies the block's INITIAL accumulator (the forward: the
//     bias fragment, fetched one block ahead by `pol.acc_init`), so that its epilogue has no add;
//   * `pol.epi(nb, r, acc_r, pre_r)` handles accumulator register r of block nb.
//
// `wp` points at the first fragment of the (sub-range of the) layer inside the packed image;
// STRIDE = row blocks per k-block there ([k-block][row block][lane][4]).
// sched_barrier(0) after every k-block pins this order (hipcc otherwise sinks the loads next to
// their uses and gathers the epilogue at the end).
// ---------------------------------------------------------------------------------------------
// What one layer hands to the next (NIW_LAYER_CARRY): the first NIW_RING_DEPTH weight fragments of the next layer, requested
// while the current layer's last k-blocks run, and the bias fragment of its first row block -- otherwise every layer starts with
// one exposed L2 round trip for each.
struct LayerCarry {
    f32x4 ring[NIW_RING_DEPTH];
    f32x16 cin0;
    bool valid = false;      // compile-time known at every use (the kernel bodies are straight-line)
};
struct NextLayer {
    int w_base = -1;         // byte offset of the next layer's first fragment in the packed image (-1: nothing to prefetch)
    int w_stride = 0;        // bytes between its consecutive k-blocks of row block 0 (STRIDE * 1024)
    int bias_bytes = -1;     // byte offset of its packed bias (policies with kAccInit), or -1
    int hoff = 0;
};

template <int KB1, int KB2, int NB, int STRIDE, typename Policy, bool CARRY_IN = false, bool CARRY_OUT = false>
__device__ __forceinline__ void stream_layer(const PackedWeights& pw, const f32x4* __restrict__ wp, const float (&b1)[4 * KB1],
                                             const float (&b2)[4 * (KB2 > 0 ? KB2 : 1)], Policy& pol, LayerCarry* carry = nullptr,
                                             NextLayer next = NextLayer{}) {
    constexpr int KB = KB1 + KB2, N = NB * KB, D = NIW_RING_DEPTH, GAPS = 4 * KB;
    constexpr int G0 = GAPS >= 32 ? 8 : 0;                 // first gap used by the epilogue
    constexpr int GS = (GAPS - G0) / 16 > 0 ? (GAPS - G0) / 16 : 1;
    const int base = (int)(reinterpret_cast<const char*>(wp) - reinterpret_cast<const char*>(pw.base));   // wave-uniform
    // Policies with kAccInit start every block's accumulation from a 16-register value of their own (the forward: the bias
    // fragment, fetched one block ahead) instead of a literal zero, which takes the bias add out of the epilogue.
    constexpr bool INIT = Policy::kAccInit;
    static_assert(!(CARRY_IN || CARRY_OUT) || (N % D == 0 && KB >= D), "carried rings need whole ring turns per layer");
    f32x16 cin[2];
    f32x4 ring[D];
    if (CARRY_IN) {
        if (INIT) cin[0] = carry->cin0;
#pragma unroll
        for (int i = 0; i < D; ++i) ring[i] = carry->ring[i];
    } else {
        if (INIT) pol.acc_init(0, cin[0]);
#pragma unroll
        for (int i = 0; i < D; ++i)
            if (i < N) ring[i] = buf_load4(pw.rsrc, pw.lane16, base + ((i % KB) * STRIDE + i / KB) * 1024);
    }
    f32x16 acc[2];
    float pre[2][16];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        f32x16& cur = acc[nb & 1];
        pol.pre(nb, pre[nb & 1]);
        if (INIT && nb + 1 < NB) pol.acc_init(nb + 1, cin[(nb + 1) & 1]);
        if (CARRY_OUT && INIT && nb == NB - 1) {           // the next layer's first bias fragment
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = buf_load4(pw.rsrc, next.hoff, next.bias_bytes + g * 16);
                carry->cin0[4 * g] = v[0]; carry->cin0[4 * g + 1] = v[1]; carry->cin0[4 * g + 2] = v[2]; carry->cin0[4 * g + 3] = v[3];
            }
        }
#pragma unroll
        for (int q = 0; q < KB; ++q) {
            const int i = nb * KB + q;
            const f32x4 a = ring[i % D];
            if (i + D < N) ring[i % D] = buf_load4(pw.rsrc, pw.lane16, base + (((i + D) % KB) * STRIDE + (i + D) / KB) * 1024);
            else if (CARRY_OUT) ring[i % D] = buf_load4(pw.rsrc, pw.lane16, next.w_base + (i + D - N) * next.w_stride);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float bv = q < KB1 ? b1[4 * (q < KB1 ? q : 0) + t] : b2[4 * (q >= KB1 ? q - KB1 : 0) + t];
                // the block's first MFMA takes a literal zero accumulator (no 16-register clear, no VALU->MFMA hazard)
                cur = mfma32(a[t], bv, (q == 0 && t == 0) ? (INIT ? cin[nb & 1] : f32x16{0.f}) : cur);
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
    if (CARRY_OUT) {
#pragma unroll
        for (int i = 0; i < D; ++i) carry->ring[i] = ring[i];
    }
}

}  // namespace niw
