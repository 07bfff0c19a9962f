// Positional encoding of a point / unit direction in MFMA slot order, shared by the exact-fp32 forward (niw_mlp_fwd.hip) and the
// fast-precision forward (niw_mlp_fast.hip): both form the encodings in fp32 with the same arithmetic.
// Reference: NeRF.positional_encoding, model/nerf.py:476-483 (+ the c2f band mask, model/barf_inn_llff.py:427-442).
#pragma once
#include "niw_common.h"
#include "niw_mlp_device.h"

namespace niw {

// One sincos pair (compile-time pair index per lane half, selected by h).  rev[c] = fl32(p_c*pi32)/(2 pi).
template <int L>
__device__ __forceinline__ void enc_pair(const double (&rev)[3], const float* __restrict__ w, int h, int pair0, int pair1,
                                         float& s_out, float& c_out) {
    // pair index -> (coordinate, band); pairs >= 3L are zero padding
    const bool valid0 = pair0 < 3 * L, valid1 = pair1 < 3 * L;
    const int c0 = valid0 ? pair0 / L : 0, k0 = valid0 ? pair0 % L : 0;
    const int c1 = valid1 ? pair1 / L : 0, k1 = valid1 ? pair1 % L : 0;
    const double t = h ? rev[c1] : rev[c0];
    const int k = h ? k1 : k0;
    const float wk = h ? w[k1] : w[k0];
    float s, c;
    sincos_band(t, k, s, c);
    const bool valid = h ? valid1 : valid0;
    s_out = valid ? s * wk : 0.f;
    c_out = valid ? c * wk : 0.f;
}

// enc[4q+t] = feature of slot 8q+4h+t for this lane's half h
template <int L, int NQ>
__device__ __forceinline__ void encode_slots(const float (&p)[3], const float* __restrict__ w, int h, float (&enc)[4 * NQ]) {
    double rev[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) rev[c] = (double)mul_rn(p[c], 3.14159274101257324f) * 0.15915494309189533577;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        // combo g = 2q + h; g == 0: raw coordinates; else pairs 2(g-1), 2(g-1)+1
        const int pa0 = 2 * (2 * q - 1), pa1 = 2 * (2 * q);      // first pair for h = 0 / h = 1
        float s0, c0, s1, c1;
        if (q == 0) {
            enc_pair<L>(rev, w, 1, 0, 0, s0, c0);                   // only h = 1 lanes use these
            enc_pair<L>(rev, w, 1, 1, 1, s1, c1);
            enc[0] = h ? s0 : p[0];
            enc[1] = h ? c0 : p[1];
            enc[2] = h ? s1 : p[2];
            enc[3] = h ? c1 : 0.f;
        } else {
            enc_pair<L>(rev, w, h, pa0, pa1, s0, c0);
            enc_pair<L>(rev, w, h, pa0 + 1, pa1 + 1, s1, c1);
            enc[4 * q + 0] = s0;
            enc[4 * q + 1] = c0;
            enc[4 * q + 2] = s1;
            enc[4 * q + 3] = c1;
        }
    }
}


// d(point)/d(unit dir) from the gradient of the encoding slots and the saved encoding values:
// d/dx [w sin(f x)] = f * (w cos(f x)),  d/dx [w cos(f x)] = -f * (w sin(f x)).
// HALF: the saved encoding is a bf16 quad-row image (niw_mlp_fast.hip kHalfWorkspace): 8 bytes per (quad, sample), rows half as far apart;
// enc_row0 then points at the image's first byte for row 0 of the encoding.
template <int L, int NQ, bool HALF = false>
__device__ __forceinline__ void enc_backward(const float (&de)[4 * NQ], const float* __restrict__ enc_row0, long long mpad,
                                             unsigned qoff, int h, float (&dp)[3]) {
    dp[0] = dp[1] = dp[2] = 0.f;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        // rows 8q + 4h + {0..3} of this sample: one quad of the saved encoding (qoff = h*Mpad + m)
        f32x4 e;
        if (HALF) {
            typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
            const u32x2_t w = reinterpret_cast<const u32x2_t*>(reinterpret_cast<const char*>(enc_row0) + (long long)(8 * q) * mpad * 2)[qoff];
            e = f32x4{__builtin_bit_cast(float, w[0] << 16), __builtin_bit_cast(float, w[0] & 0xffff0000u), __builtin_bit_cast(float, w[1] << 16),
                      __builtin_bit_cast(float, w[1] & 0xffff0000u)};
        } else {
            e = reinterpret_cast<const f32x4*>(enc_row0 + (long long)(8 * q) * mpad)[qoff];
        }
        if (q == 0) {
            // half 0: raw coordinates; half 1: pairs 0 and 1
#pragma unroll
            for (int c = 0; c < 3; ++c) dp[c] += h ? 0.f : de[c];
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                const int pair = pr;
                if (pair < 3 * L) {
                    const float v = band_freq(pair % L) * (de[2 * pr] * e[2 * pr + 1] - de[2 * pr + 1] * e[2 * pr]);
                    dp[pair / L] += h ? v : 0.f;
                }
            }
        } else {
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                const int pair0 = 2 * (2 * q - 1) + pr, pair1 = 4 * q + pr;   // half 0 / half 1
                const float core = de[4 * q + 2 * pr] * e[2 * pr + 1] - de[4 * q + 2 * pr + 1] * e[2 * pr];
                if (pair0 < 3 * L) dp[pair0 / L] += h ? 0.f : band_freq(pair0 % L) * core;
                if (pair1 < 3 * L) dp[pair1 / L] += h ? band_freq(pair1 % L) * core : 0.f;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) dp[c] += __shfl_xor(dp[c], 32);
}

}  // namespace niw
