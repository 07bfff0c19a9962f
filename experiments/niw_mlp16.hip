// EXPERIMENT (round 3): the field-MLP forward with 16-sample waves, two waves per SIMD.
//
// The product kernels (niw_mlp_fwd.hip, niw_mlp_bwd.hip) give every wave 32 samples and v_mfma_f32_32x32x2_f32: 400-450 registers, ONE
// wave per SIMD, and whatever that in-order wave cannot place in its MFMA shadow (waits, epilogue tails, layer prologues) is matrix-pipe
// idle time -- ~10 % (MFMA busy 88-90 %).  Here a wave owns 16 samples and uses v_mfma_f32_16x16x4_f32 (same 64 FLOP / clk / SIMD, 32-cycle
// issue, 40-cycle dependent latency): a layer's input and output are 64 + 64 registers, the kernel fits 256, and TWO waves share a SIMD
// -- their dependent chains interleave (32 + 32 >= 40) and each covers the other's stalls.
//
// Register chaining works the same way: C/D of 16x16x4 has column m = lane & 15 on the lane and rows 4 g + r (g = lane >> 4) in the four
// registers; the B operand of 16x16x4 wants B[k = g][m] -- so accumulator register r of a 16-row block is, as it stands, the B operand of
// the MFMA that reduces over rows {4 g + r : g = 0..3} of that block.  Slot s = 16 nb + 4 g + r is simply row s.
//   A fragment (output block ob, input block nb): lane (i = lane & 15, g) holds W[16 ob + i][16 nb + 4 g + r], r = 0..3: 16 bytes, four
//   consecutive weights of one row.
#include "niw_common.h"
#include "niw_mlp_device.h"

using namespace niw;

typedef float f32x4_t __attribute__((ext_vector_type(4)));
#ifndef NIW_V16_RING
#define NIW_V16_RING 8
#endif

namespace {

__host__ __device__ constexpr int v16_ob(int l) { return l == 8 ? 8 : l == 9 ? 0 : 16; }                 // 16-row output blocks
__host__ __device__ constexpr int v16_kb(int l) { return l == 0 ? 4 : l == 4 ? 20 : l == 8 ? 18 : l == 9 ? 8 : 16; }   // 16-slot input blocks
__host__ __device__ constexpr int v16_fwd_off(int l) {       // floats
    int o = 0;
    for (int i = 0; i < l; ++i) o += v16_ob(i) * v16_kb(i) * 256;
    return o;
}
constexpr int kV16FwdFloats = v16_fwd_off(kLayers);
constexpr int kV16BiasOff = kV16FwdFloats;                    // kernel-order biases, 256 per layer
constexpr int kV16SigOff = kV16BiasOff + kLayers * 256;       // density row of layer 7 (256 input slots, natural order)
constexpr int kV16RgbOff = kV16SigOff + 256;                  // colour layer 9: [3][128]
constexpr int kV16HeadBiasOff = kV16RgbOff + 384;             // density bias, 3 colour biases
constexpr int kV16Floats = kV16HeadBiasOff + 4;

__device__ __forceinline__ int v16_source(int idx) {
    if (idx < kV16FwdFloats) {
        int l = 0;
        while (l + 1 < kLayers && idx >= v16_fwd_off(l + 1)) ++l;
        const int local = idx - v16_fwd_off(l);
        const int r = local & 3, lane = (local >> 2) & 63, frag = local >> 8;
        const int ob = 2 * ((frag >> 1) / v16_kb(l)) + (frag & 1), nb = (frag >> 1) % v16_kb(l);      // [pair][k-block][2]
        const int i = lane & 15, g = lane >> 4;
        const int row = out_row(l, 16 * ob + i), col = fwd_slot_col(l, 16 * nb + 4 * g + r);
        return (row >= 0 && col >= 0) ? weight_off(l) + row * layer_k(l) + col : -1;
    }
    if (idx < kV16SigOff) {
        const int local = idx - kV16BiasOff, l = local >> 8, n = local & 255;
        const int row = out_row(l, n);
        return (l < kLayers && row >= 0 && n < 256) ? bias_off(l) + row : -1;
    }
    if (idx < kV16RgbOff) return weight_off(7) + (idx - kV16SigOff);                        // reference row 0 of layer 7
    if (idx < kV16HeadBiasOff) return weight_off(9) + (idx - kV16RgbOff);                   // [3][128] as stored
    const int b = idx - kV16HeadBiasOff;
    return b == 0 ? bias_off(7) : bias_off(9) + (b - 1);
}

__global__ void pack16_kernel(const float* __restrict__ params, float* __restrict__ packed) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= kV16Floats) return;
    const int src = v16_source(idx);
    packed[idx] = src >= 0 ? params[src] : 0.f;
}

__device__ __forceinline__ f32x4_t mfma16(float a, float b, f32x4_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// Output blocks are produced in PAIRS (two independent accumulation chains, interleaved): a dependent v_mfma_f32_16x16x4_f32 can issue
// only every 40 cycles although the pipe is free after 32, so a single chain runs the matrix pipe at 80 % and a SIMD whose other wave is
// busy elsewhere (epilogue, prologue, waiting for a fragment) would drop to that.  With two chains every wave alone can saturate the pipe.
// Fragments lie in consumption order: [pair][k-block][2][lane][4].  The eight epilogue values of pair p - 1 are spread over pair p.
// The fragment stream and the bias blocks run on ACROSS layers (Carry): the ring keeps fetching past the end of a layer -- the next layer's
// fragments follow in the image -- and the last pair of a layer fetches the first two bias blocks of the next, so that no layer starts
// with an exposed L2 round trip.  (Every layer has a multiple of D fragments, so the ring slots line up.)
struct Carry16 {
    f32x4_t ring[NIW_V16_RING];
    f32x4_t cin0[2];
};
template <int KB1, int KB2, int NB, typename Policy>
__device__ __forceinline__ void stream_layer16(rsrc_t rsrc, int lane16, int w_base, const float (&b1)[4 * KB1], const float (&b2)[4 * (KB2 > 0 ? KB2 : 1)],
                                               Policy& pol, Carry16& carry) {
    constexpr int KB = KB1 + KB2, NP = NB / 2, N = NB * KB, D = NIW_V16_RING;
    static_assert(NB % 2 == 0 && N % D == 0, "output blocks come in pairs; whole ring turns per layer");
    f32x4_t cin[2][2], acc[2][2];
    cin[0][0] = carry.cin0[0];
    cin[0][1] = carry.cin0[1];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        pol.pre(p);
        if (p + 1 < NP) {
            pol.acc_init(2 * p + 2, cin[(p + 1) & 1][0]);
            pol.acc_init(2 * p + 3, cin[(p + 1) & 1][1]);
        } else {
            pol.acc_init(16, carry.cin0[0]);       // the next layer's bias blocks 0, 1 (256 floats per layer)
            pol.acc_init(17, carry.cin0[1]);
        }
        acc[p & 1][0] = cin[p & 1][0];
        acc[p & 1][1] = cin[p & 1][1];
#pragma unroll
        for (int q = 0; q < KB; ++q) {
            const int i = (p * KB + q) * 2;
            const f32x4_t a0 = carry.ring[i % D], a1 = carry.ring[(i + 1) % D];
            carry.ring[i % D] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane16, w_base + (i + D) * 1024, 0));
            carry.ring[(i + 1) % D] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane16, w_base + (i + 1 + D) * 1024, 0));
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float bv = q < KB1 ? b1[4 * (q < KB1 ? q : 0) + r] : b2[4 * (q >= KB1 ? q - KB1 : 0) + r];
                acc[p & 1][0] = mfma16(a0[r], bv, acc[p & 1][0]);
                acc[p & 1][1] = mfma16(a1[r], bv, acc[p & 1][1]);
            }
#pragma unroll
            for (int v = 0; v < 8; ++v) {
                if (v >= (q * 8) / KB && v < ((q + 1) * 8) / KB) {
                    pol.gap(2 * p + (v >> 2), v & 3);
                    if (p > 0) pol.epi(2 * p - 2 + (v >> 2), v & 3, acc[(p - 1) & 1][v >> 2][v & 3]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    pol.pre(NP);
#pragma unroll
    for (int v = 0; v < 8; ++v) pol.epi(NB - 2 + (v >> 2), v & 3, acc[(NP - 1) & 1][v >> 2][v & 3]);
}

// bias (kernel order, natural rows) as the initial accumulator; ReLU; hand-over; (training) quad-row store + sign bits
// HEAD 1: density row from the layer's INPUT registers (one FMA per epilogue value: 16 blocks x 4 = the lane's 64 input slots)
// HEAD 2: colour outputs from the layer's OUTPUT values
template <int NBOUT, bool SAVE, int HEAD>
struct Fwd16Epilogue {
    rsrc_t rsrc;
    int bias_bytes;                      // byte offset of the layer's kernel-order bias
    int goff;                            // g * 16 bytes
    float (&out)[4 * NBOUT];
    const float (&in)[64];
    RowWindow win;
    const char* mrec;
    int lane;
    float sig = 0.f;
    float col[3] = {0.f, 0.f, 0.f};
    f32x4_t hw[2] = {};
    f32x4_t cw[2][3] = {};
    unsigned mbits[2] = {0u, 0u};
    float keep[3] = {0.f, 0.f, 0.f};

    __device__ __forceinline__ void acc_init(int ob, f32x4_t& c) const {
        c = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsrc, goff, bias_bytes + ob * 64, 0));
    }
    __device__ __forceinline__ void pre(int p) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            if (HEAD == 1 && p < 8) hw[e] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsrc, goff, 4 * kV16SigOff + (2 * p + e) * 64, 0));
            if (HEAD == 2 && p > 0) {       // consumed by epi(2 (p - 1) + e, ..), which runs during pair p (or after the last pair: pre(NP))
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    cw[e][c] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsrc, goff, 4 * (kV16RgbOff + c * 128) + (2 * p - 2 + e) * 64, 0));
            }
        }
    }
    // called four times during block ob, whatever ob: the density row's share of input block ob
    __device__ __forceinline__ void gap(int ob, int r) {
        if (HEAD == 1) sig = fmaf(hw[ob & 1][r], in[ob * 4 + r], sig);
    }
    __device__ __forceinline__ void epi(int ob, int r, float a) {
        const float v = __builtin_bit_cast(float, max(__builtin_bit_cast(int, a), 0));
        out[ob * 4 + r] = v;
        if (HEAD == 2) {
#pragma unroll
            for (int c = 0; c < 3; ++c) col[c] = fmaf(cw[ob & 1][c][r], v, col[c]);
        }
        if (SAVE) {
            if (r == 3) buf_store4(keep[0], keep[1], keep[2], v, win.rsrc(ob * 16), win.voff4, 0);
            else keep[r] = v;
            mbits[ob >> 3] = __builtin_amdgcn_alignbit(mbits[ob >> 3], __builtin_bit_cast(unsigned, v) + 0x7fffffffu, 31);
            if (ob == NBOUT - 1 && r == 3) {
                // The dX chain (niw_mlp_bwd.hip) owns 32 samples per wave and expects, per lane (sample i, half h), dword nb / 2 with
                // bit 31 - (16 (nb & 1) + r32) for row 32 nb + 8 (r32 >> 2) + 4 h + (r32 & 3).  This lane (sample j, group g) holds rows
                // 16 ob + 4 g + r: for the dX lane (i = 16 (wave & 1) + j, h = g & 1) that is every other nibble of the record -- the
                // even nibbles on the lanes g < 2, the odd ones on their partners g + 2.  Spread the four nibbles of each 16-bit half
                // over alternate nibble slots, merge the partner's through one cross-half exchange, and let the lanes g < 2 write.
                unsigned D[4];
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const unsigned s16 = (d & 1) ? (mbits[d >> 1] & 0xffffu) : (mbits[d >> 1] >> 16);
                    unsigned x = (s16 | (s16 << 8)) & 0x00ff00ffu;
                    x = (x | (x << 4)) & 0x0f0f0f0fu;
                    D[d] = (lane & 32) ? x : (x << 4);
                }
#pragma unroll
                for (int d = 0; d < 4; ++d) {       // v_permlane32_swap: lanes l and l + 32 see each other's value (no LDS round trip)
                    const auto sw = __builtin_amdgcn_permlane32_swap(D[d], D[d], false, false);
                    D[d] = sw[0] | sw[1];
                }
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                // (the lanes g >= 2 hold the same merged words as their partners and write them to the same place)
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{D[0], D[1], D[2], D[3]}, make_rsrc(mrec), (2 * (lane & 16) + (lane & 15)) * 16, 0, 0);
            }
        }
    }
};

struct Mlp16Args {
    const float* packed;
    const float* center;
    const float* ray;
    const float* depth;
    const float* noise;
    float* rgb;
    float* sigma;
    float* save;
    long long M, Mpad;
    int S, act;
    float w3d[NIW_L3D];
    float wview[NIW_LVIEW];
    const float* band_dev;
};

__device__ __forceinline__ float density_act16(float x, int kind) {
    if (kind == NIW_ACT_RELU) return fmaxf(x, 0.f);
    return x > 20.f ? x : log1pf(expf(x));
}

// the four encoding slots 4 c .. 4 c + 3 of combo c (c = 0: the raw coordinates; else sin / cos of pairs 2 (c - 1), 2 (c - 1) + 1)
template <int L>
__device__ __forceinline__ void encode_combo(const float (&p)[3], const double (&rev)[3], const float* __restrict__ wtab, int c, float (&o)[4]) {
    if (c == 0) { o[0] = p[0]; o[1] = p[1]; o[2] = p[2]; o[3] = 0.f; return; }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int pair = 2 * (c - 1) + e;
        const bool ok = pair < 3 * L;
        const int coord = ok ? pair / L : 0, band = ok ? pair % L : 0;
        const double t = coord == 0 ? rev[0] : coord == 1 ? rev[1] : rev[2];
        float s, cs;
        sincos_band(t, band, s, cs);
        const float w = wtab[band];
        o[2 * e] = ok ? s * w : 0.f;
        o[2 * e + 1] = ok ? cs * w : 0.f;
    }
}

template <bool SAVE>
#ifndef NIW_V16_WAVES
#define NIW_V16_WAVES 4
#endif
__global__ __launch_bounds__(64 * NIW_V16_WAVES, 2) void mlp_fwd16_kernel(Mlp16Args a) {
    __shared__ float wtab[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, g = lane >> 4;
    if (threadIdx.x < NIW_L3D + NIW_LVIEW)
        wtab[threadIdx.x] = a.band_dev ? a.band_dev[threadIdx.x] : (threadIdx.x < NIW_L3D ? a.w3d[threadIdx.x] : a.wview[threadIdx.x - NIW_L3D]);
    __syncthreads();
    const long long m = ((long long)blockIdx.x * NIW_V16_WAVES + wave) * 16 + j;
    const bool valid = m < a.M;
    const long long mc = valid ? m : a.M - 1;
    const long long ri = mc / a.S;
    float p[3], u[3];
    {
        const float d = a.depth[mc];
        const float rx = a.ray[ri * 3 + 0], ry = a.ray[ri * 3 + 1], rz = a.ray[ri * 3 + 2];
        p[0] = add_rn(a.center[ri * 3 + 0], mul_rn(rx, d));
        p[1] = add_rn(a.center[ri * 3 + 1], mul_rn(ry, d));
        p[2] = add_rn(a.center[ri * 3 + 2], mul_rn(rz, d));
        const float nrm = fmaxf(sqrtf(rx * rx + ry * ry + rz * rz), 1e-12f);
        u[0] = rx / nrm; u[1] = ry / nrm; u[2] = rz / nrm;
    }
    float enc[16], venc[8];
    {
        double rev[3], vrev[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            rev[c] = (double)mul_rn(p[c], 3.14159274101257324f) * 0.15915494309189533577;
            vrev[c] = (double)mul_rn(u[c], 3.14159274101257324f) * 0.15915494309189533577;
        }
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            float o[4];
            encode_combo<NIW_L3D>(p, rev, wtab, 4 * nb + g, o);
            enc[4 * nb] = o[0]; enc[4 * nb + 1] = o[1]; enc[4 * nb + 2] = o[2]; enc[4 * nb + 3] = o[3];
        }
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            float o[4];
            encode_combo<NIW_LVIEW>(u, vrev, wtab + NIW_L3D, 4 * nb + g, o);
            venc[4 * nb] = o[0]; venc[4 * nb + 1] = o[1]; venc[4 * nb + 2] = o[2]; venc[4 * nb + 3] = o[3];
        }
    }
    const rsrc_t rsrc = make_rsrc(a.packed);
    const int lane16 = lane * 16, goff = g * 16;
    const int pitch4 = (int)(a.Mpad * 4), voff4 = (int)(((long long)g * a.Mpad + m) * 16);
    const unsigned mpad32 = (unsigned)a.Mpad;
    auto row_off = [&](int r) { return (long long)((unsigned long long)(unsigned)r * (unsigned long long)mpad32); };
    auto window = [&](int r) { return RowWindow{SAVE ? a.save + row_off(r) : nullptr, pitch4, voff4}; };
    // ReLU sign-mask records: the 1 KiB record of a 32-sample pair of waves holds this wave's 512 bytes ([lane][8 bytes]) in its half
    const long long pair_id = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * NIW_V16_WAVES + wave) >> 1));
    const int half = __builtin_amdgcn_readfirstlane(wave & 1);
    auto mask_rec = [&](int i) -> const char* {
        if (!SAVE) return nullptr;
        const unsigned long long p = reinterpret_cast<unsigned long long>(a.save + row_off(kSaveMask)) + (pair_id * kMaskRecords + i) * kMaskRecBytes + half * 256;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)p), hi = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));   // wave-uniform: says so
        return reinterpret_cast<const char*>(((unsigned long long)hi << 32) | lo);
    };
    if (SAVE) {
        const RowWindow we = window(kSaveEnc), wv = window(kSaveVenc);
#pragma unroll
        for (int q = 0; q < 4; ++q) buf_store4(enc[4 * q], enc[4 * q + 1], enc[4 * q + 2], enc[4 * q + 3], we.rsrc(16 * q), voff4, 0);
#pragma unroll
        for (int q = 0; q < 2; ++q) buf_store4(venc[4 * q], venc[4 * q + 1], venc[4 * q + 2], venc[4 * q + 3], wv.rsrc(16 * q), voff4, 0);
    }
    const float none[4] = {0.f, 0.f, 0.f, 0.f};
    float act[64], nxt[64];
    Carry16 carry;
#pragma unroll
    for (int i = 0; i < NIW_V16_RING; ++i) carry.ring[i] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane16, i * 1024, 0));
#pragma unroll
    for (int e = 0; e < 2; ++e) carry.cin0[e] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsrc, goff, 4 * kV16BiasOff + e * 64, 0));
    auto advance = [&]() {
#pragma unroll
        for (int i = 0; i < 64; ++i) act[i] = nxt[i];
    };
    {
        Fwd16Epilogue<16, SAVE, 0> ep{rsrc, 4 * (kV16BiasOff + 0 * 256), goff, nxt, act, window(save_h(1)), mask_rec(0), lane};
        stream_layer16<4, 0, 16>(rsrc, lane16, 4 * v16_fwd_off(0), enc, none, ep, carry);
        advance();
    }
#pragma unroll 1
    for (int l = 1; l <= 3; ++l) {
        Fwd16Epilogue<16, SAVE, 0> ep{rsrc, 4 * (kV16BiasOff + l * 256), goff, nxt, act, window(save_h(l + 1)), mask_rec(l), lane};
        stream_layer16<16, 0, 16>(rsrc, lane16, 4 * v16_fwd_off(1) + (l - 1) * (16 * 16 * 1024), act, none, ep, carry);
        advance();
    }
    {
        Fwd16Epilogue<16, SAVE, 0> ep{rsrc, 4 * (kV16BiasOff + 4 * 256), goff, nxt, act, window(save_h(5)), mask_rec(4), lane};
        stream_layer16<16, 4, 16>(rsrc, lane16, 4 * v16_fwd_off(4), act, enc, ep, carry);
        advance();
    }
#pragma unroll 1
    for (int l = 5; l <= 6; ++l) {
        Fwd16Epilogue<16, SAVE, 0> ep{rsrc, 4 * (kV16BiasOff + l * 256), goff, nxt, act, window(save_h(l + 1)), mask_rec(l), lane};
        stream_layer16<16, 0, 16>(rsrc, lane16, 4 * v16_fwd_off(5) + (l - 5) * (16 * 16 * 1024), act, none, ep, carry);
        advance();
    }
    {
        Fwd16Epilogue<16, SAVE, 1> ep{rsrc, 4 * (kV16BiasOff + 7 * 256), goff, nxt, act, window(kSaveFeat), mask_rec(7), lane};
        stream_layer16<16, 0, 16>(rsrc, lane16, 4 * v16_fwd_off(7), act, none, ep, carry);
        advance();
        float sig_raw = ep.sig;
        sig_raw += __shfl_xor(sig_raw, 16);
        sig_raw += __shfl_xor(sig_raw, 32);
        sig_raw += buf_load1(rsrc, 0, 4 * kV16HeadBiasOff);
        if (a.noise != nullptr) sig_raw += a.noise[mc];
        if (g == 0) {
            if (SAVE) (a.save + row_off(kSaveSigma))[m] = sig_raw;
            if (valid) a.sigma[m] = density_act16(sig_raw, a.act);
        }
    }
    float hr[32];
    {
        Fwd16Epilogue<8, SAVE, 2> ep{rsrc, 4 * (kV16BiasOff + 8 * 256), goff, hr, act, window(kSaveHr), mask_rec(8), lane};
        stream_layer16<16, 2, 8>(rsrc, lane16, 4 * v16_fwd_off(8), act, venc, ep, carry);
        float o[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = ep.col[c];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            o[c] = v + buf_load1(rsrc, 0, 4 * (kV16HeadBiasOff + 1 + c));
        }
        if (g == 0 && valid) {
#pragma unroll
            for (int c = 0; c < 3; ++c) a.rgb[m * 3 + c] = 1.f / (1.f + expf(-o[c]));
        }
    }
}

}  // namespace

extern "C" int64_t niw_mlp16_packed_floats(void) { return kV16Floats; }

extern "C" int niw_mlp16_pack_weights(const float* params, float* packed, niw_stream_t stream) {
    NIW_REQUIRE(params && packed, "niw_mlp16_pack_weights: null pointer");
    pack16_kernel<<<(kV16Floats + 255) / 256, 256, 0, (hipStream_t)stream>>>(params, packed);
    NIW_LAUNCH_CHECK("niw_mlp16_pack_weights");
    return NIW_OK;
}

extern "C" int niw_mlp16_fwd(const float* packed, const float* center, const float* ray, const float* depth, const float* noise, int64_t n_rays,
                             int n_samples, const float* band_w3d, const float* band_wview, const float* band_dev, int density_activ, float* rgb,
                             float* sigma, float* save, niw_stream_t stream) {
    NIW_REQUIRE(packed && center && ray && depth && rgb && sigma, "niw_mlp16_fwd: null pointer");
    Mlp16Args a;
    a.packed = packed; a.center = center; a.ray = ray; a.depth = depth; a.noise = noise; a.rgb = rgb; a.sigma = sigma; a.save = save;
    a.M = n_rays * (int64_t)n_samples; a.Mpad = niw_mlp_padded_rows(n_rays, n_samples);
    a.S = n_samples; a.act = density_activ;
    for (int i = 0; i < NIW_L3D; ++i) a.w3d[i] = band_w3d ? band_w3d[i] : 1.f;
    for (int i = 0; i < NIW_LVIEW; ++i) a.wview[i] = band_wview ? band_wview[i] : 1.f;
    a.band_dev = band_dev;
    const int blocks = (int)(a.Mpad / (16 * NIW_V16_WAVES));
    if (save) mlp_fwd16_kernel<true><<<blocks, 64 * NIW_V16_WAVES, 0, (hipStream_t)stream>>>(a);
    else mlp_fwd16_kernel<false><<<blocks, 64 * NIW_V16_WAVES, 0, (hipStream_t)stream>>>(a);
    NIW_LAUNCH_CHECK("niw_mlp16_fwd");
    return NIW_OK;
}
