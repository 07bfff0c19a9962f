// Field MLP forward (points -> positional encoding -> 8x256 MLP -> density, colour) and the
// weight re-packing kernel.  Replaces NeRF.forward_samples / NeRF.forward /
// NeRF.positional_encoding (reference model/nerf.py:416-456, 476-483; c2f mask
// model/barf_inn_llff.py:427-442) and camera.get_3D_points_from_depth (camera.py:517-521).
//
// Design (gfx950): every wave owns 32 samples and carries them through the whole network with
// the activations held in registers.  The GEMM of a layer is computed transposed,
//     out[n][m] = sum_k W[n][k] * act[k][m]       (A operand = W, B operand = act)
// with v_mfma_f32_32x32x2_f32 (exact fp32).  The C/D fragment of that product (column m on the
// lane, rows n in the 16 registers, rows (r&3)+8(r>>2)+4h) is, register for register, the B
// fragment the next layer needs (k-slot 8q+4h+t <-> register 4q+t), so no LDS, no transposes
// and no barriers are needed between layers.  Weights are pre-packed (niw_mlp_pack_weights)
// into A-fragment order so that every weight load is one coalesced 1 KiB wave access served
// from L2 (2.1 MB total, resident in every XCD's 4 MiB L2).
#include "niw_common.h"
#include "niw_mlp_device.h"
#include "niw_mlp_encode.h"
#include "niw_trace.h"

using namespace niw;
NIW_TRACE_SETTER(niw_trace_set_fwd)

// fast-precision modes: niw_mlp_fast.hip
extern "C" int64_t niw_mlp_fast_image_bytes(void);
extern "C" int niw_mlp_pack_weights_fast(const float* params, void* image, niw_stream_t stream);
int niw_launch_mlp_fwd_fast(int precision, const void* image, const float* center, const float* ray, const float* depth, const float* noise,
                            int64_t n_rays, int n_samples, const float* band_w3d, const float* band_wview, const float* band_dev,
                            int density_activ, float* rgb, float* sigma, float* save, hipStream_t stream);

// ---------------------------------------------------------------------------------------
// weight packing
// ---------------------------------------------------------------------------------------
// index of the parameter float that lands in packed float `idx` (-1: zero padding)
__device__ __forceinline__ int pack_source(int idx) {
    if (idx < kFwdPackFloats) {
        int l = 0;
        while (l + 1 < kLayers && idx >= fwd_pack_off(l + 1)) ++l;
        int local = idx - fwd_pack_off(l);
        int t = local & 3, lane = (local >> 2) & 63, blk = local >> 8;
        int nb = blk % fwd_nb(l), q = blk / fwd_nb(l);
        int i = lane & 31, h = lane >> 5;
        int row = out_row(l, nb * 32 + i), col = fwd_slot_col(l, 8 * q + 4 * h + t);
        return (row >= 0 && col >= 0) ? weight_off(l) + row * layer_k(l) + col : -1;
    }
    if (idx < kBiasPackOff) {
        int l = 0;
        while (l + 1 < kLayers && idx >= bwd_pack_off(l + 1)) ++l;
        int local = idx - bwd_pack_off(l);
        int t = local & 3, lane = (local >> 2) & 63, blk = local >> 8;
        int ob = blk % bwd_ob(l), rb = blk / bwd_ob(l);
        int i = lane & 31, h = lane >> 5;
        int row = out_row(l, rb * 8 + 4 * h + t), col = fwd_slot_col(l, ob * 32 + i);
        return (row >= 0 && col >= 0) ? weight_off(l) + row * layer_k(l) + col : -1;
    }
    if (idx < kHeadSigOff) {
        int l = 0;
        while (l + 1 < kLayers && idx >= bias_pack_off(l + 1)) ++l;
        int local = idx - bias_pack_off(l);
        int r = local & 15, h = (local >> 4) & 1, nb = local >> 5;
        int row = out_row(l, nb * 32 + acc_row(r, h));
        return row >= 0 ? bias_off(l) + row : -1;
    }
    if (idx < kHeadRgbOff) {             // density row of layer 7 (reference row 0): register 4q+t of half h <-> input slot 8q+4h+t
        int local = idx - kHeadSigOff, h = local >> 7, reg = local & 127;
        return weight_off(7) + 8 * (reg >> 2) + 4 * h + (reg & 3);
    }
    if (idx < kHeadBiasOff) {            // colour layer 9: register nb*16+r of half h <-> input row nb*32 + acc_row(r, h)
        int local = idx - kHeadRgbOff, c = local >> 7, h = (local >> 6) & 1, reg = local & 63;
        return weight_off(9) + c * 128 + (reg >> 4) * 32 + acc_row(reg & 15, h);
    }
    int b = idx - kHeadBiasOff;
    return b == 0 ? bias_off(7) : bias_off(9) + (b - 1);
}

__global__ void pack_weights_kernel(const float* __restrict__ params, float* __restrict__ packed) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= kPackedFloats) return;
    const int src = pack_source(idx);
    packed[idx] = src >= 0 ? params[src] : 0.f;
}

__global__ void pack_index_kernel(int* __restrict__ index) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < kPackedFloats) index[idx] = pack_source(idx);
}

// four packed floats per thread: 16-byte index load, four L2-resident gathers, 16-byte store
__global__ void pack_gather_kernel(const float* __restrict__ params, const int4* __restrict__ index, f32x4* __restrict__ packed) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= kPackedFloats / 4) return;
    const int4 s = index[i];
    packed[i] = f32x4{s.x >= 0 ? params[s.x] : 0.f, s.y >= 0 ? params[s.y] : 0.f, s.z >= 0 ? params[s.z] : 0.f, s.w >= 0 ? params[s.w] : 0.f};
}
static_assert(kPackedFloats % 4 == 0, "packed image is a whole number of float4");

// positional encoding in MFMA slot order: niw_mlp_encode.h (shared with the fast-precision forward)

// Epilogue policy of one forward layer for stream_layer(): bias add, ReLU, hand the value to the next
// layer's operand registers and (training) store it feature-major [row][Mpad].  Row of (nb, r, h) =
// nb*32 + (r&3) + 8(r>>2) + 4h; the lane-dependent part (4h rows + sample m) is folded into ONE 32-bit
// element offset `voff`, so every store is scalar-base + vector-offset.
// HEAD 1 (layer 7): the density row as a dot product of the layer's INPUT registers with the head weights, one FMA per epilogue
// call (128 calls = the 128 input registers of this lane; the two lane halves hold complementary slots and are added at the end).
// HEAD 2 (layer 8): the three colour outputs accumulated from the layer's OUTPUT values as they are produced.
template <int NBOUT, bool RELU, bool SAVE, int HEAD = 0>
struct FwdEpilogue {
    const PackedWeights& pw;
    int bias_bytes;                      // byte offset of the layer's packed bias ([row block][half][16])
    int hoff;                            // this lane half's offset inside it: h * 64
    float (&out)[16 * NBOUT];
    const float (&in)[128];              // the layer's input registers (HEAD 1 reads them; others pass them along unused)
    RowWindow win;                       // where the layer's output rows live in the activation workspace
    float sig_raw;                       // HEAD 1: this lane half's partial density dot product
    const char* mrec = nullptr;          // this wave's ReLU sign-mask record of the layer (wave-uniform; training only)
    int lane = 0;
    unsigned mbits[4] = {0u, 0u, 0u, 0u};   // this lane's sign bits: dword nb/2, bit 31 - (16*(nb&1) + r)  (pushed LSB-first)
    float hw[HEAD == 1 ? 2 : 1][16] = {};   // HEAD 1: head weights of the current / next row block's 16 input registers
    float cw[HEAD == 2 ? 192 : 1] = {};  // HEAD 2: colour-layer weights [channel][64] of this lane half
    float col[3] = {0.f, 0.f, 0.f};      // HEAD 2: this lane half's partial colour outputs

    static constexpr bool kAccInit = true;
    // the block's accumulation starts from its bias fragment (this lane half's 16 rows)
    __device__ __forceinline__ void acc_init(int nb, f32x16& c) const {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 v = buf_load4(pw.rsrc, hoff, bias_bytes + nb * 128 + g * 16);
            c[4 * g] = v[0]; c[4 * g + 1] = v[1]; c[4 * g + 2] = v[2]; c[4 * g + 3] = v[3];
        }
    }
    __device__ __forceinline__ void pre(int nb, float (&)[16]) { pre_head(nb); }
    // head weights: HEAD 1 fetches the 16 weights that epi(nb, .) multiplies, one row block ahead like the biases; HEAD 2 fetches
    // all 192 at the start of the layer, behind the ring's first fragments (they are first used a whole row block later)
    __device__ __forceinline__ void pre_head(int nb) {
        if (HEAD == 1) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = buf_load4(pw.rsrc, hoff * 8, 4 * kHeadSigOff + nb * 64 + g * 16);      // hoff * 8 = h * 512 bytes
                hw[nb & 1][4 * g] = v[0]; hw[nb & 1][4 * g + 1] = v[1]; hw[nb & 1][4 * g + 2] = v[2]; hw[nb & 1][4 * g + 3] = v[3];
            }
        }
        if (HEAD == 2 && nb == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const f32x4 v = buf_load4(pw.rsrc, hoff * 4, 4 * kHeadRgbOff + c * 512 + g * 16);  // hoff * 4 = h * 256 bytes
                    cw[c * 64 + 4 * g] = v[0]; cw[c * 64 + 4 * g + 1] = v[1]; cw[c * 64 + 4 * g + 2] = v[2]; cw[c * 64 + 4 * g + 3] = v[3];
                }
        }
    }
    __device__ __forceinline__ void epi(int nb, int r, float a, float p) {
        if (nb < NBOUT) {
            // the bias is already in the accumulator (acc_init).  ReLU as a signed-integer max of the bit pattern: positive floats
            // are positive integers, negative ones (and -0.0) negative -- one v_max_i32, where fmaxf on a raw MFMA result costs a
            // second v_max_f32 to canonicalise its operand
            const float v = RELU ? __builtin_bit_cast(float, max(__builtin_bit_cast(int, a), 0)) : a;
            out[nb * 16 + r] = v;
            if (HEAD == 1) sig_raw = fmaf(hw[nb & 1][r], in[nb * 16 + r], sig_raw);
            if (HEAD == 2) {
#pragma unroll
                for (int c = 0; c < 3; ++c) col[c] = fmaf(cw[c * 64 + nb * 16 + r], v, col[c]);
            }
            if (SAVE && (r & 3) == 3) {    // registers r-3 .. r = four consecutive rows: one 16-byte store into the quad-row image
                buf_store4(out[nb * 16 + r - 3], out[nb * 16 + r - 2], out[nb * 16 + r - 1], v, win.rsrc(nb * 32), win.voff4, 8 * (r >> 2) * win.pitch4);
#ifndef NIW_FWD_KEEP_HAZARD      // (diagnostic build: make VARIANT=hazard EXTRA=-DNIW_FWD_KEEP_HAZARD -- the pairs as rounds 3-5 shipped them)
                if (HEAD == 2) {
                    // The colour layer's hidden values die with this store, so the compiler reuses their registers at once -- for the mask
                    // arithmetic below (`bits(v) + 0x7fffffff`) -- and a vector write of a buffer_store_dwordx4's data registers in the very
                    // next instruction reaches memory INSTEAD of the data whenever another wave shares the SIMD (the gfx950 store-data
                    // hazard; LLVM pads only immediate soffsets).  Rounds 3-5 relied on this kernel's one wave per SIMD; round 6 measured
                    // that a wave of ANOTHER kernel -- the library's own second stream, another process on the device -- triggers it just
                    // the same (tools/store_war_hazard_foreign.hip: 2.2e5 of 5.4e8 stores; a saved +0.0 then reads 0x7fffffff = NaN, and one
                    // did: a rank diverged to NaN while four processes shared a GPU).  One s_nop 0 between the two was enough in every measurement (0 of 5.4e8)
                    // and free beside an MFMA chain; the scheduling barriers keep it where it is.
                    __builtin_amdgcn_sched_barrier(0);
                    asm volatile("s_nop 1");      // two wait states: what LLVM itself puts behind the store forms it knows (global stores need both: tools/store_war_hazard_global.hip)
                    __builtin_amdgcn_sched_barrier(0);
                }
#endif
            }
            if (SAVE && RELU) {
                // v = max(x, 0) is +0.0 or positive: v > 0  <=>  bits(v) + 0x7fffffff carries into bit 31.  One add and one
                // funnel shift per value ((mbits << 1) | bit 31 of the sum); the compare / select / or form cost three.
                mbits[nb >> 1] = __builtin_amdgcn_alignbit(mbits[nb >> 1], __builtin_bit_cast(unsigned, v) + 0x7fffffffu, 31);
                if (nb == NBOUT - 1 && r == 15) {     // one coalesced 16 B/lane store per layer: the wave's 1 KiB record
                    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4{mbits[0], mbits[1], mbits[2], mbits[3]}, make_rsrc(mrec), lane * 16, 0, 0);
                }
            }
        }
    }
};

struct MlpFwdArgs {
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
    const float* band_dev;      // device copy of {w3d, wview} read at run time (HIP-graph replays), or NULL: the by-value copies above
};

__device__ __forceinline__ float density_act(float x, int kind) {
    if (kind == NIW_ACT_RELU) return fmaxf(x, 0.f);
    return x > 20.f ? x : log1pf(expf(x));      // F.softplus(beta=1, threshold=20)
}

template <bool SAVE>
__global__ __launch_bounds__(256, 1) void mlp_fwd_kernel(MlpFwdArgs a) {
    NIW_STAMP(0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const long long m = ((long long)blockIdx.x * 4 + wave) * 32 + j;
    const bool valid = m < a.M;
    const long long mc = valid ? m : a.M - 1;
    const long long ri = mc / a.S;

    // ---- sample point and unit view direction (camera.py:517-521, nerf.py:452)
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
    NIW_STAMP(11);                                   // inputs of the sample are in registers; the encodings follow
    float enc[32], venc[16];
    {
        float w3[NIW_L3D], wv[NIW_LVIEW];
#pragma unroll
        for (int i = 0; i < NIW_L3D; ++i) w3[i] = a.band_dev ? a.band_dev[i] : a.w3d[i];
#pragma unroll
        for (int i = 0; i < NIW_LVIEW; ++i) wv[i] = a.band_dev ? a.band_dev[NIW_L3D + i] : a.wview[i];
        encode_slots<NIW_L3D, 8>(p, w3, h, enc);
        encode_slots<NIW_LVIEW, 4>(u, wv, h, venc);
    }
    NIW_STAMP(12);                                   // encodings in registers; their stores (training) follow
    // Workspace layout: the quad-row image of niw_mlp_device.h ([row / 4][Mpad][4]; rows >= kSaveSigma -- raw density and the mask
    // records -- stay plain [row][Mpad]).  (A blocked [128-sample block][row][128] image was measured 10-14 % slower for this
    // kernel and the dX chain on MI355X.)
    // All hot-loop memory traffic uses buffer addressing (see niw_mlp_device.h): host guarantees 128*Mpad < 2^31.
    const PackedWeights pw = packed_weights(a.packed, lane);
    const int pitch4 = (int)(a.Mpad * 4), voff4 = (int)(((long long)h * a.Mpad + m) * 16), hoff = h * 64;
    // row * Mpad as a 32 x 32 -> 64-bit product: it stays on the scalar ALU.  (As a 64 x 64-bit product of the kernel argument with
    // the layer loop's row index it was formed by v_mad_u64_u32, the window bases lived in VGPRs and every store that used a
    // descriptor built from them became a readfirstlane "waterfall" loop: 66 loops in the training kernel.)
    const unsigned mpad32 = (unsigned)a.Mpad;
    auto row_off = [&](int r) { return (long long)((unsigned long long)(unsigned)r * (unsigned long long)mpad32); };
    auto window = [&](int r) { return RowWindow{SAVE ? a.save + row_off(r) : nullptr, pitch4, voff4}; };
    // ReLU sign-mask records of this wave (niw_common.h kSaveMask): record i = output of layer i (0..6), 7 = feat, 8 = hr
    const long long wave_id = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + wave));
    auto mask_rec = [&](int i) {
        return SAVE ? reinterpret_cast<const char*>(a.save + row_off(kSaveMask)) + (wave_id * kMaskRecords + i) * kMaskRecBytes : nullptr;
    };
    if (SAVE) {
        const RowWindow we = window(kSaveEnc), wv = window(kSaveVenc);
#pragma unroll
        for (int q = 0; q < 8; ++q) buf_store4(enc[4 * q], enc[4 * q + 1], enc[4 * q + 2], enc[4 * q + 3], we.rsrc(0), voff4, 8 * q * pitch4);
#pragma unroll
        for (int q = 0; q < 4; ++q) buf_store4(venc[4 * q], venc[4 * q + 1], venc[4 * q + 2], venc[4 * q + 3], wv.rsrc(0), voff4, 8 * q * pitch4);
    }

    const f32x4* wp = reinterpret_cast<const f32x4*>(a.packed);
    const float none[4] = {0.f, 0.f, 0.f, 0.f};
    // The 128 activation registers of a layer's input and of its output alternate between two arrays (xa -> xb -> xa ...) instead of
    // being copied back after every layer (128 vector moves each, paid in matrix-pipe time: tools/mfma_valu_contention.hip):
    //   0: enc -> xa | 1: xa -> xb | 2: xb -> xa | 3: xa -> xb | 4: xb (+enc) -> xa | 5: xa -> xb | 6: xb -> xa | 7: xa -> xb | 8: xb (+venc) -> hr
    // The loop below runs (xa -> xb, xb -> xa) three times; the second "xb -> xa" is layer 4 with its skip connection.
    float xa[128], xb[128];

    // every layer leaves the next one its first weight fragments and its first bias fragment (niw_mlp_device.h LayerCarry)
    LayerCarry carry;
    constexpr int kLayerBytes = 32 * 8 * 1024;       // one 256 -> 256 layer of the packed image
    auto w_bytes = [&](int l) {                       // byte offset of layer l's fragments (l = 1..8), scalar arithmetic
        return l <= 3 ? 4 * fwd_pack_off(1) + (l - 1) * kLayerBytes : l == 4 ? 4 * fwd_pack_off(4)
             : l <= 6 ? 4 * fwd_pack_off(5) + (l - 5) * kLayerBytes : l == 7 ? 4 * fwd_pack_off(7) : 4 * fwd_pack_off(8);
    };
    auto b_bytes = [&](int l) { return 4 * (bias_pack_off(0) + l * 256); };      // 8 row blocks x 32 floats per layer up to layer 8
    static_assert(bias_pack_off(7) == bias_pack_off(0) + 7 * 256 && bias_pack_off(8) == bias_pack_off(0) + 8 * 256, "uniform bias stride");
    auto next_of = [&](int l) { return NextLayer{w_bytes(l + 1), (l + 1 == 8 ? 4 : 8) * 1024, b_bytes(l + 1), hoff}; };
    auto plain = [&](int l, const float (&in)[128], float (&out)[128]) {          // layers 1, 2, 3, 5, 6
        FwdEpilogue<8, true, SAVE> ep{pw, b_bytes(l), hoff, out, in, window(save_h(l + 1)), 0.f, mask_rec(l), lane};
        stream_layer<32, 0, 8, 8, decltype(ep), true, true>(pw, wp + w_bytes(l) / 16, in, none, ep, &carry, next_of(l));
    };
    // ---- layer 0: 63 -> 256
    NIW_STAMP(1);                                    // encodings done: the first matrix instruction follows
    {
        FwdEpilogue<8, true, SAVE> ep{pw, b_bytes(0), hoff, xa, xb, window(save_h(1)), 0.f, mask_rec(0), lane};
        stream_layer<8, 0, 8, 8, decltype(ep), false, true>(pw, wp + fwd_pack_off(0) / 4, enc, none, ep, &carry, next_of(0));
    }
    NIW_STAMP(2);
#pragma unroll 1
    for (int k = 0; k < 3; ++k) {
        plain(2 * k + 1, xa, xb);                    // layers 1, 3, 5
        NIW_STAMP(3 + 2 * k);
        if (k != 1) {
            plain(2 * k + 2, xb, xa);                // layers 2, 6
        } else {
            // ---- layer 4: cat[feat, points_enc] (319) -> 256
            FwdEpilogue<8, true, SAVE> ep{pw, b_bytes(4), hoff, xa, xb, window(save_h(5)), 0.f, mask_rec(4), lane};
            stream_layer<32, 8, 8, 8, decltype(ep), true, true>(pw, wp + fwd_pack_off(4) / 4, xb, enc, ep, &carry, next_of(4));
        }
        NIW_STAMP(4 + 2 * k);
    }
    // ---- layer 7: 256 -> 256 features (+ density row 256 = row block 8)
    {
        FwdEpilogue<8, true, SAVE, 1> ep{pw, b_bytes(7), hoff, xb, xa, window(kSaveFeat), 0.f, mask_rec(7), lane};
        stream_layer<32, 0, 8, 8, decltype(ep), true, true>(pw, wp + fwd_pack_off(7) / 4, xa, none, ep, &carry, next_of(7));
        // density row: the two lane halves hold complementary input slots
        float sig_raw = ep.sig_raw + __shfl_xor(ep.sig_raw, 32) + buf_load1(pw.rsrc, 0, 4 * kHeadBiasOff);
        if (a.noise != nullptr) sig_raw += a.noise[mc];
        if (h == 0) {
            if (SAVE) (a.save + row_off(kSaveSigma))[m] = sig_raw;
            if (valid) a.sigma[m] = density_act(sig_raw, a.act);
        }
    }
    NIW_STAMP(9);
    // ---- colour layer 0: cat[feat, view_enc] (283) -> 128
    // ---- colour layer 1 (128 -> 3, sigmoid) rides on layer 0's epilogue: three FMAs per hidden value as it is produced
    float hr[64];
    {
        FwdEpilogue<4, true, SAVE, 2> ep{pw, b_bytes(8), hoff, hr, xb, window(kSaveHr), 0.f, mask_rec(8), lane};
        stream_layer<32, 4, 4, 4, decltype(ep), true, false>(pw, wp + fwd_pack_off(8) / 4, xb, venc, ep, &carry);
        float o[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) o[c] = ep.col[c] + __shfl_xor(ep.col[c], 32) + buf_load1(pw.rsrc, 0, 4 * (kHeadBiasOff + 1 + c));
        if (h == 0 && valid) {
#pragma unroll
            for (int c = 0; c < 3; ++c) a.rgb[m * 3 + c] = 1.f / (1.f + expf(-o[c]));
        }
    }
    NIW_STAMP_LAST(10);
}

// ---------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------
extern "C" int64_t niw_mlp_padded_rows(int64_t n_rays, int n_samples) {
    int64_t m = n_rays * (int64_t)n_samples;
    return (m + 127) / 128 * 128;
}

extern "C" int64_t niw_mlp_packed_floats(void) { return kPackedFloats; }

extern "C" int niw_mlp_pack_weights(const float* params, float* packed, niw_stream_t stream);

extern "C" int64_t niw_mlp_packed_bytes(int precision) {
    return precision == NIW_PREC_FP32 ? 4ll * kPackedFloats : niw_mlp_fast_image_bytes();
}

extern "C" int niw_mlp_pack_weights_prec(const float* params, int precision, void* packed, niw_stream_t stream) {
    if (precision == NIW_PREC_FP32) return niw_mlp_pack_weights(params, reinterpret_cast<float*>(packed), stream);
    NIW_REQUIRE(precision == NIW_PREC_BF16X3 || precision == NIW_PREC_BF16, "niw_mlp_pack_weights_prec: unknown precision %d", precision);
    return niw_mlp_pack_weights_fast(params, packed, stream);
}

extern "C" int niw_mlp_pack_weights(const float* params, float* packed, niw_stream_t stream) {
    NIW_REQUIRE(params && packed, "niw_mlp_pack_weights: null pointer");
    const int threads = 256, blocks = (kPackedFloats + threads - 1) / threads;
    pack_weights_kernel<<<blocks, threads, 0, (hipStream_t)stream>>>(params, packed);
    NIW_LAUNCH_CHECK("niw_mlp_pack_weights");
    return NIW_OK;
}

extern "C" int niw_mlp_pack_index(int32_t* index, niw_stream_t stream) {
    NIW_REQUIRE(index, "niw_mlp_pack_index: null pointer");
    pack_index_kernel<<<(kPackedFloats + 255) / 256, 256, 0, (hipStream_t)stream>>>(index);
    NIW_LAUNCH_CHECK("niw_mlp_pack_index");
    return NIW_OK;
}

extern "C" int niw_mlp_pack_weights_indexed(const float* params, const int32_t* index, float* packed, niw_stream_t stream) {
    NIW_REQUIRE(params && index && packed, "niw_mlp_pack_weights_indexed: null pointer");
    pack_gather_kernel<<<(kPackedFloats / 4 + 255) / 256, 256, 0, (hipStream_t)stream>>>(params, reinterpret_cast<const int4*>(index),
                                                                                            reinterpret_cast<f32x4*>(packed));
    NIW_LAUNCH_CHECK("niw_mlp_pack_weights_indexed");
    return NIW_OK;
}

extern "C" int niw_mlp_fwd(const float* packed, const float* center, const float* ray,
                           const float* depth, const float* noise, int64_t n_rays, int n_samples,
                           const float* band_w3d, const float* band_wview, const float* band_dev, int density_activ, int precision,
                           float* rgb, float* sigma, float* save, niw_stream_t stream) {
    NIW_REQUIRE(packed && center && ray && depth && rgb && sigma, "niw_mlp_fwd: null pointer");
    NIW_REQUIRE(precision == NIW_PREC_FP32 || precision == NIW_PREC_BF16X3 || precision == NIW_PREC_BF16, "niw_mlp_fwd: unknown precision %d", precision);
    NIW_REQUIRE(n_rays > 0 && n_samples > 0, "niw_mlp_fwd: n_rays=%lld n_samples=%d must be positive", (long long)n_rays, n_samples);
    NIW_REQUIRE(niw_mlp_padded_rows(n_rays, n_samples) < (1ll << 24), "niw_mlp_fwd: too many samples per call (%lld)", (long long)(n_rays * n_samples));
    NIW_REQUIRE(density_activ == NIW_ACT_RELU || density_activ == NIW_ACT_SOFTPLUS, "niw_mlp_fwd: unknown density activation %d", density_activ);
    if (precision != NIW_PREC_FP32)
        return niw_launch_mlp_fwd_fast(precision, packed, center, ray, depth, noise, n_rays, n_samples, band_w3d, band_wview, band_dev, density_activ,
                                       rgb, sigma, save, (hipStream_t)stream);
    MlpFwdArgs a;
    a.packed = packed; a.center = center; a.ray = ray; a.depth = depth; a.noise = noise;
    a.rgb = rgb; a.sigma = sigma; a.save = save;
    a.M = n_rays * (int64_t)n_samples; a.Mpad = niw_mlp_padded_rows(n_rays, n_samples);
    a.S = n_samples; a.act = density_activ;
    for (int i = 0; i < NIW_L3D; ++i) a.w3d[i] = band_w3d ? band_w3d[i] : 1.f;
    for (int i = 0; i < NIW_LVIEW; ++i) a.wview[i] = band_wview ? band_wview[i] : 1.f;
    a.band_dev = band_dev;
    const int blocks = (int)(a.Mpad / 128);
    if (save)
        mlp_fwd_kernel<true><<<blocks, 256, 0, (hipStream_t)stream>>>(a);
    else
        mlp_fwd_kernel<false><<<blocks, 256, 0, (hipStream_t)stream>>>(a);
    NIW_LAUNCH_CHECK("niw_mlp_fwd");
    return NIW_OK;
}
