// Field MLP backward, pass 1: the dX chain (autograd of reference model/nerf.py:416-456).
// Same register-chained structure as the forward: every wave owns 32 samples; for a layer
//     dX[k][m] = sum_n W[n][k] * dY[n][m]        (A operand = W^T fragments, B operand = dY)
// whose C/D fragment, after the ReLU mask (read from the activations the forward saved), is
// register for register the B operand of the next (earlier) layer.  Every dY is also stored
// feature-major for pass 2 (niw_dw_gemm.hip: dW = dY . X^T over all samples).
// The chain ends in d(points), d(view dirs) -> d_center / d_ray (gradient routes (i) and (ii)
// of SURVEY section 8a; route (iii), the ray length, is niw_composite_bwd's).
#include "niw_common.h"
#include "niw_mlp_device.h"
#include "niw_mlp_encode.h"
#include "niw_trace.h"

using namespace niw;
NIW_TRACE_SETTER(niw_trace_set_bwd)

struct MlpBwdArgs {
    const float* packed;
    const float* center;
    const float* ray;
    const float* depth;
    const float* rgb;
    const float* d_rgb;
    const float* d_sigma;
    const float* save;
    float* grad;
    float* d_center;
    float* d_ray;
    long long M, Mpad;
    int S, act, ray_grad;
    int zero;          // always 0: added to the bit positions of the mask decode so that they are not compile-time constants (MaskEpilogue)
};

// Epilogue policies for stream_layer() (niw_mlp_device.h); all workspace traffic uses buffer addressing.
// ReLU mask + hand-over + store.  The masks are the sign bits the forward recorded (niw_common.h kSaveMask):
// one 1 KiB record per layer and wave, fetched as ONE 16 B/lane load a whole layer ahead, so that nothing with
// HBM latency sits in the in-order vmcnt queue in front of the weight-fragment ring (re-reading the saved fp32
// activations for their signs did: 16 loads per row block, 6.8 GB per launch).  The lane's 16 bytes hold its own
// sign bits: dword nb/2, bit 31 - (16*(nb&1) + r), zeroed for padding samples.  dY is stored feature-major for the dW pass.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int NBOUT>
struct MaskEpilogue {
    u32x4 mk;                            // this lane's 16 bytes of the mask record of the layer input
    int zero;                            // 0 at run time
    float (&out)[16 * NBOUT];
    RowWindow grad;                      // where dY of the producing layer is stored
    static constexpr bool kAccInit = false;
    __device__ __forceinline__ void acc_init(int, f32x16&) const {}
    __device__ __forceinline__ void pre(int, float (&)[16]) const {}
    __device__ __forceinline__ void epi(int nb, int r, float a, float) {
        // sign-extended 1-bit field = all-ones / zero, then one AND on the float's bits: v_bfe_i32 + v_and_b32, two VALU instructions.
        // The bit position carries a run-time zero (a kernel argument; the sum stays on the scalar ALU): with a constant position
        // LLVM rewrites the pair as test-bit / compare / select -- three instructions plus the wait states of the compare's VCC --
        // and any inline asm that would pin the pair stops the unrolling of the layer loops.  VALU instructions are what this kernel
        // pays for: each one issued between the MFMAs of the fp32 chain costs the matrix pipe 4-6 cycles
        // (tools/mfma_valu_contention.hip).
        const int keep = __builtin_amdgcn_sbfe((int)mk[nb >> 1], 31 - (16 * (nb & 1) + r) + zero, 1);
        const float g = __builtin_bit_cast(float, __builtin_bit_cast(int, a) & keep);
        out[nb * 16 + r] = g;
        if ((r & 3) == 3)                  // four consecutive rows of this lane: one 16-byte store (quad-row image, niw_mlp_device.h)
            buf_store4(out[nb * 16 + r - 3], out[nb * 16 + r - 2], out[nb * 16 + r - 1], g, grad.rsrc(nb * 32), grad.voff4, 8 * (r >> 2) * grad.pitch4);
    }
};
// park a result in the workspace (d encoding slots of the skip connection, d view-encoding slots)
struct StashEpilogue {
    RowWindow win;
    float q[3] = {0.f, 0.f, 0.f};
    static constexpr bool kAccInit = false;
    __device__ __forceinline__ void acc_init(int, f32x16&) const {}
    __device__ __forceinline__ void pre(int, float (&)[16]) const {}
    __device__ __forceinline__ void epi(int nb, int r, float a, float) {
        if ((r & 3) == 3) buf_store4(q[0], q[1], q[2], a, win.rsrc(nb * 32), win.voff4, 8 * (r >> 2) * win.pitch4);
        else q[r & 3] = a;
    }
};
// add a parked result and keep the sum in registers
template <int NBOUT>
struct AddStashEpilogue {
    RowWindow win;
    float (&out)[16 * NBOUT];
    static constexpr bool kAccInit = false;
    __device__ __forceinline__ void acc_init(int, f32x16&) const {}
    __device__ __forceinline__ void pre(int nb, float (&buf)[16]) const {
        const rsrc_t r0 = win.rsrc(nb * 32);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = buf_load4(r0, win.voff4, 8 * q * win.pitch4);
            buf[4 * q] = v[0]; buf[4 * q + 1] = v[1]; buf[4 * q + 2] = v[2]; buf[4 * q + 3] = v[3];
        }
    }
    __device__ __forceinline__ void epi(int nb, int r, float a, float p) { out[nb * 16 + r] = a + p; }
};

// enc_backward(): niw_mlp_encode.h (shared with the fast-precision dX chain)

__global__ __launch_bounds__(256, 1) void mlp_bwd_dx_kernel(MlpBwdArgs a) {
    NIW_STAMP(0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const long long m = ((long long)blockIdx.x * 4 + wave) * 32 + j;
    const bool valid = m < a.M;
    const long long mc = valid ? m : a.M - 1;
    const unsigned qoff = (unsigned)((long long)h * a.Mpad + m);     // quad index of rows R + 4h .. R + 4h + 3 (R % 8 == 0) of this sample,
                                                                      // counted from row R of a quad-row image (niw_mlp_device.h)
    const f32x4* wp = reinterpret_cast<const f32x4*>(a.packed);
    const long long P = a.Mpad;
    const int zero = a.zero;
    const PackedWeights pw = packed_weights(a.packed, lane);
    const int pitch4 = (int)(P * 4), voff4 = (int)(((long long)h * P + m) * 16);
    auto gwin = [&](int r) { return RowWindow{a.grad + (long long)r * P, pitch4, voff4}; };     // gradient rows
    const float none[4] = {0.f, 0.f, 0.f, 0.f};
    // ReLU sign-mask records of this wave: record i = output of layer i (0..6), 7 = feat, 8 = hr; each is loaded while
    // the layer before its consumer runs
    const long long wave_id = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + wave));
    const char* mbase = reinterpret_cast<const char*>(a.save + (long long)kSaveMask * P) + wave_id * kMaskRecords * kMaskRecBytes;
    auto mask_rec = [&](int i) {
        const u32x4 v = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(make_rsrc(mbase), lane * 16, i * kMaskRecBytes, 0));
        return valid ? v : u32x4{0u, 0u, 0u, 0u};
    };
    u32x4 mk_cur = mask_rec(8), mk_nxt = mask_rec(7);

    float dy[128], nxt[128];
    auto advance = [&]() {
#pragma unroll
        for (int i = 0; i < 128; ++i) dy[i] = nxt[i];
    };

    // ---- colour head: sigmoid' and W_rgb1^T
    float dy9[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        float g = 0.f;
        if (h == 0 && t < 3 && valid) {
            const float o = a.rgb[mc * 3 + t];
            g = a.d_rgb[mc * 3 + t] * o * (1.f - o);
        }
        dy9[t] = g;
    }
    reinterpret_cast<f32x4*>(a.grad + (long long)kGradRgb1 * P)[qoff] = f32x4{dy9[0], dy9[1], dy9[2], dy9[3]};   // rows 4h + t of the 8-row slot block
    float dyr[64];
    NIW_STAMP(1);                                    // colour head done: the first matrix instruction follows
    {
        MaskEpilogue<4> ep{mk_cur, zero, dyr, gwin(kGradRgb0)};
        stream_layer<1, 0, 4, 4>(pw, wp + bwd_pack_off(9) / 4, dy9, none, ep);
    }
    // ---- colour layer 0 transposed: 128 -> 256 features (+ 32 view-encoding slots = row block 8 of 9)
    if (a.ray_grad) {
        StashEpilogue ep{gwin(kGradStashVenc)};
        stream_layer<16, 0, 1, 9>(pw, wp + bwd_pack_off(8) / 4 + 8 * 64, dyr, none, ep);
    }
    {
        mk_cur = mk_nxt; mk_nxt = mask_rec(6);
        MaskEpilogue<8> ep{mk_cur, zero, dy, gwin(kGradY7)};
        stream_layer<16, 0, 8, 9>(pw, wp + bwd_pack_off(8) / 4, dyr, none, ep);
    }
    NIW_STAMP(2);                                    // colour layers transposed
    // ---- density head: d sigma_raw (kernel row 256 of layer 7)
    float dsig[4] = {0.f, 0.f, 0.f, 0.f};
    {
        float g = 0.f;
        if (h == 0 && valid) {
            const float raw = (a.save + (long long)kSaveSigma * P)[m];          // plain row (h == 0 here)
            const float dact = a.act == NIW_ACT_RELU ? (raw > 0.f ? 1.f : 0.f) : (raw > 20.f ? 1.f : 1.f / (1.f + expf(-raw)));
            g = a.d_sigma[mc] * dact;
        }
        dsig[0] = g;
        reinterpret_cast<f32x4*>(a.grad + (long long)(kGradY7 + 256) * P)[qoff] = f32x4{dsig[0], dsig[1], dsig[2], dsig[3]};
    }
    // ---- layer 7 transposed (257 -> 256), mask with h7
    {
        mk_cur = mk_nxt; mk_nxt = mask_rec(5);
        MaskEpilogue<8> ep{mk_cur, zero, nxt, gwin(6 * 256)};
        stream_layer<32, 1, 8, 8>(pw, wp + bwd_pack_off(7) / 4, dy, dsig, ep);
        advance();
    }
    NIW_STAMP(3);
    // ---- layers 6, 5 transposed: produce dY5, dY4
#pragma unroll 1
    for (int l = 6; l >= 5; --l) {
        mk_cur = mk_nxt; mk_nxt = mask_rec(l - 2);       // this layer masks with record l-1 (output of layer l-1)
        MaskEpilogue<8> ep{mk_cur, zero, nxt, gwin((l - 1) * 256)};
        stream_layer<32, 0, 8, 8>(pw, wp + bwd_pack_off(5) / 4 + (l - 5) * (32 * 8 * 64), dy, none, ep);
        advance();
        NIW_STAMP(10 - l);                           // 4: layer 6, 5: layer 5
    }
    // ---- layer 4 transposed: 256 -> 256 features (+ 64 encoding slots = row blocks 8, 9 of 10)
    if (a.ray_grad) {
        StashEpilogue ep{gwin(kGradStashEnc)};
        stream_layer<32, 0, 2, 10>(pw, wp + bwd_pack_off(4) / 4 + 8 * 64, dy, none, ep);
    }
    {
        mk_cur = mk_nxt; mk_nxt = mask_rec(2);
        MaskEpilogue<8> ep{mk_cur, zero, nxt, gwin(3 * 256)};
        stream_layer<32, 0, 8, 10>(pw, wp + bwd_pack_off(4) / 4, dy, none, ep);
        advance();
    }
    NIW_STAMP(6);
    // ---- layers 3, 2, 1 transposed: produce dY2, dY1, dY0
#pragma unroll 1
    for (int l = 3; l >= 1; --l) {
        mk_cur = mk_nxt; mk_nxt = mask_rec(l >= 2 ? l - 2 : 0);
        MaskEpilogue<8> ep{mk_cur, zero, nxt, gwin((l - 1) * 256)};
        stream_layer<32, 0, 8, 8>(pw, wp + bwd_pack_off(1) / 4 + (l - 1) * (32 * 8 * 64), dy, none, ep);
        advance();
        NIW_STAMP(10 - l);                           // 7, 8, 9: layers 3, 2, 1
    }
    if (!a.ray_grad) return;
    // ---- layer 0 transposed: 256 -> 64 encoding slots (+ the parked skip-connection part)
    float denc[32], dvenc[16];
    {
        AddStashEpilogue<2> ep{gwin(kGradStashEnc), denc};
        stream_layer<32, 0, 2, 2>(pw, wp + bwd_pack_off(0) / 4, dy, none, ep);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = reinterpret_cast<const f32x4*>(a.grad + (long long)(kGradStashVenc + 8 * q) * P)[qoff];
            dvenc[4 * q] = v[0]; dvenc[4 * q + 1] = v[1]; dvenc[4 * q + 2] = v[2]; dvenc[4 * q + 3] = v[3];
        }
    }
    // ---- encodings -> point / direction -> ray gradients
    float dp[3], du[3];
    enc_backward<NIW_L3D, 8>(denc, a.save + (long long)kSaveEnc * P, P, qoff, h, dp);
    enc_backward<NIW_LVIEW, 4>(dvenc, a.save + (long long)kSaveVenc * P, P, qoff, h, du);
    const long long ri = mc / a.S;
    const float d = a.depth[mc];
    const float rx = a.ray[ri * 3], ry = a.ray[ri * 3 + 1], rz = a.ray[ri * 3 + 2];
    const float nrm = fmaxf(sqrtf(rx * rx + ry * ry + rz * rz), 1e-12f);
    const float ux = rx / nrm, uy = ry / nrm, uz = rz / nrm;
    const float dot = ux * du[0] + uy * du[1] + uz * du[2];
    float gc[3] = {dp[0], dp[1], dp[2]};
    float gr[3] = {dp[0] * d + (du[0] - ux * dot) / nrm, dp[1] * d + (du[1] - uy * dot) / nrm, dp[2] * d + (du[2] - uz * dot) / nrm};
    if (!valid) {
#pragma unroll
        for (int c = 0; c < 3; ++c) gc[c] = gr[c] = 0.f;
    }
    // Per-sample ray gradients, parked as two quads in this wave's own columns of the first eight skip-stash rows (read for the last
    // time by the layer above; the dW pass does not touch the stash).  ray_grad_reduce_kernel sums them per ray in a fixed order:
    // d_center / d_ray are bit-reproducible (rounds 1-2 accumulated them with float atomics, whose order varies from run to run)
    // and need no zero-fill.
    reinterpret_cast<f32x4*>(a.grad + (long long)kGradStashEnc * P)[qoff] =
        h == 0 ? f32x4{gc[0], gc[1], gc[2], gr[0]} : f32x4{gr[1], gr[2], 0.f, 0.f};
    NIW_STAMP_LAST(10);
}

// d_center[r], d_ray[r] = sum over the S samples of ray r of the parked per-sample gradients: one wave per ray, every lane a strided
// partial sum in sample order, then a fixed xor-shuffle tree.
__global__ __launch_bounds__(256) void ray_grad_reduce_kernel(const float* __restrict__ stash, long long P, long long n_rays, int S,
                                                              float* __restrict__ d_center, float* __restrict__ d_ray) {
    const int lane = threadIdx.x & 63;
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_rays) return;
    const f32x4* q0 = reinterpret_cast<const f32x4*>(stash) + r * S;
    const f32x4* q1 = q0 + P;
    float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int s = lane; s < S; s += 64) {
        const f32x4 u = q0[s], v = q1[s];
        acc[0] += u[0]; acc[1] += u[1]; acc[2] += u[2]; acc[3] += u[3]; acc[4] += v[0]; acc[5] += v[1];
    }
#pragma unroll
    for (int c = 0; c < 6; ++c) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc[c] += __shfl_xor(acc[c], o);
    }
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            d_center[r * 3 + c] = acc[c];
            d_ray[r * 3 + c] = acc[3 + c];
        }
    }
}

// per-ray sums of the ray gradients the dX chain (either precision) parked in the first stash rows of `gradws`
int niw_launch_ray_grad_reduce(const float* gradws, long long mpad, int64_t n_rays, int n_samples, float* d_center, float* d_ray, hipStream_t stream) {
    ray_grad_reduce_kernel<<<(int)((n_rays + 3) / 4), 256, 0, stream>>>(gradws + (long long)kGradStashEnc * mpad, mpad, n_rays, n_samples, d_center, d_ray);
    NIW_LAUNCH_CHECK("niw_mlp_bwd (ray-gradient reduction)");
    return NIW_OK;
}

int niw_launch_mlp_bwd_dx(const float* packed, const float* center, const float* ray, const float* depth,
                          int64_t n_rays, int n_samples, int density_activ, const float* rgb, const float* d_rgb,
                          const float* d_sigma, const float* save, float* gradws, float* d_center, float* d_ray,
                          hipStream_t stream) {
    MlpBwdArgs a;
    a.packed = packed; a.center = center; a.ray = ray; a.depth = depth; a.rgb = rgb; a.d_rgb = d_rgb; a.d_sigma = d_sigma;
    a.save = save; a.grad = gradws; a.d_center = d_center; a.d_ray = d_ray;
    a.M = n_rays * (int64_t)n_samples; a.Mpad = niw_mlp_padded_rows(n_rays, n_samples);
    a.S = n_samples; a.act = density_activ; a.ray_grad = (d_center != nullptr && d_ray != nullptr) ? 1 : 0;
    a.zero = 0;
    mlp_bwd_dx_kernel<<<(int)(a.Mpad / 128), 256, 0, stream>>>(a);
    NIW_LAUNCH_CHECK("niw_mlp_bwd (dX chain)");
    if (a.ray_grad) return niw_launch_ray_grad_reduce(gradws, a.Mpad, n_rays, n_samples, d_center, d_ray, stream);
    return NIW_OK;
}
