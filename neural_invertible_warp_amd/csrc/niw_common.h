// Shared host/device helpers for libniw_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <atomic>
#include "niw.h"

void niw_set_error(const char* fmt, ...);

#define NIW_REQUIRE(cond, ...)                       \
    do {                                             \
        if (!(cond)) {                               \
            niw_set_error(__VA_ARGS__);              \
            return NIW_ERR_INVALID_ARG;              \
        }                                            \
    } while (0)

#define NIW_LAUNCH_CHECK(name)                                                       \
    do {                                                                             \
        hipError_t e_ = hipGetLastError();                                           \
        if (e_ != hipSuccess) {                                                      \
            niw_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));     \
            return NIW_ERR_LAUNCH;                                                   \
        }                                                                            \
    } while (0)

// Raise a kernel's dynamic-LDS limit once per device.  `cache` is a function-local static of the caller: one bit per device
// ordinal, write-once (the only process-global state of the library: a read-mostly kernel-attribute cache; two threads racing
// here both set the attribute, which is harmless).
inline int niw_ensure_dynamic_lds(const void* kernel, size_t bytes, std::atomic<unsigned long long>& cache, const char* what) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (cache.load(std::memory_order_acquire) & bit) return NIW_OK;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
        niw_set_error("%s: cannot raise the dynamic LDS limit to %zu bytes", what, bytes);
        return NIW_ERR_LAUNCH;
    }
    cache.fetch_or(bit, std::memory_order_release);
    return NIW_OK;
}

// Operand of the NT GEMM (niw_dw_gemm.hip): feature-major rows of samples.
struct NiwGemmOperand {
    const float* p;
    int rows;                 // valid rows (rows beyond are read as zero)
    long long batch_stride;   // floats between batches (blockIdx.y)
    long long row_stride;     // floats between rows (>= the padded sample count)
};

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ----------------------------------------------------------------------------------------
// Field-MLP geometry (model/nerf.py:373-400 with the arch block of every reference yaml)
// ----------------------------------------------------------------------------------------
namespace niw {

constexpr int kLayers = 10;                       // mlp_feat.0..7, mlp_rgb.0, mlp_rgb.1
constexpr int kWidth = 256;
constexpr int kIn3D = 3 + 6 * NIW_L3D;            // 63
constexpr int kInView = 3 + 6 * NIW_LVIEW;        // 27

__host__ __device__ constexpr int layer_k(int l) {   // reference in_features
    return l == 0 ? kIn3D : l == 4 ? kWidth + kIn3D : l == 8 ? kWidth + kInView : l == 9 ? 128 : kWidth;
}
__host__ __device__ constexpr int layer_n(int l) {   // reference out_features
    return l == 7 ? kWidth + 1 : l == 8 ? 128 : l == 9 ? 3 : kWidth;
}
__host__ __device__ constexpr int weight_off(int l) {   // offset of layer l's weight in the flat parameter vector
    int o = 0;
    for (int i = 0; i < l; ++i) o += layer_n(i) * layer_k(i) + layer_n(i);
    return o;
}
__host__ __device__ constexpr int bias_off(int l) { return weight_off(l) + layer_n(l) * layer_k(l); }
static_assert(bias_off(9) + 3 == NIW_NERF_PARAM_FLOATS, "parameter count");

// MFMA-order geometry.  Forward:  out[n][m] = sum_k W[n][k] act[k][m]   (A = W, B = act)
// slots: k index as consumed by the kernel; 8 slots per k-block (2 lane halves x 4 regs).
__host__ __device__ constexpr int fwd_kb(int l) {       // k-blocks of 8 slots
    return l == 0 ? 8 : l == 4 ? 40 : l == 8 ? 36 : l == 9 ? 16 : 32;
}
// The two skinny heads are NOT row blocks of the forward: the density row (1 output) and the colour layer 9 (3 outputs)
// would each cost whole 32-row MFMA blocks (128 + 64 MFMAs of 8448 per 32 samples, 2.3 %) for 4 useful rows; the forward forms
// them on the vector ALU in the MFMA gaps of layers 7 and 8 from the "head" section of the packed image (below).
__host__ __device__ constexpr int fwd_nb(int l) {       // 32-row output blocks
    return l == 8 ? 4 : l == 9 ? 0 : 8;
}
// Backward: dX[k][m] = sum_n W[n][k] dY[n][m]   (A = W^T, B = dY)
__host__ __device__ constexpr int bwd_rb(int l) {       // reduction k-blocks (over n)
    return l == 7 ? 33 : l == 8 ? 16 : l == 9 ? 1 : 32;
}
__host__ __device__ constexpr int bwd_ob(int l) {       // 32-row output blocks (over input slots)
    return l == 0 ? 2 : l == 4 ? 10 : l == 8 ? 9 : l == 9 ? 4 : 8;
}
__host__ __device__ constexpr int fwd_pack_off(int l) {
    int o = 0;
    for (int i = 0; i < l; ++i) o += fwd_kb(i) * fwd_nb(i) * 256;
    return o;
}
constexpr int kFwdPackFloats = fwd_pack_off(kLayers);
__host__ __device__ constexpr int bwd_pack_off(int l) {
    int o = kFwdPackFloats;
    for (int i = 0; i < l; ++i) o += bwd_rb(i) * bwd_ob(i) * 256;
    return o;
}
constexpr int kBiasPackOff = bwd_pack_off(kLayers);
__host__ __device__ constexpr int bias_pack_off(int l) {     // [nb][h][16] floats per layer
    int o = kBiasPackOff;
    for (int i = 0; i < l; ++i) o += fwd_nb(i) * 32;
    return o;
}
// head section: density weights in the operand order of layer 7's input registers [half][128], colour-layer weights in the
// order of layer 8's output registers [channel][half][64], then {density bias, 3 colour biases}
constexpr int kHeadSigOff = bias_pack_off(kLayers), kHeadRgbOff = kHeadSigOff + 256, kHeadBiasOff = kHeadRgbOff + 384;
constexpr int kPackedFloats = kHeadBiasOff + 4;

// reference column of an encoding slot (see DESIGN.md "encoding slot order"); -1 = zero pad.
// slot = 8q + 4h + t; combo g = 2q + h; g == 0 -> raw xyz; else sincos pairs 2(g-1), 2(g-1)+1.
__host__ __device__ constexpr int enc_slot_col(int slot, int L) {
    int q = slot >> 3, h = (slot >> 2) & 1, t = slot & 3, g = 2 * q + h;
    if (g == 0) return t < 3 ? t : -1;
    int pair = 2 * (g - 1) + (t >> 1);
    if (pair >= 3 * L) return -1;
    int c = pair / L, k = pair % L;
    return 3 + c * 2 * L + ((t & 1) ? L : 0) + k;
}
// reference input column of forward slot s of layer l (-1 = pad)
__host__ __device__ constexpr int fwd_slot_col(int l, int s) {
    if (l == 0) return enc_slot_col(s, NIW_L3D);
    if (l == 4) return s < kWidth ? s : (enc_slot_col(s - kWidth, NIW_L3D) < 0 ? -1 : kWidth + enc_slot_col(s - kWidth, NIW_L3D));
    if (l == 8) return s < kWidth ? s : (enc_slot_col(s - kWidth, NIW_LVIEW) < 0 ? -1 : kWidth + enc_slot_col(s - kWidth, NIW_LVIEW));
    return s < layer_k(l) ? s : -1;
}
// reference output row of kernel row n of layer l (-1 = pad).  Layer 7: kernel rows 0..255 are
// the feature rows (reference rows 1..256) and kernel row 256 is the density row (reference 0).
__host__ __device__ constexpr int out_row(int l, int n) {
    if (l == 7) return n < kWidth ? n + 1 : (n == kWidth ? 0 : -1);
    return n < layer_n(l) ? n : -1;
}

// rows of the activation / gradient workspaces
constexpr int kSaveEnc = 0, kSaveH1 = 64, kSaveFeat = 64 + 7 * 256, kSaveVenc = kSaveFeat + 256,
              kSaveHr = kSaveVenc + 32, kSaveSigma = kSaveHr + 128;
// ReLU sign bits of every hidden activation, for the dX chain (which otherwise re-read all of h1..h7, feat, hr --
// 8.7 KB per sample -- only to recover them): per wave (32 samples) 9 records (output of layers 0..6, feat, hr) of
// 1 KiB = [lane 0..63][4 dwords]: dword nb/2, bit 31 - (16*(nb&1) + r) = (activation > 0) of accumulator register r of row block nb of that
// lane.  72 float-rows' worth of space.
constexpr int kSaveMask = kSaveSigma + 2, kMaskRecords = 9, kMaskRecBytes = 1024, kMaskRows = kMaskRecords * kMaskRecBytes / (4 * 32);
static_assert(kSaveMask + kMaskRows == NIW_SAVE_ROWS, "save rows");
__host__ __device__ constexpr int save_h(int l) { return kSaveH1 + (l - 1) * 256; }   // output of layer l-1, l = 1..7
constexpr int kGradY7 = 7 * 256, kGradRgb0 = kGradY7 + 288, kGradRgb1 = kGradRgb0 + 128;
// stash rows: d(encoding slots) from the layer-4 skip and d(view-encoding slots), parked in the
// workspace between their producer and the end of the chain instead of occupying 48 registers
constexpr int kGradStashEnc = kGradRgb1 + 32, kGradStashVenc = kGradStashEnc + 64;
static_assert(kGradStashVenc + 32 == NIW_GRAD_ROWS, "grad rows");

// Separately rounded fp32 multiply / add / divide, for arithmetic the reference performs as separate tensor ops.
// hipcc compiles with -ffp-contract=fast and HIP's __fmul_rn / __fadd_rn are plain `*` / `+` in a header, so
// __fadd_rn(c, __fmul_rn(a, b)) is emitted as ONE v_fma_f32 (single rounding) -- a 1-ulp difference in a sample
// position that the 2^9*pi encoding band multiplies by 1.6e3.  Without the `contract` flag on these
// instructions LLVM cannot fuse them, also after inlining.
__device__ __forceinline__ float mul_rn(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ float add_rn(float a, float b) {
#pragma clang fp contract(off)
    return a + b;
}
__device__ __forceinline__ float sub_rn(float a, float b) {
#pragma clang fp contract(off)
    return a - b;
}

// sin / cos of the band argument fl32(x * 2^k * pi32).
// The reference evaluates sin(fl32(x * f_k)) with f_k = 2^k * fl32(pi) (nerf.py:478-480; the warp's embedder alike:
// model/nvp/embedder.py:26-32).  Scaling by a
// power of two commutes with rounding, so fl32(x * f_k) == 2^k * fl32(x * pi32) EXACTLY: one range
// reduction per coordinate serves all bands.  t = arg0 / (2 pi) is formed in fp64 (arguments reach
// 1e8 with inverse-depth sampling; fp64 keeps the reduced angle good to < 1e-6 even there), the
// band's revolution fraction is frac(2^k t), and sin / cos come from the Cephes minimax polynomials
// on [-pi/4, pi/4] (~1 ulp) with quadrant rotation.  ~35 instructions instead of the ~500 of a
// full-range sincosf, which was 8 % of the forward kernel as an un-overlappable prologue.
__device__ __forceinline__ void sincos_band(double t, int k, float& s, float& c) {
    const double tk = t * (double)(1 << k);
    const double fr = tk - rint(tk);                       // revolutions in [-0.5, 0.5]
    const double q = rint(fr * 4.0);                       // quadrant -2..2
    const float th = (float)((fr - q * 0.25) * 6.283185307179586476925);   // [-pi/4, pi/4]
    const float z = th * th;
    const float ps = th + th * z * (-1.6666654611e-1f + z * (8.3321608736e-3f + z * -1.9515295891e-4f));
    const float pc = 1.f - 0.5f * z + z * z * (4.166664568298827e-2f + z * (-1.388731625493765e-3f + z * 2.443315711809948e-5f));
    const int qi = (int)q & 3;                             // rotate by q * 90 degrees
    const float ss = (qi & 1) ? pc : ps, cc = (qi & 1) ? ps : pc;
    s = (qi == 2 || qi == 3) ? -ss : ss;
    c = (qi == 1 || qi == 2) ? -cc : cc;
}

// One element of torch.optim.Adam's single-tensor update (no amsgrad / weight decay), shared by adam_kernel (niw_sampling.hip) and
// adam_multi_kernel (niw_step.hip) so that both round alike.  Explicit about what is fused: torch's lerp_ and addcmul_ / addcdiv_
// kernels evaluate `a + w * b` forms (one fma each), while mul_(beta2), the division by sqrt(bias_correction2) and add_(eps) are
// separate tensor ops (separately rounded).
__device__ __forceinline__ void adam_update(float& p, float g, float& m, float& v, float w1, float b2, float w2, float eps, float step_size,
                                            float bc2_sqrt) {
    const float mi = fmaf(w1, sub_rn(g, m), m);                        // exp_avg.lerp_(grad, 1 - beta1)
    const float vi = fmaf(mul_rn(w2, g), g, mul_rn(v, b2));            // mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
    m = mi; v = vi;
    const float denom = add_rn(__fdiv_rn(__fsqrt_rn(vi), bc2_sqrt), eps);   // (exp_avg_sq.sqrt() / sqrt(bias_correction2)).add_(eps)
    p = fmaf(-step_size, __fdiv_rn(mi, denom), p);                     // addcdiv_(exp_avg, denom, value=-lr / bias_correction1)
}

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
// feature row (within its 32-block) held by accumulator register r of lane half h
__device__ __forceinline__ constexpr int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

}  // namespace niw
