// NVP invertible warp: per-point part of DeformNetwork.forward / .inverse (reference
// model/nvp/nvp_ndr.py:365-468, 471-567) with the annealed embedder (model/nvp/embedder.py:41-50).
//
// Three coupling blocks, form 0, focus axis z, y, x (nvp_ndr.py:389-399).  Per block:
//   part a:  focus' = focus - head_a( softplus100( W_a . emb26(other) + view_bias_a ) )
//   part b:  (theta, t) = head_b( softplus100( W_b . emb13(focus') + view_bias_b ) )
//            other' = [[cos, sin], [-sin, cos]] (other - t)                 (euler2rot_2dinv, :166-174)
// The 128 latent columns of the first layers are per-view constants and arrive folded into
// `view_b` (computed by the host mirror together with weight-norm), so a point costs
// 3 * 128 * (26 + 13) MACs instead of 3 * 128 * (154 + 141).
//
// Forward / inverse: a group of 16 lanes per point (each lane 8 of the 128 hidden units, group
// sums by xor shuffles), one view per blockIdx.y, first-layer weights of all
// blocks staged once in LDS (69 KB) and read as broadcasts.
// Backward: the same thread-per-point kernel walks the blocks in reverse for d(points) and
// writes, feature-major, the per-point factors of every parameter gradient (pre-activation
// gradients, embeddings, hidden activations, head gradients, view indicator rows); the
// parameter gradients themselves are then three batched NT GEMMs over the points on the
// fp32 MFMA path (niw_dw_gemm.hip) -- no atomics, deterministic.
#include "niw_common.h"
#include <stdlib.h>

int niw_launch_nt_gemm(int wide, NiwGemmOperand A, NiwGemmOperand B, long long mpad, int batches, float* partial,
                       int bias_side, int* nsplit_out, hipStream_t st);
int niw_launch_nt_gemm_pairs(int n, const NiwGemmOperand* A, const NiwGemmOperand* B, const int* bias_side, long long mpad, float* partial,
                             int* nsplit_out, hipStream_t st);

namespace {

constexpr int kHid = 128, kEa = 26, kEb = 13, kNF = 6;
// w_emb / d_w_emb rows are padded to 28 / 16 floats: every row starts 16-byte aligned and is read from LDS as ds_read_b128
// broadcasts (with packed 26- / 13-float rows every weight was its own ds_read_b32: one LDS instruction per FMA, and the
// backward read every row twice)
constexpr int kSa = 28, kSb = 16;
constexpr int kWembBlock = kHid * (kSa + kSb);          // 5632 floats per coupling block
constexpr int kHeadBlock = kHid + 1 + 3 * kHid + 3;      // 516
constexpr float kPi32 = 3.14159274101257324f;            // fp32(pi): the band table is an fp32 tensor (embedder.py:26)

// workspace rows per coupling block (feature-major [rows][Ppad], Ppad = all views' points)
constexpr int kRowGa = 0, kRowGb = 128;                  // A1: pre-activation gradients, parts a / b
constexpr int kRowEa = 256, kRowEb = 288, kRowInd = 320; // B1: embeddings (26 of 32, 13 of 32) and 64 view indicator rows
constexpr int kRowHa = 384, kRowHb = 512;                // A2: hidden activations
constexpr int kRowGo = 640;                              // B2: head gradients (d delta, d theta, d t0, d t1)
constexpr int kRowsPerBlock = 644;


// Softplus(beta = 100, threshold = 20) of the reference's coupling MLPs (nvp_ndr.py actfn, torch.nn.Softplus) and its derivative:
//     z = 100 x;   softplus = z > 20 ? x : log1p(e^z) / 100;   d softplus / dx = z > 20 ? 1 : e^z / (1 + e^z)
// (torch's softplus backward uses exactly e^z / (e^z + 1)).  libm's expf + log1pf cost ~250 instructions per value and were
// 45 % of the forward kernel (measured with the activation stubbed out); this form is ~25:
//   e^z   = 2^t (1 + r ln 2) with t = fl(z log2 e) on the hardware exp2 and r the exact rounding residue of that product plus the
//           low word of log2 e -- without the correction the argument's rounding alone costs 7 ulp at |z| = 20;
//   log1p = e (1 - e/2 + e^2/3 - e^3/4) for e < 2^-6 (truncation < 2e-10 relative), else ln(u) + (e - (u - 1)) / u with
//           u = fl(1 + e) on the hardware log2 (the second term restores what rounding 1 + e lost).
// Accuracy: a few ulp of the result, i.e. absolute errors < 1e-9 on activations of O(0.01 .. 0.2) -- below the rounding noise
// of the 26-term dot products that feed them.
struct Softplus100 {
    float value, slope;
};
__device__ __forceinline__ Softplus100 softplus100_pair(float x) {
    const float z = 100.f * x;
    const float kL2eHi = 1.44269502162933349609375f, kL2eLo = 1.925963033500011e-8f, kLn2 = 0.693147182464599609375f;
    const float zc = fminf(z, 20.f);
    const float t = zc * kL2eHi;
    const float r = fmaf(zc, kL2eLo, fmaf(zc, kL2eHi, -t));
    float e = __builtin_amdgcn_exp2f(t);
    e = fmaf(e, r * kLn2, e);
    const float u = 1.f + e;
    const float inv_u = 1.f / u;
    const float series = e * fmaf(e, fmaf(e, fmaf(e, -0.25f, 0.333333343267440796f), -0.5f), 1.f);
    const float general = fmaf(__builtin_amdgcn_logf(u), kLn2, (e - (u - 1.f)) * inv_u);
    const float l1p = e < 0.015625f ? series : general;
    Softplus100 o;
    o.value = z > 20.f ? x : l1p / 100.f;
    o.slope = z > 20.f ? 1.f : e * inv_u;
    return o;
}
__device__ __forceinline__ float softplus100(float x) { return softplus100_pair(x).value; }
__device__ __forceinline__ float dsoftplus100(float x) { return softplus100_pair(x).slope; }

// sin / cos of the coupling rotation angle (any magnitude): revolutions in fp64, then the short polynomial of sincos_band --
// libm's full-range sincosf keeps a Payne-Hanek work array in scratch memory (56 bytes per lane) and costs ~150 instructions
__device__ __forceinline__ void rotation_sincos(float theta, float& s, float& c) {
    niw::sincos_band((double)theta * 0.15915494309189533577, 0, s, c);
}

// A point is served by a GROUP of kGroup adjacent lanes; lane `sub` owns hidden units sub, sub+kGroup, ...
// and the group combines partial sums with xor shuffles.  (4 lanes per point left half of the SIMDs idle at
// the 8k points of a training step and made every lane evaluate all 18 sincos of a block's embeddings.)
#ifndef NIW_WARP_GROUP
#define NIW_WARP_GROUP 16
#endif
constexpr int kGroup = NIW_WARP_GROUP, kPtsPerWg = 256 / kGroup;
static_assert(kGroup == 4 || kGroup == 8 || kGroup == 16 || kGroup == 32 || kGroup == 64, "lanes per point");
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = 1; o < kGroup; o <<= 1) v += __shfl_xor(v, o);
    return v;
}

// embedding of D inputs: [x (D), then per band sin(D), cos(D)], times ps * cw[band].  The 6*D sincos
// evaluations are spread over the group's lanes and exchanged by shuffles.
template <int D>
__device__ __forceinline__ void embed(const float (&x)[D], const float* __restrict__ cw, float ps, int sub, float (&e)[D * (1 + 2 * kNF)]) {
    constexpr int N = kNF * D, T = (N + kGroup - 1) / kGroup;
    float sl[T], cl[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int n = t * kGroup + sub;
        const int i = n / D, d = n % D;
        float xv = x[0];
#pragma unroll
        for (int dd = 1; dd < D; ++dd) xv = d == dd ? x[dd] : xv;
        sl[t] = cl[t] = 0.f;
        // fl32(x * 2^i pi32) = 2^i fl32(x pi32) exactly: one fp64 range reduction, then a short polynomial (niw_common.h sincos_band)
        if (n < N) niw::sincos_band((double)niw::mul_rn(xv, kPi32) * 0.15915494309189533577, i < kNF ? i : 0, sl[t], cl[t]);
    }
#pragma unroll
    for (int d = 0; d < D; ++d) e[d] = ps * x[d];
#pragma unroll
    for (int i = 0; i < kNF; ++i) {
        const float w = ps * cw[i];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int n = i * D + d;
            e[D * (1 + 2 * i) + d] = w * __shfl(sl[n / kGroup], n % kGroup, kGroup);
            e[D * (2 + 2 * i) + d] = w * __shfl(cl[n / kGroup], n % kGroup, kGroup);
        }
    }
}
// gradient of the embedding w.r.t. its inputs given d(e); uses the embedding values themselves
template <int D>
__device__ __forceinline__ void embed_bwd(const float (&e)[D * (1 + 2 * kNF)], const float (&ge)[D * (1 + 2 * kNF)], float ps,
                                          float (&gx)[D]) {
#pragma unroll
    for (int d = 0; d < D; ++d) gx[d] = ps * ge[d];
#pragma unroll
    for (int i = 0; i < kNF; ++i) {
        const float f = kPi32 * (float)(1 << i);
#pragma unroll
        for (int d = 0; d < D; ++d)   // d(w sin)/dx = f (w cos),  d(w cos)/dx = -f (w sin)
            gx[d] += f * (ge[D * (1 + 2 * i) + d] * e[D * (2 + 2 * i) + d] - ge[D * (2 + 2 * i) + d] * e[D * (1 + 2 * i) + d]);
    }
}

// one padded weight row from LDS as 16-byte reads (S floats, E of them meaningful)
template <int S>
__device__ __forceinline__ void load_row(const float* __restrict__ w, float (&r)[S]) {
#pragma unroll
    for (int c = 0; c < S; c += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(w + c);
        r[c] = v[0]; r[c + 1] = v[1]; r[c + 2] = v[2]; r[c + 3] = v[3];
    }
}
template <int E, int S>
__device__ __forceinline__ float dot_row(const float (&r)[S], const float (&e)[E]) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < E; ++c) s += r[c] * e[c];
    return s;
}

struct WarpArgs {
    const float* w_emb;
    const float* view_b;
    const float* w_head;
    const float* pts;
    const float* ps_a;
    const float* ps_b;
    const float* d_out;
    const float* xin_in;    // backward: the block inputs the forward saved ([views][points][3 blocks][3]) or NULL (recompute them)
    float* xin_save;        // forward: where to save them, or NULL
    float* out;
    float* d_pts;
    float* ws;
    long long n_pts, ppad;
    int n_views, inverse, use_iw;
    float cw[kNF];
    float iw[kNF];      // index window (reference_exact): scales whole points by their index
    const float* win_dev;   // device copy of {cw, iw} read at run time (HIP-graph replays) or NULL: the by-value copies above
};

// the annealing windows of this launch: by value, or from the device buffer a captured graph re-reads on every replay
struct Windows {
    float cw[kNF], iw[kNF];
    int use_iw;
};
__device__ __forceinline__ Windows load_windows(const WarpArgs& a) {
    Windows w;
#pragma unroll
    for (int i = 0; i < kNF; ++i) {
        w.cw[i] = a.win_dev ? a.win_dev[i] : a.cw[i];
        w.iw[i] = a.win_dev ? a.win_dev[kNF + i] : a.iw[i];
    }
    w.use_iw = a.use_iw;
    return w;
}

// embedder.py:47 on 4-D input: window value i multiplies points (2i+1)d .. (2i+3)d-1 along dim 1
__device__ __forceinline__ float index_scale(const Windows& w, long long p, int d) {
    const long long q = p / d;
    if (!(w.use_iw && q >= 1 && q < 1 + 2 * kNF)) return 1.f;
    const int k = (int)((q - 1) >> 1);
    float v = w.iw[0];
#pragma unroll
    for (int i = 1; i < kNF; ++i) v = k == i ? w.iw[i] : v;      // register select (a dynamic index would spill the array)
    return v;
}

// 69 KB of weights per workgroup: 16-byte loads, all of a thread's loads in flight before the first LDS write.  (As a scalar
// load -> store loop of 68 dependent round trips this prologue was ~20 of the forward kernel's 30 us at every size.)
template <int N4>
__device__ __forceinline__ void stage_block(const float* __restrict__ src, float* __restrict__ dst) {
    constexpr int PER = (N4 + 255) / 256;
    f32x4 v[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int i = threadIdx.x + k * 256;
        if (i < N4) v[k] = reinterpret_cast<const f32x4*>(src)[i];
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int i = threadIdx.x + k * 256;
        if (i < N4) reinterpret_cast<f32x4*>(dst)[i] = v[k];
    }
}
__device__ __forceinline__ void stage_weights(const WarpArgs& a, float* lw, float* lh, float* lv, int view) {
    static_assert((3 * kWembBlock) % 4 == 0 && (3 * kHeadBlock) % 4 == 0 && (3 * 2 * kHid) % 4 == 0, "16-byte staging");
    stage_block<3 * kWembBlock / 4>(a.w_emb, lw);
    stage_block<3 * kHeadBlock / 4>(a.w_head, lh);
    stage_block<3 * 2 * kHid / 4>(a.view_b + (long long)view * 3 * 2 * kHid, lv);
    __syncthreads();
}

// part a: delta = head_a(softplus(W_a e + v_a))
__device__ __forceinline__ float part_a(const float* lw, const float* lh, const float* lv, int b, int sub, const float (&ea)[kEa]) {
    const float* W = lw + b * kWembBlock;
    const float* hd = lh + b * kHeadBlock;
    const float* vb = lv + (b * 2 + 0) * kHid;
    float delta = 0.f;
    for (int u = sub; u < kHid; u += kGroup) {
        float row[kSa];
        load_row<kSa>(W + u * kSa, row);
        delta += hd[u] * softplus100(vb[u] + dot_row<kEa, kSa>(row, ea));
    }
    return group_sum(delta) + hd[kHid];
}
// part b: (theta, t0, t1) = head_b(softplus(W_b e + v_b))
__device__ __forceinline__ void part_b(const float* lw, const float* lh, const float* lv, int b, int sub, const float (&eb)[kEb], float (&o)[3]) {
    const float* W = lw + b * kWembBlock + kHid * kSa;
    const float* hd = lh + b * kHeadBlock + kHid + 1;
    const float* vb = lv + (b * 2 + 1) * kHid;
    o[0] = o[1] = o[2] = 0.f;
    for (int u = sub; u < kHid; u += kGroup) {
        float row[kSb];
        load_row<kSb>(W + u * kSb, row);
        const float hh = softplus100(vb[u] + dot_row<kEb, kSb>(row, eb));
        o[0] += hd[u] * hh; o[1] += hd[kHid + u] * hh; o[2] += hd[2 * kHid + u] * hh;
    }
    o[0] = group_sum(o[0]) + hd[3 * kHid];
    o[1] = group_sum(o[1]) + hd[3 * kHid + 1];
    o[2] = group_sum(o[2]) + hd[3 * kHid + 2];
}

__device__ __forceinline__ void axes(int b, int& f, int& o0, int& o1) {
    f = 2 - b;                        // focus z, y, x
    o0 = b == 2 ? 1 : 0;
    o1 = b == 0 ? 1 : 2;
}

__device__ __forceinline__ void block_fwd(const float* lw, const float* lh, const float* lv, const float* cw, float psa, float psb,
                                          int b, int sub, float (&x)[3]) {
    int f, o0, o1;
    axes(b, f, o0, o1);
    const float oth[2] = {x[o0], x[o1]};
    float ea[kEa];
    embed<2>(oth, cw, psa, sub, ea);
    const float foc[1] = {x[f] - part_a(lw, lh, lv, b, sub, ea)};
    float eb[kEb], o[3];
    embed<1>(foc, cw, psb, sub, eb);
    part_b(lw, lh, lv, b, sub, eb, o);
    float s, c;
    rotation_sincos(o[0], s, c);
    const float d0 = oth[0] - o[1], d1 = oth[1] - o[2];
    x[f] = foc[0];
    x[o0] = c * d0 + s * d1;
    x[o1] = -s * d0 + c * d1;
}

__device__ __forceinline__ void block_inv(const float* lw, const float* lh, const float* lv, const float* cw, float psa, float psb,
                                          int b, int sub, float (&x)[3]) {
    int f, o0, o1;
    axes(b, f, o0, o1);
    const float single[1] = {x[f]};
    float eb[kEb], o[3];
    embed<1>(single, cw, psb, sub, eb);
    part_b(lw, lh, lv, b, sub, eb, o);
    float s, c;
    rotation_sincos(o[0], s, c);                            // euler2rot_2d: [[cos, -sin], [sin, cos]]
    const float pr[2] = {c * x[o0] - s * x[o1] + o[1], s * x[o0] + c * x[o1] + o[2]};
    float ea[kEa];
    embed<2>(pr, cw, psa, sub, ea);
    x[f] = single[0] + part_a(lw, lh, lv, b, sub, ea);
    x[o0] = pr[0];
    x[o1] = pr[1];
}

__global__ __launch_bounds__(256) void warp_fwd_kernel(WarpArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* lw = lds;
    float* lh = lw + 3 * kWembBlock;
    float* lv = lh + 3 * kHeadBlock;
    const int view = blockIdx.y;
    stage_weights(a, lw, lh, lv, view);
    const Windows win = load_windows(a);
    const int sub = threadIdx.x & (kGroup - 1);
    const long long p = (long long)blockIdx.x * (blockDim.x / kGroup) + threadIdx.x / kGroup;
    if (p >= a.n_pts) return;                     // whole groups leave together
    const long long gi = (long long)view * a.n_pts + p;
    float x[3] = {a.pts[gi * 3], a.pts[gi * 3 + 1], a.pts[gi * 3 + 2]};
    const float psa = (a.ps_a ? a.ps_a[p] : 1.f) * index_scale(win, p, 2), psb = (a.ps_b ? a.ps_b[p] : 1.f) * index_scale(win, p, 1);
    if (!a.inverse) {
        for (int b = 0; b < 3; ++b) {
            if (a.xin_save && sub == 0) {
                float* xs = a.xin_save + gi * 9 + b * 3;
                xs[0] = x[0]; xs[1] = x[1]; xs[2] = x[2];
            }
            block_fwd(lw, lh, lv, win.cw, psa, psb, b, sub, x);
        }
    } else {
        for (int b = 2; b >= 0; --b) block_inv(lw, lh, lv, win.cw, psa, psb, b, sub, x);
    }
    if (sub == 0) { a.out[gi * 3] = x[0]; a.out[gi * 3 + 1] = x[1]; a.out[gi * 3 + 2] = x[2]; }
}

// one per-point factor of the parameter gradients -> row `row` of the block's feature-major workspace
#define NIW_WS(row, v) ws[(long long)(row) * P] = (v)
__global__ __launch_bounds__(256) void warp_bwd_kernel(WarpArgs a) {
#ifdef NIW_WARP_BWD_PRIO
    __builtin_amdgcn_s_setprio(NIW_WARP_BWD_PRIO);       // (diagnostic: wave priority of this vector-ALU kernel beside the dW launch's matrix waves)
#endif
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* lw = lds;
    float* lh = lw + 3 * kWembBlock;
    float* lv = lh + 3 * kHeadBlock;
    const int view = blockIdx.y;
    stage_weights(a, lw, lh, lv, view);
    const Windows win = load_windows(a);
    const int sub = threadIdx.x & (kGroup - 1);
    const long long p = (long long)blockIdx.x * (blockDim.x / kGroup) + threadIdx.x / kGroup;
    if (p >= a.n_pts) return;                     // whole groups leave together
    const long long gi = (long long)view * a.n_pts + p;
    const float psa = (a.ps_a ? a.ps_a[p] : 1.f) * index_scale(win, p, 2), psb = (a.ps_b ? a.ps_b[p] : 1.f) * index_scale(win, p, 1);
    float xin[3][3];
    if (a.xin_in) {              // the forward pass of this step left every block's input behind: a third of this kernel's work
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int c = 0; c < 3; ++c) xin[b][c] = a.xin_in[gi * 9 + b * 3 + c];
    } else {
        float x[3] = {a.pts[gi * 3], a.pts[gi * 3 + 1], a.pts[gi * 3 + 2]};
        for (int b = 0; b < 3; ++b) {
            xin[b][0] = x[0]; xin[b][1] = x[1]; xin[b][2] = x[2];
            block_fwd(lw, lh, lv, win.cw, psa, psb, b, sub, x);
        }
    }
    float gx[3] = {a.d_out[gi * 3], a.d_out[gi * 3 + 1], a.d_out[gi * 3 + 2]};
    for (int b = 2; b >= 0; --b) {
        int f, o0, o1;
        axes(b, f, o0, o1);
        float* ws = a.ws + (long long)b * kRowsPerBlock * a.ppad + gi;     // column gi of this block's rows
        const long long P = a.ppad;
        const float* Wa = lw + b * kWembBlock;
        const float* Wb = Wa + kHid * kSa;
        const float* hda = lh + b * kHeadBlock;
        const float* hdb = hda + kHid + 1;
        const float* va = lv + (b * 2 + 0) * kHid;
        const float* vb = lv + (b * 2 + 1) * kHid;
        // ---- recompute the block forward
        const float oth[2] = {xin[b][o0], xin[b][o1]};
        float ea[kEa], eb[kEb], o[3];
        embed<2>(oth, win.cw, psa, sub, ea);
        const float foc[1] = {xin[b][f] - part_a(lw, lh, lv, b, sub, ea)};
        embed<1>(foc, win.cw, psb, sub, eb);
        part_b(lw, lh, lv, b, sub, eb, o);
        float s, c;
        rotation_sincos(o[0], s, c);
        const float d0 = oth[0] - o[1], d1 = oth[1] - o[2];
        const float n0 = c * d0 + s * d1, n1 = -s * d0 + c * d1;
        // ---- rotation / translation
        const float g_n0 = gx[o0], g_n1 = gx[o1];
        const float g_d0 = c * g_n0 - s * g_n1, g_d1 = s * g_n0 + c * g_n1;
        const float go[3] = {g_n0 * n1 - g_n1 * n0, -g_d0, -g_d1};          // d theta, d t0, d t1
        // ---- part b backward
        float geb[kEb];
#pragma unroll
        for (int k = 0; k < kEb; ++k) geb[k] = 0.f;
        for (int u = sub; u < kHid; u += kGroup) {
            float row[kSb];
            load_row<kSb>(Wb + u * kSb, row);
            const float pre = vb[u] + dot_row<kEb, kSb>(row, eb);
            const float gh = hdb[u] * go[0] + hdb[kHid + u] * go[1] + hdb[2 * kHid + u] * go[2];
            const Softplus100 act = softplus100_pair(pre);
            const float gp = gh * act.slope;
            NIW_WS(kRowGb + u, gp);
            NIW_WS(kRowHb + u, act.value);
#pragma unroll
            for (int k = 0; k < kEb; ++k) geb[k] += row[k] * gp;
        }
#pragma unroll
        for (int k = 0; k < kEb; ++k) geb[k] = group_sum(geb[k]);
        float gfoc[1];
        embed_bwd<1>(eb, geb, psb, gfoc);
        const float g_foc = gx[f] + gfoc[0];          // d focus'
        const float g_delta = -g_foc;
        // ---- part a backward
        float gea[kEa];
#pragma unroll
        for (int k = 0; k < kEa; ++k) gea[k] = 0.f;
        for (int u = sub; u < kHid; u += kGroup) {
            float row[kSa];
            load_row<kSa>(Wa + u * kSa, row);
            const float pre = va[u] + dot_row<kEa, kSa>(row, ea);
            const Softplus100 act = softplus100_pair(pre);
            const float gp = g_delta * hda[u] * act.slope;
            NIW_WS(kRowGa + u, gp);
            NIW_WS(kRowHa + u, act.value);
#pragma unroll
            for (int k = 0; k < kEa; ++k) gea[k] += row[k] * gp;
        }
#pragma unroll
        for (int k = 0; k < kEa; ++k) gea[k] = group_sum(gea[k]);
        float goth[2];
        embed_bwd<2>(ea, gea, psa, goth);
        // ---- per-point factors of the parameter gradients
        // (the factor rows that do not depend on the hidden unit are split over the group)
#pragma unroll
        for (int k = 0; k < 32; ++k)
            if (k % kGroup == sub) NIW_WS(kRowEa + k, k < kEa ? ea[k] : 0.f);
#pragma unroll
        for (int k = 0; k < 32; ++k)
            if (k % kGroup == sub) NIW_WS(kRowEb + k, k < kEb ? eb[k] : 0.f);
        for (int v = sub; v < 64; v += kGroup) NIW_WS(kRowInd + v, v == view ? 1.f : 0.f);
        if (sub == 0) {
            NIW_WS(kRowGo + 0, g_delta);
            NIW_WS(kRowGo + 1, go[0]);
            NIW_WS(kRowGo + 2, go[1]);
            NIW_WS(kRowGo + 3, go[2]);
        }
        gx[f] = g_foc;
        gx[o0] = g_d0 + goth[0];
        gx[o1] = g_d1 + goth[1];
    }
    if (a.d_pts && sub == 0) { a.d_pts[gi * 3] = gx[0]; a.d_pts[gi * 3 + 1] = gx[1]; a.d_pts[gi * 3 + 2] = gx[2]; }
}

// fixed-order sum over split-M partial tiles with 8 independent accumulators: 8 loads in flight per thread instead of an
// nsplit-long chain of dependent cold reads (the order does not depend on timing)
__device__ __forceinline__ float sum_tiles(const float* __restrict__ p, long long stride, int n) {
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int w = 0;
    // sixteen loads in flight per trip (a short reduction has 15 partial tiles: one round trip instead of two on the chain of small
    // launches that ends a rank's iteration); the additions keep the order of the eight-accumulator loop
    for (; w + 16 <= n; w += 16) {
        float x[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) x[k] = p[(long long)(w + k) * stride];
#pragma unroll
        for (int k = 0; k < 16; ++k) s[k & 7] += x[k];
    }
    if (w < n) {
        float x[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) x[k] = w + k < n ? p[(long long)(w + k) * stride] : 0.f;
        const int full = (n - w) & ~7;                          // whole groups of eight go to the eight accumulators, the rest to s[0]
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (w + k < n) { if (k < full) s[k & 7] += x[k]; else s[0] += x[k]; }
    }
    return ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
}

// reduce the partial tiles of the two GEMM families and scatter into d_w_emb / d_view_b / d_w_head
// tk2: columns of the second family's tiles -- 64 (its own 256 x 64 launch) or 256 (one launch with the first family, niw_launch_nt_gemm_pairs).
// One thread per OUTPUT (round 5; rounds 1-4 launched a thread per tile element, 82 k per coupling block of which 7 k had anything to
// do): [128 x 28 | 128 x 16] first-layer weights incl. their zero pad columns, [256 x views] per-view biases, [128 + 3 x 128] head weights,
// 4 head biases.  Same sums over the same partial tiles.
__global__ void warp_reduce_kernel(const float* __restrict__ p1, int nsplit1, const float* __restrict__ p2, int nsplit2, int tk2,
                                   int n_views, float* __restrict__ d_w_emb, float* __restrict__ d_view_b, float* __restrict__ d_w_head) {
    const int b = blockIdx.y;                                   // coupling block
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    constexpr int T1 = 256 * 256 + 256;
    const int T2 = 256 * tk2 + 256;
    const float* t1 = p1 + (long long)b * nsplit1 * T1;         // tile 1: rows [ga 0..127 | gb 128..255], columns [ea 0..31 | eb 32..63 | views 64..127]
    const float* t2 = p2 + (long long)b * nsplit2 * T2;         // tile 2: rows [ha 0..127 | hb 128..255], columns [d delta, d theta, d t0, d t1, ...]
    if (j < kHid * kSa) {                                       // part a: 26 embedding columns + 2 pad columns (overwritten with zero)
        const int u = j / kSa, c = j % kSa;
        d_w_emb[b * kWembBlock + u * kSa + c] = c >= kEa ? 0.f : sum_tiles(t1 + u * 256 + c, T1, nsplit1);
        return;
    }
    j -= kHid * kSa;
    if (j < kHid * kSb) {                                       // part b: 13 + 3 pad
        const int u = j / kSb, c = j % kSb;
        d_w_emb[b * kWembBlock + kHid * kSa + u * kSb + c] = c >= kEb ? 0.f : sum_tiles(t1 + (kHid + u) * 256 + 32 + c, T1, nsplit1);
        return;
    }
    j -= kHid * kSb;
    if (j < 2 * kHid * n_views) {                               // per-view biases: the view indicator columns
        const int r = j / n_views, v = j % n_views, part = r >> 7, u = r & 127;
        d_view_b[((long long)(v * 3 + b) * 2 + part) * kHid + u] = sum_tiles(t1 + r * 256 + 64 + v, T1, nsplit1);
        return;
    }
    j -= 2 * kHid * n_views;
    if (j < kHid) {                                             // lin_a_1.weight: rows ha, column d delta
        d_w_head[b * kHeadBlock + j] = sum_tiles(t2 + j * tk2, T2, nsplit2);
        return;
    }
    j -= kHid;
    if (j < 3 * kHid) {                                         // lin_b_1.weight [3][128]: rows hb, columns d theta, d t0, d t1
        const int c = j / kHid, u = j % kHid;
        d_w_head[b * kHeadBlock + kHid + 1 + c * kHid + u] = sum_tiles(t2 + (kHid + u) * tk2 + 1 + c, T2, nsplit2);
        return;
    }
    j -= 3 * kHid;
    if (j < 4) {                                                // row sums of the head gradients = head biases
        float* dst = j == 0 ? d_w_head + b * kHeadBlock + kHid : d_w_head + b * kHeadBlock + kHid + 1 + 3 * kHid + (j - 1);
        *dst = sum_tiles(t2 + 256 * tk2 + j, T2, nsplit2);
    }
}

constexpr size_t kWarpLds = (3 * kWembBlock + 3 * kHeadBlock + 3 * 2 * kHid) * sizeof(float);

int fill_args(WarpArgs& a, const float* w_emb, const float* view_b, const float* w_head, const float* pts, int n_views,
              int64_t n_pts, const float* chan_w, const float* index_window, const float* window_dev, int use_index_window,
              const float* ps_a, const float* ps_b) {
    NIW_REQUIRE(w_emb && view_b && w_head && pts, "niw_warp: null pointer");
    NIW_REQUIRE(n_views > 0 && n_pts > 0, "niw_warp: empty input (views=%d, points=%lld)", n_views, (long long)n_pts);
    a.w_emb = w_emb; a.view_b = view_b; a.w_head = w_head; a.pts = pts; a.ps_a = ps_a; a.ps_b = ps_b;
    a.n_pts = n_pts; a.n_views = n_views;
    for (int i = 0; i < kNF; ++i) a.cw[i] = chan_w ? chan_w[i] : 1.f;
    a.use_iw = window_dev ? (use_index_window != 0) : (index_window != nullptr);
    for (int i = 0; i < kNF; ++i) a.iw[i] = index_window ? index_window[i] : 1.f;
    a.win_dev = window_dev;
    return NIW_OK;
}

long long warp_ppad(int n_views, int64_t n_pts) { return ((long long)n_views * n_pts + 31) / 32 * 32; }

// The factor rows are [row][ppad] with ppad = points rounded up to 32: the GEMMs reduce over all ppad columns, so the < 32 pad
// columns of every row must be zero.  A kernel (not hipMemsetAsync): a memset NODE of a captured graph was observed to leave
// stale pad columns on replays at full size (NaN parameters after the second replay), and only rows x pad elements need writing.
__global__ void warp_zero_pad_kernel(float* ws, long long n_rows, long long ppad, int first_pad, int n_pad) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_rows * n_pad) ws[(i / n_pad) * ppad + first_pad + (int)(i % n_pad)] = 0.f;
}

}  // namespace

static_assert(NIW_WARP_WEMB_FLOATS == 3 * kWembBlock && NIW_WARP_WHEAD_FLOATS == 3 * kHeadBlock, "header constants");

extern "C" int niw_warp_fwd(const float* w_emb, const float* view_b, const float* w_head, const float* pts,
                            int n_views, int64_t n_pts, const float* chan_w, const float* index_window, const float* window_dev,
                            int use_index_window, const float* pt_scale_a, const float* pt_scale_b, int inverse, float* out,
                            float* xin_save, niw_stream_t stream) {
    WarpArgs a{};
    int rc = fill_args(a, w_emb, view_b, w_head, pts, n_views, n_pts, chan_w, index_window, window_dev, use_index_window, pt_scale_a, pt_scale_b);
    if (rc != NIW_OK) return rc;
    NIW_REQUIRE(out, "niw_warp_fwd: null output");
    NIW_REQUIRE(!(inverse && xin_save), "niw_warp_fwd: block inputs are saved for the forward warp only");
    a.out = out; a.inverse = inverse; a.xin_save = xin_save;
    static std::atomic<unsigned long long> attr_set{0ull};
    if (int rc2 = niw_ensure_dynamic_lds(reinterpret_cast<const void*>(warp_fwd_kernel), kWarpLds, attr_set, "niw_warp_fwd")) return rc2;
    warp_fwd_kernel<<<dim3((unsigned)((n_pts + kPtsPerWg - 1) / kPtsPerWg), n_views), 256, kWarpLds, (hipStream_t)stream>>>(a);
    NIW_LAUNCH_CHECK("niw_warp_fwd");
    return NIW_OK;
}

extern "C" int64_t niw_warp_bwd_workspace_floats(int n_views, int64_t n_pts) {
    const long long ppad = warp_ppad(n_views, n_pts);
    return 3ll * kRowsPerBlock * ppad + 3ll * 256 * (256 * 256 + 256) + 3ll * 256 * (256 * 64 + 256);
}

// pad columns of the factor rows of niw_warp_bwd's workspace: rows, row pitch, first pad column (niw_step.hip zeroes them in its front kernel)
void niw_warp_bwd_pad_geometry(int n_views, int64_t n_pts, long long* rows, long long* ppad, long long* n_cols) {
    *rows = 3ll * kRowsPerBlock;
    *ppad = warp_ppad(n_views, n_pts);
    *n_cols = (long long)n_views * n_pts;
}

// the two halves of niw_warp_bwd, for a caller that keeps the workspace across calls (niw_step.hip): the pad columns of the factor rows
// can be zeroed off the critical path (nothing else ever writes them)
int niw_launch_warp_bwd_pad(float* workspace, int n_views, int64_t n_pts, hipStream_t st) {
    const long long ppad = warp_ppad(n_views, n_pts);
    if (const int n_pad = (int)(ppad - (long long)n_views * n_pts)) {
        const long long n_rows = 3ll * kRowsPerBlock;
        warp_zero_pad_kernel<<<(unsigned)((n_rows * n_pad + 255) / 256), 256, 0, st>>>(workspace, n_rows, ppad, (int)(ppad - n_pad), n_pad);
        NIW_LAUNCH_CHECK("niw_warp_bwd (pad)");
    }
    return NIW_OK;
}

int niw_launch_warp_bwd_main(const float* w_emb, const float* view_b, const float* w_head, const float* pts,
                             int n_views, int64_t n_pts, const float* chan_w, const float* index_window, const float* window_dev,
                             int use_index_window, const float* pt_scale_a, const float* pt_scale_b, const float* xin_saved,
                             const float* d_out, float* workspace, float* d_w_emb, float* d_view_b, float* d_w_head, float* d_pts,
                             hipStream_t st) {
    WarpArgs a{};
    int rc = fill_args(a, w_emb, view_b, w_head, pts, n_views, n_pts, chan_w, index_window, window_dev, use_index_window, pt_scale_a, pt_scale_b);
    if (rc != NIW_OK) return rc;
    NIW_REQUIRE(d_out && workspace && d_w_emb && d_view_b && d_w_head, "niw_warp_bwd: null pointer");
    NIW_REQUIRE(n_views <= 64, "niw_warp_bwd: at most 64 views per call (got %d)", n_views);
    const long long ppad = warp_ppad(n_views, n_pts);
    a.d_out = d_out; a.d_pts = d_pts; a.ws = workspace; a.ppad = ppad; a.xin_in = xin_saved;
    static std::atomic<unsigned long long> attr_set{0ull};
    if (int rc2 = niw_ensure_dynamic_lds(reinterpret_cast<const void*>(warp_bwd_kernel), kWarpLds, attr_set, "niw_warp_bwd")) return rc2;
    warp_bwd_kernel<<<dim3((unsigned)((n_pts + kPtsPerWg - 1) / kPtsPerWg), n_views), 256, kWarpLds, st>>>(a);
    NIW_LAUNCH_CHECK("niw_warp_bwd");
    float* p1 = workspace + 3ll * kRowsPerBlock * ppad;
    float* p2 = p1 + 3ll * 256 * (256 * 256 + 256);
    const long long stride = (long long)kRowsPerBlock * ppad;
    int ns1 = 0, ns2 = 0, tk2 = 64;
    // up to 2048 points (a rank's window of views): both GEMM families as ONE launch of six products (NIW_WARP_GEMM_PAIRS=<max slices of 32
    // points>, 0 = never: diagnostic) -- the chain of small launches behind the warp backward closes the iteration of a 1/8 share
    static const long long pairs_max = [] { const char* e = getenv("NIW_WARP_GEMM_PAIRS"); return e ? atoll(e) : 64ll; }();
    if (ppad / 32 <= pairs_max) {
        NiwGemmOperand A[6], B[6];
        int bias[6];
        for (int b = 0; b < 3; ++b) {
            A[b] = NiwGemmOperand{workspace + kRowGa * ppad + b * stride, 256, 0, ppad};
            B[b] = NiwGemmOperand{workspace + kRowEa * ppad + b * stride, 128, 0, ppad};
            bias[b] = 0;
            A[3 + b] = NiwGemmOperand{workspace + kRowHa * ppad + b * stride, 256, 0, ppad};
            B[3 + b] = NiwGemmOperand{workspace + kRowGo * ppad + b * stride, 4, 0, ppad};
            bias[3 + b] = 2;
        }
        rc = niw_launch_nt_gemm_pairs(6, A, B, bias, ppad, p1, &ns1, st);
        if (rc != NIW_OK) return rc;
        ns2 = ns1; tk2 = 256;
        p2 = p1 + 3ll * ns1 * (256 * 256 + 256);          // (six products x <= 42 ranges: inside the first family's 3 x 256 tiles)
    } else {
        rc = niw_launch_nt_gemm(1, NiwGemmOperand{workspace + kRowGa * ppad, 256, stride, ppad},
                                NiwGemmOperand{workspace + kRowEa * ppad, 128, stride, ppad}, ppad, 3, p1, 0, &ns1, st);
        if (rc != NIW_OK) return rc;
        rc = niw_launch_nt_gemm(0, NiwGemmOperand{workspace + kRowHa * ppad, 256, stride, ppad},
                                NiwGemmOperand{workspace + kRowGo * ppad, 4, stride, ppad}, ppad, 3, p2, 2, &ns2, st);
        if (rc != NIW_OK) return rc;
    }
    const int n_out = kHid * kSa + kHid * kSb + 2 * kHid * n_views + 4 * kHid + 4;
    warp_reduce_kernel<<<dim3((n_out + 255) / 256, 3), 256, 0, st>>>(p1, ns1, p2, ns2, tk2, n_views, d_w_emb, d_view_b, d_w_head);
    NIW_LAUNCH_CHECK("niw_warp_bwd (reduce)");
    return NIW_OK;
}

extern "C" int niw_warp_bwd(const float* w_emb, const float* view_b, const float* w_head, const float* pts,
                            int n_views, int64_t n_pts, const float* chan_w, const float* index_window, const float* window_dev,
                            int use_index_window, const float* pt_scale_a, const float* pt_scale_b, const float* xin_saved,
                            const float* d_out, float* workspace, float* d_w_emb, float* d_view_b, float* d_w_head, float* d_pts,
                            niw_stream_t stream) {
    NIW_REQUIRE(workspace && n_views > 0 && n_pts > 0, "niw_warp_bwd: null workspace or empty input");
    // padded columns of the factor rows must be zero
    if (int rc = niw_launch_warp_bwd_pad(workspace, n_views, n_pts, (hipStream_t)stream)) return rc;
    return niw_launch_warp_bwd_main(w_emb, view_b, w_head, pts, n_views, n_pts, chan_w, index_window, window_dev, use_index_window, pt_scale_a, pt_scale_b,
                                    xin_saved, d_out, workspace, d_w_emb, d_view_b, d_w_head, d_pts, (hipStream_t)stream);
}
