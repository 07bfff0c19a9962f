// NVP warp: per-parameter / per-view operand preparation of DeformNetwork (reference
// model/nvp/nvp_ndr.py) and its backward, fused into one launch each.
//
//   weight norm          w = g * v / ||v||_row                        (nvp_ndr.py:291-292, nn.utils.weight_norm)
//   code projection      code_b = lin_c(code) + code                  (nvp_ndr.py:381)
//   latent folding       view_b[v][b][part][u] = w[u, E:] . code_b[v] + bias[u]
//                        (the latent half of lin{b}_{a,b}_0 applied to the per-view code, nvp_ndr.py:416-420, 433-437)
//   w_emb = w[:, :E],  w_head = (lin{b}_a_1, lin{b}_b_1) verbatim
//
// These are O(parameters) = 166k MACs x views, i.e. launch-latency sized; as ~150 separate torch
// kernels (forward + autograd) they made small configurations host-dispatch bound.
//
// Flat parameter layout (= DeformNetwork.parameters() order, reference and host mirror alike: old-style
// weight_norm leaves `bias` registered before `weight_g` / `weight_v`; 165,900 floats):
//   for b in 0..2: lin{b}_a_0.bias[128] .weight_g[128] .weight_v[128x154]  lin{b}_a_1.weight[128] .bias[1]
//   for b in 0..2: lin{b}_b_0.bias[128] .weight_g[128] .weight_v[128x141]  lin{b}_b_1.weight[3x128] .bias[3]
//   for b in 0..2: lin{b}_c.weight[128x128] .bias[128]
#include "niw_common.h"

namespace {

constexpr int kHid = 128, kLat = 128, kEa = 26, kEb = 13;
constexpr int kKa = kEa + kLat, kKb = kEb + kLat;                       // 154, 141
constexpr int kBlkA = kHid + kHid * kKa + kHid + kHid + 1;              // 20097
constexpr int kBlkB = kHid + kHid * kKb + kHid + 3 * kHid + 3;          // 18691
constexpr int kBlkC = kLat * kLat + kLat;                               // 16512
constexpr int kOffB = 3 * kBlkA, kOffC = kOffB + 3 * kBlkB;
static_assert(kOffC + 3 * kBlkC == NIW_WARP_PARAM_FLOATS, "flat warp parameter count");
constexpr int kWembBlock = kHid * (kEa + kEb), kHeadBlock = kHid + 1 + 3 * kHid + 3;
constexpr int kMaxViews = 64;

struct Layer {                 // first layer of part a / b of block b inside the flat buffer
    int g, v, bias, head, E, K, nhead;
};
__device__ __forceinline__ Layer layer_of(int b, int part) {
    Layer l;
    if (part == 0) { l.bias = b * kBlkA; l.E = kEa; l.K = kKa; l.nhead = kHid + 1; }
    else           { l.bias = kOffB + b * kBlkB; l.E = kEb; l.K = kKb; l.nhead = 3 * kHid + 3; }
    l.g = l.bias + kHid;
    l.v = l.g + kHid;
    l.head = l.v + kHid * l.K;
    return l;
}

// code_b[v][j] = bc[j] + code[v][j] + sum_k Wc[j][k] code[v][k]  -> LDS
__device__ __forceinline__ void project_code(const float* __restrict__ P, const float* __restrict__ code, int B, int b, float* codeb) {
    const float* Wc = P + kOffC + b * kBlkC;
    const float* bc = Wc + kLat * kLat;
    for (int idx = threadIdx.x; idx < B * kLat; idx += blockDim.x) {
        const int v = idx / kLat, j = idx % kLat;
        float acc = 0.f;
        for (int k = 0; k < kLat; ++k) acc += Wc[j * kLat + k] * code[v * kLat + k];
        codeb[idx] = acc + bc[j] + code[v * kLat + j];
    }
}

__global__ __launch_bounds__(256) void warp_prep_fwd_kernel(const float* __restrict__ P, const float* __restrict__ code, int B,
                                                            float* __restrict__ w_emb, float* __restrict__ view_b, float* __restrict__ w_head) {
    __shared__ float codeb[kMaxViews * kLat];
    __shared__ float scale[kHid];
    const int b = blockIdx.x >> 1, part = blockIdx.x & 1, tid = threadIdx.x;
    const Layer l = layer_of(b, part);
    project_code(P, code, B, b, codeb);
    if (tid < kHid) {
        float n2 = 0.f;
        for (int c = 0; c < l.K; ++c) { const float x = P[l.v + tid * l.K + c]; n2 += x * x; }
        scale[tid] = P[l.g + tid] / sqrtf(n2);
    }
    __syncthreads();
    float* we = w_emb + b * kWembBlock + (part ? kHid * kEa : 0);
    for (int idx = tid; idx < kHid * l.E; idx += blockDim.x) {
        const int u = idx / l.E, c = idx % l.E;
        we[idx] = P[l.v + u * l.K + c] * scale[u];
    }
    for (int idx = tid; idx < B * kHid; idx += blockDim.x) {
        const int v = idx / kHid, u = idx % kHid;
        const float* row = P + l.v + u * l.K + l.E;
        float acc = 0.f;
        for (int k = 0; k < kLat; ++k) acc += row[k] * codeb[v * kLat + k];
        view_b[((v * 3 + b) * 2 + part) * kHid + u] = P[l.bias + u] + scale[u] * acc;
    }
    float* wh = w_head + b * kHeadBlock + (part ? kHid + 1 : 0);
    for (int idx = tid; idx < l.nhead; idx += blockDim.x) wh[idx] = P[l.head + idx];
}

// One workgroup per coupling block: both first layers (weight-norm backward), the code projection and
// this block's share of d(code) (summed over the blocks by warp_prep_sum_kernel, fixed order).
__global__ __launch_bounds__(256) void warp_prep_bwd_kernel(const float* __restrict__ P, const float* __restrict__ code, int B,
                                                            const float* __restrict__ d_w_emb, const float* __restrict__ d_view_b,
                                                            const float* __restrict__ d_w_head, float* __restrict__ dP,
                                                            float* __restrict__ d_code_blk) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* codeb = lds;                               // [B][128]
    float* dcodeb = codeb + kMaxViews * kLat;         // [B][128]
    float* dwl = dcodeb + kMaxViews * kLat;           // [128 u][128 k]  latent half of dW
    float* scale = dwl + kHid * kLat;                 // [128]
    const int b = blockIdx.x, tid = threadIdx.x;
    project_code(P, code, B, b, codeb);
    for (int idx = tid; idx < B * kLat; idx += blockDim.x) dcodeb[idx] = 0.f;
    __syncthreads();
    for (int part = 0; part < 2; ++part) {
        const Layer l = layer_of(b, part);
        // latent half of dW:  dwl[u][k] = sum_v d_view_b[v][u] * code_b[v][k]
        for (int idx = tid; idx < kHid * kLat; idx += blockDim.x) {
            const int u = idx / kLat, k = idx % kLat;
            float acc = 0.f;
            for (int v = 0; v < B; ++v) acc += d_view_b[((v * 3 + b) * 2 + part) * kHid + u] * codeb[v * kLat + k];
            dwl[idx] = acc;
        }
        __syncthreads();
        if (tid < kHid) {
            const int u = tid;
            const float* vrow = P + l.v + u * l.K;
            const float* dwe = d_w_emb + b * kWembBlock + (part ? kHid * kEa : 0) + u * l.E;
            float n2 = 0.f, dot = 0.f;
            for (int c = 0; c < l.K; ++c) {
                const float x = vrow[c];
                n2 += x * x;
                dot += (c < l.E ? dwe[c] : dwl[u * kLat + c - l.E]) * x;
            }
            const float n = sqrtf(n2), g = P[l.g + u], s = g / n;
            scale[u] = s;
            dP[l.g + u] = dot / n;                                    // d weight_g
            const float coef = g * dot / (n2 * n);
            for (int c = 0; c < l.K; ++c)                             // d weight_v = s dW - (g/n^3)(dW.v) v
                dP[l.v + u * l.K + c] = s * (c < l.E ? dwe[c] : dwl[u * kLat + c - l.E]) - coef * vrow[c];
            float db = 0.f;
            for (int v = 0; v < B; ++v) db += d_view_b[((v * 3 + b) * 2 + part) * kHid + u];
            dP[l.bias + u] = db;
        }
        __syncthreads();
        // d code_b[v][k] += sum_u d_view_b[v][u] * s[u] * V[u][E+k]
        for (int idx = tid; idx < B * kLat; idx += blockDim.x) {
            const int v = idx / kLat, k = idx % kLat;
            float acc = 0.f;
            for (int u = 0; u < kHid; ++u) acc += d_view_b[((v * 3 + b) * 2 + part) * kHid + u] * scale[u] * P[l.v + u * l.K + l.E + k];
            dcodeb[idx] += acc;
        }
        const float* dh = d_w_head + b * kHeadBlock + (part ? kHid + 1 : 0);
        for (int idx = tid; idx < l.nhead; idx += blockDim.x) dP[l.head + idx] = dh[idx];
        __syncthreads();
    }
    // code projection backward
    const int oc = kOffC + b * kBlkC;
    for (int idx = tid; idx < kLat * kLat; idx += blockDim.x) {
        const int j = idx / kLat, k = idx % kLat;
        float acc = 0.f;
        for (int v = 0; v < B; ++v) acc += dcodeb[v * kLat + j] * code[v * kLat + k];
        dP[oc + idx] = acc;
    }
    for (int j = tid; j < kLat; j += blockDim.x) {
        float acc = 0.f;
        for (int v = 0; v < B; ++v) acc += dcodeb[v * kLat + j];
        dP[oc + kLat * kLat + j] = acc;
    }
    for (int idx = tid; idx < B * kLat; idx += blockDim.x) {
        const int v = idx / kLat, k = idx % kLat;
        float acc = dcodeb[idx];
        for (int j = 0; j < kLat; ++j) acc += dcodeb[v * kLat + j] * P[oc + j * kLat + k];
        d_code_blk[(b * B + v) * kLat + k] = acc;
    }
}

__global__ void warp_prep_sum_kernel(const float* __restrict__ d_code_blk, int n, float* __restrict__ d_code) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) d_code[i] = (d_code_blk[i] + d_code_blk[n + i]) + d_code_blk[2 * n + i];
}

constexpr size_t kBwdLds = (2 * kMaxViews * kLat + kHid * kLat + kHid) * sizeof(float);

}  // namespace

extern "C" int niw_warp_prep_fwd(const float* params, const float* code, int n_views, float* w_emb, float* view_b, float* w_head,
                                 niw_stream_t stream) {
    NIW_REQUIRE(params && code && w_emb && view_b && w_head, "niw_warp_prep_fwd: null pointer");
    NIW_REQUIRE(n_views > 0 && n_views <= kMaxViews, "niw_warp_prep_fwd: 1..%d views per call (got %d)", kMaxViews, n_views);
    warp_prep_fwd_kernel<<<6, 256, 0, (hipStream_t)stream>>>(params, code, n_views, w_emb, view_b, w_head);
    NIW_LAUNCH_CHECK("niw_warp_prep_fwd");
    return NIW_OK;
}

extern "C" int niw_warp_prep_bwd(const float* params, const float* code, int n_views, const float* d_w_emb, const float* d_view_b,
                                 const float* d_w_head, float* scratch, float* d_params, float* d_code, niw_stream_t stream) {
    NIW_REQUIRE(params && code && d_w_emb && d_view_b && d_w_head && scratch && d_params && d_code, "niw_warp_prep_bwd: null pointer");
    NIW_REQUIRE(n_views > 0 && n_views <= kMaxViews, "niw_warp_prep_bwd: 1..%d views per call (got %d)", kMaxViews, n_views);
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(warp_prep_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBwdLds); attr = true; }
    warp_prep_bwd_kernel<<<3, 256, kBwdLds, (hipStream_t)stream>>>(params, code, n_views, d_w_emb, d_view_b, d_w_head, d_params, scratch);
    NIW_LAUNCH_CHECK("niw_warp_prep_bwd");
    const int n = n_views * kLat;
    warp_prep_sum_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(scratch, n, d_code);
    NIW_LAUNCH_CHECK("niw_warp_prep_bwd (sum)");
    return NIW_OK;
}
