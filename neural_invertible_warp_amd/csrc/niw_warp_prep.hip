// NVP warp: per-parameter / per-view operand preparation of DeformNetwork (reference
// model/nvp/nvp_ndr.py) and its backward.
//
//   weight norm          w = g * v / ||v||_row                        (nvp_ndr.py:291-292, nn.utils.weight_norm)
//   code projection      code_b = lin_c(code) + code                  (nvp_ndr.py:381)
//   latent folding       view_b[v][b][part][u] = w[u, E:] . code_b[v] + bias[u]
//                        (the latent half of lin{b}_{a,b}_0 applied to the per-view code, nvp_ndr.py:416-420, 433-437)
//   w_emb = w[:, :E],  w_head = (lin{b}_a_1, lin{b}_b_1) verbatim
//
// These are O(parameters) = 166k MACs x views, i.e. latency sized; as ~150 separate torch kernels (forward +
// autograd) they made small configurations host-dispatch bound.  Here: two launches forward, three backward,
// every reduction a wave-wide butterfly over a coalesced row (one wave per weight row), every cross-workgroup
// sum a fixed-order pass over partials (deterministic, no float atomics).
//
// Flat parameter layout (= DeformNetwork.parameters() order, reference and host mirror alike: old-style
// weight_norm leaves `bias` registered before `weight_g` / `weight_v`; 165,900 floats):
//   for b in 0..2: lin{b}_a_0.bias[128] .weight_g[128] .weight_v[128x154]  lin{b}_a_1.weight[128] .bias[1]
//   for b in 0..2: lin{b}_b_0.bias[128] .weight_g[128] .weight_v[128x141]  lin{b}_b_1.weight[3x128] .bias[3]
//   for b in 0..2: lin{b}_c.weight[128x128] .bias[128]
#include "niw_common.h"
#include "niw_warp_prep_device.h"

namespace {

using namespace niw_warp_prep;
constexpr int kSa = 28, kSb = 16;                                       // row pitch of w_emb / d_w_emb (26 / 13 columns + pad)
constexpr int kWembBlock = kHid * (kSa + kSb), kHeadBlock = kHid + 1 + 3 * kHid + 3;
static_assert(3 * kWembBlock == NIW_WARP_WEMB_FLOATS, "w_emb layout");
constexpr int kMaxViews = 64;
constexpr int kGroups = 8, kRowsPerWave = 4;            // a workgroup (4 waves) owns 16 of the 128 rows of one first layer
constexpr int kParts = 2 * kGroups * 4;                 // d(code_b) partials per coupling block: (part, group, wave)
static_assert(kGroups * 4 * kRowsPerWave == kHid, "row decomposition");

struct Layer {                 // first layer of part a / b of block b inside the flat buffer
    int g, v, bias, head, E, K, nhead, S;     // S: row pitch of this part inside w_emb
};
__device__ __forceinline__ Layer layer_of(int b, int part) {
    Layer l;
    if (part == 0) { l.bias = b * kBlkA; l.E = kEa; l.K = kKa; l.nhead = kHid + 1; l.S = kSa; }
    else           { l.bias = kOffB + b * kBlkB; l.E = kEb; l.K = kKb; l.nhead = 3 * kHid + 3; l.S = kSb; }
    l.g = l.bias + kHid;
    l.v = l.g + kHid;
    l.head = l.v + kHid * l.K;
    return l;
}

// code_b[b][v][j] = bc[j] + code[v][j] + sum_k Wc[j][k] code[v][k];  grid (3, B), one wave per 32 rows j
__global__ __launch_bounds__(256) void warp_prep_code_kernel(const float* __restrict__ P, const float* __restrict__ code, int B,
                                                             float* __restrict__ codeb) {
    code_projection(blockIdx.x, blockIdx.y, P, code, B, codeb);
}

struct RowCtx { int b, part, u0; Layer l; };
__device__ __forceinline__ RowCtx row_ctx() {           // grid 3 * 2 * kGroups; wave w owns rows u0 .. u0+3
    RowCtx c;
    c.b = blockIdx.x / (2 * kGroups);
    c.part = (blockIdx.x / kGroups) & 1;
    c.u0 = (blockIdx.x % kGroups) * 16 + (threadIdx.x >> 6) * kRowsPerWave;
    c.l = layer_of(c.b, c.part);
    return c;
}

__global__ __launch_bounds__(256) void warp_prep_fwd_kernel(const float* __restrict__ P, const float* __restrict__ codeb, int B,
                                                            float* __restrict__ w_emb, float* __restrict__ view_b, float* __restrict__ w_head) {
    __shared__ float cb[kMaxViews * kLat];
    const RowCtx c = row_ctx();
    const Layer l = c.l;
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < B * kLat; i += blockDim.x) cb[i] = codeb[(long long)c.b * B * kLat + i];
    __syncthreads();
    float* we = w_emb + c.b * kWembBlock + (c.part ? kHid * kSa : 0);
    // the four rows of the wave side by side: their loads are in flight together and their butterflies interleave
    float e[kRowsPerWave], y0[kRowsPerWave], y1[kRowsPerWave], s[kRowsPerWave], bias[kRowsPerWave];
#pragma unroll
    for (int r = 0; r < kRowsPerWave; ++r) {
        const float* vrow = P + l.v + (c.u0 + r) * l.K;
        e[r] = lane < l.E ? vrow[lane] : 0.f;                             // embedding columns
        y0[r] = vrow[l.E + lane]; y1[r] = vrow[l.E + 64 + lane];         // latent columns
        s[r] = P[l.g + c.u0 + r];
        bias[r] = P[l.bias + c.u0 + r];
    }
#pragma unroll
    for (int r = 0; r < kRowsPerWave; ++r) {
        s[r] = s[r] / sqrtf(wave_sum(e[r] * e[r] + y0[r] * y0[r] + y1[r] * y1[r]));
        if (lane < l.S) we[(c.u0 + r) * l.S + lane] = lane < l.E ? e[r] * s[r] : 0.f;       // pad columns zero
    }
    for (int v = 0; v < B; ++v) {
        const float c0 = cb[v * kLat + lane], c1 = cb[v * kLat + 64 + lane];
        float d[kRowsPerWave];
#pragma unroll
        for (int r = 0; r < kRowsPerWave; ++r) d[r] = y0[r] * c0 + y1[r] * c1;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
            for (int r = 0; r < kRowsPerWave; ++r) d[r] += __shfl_xor(d[r], o);
        }
        if (lane < kRowsPerWave) {
            float dv = d[0], sv = s[0], bv = bias[0];
#pragma unroll
            for (int r = 1; r < kRowsPerWave; ++r) { dv = lane == r ? d[r] : dv; sv = lane == r ? s[r] : sv; bv = lane == r ? bias[r] : bv; }
            view_b[((v * 3 + c.b) * 2 + c.part) * kHid + c.u0 + lane] = bv + sv * dv;
        }
    }
    if (blockIdx.x % kGroups == 0) {
        float* wh = w_head + c.b * kHeadBlock + (c.part ? kHid + 1 : 0);
        for (int i = threadIdx.x; i < l.nhead; i += blockDim.x) wh[i] = P[l.head + i];
    }
}

// Weight-norm backward of the first layers + this wave's share of d(code_b)  (partials [b][kParts][B][128]).
__global__ __launch_bounds__(256) void warp_prep_bwd_kernel(const float* __restrict__ P, const float* __restrict__ codeb, int B,
                                                            const float* __restrict__ d_w_emb, const float* __restrict__ d_view_b,
                                                            const float* __restrict__ d_w_head, float* __restrict__ dP,
                                                            float* __restrict__ partial) {
    __shared__ float cb[kMaxViews * kLat];
    const RowCtx c = row_ctx();
    const Layer l = c.l;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < B * kLat; i += blockDim.x) cb[i] = codeb[(long long)c.b * B * kLat + i];
    __syncthreads();
    const float* dvb = d_view_b + (c.b * 2 + c.part) * kHid;             // + v * 3*2*128 + u
    // the four rows of the wave side by side (like the forward): every global load of the wave is in flight before the first butterfly --
    // row after row, each row waited for its own loads and the kernel, which sits on the chain of small launches that ends a rank's
    // iteration, took 13-15 us for 154 k multiply-adds.  Same sums in the same order.
    float sr[kRowsPerWave], y0r[kRowsPerWave], y1r[kRowsPerWave];
    float e[kRowsPerWave], de[kRowsPerWave], gq[kRowsPerWave], dwl0[kRowsPerWave], dwl1[kRowsPerWave], db[kRowsPerWave];
#pragma unroll
    for (int r = 0; r < kRowsPerWave; ++r) {
        const int u = c.u0 + r;
        const float* vrow = P + l.v + u * l.K;
        const float* dwe = d_w_emb + c.b * kWembBlock + (c.part ? kHid * kSa : 0) + u * l.S;
        e[r] = lane < l.E ? vrow[lane] : 0.f;
        de[r] = lane < l.E ? dwe[lane] : 0.f;
        y0r[r] = vrow[l.E + lane];
        y1r[r] = vrow[l.E + 64 + lane];
        gq[r] = P[l.g + u];
        dwl0[r] = dwl1[r] = db[r] = 0.f;
    }
    for (int v = 0; v < B; ++v) {                                       // latent half of dW:  dwl[k] = sum_v d_view_b[v][u] * code_b[v][k]
        const float c0 = cb[v * kLat + lane], c1 = cb[v * kLat + 64 + lane];
#pragma unroll
        for (int r = 0; r < kRowsPerWave; ++r) {
            const float t = dvb[v * 6 * kHid + c.u0 + r];
            dwl0[r] += t * c0;
            dwl1[r] += t * c1;
            db[r] += t;
        }
    }
    float n2[kRowsPerWave], dot[kRowsPerWave];
#pragma unroll
    for (int r = 0; r < kRowsPerWave; ++r) {
        n2[r] = e[r] * e[r] + y0r[r] * y0r[r] + y1r[r] * y1r[r];
        dot[r] = de[r] * e[r] + dwl0[r] * y0r[r] + dwl1[r] * y1r[r];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int r = 0; r < kRowsPerWave; ++r) {
            n2[r] += __shfl_xor(n2[r], o);
            dot[r] += __shfl_xor(dot[r], o);
        }
    }
#pragma unroll
    for (int r = 0; r < kRowsPerWave; ++r) {
        const int u = c.u0 + r;
        const float n = sqrtf(n2[r]), g = gq[r], s = g / n, coef = g * dot[r] / (n2[r] * n);
        float* drow = dP + l.v + u * l.K;                                // d weight_v = s dW - (g/n^3)(dW.v) v
        if (lane < l.E) drow[lane] = s * de[r] - coef * e[r];
        drow[l.E + lane] = s * dwl0[r] - coef * y0r[r];
        drow[l.E + 64 + lane] = s * dwl1[r] - coef * y1r[r];
        if (lane == 0) {
            dP[l.g + u] = dot[r] / n;                                    // d weight_g
            dP[l.bias + u] = db[r];
        }
        sr[r] = s;
    }
    // d code_b[v][k] (this wave's 4 rows) = sum_r d_view_b[v][u_r] * s_r * V[u_r][E+k]
    float* out = partial + (((long long)c.b * kParts + (c.part * kGroups + blockIdx.x % kGroups) * 4 + wave) * B) * kLat;
    for (int v = 0; v < B; ++v) {
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int r = 0; r < kRowsPerWave; ++r) {
            const float t = dvb[v * 6 * kHid + c.u0 + r] * sr[r];
            a0 += t * y0r[r];
            a1 += t * y1r[r];
        }
        out[v * kLat + lane] = a0;
        out[v * kLat + 64 + lane] = a1;
    }
    if (blockIdx.x % kGroups == 0) {
        const float* dh = d_w_head + c.b * kHeadBlock + (c.part ? kHid + 1 : 0);
        for (int i = threadIdx.x; i < l.nhead; i += blockDim.x) dP[l.head + i] = dh[i];
    }
}

__device__ __forceinline__ float sum_parts(const float* __restrict__ p, long long stride) {   // fixed order over kParts partials
    // all 64 loads in flight before the first add (this kernel closes the chain of small launches that ends a rank's iteration: rolled,
    // it paid sixteen dependent round trips); the additions keep their order
    float x[kParts];
#pragma unroll
    for (int w = 0; w < kParts; ++w) x[w] = p[w * stride];
    float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < kParts; w += 4) {
#pragma unroll
        for (int k = 0; k < 4; ++k) s[k] += x[w + k];
    }
    return (s[0] + s[1]) + (s[2] + s[3]);
}

// Code projection backward.  Workgroups 0..B-1: d code[v];  workgroups B..B+23: 16 rows of d lin{b}_c.
__global__ __launch_bounds__(256) void warp_prep_bwd_code_kernel(const float* __restrict__ P, const float* __restrict__ code, int B,
                                                                 const float* __restrict__ partial, float* __restrict__ dP,
                                                                 float* __restrict__ d_code) {
    __shared__ float sh[kMaxViews * 16 > 3 * kLat ? kMaxViews * 16 : 3 * kLat];
    const int tid = threadIdx.x;
    const long long pstride = (long long)B * kLat;
    if ((int)blockIdx.x < B) {
        const int v = blockIdx.x;
        for (int i = tid; i < 3 * kLat; i += blockDim.x) {                // d code_b[b][v][j], all three blocks
            const int b = i / kLat, j = i % kLat;
            sh[i] = sum_parts(partial + (long long)b * kParts * pstride + v * kLat + j, pstride);
        }
        __syncthreads();
        if (tid < kLat) {
            float acc = 0.f;
            for (int b = 0; b < 3; ++b) {
                const float* Wc = P + kOffC + b * kBlkC;
                float a = sh[b * kLat + tid];                             // identity path of code_b = lin_c(code) + code
#pragma unroll 1
                for (int j0 = 0; j0 < kLat; j0 += 32) {                   // 32 weight loads in flight per trip, same order of additions
                    float wc[32];
#pragma unroll
                    for (int jj = 0; jj < 32; ++jj) wc[jj] = Wc[(j0 + jj) * kLat + tid];
#pragma unroll
                    for (int jj = 0; jj < 32; ++jj) a += sh[b * kLat + j0 + jj] * wc[jj];
                }
                acc += a;
            }
            d_code[v * kLat + tid] = acc;
        }
        return;
    }
    const int idx = blockIdx.x - B, b = idx / 8, j0 = (idx % 8) * 16;
    for (int i = tid; i < B * 16; i += blockDim.x) {                      // d code_b[b][v][j0 + jj]
        const int v = i / 16, jj = i % 16;
        sh[i] = sum_parts(partial + (long long)b * kParts * pstride + v * kLat + j0 + jj, pstride);
    }
    __syncthreads();
    const int oc = kOffC + b * kBlkC;
    for (int i = tid; i < 16 * kLat; i += blockDim.x) {                   // d Wc[j][k] = sum_v d code_b[v][j] code[v][k]
        const int jj = i / kLat, k = i % kLat;
        float acc = 0.f;
        for (int v = 0; v < B; ++v) acc += sh[v * 16 + jj] * code[v * kLat + k];
        dP[oc + (j0 + jj) * kLat + k] = acc;
    }
    if (tid < 16) {
        float acc = 0.f;
        for (int v = 0; v < B; ++v) acc += sh[v * 16 + tid];
        dP[oc + kLat * kLat + j0 + tid] = acc;
    }
}

}  // namespace

extern "C" int64_t niw_warp_prep_fwd_workspace_floats(int n_views) { return 3ll * n_views * kLat; }
extern "C" int64_t niw_warp_prep_bwd_workspace_floats(int n_views) { return (3ll + 3ll * kParts) * n_views * kLat; }

extern "C" int niw_warp_prep_fwd(const float* params, const float* code, int n_views, float* workspace, float* w_emb, float* view_b,
                                 float* w_head, niw_stream_t stream) {
    NIW_REQUIRE(params && code && workspace && w_emb && view_b && w_head, "niw_warp_prep_fwd: null pointer");
    NIW_REQUIRE(n_views > 0 && n_views <= kMaxViews, "niw_warp_prep_fwd: 1..%d views per call (got %d)", kMaxViews, n_views);
    hipStream_t st = (hipStream_t)stream;
    warp_prep_code_kernel<<<dim3(3, n_views), 256, 0, st>>>(params, code, n_views, workspace);
    NIW_LAUNCH_CHECK("niw_warp_prep_fwd (code projection)");
    warp_prep_fwd_kernel<<<3 * 2 * kGroups, 256, 0, st>>>(params, workspace, n_views, w_emb, view_b, w_head);
    NIW_LAUNCH_CHECK("niw_warp_prep_fwd");
    return NIW_OK;
}

// the forward without its first launch: `workspace` already holds the code projection (the train iteration's front kernel made it)
int niw_launch_warp_prep_fwd_main(const float* params, int n_views, const float* workspace, float* w_emb, float* view_b, float* w_head, hipStream_t st) {
    NIW_REQUIRE(params && workspace && w_emb && view_b && w_head, "niw_warp_prep_fwd: null pointer");
    NIW_REQUIRE(n_views > 0 && n_views <= kMaxViews, "niw_warp_prep_fwd: 1..%d views per call (got %d)", kMaxViews, n_views);
    warp_prep_fwd_kernel<<<3 * 2 * kGroups, 256, 0, st>>>(params, workspace, n_views, w_emb, view_b, w_head);
    NIW_LAUNCH_CHECK("niw_warp_prep_fwd");
    return NIW_OK;
}

// codeb_ready: the code projection niw_warp_prep_fwd left at the head of ITS workspace for the same params / code (niw_step.hip keeps
// it across the iteration), or NULL: recomputed into this call's workspace
int niw_launch_warp_prep_bwd(const float* params, const float* code, int n_views, const float* d_w_emb, const float* d_view_b,
                             const float* d_w_head, float* workspace, const float* codeb_ready, float* d_params, float* d_code, hipStream_t st) {
    NIW_REQUIRE(params && code && d_w_emb && d_view_b && d_w_head && workspace && d_params && d_code, "niw_warp_prep_bwd: null pointer");
    NIW_REQUIRE(n_views > 0 && n_views <= kMaxViews, "niw_warp_prep_bwd: 1..%d views per call (got %d)", kMaxViews, n_views);
    const float* codeb = codeb_ready;
    float* partial = workspace + 3ll * n_views * kLat;
    if (!codeb) {
        warp_prep_code_kernel<<<dim3(3, n_views), 256, 0, st>>>(params, code, n_views, workspace);
        NIW_LAUNCH_CHECK("niw_warp_prep_bwd (code projection)");
        codeb = workspace;
    }
    warp_prep_bwd_kernel<<<3 * 2 * kGroups, 256, 0, st>>>(params, codeb, n_views, d_w_emb, d_view_b, d_w_head, d_params, partial);
    NIW_LAUNCH_CHECK("niw_warp_prep_bwd");
    warp_prep_bwd_code_kernel<<<n_views + 24, 256, 0, st>>>(params, code, n_views, partial, d_params, d_code);
    NIW_LAUNCH_CHECK("niw_warp_prep_bwd (code projection backward)");
    return NIW_OK;
}

extern "C" int niw_warp_prep_bwd(const float* params, const float* code, int n_views, const float* d_w_emb, const float* d_view_b,
                                 const float* d_w_head, float* workspace, float* d_params, float* d_code, niw_stream_t stream) {
    return niw_launch_warp_prep_bwd(params, code, n_views, d_w_emb, d_view_b, d_w_head, workspace, nullptr, d_params, d_code, (hipStream_t)stream);
}
