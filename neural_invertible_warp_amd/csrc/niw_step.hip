// One INN train iteration (forward + backward) as ONE call, and the Adam step of all optimizer groups as one launch.
//
// niw_train_step replaces, for the engine, what the reference runs between zero_grad and the optimizers: Graph.forward in train
// mode (model/nerf_inn_llff.py:493-546, model/nerf_inn_dtu.py:371-396) with get_pose = ray generation + NVP warp
// (model/barf_inn_llff.py:305-364, model/pose_models/inn.py:63-102), compute_loss (nerf_inn_llff.py:548-573, nerf_inn_dtu.py:398-415),
// summarize_loss (model/base.py:130-142) and loss.all.backward().  It only SEQUENCES the library's own per-stage entry points -- the
// same kernels and launch shapes the autograd mirror (ops.py) uses -- into one caller-provided workspace, and adds three small
// kernels for what the reference does as tensor glue around them:
//   split_rays_kernel   warped [grid ; centre] point sets -> contiguous rays (grid - centre) and centres   (barf_inn_llff.py:360-363)
//   combine_kernel      the sum of every gradient route into the warped points (sample points and view directions of both field
//                       networks, ray lengths of both compositing passes, the alignment residual), the zero rows of the latent
//                       gradient outside a rank's window, and loss.all = sum_k 10^w_k loss_k
//   fill_kernel         zero gradients of a network that receives no loss term
// Route sums are formed in the order autograd's accumulation would (fine compositing, fine field, coarse compositing, coarse field;
// separately rounded), so the engine's parameters agree bit for bit with the autograd mirror on the same inputs.
#include <mutex>
#include "niw_common.h"
#include "niw_loss_device.h"
#include <stdlib.h>

// launches that exist for this sequencer only (each = two launches of the public entry points in one, same arithmetic)
int niw_launch_step_front(int64_t n_pixels, uint64_t seed, uint64_t draw, const uint64_t* draw_dev, const float* intr, const float* pose, int n_views,
                          long long R, int H, int W, int64_t* ray_idx, float* stacked,
                          uint64_t depth_seed, int stratified, long long n_rays, int S, double depth_min, double depth_max, int inverse, float* depth,
                          const float* params0, const float* params1, const int32_t* index, float* packed0, float* packed1,
                          float* pad_ws, long long pad_rows, long long ppad, long long n_cols,
                          const float* warp_params, const float* code, int n_code_views, float* codeb, hipStream_t st);
int niw_launch_warp_prep_fwd_main(const float* params, int n_views, const float* workspace, float* w_emb, float* view_b, float* w_head, hipStream_t st);
void niw_warp_bwd_pad_geometry(int n_views, int64_t n_pts, long long* rows, long long* ppad, long long* n_cols);
int niw_launch_align_register(const float* target, const float* source, int n_views, int64_t n_points, double* moments, float* poses, hipStream_t st);
int niw_launch_warp_prep_bwd(const float* params, const float* code, int n_views, const float* d_w_emb, const float* d_view_b,
                             const float* d_w_head, float* workspace, const float* codeb_ready, float* d_params, float* d_code, hipStream_t st);
int niw_launch_warp_bwd_pad(float* workspace, int n_views, int64_t n_pts, hipStream_t st);
int niw_launch_warp_bwd_main(const float* w_emb, const float* view_b, const float* w_head, const float* pts,
                             int n_views, int64_t n_pts, const float* chan_w, const float* index_window, const float* window_dev,
                             int use_index_window, const float* pt_scale_a, const float* pt_scale_b, const float* xin_saved,
                             const float* d_out, float* workspace, float* d_w_emb, float* d_view_b, float* d_w_head, float* d_pts,
                             hipStream_t st);

namespace {

using niw::add_rn;
using niw::mul_rn;
using niw::sub_rn;

#ifndef NIW_CARVE_FLOATS
#define NIW_CARVE_FLOATS 64
#endif
constexpr long long kCarveFloats = NIW_CARVE_FLOATS;

struct Carve {             // hands out consecutive pieces of the workspace, each starting on a 256-byte boundary of it (two cache lines: the
    float* p;              // kernels' 16-byte vector accesses, the LDS-DMA of the fast-precision kernels and whole-line streaming stores all
    long long used = 0;    // see the alignment separate allocations had)
    float* take(long long n) {
        float* r = p ? p + used : nullptr;
        used += (n + kCarveFloats - 1) / kCarveFloats * kCarveFloats;
        return r;
    }
};

// ---------------------------------------------------------------------------------------------------------------- glue kernels
__global__ void split_rays_kernel(const float* __restrict__ warped, long long V, long long R, float* __restrict__ ray,
                                  float* __restrict__ center) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= V * R * 3) return;
    const long long v = i / (3 * R), rem = i - v * 3 * R;
    const float g = warped[v * 6 * R + rem], c = warped[v * 6 * R + 3 * R + rem];
    ray[i] = sub_rn(g, c);
    center[i] = c;
}

struct CombineArgs {
    const float* d_ray[4];      // per-ray gradients of the share [n,3], in accumulation order; NULL = route absent
    const float* d_center[2];
    const float* d_target;      // [V,2R,3] alignment gradient (rows of the owned views valid) or NULL
    float* d_warped;            // [V,2R,3]
    long long V, R, a, b;       // window views, rays per view, the share as [a, b) of the window's flattened rays
    int own_lo, own_hi;         // owned views, window-local
    float w_align;
    // latent rows outside the window
    float* d_latent;
    long long lat_lo, lat_hi, lat_n;   // floats [lat_lo, lat_hi) are the window's rows of the lat_n-float table
    // loss total
    float* loss;
    float w[3];
    int present[3];             // term k was computed by the LOSS stage (else it is written as 0 here)
    // photometric terms whose residuals the one-launch compositing kernels left behind: their value is formed HERE (workgroup 0), in
    // mse_kernel's order -- NULL: the term came from niw_mse_fwd_bwd
    const float* resid[2];
    long long resid_n;          // residuals per term (3 per ray)
    double n_norm;
};

__global__ void combine_kernel(CombineArgs a) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < a.V * a.R * 3) {
        const long long v = i / (3 * a.R), rem = i - v * 3 * a.R, q = i / 3;
        float g = 0.f, c = 0.f;
        if (q >= a.a && q < a.b) {
            const long long j = i - a.a * 3;
            float dr = 0.f, dc = 0.f;
            bool first = true;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (a.d_ray[k]) { dr = first ? a.d_ray[k][j] : add_rn(dr, a.d_ray[k][j]); first = false; }
            first = true;
#pragma unroll
            for (int k = 0; k < 2; ++k)
                if (a.d_center[k]) { dc = first ? a.d_center[k][j] : add_rn(dc, a.d_center[k][j]); first = false; }
            g = dr;                       // d grid   = d ray
            c = sub_rn(dc, dr);           // d centre = d centre - d ray        (ray = grid - centre)
        }
        if (a.d_target && v >= a.own_lo && v < a.own_hi) {
            g = add_rn(g, mul_rn(a.d_target[v * 6 * a.R + rem], a.w_align));
            c = add_rn(c, mul_rn(a.d_target[v * 6 * a.R + 3 * a.R + rem], a.w_align));
        }
        a.d_warped[v * 6 * a.R + rem] = g;
        a.d_warped[v * 6 * a.R + 3 * a.R + rem] = c;
    }
    if (a.d_latent && i < a.lat_n && (i < a.lat_lo || i >= a.lat_hi)) a.d_latent[i] = 0.f;
    if (blockIdx.x == 0 && (a.resid[0] || a.resid[1])) {
        __shared__ double red[16];
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (a.resid[k]) {                                             // (uniform over the workgroup)
                const double t = niw::sq_sum_in_mse_order(a.resid[k], a.resid_n, red);
                if (threadIdx.x == 0) a.loss[k] = (float)(t / a.n_norm);
            }
    }
    if (i == 0) {
        float total = 0.f;
        bool first = true;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (!a.present[k]) { a.loss[k] = 0.f; continue; }      // absent term (weight `null`, no fine pass, no owned view): reported as 0
            total = first ? mul_rn(a.loss[k], a.w[k]) : fmaf(a.w[k], a.loss[k], total);
            first = false;
        }
        a.loss[3] = total;
    }
}

// warp + NDC: the gradient routes into the NDC rays of the share, summed in autograd's accumulation order over the WHOLE window (zero
// outside the share), as the two operands of niw_convert_ndc_bwd
struct NdcRoutesArgs {
    const float* d_ray[4];
    const float* d_center[2];
    float* g_ray;               // [V R 3]
    float* g_center;
    long long total, a3, b3;    // window floats; the share as floats [a3, b3)
};
__global__ void ndc_routes_kernel(NdcRoutesArgs a) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.total) return;
    float dr = 0.f, dc = 0.f;
    if (i >= a.a3 && i < a.b3) {
        const long long j = i - a.a3;
        bool first = true;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (a.d_ray[k]) { dr = first ? a.d_ray[k][j] : add_rn(dr, a.d_ray[k][j]); first = false; }
        first = true;
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (a.d_center[k]) { dc = first ? a.d_center[k][j] : add_rn(dc, a.d_center[k][j]); first = false; }
    }
    a.g_ray[i] = dr;
    a.g_center[i] = dc;
}

__global__ void fill_kernel(float* __restrict__ p, long long n, float v) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

int fill(float* p, long long n, float v, hipStream_t st) {
    fill_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(p, n, v);
    NIW_LAUNCH_CHECK("niw_train_step (fill)");
    return NIW_OK;
}

// ---------------------------------------------------------------------------------------------------------------- Adam, all groups
constexpr int kMaxAdamGroups = 8;
struct AdamGroupDev {
    float* p; const float* g; float* m; float* v;
    long long n;
    float step_size, bc2_sqrt;
    int first_block;
};
struct AdamBatch {
    AdamGroupDev g[kMaxAdamGroups];
    int n_groups;
    float w1, b2, w2, eps;
};

__global__ void adam_multi_kernel(AdamBatch b, const float* __restrict__ hyper_dev) {
    int gi = 0;
#pragma unroll
    for (int k = 1; k < kMaxAdamGroups; ++k)
        if (k < b.n_groups && (int)blockIdx.x >= b.g[k].first_block) gi = k;
    const AdamGroupDev& G = b.g[gi];
    const long long i = (long long)((int)blockIdx.x - G.first_block) * blockDim.x + threadIdx.x;
    if (i >= G.n) return;
    float step_size = G.step_size, bc2_sqrt = G.bc2_sqrt;
    if (hyper_dev) { step_size = hyper_dev[2 * gi]; bc2_sqrt = hyper_dev[2 * gi + 1]; }
    // the arithmetic of adam_kernel (niw_sampling.hip) = torch.optim.Adam's single-tensor path, element for element
    float pi = G.p[i], mi = G.m[i], vi = G.v[i];
    niw::adam_update(pi, G.g[i], mi, vi, b.w1, b.b2, b.w2, b.eps, step_size, bc2_sqrt);
    G.p[i] = pi; G.m[i] = mi; G.v[i] = vi;
}

// ---------------------------------------------------------------------------------------------------------------- second stream
// The small stages that do not depend on each other run beside the field-MLP kernels (niw.h: niw_train_desc.overlap).  One
// non-blocking stream and three events per device, created on first use and kept for the life of the process (like the
// kernel-attribute cache of niw_common.h: process-global, write-once per device).
struct SideLane {
    hipStream_t s = nullptr;
    hipEvent_t warped = nullptr, dx = nullptr, join = nullptr;
    std::mutex mu;      // one niw_train_step at a time per device between its first fork and its join (the events are shared)
};

// Holds the lane for the length of one call; a call that returns early (a failed launch behind the first fork) still joins the
// lane to the caller's stream -- a stream left forked invalidates a capture in progress.
struct SideLaneHold {
    SideLane* lane;
    hipStream_t st;
    bool forked = false;
    SideLaneHold(SideLane* l, hipStream_t s) : lane(l), st(s) { if (lane) lane->mu.lock(); }
    ~SideLaneHold() {
        if (!lane) return;
        if (forked) {
            (void)hipEventRecord(lane->join, lane->s);
            (void)hipStreamWaitEvent(st, lane->join, 0);
        }
        lane->mu.unlock();
    }
};

SideLane* side_lane() {
    static SideLane lanes[64];
    static std::atomic<unsigned long long> ready{0ull};
    static std::atomic_flag busy = ATOMIC_FLAG_INIT;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    const unsigned long long bit = 1ull << dev;
    if (ready.load(std::memory_order_acquire) & bit) return &lanes[dev];
    while (busy.test_and_set(std::memory_order_acquire)) {}
    bool ok = true;
    if (!(ready.load(std::memory_order_acquire) & bit)) {
        SideLane& L = lanes[dev];
        // NORMAL priority.  Measured (round 4, cfg3 through a one-rank RCCL group): a high-priority second stream created AFTER the
        // communicator's own (high-priority) stream cost the iteration 0.65 ms (7.08 vs 6.41 ms; created before it, or with normal
        // priority either way: 6.41-6.44 ms); without a communicator the two priorities time alike (NIW_SIDE_PRIORITY=h|l: diagnostic)
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);              // hi = numerically lowest = greatest priority
        const char* pr = getenv("NIW_SIDE_PRIORITY");
        const int prio = (pr && pr[0] == 'h') ? hi : (pr && pr[0] == 'l') ? lo : 0;
        ok = hipStreamCreateWithPriority(&L.s, hipStreamNonBlocking, prio) == hipSuccess;
        for (hipEvent_t* e : {&L.warped, &L.dx, &L.join})
            ok = ok && hipEventCreateWithFlags(e, hipEventDisableTiming) == hipSuccess;
        if (ok) ready.fetch_or(bit, std::memory_order_release);
    }
    busy.clear(std::memory_order_release);
    return ok ? &lanes[dev] : nullptr;
}

#define NIW_HIP(call, what)                                                            \
    do {                                                                               \
        const hipError_t e_ = (call);                                                  \
        if (e_ != hipSuccess) {                                                        \
            niw_set_error("niw_train_step: %s: %s", what, hipGetErrorString(e_));      \
            return NIW_ERR_LAUNCH;                                                     \
        }                                                                              \
    } while (0)

// ---------------------------------------------------------------------------------------------------------------- workspace layout
struct Layout {
    long long V, R, n, S, T, mpad_c, mpad_f;
    int n_own;
    float *ray_idx, *stacked_in, *warped, *xin, *w_emb, *view_b, *w_head, *prep_ws, *ray, *center;
    float *z, *rgb_s, *sigma_s, *prob, *rgb, *depth, *opacity;
    float *z_all, *rgb_f, *sigma_f, *rgb_fine, *depth_fine, *opacity_fine;
    float *packed_c, *packed_f, *save_c, *save_f, *gradws, *partial;
    float *d_rgb, *d_rgb_f, *d_rgb_s, *d_sigma_s, *comp_d_ray_c, *comp_d_ray_f, *mlp_d_c, *mlp_d_f;   // mlp_d_*: [2][n][3] = {d_center, d_ray}
    float *noise_c, *noise_f;        // density noise of the two passes (density_noise > 0)
    float *ray_cam, *center_cam;     // warp + NDC: the camera-frame rays the warp produced (L.ray / L.center then hold their NDC form)
    float *ndc_g;                    // warp + NDC: [4][V R 3] summed gradient routes of the NDC rays {d ray', d centre'}, then {d ray, d centre}
    float *resid_c, *resid_f, *d_rgb_s_c, *d_sigma_s_c;   // one-launch compositing + loss + backward: residuals [n,3]; the coarse pass's own sample gradients
    float *mom, *poses, *d_target, *d_warped, *warp_ws, *d_w_emb, *d_view_b, *d_w_head, *prep_bwd_ws;
    long long total;
};

int check_desc(const niw_train_desc* d) {
    NIW_REQUIRE(d, "niw_train_step: null descriptor");
    NIW_REQUIRE(d->image && d->intr && d->nerf_params, "niw_train_step: batch and field parameters are required");
    NIW_REQUIRE(d->loss && d->d_nerf, "niw_train_step: loss and gradient outputs are required");
    if (d->warp_params) {
        NIW_REQUIRE(d->latent && d->d_warp && d->d_latent, "niw_train_step: the warp needs its latent table and the two gradient outputs");
        NIW_REQUIRE(!d->ndc || d->ndc_near > 0.f, "niw_train_step: ndc_near=%g", (double)d->ndc_near);
    } else {
        // the vanilla model (model/nerf.py:251-288): rays of the given cameras, nothing pose-related is trained
        NIW_REQUIRE(d->pose_init, "niw_train_step: without a warp the rays come from the cameras `pose_init` (world->camera, required)");
        NIW_REQUIRE(d->w_align < 0.f && !d->always_register, "niw_train_step: the alignment term / registration need the warp");
        NIW_REQUIRE(!d->ndc || d->ndc_near > 0.f, "niw_train_step: ndc_near=%g", (double)d->ndc_near);
    }
    NIW_REQUIRE(d->density_noise >= 0.f, "niw_train_step: density_noise=%g", (double)d->density_noise);
    NIW_REQUIRE(d->n_views > 0 && d->n_views <= 64 && d->H > 0 && d->W > 0 && d->rays_per_view > 0, "niw_train_step: n_views=%d (1..64) H=%d W=%d rays_per_view=%lld",
                d->n_views, d->H, d->W, (long long)d->rays_per_view);
    NIW_REQUIRE(d->rays_per_view <= (int64_t)d->H * d->W, "niw_train_step: %lld rays per view from a %d x %d image", (long long)d->rays_per_view, d->H, d->W);
    NIW_REQUIRE(d->n_samples > 0 && d->n_fine >= 0, "niw_train_step: n_samples=%d n_fine=%d", d->n_samples, d->n_fine);
    NIW_REQUIRE(0 <= d->view0 && d->view0 < d->view1 && d->view1 <= d->n_views, "niw_train_step: view window [%d, %d) of %d views", d->view0, d->view1, d->n_views);
    NIW_REQUIRE(d->view0 <= d->own0 && d->own0 <= d->own1 && d->own1 <= d->view1, "niw_train_step: owned views [%d, %d) outside the window [%d, %d)", d->own0, d->own1,
                d->view0, d->view1);
    NIW_REQUIRE(d->ray_lo < d->ray_hi && d->ray_lo >= (int64_t)d->view0 * d->rays_per_view && d->ray_hi <= (int64_t)d->view1 * d->rays_per_view,
                "niw_train_step: rays [%lld, %lld) outside the window's rays [%lld, %lld)", (long long)d->ray_lo, (long long)d->ray_hi,
                (long long)d->view0 * d->rays_per_view, (long long)d->view1 * d->rays_per_view);
    NIW_REQUIRE(d->density_activ == NIW_ACT_RELU || d->density_activ == NIW_ACT_SOFTPLUS, "niw_train_step: unknown density activation %d", d->density_activ);
    NIW_REQUIRE(d->precision == NIW_PREC_FP32 || d->precision == NIW_PREC_BF16X3 || d->precision == NIW_PREC_BF16, "niw_train_step: unknown precision %d", d->precision);
    NIW_REQUIRE(d->mse_norm > 0, "niw_train_step: mse_norm must be the element count of the photometric mean");
    if (d->n_fine > 0) {
        NIW_REQUIRE(d->nerf_fine_params && d->d_nerf_fine && d->unif && d->bins, "niw_train_step: the fine pass needs nerf_fine_params, d_nerf_fine and the two inverse-CDF tables");
        NIW_REQUIRE(d->n_samples + d->n_fine <= 1024, "niw_train_step: S+Sf=%d exceeds 1024", d->n_samples + d->n_fine);
    }
    const long long n = d->ray_hi - d->ray_lo, T = d->n_fine > 0 ? d->n_samples + d->n_fine : d->n_samples;
    if (niw_mlp_padded_rows(n, (int)T) * 288ll * 4 >= (1ll << 31)) {
        niw_set_error("niw_train_step: %lld rays x %lld samples exceed one differentiable field launch (1.86 M samples); split the batch over the autograd mirror", n, T);
        return NIW_ERR_UNSUPPORTED;
    }
    return NIW_OK;
}

Layout make_layout(const niw_train_desc* d, float* base) {
    Layout L{};
    L.V = d->view1 - d->view0; L.R = d->rays_per_view; L.n = d->ray_hi - d->ray_lo; L.S = d->n_samples;
    L.T = d->n_fine > 0 ? d->n_samples + d->n_fine : 0;
    L.n_own = d->own1 - d->own0;
    L.mpad_c = niw_mlp_padded_rows(L.n, (int)L.S);
    L.mpad_f = L.T ? niw_mlp_padded_rows(L.n, (int)L.T) : 0;
    const long long P = L.V * 2 * L.R, n = L.n, S = L.S, T = L.T, M = T > S ? T : S;
    Carve ws{base};
    L.ray_idx = ws.take(2 * L.R);                      // int64 [R]
    L.stacked_in = ws.take(3 * P); L.warped = ws.take(3 * P); L.xin = ws.take(9 * P);
    L.w_emb = ws.take(NIW_WARP_WEMB_FLOATS); L.view_b = ws.take(L.V * 3 * 2 * 128); L.w_head = ws.take(NIW_WARP_WHEAD_FLOATS);
    L.prep_ws = ws.take(niw_warp_prep_fwd_workspace_floats((int)L.V));
    L.ray = ws.take(3 * L.V * L.R); L.center = ws.take(3 * L.V * L.R);
    L.z = ws.take(n * S); L.rgb_s = ws.take(3 * n * S); L.sigma_s = ws.take(n * S); L.prob = ws.take(n * S);
    L.rgb = ws.take(3 * n); L.depth = ws.take(n); L.opacity = ws.take(n);
    if (T) {
        L.z_all = ws.take(n * T); L.rgb_f = ws.take(3 * n * T); L.sigma_f = ws.take(n * T);
        L.rgb_fine = ws.take(3 * n); L.depth_fine = ws.take(n); L.opacity_fine = ws.take(n);
    }
    const long long packed = (niw_mlp_packed_bytes(d->precision) + 3) / 4;
    L.packed_c = ws.take(packed);
    if (T) L.packed_f = ws.take(packed);
    L.save_c = ws.take((long long)NIW_SAVE_ROWS * L.mpad_c);
    if (T) L.save_f = ws.take((long long)NIW_SAVE_ROWS * L.mpad_f);
    L.gradws = ws.take((long long)NIW_GRAD_ROWS * (L.mpad_f > L.mpad_c ? L.mpad_f : L.mpad_c));
    L.partial = ws.take(niw_mlp_bwd_workspace_floats(n, (int)M));
    L.d_rgb = ws.take(3 * n);
    if (T) L.d_rgb_f = ws.take(3 * n);
    L.d_rgb_s = ws.take(3 * n * M); L.d_sigma_s = ws.take(n * M);
    L.comp_d_ray_c = ws.take(3 * n); L.mlp_d_c = ws.take(6 * n);
    if (T) { L.comp_d_ray_f = ws.take(3 * n); L.mlp_d_f = ws.take(6 * n); }
    // one-launch form: the coarse pass's backward runs with its forward, long before the fine pass's -- with a fine pass it needs sample
    // gradients of its own (4 MB at cfg2) instead of sharing the fine pass's
    L.resid_c = ws.take(3 * n);
    L.d_rgb_s_c = L.d_rgb_s; L.d_sigma_s_c = L.d_sigma_s;
    if (T) { L.resid_f = ws.take(3 * n); L.d_rgb_s_c = ws.take(3 * n * S); L.d_sigma_s_c = ws.take(n * S); }
    if (d->density_noise > 0.f) { L.noise_c = ws.take(n * S); if (T) L.noise_f = ws.take(n * T); }
    if (d->warp_params && d->ndc) { L.ray_cam = ws.take(3 * L.V * L.R); L.center_cam = ws.take(3 * L.V * L.R); L.ndc_g = ws.take(4 * 3 * L.V * L.R); }
    L.mom = ws.take(2 * 16 * L.V);                     // double [V,16]
    L.poses = ws.take(12 * L.V);
    L.d_target = ws.take(3 * P); L.d_warped = ws.take(3 * P);
    L.warp_ws = ws.take(niw_warp_bwd_workspace_floats((int)L.V, 2 * L.R));
    L.d_w_emb = ws.take(NIW_WARP_WEMB_FLOATS); L.d_view_b = ws.take(L.V * 3 * 2 * 128); L.d_w_head = ws.take(NIW_WARP_WHEAD_FLOATS);
    L.prep_bwd_ws = ws.take(niw_warp_prep_bwd_workspace_floats((int)L.V));
    L.total = ws.used;
    return L;
}

int pack(const niw_train_desc* d, const float* params, float* image, niw_stream_t stream) {
    if (d->precision != NIW_PREC_FP32) return niw_mlp_pack_weights_prec(params, d->precision, image, stream);
    return d->pack_index ? niw_mlp_pack_weights_indexed(params, d->pack_index, image, stream) : niw_mlp_pack_weights(params, image, stream);
}

}  // namespace

int niw_dw_heads_prepare();   // niw_dw_gemm.hip: the stream of the vector-ALU head pieces of the weight gradient

extern "C" int niw_train_step_prepare(void) {
    if (!side_lane() || niw_dw_heads_prepare() != NIW_OK) {
        niw_set_error("niw_train_step_prepare: cannot create the second stream");
        return NIW_ERR_LAUNCH;
    }
    return NIW_OK;
}

extern "C" int64_t niw_train_step_workspace_floats(const niw_train_desc* d) {
    if (check_desc(d) != NIW_OK) return 0;
    return make_layout(d, nullptr).total;
}

#define NIW_RUN(call)                    \
    do {                                 \
        const int rc_ = (call);          \
        if (rc_ != NIW_OK) return rc_;   \
    } while (0)

extern "C" int niw_train_step(const niw_train_desc* d, float* workspace, int stage_begin, int stage_end, niw_stream_t stream) {
    NIW_RUN(check_desc(d));
    NIW_REQUIRE(workspace, "niw_train_step: null workspace");
    NIW_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "niw_train_step: the workspace must start on a 256-byte boundary");
    NIW_REQUIRE(0 <= stage_begin && stage_begin < stage_end && stage_end <= NIW_STAGE_END, "niw_train_step: stages [%d, %d) of %d", stage_begin, stage_end, (int)NIW_STAGE_END);
    const Layout L = make_layout(d, workspace);
    hipStream_t st = (hipStream_t)stream;
    const long long n = L.n, R = L.R, V = L.V;
    const int S = (int)L.S, T = (int)L.T;
    const long long a = d->ray_lo - (long long)d->view0 * R;                 // the share inside the window's flattened rays
    const float* ray = L.ray + 3 * a;
    const float* center = L.center + 3 * a;
    const int64_t* ray_idx = reinterpret_cast<const int64_t*>(L.ray_idx);
    const bool fine = T > 0;
    const bool warp = d->warp_params != nullptr;     // false: the vanilla model on the given cameras
    const bool loss_c = d->w_render >= 0.f, loss_f = fine && d->w_render_fine >= 0.f, align = d->w_align >= 0.f;
    const bool registration = align || d->always_register;
    float* poses = d->poses ? d->poses + 12ll * d->view0 : L.poses;
    float* rgb = d->rgb ? d->rgb : L.rgb;
    float* rgb_fine = d->rgb_fine ? d->rgb_fine : L.rgb_fine;
    const long long own_off = d->own0 - d->view0;
    auto in = [&](int s) { return stage_begin <= s && s < stage_end; };

    // second stream (niw.h: overlap): X carries the small independent stages, `stream` the field-MLP chain
    SideLane* lane = (warp && d->overlap && stage_begin == 0 && stage_end == NIW_STAGE_END) ? side_lane() : nullptr;   // (the vanilla chain has nothing to run beside it)
    SideLaneHold hold(lane, st);
    hipStream_t X = lane ? lane->s : st;
    niw_stream_t sx = (niw_stream_t)X;
    // ---- the front of the iteration, one launch: pixel draw + un-warped points, stratified depths, fp32 weight images, pad columns
    if (in(NIW_STAGE_FRONT)) {
        const bool gather = d->precision == NIW_PREC_FP32 && d->pack_index;
        long long pad_rows = 0, ppad = 0, n_cols = 0;
        if (warp) niw_warp_bwd_pad_geometry((int)V, 2 * R, &pad_rows, &ppad, &n_cols);
        NIW_RUN(niw_launch_step_front((int64_t)d->H * d->W, d->pixel_seed, d->draw, d->draw_dev, d->intr + 9ll * d->view0,
                                      d->pose_init ? d->pose_init + 12ll * d->view0 : nullptr, (int)V, R, d->H, d->W, reinterpret_cast<int64_t*>(L.ray_idx),
                                      L.stacked_in, d->depth_seed, d->stratified, n, S, d->depth_min, d->depth_max, d->inverse_depth, L.z, d->nerf_params,
                                      fine ? d->nerf_fine_params : nullptr, d->pack_index, gather ? L.packed_c : nullptr, gather && fine ? L.packed_f : nullptr,
                                      warp ? L.warp_ws : nullptr, pad_rows, ppad, n_cols, d->warp_params, warp ? d->latent + 128ll * d->view0 : nullptr, (int)V,
                                      warp ? L.prep_ws : nullptr, st));
        // density noise of both passes (model/nerf.py:428-429): streams keyed like the depth draw, the fine pass one key further
        if (d->density_noise > 0.f) {
            NIW_RUN(niw_normal_rng(d->noise_seed, d->draw, d->draw_dev, n * S, d->density_noise, L.noise_c, stream));
            if (fine) NIW_RUN(niw_normal_rng(d->noise_seed + 1, d->draw, d->draw_dev, n * T, d->density_noise, L.noise_f, stream));
        }
        if (!gather) {                      // split-bf16 images, or no gather table: the packing kernels of the entry points
            NIW_RUN(pack(d, d->nerf_params, L.packed_c, stream));
            if (fine) NIW_RUN(pack(d, d->nerf_fine_params, L.packed_f, stream));
        }
    }
    if (in(NIW_STAGE_WARP_FWD) && !warp) {
        // the vanilla model: camera centres and rays of the drawn pixels (camera.get_center_and_ray, camera.py:419-443), re-parametrised
        // in NDC when asked (camera.py:523-540) -- the entry points the mirror calls, nothing to differentiate
        float* c0 = d->ndc ? L.warped : L.center;            // (NDC: camera-frame rays parked where a warp would put its points)
        float* r0 = d->ndc ? L.warped + 3 * V * R : L.ray;
        NIW_RUN(niw_raygen(d->intr + 9ll * d->view0, d->pose_init + 12ll * d->view0, ray_idx, 0, (int)V, R, d->H, d->W, 1, c0, r0, stream));
        if (d->ndc) NIW_RUN(niw_convert_ndc(c0, r0, d->intr + 9ll * d->view0, (int)V, R, d->ndc_near, L.center, L.ray, stream));
    }
    if (in(NIW_STAGE_WARP_FWD) && warp) {
        // (the code projection at the head of prep_ws was made by the front kernel)
        NIW_RUN(niw_launch_warp_prep_fwd_main(d->warp_params, (int)V, L.prep_ws, L.w_emb, L.view_b, L.w_head, st));
        NIW_RUN(niw_warp_fwd(L.w_emb, L.view_b, L.w_head, L.stacked_in, (int)V, 2 * R, d->chan_w, d->index_window, d->window_dev, d->use_index_window, nullptr,
                             nullptr, 0, L.warped, L.xin, stream));
        split_rays_kernel<<<(unsigned)((V * R * 3 + 255) / 256), 256, 0, st>>>(L.warped, V, R, d->ndc ? L.ray_cam : L.ray, d->ndc ? L.center_cam : L.center);
        NIW_LAUNCH_CHECK("niw_train_step (split rays)");
        // camera.ndc: every stage below sees the NDC form of the warped rays (model/nerf_inn_llff.py:627-641 -> camera.py:523-540)
        if (d->ndc) NIW_RUN(niw_convert_ndc(L.center_cam, L.ray_cam, d->intr + 9ll * d->view0, (int)V, R, d->ndc_near, L.center, L.ray, stream));
    }
    if (lane) {
        NIW_HIP(hipEventRecord(lane->warped, st), "warped");
        NIW_HIP(hipStreamWaitEvent(X, lane->warped, 0), "warped");
        hold.forked = true;
    }
    // ---- X: rigid registration of the warped onto the un-warped points (whole views: no collective under sharding either) and the
    // alignment loss -- the part of the LOSS stage that needs only the warp
    if (in(NIW_STAGE_LOSS)) {
        if (registration) NIW_RUN(niw_launch_align_register(L.warped, L.stacked_in, (int)V, 2 * R, reinterpret_cast<double*>(L.mom), poses, X));
        if (align && L.n_own > 0)
            NIW_RUN(niw_align_loss(L.warped + own_off * 6 * R, L.stacked_in + own_off * 6 * R, poses + own_off * 12, L.n_own, 2 * R, 3.0 * d->n_views * 2.0 * (double)R,
                                   d->loss + 2, L.d_target + own_off * 6 * R, sx));
    }
    // ---- main: field forward(s), compositing, photometric loss(es), their backward
    if (in(NIW_STAGE_MLP_FWD))
        NIW_RUN(niw_mlp_fwd(L.packed_c, center, ray, L.z, L.noise_c, n, S, d->band_w3d, d->band_wview, d->band_dev, d->density_activ, d->precision, L.rgb_s, L.sigma_s,
                            (loss_c ? L.save_c : nullptr), stream));
    // compositing + photometric residual + their backward as ONE launch per pass wherever the span kernels cover the sample count
    // (niw_composite_mse_train; NIW_TRAIN_ONE_LAUNCH_LOSS=0: the three launches, diagnostic); the loss values are then formed by the
    // closing kernel of the iteration (combine_kernel) from the residuals
    static const bool one_launch_env = [] { const char* e = getenv("NIW_TRAIN_ONE_LAUNCH_LOSS"); return !e || atoi(e) != 0; }();
    const int64_t hw = (int64_t)d->H * d->W;
    const int has_bg = d->has_bg ? 1 : 0;            // opaque background (model/nerf.py:470-472): the three-launch form carries it
    const bool one_c = one_launch_env && !has_bg && loss_c && S % 4 == 0 && S <= 256;
    const bool one_f = one_launch_env && !has_bg && loss_f && T % 4 == 0 && T <= 256;
    if (in(NIW_STAGE_COMPOSITE_FWD)) {
        if (one_c)
            NIW_RUN(niw_composite_mse_train(ray, L.rgb_s, L.sigma_s, L.z, n, S, d->image, ray_idx, d->n_views, R, hw, d->ray_lo, d->mse_norm, d->w_render, rgb, L.depth,
                                            L.opacity, L.prob, L.resid_c, nullptr, L.d_rgb_s_c, L.d_sigma_s_c, L.comp_d_ray_c, stream));
        else
            NIW_RUN(niw_composite_fwd(ray, L.rgb_s, L.sigma_s, L.z, n, S, has_bg, d->bg, rgb, L.depth, L.opacity, L.prob, stream));
    }
    if (fine) {
        if (in(NIW_STAGE_RESAMPLE)) NIW_RUN(niw_sample_pdf_merge(L.prob, L.z, d->unif, d->bins, n, S, d->n_fine, nullptr, L.z_all, stream));
        if (in(NIW_STAGE_MLP_FWD_FINE))
            NIW_RUN(niw_mlp_fwd(L.packed_f, center, ray, L.z_all, L.noise_f, n, T, d->band_w3d, d->band_wview, d->band_dev, d->density_activ, d->precision, L.rgb_f,
                                L.sigma_f, (loss_f ? L.save_f : nullptr), stream));
        if (in(NIW_STAGE_COMPOSITE_FWD_FINE)) {
            if (one_f)
                NIW_RUN(niw_composite_mse_train(ray, L.rgb_f, L.sigma_f, L.z_all, n, T, d->image, ray_idx, d->n_views, R, hw, d->ray_lo, d->mse_norm, d->w_render_fine,
                                                rgb_fine, L.depth_fine, L.opacity_fine, nullptr, L.resid_f, nullptr, L.d_rgb_s, L.d_sigma_s, L.comp_d_ray_f, stream));
            else
                NIW_RUN(niw_composite_fwd(ray, L.rgb_f, L.sigma_f, L.z_all, n, T, has_bg, d->bg, rgb_fine, L.depth_fine, L.opacity_fine, nullptr, stream));
        }
    }
    if (in(NIW_STAGE_LOSS)) {
        if (loss_c && !one_c) NIW_RUN(niw_mse_fwd_bwd(rgb, d->image, ray_idx, d->n_views, R, hw, d->ray_lo, n, d->mse_norm, d->w_render, d->loss + 0, L.d_rgb, stream));
        if (loss_f && !one_f) NIW_RUN(niw_mse_fwd_bwd(rgb_fine, d->image, ray_idx, d->n_views, R, hw, d->ray_lo, n, d->mse_norm, d->w_render_fine, d->loss + 1, L.d_rgb_f, stream));
    }
    // (the coarse pass's sample gradients: its own buffers when its backward ran with its forward, else the shared ones)
    float* const d_rgb_s_c = one_c ? L.d_rgb_s_c : L.d_rgb_s;
    float* const d_sigma_s_c = one_c ? L.d_sigma_s_c : L.d_sigma_s;
    if (fine) {
        if (loss_f) {
            if (in(NIW_STAGE_COMPOSITE_BWD_FINE) && !one_f)
                NIW_RUN(niw_composite_bwd(ray, L.rgb_f, L.sigma_f, L.z_all, n, T, has_bg, d->bg, L.d_rgb_f, nullptr, nullptr, nullptr, L.d_rgb_s, L.d_sigma_s, L.comp_d_ray_f, stream));
            if (in(NIW_STAGE_MLP_BWD_DX_FINE))
                NIW_RUN(niw_mlp_bwd_dx(L.packed_f, center, ray, L.z_all, n, T, d->density_activ, d->precision, L.rgb_f, L.d_rgb_s, L.d_sigma_s, L.save_f, L.gradws,
                                       warp ? L.mlp_d_f : nullptr, warp ? L.mlp_d_f + 3 * n : nullptr, stream));
            if (in(NIW_STAGE_MLP_BWD_DW_FINE)) NIW_RUN(niw_mlp_bwd_dw(L.save_f, L.gradws, n, T, d->precision, L.partial, d->d_nerf_fine, stream));
        } else if (in(NIW_STAGE_MLP_BWD_DW_FINE)) {
            NIW_RUN(fill(d->d_nerf_fine, NIW_NERF_PARAM_FLOATS, 0.f, st));
        }
        if (d->fine_grads_ready && in(NIW_STAGE_MLP_BWD_DW_FINE)) NIW_HIP(hipEventRecord((hipEvent_t)d->fine_grads_ready, st), "fine_grads_ready");
    }
    if (loss_c) {
        if (in(NIW_STAGE_COMPOSITE_BWD) && !one_c)
            NIW_RUN(niw_composite_bwd(ray, L.rgb_s, L.sigma_s, L.z, n, S, has_bg, d->bg, L.d_rgb, nullptr, nullptr, nullptr, L.d_rgb_s, L.d_sigma_s, L.comp_d_ray_c, stream));
        if (in(NIW_STAGE_MLP_BWD_DX))
            NIW_RUN(niw_mlp_bwd_dx(L.packed_c, center, ray, L.z, n, S, d->density_activ, d->precision, L.rgb_s, d_rgb_s_c, d_sigma_s_c, L.save_c, L.gradws,
                                   warp ? L.mlp_d_c : nullptr, warp ? L.mlp_d_c + 3 * n : nullptr, stream));
    }
    if (lane) {
        NIW_HIP(hipEventRecord(lane->dx, st), "dx");
        NIW_HIP(hipStreamWaitEvent(X, lane->dx, 0), "dx");
    }
    // ---- main: the coarse network's dW GEMMs;  X beside them: everything behind the ray gradients
    if (loss_c) {
        if (in(NIW_STAGE_MLP_BWD_DW)) NIW_RUN(niw_mlp_bwd_dw(L.save_c, L.gradws, n, S, d->precision, L.partial, d->d_nerf, stream));
    } else if (in(NIW_STAGE_MLP_BWD_DW)) {
        NIW_RUN(fill(d->d_nerf, NIW_NERF_PARAM_FLOATS, 0.f, st));
    }
    if (in(NIW_STAGE_WARP_BWD)) {
        CombineArgs c{};
        // accumulation order of autograd on the shared `ray` / `center` tensors: fine compositing, fine field, coarse compositing, coarse field
        c.d_ray[0] = loss_f ? L.comp_d_ray_f : nullptr; c.d_ray[1] = loss_f ? L.mlp_d_f + 3 * n : nullptr;
        c.d_ray[2] = loss_c ? L.comp_d_ray_c : nullptr; c.d_ray[3] = loss_c ? L.mlp_d_c + 3 * n : nullptr;
        c.d_center[0] = loss_f ? L.mlp_d_f : nullptr; c.d_center[1] = loss_c ? L.mlp_d_c : nullptr;
        c.d_target = (align && L.n_own > 0) ? L.d_target : nullptr;
        c.d_warped = L.d_warped;
        c.V = warp ? V : 0; c.R = R; c.a = a; c.b = a + n;      // (vanilla: no gradient route to sum, the kernel only closes the losses)
        c.own_lo = (int)own_off; c.own_hi = (int)own_off + L.n_own;
        c.w_align = d->w_align;
        c.d_latent = warp ? d->d_latent : nullptr; c.lat_lo = 128ll * d->view0; c.lat_hi = 128ll * d->view1; c.lat_n = warp ? 128ll * d->n_views : 0;
        c.loss = d->loss;
        c.w[0] = d->w_render; c.w[1] = d->w_render_fine; c.w[2] = d->w_align;
        c.present[0] = loss_c; c.present[1] = loss_f; c.present[2] = align && L.n_own > 0;
        c.resid[0] = one_c ? L.resid_c : nullptr; c.resid[1] = one_f ? L.resid_f : nullptr;
        c.resid_n = 3 * n; c.n_norm = d->mse_norm;
        if (warp && d->ndc) {
            // the routes meet at the NDC rays: their ordered sums go back through the re-parametrisation first (the launch the mirror's
            // autograd makes, ops._ConvertNDC), and reach the combine kernel as ONE ray route and ONE centre route over the whole window
            NdcRoutesArgs r{};
            for (int k = 0; k < 4; ++k) r.d_ray[k] = c.d_ray[k];
            for (int k = 0; k < 2; ++k) r.d_center[k] = c.d_center[k];
            const long long VR3 = V * R * 3;
            r.g_ray = L.ndc_g; r.g_center = L.ndc_g + VR3; r.total = VR3; r.a3 = 3 * a; r.b3 = 3 * (a + n);
            ndc_routes_kernel<<<(unsigned)((VR3 + 255) / 256), 256, 0, X>>>(r);
            NIW_LAUNCH_CHECK("niw_train_step (NDC routes)");
            NIW_RUN(niw_convert_ndc_bwd(L.center_cam, L.ray_cam, d->intr + 9ll * d->view0, (int)V, R, d->ndc_near, r.g_center, r.g_ray, L.ndc_g + 3 * VR3,
                                        L.ndc_g + 2 * VR3, sx));
            for (int k = 0; k < 4; ++k) c.d_ray[k] = nullptr;
            for (int k = 0; k < 2; ++k) c.d_center[k] = nullptr;
            c.d_ray[0] = L.ndc_g + 2 * VR3; c.d_center[0] = L.ndc_g + 3 * VR3;
            c.a = 0; c.b = V * R;                                  // (the whole window: zero outside the share already)
        }
        const long long work = warp ? (V * R * 3 > c.lat_n ? V * R * 3 : c.lat_n) : 1;
        combine_kernel<<<(unsigned)((work + 255) / 256), 256, 0, X>>>(c);
        NIW_LAUNCH_CHECK("niw_train_step (combine)");
        if (warp) {
        NIW_RUN(niw_launch_warp_bwd_main(L.w_emb, L.view_b, L.w_head, L.stacked_in, (int)V, 2 * R, d->chan_w, d->index_window, d->window_dev, d->use_index_window,
                                         nullptr, nullptr, L.xin, L.d_warped, L.warp_ws, L.d_w_emb, L.d_view_b, L.d_w_head, nullptr, X));
        // (the code projection of the forward's operand preparation is still at the head of its workspace)
        NIW_RUN(niw_launch_warp_prep_bwd(d->warp_params, d->latent + 128ll * d->view0, (int)V, L.d_w_emb, L.d_view_b, L.d_w_head, L.prep_bwd_ws, L.prep_ws,
                                         d->d_warp, d->d_latent + 128ll * d->view0, X));
        }
    }
    if (lane) {
        hold.forked = false;
        NIW_HIP(hipEventRecord(lane->join, X), "join");
        NIW_HIP(hipStreamWaitEvent(st, lane->join, 0), "join");
    }
    return NIW_OK;
}

extern "C" int niw_adam_step_multi(const niw_adam_group* groups, int n_groups, double beta1, double beta2, double eps, const float* hyper_dev,
                                   niw_stream_t stream) {
    NIW_REQUIRE(groups && n_groups > 0 && n_groups <= kMaxAdamGroups, "niw_adam_step_multi: 1..%d groups (got %d)", kMaxAdamGroups, n_groups);
    AdamBatch b{};
    b.n_groups = n_groups;
    b.w1 = (float)(1.0 - beta1); b.b2 = (float)beta2; b.w2 = (float)(1.0 - beta2); b.eps = (float)eps;
    int blocks = 0;
    for (int k = 0; k < n_groups; ++k) {
        const niw_adam_group& g = groups[k];
        NIW_REQUIRE(g.n >= 0, "niw_adam_step_multi: group %d has n=%lld", k, (long long)g.n);
        NIW_REQUIRE(g.n == 0 || (g.param && g.grad && g.exp_avg && g.exp_avg_sq), "niw_adam_step_multi: null pointer in group %d", k);
        NIW_REQUIRE(g.n == 0 || g.step >= 1 || hyper_dev, "niw_adam_step_multi: group %d step=%d", k, g.step);
        const int step = g.step < 1 ? 1 : g.step;
        // the scalars are formed in double like torch.optim.Adam forms them in Python floats, and rounded to fp32 once (niw_adam_step)
        const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
        b.g[k] = AdamGroupDev{g.param, g.grad, g.exp_avg, g.exp_avg_sq, g.n, (float)(g.lr / bc1), (float)sqrt(bc2), blocks};
        blocks += (int)((g.n + 255) / 256);
    }
    if (blocks == 0) return NIW_OK;
    adam_multi_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(b, hyper_dev);
    NIW_LAUNCH_CHECK("niw_adam_step_multi");
    return NIW_OK;
}
