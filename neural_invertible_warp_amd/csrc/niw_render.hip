// Gradient-free render of a pixel range as ONE call: rays -> (NDC) -> stratified depths -> field MLP -> compositing
// [-> inverse-CDF depths merged with the coarse ones -> fine field MLP -> compositing].  Replaces the body of
// Graph.render (reference model/nerf.py:293-319) under torch.no_grad() and, with first_pixel / n_pixels spanning the image,
// the whole slice loop of Graph.render_by_slices (model/nerf.py:321-332).
//
// The entry point only SEQUENCES the library's own stages on the caller's stream, into a caller-provided workspace: per
// (ray, sample) the chain moves 40 bytes through HBM (depth, rgb, sigma out of the MLP; the same back into the scan) against
// 1.06 MFLOP of MLP arithmetic -- 0.008 ns of HBM time next to 6.7 ns of MFMA time -- so folding the scan into the MLP's
// last layer would save nothing measurable, while keeping the stages separate keeps every number bit-identical to the
// per-stage entry points the training path uses.
#include "niw_common.h"

namespace {

struct Carve {             // hands out consecutive pieces of the workspace
    float* p;
    float* take(long long n) {
        float* r = p;
        p += (n + 3) / 4 * 4;     // keep every piece 16-byte aligned (vector loads of the scan)
        return r;
    }
};

long long pad4(long long n) { return (n + 3) / 4 * 4; }

// One field evaluation over all rays, split so that a launch stays below the 2^24 padded samples niw_mlp_fwd takes.
int field(const niw_render_desc* d, bool fine, const float* packed, const float* center, const float* ray, const float* depth, long long n_rays,
          int S, float* rgb_s, float* sigma_s, niw_stream_t stream) {
    const long long max_rays = ((1ll << 24) - 256) / S;
    for (long long a = 0; a < n_rays; a += max_rays) {
        const long long n = n_rays - a < max_rays ? n_rays - a : max_rays;
        const int rc = niw_mlp_fwd(packed, center + 3 * a, ray + 3 * a, depth + a * S, nullptr, n, S, fine ? d->band_w3d_fine : d->band_w3d,
                                   fine ? d->band_wview_fine : d->band_wview, fine ? d->band_dev_fine : d->band_dev, d->density_activ, d->precision, rgb_s + 3 * a * S, sigma_s + a * S, nullptr, stream);
        if (rc != NIW_OK) return rc;
    }
    return NIW_OK;
}

}  // namespace

extern "C" int64_t niw_render_fwd_workspace_floats(int n_views, int64_t n_pixels, int n_samples, int n_fine) {
    const long long n = (long long)n_views * n_pixels, S = n_samples, T = n_fine > 0 ? S + n_fine : 0;
    long long f = 4 * pad4(3 * n);                       // centre, ray and their NDC images
    f += pad4(n * S) * 3 + pad4(3 * n * S);              // depths, sigma, prob, rgb of the coarse pass
    if (T) f += pad4(n * T) * 2 + pad4(3 * n * T);       // merged depths, sigma, rgb of the fine pass
    return f;
}

extern "C" int niw_render_fwd(const niw_render_desc* d, float* workspace, float* rgb, float* depth, float* opacity, float* rgb_fine,
                              float* depth_fine, float* opacity_fine, niw_stream_t stream) {
    NIW_REQUIRE(d && workspace && rgb && depth && opacity, "niw_render_fwd: null pointer");
    NIW_REQUIRE(d->intr && d->pose && d->packed, "niw_render_fwd: cameras and packed weights are required");
    NIW_REQUIRE(d->n_views > 0 && d->n_pixels > 0 && d->n_samples > 0 && d->n_fine >= 0, "niw_render_fwd: n_views=%d n_pixels=%lld n_samples=%d n_fine=%d",
                d->n_views, (long long)d->n_pixels, d->n_samples, d->n_fine);
    NIW_REQUIRE(d->first_pixel >= 0 && d->first_pixel + d->n_pixels <= (int64_t)d->H * d->W, "niw_render_fwd: pixels %lld..%lld outside the %d x %d image",
                (long long)d->first_pixel, (long long)(d->first_pixel + d->n_pixels), d->H, d->W);
    if (d->n_fine > 0) {
        NIW_REQUIRE(d->packed_fine && d->unif && d->bins && rgb_fine && depth_fine && opacity_fine,
                    "niw_render_fwd: the fine pass needs packed_fine, the two inverse-CDF tables and its three outputs");
    }
    const long long n = (long long)d->n_views * d->n_pixels;
    const int S = d->n_samples, T = d->n_fine > 0 ? S + d->n_fine : 0;
    Carve ws{workspace};
    float* center = ws.take(3 * n);
    float* ray = ws.take(3 * n);
    float* center_ndc = ws.take(3 * n);
    float* ray_ndc = ws.take(3 * n);
    float* z = ws.take(n * S);
    float* sigma_s = ws.take(n * S);
    float* prob = ws.take(n * S);
    float* rgb_s = ws.take(3 * n * S);

    int rc = niw_raygen(d->intr, d->pose, nullptr, d->first_pixel, d->n_views, d->n_pixels, d->H, d->W, 1, center, ray, stream);
    if (rc != NIW_OK) return rc;
    if (d->ndc) {
        rc = niw_convert_ndc(center, ray, d->intr, d->n_views, d->n_pixels, d->ndc_near, center_ndc, ray_ndc, stream);
        if (rc != NIW_OK) return rc;
        center = center_ndc;
        ray = ray_ndc;
    }
    rc = niw_sample_stratified(d->u, n, S, d->depth_min, d->depth_max, d->inverse_depth, z, stream);
    if (rc != NIW_OK) return rc;
    rc = field(d, false, d->packed, center, ray, z, n, S, rgb_s, sigma_s, stream);
    if (rc != NIW_OK) return rc;
    rc = niw_composite_fwd(ray, rgb_s, sigma_s, z, n, S, d->has_bg, d->bg, rgb, depth, opacity, T ? prob : nullptr, stream);
    if (rc != NIW_OK || !T) return rc;

    float* z_all = ws.take(n * T);
    float* sigma_f = ws.take(n * T);
    float* rgb_f = ws.take(3 * n * T);
    rc = niw_sample_pdf_merge(prob, z, d->unif, d->bins, n, S, d->n_fine, nullptr, z_all, stream);
    if (rc != NIW_OK) return rc;
    rc = field(d, true, d->packed_fine, center, ray, z_all, n, T, rgb_f, sigma_f, stream);
    if (rc != NIW_OK) return rc;
    return niw_composite_fwd(ray, rgb_f, sigma_f, z_all, n, T, d->has_bg, d->bg, rgb_fine, depth_fine, opacity_fine, nullptr, stream);
}
