// Alpha compositing along rays: NeRF.composite (reference model/nerf.py:458-474), forward and
// closed-form backward.  One 64-lane wave per ray; the transmittance
// T_i = exp(-sum_{j<i} sigma_j delta_j) is a wave-level exclusive scan (shuffle ladder) with a
// running carry across 64-sample chunks, the backward a suffix scan.  HBM-bound: 20 B/sample in.
#include "niw_common.h"

namespace {

__device__ __forceinline__ float wave_incl_scan(float v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        float t = __shfl_up(v, o);
        if (lane >= o) v += t;
    }
    return v;
}
__device__ __forceinline__ float wave_suffix_incl_scan(float v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        float t = __shfl_down(v, o);
        if (lane + o < 64) v += t;
    }
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// sigma*delta of sample s (reference: dist = intv * ray_length; sigma_delta = density * dist)
__device__ __forceinline__ float interval(const float* __restrict__ d, int s, int S) {
    return s == S - 1 ? 1e10f : d[s + 1] - d[s];
}

__global__ __launch_bounds__(256) void composite_fwd_kernel(const float* __restrict__ ray, const float* __restrict__ rgb_s,
                                                            const float* __restrict__ sigma_s, const float* __restrict__ depth_s,
                                                            long long n_rays, int S, int has_bg, float bg,
                                                            float* __restrict__ rgb, float* __restrict__ depth,
                                                            float* __restrict__ opacity, float* __restrict__ prob) {
    const int lane = threadIdx.x & 63;
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_rays) return;
    const float rx = ray[r * 3], ry = ray[r * 3 + 1], rz = ray[r * 3 + 2];
    const float len = sqrtf(rx * rx + ry * ry + rz * rz);
    const float* d = depth_s + r * S;
    const float* sg = sigma_s + r * S;
    const float* c = rgb_s + r * S * 3;
    float carry = 0.f, a0 = 0.f, a1 = 0.f, a2 = 0.f, ad = 0.f, ao = 0.f;
    for (int base = 0; base < S; base += 64) {
        const int s = base + lane;
        const bool v = s < S;
        float dv = 0.f, sd = 0.f;
        if (v) {
            dv = d[s];
            sd = sg[s] * (interval(d, s, S) * len);
        }
        // exclusive scan: shift by one lane first (an inclusive-minus-self form would cancel
        // catastrophically against the 1e10 closing interval)
        float sh = __shfl_up(sd, 1);
        if (lane == 0) sh = 0.f;
        const float excl = carry + wave_incl_scan(sh, lane);
        const float T = expf(-excl);
        const float w = v ? T * (1.f - expf(-sd)) : 0.f;
        if (v) {
            if (prob) prob[r * S + s] = w;
            a0 += w * c[s * 3];
            a1 += w * c[s * 3 + 1];
            a2 += w * c[s * 3 + 2];
            ad += w * dv;
            ao += w;
        }
        carry = __shfl(excl + sd, 63);   // lane 63 is valid whenever another chunk follows
    }
    a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2); ad = wave_sum(ad); ao = wave_sum(ao);
    if (lane == 0) {
        if (has_bg) { const float t = bg * (1.f - ao); a0 += t; a1 += t; a2 += t; }
        rgb[r * 3] = a0; rgb[r * 3 + 1] = a1; rgb[r * 3 + 2] = a2;
        depth[r] = ad;
        opacity[r] = ao;
    }
}

__global__ __launch_bounds__(256) void composite_bwd_kernel(const float* __restrict__ ray, const float* __restrict__ rgb_s,
                                                            const float* __restrict__ sigma_s, const float* __restrict__ depth_s,
                                                            long long n_rays, int S, int has_bg, float bg,
                                                            const float* __restrict__ g_rgb, const float* __restrict__ g_depth,
                                                            const float* __restrict__ g_opacity, const float* __restrict__ g_prob,
                                                            float* __restrict__ d_rgb_s, float* __restrict__ d_sigma_s,
                                                            float* __restrict__ d_ray) {
    __shared__ float chunk_prefix[4][17];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long r = (long long)blockIdx.x * 4 + wv;
    if (r >= n_rays) return;
    const float rx = ray[r * 3], ry = ray[r * 3 + 1], rz = ray[r * 3 + 2];
    const float len = sqrtf(rx * rx + ry * ry + rz * rz);
    const float* d = depth_s + r * S;
    const float* sg = sigma_s + r * S;
    const float* c = rgb_s + r * S * 3;
    const float gr0 = g_rgb ? g_rgb[r * 3] : 0.f, gr1 = g_rgb ? g_rgb[r * 3 + 1] : 0.f, gr2 = g_rgb ? g_rgb[r * 3 + 2] : 0.f;
    const float gd = g_depth ? g_depth[r] : 0.f;
    float go = g_opacity ? g_opacity[r] : 0.f;
    if (has_bg) go -= bg * (gr0 + gr1 + gr2);
    // pass 1: prefix of sigma*delta at the start of every chunk
    const int n_chunks = (S + 63) / 64;
    float carry = 0.f;
    for (int k = 0; k < n_chunks; ++k) {
        const int s = k * 64 + lane;
        float sd = s < S ? sg[s] * (interval(d, s, S) * len) : 0.f;
        if (lane == 0) chunk_prefix[wv][k] = carry;
        carry += wave_sum(sd);
    }
    __builtin_amdgcn_wave_barrier();
    // pass 2: chunks in reverse with a suffix carry of g_k w_k
    float suffix = 0.f, dlen = 0.f;
    for (int k = n_chunks - 1; k >= 0; --k) {
        const int s = k * 64 + lane;
        const bool v = s < S;
        float sd = 0.f, sig = 0.f, itv = 0.f, g = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f;
        if (v) {
            sig = sg[s];
            itv = interval(d, s, S);
            sd = sig * (itv * len);
            c0 = c[s * 3]; c1 = c[s * 3 + 1]; c2 = c[s * 3 + 2];
            g = gr0 * c0 + gr1 * c1 + gr2 * c2 + gd * d[s] + go + (g_prob ? g_prob[r * S + s] : 0.f);
        }
        float sh = __shfl_up(sd, 1);
        if (lane == 0) sh = 0.f;
        const float excl = chunk_prefix[wv][k] + wave_incl_scan(sh, lane);
        const float T = expf(-excl), e = expf(-sd);
        const float w = v ? T * (1.f - e) : 0.f;
        const float gw = g * w;
        // sum_{k>i} g_k w_k = (inclusive suffix within chunk) - own + carry from later chunks
        float shd = __shfl_down(gw, 1);
        if (lane == 63) shd = 0.f;
        const float after = suffix + wave_suffix_incl_scan(shd, lane);
        const float ds = g * (T * e) - after;           // dL/d(sigma*delta)
        if (v) {
            d_sigma_s[r * S + s] = ds * (itv * len);
            d_rgb_s[(r * S + s) * 3] = w * gr0;
            d_rgb_s[(r * S + s) * 3 + 1] = w * gr1;
            d_rgb_s[(r * S + s) * 3 + 2] = w * gr2;
            dlen += ds * (sig * itv);
        }
        suffix = __shfl(after + gw, 0);
    }
    dlen = wave_sum(dlen);
    if (lane == 0) {
        const float inv = len > 0.f ? dlen / len : 0.f;
        d_ray[r * 3] = inv * rx; d_ray[r * 3 + 1] = inv * ry; d_ray[r * 3 + 2] = inv * rz;
    }
}

}  // namespace

extern "C" int niw_composite_fwd(const float* ray, const float* rgb_s, const float* sigma_s, const float* depth_s,
                                 int64_t n_rays, int n_samples, int has_bg, float bg,
                                 float* rgb, float* depth, float* opacity, float* prob, niw_stream_t stream) {
    NIW_REQUIRE(ray && rgb_s && sigma_s && depth_s && rgb && depth && opacity, "niw_composite_fwd: null pointer");
    NIW_REQUIRE(n_rays > 0 && n_samples > 0, "niw_composite_fwd: empty input (n_rays=%lld, S=%d)", (long long)n_rays, n_samples);
    // reference nerf.py:461-462 builds the closing 1e10 interval with empty_like(intervals[..., :1]): with a single sample
    // that slice is empty, the sample gets NO interval and every output is zero with an empty prob -- not reproduced
    NIW_REQUIRE(n_samples >= 2, "niw_composite_fwd: needs at least 2 samples per ray (the reference degenerates to all-zero outputs at S=1)");
    const int blocks = (int)((n_rays + 3) / 4);
    composite_fwd_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(ray, rgb_s, sigma_s, depth_s, n_rays, n_samples, has_bg, bg,
                                                                 rgb, depth, opacity, prob);
    NIW_LAUNCH_CHECK("niw_composite_fwd");
    return NIW_OK;
}

extern "C" int niw_composite_bwd(const float* ray, const float* rgb_s, const float* sigma_s, const float* depth_s,
                                 int64_t n_rays, int n_samples, int has_bg, float bg,
                                 const float* d_rgb, const float* d_depth, const float* d_opacity, const float* d_prob,
                                 float* d_rgb_s, float* d_sigma_s, float* d_ray, niw_stream_t stream) {
    NIW_REQUIRE(ray && rgb_s && sigma_s && depth_s && d_rgb_s && d_sigma_s && d_ray, "niw_composite_bwd: null pointer");
    NIW_REQUIRE(n_rays > 0 && n_samples >= 2 && n_samples <= 1024, "niw_composite_bwd: need 2 <= S <= 1024 (S=%d)", n_samples);
    const int blocks = (int)((n_rays + 3) / 4);
    composite_bwd_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(ray, rgb_s, sigma_s, depth_s, n_rays, n_samples, has_bg, bg,
                                                                 d_rgb, d_depth, d_opacity, d_prob, d_rgb_s, d_sigma_s, d_ray);
    NIW_LAUNCH_CHECK("niw_composite_bwd");
    return NIW_OK;
}
