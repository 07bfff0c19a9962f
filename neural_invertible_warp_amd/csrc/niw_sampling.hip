// Depth sampling (stratified + inverse-CDF resampling + merge), ray generation, photometric
// loss and the Adam update.  All are tiny next to the field MLP; they exist so that the whole
// step stays on the device without host round trips.
#include "niw_common.h"
#include "niw_loss_device.h"
#include "niw_warp_prep_device.h"

namespace {

// ---------------------------------------------------------------- S1: Graph.sample_depth (nerf.py:334-344)
// `span` = fl32(depth_max - depth_min) with the difference formed in double by the host entry point (include/niw.h)
__global__ void sample_stratified_kernel(const float* __restrict__ u, long long n, int S, float dmin, float span, int inverse,
                                         float* __restrict__ depth) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int s = (int)(i % S);
    const float r = (u ? u[i] : 0.5f) + (float)s;
    // rand_samples / S * (max - min) + min, each op rounded on its own
    float d = niw::add_rn(niw::mul_rn(__fdiv_rn(r, (float)S), span), dmin);
    if (inverse) d = __fdiv_rn(1.f, niw::add_rn(d, 1e-8f));
    depth[i] = d;
}

// The same with the stratified draw made IN the kernel (the reference calls torch.rand on the device inside the
// path, nerf.py:337).  Philox4x32-10 (Salmon et al., SC'11), counter = (index of the group of four consecutive samples, draw number),
// key = seed: every group of four samples owns one counter value, so the stream is a pure function of (seed, draw, sample index) --
// reproducible, resumable by the iteration number, independent of the launch geometry, and replayable from a captured graph with the
// draw number read from device memory.  u = the upper 24 bits * 2^-24, in [0, 1) like torch.rand's float32.
__device__ __forceinline__ void philox4x32_10(unsigned (&c)[4], unsigned k0, unsigned k1) {
#pragma unroll
    for (int round = 0; round < 10; ++round) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c[0], p1 = (unsigned long long)0xCD9E8D57u * c[2];
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1;
        c[1] = (unsigned)p1; c[3] = (unsigned)p0; c[0] = n0; c[2] = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

__global__ void sample_stratified_rng_kernel(unsigned long long seed, unsigned long long draw, const unsigned long long* __restrict__ draw_dev,
                                             long long n, int S, float dmin, float span, int inverse, float* __restrict__ depth,
                                             float* __restrict__ u_out) {
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;      // group of four consecutive samples
    if (4 * g >= n) return;
    if (draw_dev) draw = *draw_dev;
    unsigned c[4] = {(unsigned)g, (unsigned)(g >> 32), (unsigned)draw, (unsigned)(draw >> 32)};
    philox4x32_10(c, (unsigned)seed, (unsigned)(seed >> 32));
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const long long i = 4 * g + t;
        if (i >= n) break;
        const float u = (float)(c[t] >> 8) * 5.9604644775390625e-08f;           // 2^-24
        const float r = u + (float)(int)(i % S);
        float d = niw::add_rn(niw::mul_rn(__fdiv_rn(r, (float)S), span), dmin);
        if (inverse) d = __fdiv_rn(1.f, niw::add_rn(d, 1e-8f));
        depth[i] = d;
        if (u_out) u_out[i] = u;
    }
}

// Density noise (reference model/nerf.py:428-429: density += randn_like(density) * density_noise_reg in train mode) drawn in a kernel:
// the same Philox4x32-10 stream layout as the stratified draw -- counter = (group of four consecutive samples, draw number), key = seed --
// and Box-Muller on the two uniform pairs of a counter value: r = sqrt(-2 ln u1) with u1 in (0, 1] from the upper 24 bits, theta = 2 pi u2;
// samples 4g .. 4g+3 = r1 cos t1, r1 sin t1, r2 cos t2, r2 sin t2, times `scale`.  fp32 libm (logf, sqrtf, sincosf): a pure function of
// (seed, draw, sample index), reproducible and replayable from a captured graph (draw_dev); the reference's torch.randn stream is not
// reproduced (no fixture could pin it: the reference draws on whatever device it runs on), its distribution is.
__global__ void normal_rng_kernel(unsigned long long seed, unsigned long long draw, const unsigned long long* __restrict__ draw_dev, long long n,
                                  float scale, float* __restrict__ out) {
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (4 * g >= n) return;
    if (draw_dev) draw = *draw_dev;
    unsigned c[4] = {(unsigned)g, (unsigned)(g >> 32), (unsigned)draw, (unsigned)(draw >> 32)};
    philox4x32_10(c, (unsigned)seed, (unsigned)(seed >> 32));
    float z[4];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const float u1 = (float)((c[2 * t] >> 8) + 1u) * 5.9604644775390625e-08f;        // (0, 1]
        const float u2 = (float)(c[2 * t + 1] >> 8) * 5.9604644775390625e-08f;           // [0, 1)
        const float r = sqrtf(niw::mul_rn(-2.f, logf(u1)));
        float sn, cs;
        sincosf(niw::mul_rn(6.28318530717958647692f, u2), &sn, &cs);
        z[2 * t] = niw::mul_rn(r, cs);
        z[2 * t + 1] = niw::mul_rn(r, sn);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
        if (4 * g + t < n) out[4 * g + t] = niw::mul_rn(z[t], scale);
}

// ---------------------------------------------------------------- H1 + H2: nerf.py:346-365, 313-315
// One wave per ray.  cdf in LDS (accumulated sequentially in fp64 and rounded per element, the
// arithmetic of the CPU reference's cumsum), Sf binary searches, then a rank sort of the S+Sf
// depths (ties broken by position) which reproduces an ascending sort exactly.
__global__ __launch_bounds__(256) void sample_pdf_merge_kernel(const float* __restrict__ pdf, const float* __restrict__ coarse,
                                                               const float* __restrict__ unif, const float* __restrict__ bins,
                                                               long long n_rays, int S, int Sf, float* __restrict__ fine_out,
                                                               float* __restrict__ merged) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long r = (long long)blockIdx.x * 4 + wv;
    const int n = S + Sf;
    float* cdf = smem + wv * (S + 1 + n);
    float* val = cdf + S + 1;
    if (r >= n_rays) return;
    if (lane == 0) {
        double acc = 0.0;
        cdf[0] = 0.f;
        for (int i = 0; i < S; ++i) {
            acc += (double)pdf[r * S + i];
            cdf[i + 1] = (float)acc;
        }
    }
    for (int i = lane; i < S; i += 64) val[i] = coarse[r * S + i];
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0);
    for (int jf = lane; jf < Sf; jf += 64) {
        const float uu = unif[jf];
        // searchsorted(cdf, u, right=True): first index with cdf[idx] > u, in [0, S+1]
        int lo = 0, hi = S + 1;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cdf[mid] > uu) hi = mid; else lo = mid + 1;
        }
        const int il = max(lo - 1, 0), ih = min(lo, S);
        const float cl = cdf[il], ch = cdf[ih], dl = bins[il], dh = bins[ih];
        const float t = __fdiv_rn(uu - cl, niw::add_rn(ch - cl, 1e-8f));
        const float d = niw::add_rn(dl, niw::mul_rn(t, dh - dl));
        val[S + jf] = d;
        if (fine_out) fine_out[r * Sf + jf] = d;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0);
    for (int e = lane; e < n; e += 64) {
        const float v = val[e];
        int rank = 0;
        for (int k = 0; k < n; ++k) {
            const float o = val[k];
            rank += (o < v) || (o == v && k < e);
        }
        merged[r * n + rank] = v;
    }
}

// ---------------------------------------------------------------- R1 / R2: camera.py:359-390, 419-443
__device__ __forceinline__ void inv3x3(const float* __restrict__ K, float (&o)[9]) {
    const double a = K[0], b = K[1], c = K[2], d = K[3], e = K[4], f = K[5], g = K[6], h = K[7], i = K[8];
    const double A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
    const double det = a * A + b * B + c * C, id = 1.0 / det;
    o[0] = (float)(A * id); o[1] = (float)(-(b * i - c * h) * id); o[2] = (float)((b * f - c * e) * id);
    o[3] = (float)(B * id); o[4] = (float)((a * i - c * g) * id);  o[5] = (float)(-(a * f - c * d) * id);
    o[6] = (float)(C * id); o[7] = (float)(-(a * h - b * g) * id); o[8] = (float)((a * e - b * d) * id);
}

__global__ void raygen_kernel(const float* __restrict__ intr, const float* __restrict__ pose, const int64_t* __restrict__ ray_idx,
                              long long first_pixel, int B, long long R, int H, int W, int mode, long long view_stride,
                              float* __restrict__ out_a, float* __restrict__ out_b) {
    // view_stride: output rows between two views (R: dense [B,R,3] outputs; 2R: the two halves of a stacked [B,2R,3] point set)
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)B * R) return;
    const int b = (int)(i / R);
    const long long r = i % R;
    const long long o = b * view_stride + r;
    const long long pix = ray_idx ? ray_idx[r] : first_pixel + r;
    const float x = (float)(pix % W) + 0.5f, y = (float)(pix / W) + 0.5f;
    float Ki[9];
    inv3x3(intr + b * 9, Ki);
    float g[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) g[k] = Ki[k * 3] * x + Ki[k * 3 + 1] * y + Ki[k * 3 + 2];
    float c[3] = {0.f, 0.f, 0.f};
    if (pose) {
        // cam2world: X_hom @ [R^T | -R^T t]^T  (pose is world->camera, camera.py:89-95, 343-346)
        const float* P = pose + b * 12;
        float gw[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float tk = -(P[0 * 4 + k] * P[3] + P[1 * 4 + k] * P[7] + P[2 * 4 + k] * P[11]);
            gw[k] = P[0 * 4 + k] * g[0] + P[1 * 4 + k] * g[1] + P[2 * 4 + k] * g[2] + tk;
            c[k] = tk;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) g[k] = gw[k];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        out_a[o * 3 + k] = c[k];
        out_b[o * 3 + k] = mode == 1 ? g[k] - c[k] : g[k];
    }
}

// ---------------------------------------------------------------- R3: camera.convert_NDC (camera.py:523-540)
__global__ void ndc_kernel(const float* __restrict__ center, const float* __restrict__ ray, const float* __restrict__ intr,
                           int B, long long R, float near, float* __restrict__ oc, float* __restrict__ orr) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)B * R) return;
    const int b = (int)(i / R);
    const float rx = ray[i * 3], ry = ray[i * 3 + 1], rz = ray[i * 3 + 2];
    const float t = (near - center[i * 3 + 2]) / rz;
    const float cx = center[i * 3] + t * rx, cy = center[i * 3 + 1] + t * ry, cz = center[i * 3 + 2] + t * rz;
    const float sx = intr[b * 9 + 0] / intr[b * 9 + 2], sy = intr[b * 9 + 4] / intr[b * 9 + 5];
    oc[i * 3] = sx * (cx / cz); oc[i * 3 + 1] = sy * (cy / cz); oc[i * 3 + 2] = 1.f - 2.f * near / cz;
    orr[i * 3] = sx * (rx / rz - cx / cz); orr[i * 3 + 1] = sy * (ry / rz - cy / cz); orr[i * 3 + 2] = 2.f * near / cz;
}

// The reverse pass of ndc_kernel (round 6): d(centre), d(ray) of the camera-frame rays from the gradients of the NDC centre and ray --
// what autograd does to the reference's formulas (camera.py:523-540) when warped rays are re-parametrised in training.  With
// t = (near - c_z) / r_z, p = c + t r (the origin slid onto the near plane), a = p_x / p_z, b = p_y / p_z:
//   c' = (sx a, sy b, 1 - 2 near / p_z),   r' = (sx (r_x / r_z - a), sy (r_y / r_z - b), 2 near / p_z).
// No gradient reaches the intrinsics (data).
__global__ void ndc_bwd_kernel(const float* __restrict__ center, const float* __restrict__ ray, const float* __restrict__ intr, int B, long long R,
                               float near, const float* __restrict__ g_oc, const float* __restrict__ g_or, float* __restrict__ g_c,
                               float* __restrict__ g_r) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)B * R) return;
    const int b = (int)(i / R);
    const float rx = ray[i * 3], ry = ray[i * 3 + 1], rz = ray[i * 3 + 2];
    const float c0 = center[i * 3], c1 = center[i * 3 + 1], c2 = center[i * 3 + 2];
    const float t = (near - c2) / rz;
    const float px = c0 + t * rx, py = c1 + t * ry, pz = c2 + t * rz;
    const float sx = intr[b * 9 + 0] / intr[b * 9 + 2], sy = intr[b * 9 + 4] / intr[b * 9 + 5];
    const float gc0 = g_oc ? g_oc[i * 3] : 0.f, gc1 = g_oc ? g_oc[i * 3 + 1] : 0.f, gc2 = g_oc ? g_oc[i * 3 + 2] : 0.f;
    const float gr0 = g_or ? g_or[i * 3] : 0.f, gr1 = g_or ? g_or[i * 3 + 1] : 0.f, gr2 = g_or ? g_or[i * 3 + 2] : 0.f;
    const float ipz = 1.f / pz, irz = 1.f / rz;
    const float ga = sx * (gc0 - gr0), gb = sy * (gc1 - gr1);                       // d a, d b
    const float gpx = ga * ipz, gpy = gb * ipz;
    const float gpz = ((gc2 - gr2) * 2.f * near - ga * px - gb * py) * ipz * ipz;  // 1 - 2 near / p_z, 2 near / p_z, a, b
    float drx = sx * gr0 * irz, dry = sy * gr1 * irz;
    float drz = -(sx * gr0 * rx + sy * gr1 * ry) * irz * irz;
    const float gt = gpx * rx + gpy * ry + gpz * rz;                                // p = c + t r
    drx += t * gpx; dry += t * gpy; drz += t * gpz;
    drz -= gt * t * irz;                                                             // t = (near - c_z) / r_z
    g_c[i * 3] = gpx; g_c[i * 3 + 1] = gpy; g_c[i * 3 + 2] = gpz - gt * irz;
    g_r[i * 3] = drx; g_r[i * 3 + 1] = dry; g_r[i * 3 + 2] = drz;
}

// ---------------------------------------------------------------- L1: MSE (base.py:209-211) + gather (nerf.py:279-281)
// one workgroup of 1024 threads over all B*R*3 elements: fixed-order reduction, loss[0] overwritten
__global__ __launch_bounds__(1024) void mse_kernel(const float* __restrict__ rgb, const float* __restrict__ image,
                                                   const int64_t* __restrict__ ray_idx, int B, long long R, long long hw,
                                                   long long first_ray, long long n_rays,
                                                   double n_norm, float grad_scale, float* __restrict__ loss, float* __restrict__ d_rgb) {
    // rgb / d_rgb hold the rays first_ray .. first_ray + n_rays - 1 of the flattened (view-major) [B][R] ray list: the whole batch, or one
    // rank's contiguous share of it under ray sharding
    __shared__ double red[16];
    const long long total = n_rays * 3;
    const unsigned first32 = (unsigned)first_ray;
    const double slope = (double)grad_scale * 2.0 / n_norm;        // d mean / d rgb = 2 diff / n (one fp64 divide per launch, not per element)
    double acc = 0.0;
    // Eight elements per thread per round, every load of the round issued before the first use: the pixel index and the image
    // value are two dependent cold reads each (the rolled loop paid that latency per element: 18 us for 6 k elements), and the
    // element -> (view, ray, channel) split is 32-bit arithmetic (64-bit divisions were most of the rest).
    constexpr int U = 8;
    const unsigned R32 = (unsigned)R;
    for (long long base = 0; base < total; base += 1024ll * U) {
        float pred[U], img[U];
        bool ok[U];
        long long idx[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            idx[k] = base + threadIdx.x + 1024ll * k;
            ok[k] = idx[k] < total;
            const unsigned i = ok[k] ? (unsigned)idx[k] : 0u;             // host guarantees total < 2^32
            const unsigned lr = i / 3u, c = i - lr * 3u, br = first32 + lr, b = br / R32, r = br - b * R32;
            const long long pix = ray_idx ? ray_idx[r] : (long long)r;
            pred[k] = rgb[i];
            img[k] = image[((long long)b * 3 + c) * hw + pix];
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const float diff = ok[k] ? pred[k] - img[k] : 0.f;
            acc += diff * diff;
            if (d_rgb && ok[k]) d_rgb[idx[k]] = (float)(slope * (double)diff);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < 16; ++w) t += red[w];
        loss[0] = (float)(t / n_norm);
    }
}

// the loss value from residuals left by the training form of the compositing kernel (niw_composite_mse_train): same sum, same order
__global__ __launch_bounds__(256) void mse_from_residuals_kernel(const float* __restrict__ resid, long long total, double n_norm, float* __restrict__ loss) {
    __shared__ double red[16];
    const double t = niw::sq_sum_in_mse_order(resid, total, red);
    if (threadIdx.x == 0) loss[0] = (float)(t / n_norm);
}

// ---------------------------------------------------------------- torch.optim.Adam (single tensor, no amsgrad / weight decay)
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            long long n, float w1, float b2, float w2, float eps, float step_size, float bc2_sqrt,
                            const float* __restrict__ hyper_dev) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (hyper_dev) { step_size = hyper_dev[0]; bc2_sqrt = hyper_dev[1]; }
    float pi = p[i], mi = m[i], vi = v[i];
    niw::adam_update(pi, g[i], mi, vi, w1, b2, w2, eps, step_size, bc2_sqrt);
    p[i] = pi; m[i] = mi; v[i] = vi;
}

// ---------------------------------------------------------------- G0: var.ray_idx = randperm(H*W)[:n] (nerf_inn_llff.py:510)
__device__ __forceinline__ unsigned mix32(unsigned x) {          // murmur3 finaliser
    x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
    return x;
}
// keyed permutation of [0, 2^(2*half)): 4-round balanced Feistel network
__device__ __forceinline__ unsigned long long feistel(unsigned long long v, int half, const unsigned (&key)[4]) {
    const unsigned mask = (1u << half) - 1u;
    unsigned l = (unsigned)(v >> half) & mask, r = (unsigned)v & mask;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned f = mix32(r ^ key[k]) & mask;
        const unsigned t = l ^ f;
        l = r;
        r = t;
    }
    return ((unsigned long long)l << half) | r;
}
__global__ void draw_ray_idx_kernel(long long n_pixels, long long n, unsigned long long seed, unsigned long long draw,
                                    const unsigned long long* __restrict__ draw_dev, long long first, long long stride, int half,
                                    int64_t* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (draw_dev) draw = draw_dev[0];
    unsigned key[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
        key[k] = mix32((unsigned)(seed >> (16 * (k & 1))) + 0x9e3779b9u * (unsigned)(k + 1)) ^ mix32((unsigned)draw + 0x7f4a7c15u * (unsigned)(k + 1)) ^
                 mix32((unsigned)(draw >> 32) ^ (unsigned)(seed >> 32));
    unsigned long long v = (unsigned long long)(first + i * stride);
    do { v = feistel(v, half, key); } while (v >= (unsigned long long)n_pixels);       // cycle walking keeps it a permutation of [0, n_pixels)
    out[i] = (int64_t)v;
}

// draw_ray_idx_kernel + raygen_kernel (mode 0, stacked outputs) in one launch: every (view, ray) thread forms its ray's pixel index from
// the keyed permutation itself (a few dozen integer instructions), the threads of the first view also leave the indices behind for the
// photometric loss.  Same indices, same points.
__device__ __forceinline__ void draw_raygen_body(long long i, long long n_pixels, unsigned long long seed, unsigned long long draw,
                                                 const unsigned long long* __restrict__ draw_dev, int half, const float* __restrict__ intr,
                                                 const float* __restrict__ pose, int B, long long R, int W, int64_t* __restrict__ ray_idx,
                                                 float* __restrict__ stacked) {
    if (i >= (long long)B * R) return;
    const int b = (int)(i / R);
    const long long r = i % R;
    if (draw_dev) draw = draw_dev[0];
    unsigned key[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
        key[k] = mix32((unsigned)(seed >> (16 * (k & 1))) + 0x9e3779b9u * (unsigned)(k + 1)) ^ mix32((unsigned)draw + 0x7f4a7c15u * (unsigned)(k + 1)) ^
                 mix32((unsigned)(draw >> 32) ^ (unsigned)(seed >> 32));
    unsigned long long v = (unsigned long long)r;
    do { v = feistel(v, half, key); } while (v >= (unsigned long long)n_pixels);
    const long long pix = (long long)v;
    if (b == 0) ray_idx[r] = pix;
    const float x = (float)(pix % W) + 0.5f, y = (float)(pix / W) + 0.5f;
    float Ki[9];
    inv3x3(intr + b * 9, Ki);
    float g[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) g[k] = Ki[k * 3] * x + Ki[k * 3 + 1] * y + Ki[k * 3 + 2];
    float c[3] = {0.f, 0.f, 0.f};
    if (pose) {
        const float* P = pose + b * 12;
        float gw[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float tk = -(P[0 * 4 + k] * P[3] + P[1 * 4 + k] * P[7] + P[2 * 4 + k] * P[11]);
            gw[k] = P[0 * 4 + k] * g[0] + P[1 * 4 + k] * g[1] + P[2 * 4 + k] * g[2] + tk;
            c[k] = tk;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) g[k] = gw[k];
    }
    const long long o = b * 2 * R + r;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        stacked[o * 3 + k] = g[k];                 // [grid ; centre] per view
        stacked[(o + R) * 3 + k] = c[k];
    }
}

// The FRONT of a train iteration as one launch (niw_step.hip): everything that depends on nothing but the iteration number and the
// parameters -- pixel draw + un-warped points, stratified depths, the fp32 weight images of the field network(s) (the gather of
// niw_mlp_pack_weights_indexed), the zero pad columns of the warp backward's factor rows.  Block ranges select the job; every job is
// the body of the stand-alone kernel of the same name.
struct FrontArgs {
    int first_block[6];            // jobs: 0 rays, 1 depths, 2 weight image (coarse), 3 weight image (fine), 4 pad columns, 5 the warp's code projection
    int end_block;
    // code projection (job 5): one workgroup per (coupling block, view)
    const float* warp_params; const float* code; float* codeb; int n_code_views;
    // rays
    long long n_pixels; unsigned long long seed, draw; const unsigned long long* draw_dev; int half;
    const float* intr; const float* pose; int B; long long R; int W; int64_t* ray_idx; float* stacked;
    // depths
    unsigned long long depth_seed; long long n_depth; int S; float dmin, span; int inverse, stratified; float* depth;
    // weight images
    const float* params[2]; const int4* index; f32x4* packed[2];
    // pad columns
    float* pad_ws; long long pad_rows, ppad; int first_pad, n_pad;
};

__global__ void step_front_kernel(FrontArgs a) {
    const int blk = blockIdx.x;
    int job = 0;
#pragma unroll
    for (int k = 1; k < 6; ++k)
        if (blk >= a.first_block[k]) job = k;
    if (job == 5) {                                                   // (whole workgroups: the body uses wave butterflies)
        const int w = blk - a.first_block[5];
        niw_warp_prep::code_projection(w % 3, w / 3, a.warp_params, a.code, a.n_code_views, a.codeb);
        return;
    }
    const long long i = (long long)(blk - a.first_block[job]) * blockDim.x + threadIdx.x;
    if (job == 0) {
        draw_raygen_body(i, a.n_pixels, a.seed, a.draw, a.draw_dev, a.half, a.intr, a.pose, a.B, a.R, a.W, a.ray_idx, a.stacked);
    } else if (job == 1) {
        if (4 * i >= a.n_depth) return;
        unsigned long long draw = a.draw_dev ? *a.draw_dev : a.draw;
        unsigned c[4] = {(unsigned)i, (unsigned)(i >> 32), (unsigned)draw, (unsigned)(draw >> 32)};
        if (a.stratified) philox4x32_10(c, (unsigned)a.depth_seed, (unsigned)(a.depth_seed >> 32));
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const long long e = 4 * i + t;
            if (e >= a.n_depth) break;
            const float u = a.stratified ? (float)(c[t] >> 8) * 5.9604644775390625e-08f : 0.5f;
            const float r = u + (float)(int)(e % a.S);
            float d = niw::add_rn(niw::mul_rn(__fdiv_rn(r, (float)a.S), a.span), a.dmin);
            if (a.inverse) d = __fdiv_rn(1.f, niw::add_rn(d, 1e-8f));
            a.depth[e] = d;
        }
    } else if (job == 2 || job == 3) {
        if (i >= niw::kPackedFloats / 4) return;
        const float* __restrict__ params = a.params[job - 2];
        const int4 s = a.index[i];
        a.packed[job - 2][i] = f32x4{s.x >= 0 ? params[s.x] : 0.f, s.y >= 0 ? params[s.y] : 0.f, s.z >= 0 ? params[s.z] : 0.f, s.w >= 0 ? params[s.w] : 0.f};
    } else {
        if (i < a.pad_rows * a.n_pad) a.pad_ws[(i / a.n_pad) * a.ppad + a.first_pad + (int)(i % a.n_pad)] = 0.f;
    }
}

__global__ void draw_raygen_kernel(long long n_pixels, unsigned long long seed, unsigned long long draw, const unsigned long long* __restrict__ draw_dev,
                                   int half, const float* __restrict__ intr, const float* __restrict__ pose, int B, long long R, int W,
                                   int64_t* __restrict__ ray_idx, float* __restrict__ stacked) {
    draw_raygen_body((long long)blockIdx.x * blockDim.x + threadIdx.x, n_pixels, seed, draw, draw_dev, half, intr, pose, B, R, W, ray_idx, stacked);
}

}  // namespace

extern "C" int niw_sample_stratified(const float* u, int64_t n_rays, int n_samples, double depth_min, double depth_max,
                                     int inverse, float* depth, niw_stream_t stream) {
    NIW_REQUIRE(depth, "niw_sample_stratified: null output");
    NIW_REQUIRE(n_rays > 0 && n_samples > 0, "niw_sample_stratified: empty input");
    const long long n = n_rays * (long long)n_samples;
    sample_stratified_kernel<<<(int)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(u, n, n_samples, (float)depth_min, (float)(depth_max - depth_min), inverse, depth);
    NIW_LAUNCH_CHECK("niw_sample_stratified");
    return NIW_OK;
}

extern "C" int niw_sample_stratified_rng(uint64_t seed, uint64_t draw, const uint64_t* draw_dev, int64_t n_rays, int n_samples,
                                         double depth_min, double depth_max, int inverse, float* depth, float* u_out, niw_stream_t stream) {
    NIW_REQUIRE(depth, "niw_sample_stratified_rng: null output");
    NIW_REQUIRE(n_rays > 0 && n_samples > 0, "niw_sample_stratified_rng: empty input");
    const long long n = n_rays * (long long)n_samples, groups = (n + 3) / 4;
    sample_stratified_rng_kernel<<<(int)((groups + 255) / 256), 256, 0, (hipStream_t)stream>>>(
        seed, draw, reinterpret_cast<const unsigned long long*>(draw_dev), n, n_samples, (float)depth_min, (float)(depth_max - depth_min), inverse, depth, u_out);
    NIW_LAUNCH_CHECK("niw_sample_stratified_rng");
    return NIW_OK;
}

extern "C" int niw_normal_rng(uint64_t seed, uint64_t draw, const uint64_t* draw_dev, int64_t n, float scale, float* out, niw_stream_t stream) {
    NIW_REQUIRE(out && n > 0, "niw_normal_rng: null output or empty draw");
    const long long groups = (n + 3) / 4;
    normal_rng_kernel<<<(int)((groups + 255) / 256), 256, 0, (hipStream_t)stream>>>(seed, draw, reinterpret_cast<const unsigned long long*>(draw_dev), n, scale, out);
    NIW_LAUNCH_CHECK("niw_normal_rng");
    return NIW_OK;
}

extern "C" int niw_sample_pdf_merge(const float* pdf, const float* depth_coarse, const float* unif, const float* bins,
                                    int64_t n_rays, int n_samples, int n_fine, float* depth_fine, float* depth_merged,
                                    niw_stream_t stream) {
    NIW_REQUIRE(pdf && depth_coarse && unif && bins && depth_merged, "niw_sample_pdf_merge: null pointer");
    NIW_REQUIRE(n_rays > 0 && n_samples > 0 && n_fine > 0, "niw_sample_pdf_merge: empty input");
    NIW_REQUIRE(n_samples + n_fine <= 1024, "niw_sample_pdf_merge: S+Sf=%d exceeds 1024", n_samples + n_fine);
    const size_t lds = 4 * (size_t)(n_samples + 1 + n_samples + n_fine) * sizeof(float);
    sample_pdf_merge_kernel<<<(int)((n_rays + 3) / 4), 256, lds, (hipStream_t)stream>>>(pdf, depth_coarse, unif, bins, n_rays, n_samples,
                                                                                       n_fine, depth_fine, depth_merged);
    NIW_LAUNCH_CHECK("niw_sample_pdf_merge");
    return NIW_OK;
}

extern "C" int niw_raygen(const float* intr, const float* pose, const int64_t* ray_idx, int64_t first_pixel, int n_views,
                          int64_t n_rays_per_view, int H, int W, int mode, float* out_a, float* out_b, niw_stream_t stream) {
    NIW_REQUIRE(intr && out_a && out_b, "niw_raygen: null pointer");
    NIW_REQUIRE(n_views > 0 && n_rays_per_view > 0 && H > 0 && W > 0, "niw_raygen: empty input");
    NIW_REQUIRE(mode == 0 || (mode == 1 && pose), "niw_raygen: mode %d needs a pose", mode);
    NIW_REQUIRE(ray_idx || (first_pixel >= 0 && first_pixel + n_rays_per_view <= (int64_t)H * W),
                "niw_raygen: pixel range [%lld, %lld) leaves the %dx%d image", (long long)first_pixel, (long long)(first_pixel + n_rays_per_view), H, W);
    const long long n = (long long)n_views * n_rays_per_view;
    raygen_kernel<<<(int)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(intr, pose, ray_idx, first_pixel, n_views, n_rays_per_view, H, W, mode,
                                                                           n_rays_per_view, out_a, out_b);
    NIW_LAUNCH_CHECK("niw_raygen");
    return NIW_OK;
}

// The front of a train iteration in one launch (step_front_kernel): niw_draw_ray_idx + stacked mode-0 ray generation; niw_sample_stratified(_rng);
// niw_mlp_pack_weights_indexed of one or two networks (packed[k] == NULL: skipped); the pad columns [n_cols, ppad) of `pad_rows` factor rows;
// the warp's code projection (codeb == NULL: skipped), the first of niw_warp_prep_fwd's two launches
int niw_launch_step_front(int64_t n_pixels, uint64_t seed, uint64_t draw, const uint64_t* draw_dev, const float* intr, const float* pose, int n_views,
                          long long R, int H, int W, int64_t* ray_idx, float* stacked,
                          uint64_t depth_seed, int stratified, long long n_rays, int S, double depth_min, double depth_max, int inverse, float* depth,
                          const float* params0, const float* params1, const int32_t* index, float* packed0, float* packed1,
                          float* pad_ws, long long pad_rows, long long ppad, long long n_cols,
                          const float* warp_params, const float* code, int n_code_views, float* codeb, hipStream_t st) {
    NIW_REQUIRE(n_pixels == (int64_t)H * W && R > 0 && R <= n_pixels && n_pixels <= (1ll << 40), "niw_train_step (pixel draw): %lld rays of %lld pixels", R, (long long)n_pixels);
    NIW_REQUIRE(intr && ray_idx && stacked && depth && n_rays > 0 && S > 0, "niw_train_step (front): null pointer or empty input");
    NIW_REQUIRE(!codeb || (warp_params && code && n_code_views > 0), "niw_train_step (front): the code projection needs the warp parameters and the latent codes");
    NIW_REQUIRE((!packed0 && !packed1) || index, "niw_train_step (front): weight images need the pack index");
    FrontArgs a{};
    int bits = 1;
    while ((1ll << bits) < n_pixels) ++bits;
    a.n_pixels = n_pixels; a.seed = seed; a.draw = draw; a.draw_dev = reinterpret_cast<const unsigned long long*>(draw_dev);
    a.half = (bits + 1) / 2 < 1 ? 1 : (bits + 1) / 2;
    a.intr = intr; a.pose = pose; a.B = n_views; a.R = R; a.W = W; a.ray_idx = ray_idx; a.stacked = stacked;
    a.depth_seed = depth_seed; a.n_depth = n_rays * S; a.S = S; a.dmin = (float)depth_min; a.span = (float)(depth_max - depth_min);
    a.inverse = inverse; a.stratified = stratified; a.depth = depth;
    a.params[0] = params0; a.params[1] = params1; a.index = reinterpret_cast<const int4*>(index);
    a.packed[0] = reinterpret_cast<f32x4*>(packed0); a.packed[1] = reinterpret_cast<f32x4*>(packed1);
    const int n_pad = pad_ws ? (int)(ppad - n_cols) : 0;
    a.pad_ws = pad_ws; a.pad_rows = pad_rows; a.ppad = ppad; a.first_pad = (int)n_cols; a.n_pad = n_pad;
    a.warp_params = warp_params; a.code = code; a.codeb = codeb; a.n_code_views = n_code_views;
    const long long work[6] = {(long long)n_views * R, (a.n_depth + 3) / 4, packed0 ? niw::kPackedFloats / 4 : 0, packed1 ? niw::kPackedFloats / 4 : 0,
                               pad_rows * n_pad, codeb ? 256ll * 3 * n_code_views : 0};
    int blocks = 0;
    for (int k = 0; k < 6; ++k) {
        a.first_block[k] = blocks;
        blocks += (int)((work[k] + 255) / 256);
    }
    a.end_block = blocks;
    step_front_kernel<<<blocks, 256, 0, st>>>(a);
    NIW_LAUNCH_CHECK("niw_train_step (front)");
    return NIW_OK;
}

// niw_draw_ray_idx (first = 0, stride = 1) + the stacked mode-0 ray generation below as ONE launch
int niw_launch_draw_raygen_stacked(int64_t n_pixels, uint64_t seed, uint64_t draw, const uint64_t* draw_dev, const float* intr, const float* pose,
                                   int n_views, long long R, int H, int W, int64_t* ray_idx, float* stacked, hipStream_t st) {
    NIW_REQUIRE(n_pixels == (int64_t)H * W && R > 0 && R <= n_pixels && n_pixels <= (1ll << 40), "niw_train_step (pixel draw): %lld rays of %lld pixels", R, (long long)n_pixels);
    int bits = 1;
    while ((1ll << bits) < n_pixels) ++bits;
    const int half = (bits + 1) / 2 < 1 ? 1 : (bits + 1) / 2;
    const long long n = (long long)n_views * R;
    draw_raygen_kernel<<<(int)((n + 255) / 256), 256, 0, st>>>(n_pixels, seed, draw, reinterpret_cast<const unsigned long long*>(draw_dev), half, intr, pose,
                                                               n_views, R, W, ray_idx, stacked);
    NIW_LAUNCH_CHECK("niw_train_step (pixel draw + ray generation)");
    return NIW_OK;
}

// mode 0 straight into a stacked point set [n_views][2R][3] = [grid ; centre] per view (the warp's input batch; niw_step.hip)
int niw_launch_raygen_stacked(const float* intr, const float* pose, const int64_t* ray_idx, int n_views, long long R, int H, int W,
                              float* stacked, hipStream_t st) {
    const long long n = (long long)n_views * R;
    raygen_kernel<<<(int)((n + 255) / 256), 256, 0, st>>>(intr, pose, ray_idx, 0, n_views, R, H, W, 0, 2 * R, stacked + 3 * R, stacked);
    NIW_LAUNCH_CHECK("niw_train_step (ray generation)");
    return NIW_OK;
}

extern "C" int niw_mse_from_residuals(const float* resid, int64_t n_rays, double n_norm, float* loss, niw_stream_t stream) {
    NIW_REQUIRE(resid && loss, "niw_mse_from_residuals: null pointer");
    NIW_REQUIRE(n_rays > 0 && n_norm > 0, "niw_mse_from_residuals: empty input");
    mse_from_residuals_kernel<<<1, 256, 0, (hipStream_t)stream>>>(resid, (long long)n_rays * 3, n_norm, loss);
    NIW_LAUNCH_CHECK("niw_mse_from_residuals");
    return NIW_OK;
}

extern "C" int niw_convert_ndc(const float* center, const float* ray, const float* intr, int n_views, int64_t n_rays_per_view,
                               float near, float* center_ndc, float* ray_ndc, niw_stream_t stream) {
    NIW_REQUIRE(center && ray && intr && center_ndc && ray_ndc, "niw_convert_ndc: null pointer");
    NIW_REQUIRE(n_views > 0 && n_rays_per_view > 0, "niw_convert_ndc: empty input");
    const long long n = (long long)n_views * n_rays_per_view;
    ndc_kernel<<<(int)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(center, ray, intr, n_views, n_rays_per_view, near, center_ndc, ray_ndc);
    NIW_LAUNCH_CHECK("niw_convert_ndc");
    return NIW_OK;
}

extern "C" int niw_convert_ndc_bwd(const float* center, const float* ray, const float* intr, int n_views, int64_t n_rays_per_view, float near,
                                   const float* d_center_ndc, const float* d_ray_ndc, float* d_center, float* d_ray, niw_stream_t stream) {
    NIW_REQUIRE(center && ray && intr && d_center && d_ray && (d_center_ndc || d_ray_ndc), "niw_convert_ndc_bwd: null pointer");
    NIW_REQUIRE(n_views > 0 && n_rays_per_view > 0 && near > 0.f, "niw_convert_ndc_bwd: empty input or near <= 0");
    const long long n = (long long)n_views * n_rays_per_view;
    ndc_bwd_kernel<<<(int)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(center, ray, intr, n_views, n_rays_per_view, near, d_center_ndc, d_ray_ndc, d_center, d_ray);
    NIW_LAUNCH_CHECK("niw_convert_ndc_bwd");
    return NIW_OK;
}

extern "C" int niw_mse_fwd_bwd(const float* rgb, const float* image, const int64_t* ray_idx, int n_views,
                               int64_t n_rays_per_view, int64_t hw, int64_t first_ray, int64_t n_rays, double n_norm, float grad_scale,
                               float* loss, float* d_rgb, niw_stream_t stream) {
    NIW_REQUIRE(rgb && image && loss, "niw_mse_fwd_bwd: null pointer");
    NIW_REQUIRE(n_views > 0 && n_rays_per_view > 0 && hw > 0 && n_norm > 0, "niw_mse_fwd_bwd: empty input");
    NIW_REQUIRE((long long)n_views * n_rays_per_view * 3 < (1ll << 32), "niw_mse_fwd_bwd: at most 2^32 colour values per call");
    if (n_rays <= 0) { first_ray = 0; n_rays = (int64_t)n_views * n_rays_per_view; }      // the whole batch
    NIW_REQUIRE(first_ray >= 0 && first_ray + n_rays <= (int64_t)n_views * n_rays_per_view,
                "niw_mse_fwd_bwd: rays [%lld, %lld) leave the %d x %lld batch", (long long)first_ray, (long long)(first_ray + n_rays), n_views, (long long)n_rays_per_view);
    mse_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(rgb, image, ray_idx, n_views, n_rays_per_view, hw, first_ray, n_rays, n_norm, grad_scale, loss, d_rgb);
    NIW_LAUNCH_CHECK("niw_mse_fwd_bwd");
    return NIW_OK;
}

extern "C" int niw_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                             double lr, double beta1, double beta2, double eps, int step, const float* hyper_dev,
                             niw_stream_t stream) {
    NIW_REQUIRE(param && grad && exp_avg && exp_avg_sq, "niw_adam_step: null pointer");
    NIW_REQUIRE(n > 0 && (step >= 1 || hyper_dev), "niw_adam_step: n=%lld step=%d", (long long)n, step);
    if (step < 1) step = 1;
    // the scalars are formed in double like torch.optim.Adam forms them in Python floats, and rounded to fp32 once
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    adam_kernel<<<(int)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(param, grad, exp_avg, exp_avg_sq, n, (float)(1.0 - beta1), (float)beta2,
                                                                        (float)(1.0 - beta2), (float)eps, (float)(lr / bc1), (float)sqrt(bc2), hyper_dev);
    NIW_LAUNCH_CHECK("niw_adam_step");
    return NIW_OK;
}

extern "C" int niw_draw_ray_idx(int64_t n_pixels, int64_t n, uint64_t seed, uint64_t draw, const uint64_t* draw_dev, int64_t first,
                                int64_t stride, int64_t* out, niw_stream_t stream) {
    NIW_REQUIRE(out, "niw_draw_ray_idx: null output");
    NIW_REQUIRE(n_pixels > 0 && n > 0 && first >= 0 && stride >= 1, "niw_draw_ray_idx: n_pixels=%lld n=%lld first=%lld stride=%lld",
                (long long)n_pixels, (long long)n, (long long)first, (long long)stride);
    NIW_REQUIRE(first + (n - 1) * stride < n_pixels, "niw_draw_ray_idx: %lld indices from %lld by %lld leave the %lld-pixel permutation",
                (long long)n, (long long)first, (long long)stride, (long long)n_pixels);
    NIW_REQUIRE(n_pixels <= (1ll << 40), "niw_draw_ray_idx: at most 2^40 pixels");
    int bits = 1;
    while ((1ll << bits) < n_pixels) ++bits;
    const int half = (bits + 1) / 2 < 1 ? 1 : (bits + 1) / 2;
    draw_ray_idx_kernel<<<(int)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(n_pixels, n, seed, draw, (const unsigned long long*)draw_dev, first,
                                                                                  stride, half, out);
    NIW_LAUNCH_CHECK("niw_draw_ray_idx");
    return NIW_OK;
}
