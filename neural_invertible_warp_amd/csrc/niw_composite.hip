// Alpha compositing along rays: NeRF.composite (reference model/nerf.py:458-474), forward and
// closed-form backward.  HBM-bound: 20 B/sample in (+4 B/sample prob out) forward,
// 20 (+4) B in and 16 B out per sample backward.
//
// Vector kernels (S % 4 == 0, 16-byte aligned rows -- every shipped configuration).  A lane owns QUADS of four consecutive samples
// (sigma, depth, prob: one 16-byte access per quad); the transmittance T_i = exp(-sum_{j<i} sigma_j delta_j) is a segmented wave scan:
// a serial exclusive prefix over the quad plus a ladder over the quad totals of the ray's lane group (shift-then-scan: an
// inclusive-minus-self form would cancel catastrophically against the 1e10 closing interval); the backward is the mirrored suffix
// scan of g_k w_k.
//   S <= 256      16 lanes per ray -- one DPP row -- and 1 .. 4 quads per lane: the SPAN kernels below (round 5)
//   S  > 256      64 lanes per ray, one quad per lane, chunks of 256 samples with a running carry (composite_*_kernel<64>, shuffle ladder)
// Any other shape (S % 4 != 0, unaligned views) takes the scalar one-wave-per-ray kernels at the end of the file.
// What bounds the span kernels is the HBM rate of their read : write mix, and the mix only streams at the rate of a copy when every
// access instruction covers whole cache lines and carries the non-temporal hint (tools/composite_variants.hip, 120,000 x 192:
// plain streaming kernels of the forward's 5 : 1 mix 5.1 TB/s with default loads, 6.0 non-temporal; a copy 5.1 / 6.5).  The colours
// (12 B per sample: a lane's quad is 48 B) therefore move as fully coalesced 16-byte accesses and are transposed through LDS
// (forward 111 -> 88 us, backward 163 -> 137 us; with 48-byte lane strides the hint costs 40 %: partial lines are fetched again).
#include "niw_common.h"

namespace {

constexpr int kMaxChunks = 64;      // backward: LDS table of chunk prefixes, 256 samples per chunk -> S <= 16384

template <int G>
__device__ __forceinline__ float group_excl_scan_up(float v, int gl) {
    // exclusive prefix over the G lanes of a group: shift by one lane, then an inclusive ladder
    float s = __shfl_up(v, 1, G);
    if (gl == 0) s = 0.f;
#pragma unroll
    for (int o = 1; o < G; o <<= 1) {
        const float t = __shfl_up(s, o, G);
        if (gl >= o) s += t;
    }
    return s;
}
template <int G>
__device__ __forceinline__ float group_excl_scan_down(float v, int gl) {
    float s = __shfl_down(v, 1, G);
    if (gl == G - 1) s = 0.f;
#pragma unroll
    for (int o = 1; o < G; o <<= 1) {
        const float t = __shfl_down(s, o, G);
        if (gl + o < G) s += t;
    }
    return s;
}
template <int G>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, G);
    return v;
}

struct Quad {
    float sig[4], dep[4], itv[4], sd[4], col[4][3];
};

// The lane's four samples of chunk `k` of its ray: sigma, depth, the interval to the next sample (1e10 after the
// last one, nerf.py:461-462), sigma * (interval * |ray|) (nerf.py:463-464) and, when WITH_RGB, the colours.
template <int G, bool WITH_RGB>
__device__ __forceinline__ bool load_quad(Quad& q, const float* __restrict__ sg, const float* __restrict__ d, const float* __restrict__ c,
                                          int S, int k, int gl, float len) {
    const int s0 = k * 4 * G + 4 * gl;
    const bool v = s0 < S;
    f32x4 s4 = {0.f, 0.f, 0.f, 0.f}, d4 = s4;
    if (v) {
        s4 = *reinterpret_cast<const f32x4*>(sg + s0);
        d4 = *reinterpret_cast<const f32x4*>(d + s0);
    }
    if (WITH_RGB) {
        f32x4 c4[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) c4[j] = v ? *reinterpret_cast<const f32x4*>(c + 3 * s0 + 4 * j) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 12; ++t) q.col[t / 3][t % 3] = c4[t / 4][t % 4];
    }
    // depth of the sample after the lane's last one: the next lane's first (the group's last lane: next chunk, if any)
    float dn = __shfl_down(d4[0], 1, G);
    if (gl == G - 1) dn = (s0 + 4 < S) ? d[s0 + 4] : 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        q.sig[t] = s4[t];
        q.dep[t] = d4[t];
    }
    q.itv[0] = d4[1] - d4[0];
    q.itv[1] = d4[2] - d4[1];
    q.itv[2] = d4[3] - d4[2];
    q.itv[3] = (s0 + 4 >= S) ? 1e10f : dn - d4[3];
#pragma unroll
    for (int t = 0; t < 4; ++t) q.sd[t] = v ? q.sig[t] * (q.itv[t] * len) : 0.f;
    return v;
}

template <int G>
__global__ __launch_bounds__(256) void composite_fwd_kernel(const float* __restrict__ ray, const float* __restrict__ rgb_s,
                                                            const float* __restrict__ sigma_s, const float* __restrict__ depth_s,
                                                            long long n_rays, int S, int has_bg, float bg,
                                                            float* __restrict__ rgb, float* __restrict__ depth,
                                                            float* __restrict__ opacity, float* __restrict__ prob) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, gl = threadIdx.x % G;
    constexpr int RPW = 64 / G;                                  // rays per wave
    const long long r_wave = ((long long)blockIdx.x * 4 + wave) * RPW;
    if (r_wave >= n_rays) return;                               // wave-uniform: no ray left for this wave
    const long long r_raw = r_wave + lane / G;
    const bool live = r_raw < n_rays;
    const long long r = live ? r_raw : n_rays - 1;              // idle groups shadow the last ray (no stores)
    const float rx = ray[r * 3], ry = ray[r * 3 + 1], rz = ray[r * 3 + 2];
    const float len = sqrtf(rx * rx + ry * ry + rz * rz);
    const float* d = depth_s + r * S;
    const float* sg = sigma_s + r * S;
    const float* c = rgb_s + r * S * 3;
    const int n_chunks = (S + 4 * G - 1) / (4 * G);
    float carry = 0.f, a0 = 0.f, a1 = 0.f, a2 = 0.f, ad = 0.f, ao = 0.f;
    for (int k = 0; k < n_chunks; ++k) {
        Quad q;
        const bool v = load_quad<G, true>(q, sg, d, c, S, k, gl, len);
        const float e1 = q.sd[0], e2 = e1 + q.sd[1], e3 = e2 + q.sd[2], tot = e3 + q.sd[3];
        const float base = carry + group_excl_scan_up<G>(tot, gl);
        const float ex[4] = {base, base + e1, base + e2, base + e3};
        f32x4 w4;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float w = v ? expf(-ex[t]) * (1.f - expf(-q.sd[t])) : 0.f;
            w4[t] = w;
            a0 += w * q.col[t][0];
            a1 += w * q.col[t][1];
            a2 += w * q.col[t][2];
            ad += w * q.dep[t];
            ao += w;
        }
        if (v && live && prob) *reinterpret_cast<f32x4*>(prob + r * S + k * 4 * G + 4 * gl) = w4;
        if (k + 1 < n_chunks) carry = __shfl(base + tot, G - 1, G);      // every lane is valid when another chunk follows
    }
    a0 = group_sum<G>(a0); a1 = group_sum<G>(a1); a2 = group_sum<G>(a2); ad = group_sum<G>(ad); ao = group_sum<G>(ao);
    if (gl == 0 && live) {
        if (has_bg) { const float t = bg * (1.f - ao); a0 += t; a1 += t; a2 += t; }
        rgb[r * 3] = a0; rgb[r * 3 + 1] = a1; rgb[r * 3 + 2] = a2;
        depth[r] = ad;
        opacity[r] = ao;
    }
}

template <int G>
__global__ __launch_bounds__(256) void composite_bwd_kernel(const float* __restrict__ ray, const float* __restrict__ rgb_s,
                                                            const float* __restrict__ sigma_s, const float* __restrict__ depth_s,
                                                            long long n_rays, int S, int has_bg, float bg,
                                                            const float* __restrict__ g_rgb, const float* __restrict__ g_depth,
                                                            const float* __restrict__ g_opacity, const float* __restrict__ g_prob,
                                                            float* __restrict__ d_rgb_s, float* __restrict__ d_sigma_s,
                                                            float* __restrict__ d_ray) {
    __shared__ float chunk_prefix[G == 64 ? 4 * kMaxChunks : 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, gl = threadIdx.x % G;
    constexpr int RPW = 64 / G;
    const long long r_wave = ((long long)blockIdx.x * 4 + wave) * RPW;
    if (r_wave >= n_rays) return;                               // wave-uniform: no ray left for this wave
    const long long r_raw = r_wave + lane / G;
    const bool live = r_raw < n_rays;
    const long long r = live ? r_raw : n_rays - 1;
    const float rx = ray[r * 3], ry = ray[r * 3 + 1], rz = ray[r * 3 + 2];
    const float len = sqrtf(rx * rx + ry * ry + rz * rz);
    const float* d = depth_s + r * S;
    const float* sg = sigma_s + r * S;
    const float* c = rgb_s + r * S * 3;
    const float gr0 = g_rgb ? g_rgb[r * 3] : 0.f, gr1 = g_rgb ? g_rgb[r * 3 + 1] : 0.f, gr2 = g_rgb ? g_rgb[r * 3 + 2] : 0.f;
    const float gd = g_depth ? g_depth[r] : 0.f;
    float go = g_opacity ? g_opacity[r] : 0.f;
    if (has_bg) go -= bg * (gr0 + gr1 + gr2);
    const int n_chunks = (S + 4 * G - 1) / (4 * G);
    // pass 1 (rays of more than one chunk, G = 64): sum of sigma*delta in front of every chunk
    if (G == 64 && n_chunks > 1) {
        float run = 0.f;
        for (int k = 0; k < n_chunks; ++k) {
            Quad q;
            load_quad<G, false>(q, sg, d, c, S, k, gl, len);
            if (gl == 0) chunk_prefix[wave * kMaxChunks + k] = run;
            run += group_sum<G>((q.sd[0] + q.sd[1]) + (q.sd[2] + q.sd[3]));
        }
        __builtin_amdgcn_wave_barrier();
    }
    // pass 2: chunks in reverse with a suffix carry of g_k w_k
    float suffix = 0.f, dlen = 0.f;
    for (int k = n_chunks - 1; k >= 0; --k) {
        Quad q;
        const bool v = load_quad<G, true>(q, sg, d, c, S, k, gl, len);
        const int s0 = k * 4 * G + 4 * gl;
        f32x4 gp = {0.f, 0.f, 0.f, 0.f};
        if (v && g_prob) gp = *reinterpret_cast<const f32x4*>(g_prob + r * S + s0);
        const float e1 = q.sd[0], e2 = e1 + q.sd[1], e3 = e2 + q.sd[2], tot = e3 + q.sd[3];
        const float first = (G == 64 && n_chunks > 1) ? chunk_prefix[wave * kMaxChunks + k] : 0.f;
        const float base = first + group_excl_scan_up<G>(tot, gl);
        const float ex[4] = {base, base + e1, base + e2, base + e3};
        float T[4], E[4], w[4], g[4], gw[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            T[t] = expf(-ex[t]);
            E[t] = expf(-q.sd[t]);
            w[t] = v ? T[t] * (1.f - E[t]) : 0.f;
            g[t] = v ? gr0 * q.col[t][0] + gr1 * q.col[t][1] + gr2 * q.col[t][2] + gd * q.dep[t] + go + gp[t] : 0.f;
            gw[t] = g[t] * w[t];
        }
        // sum_{j>i} g_j w_j = later samples of the lane + later lanes of the group + later chunks
        const float b2 = gw[3], b1 = b2 + gw[2], b0 = b1 + gw[1], btot = b0 + gw[0];
        const float after = suffix + group_excl_scan_down<G>(btot, gl);
        const float aft[4] = {after + b0, after + b1, after + b2, after};
        f32x4 ds4, dc4[3];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float ds = g[t] * (T[t] * E[t]) - aft[t];           // dL/d(sigma*delta)
            ds4[t] = ds * (q.itv[t] * len);
            if (v) dlen += ds * (q.sig[t] * q.itv[t]);
        }
#pragma unroll
        for (int t = 0; t < 12; ++t) dc4[t / 4][t % 4] = w[t / 3] * (t % 3 == 0 ? gr0 : t % 3 == 1 ? gr1 : gr2);
        if (v && live) *reinterpret_cast<f32x4*>(d_sigma_s + r * S + s0) = ds4;
        if (v && live) {
#pragma unroll
            for (int j = 0; j < 3; ++j) *reinterpret_cast<f32x4*>(d_rgb_s + (r * S + s0) * 3 + 4 * j) = dc4[j];
        }
        if (k > 0) suffix = __shfl(after + btot, 0, G);
    }
    dlen = group_sum<G>(dlen);
    if (gl == 0 && live) {
        const float inv = len > 0.f ? dlen / len : 0.f;
        d_ray[r * 3] = inv * rx; d_ray[r * 3 + 1] = inv * ry; d_ray[r * 3 + 2] = inv * rz;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Span kernels (round 5): rays of 65..256 samples keep the SIXTEEN-lane group of the 64-sample form -- four rays per wave,
// one DPP row per ray -- and give every lane Q = 2 / 3 / 4 quads instead of widening the group to 32 / 64 lanes: no idle
// lanes at S = 192 (the 64-lane group left 16 of 64 idle), four-step ladders instead of six, the closing sums amortised
// over four rays, every load of the wave issued before the first use.  Within a 16-lane group the ladder steps are DPP row
// shifts on the vector ALU (row_shr / row_shl / row_ror with zero fill; the same sums in the same order as the shuffle
// ladder), not ds_bpermute round trips through the LDS crossbar.
// Quads are INTERLEAVED over the lanes: lane gl owns quads gl, gl + 16, gl + 32, .. -- every 16-byte access of a group is 256
// contiguous bytes (768 for the colours); Q ladders and Q - 1 broadcasts carry the sum from one row of quads into the next.
// (Measured against the consecutive assignment -- lane gl owns quads gl Q .. gl Q + Q - 1, one ladder, no broadcast: 202 us
// against 108 forward at 120,000 x 192, 217 against 155 backward; its loads put every lane of a wave on a cache line of
// its own, 9 instructions long: bound by the L1's tag rate, not by HBM.)
// WAVES: waves per workgroup (the kernels have no LDS and no barrier: a workgroup is just the unit CUs are refilled in).
// ------------------------------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_zero_fill(float v) {      // lanes whose source lies outside their 16-lane row read 0
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
constexpr int kRowShr = 0x110, kRowShl = 0x100, kRowRor = 0x120;
__device__ __forceinline__ float row16_excl_scan_up(float v) {
    float s = dpp_zero_fill<kRowShr | 1>(v);
    s += dpp_zero_fill<kRowShr | 1>(s);
    s += dpp_zero_fill<kRowShr | 2>(s);
    s += dpp_zero_fill<kRowShr | 4>(s);
    s += dpp_zero_fill<kRowShr | 8>(s);
    return s;
}
__device__ __forceinline__ float row16_excl_scan_down(float v) {
    float s = dpp_zero_fill<kRowShl | 1>(v);
    s += dpp_zero_fill<kRowShl | 1>(s);
    s += dpp_zero_fill<kRowShl | 2>(s);
    s += dpp_zero_fill<kRowShl | 4>(s);
    s += dpp_zero_fill<kRowShl | 8>(s);
    return s;
}
__device__ __forceinline__ float row16_sum(float v) {           // the row's total in every lane (lane 0's order is the one used)
    v += dpp_zero_fill<kRowRor | 8>(v);
    v += dpp_zero_fill<kRowRor | 4>(v);
    v += dpp_zero_fill<kRowRor | 2>(v);
    v += dpp_zero_fill<kRowRor | 1>(v);
    return v;
}

// diagnostic switches of tools/composite_variants.hip (the product instantiates <.., false, false>)
template <bool NT>
__device__ __forceinline__ f32x4 ld4(const float* p) {
    return NT ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p)) : *reinterpret_cast<const f32x4*>(p);
}
template <bool NT>
__device__ __forceinline__ void st4(float* p, f32x4 v) {
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
    else *reinterpret_cast<f32x4*>(p) = v;
}
template <bool FAST>
__device__ __forceinline__ float exp_neg(float x) { return FAST ? __expf(-x) : expf(-x); }

template <int Q>
struct Span {
    float sig[4 * Q], dep[4 * Q], itv[4 * Q], sd[4 * Q], col[4 * Q][3];
    bool v[Q];
    int s0[Q];
};

// The lane's Q quads of its ray: sigma, depth, the interval to the next sample (1e10 after the ray's last one), sigma *
// (interval * |ray|) and the colours.  Invalid quads (beyond S) read as zero and contribute nothing.
// XP: the colours arrive as coalesced 16-byte accesses -- lane gl takes float4 16 j + gl of the 48 a row of quads holds, whole cache
// lines per instruction, which is what lets them be non-temporal -- and reach the lane that owns their samples through `stage` (this
// ray's [Q][192] floats of LDS; written and read by the same wave, in order).
template <int Q, bool NT, bool XP = false>
__device__ __forceinline__ void load_span(Span<Q>& p, const float* __restrict__ sg, const float* __restrict__ d, const float* __restrict__ c,
                                          int S, int gl, float len, float* stage = nullptr) {
#pragma clang fp contract(off)
    constexpr int G = 16;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 s4[Q], d4[Q], c4[Q][3];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        p.s0[q] = 4 * (q * G + gl);
        p.v[q] = p.s0[q] < S;
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        s4[q] = p.v[q] ? ld4<NT>(sg + p.s0[q]) : zero;
        d4[q] = p.v[q] ? ld4<NT>(d + p.s0[q]) : zero;
    }
    if (XP) {
#pragma unroll
        for (int q = 0; q < Q; ++q)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int e = 192 * q + 4 * (16 * j + gl);                     // first float of the lane's float4 within the ray's colours
                c4[q][j] = e < 3 * S ? ld4<NT>(c + e) : zero;
            }
#pragma unroll
        for (int q = 0; q < Q; ++q)
#pragma unroll
            for (int j = 0; j < 3; ++j) *reinterpret_cast<f32x4*>(stage + 192 * q + 4 * (16 * j + gl)) = c4[q][j];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < Q; ++q)
#pragma unroll
            for (int j = 0; j < 3; ++j) c4[q][j] = *reinterpret_cast<const f32x4*>(stage + 192 * q + 12 * gl + 4 * j);
    } else {
#pragma unroll
        for (int q = 0; q < Q; ++q)
#pragma unroll
            for (int j = 0; j < 3; ++j) c4[q][j] = p.v[q] ? ld4<NT>(c + 3 * p.s0[q] + 4 * j) : zero;
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        // depth of the sample after the quad's last one: the next lane's quad of the same row; for the group's last lane, lane 0's quad of the next row
        float dn = dpp_zero_fill<kRowShl | 1>(d4[q][0]);
        if (q + 1 < Q) {
            const float wrap = __shfl(d4[q + 1][0], 0, G);
            if (gl == G - 1) dn = wrap;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            p.sig[4 * q + t] = s4[q][t];
            p.dep[4 * q + t] = d4[q][t];
        }
        p.itv[4 * q + 0] = d4[q][1] - d4[q][0];
        p.itv[4 * q + 1] = d4[q][2] - d4[q][1];
        p.itv[4 * q + 2] = d4[q][3] - d4[q][2];
        p.itv[4 * q + 3] = (p.s0[q] + 4 >= S) ? 1e10f : dn - d4[q][3];
#pragma unroll
        for (int t = 0; t < 4; ++t) p.sd[4 * q + t] = p.v[q] ? p.sig[4 * q + t] * (p.itv[4 * q + t] * len) : 0.f;
#pragma unroll
        for (int t = 0; t < 12; ++t) p.col[4 * q + t / 3][t % 3] = c4[q][t / 4][t % 4];
    }
}

// sum of sigma * delta in front of each of the lane's samples (the exponent of the transmittance, nerf.py:466)
template <int Q>
__device__ __forceinline__ void span_prefix(const Span<Q>& p, float (&ex)[4 * Q]) {
    float carry = 0.f;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const float e1 = p.sd[4 * q], e2 = e1 + p.sd[4 * q + 1], e3 = e2 + p.sd[4 * q + 2], tot = e3 + p.sd[4 * q + 3];
        const float base = carry + row16_excl_scan_up(tot);
        ex[4 * q] = base; ex[4 * q + 1] = base + e1; ex[4 * q + 2] = base + e2; ex[4 * q + 3] = base + e3;
        if (q + 1 < Q) carry = __shfl(base + tot, 15, 16);
    }
}

// transmittance in front of each sample, the sample's own attenuation and its weight (nerf.py:465-468) -- ONE definition for the
// forward, the backward (which recomputes them) and the training kernel (which keeps them), so that the three agree bit for bit
// (`#pragma clang fp contract(off)` + explicit fmaf in the shared functions: under -ffp-contract=fast LLVM chooses where to fuse per
// kernel -- T * (1 - E) became fma(-E, T, T) in the forward-only kernel and stayed a subtract and a multiply where T * E is also needed,
// a 1-ulp difference in prob between kernels that must agree.  Here every rounding is written down.)
__device__ __forceinline__ float ray_length(float rx, float ry, float rz) {
#pragma clang fp contract(off)
    return sqrtf(fmaf(rz, rz, fmaf(ry, ry, rx * rx)));
}
template <int Q, bool FAST>
__device__ __forceinline__ void span_weights(const Span<Q>& p, const float (&ex)[4 * Q], float (&T)[4 * Q], float (&E)[4 * Q], float (&w)[4 * Q]) {
#pragma clang fp contract(off)
#pragma unroll
    for (int i = 0; i < 4 * Q; ++i) {
        T[i] = exp_neg<FAST>(ex[i]);
        E[i] = exp_neg<FAST>(p.sd[i]);
        w[i] = p.v[i / 4] ? T[i] * (1.f - E[i]) : 0.f;
    }
}

struct RaySums { float a0, a1, a2, ad, ao; };

// weighted sums of a ray (every lane of the group returns the totals; lane 0's summation order is the one that is stored) + prob
template <int Q, bool NTS>
__device__ __forceinline__ RaySums span_forward(const Span<Q>& p, const float (&w)[4 * Q], bool live, float* __restrict__ prob_ray) {
#pragma clang fp contract(off)
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, ad = 0.f, ao = 0.f;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        f32x4 w4;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int i = 4 * q + t;
            w4[t] = w[i];
            a0 = fmaf(w[i], p.col[i][0], a0);
            a1 = fmaf(w[i], p.col[i][1], a1);
            a2 = fmaf(w[i], p.col[i][2], a2);
            ad = fmaf(w[i], p.dep[i], ad);
            ao += w[i];
        }
        if (p.v[q] && live && prob_ray) st4<NTS>(prob_ray + p.s0[q], w4);
    }
    return RaySums{row16_sum(a0), row16_sum(a1), row16_sum(a2), row16_sum(ad), row16_sum(ao)};
}

// closed-form backward of a ray: writes d sigma and d colour of the lane's samples, returns the group's d |ray|
template <int Q, bool NTS, bool XP>
__device__ __forceinline__ float span_backward(const Span<Q>& p, const float (&T)[4 * Q], const float (&E)[4 * Q], const float (&w)[4 * Q],
                                               float gr0, float gr1, float gr2, float gd, float go, const f32x4 (&gp)[Q], float len, bool live, int S,
                                               int gl, float* stage, float* __restrict__ d_rgb_ray, float* __restrict__ d_sigma_ray) {
#pragma clang fp contract(off)
    float TE[4 * Q], g[4 * Q], gw[4 * Q];
#pragma unroll
    for (int i = 0; i < 4 * Q; ++i) {
        TE[i] = T[i] * E[i];
        const float dot = fmaf(gd, p.dep[i], fmaf(gr2, p.col[i][2], fmaf(gr1, p.col[i][1], gr0 * p.col[i][0])));
        g[i] = p.v[i / 4] ? (dot + go) + gp[i / 4][i % 4] : 0.f;
        gw[i] = g[i] * w[i];
    }
    // sum_{j>i} g_j w_j = later samples of the quad + later lanes of its row + later rows of quads
    float aft[4 * Q], suffix = 0.f;
#pragma unroll
    for (int q = Q - 1; q >= 0; --q) {
        const float b2 = gw[4 * q + 3], b1 = b2 + gw[4 * q + 2], b0 = b1 + gw[4 * q + 1], btot = b0 + gw[4 * q];
        const float after = suffix + row16_excl_scan_down(btot);
        aft[4 * q] = after + b0; aft[4 * q + 1] = after + b1; aft[4 * q + 2] = after + b2; aft[4 * q + 3] = after;
        if (q > 0) suffix = __shfl(after + btot, 0, 16);
    }
    float dlen = 0.f;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        f32x4 ds4, dc4[3];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int i = 4 * q + t;
            const float ds = fmaf(g[i], TE[i], -aft[i]);              // dL/d(sigma*delta)
            ds4[t] = ds * (p.itv[i] * len);
            if (p.v[q]) dlen = fmaf(ds, p.sig[i] * p.itv[i], dlen);
        }
#pragma unroll
        for (int t = 0; t < 12; ++t) dc4[t / 4][t % 4] = w[4 * q + t / 3] * (t % 3 == 0 ? gr0 : t % 3 == 1 ? gr1 : gr2);
        if (p.v[q] && live) st4<NTS>(d_sigma_ray + p.s0[q], ds4);
        if (XP) {
            // the colour gradients leave the way the colours came: through the ray's staging rows (every read of the inbound direction
            // is complete: the wave's LDS accesses execute in order), as whole cache lines per store
#pragma unroll
            for (int j = 0; j < 3; ++j) *reinterpret_cast<f32x4*>(stage + 192 * q + 12 * gl + 4 * j) = dc4[j];
        } else if (p.v[q] && live) {
#pragma unroll
            for (int j = 0; j < 3; ++j) st4<NTS>(d_rgb_ray + 3 * p.s0[q] + 4 * j, dc4[j]);
        }
    }
    if (XP) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < Q; ++q)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int e = 192 * q + 4 * (16 * j + gl);
                if (e < 3 * S && live) st4<NTS>(d_rgb_ray + e, *reinterpret_cast<const f32x4*>(stage + e));
            }
    }
    return row16_sum(dlen);
}

template <int Q, int WAVES, bool NT = false, bool FAST = false, bool NTS = NT, bool XP = false>
__global__ __launch_bounds__(64 * WAVES) void composite_fwd_span_kernel(const float* __restrict__ ray, const float* __restrict__ rgb_s,
                                                                        const float* __restrict__ sigma_s, const float* __restrict__ depth_s,
                                                                        long long n_rays, int S, int has_bg, float bg,
                                                                        float* __restrict__ rgb, float* __restrict__ depth,
                                                                        float* __restrict__ opacity, float* __restrict__ prob) {
    constexpr int G = 16, RPW = 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, gl = threadIdx.x & 15;
    const long long r_wave = ((long long)blockIdx.x * WAVES + wave) * RPW;
    if (r_wave >= n_rays) return;                               // wave-uniform: no ray left for this wave
    const long long r_raw = r_wave + lane / G;
    const bool live = r_raw < n_rays;
    const long long r = live ? r_raw : n_rays - 1;              // idle groups shadow the last ray (no stores)
    const float rx = ray[r * 3], ry = ray[r * 3 + 1], rz = ray[r * 3 + 2];
    const float len = ray_length(rx, ry, rz);
    __shared__ float stage[XP ? WAVES * RPW * Q * 192 : 1];
    Span<Q> p;
    load_span<Q, NT, XP>(p, sigma_s + r * S, depth_s + r * S, rgb_s + r * S * 3, S, gl, len, stage + (XP ? (wave * RPW + lane / G) * Q * 192 : 0));
    float ex[4 * Q], T[4 * Q], E[4 * Q], w[4 * Q];
    span_prefix<Q>(p, ex);
    span_weights<Q, FAST>(p, ex, T, E, w);
    RaySums a = span_forward<Q, NTS>(p, w, live, prob ? prob + r * S : nullptr);
    if (gl == 0 && live) {
        if (has_bg) { const float t = bg * (1.f - a.ao); a.a0 += t; a.a1 += t; a.a2 += t; }
        rgb[r * 3] = a.a0; rgb[r * 3 + 1] = a.a1; rgb[r * 3 + 2] = a.a2;
        depth[r] = a.ad;
        opacity[r] = a.ao;
    }
}

template <int Q, int WAVES, bool NT = false, bool FAST = false, bool NTS = NT, bool XP = false>
__global__ __launch_bounds__(64 * WAVES) void composite_bwd_span_kernel(const float* __restrict__ ray, const float* __restrict__ rgb_s,
                                                                        const float* __restrict__ sigma_s, const float* __restrict__ depth_s,
                                                                        long long n_rays, int S, int has_bg, float bg,
                                                                        const float* __restrict__ g_rgb, const float* __restrict__ g_depth,
                                                                        const float* __restrict__ g_opacity, const float* __restrict__ g_prob,
                                                                        float* __restrict__ d_rgb_s, float* __restrict__ d_sigma_s,
                                                                        float* __restrict__ d_ray) {
    constexpr int G = 16, RPW = 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, gl = threadIdx.x & 15;
    const long long r_wave = ((long long)blockIdx.x * WAVES + wave) * RPW;
    if (r_wave >= n_rays) return;
    const long long r_raw = r_wave + lane / G;
    const bool live = r_raw < n_rays;
    const long long r = live ? r_raw : n_rays - 1;
    const float rx = ray[r * 3], ry = ray[r * 3 + 1], rz = ray[r * 3 + 2];
    const float len = ray_length(rx, ry, rz);
    const float gr0 = g_rgb ? g_rgb[r * 3] : 0.f, gr1 = g_rgb ? g_rgb[r * 3 + 1] : 0.f, gr2 = g_rgb ? g_rgb[r * 3 + 2] : 0.f;
    const float gd = g_depth ? g_depth[r] : 0.f;
    float go = g_opacity ? g_opacity[r] : 0.f;
    if (has_bg) go -= bg * (gr0 + gr1 + gr2);
    __shared__ float stage_all[XP ? WAVES * RPW * Q * 192 : 1];
    float* stage = stage_all + (XP ? (wave * RPW + lane / G) * Q * 192 : 0);
    Span<Q> p;
    load_span<Q, NT, XP>(p, sigma_s + r * S, depth_s + r * S, rgb_s + r * S * 3, S, gl, len, stage);
    f32x4 gp[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) gp[q] = (p.v[q] && g_prob) ? ld4<NT>(g_prob + r * S + p.s0[q]) : f32x4{0.f, 0.f, 0.f, 0.f};
    float ex[4 * Q], T[4 * Q], E[4 * Q], w[4 * Q];
    span_prefix<Q>(p, ex);
    span_weights<Q, FAST>(p, ex, T, E, w);
    const float dlen = span_backward<Q, NTS, XP>(p, T, E, w, gr0, gr1, gr2, gd, go, gp, len, live, S, gl, stage, d_rgb_s + r * S * 3, d_sigma_s + r * S);
    if (gl == 0 && live) {
        const float inv = len > 0.f ? dlen / len : 0.f;
        d_ray[r * 3] = inv * rx; d_ray[r * 3 + 1] = inv * ry; d_ray[r * 3 + 2] = inv * rz;
    }
}

// The training form (niw_composite_mse_train): compositing, the photometric residual of the ray against its pixel, and the backward of
// both, in ONE pass over the samples -- the three launches composite_fwd -> mse -> composite_bwd of a train iteration are 5-7 us each
// around no work at a rank's share of the batch, and the backward re-read what the forward had in registers.  The residual is local
// to the ray (the mean's normaliser is a constant of the batch: d rgb = 2 (rgb - pixel) / n * weight), so nothing waits for the loss
// VALUE: every ray leaves its three residuals in `resid`, and niw_mse_from_residuals / the train step's closing kernel sums their
// squares in mse_kernel's order (bit-identical loss).  Same device functions as the two kernels above: same rgb, same gradients.
struct TrainPixels {
    const float* image;           // [B][3][hw]
    const int64_t* ray_idx;       // [R] pixel of ray r of every view, or NULL (r itself)
    long long R, hw, first_ray;   // rays per view; pixels per image; the flattened-[B][R] number of ray 0 of this launch
    double slope;                 // grad_scale * 2 / n_norm, formed like mse_kernel forms it
};

template <int Q, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void composite_train_span_kernel(const float* __restrict__ ray, const float* __restrict__ rgb_s,
                                                                          const float* __restrict__ sigma_s, const float* __restrict__ depth_s,
                                                                          long long n_rays, int S, TrainPixels px,
                                                                          float* __restrict__ rgb, float* __restrict__ depth, float* __restrict__ opacity,
                                                                          float* __restrict__ prob, float* __restrict__ resid, float* __restrict__ d_rgb,
                                                                          float* __restrict__ d_rgb_s, float* __restrict__ d_sigma_s, float* __restrict__ d_ray) {
    constexpr int G = 16, RPW = 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, gl = threadIdx.x & 15;
    const long long r_wave = ((long long)blockIdx.x * WAVES + wave) * RPW;
    if (r_wave >= n_rays) return;
    const long long r_raw = r_wave + lane / G;
    const bool live = r_raw < n_rays;
    const long long r = live ? r_raw : n_rays - 1;
    // the ray's pixel: lanes 0..2 of the group fetch one colour channel each (two dependent cold reads, issued before everything else)
    float pixel = 0.f;
    if (gl < 3) {
        const long long br = px.first_ray + r, b = br / px.R, rr = br - b * px.R;
        const long long pix = px.ray_idx ? px.ray_idx[rr] : rr;
        pixel = px.image[(b * 3 + gl) * px.hw + pix];
    }
    const float rx = ray[r * 3], ry = ray[r * 3 + 1], rz = ray[r * 3 + 2];
    const float len = ray_length(rx, ry, rz);
    __shared__ float stage_all[WAVES * RPW * Q * 192];
    float* stage = stage_all + (wave * RPW + lane / G) * Q * 192;
    Span<Q> p;
    load_span<Q, false, true>(p, sigma_s + r * S, depth_s + r * S, rgb_s + r * S * 3, S, gl, len, stage);
    float ex[4 * Q], T[4 * Q], E[4 * Q], w[4 * Q];
    span_prefix<Q>(p, ex);
    span_weights<Q, false>(p, ex, T, E, w);
    const RaySums a = span_forward<Q, false>(p, w, live, prob ? prob + r * S : nullptr);
    // lane 0 holds the stored colour; the residuals and the colour's gradient are formed there exactly as mse_kernel forms them
    const float p1 = dpp_zero_fill<kRowShl | 1>(pixel), p2 = dpp_zero_fill<kRowShl | 2>(pixel);
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
    if (gl == 0) {
        const float e0 = a.a0 - pixel, e1 = a.a1 - p1, e2 = a.a2 - p2;
        g0 = (float)(px.slope * (double)e0); g1 = (float)(px.slope * (double)e1); g2 = (float)(px.slope * (double)e2);
        if (live) {
            rgb[r * 3] = a.a0; rgb[r * 3 + 1] = a.a1; rgb[r * 3 + 2] = a.a2;
            depth[r] = a.ad;
            opacity[r] = a.ao;
            resid[r * 3] = e0; resid[r * 3 + 1] = e1; resid[r * 3 + 2] = e2;
            if (d_rgb) { d_rgb[r * 3] = g0; d_rgb[r * 3 + 1] = g1; d_rgb[r * 3 + 2] = g2; }
        }
    }
    const float gr0 = __shfl(g0, 0, G), gr1 = __shfl(g1, 0, G), gr2 = __shfl(g2, 0, G);
    f32x4 gp[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) gp[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float dlen = span_backward<Q, false, true>(p, T, E, w, gr0, gr1, gr2, 0.f, 0.f, gp, len, live, S, gl, stage, d_rgb_s + r * S * 3, d_sigma_s + r * S);
    if (gl == 0 && live) {
        const float inv = len > 0.f ? dlen / len : 0.f;
        d_ray[r * 3] = inv * rx; d_ray[r * 3 + 1] = inv * ry; d_ray[r * 3 + 2] = inv * rz;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// scalar fallback: one 64-lane wave per ray, one sample per lane, 64-sample chunks (any S >= 2, any alignment)
// ------------------------------------------------------------------------------------------------------------------
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
__device__ __forceinline__ float interval(const float* __restrict__ d, int s, int S) {
    return s == S - 1 ? 1e10f : d[s + 1] - d[s];
}

__global__ __launch_bounds__(256) void composite_fwd_scalar_kernel(const float* __restrict__ ray, const float* __restrict__ rgb_s,
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

__global__ __launch_bounds__(256) void composite_bwd_scalar_kernel(const float* __restrict__ ray, const float* __restrict__ rgb_s,
                                                                   const float* __restrict__ sigma_s, const float* __restrict__ depth_s,
                                                                   long long n_rays, int S, int has_bg, float bg,
                                                                   const float* __restrict__ g_rgb, const float* __restrict__ g_depth,
                                                                   const float* __restrict__ g_opacity, const float* __restrict__ g_prob,
                                                                   float* __restrict__ d_rgb_s, float* __restrict__ d_sigma_s,
                                                                   float* __restrict__ d_ray) {
    __shared__ float chunk_prefix[4][4 * kMaxChunks];
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
    const int n_chunks = (S + 63) / 64;
    float carry = 0.f;
    for (int k = 0; k < n_chunks; ++k) {
        const int s = k * 64 + lane;
        float sd = s < S ? sg[s] * (interval(d, s, S) * len) : 0.f;
        if (lane == 0) chunk_prefix[wv][k] = carry;
        carry += wave_sum(sd);
    }
    __builtin_amdgcn_wave_barrier();
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
        float shd = __shfl_down(gw, 1);
        if (lane == 63) shd = 0.f;
        const float after = suffix + wave_suffix_incl_scan(shd, lane);
        const float ds = g * (T * e) - after;
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

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline int group_lanes(int S) {      // lanes per ray: 16 up to 256 samples (one to four quads per lane), 64 beyond (chunks of 256)
    return S <= 256 ? 16 : 64;
}
// From 2 M samples per launch (a full image: 7.7-23 M) the accesses carry the non-temporal hint: the operands are touched once and
// exceed the L2s (32 MiB) -- the training launches (<= 0.8 M samples, produced and consumed by the neighbouring kernels) keep the default.
inline bool streaming_size(long long n_rays, int S) { return n_rays * S >= (2ll << 20); }
// S = 1, as the reference behaves (nerf.py:461-462): the closing 1e10 interval is built from an EMPTY slice of the (empty) interval
// tensor, so the single sample gets no interval at all -- every weight tensor is empty and rgb / depth / opacity are sums over nothing.
__global__ void composite_single_sample_fwd_kernel(long long n_rays, int has_bg, float bg, float* __restrict__ rgb, float* __restrict__ depth,
                                                   float* __restrict__ opacity) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rays) return;
    const float c = has_bg ? bg : 0.f;                       // rgb + bg * (1 - opacity) with opacity = 0
    rgb[3 * r] = c; rgb[3 * r + 1] = c; rgb[3 * r + 2] = c;
    depth[r] = 0.f;
    opacity[r] = 0.f;
}
__global__ void composite_single_sample_bwd_kernel(long long n_rays, float* __restrict__ d_rgb_s, float* __restrict__ d_sigma_s, float* __restrict__ d_ray) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rays) return;
#pragma unroll
    for (int c = 0; c < 3; ++c) d_rgb_s[3 * r + c] = d_ray[3 * r + c] = 0.f;
    d_sigma_s[r] = 0.f;
}
}  // namespace

extern "C" int niw_composite_fwd(const float* ray, const float* rgb_s, const float* sigma_s, const float* depth_s,
                                 int64_t n_rays, int n_samples, int has_bg, float bg,
                                 float* rgb, float* depth, float* opacity, float* prob, niw_stream_t stream) {
    NIW_REQUIRE(ray && rgb_s && sigma_s && depth_s && rgb && depth && opacity, "niw_composite_fwd: null pointer");
    NIW_REQUIRE(n_rays > 0 && n_samples > 0, "niw_composite_fwd: empty input (n_rays=%lld, S=%d)", (long long)n_rays, n_samples);
    hipStream_t st = (hipStream_t)stream;
    const int S = n_samples;
    if (S == 1) {      // the reference's degenerate case: zero outputs, EMPTY prob (`prob`, if given, is not written: it has no elements there)
        composite_single_sample_fwd_kernel<<<(int)((n_rays + 255) / 256), 256, 0, st>>>(n_rays, has_bg, bg, rgb, depth, opacity);
        NIW_LAUNCH_CHECK("niw_composite_fwd (S = 1)");
        return NIW_OK;
    }
    if (S % 4 == 0 && aligned16(rgb_s) && aligned16(sigma_s) && aligned16(depth_s) && (!prob || aligned16(prob))) {
        const int G = group_lanes(S);
        const int blocks = (int)((n_rays + 4 * (64 / G) - 1) / (4 * (64 / G)));
        const bool nt = streaming_size(n_rays, S);
#define NIW_CFWD(GG) composite_fwd_kernel<GG><<<blocks, 256, 0, st>>>(ray, rgb_s, sigma_s, depth_s, n_rays, S, has_bg, bg, rgb, depth, opacity, prob)
        // span kernels: colours through LDS; streaming sizes read non-temporally (prob keeps the default policy: the resampling kernel reads it next)
#define NIW_CFWD_SPAN(QQ)                                                                                                                              \
    do {                                                                                                                                               \
        if (nt) composite_fwd_span_kernel<QQ, 4, true, false, false, true><<<blocks, 256, 0, st>>>(ray, rgb_s, sigma_s, depth_s, n_rays, S, has_bg, bg, rgb, depth, opacity, prob); \
        else composite_fwd_span_kernel<QQ, 4, false, false, false, true><<<blocks, 256, 0, st>>>(ray, rgb_s, sigma_s, depth_s, n_rays, S, has_bg, bg, rgb, depth, opacity, prob);  \
    } while (0)
        if (S <= 64) NIW_CFWD_SPAN(1);
        else if (S <= 128) NIW_CFWD_SPAN(2);
        else if (S <= 192) NIW_CFWD_SPAN(3);
        else if (S <= 256) NIW_CFWD_SPAN(4);
        else NIW_CFWD(64);
#undef NIW_CFWD
#undef NIW_CFWD_SPAN
    } else {
        const int blocks = (int)((n_rays + 3) / 4);
        composite_fwd_scalar_kernel<<<blocks, 256, 0, st>>>(ray, rgb_s, sigma_s, depth_s, n_rays, S, has_bg, bg, rgb, depth, opacity, prob);
    }
    NIW_LAUNCH_CHECK("niw_composite_fwd");
    return NIW_OK;
}

extern "C" int niw_composite_bwd(const float* ray, const float* rgb_s, const float* sigma_s, const float* depth_s,
                                 int64_t n_rays, int n_samples, int has_bg, float bg,
                                 const float* d_rgb, const float* d_depth, const float* d_opacity, const float* d_prob,
                                 float* d_rgb_s, float* d_sigma_s, float* d_ray, niw_stream_t stream) {
    NIW_REQUIRE(ray && rgb_s && sigma_s && depth_s && d_rgb_s && d_sigma_s && d_ray, "niw_composite_bwd: null pointer");
    NIW_REQUIRE(n_rays > 0 && n_samples > 0, "niw_composite_bwd: empty input (n_rays=%lld, S=%d)", (long long)n_rays, n_samples);
    if (n_samples == 1) {      // nothing depends on the inputs (see niw_composite_fwd)
        composite_single_sample_bwd_kernel<<<(int)((n_rays + 255) / 256), 256, 0, (hipStream_t)stream>>>(n_rays, d_rgb_s, d_sigma_s, d_ray);
        NIW_LAUNCH_CHECK("niw_composite_bwd (S = 1)");
        return NIW_OK;
    }
    NIW_REQUIRE(n_samples <= 256 * kMaxChunks, "niw_composite_bwd: S=%d exceeds the %d samples per ray the chunk-prefix table holds",
                n_samples, 256 * kMaxChunks);
    hipStream_t st = (hipStream_t)stream;
    const int S = n_samples;
    if (S % 4 == 0 && aligned16(rgb_s) && aligned16(sigma_s) && aligned16(depth_s) && aligned16(d_rgb_s) && aligned16(d_sigma_s) &&
        (!d_prob || aligned16(d_prob))) {
        const int G = group_lanes(S);
        const int blocks = (int)((n_rays + 4 * (64 / G) - 1) / (4 * (64 / G)));
        const bool nt = streaming_size(n_rays, S);
#define NIW_CBWD(GG) composite_bwd_kernel<GG><<<blocks, 256, 0, st>>>(ray, rgb_s, sigma_s, depth_s, n_rays, S, has_bg, bg, d_rgb, d_depth, d_opacity, d_prob, d_rgb_s, d_sigma_s, d_ray)
#define NIW_CBWD_SPAN(QQ)                                                                                                                              \
    do {                                                                                                                                               \
        if (nt) composite_bwd_span_kernel<QQ, 4, true, false, true, true><<<blocks, 256, 0, st>>>(ray, rgb_s, sigma_s, depth_s, n_rays, S, has_bg, bg, d_rgb, d_depth, d_opacity, d_prob, d_rgb_s, d_sigma_s, d_ray); \
        else composite_bwd_span_kernel<QQ, 4, false, false, false, true><<<blocks, 256, 0, st>>>(ray, rgb_s, sigma_s, depth_s, n_rays, S, has_bg, bg, d_rgb, d_depth, d_opacity, d_prob, d_rgb_s, d_sigma_s, d_ray); \
    } while (0)
        if (S <= 64) NIW_CBWD_SPAN(1);
        else if (S <= 128) NIW_CBWD_SPAN(2);
        else if (S <= 192) NIW_CBWD_SPAN(3);
        else if (S <= 256) NIW_CBWD_SPAN(4);
        else NIW_CBWD(64);
#undef NIW_CBWD
#undef NIW_CBWD_SPAN
    } else {
        const int blocks = (int)((n_rays + 3) / 4);
        composite_bwd_scalar_kernel<<<blocks, 256, 0, st>>>(ray, rgb_s, sigma_s, depth_s, n_rays, S, has_bg, bg, d_rgb, d_depth, d_opacity,
                                                            d_prob, d_rgb_s, d_sigma_s, d_ray);
    }
    NIW_LAUNCH_CHECK("niw_composite_bwd");
    return NIW_OK;
}

// Training form: niw_composite_fwd + niw_mse_fwd_bwd + niw_composite_bwd of one ray batch as ONE launch (composite_train_span_kernel).
// NIW_ERR_UNSUPPORTED (nothing launched) for shapes outside the span kernels: the caller then makes the three calls.
extern "C" int niw_composite_mse_train(const float* ray, const float* rgb_s, const float* sigma_s, const float* depth_s, int64_t n_rays, int n_samples,
                                       const float* image, const int64_t* ray_idx, int n_views, int64_t n_rays_per_view, int64_t hw, int64_t first_ray,
                                       double n_norm, float grad_scale, float* rgb, float* depth, float* opacity, float* prob, float* resid, float* d_rgb,
                                       float* d_rgb_s, float* d_sigma_s, float* d_ray, niw_stream_t stream) {
    NIW_REQUIRE(ray && rgb_s && sigma_s && depth_s && image && rgb && depth && opacity && resid && d_rgb_s && d_sigma_s && d_ray, "niw_composite_mse_train: null pointer");
    NIW_REQUIRE(n_rays > 0 && n_samples > 0, "niw_composite_mse_train: empty input (n_rays=%lld, S=%d)", (long long)n_rays, n_samples);
    NIW_REQUIRE(n_views > 0 && n_rays_per_view > 0 && hw > 0 && n_norm > 0, "niw_composite_mse_train: empty batch");
    NIW_REQUIRE(first_ray >= 0 && first_ray + n_rays <= (int64_t)n_views * n_rays_per_view, "niw_composite_mse_train: rays [%lld, %lld) leave the %d x %lld batch",
                (long long)first_ray, (long long)(first_ray + n_rays), n_views, (long long)n_rays_per_view);
    const int S = n_samples;
    if (S < 2 || S > 256 || S % 4 != 0 || !aligned16(rgb_s) || !aligned16(sigma_s) || !aligned16(depth_s) || !aligned16(d_rgb_s) || !aligned16(d_sigma_s) ||
        (prob && !aligned16(prob))) {
        niw_set_error("niw_composite_mse_train: S=%d or the operands' alignment is outside the one-launch form (S %% 4 == 0, S <= 256, 16-byte rows)", S);
        return NIW_ERR_UNSUPPORTED;
    }
    TrainPixels px{image, ray_idx, (long long)n_rays_per_view, (long long)hw, (long long)first_ray, (double)grad_scale * 2.0 / n_norm};
    const int blocks = (int)((n_rays + 15) / 16);
    hipStream_t st = (hipStream_t)stream;
#define NIW_CTRAIN(QQ) composite_train_span_kernel<QQ, 4><<<blocks, 256, 0, st>>>(ray, rgb_s, sigma_s, depth_s, n_rays, S, px, rgb, depth, opacity, prob, resid, d_rgb, d_rgb_s, d_sigma_s, d_ray)
    if (S <= 64) NIW_CTRAIN(1);
    else if (S <= 128) NIW_CTRAIN(2);
    else if (S <= 192) NIW_CTRAIN(3);
    else NIW_CTRAIN(4);
#undef NIW_CTRAIN
    NIW_LAUNCH_CHECK("niw_composite_mse_train");
    return NIW_OK;
}
