// Diagnostic (not part of libniw_hip.so): the compositing kernels of csrc/niw_composite.hip launched directly in their template
// variants -- waves per workgroup, non-temporal accesses, fast exponential, the 64-lane form of rounds 2-4 -- next to plain streaming
// kernels of the same footprint (read-only, copy, the forward's 5 : 1 and the backward's 5 : 4 read : write mix), all timed with device
// events on one stream.  Prints one JSON line per variant.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Ineural_invertible_warp_amd/csrc tools/composite_variants.hip -o scratch/composite_variants
//   scratch/composite_variants [n_rays=120000] [S=192] [iters=50]
#include "../neural_invertible_warp_amd/csrc/niw_composite.hip"

#include <cstdarg>
#include <cstdlib>
#include <random>
#include <vector>

void niw_set_error(const char*, ...) {}

#define CK(x)                                                                            \
    do {                                                                                 \
        hipError_t e_ = (x);                                                             \
        if (e_ != hipSuccess) {                                                          \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));    \
            exit(1);                                                                     \
        }                                                                                \
    } while (0)

namespace {
// streaming kernels: every lane moves 16 bytes per access, a workgroup of 256 walks `per_block` float4s
__global__ __launch_bounds__(256) void stream_read_kernel(const f32x4* __restrict__ a, long long n4, float* __restrict__ out) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) acc += a[i];
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = 1.f;      // (never: keeps the loads)
}
// reads R float4 per thread and trip, writes W of them
template <int R, int W, bool NTL = false, bool NTS = false>
__global__ __launch_bounds__(256) void stream_mix_kernel(const f32x4* __restrict__ a, f32x4* __restrict__ b, long long trips) {
    for (long long t = (long long)blockIdx.x; t < trips; t += gridDim.x) {
        f32x4 v[R];
#pragma unroll
        for (int j = 0; j < R; ++j) v[j] = NTL ? __builtin_nontemporal_load(a + (t * R + j) * 256 + threadIdx.x) : a[(t * R + j) * 256 + threadIdx.x];
        f32x4 s = v[0];
#pragma unroll
        for (int j = 1; j < R; ++j) s += v[j];
#pragma unroll
        for (int j = 0; j < W; ++j) {
            if (NTS) __builtin_nontemporal_store(s + v[j % R], b + (t * W + j) * 256 + threadIdx.x);
            else b[(t * W + j) * 256 + threadIdx.x] = s + v[j % R];
        }
    }
}

struct Timer {
    hipEvent_t a, b;
    Timer() { CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); }
    template <class F>
    float us(F&& f, int iters) {
        f(); f();
        CK(hipEventRecord(a, 0));
        for (int i = 0; i < iters; ++i) f();
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, a, b));
        return ms * 1e3f / iters;
    }
};

double checksum(const float* d, size_t n) {
    std::vector<float> h(n);
    CK(hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost));
    double s = 0;
    for (size_t i = 0; i < n; ++i) s += (double)h[i] * (double)((i % 97) + 1);
    return s;
}
}  // namespace

int main(int argc, char** argv) {
    const long long N = argc > 1 ? atoll(argv[1]) : 120000;
    const int S = argc > 2 ? atoi(argv[2]) : 192;
    const int iters = argc > 3 ? atoi(argv[3]) : 50;
    const size_t NS = (size_t)N * S;
    std::mt19937 gen(7);
    std::uniform_real_distribution<float> U(0.f, 1.f);
    std::vector<float> h_ray(N * 3), h_rgb(NS * 3), h_sig(NS), h_dep(NS), h_g(N * 3);
    for (auto& x : h_ray) x = U(gen) * 2 - 1;
    for (auto& x : h_rgb) x = U(gen);
    for (auto& x : h_sig) x = U(gen) * 2;
    for (long long r = 0; r < N; ++r)
        for (int s = 0; s < S; ++s) h_dep[r * S + s] = 1.f + (s + 0.9f * U(gen)) / S;
    for (auto& x : h_g) x = U(gen) * 2 - 1;
    float *ray, *rgb_s, *sig, *dep, *g_rgb, *rgb, *depth, *opa, *prob, *d_rgb_s, *d_sig, *d_ray;
    CK(hipMalloc(&ray, N * 12)); CK(hipMalloc(&rgb_s, NS * 12)); CK(hipMalloc(&sig, NS * 4)); CK(hipMalloc(&dep, NS * 4));
    CK(hipMalloc(&g_rgb, N * 12)); CK(hipMalloc(&rgb, N * 12)); CK(hipMalloc(&depth, N * 4)); CK(hipMalloc(&opa, N * 4));
    CK(hipMalloc(&prob, NS * 4)); CK(hipMalloc(&d_rgb_s, NS * 12)); CK(hipMalloc(&d_sig, NS * 4)); CK(hipMalloc(&d_ray, N * 12));
    CK(hipMemcpy(ray, h_ray.data(), N * 12, hipMemcpyHostToDevice)); CK(hipMemcpy(rgb_s, h_rgb.data(), NS * 12, hipMemcpyHostToDevice));
    CK(hipMemcpy(sig, h_sig.data(), NS * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dep, h_dep.data(), NS * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(g_rgb, h_g.data(), N * 12, hipMemcpyHostToDevice));
    Timer T;
    const double bytes_f = (double)NS * 24 + N * 32, bytes_b = (double)NS * 36 + N * 36;
    auto report = [&](const char* name, float us, double bytes, double chk) {
        printf("{\"variant\": \"%s\", \"n_rays\": %lld, \"samples\": %d, \"us\": %.1f, \"gbps\": %.0f, \"frac_of_6290\": %.3f, \"checksum\": %.9g}\n", name, N, S, us,
               bytes / us / 1e3, bytes / us / 1e3 / 6290, chk);
        fflush(stdout);
    };
    // ---- streaming references over the same footprint (buffers of their own: A holds the forward's bytes, Bw the backward's stores)
    {
        f32x4 *A, *Bw;
        const size_t a_bytes = NS * 24 + (1 << 20), b_bytes = NS * 16 + (1 << 20);
        CK(hipMalloc(&A, a_bytes)); CK(hipMalloc(&Bw, b_bytes));
        CK(hipMemset(A, 0, a_bytes)); CK(hipMemset(Bw, 0, b_bytes));
        const long long n4 = (long long)(NS * 20 / 16);                  // the forward's read bytes
        for (int blocks : {2048, 8192}) {
            char nm[64];
            snprintf(nm, sizeof nm, "stream_read_%d", blocks);
            report(nm, T.us([&] { stream_read_kernel<<<blocks, 256>>>(A, n4, opa); }, iters), n4 * 16.0, 0);
        }
        const long long trips5 = (long long)NS * 20 / (5 * 256 * 16);    // 20 B in per sample
        report("stream_mix_5r_1w", T.us([&] { stream_mix_kernel<5, 1><<<4096, 256>>>(A, Bw, trips5); }, iters), trips5 * 6 * 4096.0, 0);
        report("stream_mix_5r_1w_nts", T.us([&] { stream_mix_kernel<5, 1, false, true><<<4096, 256>>>(A, Bw, trips5); }, iters), trips5 * 6 * 4096.0, 0);
        report("stream_mix_5r_1w_ntl", T.us([&] { stream_mix_kernel<5, 1, true, false><<<4096, 256>>>(A, Bw, trips5); }, iters), trips5 * 6 * 4096.0, 0);
        report("stream_mix_5r_1w_b1024", T.us([&] { stream_mix_kernel<5, 1><<<1024, 256>>>(A, Bw, trips5); }, iters), trips5 * 6 * 4096.0, 0);
        report("stream_mix_5r_1w_b16384", T.us([&] { stream_mix_kernel<5, 1><<<16384, 256>>>(A, Bw, trips5); }, iters), trips5 * 6 * 4096.0, 0);
        report("stream_mix_5r_4w_nts", T.us([&] { stream_mix_kernel<5, 4, false, true><<<4096, 256>>>(A, Bw, trips5); }, iters), trips5 * 9 * 4096.0, 0);
        report("stream_copy_1r_1w_nts", T.us([&] { stream_mix_kernel<1, 1, false, true><<<4096, 256>>>(A, Bw, (long long)NS * 12 / 4096); }, iters),
               (double)((long long)NS * 12 / 4096) * 2 * 4096.0, 0);
        report("stream_mix_5r_4w", T.us([&] { stream_mix_kernel<5, 4><<<4096, 256>>>(A, Bw, trips5); }, iters), trips5 * 9 * 4096.0, 0);
        report("stream_copy_1r_1w", T.us([&] { stream_mix_kernel<1, 1><<<4096, 256>>>(A, Bw, (long long)NS * 12 / 4096); }, iters),
               (double)((long long)NS * 12 / 4096) * 2 * 4096.0, 0);
        // reads: n4 * 16 = NS * 20 <= a_bytes; 5r: trips5 * 5 * 4096 <= NS * 20; writes: 5r_4w trips5 * 4 * 4096 <= NS * 16 <= b_bytes; copy NS * 12 each
        CK(hipFree(A)); CK(hipFree(Bw));
    }
    if (S % 4 || S <= 64 || S > 256) {
        printf("{\"note\": \"span variants need 64 < S <= 256, S %% 4 == 0\"}\n");
        return 0;
    }
#define ARGS_F ray, rgb_s, sig, dep, N, S, 0, 0.f, rgb, depth, opa
#define ARGS_B ray, rgb_s, sig, dep, N, S, 0, 0.f, g_rgb, nullptr, nullptr, nullptr, d_rgb_s, d_sig, d_ray
#define FWD(Q, W, NT, FAST, PROB, name)                                                                                              \
    {                                                                                                                                \
        const int blocks = (int)((N + 4 * W - 1) / (4 * W));                                                                         \
        const float us = T.us([&] { composite_fwd_span_kernel<Q, W, NT, FAST><<<blocks, 64 * W>>>(ARGS_F, PROB); }, iters);           \
        report(name, us, bytes_f - (PROB ? 0 : (double)NS * 4), checksum(rgb, N * 3) + (PROB ? checksum(prob, NS) : 0));              \
    }
#define BWD(Q, W, NT, FAST, name)                                                                                                    \
    {                                                                                                                                \
        const int blocks = (int)((N + 4 * W - 1) / (4 * W));                                                                         \
        const float us = T.us([&] { composite_bwd_span_kernel<Q, W, NT, FAST><<<blocks, 64 * W>>>(ARGS_B); }, iters);                 \
        report(name, us, bytes_b, checksum(d_sig, NS) + checksum(d_ray, N * 3) + 1e-3 * checksum(d_rgb_s, NS * 3));                                                      \
    }
#define FWDX(Q, W, NTL, NTS, name)                                                                                                   \
    {                                                                                                                                \
        const int blocks = (int)((N + 4 * W - 1) / (4 * W));                                                                         \
        const float us = T.us([&] { composite_fwd_span_kernel<Q, W, NTL, false, NTS><<<blocks, 64 * W>>>(ARGS_F, prob); }, iters);    \
        report(name, us, bytes_f, checksum(rgb, N * 3) + checksum(prob, NS));                                                         \
    }
#define BWDX(Q, W, NTL, NTS, name)                                                                                                   \
    {                                                                                                                                \
        const int blocks = (int)((N + 4 * W - 1) / (4 * W));                                                                         \
        const float us = T.us([&] { composite_bwd_span_kernel<Q, W, NTL, false, NTS><<<blocks, 64 * W>>>(ARGS_B); }, iters);          \
        report(name, us, bytes_b, checksum(d_sig, NS) + checksum(d_ray, N * 3));                                                      \
    }
#define FWDP(Q, W, NTL, NTS, name)                                                                                                   \
    {                                                                                                                                \
        const int blocks = (int)((N + 4 * W - 1) / (4 * W));                                                                         \
        const float us = T.us([&] { composite_fwd_span_kernel<Q, W, NTL, false, NTS, true><<<blocks, 64 * W>>>(ARGS_F, prob); }, iters); \
        report(name, us, bytes_f, checksum(rgb, N * 3) + checksum(prob, NS));                                                         \
    }
#define BWDP(Q, W, NTL, NTS, name)                                                                                                   \
    {                                                                                                                                \
        CK(hipMemset(d_rgb_s, 0, NS * 12));                                                                                          \
        const int blocks = (int)((N + 4 * W - 1) / (4 * W));                                                                         \
        const float us = T.us([&] { composite_bwd_span_kernel<Q, W, NTL, false, NTS, true><<<blocks, 64 * W>>>(ARGS_B); }, iters);    \
        report(name, us, bytes_b, checksum(d_sig, NS) + checksum(d_ray, N * 3) + 1e-3 * checksum(d_rgb_s, NS * 3));                   \
    }
#define ALL(Q)                                                  \
    FWD(Q, 4, false, false, prob, "fwd_w4")                     \
    FWD(Q, 2, false, false, prob, "fwd_w2")                     \
    FWD(Q, 1, false, false, prob, "fwd_w1")                     \
    FWD(Q, 8, false, false, prob, "fwd_w8")                     \
    FWD(Q, 4, true, false, prob, "fwd_w4_nt")                   \
    FWDX(Q, 4, true, false, "fwd_w4_ntl")                       \
    FWDX(Q, 4, false, true, "fwd_w4_nts")                       \
    FWD(Q, 4, false, true, prob, "fwd_w4_fastexp")              \
    FWD(Q, 4, false, false, (float*)nullptr, "fwd_w4_noprob")   \
    FWDP(Q, 4, false, false, "fwd_w4_xp")                       \
    FWDP(Q, 4, true, false, "fwd_w4_xp_ntl")                    \
    FWDP(Q, 4, true, true, "fwd_w4_xp_nt")                      \
    FWDP(Q, 2, true, true, "fwd_w2_xp_nt")                      \
    FWDP(Q, 1, true, true, "fwd_w1_xp_nt")                      \
    BWDP(Q, 4, false, false, "bwd_w4_xp")                       \
    BWDP(Q, 4, true, false, "bwd_w4_xp_ntl")                    \
    BWDP(Q, 4, true, true, "bwd_w4_xp_nt")                      \
    BWDP(Q, 2, true, true, "bwd_w2_xp_nt")                      \
    BWDP(Q, 4, false, true, "bwd_w4_xp_nts")                    \
    BWD(Q, 4, false, false, "bwd_w4")                           \
    BWD(Q, 2, false, false, "bwd_w2")                           \
    BWD(Q, 1, false, false, "bwd_w1")                           \
    BWD(Q, 4, true, false, "bwd_w4_nt")                         \
    BWDX(Q, 4, true, false, "bwd_w4_ntl")                       \
    BWDX(Q, 4, false, true, "bwd_w4_nts")                       \
    BWD(Q, 4, false, true, "bwd_w4_fastexp")
    if (S <= 128) { ALL(2) } else if (S <= 192) { ALL(3) } else { ALL(4) }
    // rounds 2-4: one quad per lane, 32 / 64 lanes per ray
    {
        const int G = S <= 128 ? 32 : 64;
        const int blocks = (int)((N + 4 * (64 / G) - 1) / (4 * (64 / G)));
        float us;
        if (G == 32) us = T.us([&] { composite_fwd_kernel<32><<<blocks, 256>>>(ARGS_F, prob); }, iters);
        else us = T.us([&] { composite_fwd_kernel<64><<<blocks, 256>>>(ARGS_F, prob); }, iters);
        report("fwd_wide_groups", us, bytes_f, checksum(rgb, N * 3) + checksum(prob, NS));
        if (G == 32) us = T.us([&] { composite_bwd_kernel<32><<<blocks, 256>>>(ARGS_B); }, iters);
        else us = T.us([&] { composite_bwd_kernel<64><<<blocks, 256>>>(ARGS_B); }, iters);
        report("bwd_wide_groups", us, bytes_b, checksum(d_sig, NS) + checksum(d_ray, N * 3));
    }
    return 0;
}
