/* Calling the boundary from plain C: alpha compositing (NeRF.composite, reference model/nerf.py:458-474) of a few
 * rays through niw_composite_fwd, checked against a scalar loop written from the same formulas.
 *
 *   gcc -std=c11 -D__HIP_PLATFORM_AMD__ examples/composite_c_abi.c -Iinclude -I/opt/rocm/include \
 *       -Lneural_invertible_warp_amd -L/opt/rocm/lib -lniw_hip -lamdhip64 -lm \
 *       -Wl,-rpath,$PWD/neural_invertible_warp_amd -Wl,-rpath,/opt/rocm/lib -o /tmp/composite_c_abi && /tmp/composite_c_abi
 * (a C compiler, not hipcc: the define only selects the AMD flavour of the HIP runtime headers)
 *
 * Nothing but <hip/hip_runtime_api.h> (device memory, one stream) and include/niw.h is needed: no C++ types cross
 * the boundary, the caller owns every buffer, errors come back as status codes + niw_last_error_string(). */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "niw.h"

#define N 7
#define S 48
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

static float frand(unsigned* s) { *s = *s * 1664525u + 1013904223u; return (float)(*s >> 8) / 16777216.0f; }

int main(void) {
    static float ray[N * 3], rgb_s[N * S * 3], sigma[N * S], depth[N * S], rgb[N * 3], dep[N], opa[N], prob[N * S];
    unsigned seed = 12345u;
    for (int r = 0; r < N; ++r) {
        for (int c = 0; c < 3; ++c) ray[r * 3 + c] = frand(&seed) * 2.f - 1.f;
        float d = 0.5f;
        for (int s = 0; s < S; ++s) {
            d += 0.02f + 0.1f * frand(&seed);
            depth[r * S + s] = d;
            sigma[r * S + s] = frand(&seed) < 0.4f ? 0.f : 3.f * frand(&seed);
            for (int c = 0; c < 3; ++c) rgb_s[(r * S + s) * 3 + c] = frand(&seed);
        }
    }
    float *d_ray, *d_rgb_s, *d_sigma, *d_depth, *d_rgb, *d_dep, *d_opa, *d_prob;
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    CHECK(hipMalloc((void**)&d_ray, sizeof ray)); CHECK(hipMalloc((void**)&d_rgb_s, sizeof rgb_s));
    CHECK(hipMalloc((void**)&d_sigma, sizeof sigma)); CHECK(hipMalloc((void**)&d_depth, sizeof depth));
    CHECK(hipMalloc((void**)&d_rgb, sizeof rgb)); CHECK(hipMalloc((void**)&d_dep, sizeof dep));
    CHECK(hipMalloc((void**)&d_opa, sizeof opa)); CHECK(hipMalloc((void**)&d_prob, sizeof prob));
    CHECK(hipMemcpy(d_ray, ray, sizeof ray, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_rgb_s, rgb_s, sizeof rgb_s, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_sigma, sigma, sizeof sigma, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_depth, depth, sizeof depth, hipMemcpyHostToDevice));

    int rc = niw_composite_fwd(d_ray, d_rgb_s, d_sigma, d_depth, N, S, 0, 0.f, d_rgb, d_dep, d_opa, d_prob, (niw_stream_t)st);
    if (rc != NIW_OK) { fprintf(stderr, "niw_composite_fwd: %d %s\n", rc, niw_last_error_string()); return 1; }
    CHECK(hipStreamSynchronize(st));
    CHECK(hipMemcpy(rgb, d_rgb, sizeof rgb, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(opa, d_opa, sizeof opa, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(dep, d_dep, sizeof dep, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(prob, d_prob, sizeof prob, hipMemcpyDeviceToHost));

    double worst = 0.0;
    for (int r = 0; r < N; ++r) {
        const double len = sqrt((double)ray[r * 3] * ray[r * 3] + (double)ray[r * 3 + 1] * ray[r * 3 + 1] + (double)ray[r * 3 + 2] * ray[r * 3 + 2]);
        double acc = 0.0, c3[3] = {0, 0, 0}, o = 0.0, dd = 0.0;
        for (int s = 0; s < S; ++s) {
            const double delta = s == S - 1 ? 1e10 : (double)depth[r * S + s + 1] - depth[r * S + s];
            const double sd = sigma[r * S + s] * delta * len, w = exp(-acc) * (1.0 - exp(-sd));
            acc += sd;
            for (int c = 0; c < 3; ++c) c3[c] += w * rgb_s[(r * S + s) * 3 + c];
            o += w; dd += w * depth[r * S + s];
            worst = fmax(worst, fabs(w - prob[r * S + s]));
        }
        for (int c = 0; c < 3; ++c) worst = fmax(worst, fabs(c3[c] - rgb[r * 3 + c]));
        worst = fmax(worst, fabs(o - opa[r]));
        worst = fmax(worst, fabs(dd - dep[r]) / fmax(1.0, fabs(dd)));
    }
    /* error path: a null pointer is refused with a status code and a message, nothing is launched */
    rc = niw_composite_fwd(NULL, d_rgb_s, d_sigma, d_depth, N, S, 0, 0.f, d_rgb, d_dep, d_opa, d_prob, (niw_stream_t)st);
    printf("libniw_hip version %d, max |diff| vs scalar loop %.3g, null-pointer status %d (%s)\n", niw_version(), worst, rc, niw_last_error_string());
    hipFree(d_ray); hipFree(d_rgb_s); hipFree(d_sigma); hipFree(d_depth); hipFree(d_rgb); hipFree(d_dep); hipFree(d_opa); hipFree(d_prob);
    if (worst > 2e-5 || rc != NIW_ERR_INVALID_ARG) { printf("FAIL\n"); return 1; }
    printf("OK\n");
    return 0;
}
