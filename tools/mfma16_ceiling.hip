// Microbenchmark: what do one / two waves per SIMD sustain on v_mfma_f32_16x16x4_f32 (32-cycle issue, 40-cycle dependent latency)?
// hipcc --offload-arch=gfx950 -O3 tools/mfma16_ceiling.hip -o tools/mfma16_ceiling
// V = number of independent accumulation chains per wave (1, 2, 4); OCC = waves per SIMD (1, 2); L = 1: weight fragments from L2 through a ring
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)
constexpr int STEPS = 4096, REPS = 16;     // MFMAs per wave = STEPS * 4 * REPS

template <int V, int OCC, int L>
__global__ __launch_bounds__(256, OCC) void k(const f32x4* __restrict__ w, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    float b[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) b[i] = (float)((lane * 131 + i * 71) % 257 - 128) * 3e-3f;
    f32x4 acc[V];
#pragma unroll
    for (int i = 0; i < V; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, 0x7fffffff, 0x00020000);
#pragma unroll 1
    for (int rep = 0; rep < REPS; ++rep) {
        f32x4 ring[8];
        if (L) {
#pragma unroll
            for (int i = 0; i < 8; ++i) ring[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wr, lane * 16, i * 1024, 0));
        }
#pragma unroll 1
        for (int s0 = 0; s0 < STEPS; s0 += 64) {
#pragma unroll
            for (int s = 0; s < 64; ++s) {
                f32x4 a;
                if (L) {
                    a = ring[s % 8];
                    ring[s % 8] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wr, lane * 16, ((s0 + s + 8) % 2048) * 1024, 0));
                } else {
                    a = f32x4{b[s % 64], b[(s + 1) % 64], b[(s + 2) % 64], b[(s + 3) % 64]};
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[(4 * s + t) % V] = MFMA(a[t], b[(4 * s + t) % 64], acc[(4 * s + t) % V]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) v += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = v;
}

template <int V, int OCC, int L>
void run(const f32x4* w, float* out) {
    const int blocks = 256 * OCC * 4;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<V, OCC, L><<<blocks, 256>>>(w, out);
    hipEventRecord(e0);
    k<V, OCC, L><<<blocks, 256>>>(w, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * 4 * STEPS * 4 * REPS * 2048.0;
    printf("{\"chains\": %d, \"waves_per_simd\": %d, \"fragments_from_l2\": %d, \"ms\": %.3f, \"tflops\": %.1f, \"frac_of_157p3\": %.3f}\n", V, OCC, L, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3);
}

int main() {
    f32x4* w; float* out;
    hipMalloc(&w, 2048 * 1024 + 65536); hipMemset(w, 0, 2048 * 1024 + 65536);
    hipMalloc(&out, 256 * 8 * 256 * 4);
    run<1, 1, 0>(w, out); run<2, 1, 0>(w, out); run<4, 1, 0>(w, out);
    run<1, 2, 0>(w, out); run<2, 2, 0>(w, out);
    run<1, 1, 1>(w, out); run<2, 1, 1>(w, out); run<1, 2, 1>(w, out); run<2, 2, 1>(w, out);
    return 0;
}
