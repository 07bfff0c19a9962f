// Microbenchmark: fp32 MFMA shapes on RANDOM register operands (zero operands let the chip hold a higher clock and
// flatter the result -- MI355X_MICROARCH.md, DVFS give-back).  One wave per SIMD, 256 CUs, no memory traffic in the loop.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_shapes.hip -o /tmp/mfma_shapes && /tmp/mfma_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float rnd(unsigned& s) { s = s * 1664525u + 1013904223u; return (float)(s >> 9) * (1.0f / 8388608.0f) - 0.5f; }

// V 0: 32x32x2, one dependent chain   V 1: 32x32x2, 4 accumulators   V 2: 16x16x4, 4 accumulators   V 3: 16x16x4, 8 accumulators
// ZERO: all operands zero (shows the clock effect)
template <int V, bool ZERO>
__global__ __launch_bounds__(256, 1) void k(float* __restrict__ sink, int iters) {
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 17u;
    float a[32], b[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) { a[i] = ZERO ? 0.f : rnd(s); b[i] = ZERO ? 0.f : rnd(s) * 0.1f; }
    float total = 0.f;
    if (V <= 1) {
        f32x16 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                const int t = V == 0 ? 0 : (j & 3);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[(j * 7) & 31], acc[t], 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) total += acc[i][r];
    } else {
        constexpr int NA = V == 2 ? 4 : 8;
        f32x4 acc[NA];
#pragma unroll
        for (int i = 0; i < NA; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 64; ++j) acc[j % NA] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j & 31], b[(j * 7) & 31], acc[j % NA], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NA; ++i) total += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    }
    if (total == 12345.678f) sink[0] = total;
}

template <int V, bool ZERO>
void run(float* sink, const char* name) {
    const int blocks = 256 * 8, iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) k<V, ZERO><<<blocks, 256>>>(sink, iters);
    hipEventRecord(e0);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) k<V, ZERO><<<blocks, 256>>>(sink, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double per_iter = V <= 1 ? 32 * 4096.0 : 64 * 2048.0;      // flop per wave per iteration
    const double flop = (double)blocks * 4 * iters * per_iter;
    printf("%-44s %8.3f ms  %6.1f TFLOP/s  (%.1f %% of 157.3)\n", name, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3 * 100);
}

int main() {
    float* sink;
    hipMalloc(&sink, 16);
    run<0, true>(sink, "32x32x2 one chain, zero operands");
    run<0, false>(sink, "32x32x2 one chain, random operands");
    run<1, false>(sink, "32x32x2 four accumulators, random");
    run<2, false>(sink, "16x16x4 four accumulators, random");
    run<3, false>(sink, "16x16x4 eight accumulators, random");
    run<3, true>(sink, "16x16x4 eight accumulators, zero");
    return 0;
}
