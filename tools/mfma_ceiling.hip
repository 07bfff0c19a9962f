// Microbenchmark: what does one wave per SIMD sustain on v_mfma_f32_32x32x2_f32 in the shapes the
// MLP kernels use?  hipcc --offload-arch=gfx950 -O3 tools/mfma_ceiling.hip -o tools/mfma_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0)

constexpr int KB = 32, NBLK = 8, LAYERS = 8;   // 8 * 8 * 32 * 4 = 8192 MFMAs per wave

// V: 0 = single dependent chain, registers only; 1 = 8 independent accumulators round robin;
//    2 = chain + fragment ring loads (L2 resident weights); 3 = 2 + one store per 8 MFMAs
// V: 4 = like 2 but weight fragments through buffer_load (SGPR descriptor + SGPR offset + one 32-bit lane offset);
//    5 = 4 + stores through buffer_store_dword
template <int V>
__global__ __launch_bounds__(256, 1) void k(const f32x4* __restrict__ w, float* __restrict__ out, float* __restrict__ sink, long long pitch) {
    const int lane = threadIdx.x & 63;
    float b[128];
#pragma unroll
    for (int i = 0; i < 128; ++i) b[i] = (float)((lane * 131 + i * 71) % 257 - 128) * 3e-3f;
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const long long m = (long long)blockIdx.x * 256 + threadIdx.x;
    __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, 0x7fffffff, 0x00020000);
    const int lane16 = lane * 16;
    const int m4 = (int)(m * 4);
#pragma unroll 1
    for (int l = 0; l < LAYERS; ++l) {
        const f32x4* wp = w + (long long)l * KB * NBLK * 64;
        if (V == 1) {
#pragma unroll
            for (int q = 0; q < KB; ++q) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int nb = 0; nb < NBLK; ++nb) acc[nb] = MFMA(b[(4 * q + t + nb) & 127], b[4 * q + t], acc[nb]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            f32x4 ring[8];
            const int lbase = l * KB * NBLK * 1024;
            if (V >= 4) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    ring[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wr, lane16, lbase + ((i % KB) * NBLK + i / KB) * 1024, 0));
            } else if (V >= 2) {
#pragma unroll
                for (int i = 0; i < 8; ++i) ring[i] = wp[((i % KB) * NBLK + i / KB) * 64 + lane];
            }
#pragma unroll
            for (int nb = 0; nb < NBLK; ++nb) {
#pragma unroll
                for (int q = 0; q < KB; ++q) {
                    const int i = nb * KB + q;
                    f32x4 a;
                    if (V >= 2) {
                        a = ring[i % 8];
                        if (i + 8 < NBLK * KB) {
                            if (V >= 4)
                                ring[i % 8] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wr, lane16, lbase + (((i + 8) % KB) * NBLK + (i + 8) / KB) * 1024, 0));
                            else
                                ring[i % 8] = wp[(((i + 8) % KB) * NBLK + (i + 8) / KB) * 64 + lane];
                        }
                    } else {
                        a = f32x4{b[(q + 1) & 127], b[(q + 2) & 127], b[(q + 3) & 127], b[(q + 4) & 127]};
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        acc[0] = MFMA(a[t], b[4 * q + t], acc[0]);
                        if (V == 3 && (q & 1) == 0 && t == 1) (out + (long long)(l * 256 + nb * 32 + q / 2) * pitch)[m] = b[q];
                        if (V == 5 && (q & 1) == 0 && t == 1) {
                            __amdgpu_buffer_rsrc_t orr = __builtin_amdgcn_make_buffer_rsrc((void*)(out + (long long)l * 256 * pitch), 0, 0x7fffffff, 0x00020000);
                            __builtin_amdgcn_raw_buffer_store_b32(b[q], orr, m4, (int)((nb * 32 + q / 2) * pitch * 4), 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) sink[0] = s;
}

template <int V>
void run(const f32x4* w, float* out, float* sink, int blocks, long long pitch) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 2; ++i) k<V><<<blocks, 256>>>(w, out, sink, pitch);
    hipEventRecord(a);
    const int reps = 5;
    for (int i = 0; i < reps; ++i) k<V><<<blocks, 256>>>(w, out, sink, pitch);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    ms /= reps;
    const double flop = (double)blocks * 4 * LAYERS * NBLK * KB * 4 * 4096.0;
    printf("variant %d: %.3f ms  %.1f TFLOP/s  (%.1f %% of 157.3)\n", V, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3 * 100);
}

int main() {
    const int blocks = 6129;
    const long long pitch = (long long)blocks * 256;
    f32x4* w; float *out, *sink;
    hipMalloc(&w, sizeof(f32x4) * LAYERS * KB * NBLK * 64);
    {   // random weights: all-zero operands let the chip hold a higher clock (MI355X_MICROARCH.md, DVFS give-back)
        std::vector<float> h((size_t)LAYERS * KB * NBLK * 64 * 4);
        unsigned s = 12345u;
        for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((float)(s >> 9) * (1.0f / 8388608.0f) - 0.5f) * 0.1f; }
        hipMemcpy(w, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
    }
    hipMalloc(&out, sizeof(float) * pitch * LAYERS * 256);
    hipMalloc(&sink, 16);
    run<0>(w, out, sink, blocks, pitch);
    run<1>(w, out, sink, blocks, pitch);
    run<2>(w, out, sink, blocks, pitch);
    run<3>(w, out, sink, blocks, pitch);
    run<4>(w, out, sink, blocks, pitch);
    run<5>(w, out, sink, blocks, pitch);
    return 0;
}
