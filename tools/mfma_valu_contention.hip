// Microbenchmark: what does an instruction of each class cost a dependent chain of fp32 MFMAs when it is issued between them?
// hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_contention.hip -o tools/mfma_valu_contention
// One wave per SIMD, v_mfma_f32_32x32x2_f32 (64 cycles each); K instructions of class OP per group of 4 MFMAs.
// Result (MI355X, round 3): see HISTORY.md ("What an instruction costs the chain") -- ordinary VALU instructions are NOT free in the shadow of an fp32 MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int STEPS = 2048, REPS = 8;
enum Op { NONE, VADD, VCMP_SGPR, VCNDMASK_SGPR, ACC_WRITE, ACC_READ, SALU, SNOP0, SNOP7, VMEM_LOAD, VMEM_STORE, LDS_READ, SSTORE, VMAX, NOPS };
const char* names[] = {"none", "v_add_u32", "v_cmp_lt_f32 -> sgpr pair", "v_cndmask_b32 (sgpr mask)", "v_accvgpr_write", "v_accvgpr_read", "s_add_u32", "s_nop 0", "s_nop 7",
                       "buffer_load_dwordx4", "buffer_store_dwordx4", "ds_read_b128", "s_store_dwordx2", "v_max_i32"};

template <int OP, int K>
__global__ __launch_bounds__(256, 1) void k(float* __restrict__ out, const float* __restrict__ in, unsigned long long* __restrict__ sout) {
    __shared__ f32x4 lds[256];
    const int lane = threadIdx.x & 63;
    lds[threadIdx.x] = f32x4{1.f, 2.f, 3.f, 4.f};
    __syncthreads();
    float b[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) b[i] = (float)((lane * 131 + i * 71) % 257 - 128) * 3e-3f;
    f32x16 a16;
#pragma unroll
    for (int r = 0; r < 16; ++r) a16[r] = 0.f;
    unsigned x[4] = {1u, 2u, 3u, 4u};
    float fx[4] = {1.f, -2.f, 3.f, -4.f};
    unsigned long long sm = 0;
    unsigned su = 0;
    f32x4 ld = {0.f, 0.f, 0.f, 0.f};
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, 0x7fffffff, 0x00020000);
    __amdgpu_buffer_rsrc_t ws = __builtin_amdgcn_make_buffer_rsrc((void*)(out + 65536), 0, 0x7fffffff, 0x00020000);
    unsigned long long* sp = sout + (blockIdx.x * 4 + (threadIdx.x >> 6)) * 8;
    sp = (unsigned long long*)__builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)sp) ;   // low half only: rebuilt below
    unsigned long long spb = ((unsigned long long)sout & 0xffffffff00000000ull) | (unsigned long long)sp;
#pragma unroll 1
    for (int rep = 0; rep < REPS; ++rep) {
#pragma unroll 1
        for (int s0 = 0; s0 < STEPS; s0 += 16) {
#pragma unroll
            for (int s = 0; s < 16; ++s) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    a16 = __builtin_amdgcn_mfma_f32_32x32x2f32(b[(s + t) % 16], b[(4 * s + t) % 16], a16, 0, 0, 0);
#pragma unroll
                    for (int v = 0; v < K; ++v) {
                        if (v % 4 != t) continue;
                        if (OP == VADD) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[v % 4]) : "v"(x[(v + 1) % 4]));
                        if (OP == VMAX) asm volatile("v_max_i32 %0, %0, %1" : "+v"(x[v % 4]) : "v"(x[(v + 1) % 4]));
                        if (OP == VCMP_SGPR) asm volatile("v_cmp_lt_f32 %0, 0, %1" : "=s"(sm) : "v"(fx[v % 4]));
                        if (OP == VCNDMASK_SGPR) asm volatile("v_cndmask_b32 %0, 0, %1, %2" : "=v"(x[v % 4]) : "v"(x[(v + 1) % 4]), "s"(sm));
                        if (OP == ACC_WRITE) asm volatile("v_accvgpr_write_b32 a0, %0" ::"v"(x[v % 4]) : "a0");
                        if (OP == ACC_READ) asm volatile("v_accvgpr_read_b32 %0, a0" : "=v"(x[v % 4]));
                        if (OP == SALU) asm volatile("s_add_u32 %0, %0, 3" : "+s"(su));
                        if (OP == SNOP0) asm volatile("s_nop 0");
                        if (OP == SNOP7) asm volatile("s_nop 7");
                        if (OP == VMEM_LOAD) ld = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, (s * 4 + t) * 1024, 0));
                        if (OP == VMEM_STORE) {
                            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                            __builtin_amdgcn_raw_buffer_store_b128(u32x4{x[0], x[1], x[2], x[3]}, ws, (int)((blockIdx.x * 256 + threadIdx.x) * 16), 0, 2);
                        }
                        if (OP == LDS_READ) ld = lds[(threadIdx.x + s) & 255];
                        if (OP == SSTORE) asm volatile("s_store_dwordx2 %0, %1, 0x0" ::"s"(sm), "s"(spb) : "memory");
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if (OP == SSTORE) asm volatile("s_dcache_wb" ::: "memory");
    out[blockIdx.x * 256 + threadIdx.x] = a16[0] + a16[5] + a16[10] + a16[15] + (float)(x[0] + x[1] + x[2] + x[3]) + ld[0] + ld[3] + (float)su + (float)(sm & 1);
}

template <int OP, int K>
float run(float* out, const float* in, unsigned long long* sout, float base_ms) {
    const int blocks = 256 * 4;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<OP, K><<<blocks, 256>>>(out, in, sout);
    (void)hipEventRecord(e0);
    k<OP, K><<<blocks, 256>>>(out, in, sout);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * 4 * STEPS * 4 * REPS * 4096.0;
    const double mf = (double)STEPS * 4 * REPS;     // MFMAs per wave
    printf("{\"op\": \"%s\", \"per_4_mfma\": %d, \"ms\": %.3f, \"frac_of_157p3\": %.3f, \"cycles_per_op_at_2p4\": %.1f}\n", names[OP], K, ms, flop / ms / 1e9 / 157.3,
           K ? (ms - base_ms) * 2.4e6 / (mf * K / 4.0 * (blocks * 4 / 1024.0)) : 0.0);      // waves run one after the other on a SIMD
    return ms;
}

int main() {
    float *out, *in; unsigned long long* sout;
    (void)hipMalloc(&out, 64 << 20); (void)hipMalloc(&in, 1 << 20); (void)hipMalloc(&sout, 1 << 20);
    (void)hipMemset(in, 0, 1 << 20);
    (void)run<NONE, 0>(out, in, sout, 0.f);                       // warm-up (clocks, code objects)
    const float base = run<NONE, 0>(out, in, sout, 0.f);         // the bare chain: what every other case is measured against
    run<VADD, 4>(out, in, sout, base); run<VADD, 16>(out, in, sout, base);
    run<VMAX, 4>(out, in, sout, base);
    run<VCMP_SGPR, 4>(out, in, sout, base); run<VCNDMASK_SGPR, 4>(out, in, sout, base);
    run<ACC_WRITE, 4>(out, in, sout, base); run<ACC_READ, 4>(out, in, sout, base);
    run<SALU, 4>(out, in, sout, base); run<SALU, 16>(out, in, sout, base);
    run<SNOP0, 4>(out, in, sout, base); run<SNOP7, 4>(out, in, sout, base);
    run<VMEM_LOAD, 4>(out, in, sout, base); run<VMEM_STORE, 4>(out, in, sout, base); run<LDS_READ, 4>(out, in, sout, base);
    run<SSTORE, 4>(out, in, sout, base);
    return 0;
}
