// Round 6 follow-up to tools/store_war_hazard.hip: does the store-data hazard of gfx950 (a buffer_store_dwordx4 with an SGPR soffset followed
// IMMEDIATELY by a vector write of its data registers stores the new value when waves share a SIMD) also fire when the wave that shares the
// SIMD belongs to ANOTHER kernel -- another stream of the same process (the library's second stream runs small kernels beside the training
// forward, which keeps 12 such adjacent pairs on the strength of its one wave per SIMD), or another process (ranks that time-slice one GPU)?
// The hazard kernel runs ONE workgroup per CU (one wave per SIMD, the forward's geometry) while a filler kernel of another stream keeps
// every SIMD supplied with extra waves of vector-ALU or memory work.
//   hipcc --offload-arch=gfx950 -O3 tools/store_war_hazard_foreign.hip -o /tmp/swh_foreign && /tmp/swh_foreign > profiles/r6_store_hazard_foreign.jsonl
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define FILL "v_mov_b32 v40, %0\n\tv_mov_b32 v41, %1\n\tv_mov_b32 v42, %2\n\tv_mov_b32 v43, %3\n\t"
#define STORE4 "buffer_store_dwordx4 v[40:43], %4, %5, %6 offen\n\t"
#define CLOBBER "v_mov_b32 v40, 0xdeadbeef\n\tv_mov_b32 v41, 0xdeadbeef\n\tv_mov_b32 v42, 0xdeadbeef\n\tv_mov_b32 v43, 0xdeadbeef"
#define ARGS :: "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(voff), "s"(rs), "s"(soff) : "memory", "v40", "v41", "v42", "v43"

template <bool NOP>
__global__ __launch_bounds__(256, 1) void hazard(unsigned* __restrict__ out, int iters, int soff_bytes) {
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, 0x7fffffff, 0x00020000);
    const unsigned tid = blockIdx.x * 256 + threadIdx.x;
    const int soff = __builtin_amdgcn_readfirstlane(soff_bytes);
    for (int it = 0; it < iters; ++it) {
        const unsigned a0 = tid, a1 = (unsigned)it, a2 = tid ^ 0x5a5a5a5au, a3 = 0x12345678u;
        const unsigned voff = (tid * (unsigned)iters + (unsigned)it) * 16u;
        if (NOP) asm volatile(FILL STORE4 "s_nop 0\n\t" CLOBBER ARGS);
        else asm volatile(FILL STORE4 CLOBBER ARGS);
    }
}

// filler: KIND 0 = a dependent vector-ALU chain, 1 = streaming loads + stores of its own buffer, 2 = both
template <int KIND>
__global__ __launch_bounds__(256) void filler(float* __restrict__ buf, long long n, int rounds) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    float x = (float)i * 1e-3f;
    for (int r = 0; r < rounds; ++r) {
        if (KIND != 1)
            for (int k = 0; k < 64; ++k) x = fmaf(x, 1.0001f, 0.5f) * 0.9999f;
        if (KIND != 0) {
            const long long j = (i + (long long)r * 4099) % n;
            x += buf[j];
            buf[j] = x;
        }
    }
    if (x == 123.456f) buf[0] = x;
}

static long check(const std::vector<unsigned>& h, int blocks, int iters) {
    long bad = 0;
    for (size_t t = 0; t < (size_t)blocks * 256; ++t)
        for (int it = 0; it < iters; ++it) {
            const unsigned* q = &h[(t * iters + it) * 4];
            bad += (q[0] == (unsigned)t && q[1] == (unsigned)it && q[2] == ((unsigned)t ^ 0x5a5a5a5au) && q[3] == 0x12345678u) ? 0 : 1;
        }
    return bad;
}

template <bool NOP, int KIND>
void run(unsigned* out, float* fbuf, long long fn, const char* what) {
    const int blocks = 256, iters = 1024, reps = 8;
    const size_t n = (size_t)blocks * 256 * iters * 4;
    hipStream_t a, b;
    (void)hipStreamCreateWithFlags(&a, hipStreamNonBlocking);
    (void)hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
    long bad = 0;
    std::vector<unsigned> h(n);
    for (int rep = 0; rep < reps; ++rep) {
        (void)hipMemsetAsync(out, 0, n * 4, a);
        (void)hipStreamSynchronize(a);
        if (KIND >= 0) filler<(KIND < 0 ? 0 : KIND)><<<8192, 256, 0, b>>>(fbuf, fn, 40);      // 32 small workgroups per CU: extra waves for every SIMD
        hazard<NOP><<<blocks, 256, 0, a>>>(out, iters, 0);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h.data(), out, n * 4, hipMemcpyDeviceToHost);
        bad += check(h, blocks, iters);
    }
    printf("{\"store\": \"buffer_store_dwordx4, soffset in an SGPR, data overwritten by the next instruction%s\", \"hazard_kernel\": \"one workgroup per CU, one wave per SIMD\", "
           "\"beside_it\": \"%s\", \"stores\": %zu, \"corrupted\": %ld}\n", NOP ? " but one (s_nop 0 between)" : "", what, (size_t)reps * n / 4, bad);
    fflush(stdout);
    (void)hipStreamDestroy(a); (void)hipStreamDestroy(b);
}

int main(int argc, char** argv) {
    unsigned* out; (void)hipMalloc(&out, (size_t)256 * 256 * 1024 * 16);
    const long long fn = 1ll << 26;
    float* fbuf; (void)hipMalloc(&fbuf, fn * 4); (void)hipMemset(fbuf, 0, fn * 4);
    const bool filler_only = argc > 1;         // a second PROCESS started with an argument only runs fillers (the foreign-process case)
    if (filler_only) {
        for (int i = 0; i < 400; ++i) { filler<2><<<8192, 256>>>(fbuf, fn, 40); }
        (void)hipDeviceSynchronize();
        return 0;
    }
    run<false, -1>(out, fbuf, fn, "nothing (alone on the device)");
    run<false, 0>(out, fbuf, fn, "a vector-ALU kernel of another stream of this process");
    run<false, 1>(out, fbuf, fn, "a load / store kernel of another stream of this process");
    run<false, 2>(out, fbuf, fn, "a vector-ALU + load / store kernel of another stream of this process");
    run<true, 2>(out, fbuf, fn, "a vector-ALU + load / store kernel of another stream of this process");
    return 0;
}
