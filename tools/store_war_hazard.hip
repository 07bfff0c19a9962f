// Does a VALU write to the data registers of a 16-byte buffer store, issued as the very next instruction, ever reach memory?
// LLVM's hazard recogniser inserts a wait state after a >64-bit store only when the store's soffset is NOT a register
// (GCNHazardRecognizer::createsVALUHazard, "this hazard only exists if the instruction is not using a register in the soffset field").
// hipcc --offload-arch=gfx950 -O3 tools/store_war_hazard.hip -o tools/store_war_hazard
// MODE 0: soffset in an SGPR, next instruction overwrites the data;  MODE 1: the same with one s_nop between.
// Result on MI355X (round 3, two runs): one wave per SIMD 0 / 67,108,864 stores corrupted; waves sharing SIMDs 2,032 and 1,984; with the
// s_nop 0 / 67,108,864.  Consequences: DESIGN.md section 3.7, tools/check_store_hazard.py (a CPU test).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int WAVES_PER_SIMD>
__global__ __launch_bounds__(256, WAVES_PER_SIMD) void k(unsigned* __restrict__ out, int iters, int soff_bytes) {
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, 0x7fffffff, 0x00020000);
    const unsigned tid = blockIdx.x * 256 + threadIdx.x;
    const int soff = __builtin_amdgcn_readfirstlane(soff_bytes);
    for (int it = 0; it < iters; ++it) {
        const unsigned a0 = tid, a1 = (unsigned)it, a2 = tid ^ 0x5a5a5a5au, a3 = 0x12345678u;
        const unsigned voff = (tid * (unsigned)iters + (unsigned)it) * 16u;
        // fixed data registers v[40:43]: filled, stored, and v40 / v42 overwritten by the instructions right behind the store
        if (MODE == 0)
            asm volatile("v_mov_b32 v40, %0\n\tv_mov_b32 v41, %1\n\tv_mov_b32 v42, %2\n\tv_mov_b32 v43, %3\n\t"
                         "buffer_store_dwordx4 v[40:43], %4, %5, %6 offen\n\tv_mov_b32 v40, 0xdeadbeef\n\tv_mov_b32 v42, 0xdeadbeef"
                         :: "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(voff), "s"(rs), "s"(soff) : "memory", "v40", "v41", "v42", "v43");
        else
            asm volatile("v_mov_b32 v40, %0\n\tv_mov_b32 v41, %1\n\tv_mov_b32 v42, %2\n\tv_mov_b32 v43, %3\n\t"
                         "buffer_store_dwordx4 v[40:43], %4, %5, %6 offen\n\ts_nop 0\n\tv_mov_b32 v40, 0xdeadbeef\n\tv_mov_b32 v42, 0xdeadbeef"
                         :: "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(voff), "s"(rs), "s"(soff) : "memory", "v40", "v41", "v42", "v43");
    }
}

template <int MODE, int W>
void run(unsigned* out, int blocks, int iters, const char* name) {
    const size_t n = (size_t)blocks * 256 * iters * 4;
    (void)hipMemset(out, 0, n * 4);
    k<MODE, W><<<blocks, 256>>>(out, iters, 0);
    (void)hipDeviceSynchronize();
    std::vector<unsigned> h(n);
    (void)hipMemcpy(h.data(), out, n * 4, hipMemcpyDeviceToHost);
    long bad = 0;
    for (size_t t = 0; t < (size_t)blocks * 256; ++t)
        for (int it = 0; it < iters; ++it) {
            const unsigned* q = &h[(t * iters + it) * 4];
            if (q[0] != (unsigned)t || q[1] != (unsigned)it || q[2] != ((unsigned)t ^ 0x5a5a5a5au) || q[3] != 0x12345678u) ++bad;
        }
    printf("{\"case\": \"%s\", \"workgroups\": %d, \"stores\": %zu, \"corrupted\": %ld}\n", name, blocks, n / 4, bad);
}

int main() {
    unsigned* out; (void)hipMalloc(&out, (size_t)4096 * 256 * 64 * 16);      // 1 GiB: 67 M stores of 16 bytes in every case
    run<0, 1>(out, 256, 1024, "soffset SGPR, data overwritten by the next instruction, 256 workgroups = one wave per SIMD");
    run<0, 2>(out, 4096, 64, "soffset SGPR, data overwritten by the next instruction, 4096 workgroups = waves share SIMDs");
    run<1, 2>(out, 4096, 64, "the same with s_nop 0 between");
    return 0;
}
