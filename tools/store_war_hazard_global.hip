// Round 6, third microbenchmark of the family (tools/store_war_hazard.hip, tools/store_war_hazard_foreign.hip): the same experiment for
// global_store_dwordx4 -- SGPR base (saddr) and 64-bit vector address -- whose data registers are rewritten by the next vector instruction,
// waves sharing SIMDs.  Raw inline assembly, so that nothing is padded: LLVM's hazard recogniser KNOWS these forms (FLAT stores of more than
// 64 bits: two wait states on gfx940+) and pads them in compiled code; this measures what it protects from, and how many wait states it takes.
//   hipcc --offload-arch=gfx950 -O3 tools/store_war_hazard_global.hip -o /tmp/swh_global && /tmp/swh_global                (adjacent)
//   hipcc --offload-arch=gfx950 -O3 -DWITH_NOP tools/store_war_hazard_global.hip -o /tmp/swh_global_nop && /tmp/swh_global_nop   (one s_nop 0)
// Result (profiles/r6_store_hazard_global.jsonl): adjacent 25 % of 6.7e7 stores corrupted in both forms; with ONE wait state still 1.3-1.5e4
// -- global stores need the two wait states LLVM gives them.  (The SGPR-soffset buffer store, which LLVM does not pad, was clean with one
// in every measurement; the training forward's fix uses two all the same.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#ifdef WITH_NOP
#define NOPSTR "s_nop 0\n\t"
#else
#define NOPSTR ""
#endif
template <int FORM>   // 0: global_store saddr; 1: global_store vaddr64 (LLVM pads this one itself when it schedules; here raw asm)
__global__ __launch_bounds__(256, 2) void k(unsigned* __restrict__ out, int iters) {
    const unsigned tid = blockIdx.x * 256 + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        const unsigned a0 = tid, a1 = (unsigned)it, a2 = tid ^ 0x5a5a5a5au, a3 = 0x12345678u;
        const unsigned voff = (tid * (unsigned)iters + (unsigned)it) * 16u;
        if (FORM == 0)
            asm volatile("v_mov_b32 v40, %0\n\tv_mov_b32 v41, %1\n\tv_mov_b32 v42, %2\n\tv_mov_b32 v43, %3\n\t"
                         "global_store_dwordx4 %4, v[40:43], %5\n\t" NOPSTR
                         "v_mov_b32 v40, 0xdeadbeef\n\tv_mov_b32 v41, 0xdeadbeef\n\tv_mov_b32 v42, 0xdeadbeef\n\tv_mov_b32 v43, 0xdeadbeef"
                         :: "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(voff), "s"(out) : "memory", "v40", "v41", "v42", "v43");
        else {
            unsigned long long addr = (unsigned long long)out + voff;
            asm volatile("v_mov_b32 v40, %0\n\tv_mov_b32 v41, %1\n\tv_mov_b32 v42, %2\n\tv_mov_b32 v43, %3\n\t"
                         "global_store_dwordx4 %4, v[40:43], off\n\t" NOPSTR
                         "v_mov_b32 v40, 0xdeadbeef\n\tv_mov_b32 v41, 0xdeadbeef\n\tv_mov_b32 v42, 0xdeadbeef\n\tv_mov_b32 v43, 0xdeadbeef"
                         :: "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(addr) : "memory", "v40", "v41", "v42", "v43");
        }
    }
}
template <int FORM> void run(unsigned* out, const char* what) {
    const int blocks = 4096, iters = 64;
    const size_t n = (size_t)blocks * 256 * iters * 4;
    (void)hipMemset(out, 0, n * 4);
    k<FORM><<<blocks, 256>>>(out, iters);
    (void)hipDeviceSynchronize();
    std::vector<unsigned> h(n);
    (void)hipMemcpy(h.data(), out, n * 4, hipMemcpyDeviceToHost);
    long bad = 0;
    for (size_t t = 0; t < (size_t)blocks * 256; ++t)
        for (int it = 0; it < iters; ++it) {
            const unsigned* q = &h[(t * iters + it) * 4];
            bad += (q[0] == (unsigned)t && q[1] == (unsigned)it && q[2] == ((unsigned)t ^ 0x5a5a5a5au) && q[3] == 0x12345678u) ? 0 : 1;
        }
    printf("{\"store\": \"%s, data overwritten by the next instruction\", \"waves\": \"share SIMDs\", \"stores\": %zu, \"corrupted\": %ld}\n", what, n / 4, bad);
}
int main() {
    unsigned* out; (void)hipMalloc(&out, (size_t)4096 * 256 * 64 * 16);
    run<0>(out, "global_store_dwordx4 with an SGPR base (saddr)");
    run<1>(out, "global_store_dwordx4 with a 64-bit vector address");
    return 0;
}
