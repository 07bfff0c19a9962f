// Round 6, fourth of the family: ds_write_b128 whose data registers the next vector instruction rewrites, waves sharing SIMDs (the
// compositing kernels have 12 such pairs; LLVM models no hazard here).  Every lane writes a known quad to its own LDS slot, clobbers the
// registers at once, reads the slot back and compares.
//   hipcc --offload-arch=gfx950 -O3 tools/store_war_hazard_lds.hip -o /tmp/swh_lds && /tmp/swh_lds
// Result (profiles/r6_store_hazard_lds.jsonl): 0 of 1.3e9 -- LDS writes take their data with the instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256, 2) void k(unsigned long long* __restrict__ bad_out, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned slot[256 * 4];
    const unsigned tid = blockIdx.x * 256 + threadIdx.x;
    slot[threadIdx.x * 4] = 0;                                       // (the array is really allocated and addressed through its own LDS address)
    const unsigned addr = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned*)&slot[threadIdx.x * 4];
    unsigned long long bad = 0;
    for (int it = 0; it < iters; ++it) {
        const unsigned a0 = tid + it, a1 = (unsigned)it * 2654435761u, a2 = tid ^ 0x5a5a5a5au, a3 = 0x12345678u + it;
        unsigned r0, r1, r2, r3;
        asm volatile("v_mov_b32 v40, %4\n\tv_mov_b32 v41, %5\n\tv_mov_b32 v42, %6\n\tv_mov_b32 v43, %7\n\t"
                     "ds_write_b128 %8, v[40:43]\n\t"
                     "v_mov_b32 v40, 0xdeadbeef\n\tv_mov_b32 v41, 0xdeadbeef\n\tv_mov_b32 v42, 0xdeadbeef\n\tv_mov_b32 v43, 0xdeadbeef\n\t"
                     "s_waitcnt lgkmcnt(0)\n\t"
                     "ds_read_b128 v[44:47], %8\n\t"
                     "s_waitcnt lgkmcnt(0)\n\t"
                     "v_mov_b32 %0, v44\n\tv_mov_b32 %1, v45\n\tv_mov_b32 %2, v46\n\tv_mov_b32 %3, v47"
                     : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)
                     : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(addr)
                     : "memory", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
        bad += (r0 != a0) || (r1 != a1) || (r2 != a2) || (r3 != a3);
    }
    if (bad) atomicAdd(bad_out, bad);
}
int main() {
    unsigned long long* bad; (void)hipMalloc(&bad, 8); (void)hipMemset(bad, 0, 8);
    const int blocks = 8192, iters = 640;
    k<<<blocks, 256>>>(bad, iters);
    (void)hipDeviceSynchronize();
    unsigned long long h = 0; (void)hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
    printf("{\"store\": \"ds_write_b128, data overwritten by the next instruction\", \"waves\": \"share SIMDs\", \"stores\": %llu, \"corrupted\": %llu}\n",
           (unsigned long long)blocks * 256 * iters, h);
    return 0;
}
