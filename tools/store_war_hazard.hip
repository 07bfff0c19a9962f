// Does a VALU write to the data registers of a buffer store whose soffset is an SGPR, issued 0, 1 or 2 instructions behind the store,
// ever reach memory?  LLVM's hazard recogniser inserts wait states after a > 64-bit store only when the store's soffset is NOT a register
// (GCNHazardRecognizer::createsVALUHazard: "this hazard only exists if the instruction is not using a register in the soffset field"),
// so hipcc can schedule such a pair; round 3 found the x4 / next-instruction case corrupting ~2,000 of 67 M stores whenever waves share a
// SIMD (HISTORY.md, "A store-data hazard of gfx950").  Round 4 closes the matrix the round-3 review asked for:
//     store width   dwordx2 | dwordx3 | dwordx4        (x2 is what the bf16 workspaces of niw_mlp_fast.hip store)
//     distance      the overwriting VALU is the next instruction | one unrelated VALU between | two between | one s_nop 0 between
//     occupancy     one wave per SIMD | waves share SIMDs
// hipcc --offload-arch=gfx950 -O3 tools/store_war_hazard.hip -o /tmp/store_war_hazard && /tmp/store_war_hazard > profiles/r4_store_hazard.jsonl
// Result and consequences: profiles/r4_store_hazard.jsonl, tools/check_store_hazard.py (a CPU test over the assembly of every kernel).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define FILL "v_mov_b32 v40, %0\n\tv_mov_b32 v41, %1\n\tv_mov_b32 v42, %2\n\tv_mov_b32 v43, %3\n\t"
#define STORE2 "buffer_store_dwordx2 v[40:41], %4, %5, %6 offen\n\t"
#define STORE3 "buffer_store_dwordx3 v[40:42], %4, %5, %6 offen\n\t"
#define STORE4 "buffer_store_dwordx4 v[40:43], %4, %5, %6 offen\n\t"
#define OTHER "v_add_u32 v44, v44, v45\n\t"                      // an unrelated vector instruction
#define CLOBBER "v_mov_b32 v40, 0xdeadbeef\n\tv_mov_b32 v41, 0xdeadbeef\n\tv_mov_b32 v42, 0xdeadbeef\n\tv_mov_b32 v43, 0xdeadbeef"
#define ARGS :: "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(voff), "s"(rs), "s"(soff) : "memory", "v40", "v41", "v42", "v43", "v44", "v45"

// GAP: 0 next instruction, 1 / 2 unrelated VALUs between, 3 one s_nop 0 between
template <int WIDTH, int GAP, int WAVES_PER_SIMD>
__global__ __launch_bounds__(256, WAVES_PER_SIMD) void k(unsigned* __restrict__ out, int iters, int soff_bytes) {
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, 0x7fffffff, 0x00020000);
    const unsigned tid = blockIdx.x * 256 + threadIdx.x;
    const int soff = __builtin_amdgcn_readfirstlane(soff_bytes);
    for (int it = 0; it < iters; ++it) {
        const unsigned a0 = tid, a1 = (unsigned)it, a2 = tid ^ 0x5a5a5a5au, a3 = 0x12345678u;
        const unsigned voff = (tid * (unsigned)iters + (unsigned)it) * 16u;
#define CASE(W, STORE)                                                                          \
        if (WIDTH == W) {                                                                       \
            if (GAP == 0) asm volatile(FILL STORE CLOBBER ARGS);                                \
            else if (GAP == 1) asm volatile(FILL STORE OTHER CLOBBER ARGS);                     \
            else if (GAP == 2) asm volatile(FILL STORE OTHER OTHER CLOBBER ARGS);               \
            else asm volatile(FILL STORE "s_nop 0\n\t" CLOBBER ARGS);                           \
        }
        CASE(2, STORE2)
        CASE(3, STORE3)
        CASE(4, STORE4)
#undef CASE
    }
}

template <int WIDTH, int GAP, int W>
void run(unsigned* out, int blocks, int iters) {
    const size_t n = (size_t)blocks * 256 * iters * 4;
    (void)hipMemset(out, 0, n * 4);
    k<WIDTH, GAP, W><<<blocks, 256>>>(out, iters, 0);
    (void)hipDeviceSynchronize();
    std::vector<unsigned> h(n);
    (void)hipMemcpy(h.data(), out, n * 4, hipMemcpyDeviceToHost);
    long bad = 0;
    for (size_t t = 0; t < (size_t)blocks * 256; ++t)
        for (int it = 0; it < iters; ++it) {
            const unsigned* q = &h[(t * iters + it) * 4];
            const unsigned want[4] = {(unsigned)t, (unsigned)it, (unsigned)t ^ 0x5a5a5a5au, 0x12345678u};
            bool ok = true;
            for (int c = 0; c < WIDTH; ++c) ok = ok && q[c] == want[c];
            bad += ok ? 0 : 1;
        }
    static const char* gap[4] = {"next instruction", "one unrelated VALU between", "two unrelated VALUs between", "one s_nop 0 between"};
    printf("{\"store\": \"buffer_store_dwordx%d, soffset in an SGPR\", \"data_overwritten_by\": \"%s\", \"waves\": \"%s\", \"workgroups\": %d, \"stores\": %zu, "
           "\"corrupted\": %ld}\n", WIDTH, gap[GAP], W == 1 ? "one per SIMD" : "share SIMDs", blocks, n / 4, bad);
    fflush(stdout);
}

template <int WIDTH>
void width(unsigned* out) {
    run<WIDTH, 0, 1>(out, 256, 1024);
    run<WIDTH, 0, 2>(out, 4096, 64);
    run<WIDTH, 1, 2>(out, 4096, 64);
    run<WIDTH, 2, 2>(out, 4096, 64);
    run<WIDTH, 3, 2>(out, 4096, 64);
}

int main() {
    unsigned* out; (void)hipMalloc(&out, (size_t)4096 * 256 * 64 * 16);      // 1 GiB: 67 M stores in every case
    width<4>(out);
    width<3>(out);
    width<2>(out);
    return 0;
}
