// In-kernel time stamps of the DIAGNOSTIC build (never part of libniw_hip.so):
//     make -C neural_invertible_warp_amd/csrc VARIANT=trace EXTRA=-DNIW_TRACE      ->  ../libniw_hip_trace.so   (NIW_LIB_PATH selects it)
// Lane 0 of every wave writes s_memrealtime -- the 100 MHz constant clock, 10 ns per tick, the same on every CU and XCD -- into
// trace[wave][slot] at fixed points of the register-chained field-MLP kernels and of the NT GEMM: kernel entry, first matrix instruction,
// every layer boundary, last store.  Slot 15 holds the wave's placement (XCC_ID << 32 | HW_ID); slots 13 / 14 hold s_memtime -- the SHADER
// clock's cycle counter -- at entry and at the last stamp, so that (delta s_memtime) / (delta s_memrealtime) x 100 MHz is the clock the wave's
// XCD actually held over the kernel (MI355X_MICROARCH.md, DVFS give-back item 6).  tools/launch_trace.py turns a launch's
// stamps into dispatch skew over its workgroups, time to the first MFMA, per-layer durations and the exposed tail.
// One buffer per translation unit (no relocatable device code in this build): niw_trace_set_<unit>(buffer, waves[, kind]).
#pragma once
#ifdef NIW_TRACE
#include <hip/hip_runtime.h>
#define NIW_TRACE_SLOTS 16
namespace niw_trace {
static __device__ unsigned long long* buf = nullptr;
static __device__ unsigned long long waves = 0;
static __device__ int kind = 0;                 // NT GEMM: only the instantiation whose tile code equals `kind` stamps (0: all)
__device__ __forceinline__ unsigned wave_index() {
    return ((unsigned)blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
}
__device__ __forceinline__ void stamp(int slot, int my_kind = 0, bool last = false) {
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long* b = buf;
    const unsigned w = wave_index();
    if (b != nullptr && w < waves && (threadIdx.x & 63) == 0 && (kind == 0 || my_kind == kind)) {
        b[(unsigned long long)w * NIW_TRACE_SLOTS + slot] = __builtin_amdgcn_s_memrealtime();
        if (last) b[(unsigned long long)w * NIW_TRACE_SLOTS + 14] = __builtin_amdgcn_s_memtime();
        if (slot == 0) {
            b[(unsigned long long)w * NIW_TRACE_SLOTS + 13] = __builtin_amdgcn_s_memtime();
            // s_getreg_b32: id | offset << 6 | (width - 1) << 11;  HW_REG_HW_ID = 4 (cu_id [11:8], sh_id [12], se_id [15:13]), HW_REG_XCC_ID = 20
            const unsigned hw = __builtin_amdgcn_s_getreg(4 | (31 << 11)), xcc = __builtin_amdgcn_s_getreg(20 | (31 << 11));
            b[(unsigned long long)w * NIW_TRACE_SLOTS + 15] = ((unsigned long long)xcc << 32) | hw;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}
inline int set(unsigned long long* p, unsigned long long n, int k) {
    if (hipMemcpyToSymbol(HIP_SYMBOL(buf), &p, sizeof(p)) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(waves), &n, sizeof(n)) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(kind), &k, sizeof(k)) != hipSuccess) return -1;
    return 0;
}
}  // namespace niw_trace
#define NIW_STAMP(slot) niw_trace::stamp(slot)
#define NIW_STAMP_KIND(slot, k) niw_trace::stamp(slot, k)
#define NIW_STAMP_LAST(slot) niw_trace::stamp(slot, 0, true)
#define NIW_STAMP_KIND_LAST(slot, k) niw_trace::stamp(slot, k, true)
#define NIW_TRACE_SETTER(name) extern "C" int name(unsigned long long* p, unsigned long long n, int k) { return niw_trace::set(p, n, k); }
#else
#define NIW_STAMP(slot) do {} while (0)
#define NIW_STAMP_KIND(slot, k) do {} while (0)
#define NIW_STAMP_LAST(slot) do {} while (0)
#define NIW_STAMP_KIND_LAST(slot, k) do {} while (0)
#define NIW_TRACE_SETTER(name)
#endif
