// Diagnostic: what an event record BETWEEN two dependent kernels of one stream costs (the fork of a second stream inside niw_train_step),
// and whether the kernel's own completion signal (hipExtLaunchKernelGGL's stopEvent) is cheaper than a hipEventRecord marker.
//   hipcc --offload-arch=gfx950 -O3 tools/event_bubble.hip -o scratch/event_bubble && scratch/event_bubble
// Prints microseconds per PAIR of ~4 us kernels: (a) back to back, (b) hipEventRecord between them, (c) first kernel launched with a
// stopEvent, (d) = (b) + a second stream that waits for the event and runs a kernel, (e) = (c) + the same, (f) a hipStreamWaitEvent on an
// event that completed long ago between the two kernels (the join of a side stream that finished early).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void spin_kernel(float* p, int iters) {
    float v = p[threadIdx.x];
    for (int i = 0; i < iters; ++i) v = v * 1.000001f + 1e-7f;
    p[threadIdx.x] = v;
}

int main() {
    float *a, *b;
    CK(hipMalloc(&a, 4096)); CK(hipMalloc(&b, 4096));
    CK(hipMemset(a, 0, 4096)); CK(hipMemset(b, 0, 4096));
    hipStream_t s, x;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
    hipEvent_t ev, old, t0, t1;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&old, hipEventDisableTiming));
    CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    const int iters = 3000, reps = 200;      // ~4 us per kernel
    CK(hipEventRecord(old, x)); CK(hipStreamSynchronize(x));
    auto time = [&](const char* name, auto&& body) {
        for (int w = 0; w < 20; ++w) body();
        CK(hipStreamSynchronize(s)); CK(hipStreamSynchronize(x));
        CK(hipEventRecord(t0, s));
        for (int r = 0; r < reps; ++r) body();
        CK(hipEventRecord(t1, s));
        CK(hipEventSynchronize(t1)); CK(hipStreamSynchronize(x));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, t0, t1));
        printf("{\"case\": \"%s\", \"us_per_pair\": %.2f}\n", name, ms * 1e3f / reps);
    };
    time("a_back_to_back", [&] { spin_kernel<<<1, 64, 0, s>>>(a, iters); spin_kernel<<<1, 64, 0, s>>>(a, iters); });
    time("b_event_record_between", [&] { spin_kernel<<<1, 64, 0, s>>>(a, iters); CK(hipEventRecord(ev, s)); spin_kernel<<<1, 64, 0, s>>>(a, iters); });
    time("c_stop_event_of_the_first_kernel", [&] {
        hipExtLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, nullptr, ev, 0, a, iters);
        spin_kernel<<<1, 64, 0, s>>>(a, iters);
    });
    time("d_event_record_and_forked_stream", [&] {
        spin_kernel<<<1, 64, 0, s>>>(a, iters); CK(hipEventRecord(ev, s)); CK(hipStreamWaitEvent(x, ev, 0));
        spin_kernel<<<1, 64, 0, x>>>(b, iters); spin_kernel<<<1, 64, 0, s>>>(a, iters);
    });
    time("e_stop_event_and_forked_stream", [&] {
        hipExtLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, nullptr, ev, 0, a, iters);
        CK(hipStreamWaitEvent(x, ev, 0));
        spin_kernel<<<1, 64, 0, x>>>(b, iters); spin_kernel<<<1, 64, 0, s>>>(a, iters);
    });
    time("f_wait_for_an_old_event_between", [&] { spin_kernel<<<1, 64, 0, s>>>(a, iters); CK(hipStreamWaitEvent(s, old, 0)); spin_kernel<<<1, 64, 0, s>>>(a, iters); });
    return 0;
}
