// Global-alignment loss of the INN models, fused (reference model/nerf_inn_llff.py:563-572, model/nerf_inn_dtu.py:410-414,
// model/pose_models/inn.py:96-102):
//
//     R, t = rigid registration of the warped points x_i (grid_3D ; centre) onto the un-warped ones y_i        [Kabsch]
//     loss = mean_i | x_i - R^T (y_i - t) |^2                      (= MSE(target, cam2world(source, [R|t])))
//
// As torch glue this was ~40 launches per step (means, centring, 3x3 matmuls through hipBLASLt, cat / neg / add kernels):
// a third of the non-MLP time of the 2048-ray configurations.  Here it is three:
//
//   niw_align_moments   per view: n, sum x, sum y, sum y x^T accumulated in fp64 (one workgroup per view).  Sixteen doubles
//                       per view -- under ray sharding these are what the ranks all-reduce, so that every rank registers
//                       the GLOBAL point set.
//   niw_align_solve     per view: centred moment matrix M = sum y x^T - n ybar xbar^T, R = U diag(1,1,det) V^T (the fp64
//                       Jacobi solver of niw_kabsch.hip), t = ybar - R xbar  ->  poses [B,3,4].
//   niw_align_loss      residuals, loss and d loss / d x_i = 2 e_i / n_norm in one pass (one workgroup, fixed-order
//                       reduction: bit-reproducible).
//
// Gradient.  R, t MINIMISE sum_i |R x_i + t - y_i|^2 = sum_i |x_i - R^T (y_i - t)|^2 over SO(3) x R^3, so the loss is
// stationary in (R, t) and, by the envelope theorem, d loss / d x through (R, t) vanishes identically: the total derivative
// is the direct term 2 e_i / n.  (Checked against autograd through the SVD in fp64: the two agree to 1e-17,
// tests/test_gpu_step_ops.py::test_fused_alignment_loss_vs_oracle_autograd_through_svd and tests/test_parallel_gloo.py.)  The reference differentiates through roma's SVD and obtains the same numbers plus roundoff; the
// DTU variant detaches the pose explicitly.
#include "niw_common.h"
#include "niw_kabsch_device.h"

namespace {

constexpr int kMom = 16;      // n, sum x[3], sum y[3], sum y x^T [9] (row-major: y index major)

__device__ __forceinline__ void solve_view(const double* m, float* __restrict__ pose);

// poses != NULL: the workgroup that has just summed a view's moments also solves the view's registration (niw_train_step: a view's
// moments are complete in its own workgroup -- no reduction over ranks in between -- so the separate solve launch is saved)
__global__ __launch_bounds__(256) void align_moments_kernel(const float* __restrict__ x, const float* __restrict__ y, long long N,
                                                            double* __restrict__ mom, float* __restrict__ poses) {
    __shared__ double red[4][kMom];
    __shared__ double total[kMom];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* xb = x + (long long)b * N * 3;
    const float* yb = y + (long long)b * N * 3;
    double acc[kMom];
#pragma unroll
    for (int k = 0; k < kMom; ++k) acc[k] = 0.0;
    for (long long i = tid; i < N; i += 256) {
        const double xv[3] = {xb[i * 3], xb[i * 3 + 1], xb[i * 3 + 2]};
        const double yv[3] = {yb[i * 3], yb[i * 3 + 1], yb[i * 3 + 2]};
        acc[0] += 1.0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            acc[1 + c] += xv[c];
            acc[4 + c] += yv[c];
#pragma unroll
            for (int d = 0; d < 3; ++d) acc[7 + 3 * c + d] += yv[c] * xv[d];
        }
    }
#pragma unroll
    for (int k = 0; k < kMom; ++k) {
        double v = acc[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (tid < kMom) {
        const double t = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
        mom[b * kMom + tid] = t;
        total[tid] = t;
    }
    if (poses) {
        __syncthreads();
        if (tid == 0) solve_view(total, poses + b * 12);
    }
}

__device__ __forceinline__ void solve_view(const double* m, float* __restrict__ pose) {
    const double n = m[0] > 0.0 ? m[0] : 1.0;
    const double xm[3] = {m[1] / n, m[2] / n, m[3] / n}, ym[3] = {m[4] / n, m[5] / n, m[6] / n};
    double M[3][3], R[3][3], U[3][3], V[3][3], s[3];
    for (int c = 0; c < 3; ++c)
        for (int d = 0; d < 3; ++d) M[c][d] = m[7 + 3 * c + d] - n * ym[c] * xm[d];
    niw::kabsch_rotation(M, R, U, V, s);
    for (int c = 0; c < 3; ++c) {
        for (int d = 0; d < 3; ++d) pose[c * 4 + d] = (float)R[c][d];
        pose[c * 4 + 3] = (float)(ym[c] - (R[c][0] * xm[0] + R[c][1] * xm[1] + R[c][2] * xm[2]));
    }
}

__global__ void align_solve_kernel(const double* __restrict__ mom, int B, float* __restrict__ poses) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    solve_view(mom + b * kMom, poses + b * 12);
}

// one workgroup of 1024 threads over all B*N points, views in order
__global__ __launch_bounds__(1024) void align_loss_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                          const float* __restrict__ poses, int B, long long N, double inv_norm,
                                                          float* __restrict__ loss, float* __restrict__ d_x) {
    __shared__ double red[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long total = (long long)B * N;
    const float g = (float)(2.0 * inv_norm);
    double acc = 0.0;
    for (long long i = tid; i < total; i += 1024) {
        const float* P = poses + (i / N) * 12;
        const float yc[3] = {y[i * 3] - P[3], y[i * 3 + 1] - P[7], y[i * 3 + 2] - P[11]};
        float sq = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            // (R^T (y - t))_c = sum_k R[k][c] (y - t)_k   (cam2world of a world-to-camera pose, camera.py:343-346)
            const float e = x[i * 3 + c] - (P[c] * yc[0] + P[4 + c] * yc[1] + P[8 + c] * yc[2]);
            sq += e * e;
            if (d_x) d_x[i * 3 + c] = g * e;
        }
        acc += sq;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int w = 0; w < 16; ++w) t += red[w];
        loss[0] = (float)(t * inv_norm);
    }
}

}  // namespace

extern "C" int niw_align_moments(const float* target, const float* source, int n_views, int64_t n_points, double* moments,
                                 niw_stream_t stream) {
    NIW_REQUIRE(target && source && moments, "niw_align_moments: null pointer");
    NIW_REQUIRE(n_views > 0 && n_points > 0, "niw_align_moments: empty input (views=%d, points=%lld)", n_views, (long long)n_points);
    align_moments_kernel<<<n_views, 256, 0, (hipStream_t)stream>>>(target, source, n_points, moments, nullptr);
    NIW_LAUNCH_CHECK("niw_align_moments");
    return NIW_OK;
}

// niw_align_moments + niw_align_solve of the same views in ONE launch (niw_step.hip)
int niw_launch_align_register(const float* target, const float* source, int n_views, int64_t n_points, double* moments, float* poses, hipStream_t st) {
    NIW_REQUIRE(target && source && moments && poses, "niw_train_step (registration): null pointer");
    align_moments_kernel<<<n_views, 256, 0, st>>>(target, source, n_points, moments, poses);
    NIW_LAUNCH_CHECK("niw_train_step (registration)");
    return NIW_OK;
}

extern "C" int niw_align_solve(const double* moments, int n_views, float* poses, niw_stream_t stream) {
    NIW_REQUIRE(moments && poses, "niw_align_solve: null pointer");
    NIW_REQUIRE(n_views > 0, "niw_align_solve: empty batch");
    align_solve_kernel<<<(n_views + 63) / 64, 64, 0, (hipStream_t)stream>>>(moments, n_views, poses);
    NIW_LAUNCH_CHECK("niw_align_solve");
    return NIW_OK;
}

extern "C" int niw_align_loss(const float* target, const float* source, const float* poses, int n_views, int64_t n_points,
                              double n_norm, float* loss, float* d_target, niw_stream_t stream) {
    NIW_REQUIRE(target && source && poses && loss, "niw_align_loss: null pointer");
    NIW_REQUIRE(n_views > 0 && n_points > 0 && n_norm > 0, "niw_align_loss: empty input or non-positive normaliser");
    align_loss_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(target, source, poses, n_views, n_points, 1.0 / n_norm, loss, d_target);
    NIW_LAUNCH_CHECK("niw_align_loss");
    return NIW_OK;
}
