// Rotation of the orthogonal Procrustes / Kabsch problem for batches of 3x3 moment matrices, forward and backward:
//     R = U diag(1, 1, det(U V^T)) V^T     for  M = U S V^T
// (the alignment loss of the INN models: `roma.rigid_points_registration` at reference model/nerf_inn_llff.py:569 and
// model/pose_models/inn.py:100; roma is not vendored in the reference -- this follows the published algorithm).
// One thread per matrix, fp64 inside.  torch.linalg.svd on the device blocks the host for ~2 ms per call (status
// check), which with 18 views made the 2048-ray configurations host-bound; these two launches never synchronise.
//
// forward:  A = M^T M, cyclic Jacobi eigen-decomposition -> V, s_i = sqrt(lambda_i) (descending), u_i = M v_i / s_i
//           (completed by cross products when a singular value vanishes), d = det(U) det(V); outputs R and, for the
//           backward, U' = U diag(1,1,d), V, s' = (s_0, s_1, d s_2).
// backward: with A = U'^T G V (G = dL/dR):  dL/dM = U' [ (A - A^T)_ij / (s'_i + s'_j) ] V^T    (differential of the
//           orthogonal polar factor; unlike the generic SVD backward it has no 1/(s_i^2 - s_j^2) terms).
#include "niw_common.h"
#include "niw_kabsch_device.h"

namespace {

__global__ void kabsch_fwd_kernel(const float* __restrict__ Min, int B, float* __restrict__ R, float* __restrict__ Us,
                                  float* __restrict__ Vo, float* __restrict__ So) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double M[3][3], Rd[3][3], U[3][3], Vs[3][3], s[3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) M[i][j] = Min[b * 9 + i * 3 + j];
    niw::kabsch_rotation(M, Rd, U, Vs, s);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            R[b * 9 + i * 3 + j] = (float)Rd[i][j];
            Us[b * 9 + i * 3 + j] = (float)U[i][j];
            Vo[b * 9 + i * 3 + j] = (float)Vs[i][j];
        }
    for (int c = 0; c < 3; ++c) So[b * 3 + c] = (float)s[c];
}

__global__ void kabsch_bwd_kernel(const float* __restrict__ Us, const float* __restrict__ Vi, const float* __restrict__ S,
                                  const float* __restrict__ G, int B, float* __restrict__ dM) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double U[3][3], V[3][3], g[3][3], s[3], A[3][3], X[3][3], T[3][3];
    for (int i = 0; i < 3; ++i) {
        s[i] = S[b * 3 + i];
        for (int j = 0; j < 3; ++j) { U[i][j] = Us[b * 9 + i * 3 + j]; V[i][j] = Vi[b * 9 + i * 3 + j]; g[i][j] = G[b * 9 + i * 3 + j]; }
    }
    for (int i = 0; i < 3; ++i)                   // T = U^T G
        for (int j = 0; j < 3; ++j) T[i][j] = U[0][i] * g[0][j] + U[1][i] * g[1][j] + U[2][i] * g[2][j];
    for (int i = 0; i < 3; ++i)                   // A = T V
        for (int j = 0; j < 3; ++j) A[i][j] = T[i][0] * V[0][j] + T[i][1] * V[1][j] + T[i][2] * V[2][j];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            const double den = s[i] + s[j];
            X[i][j] = (i == j || fabs(den) < 1e-30) ? 0.0 : (A[i][j] - A[j][i]) / den;
        }
    for (int i = 0; i < 3; ++i)                   // T = U X
        for (int j = 0; j < 3; ++j) T[i][j] = U[i][0] * X[0][j] + U[i][1] * X[1][j] + U[i][2] * X[2][j];
    for (int i = 0; i < 3; ++i)                   // dM = T V^T
        for (int j = 0; j < 3; ++j) dM[b * 9 + i * 3 + j] = (float)(T[i][0] * V[j][0] + T[i][1] * V[j][1] + T[i][2] * V[j][2]);
}

}  // namespace

extern "C" int niw_kabsch_rotation_fwd(const float* M, int n, float* R, float* Us, float* V, float* S, niw_stream_t stream) {
    NIW_REQUIRE(M && R && Us && V && S, "niw_kabsch_rotation_fwd: null pointer");
    NIW_REQUIRE(n > 0, "niw_kabsch_rotation_fwd: empty batch");
    kabsch_fwd_kernel<<<(n + 63) / 64, 64, 0, (hipStream_t)stream>>>(M, n, R, Us, V, S);
    NIW_LAUNCH_CHECK("niw_kabsch_rotation_fwd");
    return NIW_OK;
}

extern "C" int niw_kabsch_rotation_bwd(const float* Us, const float* V, const float* S, const float* dR, int n, float* dM,
                                       niw_stream_t stream) {
    NIW_REQUIRE(Us && V && S && dR && dM, "niw_kabsch_rotation_bwd: null pointer");
    NIW_REQUIRE(n > 0, "niw_kabsch_rotation_bwd: empty batch");
    kabsch_bwd_kernel<<<(n + 63) / 64, 64, 0, (hipStream_t)stream>>>(Us, V, S, dR, n, dM);
    NIW_LAUNCH_CHECK("niw_kabsch_rotation_bwd");
    return NIW_OK;
}
