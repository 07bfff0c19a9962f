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

namespace {

__device__ __forceinline__ void jacobi_rotate(double (&A)[3][3], double (&V)[3][3], int p, int q) {
    if (fabs(A[p][q]) < 1e-300) return;
    const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
    const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
    const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
    for (int k = 0; k < 3; ++k) {                 // A <- A J
        const double akp = A[k][p], akq = A[k][q];
        A[k][p] = c * akp - s * akq;
        A[k][q] = s * akp + c * akq;
    }
    for (int k = 0; k < 3; ++k) {                 // A <- J^T A
        const double apk = A[p][k], aqk = A[q][k];
        A[p][k] = c * apk - s * aqk;
        A[q][k] = s * apk + c * aqk;
    }
    for (int k = 0; k < 3; ++k) {                 // V <- V J
        const double vkp = V[k][p], vkq = V[k][q];
        V[k][p] = c * vkp - s * vkq;
        V[k][q] = s * vkp + c * vkq;
    }
}

__device__ __forceinline__ double det3(const double (&X)[3][3]) {
    return X[0][0] * (X[1][1] * X[2][2] - X[1][2] * X[2][1]) - X[0][1] * (X[1][0] * X[2][2] - X[1][2] * X[2][0]) +
           X[0][2] * (X[1][0] * X[2][1] - X[1][1] * X[2][0]);
}

__global__ void kabsch_fwd_kernel(const float* __restrict__ Min, int B, float* __restrict__ R, float* __restrict__ Us,
                                  float* __restrict__ Vo, float* __restrict__ So) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double M[3][3], A[3][3], V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) M[i][j] = Min[b * 9 + i * 3 + j];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) A[i][j] = M[0][i] * M[0][j] + M[1][i] * M[1][j] + M[2][i] * M[2][j];
    for (int sweep = 0; sweep < 12; ++sweep) {
        jacobi_rotate(A, V, 0, 1);
        jacobi_rotate(A, V, 0, 2);
        jacobi_rotate(A, V, 1, 2);
    }
    // sort eigenpairs by descending eigenvalue
    int ord[3] = {0, 1, 2};
    double lam[3] = {A[0][0], A[1][1], A[2][2]};
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2 - i; ++j)
            if (lam[ord[j]] < lam[ord[j + 1]]) { const int t = ord[j]; ord[j] = ord[j + 1]; ord[j + 1] = t; }
    double Vs[3][3], U[3][3], s[3];
    for (int c = 0; c < 3; ++c) {
        s[c] = sqrt(fmax(lam[ord[c]], 0.0));
        for (int k = 0; k < 3; ++k) Vs[k][c] = V[k][ord[c]];
    }
    const double tol = 1e-12 * fmax(s[0], 1e-300);
    for (int c = 0; c < 3; ++c) {
        if (s[c] > tol) {
            for (int k = 0; k < 3; ++k) U[k][c] = (M[k][0] * Vs[0][c] + M[k][1] * Vs[1][c] + M[k][2] * Vs[2][c]) / s[c];
        } else if (c == 2) {                      // rank 2: complete with the cross product
            U[0][2] = U[1][0] * U[2][1] - U[2][0] * U[1][1];
            U[1][2] = U[2][0] * U[0][1] - U[0][0] * U[2][1];
            U[2][2] = U[0][0] * U[1][1] - U[1][0] * U[0][1];
        } else if (c == 1) {                      // rank 1: any unit vector orthogonal to u_0
            const int k0 = fabs(U[0][0]) < fabs(U[1][0]) ? (fabs(U[0][0]) < fabs(U[2][0]) ? 0 : 2) : (fabs(U[1][0]) < fabs(U[2][0]) ? 1 : 2);
            double e[3] = {0, 0, 0};
            e[k0] = 1.0;
            const double dot = U[k0][0];
            double w[3] = {e[0] - dot * U[0][0], e[1] - dot * U[1][0], e[2] - dot * U[2][0]};
            const double n = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
            for (int k = 0; k < 3; ++k) U[k][1] = w[k] / n;
        } else {                                  // zero matrix
            U[0][0] = 1; U[1][0] = 0; U[2][0] = 0;
        }
    }
    const double d = det3(U) * det3(Vs) < 0 ? -1.0 : 1.0;
    for (int k = 0; k < 3; ++k) U[k][2] *= d;
    s[2] *= d;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            R[b * 9 + i * 3 + j] = (float)(U[i][0] * Vs[j][0] + U[i][1] * Vs[j][1] + U[i][2] * Vs[j][2]);
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
