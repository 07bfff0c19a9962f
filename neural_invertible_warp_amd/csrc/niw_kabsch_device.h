// Device side of the Kabsch rotation (see niw_kabsch.hip for the algorithm); shared with the fused alignment loss (niw_align.hip).
#pragma once
#include "niw_common.h"

namespace niw {

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

// M (double) -> R = U diag(1,1,d) V^T; also returns U' = U diag(1,1,d), V and s' = (s0, s1, d s2) for the backward
__device__ inline void kabsch_rotation(const double (&M)[3][3], double (&Rout)[3][3], double (&U)[3][3], double (&Vs)[3][3], double (&s)[3]) {
    double A[3][3], V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) A[i][j] = M[0][i] * M[0][j] + M[1][i] * M[1][j] + M[2][i] * M[2][j];
    // cyclic Jacobi converges quadratically: 4-6 sweeps bring a 3 x 3 matrix to fp64 round-off; stop there (each rotation is a
    // double-precision divide and two square roots on one lane; 12 unconditional sweeps were 15 us of a training step)
    for (int sweep = 0; sweep < 12; ++sweep) {
        const double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
        if (off <= 1e-34 * (fabs(A[0][0]) + fabs(A[1][1]) + fabs(A[2][2]))) break;
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
        for (int j = 0; j < 3; ++j) Rout[i][j] = U[i][0] * Vs[j][0] + U[i][1] * Vs[j][1] + U[i][2] * Vs[j][2];
}

}  // namespace niw
