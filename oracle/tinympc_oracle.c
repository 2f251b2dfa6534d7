/* TEST INFRASTRUCTURE ONLY — see tinympc_oracle.h.
 *
 * CPU restatement of the reference's TinyMPC solve path in plain C:
 *   orc_riccati            <- src/codegen_src/tinympc/tiny_api.cpp:124-190
 *   the ADMM loop + API    <- tinympc_oracle_body.inc (admm.cpp:13-207,
 *                             tiny_api.cpp:21-122,233-267)
 * Built by oracle/Makefile (`make port`) into oracle/libtinympc_oracle.so.
 */
#define _POSIX_C_SOURCE 200809L
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "tinympc_oracle.h"

#define ORC_MAX_DIM 64
#define ORC_MAX_LIN 8 /* rows of linear inequalities per side */

/* ---------- small dense fp64 helpers, column-major ---------- */

/* C(m x n) = A(m x k) * B(k x n) */
static void mm(double *C, const double *A, const double *B, int m, int k, int n) {
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < m; ++i) {
            double acc = 0;
            for (int l = 0; l < k; ++l) acc += A[i + (size_t)l * m] * B[l + (size_t)j * k];
            C[i + (size_t)j * m] = acc;
        }
}

/* T(n x m) = A(m x n)^T */
static void tr(double *T, const double *A, int m, int n) {
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < m; ++i) T[j + (size_t)i * n] = A[i + (size_t)j * m];
}

/* Ainv = A^-1 (n x n) by Gauss-Jordan with partial pivoting.  (The reference's
 * Eigen .inverse() on a dynamic matrix is PartialPivLU; results agree to rounding.) */
static int inv(double *Ainv, const double *A, int n) {
    double *M = (double *)malloc(sizeof(double) * n * n);
    memcpy(M, A, sizeof(double) * n * n);
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) Ainv[i + (size_t)j * n] = (i == j) ? 1.0 : 0.0;
    for (int c = 0; c < n; ++c) {
        int piv = c;
        double best = fabs(M[c + (size_t)c * n]);
        for (int r = c + 1; r < n; ++r)
            if (fabs(M[r + (size_t)c * n]) > best) {
                best = fabs(M[r + (size_t)c * n]);
                piv = r;
            }
        if (best == 0.0) {
            free(M);
            return 1;
        }
        if (piv != c)
            for (int j = 0; j < n; ++j) {
                double t = M[c + (size_t)j * n];
                M[c + (size_t)j * n] = M[piv + (size_t)j * n];
                M[piv + (size_t)j * n] = t;
                t = Ainv[c + (size_t)j * n];
                Ainv[c + (size_t)j * n] = Ainv[piv + (size_t)j * n];
                Ainv[piv + (size_t)j * n] = t;
            }
        double ip = 1.0 / M[c + (size_t)c * n];
        for (int j = 0; j < n; ++j) {
            M[c + (size_t)j * n] *= ip;
            Ainv[c + (size_t)j * n] *= ip;
        }
        for (int r = 0; r < n; ++r) {
            if (r == c) continue;
            double f = M[r + (size_t)c * n];
            if (f == 0.0) continue;
            for (int j = 0; j < n; ++j) {
                M[r + (size_t)j * n] -= f * M[c + (size_t)j * n];
                Ainv[r + (size_t)j * n] -= f * Ainv[c + (size_t)j * n];
            }
        }
    }
    free(M);
    return 0;
}

/* reference: tiny_api.cpp:124-190 (tiny_precompute_and_set_cache).
 * Qd, Rd are the diagonals handed in by tiny_setup (already diag(Q)+rho,
 * tiny_api.cpp:90-91,113); rho is added a second time here (:134-135).
 * P starts at rho*I (:148); at most 1000 iterations; the convergence test
 * max|K - K_prev| < 1e-5 is made BEFORE the shift (:157-165) so the K, P of the
 * breaking iteration are the ones kept. */
static void orc_riccati(const double *A, const double *B, const double *Qd, const double *Rd,
                        double rho, int nx, int nu, double *Kinf, double *Pinf, double *Quu_inv,
                        double *AmBKt) {
    size_t nxx = (size_t)nx * nx, nux = (size_t)nu * nx, nuu = (size_t)nu * nu;
    double *Q1 = (double *)calloc(nxx, sizeof(double));
    double *R1 = (double *)calloc(nuu, sizeof(double));
    double *Ktp1 = (double *)calloc(nux, sizeof(double));
    double *Ptp1 = (double *)calloc(nxx, sizeof(double));
    double *Bt = (double *)malloc(sizeof(double) * nux);
    double *At = (double *)malloc(sizeof(double) * nxx);
    double *BtP = (double *)malloc(sizeof(double) * nux);
    double *S = (double *)malloc(sizeof(double) * nuu);
    double *Si = (double *)malloc(sizeof(double) * nuu);
    double *T1 = (double *)malloc(sizeof(double) * nux);
    double *T2 = (double *)malloc(sizeof(double) * nux);
    double *BK = (double *)malloc(sizeof(double) * nxx);
    double *AmBK = (double *)malloc(sizeof(double) * nxx);
    double *AtP = (double *)malloc(sizeof(double) * nxx);
    double *T3 = (double *)malloc(sizeof(double) * nxx);
    for (int i = 0; i < nx; ++i) {
        Q1[i + (size_t)i * nx] = Qd[i] + rho;
        Ptp1[i + (size_t)i * nx] = rho;
    }
    for (int i = 0; i < nu; ++i) R1[i + (size_t)i * nu] = Rd[i] + rho;
    tr(Bt, B, nx, nu);
    tr(At, A, nx, nx);
    memset(Kinf, 0, sizeof(double) * nux);
    memset(Pinf, 0, sizeof(double) * nxx);
    for (int it = 0; it < 1000; ++it) {
        /* Kinf = (R1 + B'PB)^-1 * B' * P * A, evaluated left to right */
        mm(BtP, Bt, Ptp1, nu, nx, nx);
        mm(S, BtP, B, nu, nx, nu);
        for (size_t i = 0; i < nuu; ++i) S[i] += R1[i];
        inv(Si, S, nu);
        mm(T1, Si, Bt, nu, nu, nx);
        mm(T2, T1, Ptp1, nu, nx, nx);
        mm(Kinf, T2, A, nu, nx, nx);
        /* Pinf = Q1 + A' * P * (A - B K) */
        mm(BK, B, Kinf, nx, nu, nx);
        for (size_t i = 0; i < nxx; ++i) AmBK[i] = A[i] - BK[i];
        mm(AtP, At, Ptp1, nx, nx, nx);
        mm(T3, AtP, AmBK, nx, nx, nx);
        for (size_t i = 0; i < nxx; ++i) Pinf[i] = Q1[i] + T3[i];
        double m = 0;
        for (size_t i = 0; i < nux; ++i) {
            double t = fabs(Kinf[i] - Ktp1[i]);
            if (t > m) m = t;
        }
        if (m < 1e-5) break;
        memcpy(Ktp1, Kinf, sizeof(double) * nux);
        memcpy(Ptp1, Pinf, sizeof(double) * nxx);
    }
    /* Quu_inv = (R1 + B' Pinf B)^-1 ; AmBKt = (A - B Kinf)^T */
    mm(BtP, Bt, Pinf, nu, nx, nx);
    mm(S, BtP, B, nu, nx, nu);
    for (size_t i = 0; i < nuu; ++i) S[i] += R1[i];
    inv(Quu_inv, S, nu);
    mm(BK, B, Kinf, nx, nu, nx);
    for (size_t i = 0; i < nxx; ++i) AmBK[i] = A[i] - BK[i];
    tr(AmBKt, AmBK, nx, nx);
    free(Q1); free(R1); free(Ktp1); free(Ptp1); free(Bt); free(At); free(BtP); free(S);
    free(Si); free(T1); free(T2); free(BK); free(AmBK); free(AtP); free(T3);
}

#define REAL double
#define ORC_PREFIX orc64_
#include "tinympc_oracle_body.inc"
#undef REAL
#undef ORC_PREFIX

#define REAL float
#define ORC_PREFIX orc32_
#include "tinympc_oracle_body.inc"
#undef REAL
#undef ORC_PREFIX
