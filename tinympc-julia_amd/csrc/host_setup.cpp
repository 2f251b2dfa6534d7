// Host fp64 Riccati precompute — semantics of the reference's
// tiny_setup + tiny_precompute_and_set_cache
// (reference: src/codegen_src/tinympc/tiny_api.cpp:90-91,113,124-190):
//   * the stage costs kept for the ADMM linear-cost step are the DIAGONALS
//     Qd = diag(Q) + rho, Rd = diag(R) + rho (off-diagonals of Q, R are dropped);
//   * the Riccati recursion runs on Qd + rho, Rd + rho (rho enters twice);
//   * P_0 = rho I; K = (R1 + B'PB)^-1 B'PA; P' = Q1 + A'P(A - BK); at most 1000
//     sweeps; stop when max|K - K_prev| < 1e-5, keeping the K, P' of that sweep;
//   * Quu_inv = (R1 + B'Pinf B)^-1, AmBKt = (A - B Kinf)'.
#include "host_setup.h"

#include <cmath>
#include <utility>

namespace tmpc {
namespace {

Mat mul(const Mat &X, const Mat &Y) {
    Mat Z(X.r, Y.c);
    for (int j = 0; j < Y.c; ++j)
        for (int l = 0; l < X.c; ++l) {
            const double y = Y(l, j);
            if (y == 0.0) continue;
            for (int i = 0; i < X.r; ++i) Z(i, j) += X(i, l) * y;
        }
    return Z;
}

Mat transpose(const Mat &X) {
    Mat T(X.c, X.r);
    for (int j = 0; j < X.c; ++j)
        for (int i = 0; i < X.r; ++i) T(j, i) = X(i, j);
    return T;
}

Mat add(const Mat &X, const Mat &Y, double sy = 1.0) {
    Mat Z(X.r, X.c);
    for (size_t i = 0; i < Z.a.size(); ++i) Z.a[i] = X.a[i] + sy * Y.a[i];
    return Z;
}

// Solve S X = Rhs for X by LU with partial pivoting (S is nu x nu, tiny).
bool lu_solve(Mat S, Mat Rhs, Mat &X) {
    const int n = S.r;
    for (int c = 0; c < n; ++c) {
        int piv = c;
        for (int i = c + 1; i < n; ++i)
            if (std::fabs(S(i, c)) > std::fabs(S(piv, c))) piv = i;
        if (S(piv, c) == 0.0) return false;
        if (piv != c) {
            for (int j = 0; j < n; ++j) std::swap(S(c, j), S(piv, j));
            for (int j = 0; j < Rhs.c; ++j) std::swap(Rhs(c, j), Rhs(piv, j));
        }
        for (int i = c + 1; i < n; ++i) {
            const double f = S(i, c) / S(c, c);
            if (f == 0.0) continue;
            for (int j = c; j < n; ++j) S(i, j) -= f * S(c, j);
            for (int j = 0; j < Rhs.c; ++j) Rhs(i, j) -= f * Rhs(c, j);
        }
    }
    X = Mat(n, Rhs.c);
    for (int j = 0; j < Rhs.c; ++j)
        for (int i = n - 1; i >= 0; --i) {
            double acc = Rhs(i, j);
            for (int l = i + 1; l < n; ++l) acc -= S(i, l) * X(l, j);
            X(i, j) = acc / S(i, i);
        }
    return true;
}

}  // namespace

int precompute_cache(const Mat &A, const Mat &B, const Mat &Q, const Mat &R, double rho, Cache &out) {
    const int nx = A.r, nu = B.c;
    out.rho = rho;
    out.Qd.assign(nx, 0.0);
    out.Rd.assign(nu, 0.0);
    for (int i = 0; i < nx; ++i) out.Qd[i] = Q(i, i) + rho;
    for (int i = 0; i < nu; ++i) out.Rd[i] = R(i, i) + rho;
    Mat Q1(nx, nx), R1(nu, nu), P(nx, nx), Kprev(nu, nx);
    for (int i = 0; i < nx; ++i) {
        Q1(i, i) = out.Qd[i] + rho;
        P(i, i) = rho;
    }
    for (int i = 0; i < nu; ++i) R1(i, i) = out.Rd[i] + rho;
    const Mat Bt = transpose(B), At = transpose(A);
    Mat K(nu, nx), Pn(nx, nx);
    int it = 0;
    for (; it < 1000; ++it) {
        const Mat BtP = mul(Bt, P);
        if (!lu_solve(add(R1, mul(BtP, B)), mul(BtP, A), K)) return 1;
        Pn = add(Q1, mul(mul(At, P), add(A, mul(B, K), -1.0)));
        double dk = 0.0;
        for (size_t i = 0; i < K.a.size(); ++i) dk = std::fmax(dk, std::fabs(K.a[i] - Kprev.a[i]));
        if (dk < 1e-5) {
            ++it;
            break;
        }
        Kprev = K;
        P = Pn;
    }
    out.riccati_iters = it;
    out.Kinf = K;
    out.Pinf = Pn;
    Mat I(nu, nu);
    for (int i = 0; i < nu; ++i) I(i, i) = 1.0;
    if (!lu_solve(add(R1, mul(mul(Bt, Pn), B)), I, out.Quu_inv)) return 1;
    out.AmBKt = transpose(add(A, mul(B, K), -1.0));
    return 0;
}

namespace {

// The reference host's LQR for the sensitivities (TinyMPC.jl:326-352, solve_lqr) — NOT the recursion of setup():
// full Q, R with rho added ONCE, P_0 = Q + rho I, K = (R + rho I + B'PB + 1e-8 I) \ B'PA, at most 5000 sweeps,
// stop when ||K - K_prev||_F < 1e-10 (from the second sweep on).
bool lqr_for_sensitivity(const Mat &A, const Mat &B, const Mat &Q, const Mat &R, double rho, Mat &K, Mat &P, Mat &C1,
                         Mat &C2) {
    const int nx = A.r, nu = B.c;
    Mat Qr = Q, Rr = R, Rreg;
    for (int i = 0; i < nx; ++i) Qr(i, i) += rho;
    for (int i = 0; i < nu; ++i) Rr(i, i) += rho;
    Rreg = Rr;
    for (int i = 0; i < nu; ++i) Rreg(i, i) += 1e-8;
    const Mat Bt = transpose(B), At = transpose(A);
    P = Qr;
    K = Mat(nu, nx);
    for (int it = 1; it <= 5000; ++it) {
        const Mat Kprev = K, BtP = mul(Bt, P);
        if (!lu_solve(add(Rreg, mul(BtP, B)), mul(BtP, A), K)) return false;
        P = add(Qr, mul(mul(At, P), add(A, mul(B, K), -1.0)));
        double n2 = 0.0;
        for (size_t i = 0; i < K.a.size(); ++i) n2 += (K.a[i] - Kprev.a[i]) * (K.a[i] - Kprev.a[i]);
        if (it > 1 && std::sqrt(n2) < 1e-10) break;
    }
    Mat I(nu, nu);
    for (int i = 0; i < nu; ++i) I(i, i) = 1.0;
    if (!lu_solve(add(Rr, mul(mul(Bt, P), B)), I, C1)) return false;
    C2 = transpose(add(A, mul(B, K), -1.0));
    return true;
}

}  // namespace

// Forward differences with h = 1e-6 (TinyMPC.jl:301-323, compute_sensitivity_autograd).
int compute_sensitivity(const Mat &A, const Mat &B, const Mat &Q, const Mat &R, double rho, Mat &dK, Mat &dP, Mat &dC1,
                        Mat &dC2) {
    const double h = 1e-6;
    Mat K0, P0, C10, C20, K1, P1, C11, C21;
    if (!lqr_for_sensitivity(A, B, Q, R, rho, K0, P0, C10, C20) ||
        !lqr_for_sensitivity(A, B, Q, R, rho + h, K1, P1, C11, C21))
        return 1;
    auto diff = [h](const Mat &X1, const Mat &X0) {
        Mat D(X0.r, X0.c);
        for (size_t i = 0; i < D.a.size(); ++i) D.a[i] = (X1.a[i] - X0.a[i]) / h;
        return D;
    };
    dK = diff(K1, K0);
    dP = diff(P1, P0);
    dC1 = diff(C11, C10);
    dC2 = diff(C21, C20);
    return 0;
}

}  // namespace tmpc
