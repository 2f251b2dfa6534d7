// Host-side (fp64) part of setup(): infinite-horizon Riccati precompute.
// Stays on the CPU in double precision: an fp32 Riccati alone breaks the 1e-5
// parity target, and it runs once per problem family.
#pragma once
#include <vector>

namespace tmpc {

// Column-major dense fp64 matrix, just enough for the precompute.
struct Mat {
    int r = 0, c = 0;
    std::vector<double> a;
    Mat() = default;
    Mat(int r_, int c_) : r(r_), c(c_), a((size_t)r_ * c_, 0.0) {}
    Mat(int r_, int c_, const double *p) : r(r_), c(c_), a(p, p + (size_t)r_ * c_) {}
    double &operator()(int i, int j) { return a[i + (size_t)j * r]; }
    double operator()(int i, int j) const { return a[i + (size_t)j * r]; }
};

struct Cache {
    double rho = 0;
    Mat Kinf, Pinf, Quu_inv, AmBKt;  // nu x nx, nx x nx, nu x nu, nx x nx
    std::vector<double> Qd, Rd;      // diag(Q)+rho, diag(R)+rho (what the linear-cost step uses)
    int riccati_iters = 0;
};

// Returns 0 on success, 1 if (R1 + B'PB) is singular.
int precompute_cache(const Mat &A, const Mat &B, const Mat &Q, const Mat &R, double rho, Cache &out);

// d(Kinf, Pinf, Quu_inv, AmBKt)/d rho for adaptive rho, as the reference's host computes them (TinyMPC.jl:301-352).
// Returns 0 on success, 1 if a linear solve failed.
int compute_sensitivity(const Mat &A, const Mat &B, const Mat &Q, const Mat &R, double rho, Mat &dK, Mat &dP, Mat &dC1,
                        Mat &dC2);

}  // namespace tmpc
