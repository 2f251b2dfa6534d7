// TEST INFRASTRUCTURE ONLY — not part of the product path.
//
// Driver around the reference's vendored TinyMPC snapshot, compiled from the
// sources where they lie under /root/reference (see oracle/Makefile, target
// `_ref`).  No reference source is copied into this repo: this file only calls
// the snapshot's public API
//     tiny_setup / tiny_set_x0 / tiny_set_x_ref / tiny_set_u_ref / tiny_solve
//     (reference: src/codegen_src/tinympc/tiny_api.hpp:10-41)
// and reads/writes the public TinySolver structs
//     (reference: src/codegen_src/tinympc/types.hpp:32-146).
//
// The reference's own shim (src/bindings.cpp) cannot be the driver: it is
// written against a newer TinyMPC API (fdyn, SOC, linear constraints) that is
// absent from /root/reference, so this driver mirrors
// what bindings.cpp + TinyMPC.jl would set on the snapshot's structs:
//   * setup() pushes settings with every en_* flag false   (TinyMPC.jl:89-104)
//   * set_bound_constraints() stores the bounds and enables both bound flags
//                                                      (bindings.cpp:378-411)
//
// Exposes a plain C handle API (`ref_*`) consumed through ctypes by
// oracle/make_golden.py, tests/ and bench.py's cpu_baseline leg.
#include <atomic>
#include <chrono>
#include <cstring>
#include <iostream>
#include <thread>
#include <vector>

#include "tiny_api.hpp"

namespace {

struct RefSolver {
    TinySolver *s = nullptr;
    int nx = 0, nu = 0, N = 0;
};

tinyMatrix map_cm(const double *p, int r, int c) {
    return Eigen::Map<const Eigen::MatrixXd>(p, r, c);
}

void silence_stdout_once() {
    // admm.cpp:190 prints unconditionally on convergence; drop it.
    static bool done = false;
    if (!done) {
        std::cout.rdbuf(nullptr);
        done = true;
    }
}

void free_solver(TinySolver *s) {
    if (!s) return;
    delete s->solution;
    delete s->cache;
    delete s->settings;
    delete s->work;
    delete s;
}

// Zero the 12 trajectory matrices (the snapshot has no reset API; this is the
// state tiny_setup leaves behind, tiny_api.cpp:73-88).  x(:,0) is preserved.
void cold_reset(TinySolver *s) {
    TinyWorkspace *w = s->work;
    tinyVector x0 = w->x.col(0);
    w->x.setZero();
    w->u.setZero();
    w->q.setZero();
    w->r.setZero();
    w->p.setZero();
    w->d.setZero();
    w->v.setZero();
    w->vnew.setZero();
    w->z.setZero();
    w->znew.setZero();
    w->g.setZero();
    w->y.setZero();
    w->x.col(0) = x0;
    w->primal_residual_state = 0;
    w->primal_residual_input = 0;
    w->dual_residual_state = 0;
    w->dual_residual_input = 0;
}

}  // namespace

extern "C" {

// A,B,Q,R column-major fp64.  Bounds start at +-1e17 (inactive) with both
// en_*_bound flags false, as TinyMPC.jl's setup leaves them.
void *ref_create(const double *A, const double *B, const double *Q, const double *R,
                 double rho, int nx, int nu, int N) {
    silence_stdout_once();
    RefSolver *h = new RefSolver();
    h->nx = nx;
    h->nu = nu;
    h->N = N;
    tinyMatrix xmin = tinyMatrix::Constant(nx, N, -1e17);
    tinyMatrix xmax = tinyMatrix::Constant(nx, N, 1e17);
    tinyMatrix umin = tinyMatrix::Constant(nu, N - 1, -1e17);
    tinyMatrix umax = tinyMatrix::Constant(nu, N - 1, 1e17);
    int st = tiny_setup(&h->s, map_cm(A, nx, nx), map_cm(B, nx, nu), map_cm(Q, nx, nx),
                        map_cm(R, nu, nu), rho, nx, nu, N, xmin, xmax, umin, umax, 0);
    if (st != 0) {
        free_solver(h->s);
        delete h;
        return nullptr;
    }
    // TinyMPC.jl:57-61,89-104 effective defaults after setup()
    h->s->settings->abs_pri_tol = 1e-3;
    h->s->settings->abs_dua_tol = 1e-3;
    h->s->settings->max_iter = 100;
    h->s->settings->check_termination = 1;
    h->s->settings->en_state_bound = 0;
    h->s->settings->en_input_bound = 0;
    h->s->settings->adaptive_rho = 0;
    return h;
}

void ref_destroy(void *hp) {
    RefSolver *h = static_cast<RefSolver *>(hp);
    if (!h) return;
    free_solver(h->s);
    delete h;
}

void ref_update_settings(void *hp, double abs_pri_tol, double abs_dua_tol, int max_iter,
                         int check_termination, int en_state_bound, int en_input_bound) {
    RefSolver *h = static_cast<RefSolver *>(hp);
    tiny_update_settings(h->s->settings, abs_pri_tol, abs_dua_tol, max_iter, check_termination,
                         en_state_bound, en_input_bound);
}

// bindings.cpp:378-411 semantics on the snapshot's workspace fields.
void ref_set_bound_constraints(void *hp, const double *xmin, const double *xmax,
                               const double *umin, const double *umax) {
    RefSolver *h = static_cast<RefSolver *>(hp);
    h->s->work->x_min = map_cm(xmin, h->nx, h->N);
    h->s->work->x_max = map_cm(xmax, h->nx, h->N);
    h->s->work->u_min = map_cm(umin, h->nu, h->N - 1);
    h->s->work->u_max = map_cm(umax, h->nu, h->N - 1);
    h->s->settings->en_state_bound = 1;
    h->s->settings->en_input_bound = 1;
}

void ref_set_x0(void *hp, const double *x0) {
    RefSolver *h = static_cast<RefSolver *>(hp);
    tiny_set_x0(h->s, Eigen::Map<const Eigen::VectorXd>(x0, h->nx));
}
void ref_set_x_ref(void *hp, const double *xr) {
    RefSolver *h = static_cast<RefSolver *>(hp);
    tiny_set_x_ref(h->s, map_cm(xr, h->nx, h->N));
}
void ref_set_u_ref(void *hp, const double *ur) {
    RefSolver *h = static_cast<RefSolver *>(hp);
    tiny_set_u_ref(h->s, map_cm(ur, h->nu, h->N - 1));
}

void ref_reset(void *hp) { cold_reset(static_cast<RefSolver *>(hp)->s); }

int ref_solve(void *hp) { return tiny_solve(static_cast<RefSolver *>(hp)->s); }

void ref_get_solution(void *hp, double *x, double *u, int *iter, int *solved, double *res4) {
    RefSolver *h = static_cast<RefSolver *>(hp);
    if (x) std::memcpy(x, h->s->solution->x.data(), sizeof(double) * h->nx * h->N);
    if (u) std::memcpy(u, h->s->solution->u.data(), sizeof(double) * h->nu * (h->N - 1));
    if (iter) *iter = h->s->solution->iter;
    if (solved) *solved = h->s->solution->solved;
    if (res4) {
        res4[0] = h->s->work->primal_residual_state;
        res4[1] = h->s->work->dual_residual_state;
        res4[2] = h->s->work->primal_residual_input;
        res4[3] = h->s->work->dual_residual_input;
    }
}

void ref_get_cache(void *hp, double *Kinf, double *Pinf, double *Quu_inv, double *AmBKt) {
    RefSolver *h = static_cast<RefSolver *>(hp);
    const TinyCache *c = h->s->cache;
    std::memcpy(Kinf, c->Kinf.data(), sizeof(double) * h->nu * h->nx);
    std::memcpy(Pinf, c->Pinf.data(), sizeof(double) * h->nx * h->nx);
    std::memcpy(Quu_inv, c->Quu_inv.data(), sizeof(double) * h->nu * h->nu);
    std::memcpy(AmBKt, c->AmBKt.data(), sizeof(double) * h->nx * h->nx);
}

// bindings.cpp:262-293 semantics.
void ref_set_cache_terms(void *hp, const double *Kinf, const double *Pinf, const double *Quu_inv,
                         const double *AmBKt) {
    RefSolver *h = static_cast<RefSolver *>(hp);
    h->s->cache->Kinf = map_cm(Kinf, h->nu, h->nx);
    h->s->cache->Pinf = map_cm(Pinf, h->nx, h->nx);
    h->s->cache->Quu_inv = map_cm(Quu_inv, h->nu, h->nu);
    h->s->cache->AmBKt = map_cm(AmBKt, h->nx, h->nx);
}

// Adaptive rho (admm.cpp:147-174, rho_benchmark.cpp): the settings update_settings would push
// (bindings.cpp:360-364) and the public cache fields the Taylor update reads and writes (types.hpp:51-55).
void ref_set_adaptive_rho(void *hp, int enable, double rho_min, double rho_max, int clip) {
    RefSolver *h = static_cast<RefSolver *>(hp);
    h->s->settings->adaptive_rho = enable;
    h->s->settings->adaptive_rho_min = rho_min;
    h->s->settings->adaptive_rho_max = rho_max;
    h->s->settings->adaptive_rho_enable_clipping = clip;
}
// What tiny_setup left in the cache (its built-in tables, whatever the problem is); meaningful for nx=12, nu=4 only.
int ref_get_sensitivity(void *hp, double *dK, double *dP) {
    RefSolver *h = static_cast<RefSolver *>(hp);
    const TinyCache *c = h->s->cache;
    if (c->dKinf_drho.rows() != h->nu || c->dKinf_drho.cols() != h->nx || c->dPinf_drho.rows() != h->nx) return 1;
    std::memcpy(dK, c->dKinf_drho.data(), sizeof(double) * h->nu * h->nx);
    std::memcpy(dP, c->dPinf_drho.data(), sizeof(double) * h->nx * h->nx);
    return 0;
}
// Caller-supplied sensitivities; dC1/dC2 are sized and zeroed (they only reach the dead C1/C2 copies).
void ref_set_sensitivity(void *hp, const double *dK, const double *dP) {
    RefSolver *h = static_cast<RefSolver *>(hp);
    h->s->cache->dKinf_drho = map_cm(dK, h->nu, h->nx);
    h->s->cache->dPinf_drho = map_cm(dP, h->nx, h->nx);
    h->s->cache->dC1_drho = tinyMatrix::Zero(h->nu, h->nu);
    h->s->cache->dC2_drho = tinyMatrix::Zero(h->nx, h->nx);
}
void ref_get_adapted(void *hp, double *rho, double *Kinf, double *Pinf) {
    RefSolver *h = static_cast<RefSolver *>(hp);
    const TinyCache *c = h->s->cache;
    if (rho) *rho = c->rho;
    if (Kinf) std::memcpy(Kinf, c->Kinf.data(), sizeof(double) * h->nu * h->nx);
    if (Pinf) std::memcpy(Pinf, c->Pinf.data(), sizeof(double) * h->nx * h->nx);
}

// Warm-start state of the workspace (what persists between solves).
void ref_get_state(void *hp, double *d, double *y, double *g, double *v, double *z) {
    RefSolver *h = static_cast<RefSolver *>(hp);
    const TinyWorkspace *w = h->s->work;
    int ex = h->nx * h->N, eu = h->nu * (h->N - 1);
    std::memcpy(d, w->d.data(), sizeof(double) * eu);
    std::memcpy(y, w->y.data(), sizeof(double) * eu);
    std::memcpy(g, w->g.data(), sizeof(double) * ex);
    std::memcpy(v, w->v.data(), sizeof(double) * ex);
    std::memcpy(z, w->z.data(), sizeof(double) * eu);
}

// Cold-start batch: instance b uses x0[b*nx..], shared or per-instance refs.
// One TinySolver per thread (the core has no globals; only bindings.cpp does).
// Returns wall seconds of the solve loop (setup excluded).
double ref_solve_batch(const double *A, const double *B, const double *Q, const double *R,
                       double rho, int nx, int nu, int N, const double *xmin, const double *xmax,
                       const double *umin, const double *umax, int use_bounds, double abs_pri_tol,
                       double abs_dua_tol, int max_iter, int check_termination, int batch,
                       const double *x0, const double *xref, const double *uref,
                       int per_instance_ref, double *x_out, double *u_out, int *iter_out,
                       int *solved_out, double *res_out, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    const int ex = nx * N, eu = nu * (N - 1);
    std::vector<void *> hs(nthreads, nullptr);
    for (int t = 0; t < nthreads; ++t) {
        hs[t] = ref_create(A, B, Q, R, rho, nx, nu, N);
        if (!hs[t]) return -1.0;
        if (use_bounds) ref_set_bound_constraints(hs[t], xmin, xmax, umin, umax);
        RefSolver *h = static_cast<RefSolver *>(hs[t]);
        h->s->settings->abs_pri_tol = abs_pri_tol;
        h->s->settings->abs_dua_tol = abs_dua_tol;
        h->s->settings->max_iter = max_iter;
        h->s->settings->check_termination = check_termination;
        if (!per_instance_ref) {
            if (xref) ref_set_x_ref(hs[t], xref);
            if (uref) ref_set_u_ref(hs[t], uref);
        }
    }
    std::atomic<int> next(0);
    auto worker = [&](int t) {
        void *hp = hs[t];
        const int chunk = 64;
        for (;;) {
            int b0 = next.fetch_add(chunk);
            if (b0 >= batch) break;
            int b1 = b0 + chunk < batch ? b0 + chunk : batch;
            for (int b = b0; b < b1; ++b) {
                ref_set_x0(hp, x0 + (size_t)b * nx);
                if (per_instance_ref) {
                    if (xref) ref_set_x_ref(hp, xref + (size_t)b * ex);
                    if (uref) ref_set_u_ref(hp, uref + (size_t)b * eu);
                }
                ref_reset(hp);
                ref_solve(hp);
                ref_get_solution(hp, x_out ? x_out + (size_t)b * ex : nullptr,
                                 u_out ? u_out + (size_t)b * eu : nullptr,
                                 iter_out ? iter_out + b : nullptr,
                                 solved_out ? solved_out + b : nullptr,
                                 res_out ? res_out + (size_t)b * 4 : nullptr);
            }
        }
    };
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int t = 1; t < nthreads; ++t) th.emplace_back(worker, t);
    worker(0);
    for (auto &x : th) x.join();
    auto t1 = std::chrono::steady_clock::now();
    for (void *hp : hs) ref_destroy(hp);
    return std::chrono::duration<double>(t1 - t0).count();
}

}  // extern "C"
