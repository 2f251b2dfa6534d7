/* TEST INFRASTRUCTURE ONLY — CPU restatement ("port") of the reference's TinyMPC
 * ADMM solve path.  Never linked, imported or executed by the product path.
 *
 * Two instantiations of the same plain-loop C code (tinympc_oracle_body.inc):
 *   orc64_*  all arithmetic in fp64 — restates the reference exactly
 *            (reference: src/codegen_src/tinympc/types.hpp:15 `typedef double tinytype`)
 *   orc32_*  ADMM loop in fp32, fed the fp64-computed Riccati cache rounded once to
 *            fp32 — the arithmetic model of the HIP kernel
 *
 * Parity pin: orc64 is checked against golden vectors produced by the compiled
 * reference snapshot (oracle/_ref, oracle/make_golden.py -> tests/golden/*.json)
 * to <= 1e-12 in tests/test_oracle.py.
 *
 * All matrices are column-major fp64 at this API, like the reference's C-ABI
 * (reference: src/bindings.cpp:21-27, src/TinyMPC.jl:77-83).
 */
#ifndef TINYMPC_ORACLE_H
#define TINYMPC_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_DECLARE(P)                                                                          \
    void *P##create(const double *A, const double *B, const double *Q, const double *R,        \
                    double rho, int nx, int nu, int N);                                        \
    void P##destroy(void *h);                                                                  \
    void P##update_settings(void *h, double abs_pri_tol, double abs_dua_tol, int max_iter,     \
                            int check_termination, int en_state_bound, int en_input_bound);   \
    void P##set_bound_constraints(void *h, const double *xmin, const double *xmax,            \
                                  const double *umin, const double *umax);                    \
    void P##set_x0(void *h, const double *x0);                                                 \
    void P##set_x_ref(void *h, const double *xref);                                            \
    void P##set_u_ref(void *h, const double *uref);                                            \
    void P##project_soc_block(double *blk, int q, double mu);                                 \
    void P##set_fdyn(void *h, const double *fdyn);                                            \
    void P##set_cone_constraints(void *h, const int *Acu, const int *qcu, const double *cu,   \
                                 int ncu, const int *Acx, const int *qcx, const double *cx,   \
                                 int ncx);                                                    \
    void P##set_linear_constraints(void *h, const double *Alin_x, int mx, const double *blin_x, \
                                   const double *Alin_u, int mu, const double *blin_u);       \
    void P##project_halfspaces_block(double *z, int n, const double *A_rowmajor,              \
                                     const double *b, int m);                                 \
    void P##set_cache_terms(void *h, const double *Kinf, const double *Pinf,                  \
                            const double *Quu_inv, const double *AmBKt);                      \
    void P##set_adaptive_rho(void *h, int enable, double rho_min, double rho_max, int clip);   \
    void P##set_sensitivity(void *h, const double *dKinf_drho, const double *dPinf_drho);     \
    void P##get_adapted(void *h, double *rho, double *Kinf, double *Pinf);                    \
    void P##reset(void *h);                                                                    \
    void P##set_forced_exit(void *h, int k);                                                   \
    int P##solve(void *h);                                                                     \
    void P##get_solution(void *h, double *x, double *u, int *iter, int *solved, double *res4); \
    void P##get_cache(void *h, double *Kinf, double *Pinf, double *Quu_inv, double *AmBKt);    \
    void P##get_state(void *h, double *d, double *y, double *g, double *v, double *z);         \
    void P##set_state(void *h, const double *d, const double *y, const double *g,             \
                      const double *v, const double *z);                                      \
    void P##get_cone_state(void *h, double *gc, double *vc, double *yc, double *zc);           \
    void P##set_cone_state(void *h, const double *gc, const double *vc, const double *yc,      \
                           const double *zc);                                                 \
    double P##solve_batch(const double *A, const double *B, const double *Q, const double *R, \
                          double rho, int nx, int nu, int N, const double *xmin,              \
                          const double *xmax, const double *umin, const double *umax,         \
                          int use_bounds, double abs_pri_tol, double abs_dua_tol,             \
                          int max_iter, int check_termination, int batch, const double *x0,   \
                          const double *xref, const double *uref, int per_instance_ref,       \
                          double *x_out, double *u_out, int *iter_out, int *solved_out,       \
                          double *res_out, int nthreads);

ORC_DECLARE(orc64_)
ORC_DECLARE(orc32_)

#ifdef __cplusplus
}
#endif
#endif
