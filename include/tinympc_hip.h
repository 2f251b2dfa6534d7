/* libtinympc_hip.so — C-ABI of the MI355X batched TinyMPC ADMM engine.
 *
 * Drop-in boundary: the reference's Julia module reaches its solver through
 * `ccall((:sym, libtinympc_jl), Int32, ...)` into src/bindings.cpp
 * (reference: src/TinyMPC.jl:12-14, src/bindings.cpp:15-490).  This header
 * declares
 *   (1) the same symbol names with the SAME argument lists and return
 *       conventions as bindings.cpp, operating on one process-global solver, so
 *       the unmodified TinyMPC.jl can ccall this library instead; the batch
 *       dimension rides on the (rows, cols) arguments the reference already
 *       passes (x0 as nx x B, x_ref as nx x N*B, ...), and
 *   (2) a handle-based extension (`tinympc_*`) adding what a batched, device-
 *       resident engine needs: batch size, per-instance status, cold reset,
 *       device-pointer I/O and stream-ordered solves.
 *
 * Conventions (as bindings.cpp): every matrix is column-major fp64 passed as
 * pointer + (rows, cols); inputs are borrowed for the call only; outputs are
 * written into caller-allocated buffers; return 0 = OK, -1 = error (message on
 * stderr); solve_mpc passes the solver status through (0 all converged, 1 some
 * instance hit max_iter).  Plain C, no torch / HIP types except the opaque
 * stream handle (void*).  The library has no CPU fallback: every entry point
 * that computes fails with -1 when no HIP device is usable.
 */
#ifndef TINYMPC_HIP_H
#define TINYMPC_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------- */
/* (1) process-global solver: the reference's own entry points               */
/* ------------------------------------------------------------------------- */

/* replaces bindings.cpp:21-73 setup_solver.  A non-zero fdyn (affine dynamics x+ = A x + B u + f)
 * is honoured, but its arithmetic lives only in the un-vendored TinyMPC submodule (SURVEY.md §0
 * fact 2): parity for it is UNPINNED, and such problems run on the stream (or generic) kernel.
 * Batch starts at 1; see set_batch_size. */
int setup_solver(double *A_data, int A_rows, int A_cols, double *B_data, int B_rows, int B_cols,
                 double *fdyn_data, int fdyn_rows, int fdyn_cols, double *Q_data, int Q_rows,
                 int Q_cols, double *R_data, int R_rows, int R_cols, double rho, int nx, int nu,
                 int N, int verbose);

/* replaces bindings.cpp:75-96.  x0 is nx x 1 (broadcast to every instance) or nx x batch. */
int set_x0(double *x0_data, int x0_rows, int x0_cols, int verbose);
/* replaces bindings.cpp:98-119.  nx x N (shared by the batch) or nx x (N*batch). */
int set_x_ref(double *x_ref_data, int x_ref_rows, int x_ref_cols, int verbose);
/* replaces bindings.cpp:121-142.  nu x (N-1) (shared) or nu x ((N-1)*batch). */
int set_u_ref(double *u_ref_data, int u_ref_rows, int u_ref_cols, int verbose);
/* replaces bindings.cpp:144-159.  0: every instance converged, 1: some hit max_iter, -1: error. */
int solve_mpc(int verbose);
/* replace bindings.cpp:161-203.  Buffers hold nx*N*batch / nu*(N-1)*batch doubles;
 * rows = nx / nu, cols = N*batch / (N-1)*batch (Julia: reshape to (nx, N, batch)). */
int get_states(double *states_buffer, int *rows, int *cols);
int get_controls(double *controls_buffer, int *rows, int *cols);
/* replaces bindings.cpp:205-208 */
void cleanup_solver(void);
/* Additions next to bindings.cpp:75,161,183 for hosts that keep Float32 arrays: the same three per-solve transfers without
 * the fp64 narrowing / widening pass on the host (the device buffers are fp32; a 65 536-instance cartpole round trip is
 * 1.3 ms through the fp64 forms, most of it that pass).  Shapes and conventions as set_x0 / get_states / get_controls. */
int set_x0_f32(float *x0_data, int x0_rows, int x0_cols, int verbose);
int get_states_f32(float *states_buffer, int *rows, int *cols);
int get_controls_f32(float *controls_buffer, int *rows, int *cols);
/* Page-lock / release a caller-owned host array that is reused with the fp32 forms above (direct DMA instead of a bounce
 * through pageable memory).  The library never registers caller memory on its own; unpin before the array is freed. */
int set_precision(int precision);   /* tinympc_set_precision on the process-global solver: 0 default, 1 all fp32, 2 all fp64 */
int pin_host_buffer(void *ptr, size_t bytes);
int unpin_host_buffer(void *ptr);
/* replaces bindings.cpp:336-376.  en_*_soc / en_*_linear switch the sets given to set_cone_constraints /
 * set_linear_constraints on and off (parity unpinned).  adaptive_rho != 0 turns on the per-instance rho adaptation of
 * admm.cpp:147-174 (see tinympc_set_adaptive_rho; runs on the generic kernel).  check_termination <= 0 means "never check"
 * (the reference divides by it, admm.cpp:91). */
int update_settings(double abs_pri_tol, double abs_dua_tol, int max_iter, int check_termination,
                    int en_state_bound, int en_input_bound, int en_state_soc, int en_input_soc,
                    int en_state_linear, int en_input_linear, int adaptive_rho,
                    double adaptive_rho_min, double adaptive_rho_max,
                    int adaptive_rho_enable_clipping, int verbose);
/* replaces bindings.cpp:378-411.  Bounds are per knot (nx x N, nu x (N-1)), shared by the batch;
 * enables en_state_bound and en_input_bound on success. */
int set_bound_constraints(double *x_min_data, int x_min_rows, int x_min_cols, double *x_max_data,
                          int x_max_rows, int x_max_cols, double *u_min_data, int u_min_rows,
                          int u_min_cols, double *u_max_data, int u_max_rows, int u_max_cols,
                          int verbose);
/* replaces bindings.cpp:262-293 */
int set_cache_terms(double *Kinf_data, int Kinf_rows, int Kinf_cols, double *Pinf_data,
                    int Pinf_rows, int Pinf_cols, double *Quu_inv_data, int Quu_inv_rows,
                    int Quu_inv_cols, double *AmBKt_data, int AmBKt_rows, int AmBKt_cols,
                    int verbose);
/* The sensitivities the reference takes through codegen_with_sensitivity (bindings.cpp:298-334; TinyMPC.jl:374-394) for
 * its generated solver, here for the live one: dK (nu x nx), dP (nx x nx); dC1 / dC2 may be NULL (see
 * tinympc_set_sensitivity).  Optional: update_settings(adaptive_rho = 1) without it computes them on first use. */
int set_sensitivity(double *dK_data, int dK_rows, int dK_cols, double *dP_data, int dP_rows, int dP_cols,
                    double *dC1_data, int dC1_rows, int dC1_cols, double *dC2_data, int dC2_rows, int dC2_cols,
                    int verbose);
/* rho of every instance after adaptation (the reference's solver->cache->rho, one per instance); *count = batch */
int get_adaptive_rho(double *rho_buffer, int *count);
/* replaces bindings.cpp:228-259 */
int print_problem_data(int verbose);
/* replace bindings.cpp:413-490.  set_linear_constraints: Alin_x x <= blin_x and Alin_u u <= blin_u at every knot
 * (README.md:115-116), column-major (rows x nx) / (rows x nu), at most 8 rows per side (an equality is two rows,
 * TinyMPC.jl:258-267); enables the non-empty halves.  set_cone_constraints: per-knot second-order cones, inputs first; cone i covers rows
 * [Ac[i], Ac[i]+qc[i]) of each knot, the LAST row is the axis: ||head|| <= c[i] * axis
 * (rocket_landing_constraints.jl:51-57,132); at most 8 cones per side; enables the non-empty halves.
 * Cone, linear-inequality and fdyn arithmetic lives only in the absent submodule — parity UNPINNED. */
int set_linear_constraints(double *Alin_x_data, int Alin_x_rows, int Alin_x_cols,
                           double *blin_x_data, int blin_x_len, double *Alin_u_data,
                           int Alin_u_rows, int Alin_u_cols, double *blin_u_data, int blin_u_len,
                           int verbose);
int set_cone_constraints(int *Acu_data, int Acu_len, int *qcu_data, int qcu_len, double *cu_data,
                         int cu_len, int *Acx_data, int Acx_len, int *qcx_data, int qcx_len,
                         double *cx_data, int cx_len, int verbose);

/* --- batch extensions on the global solver (no counterpart in bindings.cpp) --- */
/* Re-shapes the global solver to `batch` instances; per-instance inputs/state are reset. */
int set_batch_size(int batch);
int get_batch_size(void);
/* Per-instance iteration count, solved flag and the 4 residuals
 * (pri_state, dua_state, pri_input, dua_input) — the batched form of
 * solution->iter / solution->solved / work->*_residual_* (types.hpp:32-37,128-131).
 * Any pointer may be NULL. */
int get_status(int *iter, int *solved, double *residuals4);
/* Cold start: zero d, y, g, v, z of every instance (what tiny_setup leaves, tiny_api.cpp:73-88). */
int reset_workspace(void);
/* The fused closed loop on the process-global solver (tinympc_mpc_rollout / tinympc_set_ref_sequence below): `steps`
 * repetitions of  set_x0 -> [set_x_ref / set_u_ref of the step] -> solve -> x0 = A x0 + B u0 + f  in one launch — the loop of
 * examples/cartpole_example_mpc.jl:35-51 and examples/rocket_landing_constraints.jl:97-134.  Returns the status of the last
 * step's solve (0 / 1) or -1; logs (any may be NULL) as tinympc_get_mpc_log. */
int set_ref_sequence(double *x_ref_seq, int x_rows, int x_cols, double *u_ref_seq, int u_rows, int u_cols, int steps);
int mpc_rollout(int steps, double *x_log, double *u_log, int *iter_log);

/* ------------------------------------------------------------------------- */
/* (2) handle API                                                            */
/* ------------------------------------------------------------------------- */
typedef struct tinympc_solver tinympc_solver;

/* device < 0: current HIP device. */
int tinympc_create(tinympc_solver **out, const double *A, const double *B, const double *Q,
                   const double *R, double rho, int nx, int nu, int N, int batch, int device,
                   int verbose);
/* One problem family PER INSTANCE (no counterpart in the reference, whose solver holds one
 * family): A [batch][nx*nx], B [batch][nx*nu], Q [batch][nx*nx], R [batch][nu*nu] (each block column-major),
 * rho [batch].  Every instance gets its own host-side fp64 Riccati cache; bounds, settings and references
 * behave as in the single-family solver.  Needs (nx, nu) in the stream-kernel grid
 * (nx in {2,3,4,6,8,10,12}, nu <= 4); the batch size is fixed. */
int tinympc_create_families(tinympc_solver **out, const double *A, const double *B, const double *Q,
                            const double *R, const double *rho, int nx, int nu, int N, int batch,
                            int device, int verbose);
void tinympc_destroy(tinympc_solver *s);
int tinympc_update_settings(tinympc_solver *s, double abs_pri_tol, double abs_dua_tol,
                            int max_iter, int check_termination, int en_state_bound,
                            int en_input_bound);
int tinympc_set_bound_constraints(tinympc_solver *s, const double *x_min, const double *x_max,
                                  const double *u_min, const double *u_max);
int tinympc_set_cache_terms(tinympc_solver *s, const double *Kinf, const double *Pinf,
                            const double *Quu_inv, const double *AmBKt);
/* Unpinned extensions (see setup_solver / set_cone_constraints above). */
int tinympc_set_fdyn(tinympc_solver *s, const double *fdyn /* nx, NULL = zero */);
int tinympc_set_cone_constraints(tinympc_solver *s, const int *Acu, const int *qcu, const double *cu,
                                 int n_input_cones, const int *Acx, const int *qcx, const double *cx,
                                 int n_state_cones);
int tinympc_enable_cones(tinympc_solver *s, int en_state_soc, int en_input_soc);
int tinympc_set_linear_constraints(tinympc_solver *s, const double *Alin_x, int rows_x, const double *blin_x,
                                   const double *Alin_u, int rows_u, const double *blin_u);
int tinympc_enable_linear(tinympc_solver *s, int en_state_linear, int en_input_linear);
int tinympc_get_cache_terms(tinympc_solver *s, double *Kinf, double *Pinf, double *Quu_inv,
                            double *AmBKt);
/* Adaptive rho — admm.cpp:147-174 + rho_benchmark.cpp, per instance: every 5th ADMM iteration each instance predicts
 * a new rho from its normalised residuals (clipped to [rho_min, rho_max] if enable_clipping) and moves ITS Kinf, Pinf
 * by delta_rho * sensitivity.  The adapted (rho, Kinf, Pinf) are solver state and persist between solves, as the
 * reference's mutated cache does; tinympc_reset / set_cache_terms / toggling the switch return them to the family's
 * cache.  Settings: bindings.cpp:360-364.  Not available on per-instance-family solvers or in mpc_rollout. */
int tinympc_set_adaptive_rho(tinympc_solver *s, int enable, double rho_min, double rho_max, int enable_clipping);
/* dKinf/drho (nu x nx) and dPinf/drho (nx x nx), column-major — what codegen_with_sensitivity receives
 * (bindings.cpp:298-334) and tiny_setup hard-codes for the 12x4 quadrotor (tiny_api.cpp:269-329).  dC1 / dC2 may be
 * NULL: the reference applies them to copies the iteration never reads.  If never set, the first adaptive solve
 * computes them as TinyMPC.jl:301-352 does (tinympc_compute_sensitivity). */
int tinympc_set_sensitivity(tinympc_solver *s, const double *dKinf, const double *dPinf, const double *dC1,
                            const double *dC2);
/* Host-side finite differences (h = 1e-6) of the rho-regularised LQR — TinyMPC.jl:301-352
 * (compute_sensitivity_autograd / solve_lqr).  Any output may be NULL. */
int tinympc_compute_sensitivity(tinympc_solver *s, double *dKinf, double *dPinf, double *dC1, double *dC2);
/* Per-instance adapted values: rho [batch], Kinf [batch][nu*nx], Pinf [batch][nx*nx] (column-major each); any may be
 * NULL.  Before any adaptive solve: the family's values. */
int tinympc_get_adaptive_state(tinympc_solver *s, double *rho, double *Kinf, double *Pinf);
int tinympc_set_x0(tinympc_solver *s, const double *x0, int cols);       /* cols: 1 | batch */
int tinympc_set_x_ref(tinympc_solver *s, const double *x_ref, int cols); /* cols: N | N*batch */
int tinympc_set_u_ref(tinympc_solver *s, const double *u_ref, int cols); /* N-1 | (N-1)*batch */
int tinympc_reset(tinympc_solver *s);
/* warm_start: 0 = every solve starts from the zero workspace and keeps no state (benchmark
 * configs); 1 = the workspace (d, y, g, v, z) persists between solves like the reference's
 * (SURVEY.md 3.5).  Default 1. */
int tinympc_set_warm_start(tinympc_solver *s, int warm_start);
int tinympc_solve(tinympc_solver *s);
int tinympc_get_states(tinympc_solver *s, double *buf);
int tinympc_get_controls(tinympc_solver *s, double *buf);
int tinympc_get_status(tinympc_solver *s, int *iter, int *solved, double *residuals4);
/* Warm-start state of every instance, instance-major: d,y,z [batch][N-1][nu]; g,v [batch][N][nx]. */
int tinympc_get_workspace(tinympc_solver *s, double *d, double *y, double *g, double *v, double *z);
int tinympc_set_workspace(tinympc_solver *s, const double *d, const double *y, const double *g,
                          const double *v, const double *z);

/* --- device-resident I/O (fp32, instance-major), for callers that keep data in HBM --- */
/* Device pointers owned by the solver; valid until destroy / re-batch.
 *   x0 [batch][nx]; x_ref [batch][N][nx] (or [N][nx] when shared); u_ref likewise;
 *   states [batch][N][nx]; controls [batch][N-1][nu]; iter/solved [batch] int32;
 *   residuals [batch][4]; gstat: 8 x uint32 (max residual bits [0..3], unsolved count [4]). */
int tinympc_device_buffers(tinympc_solver *s, void **x0, void **x_ref, void **u_ref,
                           void **states, void **controls, void **iter, void **solved,
                           void **residuals, void **gstat);
/* Declare how the device-side reference buffers are to be read: 0 zero, 1 shared, 2 per instance. */
int tinympc_set_ref_mode(tinympc_solver *s, int ref_mode);
/* Enqueue one batched solve on `hip_stream` (NULL = default stream) without synchronising;
 * results are in the device buffers when the stream reaches this point. */
int tinympc_solve_async(tinympc_solver *s, void *hip_stream);
/* After the stream has been synchronised: 0 all converged / 1 otherwise (reads gstat). */
int tinympc_solve_status(tinympc_solver *s);
/* fp32 host buffers (plain copies to / from the fp32 device buffers, no conversion pass): x0 nx x 1 or nx x batch;
 * states nx*N*batch, controls nu*(N-1)*batch floats, instance-major as the fp64 forms.  The library never page-locks
 * memory it does not own: a caller that reuses its buffers and wants direct DMA into them registers them itself with
 * tinympc_pin_host and removes the registration with tinympc_unpin_host BEFORE freeing the memory (whatever is still
 * registered at tinympc_destroy is unregistered there). */
int tinympc_set_x0_f32(tinympc_solver *s, const float *x0, int cols);
int tinympc_get_states_f32(tinympc_solver *s, float *buf);
int tinympc_get_controls_f32(tinympc_solver *s, float *buf);
int tinympc_pin_host(tinympc_solver *s, void *ptr, size_t bytes);
int tinympc_unpin_host(tinympc_solver *s, void *ptr);
/* Fused closed loop (the caller pattern of examples/cartpole_example_mpc.jl:35-51):
 * `steps` repetitions of  solve -> u0 = controls[:,0] -> x0 = A x0 + B u0 -> set_x0  in ONE launch,
 * the warm-start workspace staying on chip between steps.  Needs warm-start mode and a specialised
 * kernel for the shape.  Synchronous; returns the solve status of the LAST step (0/1) or -1.
 * Afterwards x0 holds the plant state after the last step and get_states/get_controls/get_status
 * describe the last solve.  Logs, instance-major: x [batch][steps][nx] (plant state after each
 * step), u [batch][steps][nu] (control applied), iter [batch][steps] (ADMM iterations of the step,
 * negated when the step hit max_iter).  Any log pointer may be NULL. */
int tinympc_mpc_rollout(tinympc_solver *s, int steps, void *hip_stream);
int tinympc_get_mpc_log(tinympc_solver *s, double *x, double *u, int *iter);
/* Shared references of EVERY step of the next closed loops — the caller pattern of
 * examples/rocket_landing_constraints.jl:97-134, which shifts x_ref by one knot per step (:107-115) before each solve:
 * x_ref_seq is nx x (N*steps), u_ref_seq nu x ((N-1)*steps), column-major, step after step (what `steps` calls of
 * set_x_ref / set_u_ref would have passed).  The plant step of such a loop includes the affine term,
 * x0 = A x0 + B u0 + f (:123).  Step 0's references become the solver's own shared references and stay installed when
 * the sequence is dropped (by a later set_x_ref / set_u_ref, which replaces them, or by steps = 0, which does not).  Needs the transposed-sets kernel (mfmat) for the shape. */
int tinympc_set_ref_sequence(tinympc_solver *s, const double *x_ref_seq, int x_rows, int x_cols, const double *u_ref_seq,
                             int u_rows, int u_cols, int steps);
/* Tolerance-terminated solves of big batches: with chunk_iters > 0 (rounded up to a multiple of check_termination)
 * a solve runs in chunks of that many iterations and, between chunks, gathers the instances still iterating into
 * a dense list, so that wavefronts do not idle behind their slowest instance.  Iterates, iteration counts and
 * residuals are those of the single-launch solve.  The solve then synchronises its stream.  Ignored for
 * fixed-iteration settings (a tolerance <= 0), per-instance families, closed-loop rollouts.  0 = off (default). */
int tinympc_set_compaction(tinympc_solver *s, int chunk_iters);
/* Kernel timing: when enabled, every solve records HIP events immediately around the ADMM kernel
 * launch on the launch stream; tinympc_kernel_elapsed_ms returns the last kernel's duration,
 * tinympc_kernel_elapsed_mean_ms the mean over the last `last_n` launches (at most 256)
 * (call after the stream has been synchronised; < 0 if unavailable). */
int tinympc_set_profiling(tinympc_solver *s, int enable);
double tinympc_kernel_elapsed_ms(tinympc_solver *s);
double tinympc_kernel_elapsed_mean_ms(tinympc_solver *s, int last_n);
/* Arithmetic of the two serial recurrences (rollout, Riccati gradient): 0 = fp64 accumulation
 * with fp64 coefficients (default; ADMM state and elementwise steps stay fp32), 1 = all fp32,
 * 2 = all fp64 — the reference's own arithmetic end to end (types.hpp:15): recurrences, slacks, duals, residual
 * comparisons and the workspace kept between solves in fp64, on the generic kernel (any shape and option set of the
 * single-family solvers; no fused closed loop); x0 / references / bounds go in and the solution comes out as the fp32
 * device arrays.  It is the mode for callers who need the reference's digits rather than the 1e-5 of the fp32-state
 * kernels (families whose duals lose digits to fp32 rounding miss 1e-5 there); it is the slowest path of the library —
 * except cold one-shot solves of cartpole-class shapes ((4,1) to N = 30, (3,2), (2,x)), which are launched on the headline
 * kernel's fp64-state variant, compiled on request (csrc/jit.cpp; tinympc_last_launch_name: "lean<...;f64>"): the same
 * digits at that kernel's speed.
 * Switching to or from precision 2 restarts the workspace cold.
 * precision = 1 is a request to save time and does NOT meet the 1e-5 parity target on every instance (3 of the 65 536
 * benchmark cartpole instances miss it, worst 1.6e-5).  Where a shape has a matrix-core kernel that kernel is faster than the
 * fp32 one, so such solves run there — with fp64 recurrences — unless tinympc_set_strict_precision(s, 1) insists on fp32.
 * tinympc_effective_precision returns the recurrence precision (0 / 1) the solver's current options run with. */
int tinympc_set_precision(tinympc_solver *s, int precision);
int tinympc_set_strict_precision(tinympc_solver *s, int strict);
/* Tuning / test aid.  The TINYMPC_HIP_* environment switches are read ONCE, when a solver is created —
 * nothing on the solve path calls getenv; this re-reads them for a live solver. */
int tinympc_reload_switches(tinympc_solver *s);
int tinympc_effective_precision(tinympc_solver *s);
/* Name of the kernel family the solver's shape / options select: "quad<nx,nu,N,gG>", "mfma<...>", "mfmat<...>",
 * "stream4<nx,nu>", "generic", ... */
const char *tinympc_kernel_name(tinympc_solver *s);
/* Name of the kernel the most recent launch actually ran: the family above, or the variant a launch of that family took
 * for its calling pattern — "lean<nx,nu,N>" for one-shot solves (cold start, workspace not kept) of a one-lane-per-instance
 * quad entry without an active state bound, zero references, fp64 recurrences (admm_lean.hip.h). */
const char *tinympc_last_launch_name(tinympc_solver *s);
/* Algorithmic HBM bytes and FLOPs of one solve of the whole batch (SURVEY.md 8d formulas);
 * flops assume `iters` ADMM iterations per instance. */
double tinympc_algorithmic_bytes(tinympc_solver *s);
double tinympc_algorithmic_flops(tinympc_solver *s, int iters);
const char *tinympc_last_error(void);
/* Specialisation at setup.  The reference takes any (nx, nu, N) at run time (tiny_api.cpp:21-71); the on-chip kernels here are
 * compiled per shape.  For a shape the library was not built with, tinympc_create / setup_solver compile the one instantiation
 * that suits it (hipcc as a child process, from the library's own csrc/ headers), cache it under $TINYMPC_HIP_CACHE or
 * ~/.cache/tinympc_hip/<source hash>/ and load it; the time of the one-off compile is printed on stderr.  Without a
 * compiler or the sources, when the shape's state does not fit the chip, or with TINYMPC_HIP_NO_JIT=1, such a solver runs
 * on the run-time-shape kernels (stream / generic) as before.  This entry point does the same ahead of time (no GPU needed):
 * returns 1 if an on-chip kernel exists for the shape afterwards, 0 if not. */
int tinympc_specialise(int nx, int nu, int N, int verbose);
/* Host-only fp64 part of setup() — the infinite-horizon Riccati precompute of
 * tiny_precompute_and_set_cache (tiny_api.cpp:124-190, incl. the rho-twice quirk of
 * tiny_setup :90-91,113).  Needs no GPU.  Outputs column-major: Kinf nu x nx, Pinf nx x nx,
 * Quu_inv nu x nu, AmBKt nx x nx.  Returns 0, or -1 if R + B'PB is singular. */
int tinympc_host_precompute(const double *A, const double *B, const double *Q, const double *R,
                            double rho, int nx, int nu, double *Kinf, double *Pinf,
                            double *Quu_inv, double *AmBKt);
/* Host-only (no GPU): the sensitivities tinympc_compute_sensitivity returns, for a family given directly. */
int tinympc_host_sensitivity(const double *A, const double *B, const double *Q, const double *R, double rho, int nx,
                             int nu, double *dKinf, double *dPinf, double *dC1, double *dC2);

/* ------------------------------------------------------------------------- */
/* (3) multi-GPU: one handle, n_gpus devices of one node, one host process   */
/* ------------------------------------------------------------------------- */
/* SURVEY.md 8(b) "what the replacement exports": setup_solver(..., batch, n_gpus); 8(e).  The reference has one
 * process-global CPU solver (bindings.cpp:15-18) and nothing to distribute; this is the surface a `ccall` host uses to
 * drive all GPUs of a node without a process launcher.  The batch is cut into contiguous shards (shard i = instances
 * [lo_i, hi_i), sizes differing by at most one), shard i on devices[i] with its own stream; inputs are scattered and
 * outputs gathered by offset on the caller's instance-major buffers; no data moves between GPUs.  The one exchange is
 * the solve status: each solve ends with ONE all-reduce(MAX) over RCCL (ncclCommInitAll at creation, a grouped
 * ncclAllReduce of 8 uint32 per solve) of the devices' status blocks, enqueued on the shards' streams behind the
 * kernels.  librccl.so is loaded at run time by the first call of tinympc_create_sharded. */
typedef struct tinympc_sharded tinympc_sharded;
/* devices: n_gpus device ordinals, NULL = 0 .. n_gpus-1.  Needs 1 <= n_gpus <= batch.  A list that repeats a device
 * cannot form an RCCL communicator: such a handle folds the status on the host (one-GPU rehearsal of the multi-shard
 * path; see tinympc_sharded_fold_backend). */
int tinympc_create_sharded(tinympc_sharded **out, const double *A, const double *B, const double *Q, const double *R,
                           double rho, int nx, int nu, int N, int batch, int n_gpus, const int *devices, int verbose);
void tinympc_sharded_destroy(tinympc_sharded *s);
int tinympc_sharded_n_shards(tinympc_sharded *s);
/* "rccl" or "host" */
const char *tinympc_sharded_fold_backend(tinympc_sharded *s);
/* shard i: its device, its instance range [lo, hi) and its single-device handle (for the setters not mirrored below:
 * cones, linear rows, cache terms, adaptive rho ... — family-level calls, to be made on every shard).  Any out may be NULL. */
int tinympc_sharded_shard(tinympc_sharded *s, int i, int *device, int *lo, int *hi, tinympc_solver **local);
/* the partition rule itself (pure arithmetic, no GPU): shard `shard` of `n_shards` over `batch` instances */
void tinympc_shard_range(int batch, int n_shards, int shard, int *lo, int *hi);
/* family-level: applied to every shard (arguments as the tinympc_* function of the same name) */
int tinympc_sharded_update_settings(tinympc_sharded *s, double abs_pri_tol, double abs_dua_tol, int max_iter,
                                    int check_termination, int en_state_bound, int en_input_bound);
int tinympc_sharded_set_bound_constraints(tinympc_sharded *s, const double *x_min, const double *x_max,
                                          const double *u_min, const double *u_max);
int tinympc_sharded_set_warm_start(tinympc_sharded *s, int warm_start);
int tinympc_sharded_reset(tinympc_sharded *s);
int tinympc_sharded_set_precision(tinympc_sharded *s, int precision);
int tinympc_sharded_set_compaction(tinympc_sharded *s, int chunk_iters);
/* per-instance inputs over the WHOLE batch (cols as tinympc_set_x0 / _x_ref / _u_ref), scattered to the shards */
int tinympc_sharded_set_x0(tinympc_sharded *s, const double *x0, int cols);
int tinympc_sharded_set_x_ref(tinympc_sharded *s, const double *x_ref, int cols);
int tinympc_sharded_set_u_ref(tinympc_sharded *s, const double *u_ref, int cols);
/* One batched solve on every device + the status all-reduce.  _solve = _solve_async + _wait.  Returns the GLOBAL status:
 * 0 iff every instance on every device converged, 1 otherwise, -1 on error (solve_mpc's convention, bindings.cpp:144-159). */
int tinympc_sharded_solve(tinympc_sharded *s);
int tinympc_sharded_solve_async(tinympc_sharded *s);
int tinympc_sharded_wait(tinympc_sharded *s);
/* the folded status block of the last solve: global residual maxima (pri_x, dua_x, pri_u, dua_u) and the largest
 * per-device count of unsolved instances (> 0 iff the status is 1).  Any out may be NULL. */
int tinympc_sharded_global_status(tinympc_sharded *s, double *residual_maxima4, int *max_unsolved_per_device);
/* outputs over the WHOLE batch, gathered from the shards into caller-allocated buffers (sizes as tinympc_get_*) */
int tinympc_sharded_get_states(tinympc_sharded *s, double *buf);
int tinympc_sharded_get_controls(tinympc_sharded *s, double *buf);
int tinympc_sharded_get_status(tinympc_sharded *s, int *iter, int *solved, double *residuals4);
int tinympc_sharded_get_workspace(tinympc_sharded *s, double *d, double *y, double *g, double *v, double *z);

/* The same on the process-global solver (the bare entry points of section (1)): after setup_solver / set_batch_size,
 * set_gpus(n) spreads the global solver's batch over devices 0 .. n-1 (n = 1: back to one device).  Every entry point
 * of section (1) then acts on the whole sharded batch — per-instance buffers are scattered / gathered, family-level
 * setters reach every shard, solve_mpc returns the all-reduced status — so src/TinyMPC.jl needs one extra ccall in
 * setup() and nothing else.  Inputs and the workspace are reset, as by set_batch_size. */
int set_gpus(int n_gpus);
int get_gpus(void);
/* warm_start = 0: every solve_mpc of the global solver starts from the zero workspace and keeps none (one-shot solves:
 * the regime of the benchmark configs, served by the on-chip kernels); 1 (default) = the reference's semantics, the
 * workspace persists between solves (admm.cpp:112-115).  The kernel the last solve ran on, for logs. */
int set_warm_start(int warm_start);
const char *get_kernel_name(void);

#ifdef __cplusplus
}
#endif
#endif /* TINYMPC_HIP_H */
