// extern "C" surface of libtinympc_hip.so (declared in include/tinympc_hip.h).
// The process-global entry points keep the names, argument lists and return
// conventions of the reference shim (reference: src/bindings.cpp:15-490) so that
// src/TinyMPC.jl's ccalls bind unchanged; the tinympc_* handle API adds the batch.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <exception>
#include <memory>

#include "../../include/tinympc_hip.h"
#include "solver.h"

using tmpc::set_error;

namespace {

std::unique_ptr<tinympc_solver> g_solver;  // bindings.cpp:15
// set_gpus(n > 1): the global solver's batch lives on n devices; g_solver then keeps the family-level state only
// (batch 1) so that a later set_gpus / set_batch_size can rebuild the shards from it
struct ShardedDeleter {
    void operator()(tinympc_sharded *p) const { tinympc_sharded_destroy(p); }
};
std::unique_ptr<tinympc_sharded, ShardedDeleter> g_sharded;
int g_sharded_batch = 0, g_verbose = 0;

tinympc_solver *global_shard(int i) {
    tinympc_solver *h = nullptr;
    tinympc_sharded_shard(g_sharded.get(), i, nullptr, nullptr, nullptr, &h);
    return h;
}
// a family-level call on the global solver: on the master and, when sharded, on every shard
template <class F>
int each_global(F &&f) {
    if (int rc = f(g_solver.get())) return rc;
    if (g_sharded)
        for (int i = 0; i < tinympc_sharded_n_shards(g_sharded.get()); ++i)
            if (int rc = f(global_shard(i))) return rc;
    return 0;
}
int global_batch() { return g_sharded ? g_sharded_batch : (g_solver ? g_solver->s.batch : 0); }

// (re)build the shards of the global solver for `batch` instances on `n_gpus` devices from the master's family state
int reshard_global(int batch, int n_gpus) {
    tmpc::Solver &m = g_solver->s;
    if (n_gpus <= 1) {
        g_sharded.reset();
        return m.alloc_batch(batch);
    }
    // TINYMPC_HIP_SHARD_DEVICES="0,0,...": explicit device list (test aid: several shards on one GPU, host status fold)
    std::vector<int> devs;
    if (const char *e = std::getenv("TINYMPC_HIP_SHARD_DEVICES")) {
        for (const char *p = e; *p;) {
            devs.push_back(std::atoi(p));
            while (*p && *p != ',') ++p;
            if (*p == ',') ++p;
        }
        if ((int)devs.size() != n_gpus) {
            set_error("set_gpus: TINYMPC_HIP_SHARD_DEVICES does not list n_gpus devices");
            return -1;
        }
    }
    tinympc_sharded *sh = nullptr;
    if (tinympc_create_sharded(&sh, m.A.a.data(), m.B.a.data(), m.Q.a.data(), m.R.a.data(), m.cache.rho, m.nx, m.nu, m.N,
                               batch, n_gpus, devs.empty() ? nullptr : devs.data(), g_verbose))
        return -1;
    // (a failure above leaves the previous arrangement untouched)
    g_sharded.reset(sh);
    g_sharded_batch = batch;
    for (int i = 0; i < n_gpus; ++i)
        if (global_shard(i)->s.copy_family_state(m)) {
            g_sharded.reset();
            (void)m.alloc_batch(batch);
            return -1;
        }
    return m.batch == 1 ? 0 : m.alloc_batch(1);
}

template <class F>
int guarded(const char *what, F &&f) {
    try {
        return f();
    } catch (const std::exception &e) {
        set_error(std::string(what) + " failed: " + e.what());
        return -1;
    } catch (...) {
        set_error(std::string(what) + " failed: unknown exception");
        return -1;
    }
}

int need_global(const char *what) {
    if (!g_solver) {
        set_error(std::string(what) + " failed: Solver not initialized");  // bindings.cpp:77-79
        return -1;
    }
    return 0;
}

bool dims_ok(const char *name, int r, int c, int er, int ec) {
    if (r == er && c == ec) return true;
    char buf[160];
    std::snprintf(buf, sizeof buf, "%s has %d x %d, expected %d x %d", name, r, c, er, ec);
    set_error(buf);
    return false;
}

int sync_status(tinympc_solver *s, hipStream_t stream) {
    if (!tmpc::hip_ok(hipStreamSynchronize(stream), "hipStreamSynchronize")) return -1;
    return s->s.solve_status();
}

}  // namespace

extern "C" {

/* ------------------------------ handle API ------------------------------ */

int tinympc_create(tinympc_solver **out, const double *A, const double *B, const double *Q,
                   const double *R, double rho, int nx, int nu, int N, int batch, int device,
                   int verbose) {
    return guarded("tinympc_create", [&]() -> int {
        if (!out || !A || !B || !Q || !R) {
            set_error("tinympc_create: null argument");
            return -1;
        }
        std::unique_ptr<tinympc_solver> s(new tinympc_solver());
        if (s->s.init(A, B, Q, R, rho, nx, nu, N, batch, device, verbose)) return -1;
        *out = s.release();
        return 0;
    });
}

int tinympc_create_families(tinympc_solver **out, const double *A, const double *B, const double *Q,
                            const double *R, const double *rho, int nx, int nu, int N, int batch, int device,
                            int verbose) {
    return guarded("tinympc_create_families", [&]() -> int {
        if (!out || !A || !B || !Q || !R || !rho) {
            set_error("tinympc_create_families: null argument");
            return -1;
        }
        std::unique_ptr<tinympc_solver> s(new tinympc_solver());
        if (s->s.init_families(A, B, Q, R, rho, nx, nu, N, batch, device, verbose)) return -1;
        *out = s.release();
        return 0;
    });
}

void tinympc_destroy(tinympc_solver *s) { delete s; }

int tinympc_update_settings(tinympc_solver *s, double abs_pri_tol, double abs_dua_tol, int max_iter,
                            int check_termination, int en_state_bound, int en_input_bound) {
    if (!s) return -1;
    if (max_iter < 0) {
        set_error("update_settings: max_iter < 0");
        return -1;
    }
    tmpc::Settings &st = s->s.st;
    const bool flags_changed = st.en_state_bound != en_state_bound || st.en_input_bound != en_input_bound;
    st.abs_pri_tol = abs_pri_tol;
    st.abs_dua_tol = abs_dua_tol;
    st.max_iter = max_iter;
    st.check_termination = check_termination;
    st.en_state_bound = en_state_bound ? 1 : 0;
    st.en_input_bound = en_input_bound ? 1 : 0;
    if (flags_changed) s->s.packs_dirty = true, s->s.route_gen += 1;
    return 0;
}

int tinympc_enable_cones(tinympc_solver *s, int en_state_soc, int en_input_soc) {
    if (!s) return -1;
    s->s.st.en_state_soc = en_state_soc ? 1 : 0;
    s->s.st.en_input_soc = en_input_soc ? 1 : 0;
    s->s.route_gen += 1;
    return 0;
}

int tinympc_set_bound_constraints(tinympc_solver *s, const double *x_min, const double *x_max,
                                  const double *u_min, const double *u_max) {
    if (!s || !x_min || !x_max || !u_min || !u_max) return -1;
    return guarded("set_bound_constraints", [&] { return s->s.set_bounds(x_min, x_max, u_min, u_max); });
}

int tinympc_set_fdyn(tinympc_solver *s, const double *fdyn) {
    if (!s) return -1;
    return guarded("set_fdyn", [&] { return s->s.set_fdyn(fdyn); });
}

int tinympc_enable_linear(tinympc_solver *s, int en_state_linear, int en_input_linear) {
    if (!s) return -1;
    s->s.st.en_state_linear = en_state_linear ? 1 : 0;
    s->s.st.en_input_linear = en_input_linear ? 1 : 0;
    s->s.lin_dirty = true;   // the device pack holds the enabled sides only
    s->s.route_gen += 1;
    return 0;
}

int tinympc_set_linear_constraints(tinympc_solver *s, const double *Alin_x, int rows_x, const double *blin_x,
                                   const double *Alin_u, int rows_u, const double *blin_u) {
    if (!s) return -1;
    return guarded("set_linear_constraints",
                   [&] { return s->s.set_linear(Alin_x, rows_x, blin_x, Alin_u, rows_u, blin_u); });
}

int tinympc_set_cone_constraints(tinympc_solver *s, const int *Acu, const int *qcu, const double *cu,
                                 int n_input_cones, const int *Acx, const int *qcx, const double *cx,
                                 int n_state_cones) {
    if (!s || n_input_cones < 0 || n_state_cones < 0) return -1;
    return guarded("set_cone_constraints",
                   [&] { return s->s.set_cones(Acu, qcu, cu, n_input_cones, Acx, qcx, cx, n_state_cones); });
}

int tinympc_set_cache_terms(tinympc_solver *s, const double *Kinf, const double *Pinf,
                            const double *Quu_inv, const double *AmBKt) {
    if (!s || !Kinf || !Pinf || !Quu_inv || !AmBKt) return -1;
    tmpc::Solver &v = s->s;
    if (v.hetero) {
        set_error("set_cache_terms: not available on a per-instance-family solver");
        return -1;
    }
    v.cache.Kinf = tmpc::Mat(v.nu, v.nx, Kinf);
    v.cache.Pinf = tmpc::Mat(v.nx, v.nx, Pinf);
    v.cache.Quu_inv = tmpc::Mat(v.nu, v.nu, Quu_inv);
    v.cache.AmBKt = tmpc::Mat(v.nx, v.nx, AmBKt);
    v.packs_dirty = true;
    v.adapt_dirty = true;  // adaptive rho restarts from the new cache
    v.cache_overridden = true;  // AmBKt may no longer equal (A - B Kinf)': the quad kernel's adaptive variant relies on that identity
    return v.select_kernel();
}

int tinympc_set_adaptive_rho(tinympc_solver *s, int enable, double rho_min, double rho_max, int enable_clipping) {
    if (!s) return -1;
    tmpc::Solver &v = s->s;
    if (enable && v.hetero) {
        set_error("adaptive_rho is not available on a per-instance-family solver");
        return -1;
    }
    if (enable && !(rho_min > 0.0 && rho_max >= rho_min)) {
        set_error("adaptive_rho: need 0 < rho_min <= rho_max");
        return -1;
    }
    if ((enable != 0) != (v.st.adaptive_rho != 0)) v.adapt_dirty = true;
    v.st.adaptive_rho = enable ? 1 : 0;
    v.st.adaptive_rho_min = rho_min;
    v.st.adaptive_rho_max = rho_max;
    v.st.adaptive_rho_clip = enable_clipping ? 1 : 0;
    return 0;
}

int tinympc_set_sensitivity(tinympc_solver *s, const double *dKinf, const double *dPinf, const double *dC1,
                            const double *dC2) {
    (void)dC1;  // accepted for the reference's call shape; they only ever reach dead copies there
    (void)dC2;
    if (!s || !dKinf || !dPinf) return -1;
    return guarded("set_sensitivity", [&] { return s->s.set_sensitivity(dKinf, dPinf); });
}

int tinympc_compute_sensitivity(tinympc_solver *s, double *dKinf, double *dPinf, double *dC1, double *dC2) {
    if (!s) return -1;
    return guarded("compute_sensitivity", [&] {
        tmpc::Solver &v = s->s;
        tmpc::Mat dK, dP, d1, d2;
        if (tmpc::compute_sensitivity(v.A, v.B, v.Q, v.R, v.cache.rho, dK, dP, d1, d2)) {
            set_error("compute_sensitivity: singular R + rho I + B'PB");
            return -1;
        }
        if (dKinf) std::copy(dK.a.begin(), dK.a.end(), dKinf);
        if (dPinf) std::copy(dP.a.begin(), dP.a.end(), dPinf);
        if (dC1) std::copy(d1.a.begin(), d1.a.end(), dC1);
        if (dC2) std::copy(d2.a.begin(), d2.a.end(), dC2);
        return 0;
    });
}

int tinympc_get_adaptive_state(tinympc_solver *s, double *rho, double *Kinf, double *Pinf) {
    if (!s) return -1;
    return guarded("get_adaptive_state", [&] { return s->s.get_adaptive_state(rho, Kinf, Pinf); });
}

int tinympc_get_cache_terms(tinympc_solver *s, double *Kinf, double *Pinf, double *Quu_inv,
                            double *AmBKt) {
    if (!s) return -1;
    const tmpc::Cache &c = s->s.cache;
    if (Kinf) std::copy(c.Kinf.a.begin(), c.Kinf.a.end(), Kinf);
    if (Pinf) std::copy(c.Pinf.a.begin(), c.Pinf.a.end(), Pinf);
    if (Quu_inv) std::copy(c.Quu_inv.a.begin(), c.Quu_inv.a.end(), Quu_inv);
    if (AmBKt) std::copy(c.AmBKt.a.begin(), c.AmBKt.a.end(), AmBKt);
    return 0;
}

int tinympc_set_x0(tinympc_solver *s, const double *x0, int cols) {
    if (!s || !x0) return -1;
    return guarded("set_x0", [&] { return s->s.set_x0(x0, cols); });
}
int tinympc_set_x_ref(tinympc_solver *s, const double *x_ref, int cols) {
    if (!s || !x_ref) return -1;
    return guarded("set_x_ref", [&] { return s->s.set_ref(true, x_ref, cols); });
}
int tinympc_set_u_ref(tinympc_solver *s, const double *u_ref, int cols) {
    if (!s || !u_ref) return -1;
    return guarded("set_u_ref", [&] { return s->s.set_ref(false, u_ref, cols); });
}
int tinympc_reset(tinympc_solver *s) { return s ? s->s.reset() : -1; }
int tinympc_set_warm_start(tinympc_solver *s, int warm_start) {
    if (!s) return -1;
    s->s.warm_start = warm_start != 0;
    return 0;
}

int tinympc_solve(tinympc_solver *s) {
    if (!s) return -1;
    return guarded("solve", [&]() -> int {
        if (s->s.solve_async(nullptr)) return -1;
        return sync_status(s, nullptr);
    });
}
int tinympc_solve_async(tinympc_solver *s, void *hip_stream) {
    if (!s) return -1;
    return guarded("solve_async", [&] { return s->s.solve_async((hipStream_t)hip_stream); });
}
int tinympc_solve_status(tinympc_solver *s) { return s ? s->s.solve_status() : -1; }

int tinympc_mpc_rollout(tinympc_solver *s, int steps, void *hip_stream) {
    if (!s || steps < 1) return -1;
    return guarded("mpc_rollout", [&]() -> int {
        if (s->s.solve_async((hipStream_t)hip_stream, steps)) return -1;
        return sync_status(s, (hipStream_t)hip_stream);
    });
}
int tinympc_set_ref_sequence(tinympc_solver *s, const double *x_ref_seq, int x_rows, int x_cols, const double *u_ref_seq,
                             int u_rows, int u_cols, int steps) {
    if (!s) return -1;
    return guarded("set_ref_sequence", [&]() -> int {
        if (steps > 0 && (x_rows != s->s.nx || u_rows != s->s.nu || (long)x_cols != (long)s->s.N * steps ||
                          (long)u_cols != (long)(s->s.N - 1) * steps)) {
            tmpc::set_error("set_ref_sequence: expected nx x (N steps) and nu x ((N-1) steps)");
            return -1;
        }
        return s->s.set_ref_sequence(x_ref_seq, u_ref_seq, steps);
    });
}
int tinympc_get_mpc_log(tinympc_solver *s, double *x, double *u, int *iter) {
    if (!s) return -1;
    return guarded("get_mpc_log", [&] { return s->s.get_mpc_log(x, u, iter); });
}

int tinympc_get_states(tinympc_solver *s, double *buf) {
    if (!s || !buf) return -1;
    return guarded("get_states", [&] { return s->s.get_traj(true, buf); });
}
int tinympc_get_controls(tinympc_solver *s, double *buf) {
    if (!s || !buf) return -1;
    return guarded("get_controls", [&] { return s->s.get_traj(false, buf); });
}
// fp32 host arrays: the device buffers are fp32, so these are plain copies — no narrowing / widening pass on the host
// (the fp64 forms spend most of a round trip on it: DESIGN.md, PCIe-inclusive rate)
int tinympc_set_x0_f32(tinympc_solver *s, const float *x0, int cols) {
    if (!s || !x0) return -1;
    return guarded("set_x0_f32", [&]() -> int {
        tmpc::Solver &v = s->s;
        if (cols != 1 && cols != v.batch) {
            tmpc::set_error("set_x0_f32: expected nx x 1 or nx x batch");
            return -1;
        }
        if (!tmpc::hip_ok(hipSetDevice(v.device), "hipSetDevice") || v.wait_last_launch()) return -1;
        if (cols == v.batch) {
            return tmpc::hip_ok(hipMemcpy(v.d_x0, x0, (size_t)v.batch * v.nx * sizeof(float), hipMemcpyHostToDevice), "hipMemcpy") ? 0 : -1;
        }
        std::vector<float> h((size_t)v.batch * v.nx);
        for (int b = 0; b < v.batch; ++b) std::copy(x0, x0 + v.nx, h.begin() + (size_t)b * v.nx);
        return tmpc::hip_ok(hipMemcpy(v.d_x0, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice), "hipMemcpy") ? 0 : -1;
    });
}
static int get_traj_f32(tinympc_solver *s, bool states, float *buf) {
    tmpc::Solver &v = s->s;
    if (!tmpc::hip_ok(hipSetDevice(v.device), "hipSetDevice") || v.wait_last_launch()) return -1;
    const size_t n = (size_t)v.batch * (states ? v.ex() : v.eu());
    return tmpc::hip_ok(hipMemcpy(buf, states ? v.d_xout : v.d_uout, n * sizeof(float), hipMemcpyDeviceToHost), "hipMemcpy") ? 0 : -1;
}
int tinympc_get_states_f32(tinympc_solver *s, float *buf) {
    if (!s || !buf) return -1;
    return guarded("get_states_f32", [&] { return get_traj_f32(s, true, buf); });
}
int tinympc_get_controls_f32(tinympc_solver *s, float *buf) {
    if (!s || !buf) return -1;
    return guarded("get_controls_f32", [&] { return get_traj_f32(s, false, buf); });
}
int tinympc_pin_host(tinympc_solver *s, void *ptr, size_t bytes) {
    if (!s) return -1;
    return guarded("pin_host", [&] { return s->s.pin_host_range(ptr, bytes); });
}
int tinympc_unpin_host(tinympc_solver *s, void *ptr) {
    if (!s) return -1;
    return guarded("unpin_host", [&] { return s->s.unpin_host_range(ptr); });
}
int tinympc_get_status(tinympc_solver *s, int *iter, int *solved, double *residuals4) {
    if (!s) return -1;
    return guarded("get_status", [&] { return s->s.get_status(iter, solved, residuals4); });
}
int tinympc_get_workspace(tinympc_solver *s, double *d, double *y, double *g, double *v, double *z) {
    if (!s) return -1;
    return guarded("get_workspace", [&] { return s->s.get_workspace(d, y, g, v, z); });
}
int tinympc_set_workspace(tinympc_solver *s, const double *d, const double *y, const double *g,
                          const double *v, const double *z) {
    if (!s) return -1;
    return guarded("set_workspace", [&] { return s->s.set_workspace(d, y, g, v, z); });
}

int tinympc_device_buffers(tinympc_solver *s, void **x0, void **x_ref, void **u_ref, void **states,
                           void **controls, void **iter, void **solved, void **residuals,
                           void **gstat) {
    if (!s) return -1;
    tmpc::Solver &v = s->s;
    if (x0) *x0 = v.d_x0;
    if (x_ref) *x_ref = v.d_xref;
    if (u_ref) *u_ref = v.d_uref;
    if (states) *states = v.d_xout;
    if (controls) *controls = v.d_uout;
    if (iter) *iter = v.d_iter;
    if (solved) *solved = v.d_solved;
    if (residuals) *residuals = v.d_res;
    if (gstat) *gstat = v.d_gstat;
    return 0;
}

int tinympc_set_ref_mode(tinympc_solver *s, int ref_mode) {
    if (!s || ref_mode < 0 || ref_mode > 2) return -1;
    return guarded("set_ref_mode", [&]() -> int {
        tmpc::Solver &v = s->s;
        // make the device buffers large enough for the mode, then hand them to the caller
        v.xref_kind = v.uref_kind = ref_mode;
        v.h_xref.assign(ref_mode == 2 ? (size_t)v.batch * v.ex() : (ref_mode == 1 ? v.ex() : 0), 0.f);
        v.h_uref.assign(ref_mode == 2 ? (size_t)v.batch * v.eu() : (ref_mode == 1 ? v.eu() : 0), 0.f);
        v.refs_dirty = true;
        v.refs_device_owned = false;
        if (v.upload_refs()) return -1;
        v.h_xref.clear();
        v.h_uref.clear();
        v.refs_device_owned = true;
        return 0;
    });
}

int tinympc_set_compaction(tinympc_solver *s, int chunk_iters) {
    if (!s || chunk_iters < 0) return -1;
    s->s.chunk_iters = chunk_iters;
    return 0;
}
int tinympc_set_profiling(tinympc_solver *s, int enable) {
    if (!s) return -1;
    s->s.profiling = enable != 0;
    return 0;
}
double tinympc_kernel_elapsed_ms(tinympc_solver *s) { return s ? s->s.kernel_elapsed_ms() : -1.0; }
double tinympc_kernel_elapsed_mean_ms(tinympc_solver *s, int last_n) {
    return s ? s->s.kernel_elapsed_mean_ms(last_n) : -1.0;
}
int tinympc_set_precision(tinympc_solver *s, int precision) {
    if (!s || precision < 0 || precision > 2) return -1;
    if (precision == 2 && s->s.hetero) {
        set_error("set_precision: precision 2 is not available on a per-instance-family solver");
        return -1;
    }
    if (s->s.precision != precision) {
        s->s.packs_dirty = true;
        // the workspace kept between solves does not carry over between the fp32 arrays and the fp64 block: cold restart
        if ((s->s.precision == 2) != (precision == 2) && s->s.solved_once && s->s.reset()) return -1;
    }
    s->s.precision = precision;
    return 0;
}

// tuning / test aid: the environment switches are read when a solver is created; this re-reads them for a live solver
int tinympc_reload_switches(tinympc_solver *s) {
    if (!s) return -1;
    s->s.sw = tmpc::read_switches();
    s->s.lean_enabled = !s->s.sw.no_lean;
    s->s.route_gen += 1;
    s->s.packs_dirty = true;
    return 0;
}
int tinympc_set_strict_precision(tinympc_solver *s, int strict) {
    if (!s) return -1;
    s->s.strict_precision = strict != 0;
    return 0;
}
int tinympc_effective_precision(tinympc_solver *s) {
    if (!s) return -1;
    tmpc::Solver &v = s->s;
    if (v.select_kernel()) return -1;
    // the matrix-core kernels and the lean kernel only exist with fp64 recurrences; every other family follows `precision`
    const bool matrix = (v.ke && v.ke->G == 16) || v.ce != nullptr;
    return v.precision == 2 ? 2 : (matrix ? 0 : v.precision);
}
const char *tinympc_kernel_name(tinympc_solver *s) { return s ? s->s.kernel_name.c_str() : ""; }
const char *tinympc_last_launch_name(tinympc_solver *s) {
    if (!s) return "";
    return s->s.last_launch_name.empty() ? s->s.kernel_name.c_str() : s->s.last_launch_name.c_str();
}

/* compulsory fp32 device I/O per solve, state on chip. */
double tinympc_algorithmic_bytes(tinympc_solver *s) {
    if (!s) return 0;
    const tmpc::Solver &v = s->s;
    const double EX = v.ex(), EU = v.eu();
    double per = 4.0 * v.nx + 4.0 * (EX + EU) + 24.0;             // x0 in, x/u out, iter/solved/4 res
    if (v.ref_mode == tmpc::REF_PER_INSTANCE) per += 4.0 * (EX + EU);
    if (v.warm_start) per += 2.0 * 4.0 * (3.0 * EU + 2.0 * EX);   // d,y,z,g,v in and out
    return per * v.batch;
}

/* 2(N-1)(2nx^2+4nx nu+nu^2) + (N-1)(nu+2nx) + 15(Ex+Eu) + 2nx^2 + 3nx per iteration. */
double tinympc_algorithmic_flops(tinympc_solver *s, int iters) {
    if (!s) return 0;
    const tmpc::Solver &v = s->s;
    const double nx = v.nx, nu = v.nu, N = v.N, EX = v.ex(), EU = v.eu();
    const double per_it = 2.0 * (N - 1) * (2 * nx * nx + 4 * nx * nu + nu * nu) + (N - 1) * (nu + 2 * nx) +
                          15.0 * (EX + EU) + 2 * nx * nx + 3 * nx;
    return per_it * iters * v.batch;
}

const char *tinympc_last_error(void) { return tmpc::last_error(); }

// Specialise the on-chip kernel of a shape now (what tinympc_create does by itself for a shape the library was not built with):
// 1 = an on-chip kernel exists for the shape afterwards (built in, cached, or just compiled), 0 = none (run-time-shape kernels)
int tinympc_specialise(int nx, int nu, int N, int verbose) {
    if (tmpc::find_quad_kernel(nx, nu, N, -1) || tmpc::find_mfma_kernel(nx, nu, N) || tmpc::find_trans_kernel(nx, nu, N)) return 1;
    return tmpc::jit_kernel_for(nx, nu, N, verbose) != nullptr ? 1 : 0;
}

int tinympc_host_precompute(const double *A, const double *B, const double *Q, const double *R,
                            double rho, int nx, int nu, double *Kinf, double *Pinf,
                            double *Quu_inv, double *AmBKt) {
    return guarded("host_precompute", [&]() -> int {
        if (!A || !B || !Q || !R || nx < 1 || nu < 1) return -1;
        tmpc::Cache c;
        if (tmpc::precompute_cache(tmpc::Mat(nx, nx, A), tmpc::Mat(nx, nu, B), tmpc::Mat(nx, nx, Q),
                                   tmpc::Mat(nu, nu, R), rho, c)) {
            set_error("Riccati precompute failed: R + B'PB is singular");
            return -1;
        }
        if (Kinf) std::copy(c.Kinf.a.begin(), c.Kinf.a.end(), Kinf);
        if (Pinf) std::copy(c.Pinf.a.begin(), c.Pinf.a.end(), Pinf);
        if (Quu_inv) std::copy(c.Quu_inv.a.begin(), c.Quu_inv.a.end(), Quu_inv);
        if (AmBKt) std::copy(c.AmBKt.a.begin(), c.AmBKt.a.end(), AmBKt);
        return 0;
    });
}

int tinympc_host_sensitivity(const double *A, const double *B, const double *Q, const double *R, double rho, int nx,
                             int nu, double *dKinf, double *dPinf, double *dC1, double *dC2) {
    return guarded("host_sensitivity", [&]() -> int {
        if (!A || !B || !Q || !R || nx < 1 || nu < 1) return -1;
        tmpc::Mat dK, dP, d1, d2;
        if (tmpc::compute_sensitivity(tmpc::Mat(nx, nx, A), tmpc::Mat(nx, nu, B), tmpc::Mat(nx, nx, Q),
                                      tmpc::Mat(nu, nu, R), rho, dK, dP, d1, d2)) {
            set_error("host_sensitivity: singular R + rho I + B'PB");
            return -1;
        }
        if (dKinf) std::copy(dK.a.begin(), dK.a.end(), dKinf);
        if (dPinf) std::copy(dP.a.begin(), dP.a.end(), dPinf);
        if (dC1) std::copy(d1.a.begin(), d1.a.end(), dC1);
        if (dC2) std::copy(d2.a.begin(), d2.a.end(), dC2);
        return 0;
    });
}

/* --------------------- process-global solver (drop-in) --------------------- */

int setup_solver(double *A_data, int A_rows, int A_cols, double *B_data, int B_rows, int B_cols,
                 double *fdyn_data, int fdyn_rows, int fdyn_cols, double *Q_data, int Q_rows,
                 int Q_cols, double *R_data, int R_rows, int R_cols, double rho, int nx, int nu,
                 int N, int verbose) {
    return guarded("setup_solver", [&]() -> int {
        g_sharded.reset();
        g_verbose = verbose;
        if (!A_data || !B_data || !Q_data || !R_data) {
            set_error("setup_solver: null matrix");
            g_solver.reset();
            return -1;
        }
        if (!dims_ok("A", A_rows, A_cols, nx, nx) || !dims_ok("B", B_rows, B_cols, nx, nu) ||
            !dims_ok("Q", Q_rows, Q_cols, nx, nx) || !dims_ok("R", R_rows, R_cols, nu, nu)) {
            g_solver.reset();
            return -1;
        }
        if (fdyn_data && (long)fdyn_rows * fdyn_cols != nx) {
            set_error("setup_solver: fdyn must have nx entries");
            g_solver.reset();
            return -1;
        }
        tinympc_solver *s = nullptr;
        if (tinympc_create(&s, A_data, B_data, Q_data, R_data, rho, nx, nu, N, 1, -1, verbose)) {
            g_solver.reset();
            return -1;
        }
        if (fdyn_data && tinympc_set_fdyn(s, fdyn_data)) {
            tinympc_destroy(s);
            g_solver.reset();
            return -1;
        }
        g_solver.reset(s);  // a second setup replaces the first (bindings.cpp:63)
        return 0;
    });
}

int set_batch_size(int batch) {
    if (need_global("set_batch_size")) return -1;
    return guarded("set_batch_size", [&] {
        return g_sharded ? reshard_global(batch, tinympc_sharded_n_shards(g_sharded.get())) : g_solver->s.alloc_batch(batch);
    });
}
int get_batch_size(void) { return global_batch(); }

int set_gpus(int n_gpus) {
    if (need_global("set_gpus")) return -1;
    if (n_gpus < 1) {
        set_error("set_gpus: n_gpus must be >= 1");
        return -1;
    }
    return guarded("set_gpus", [&] { return reshard_global(global_batch(), n_gpus); });
}
int get_gpus(void) { return g_sharded ? tinympc_sharded_n_shards(g_sharded.get()) : (g_solver ? 1 : 0); }

// warm_start = 0: every solve() of the global solver starts from the zero workspace and keeps none ("one-shot": the
// benchmark configs' regime, and what lets the on-chip kernels run); 1 (default): the reference's semantics, the
// workspace persists between solves (admm.cpp:112-115 resets only the counters)
int set_warm_start(int warm_start) {
    if (need_global("set_warm_start")) return -1;
    return each_global([&](tinympc_solver *h) { return tinympc_set_warm_start(h, warm_start); });
}
// recurrence / state precision of the global solver (tinympc_set_precision): 0 default, 1 all fp32, 2 all fp64 like the reference
int set_precision(int precision) {
    if (need_global("set_precision")) return -1;
    return each_global([&](tinympc_solver *h) { return tinympc_set_precision(h, precision); });
}
const char *get_kernel_name(void) {
    if (!g_solver) return "";
    return (g_sharded ? global_shard(0)->s : g_solver->s).kernel_name.c_str();
}

int set_x0(double *x0_data, int x0_rows, int x0_cols, int verbose) {
    (void)verbose;
    if (need_global("set_x0")) return -1;
    if (x0_rows != g_solver->s.nx) {
        set_error("set_x0: x0 is not the correct length");
        return -1;
    }
    if (g_sharded) return guarded("set_x0", [&] { return tinympc_sharded_set_x0(g_sharded.get(), x0_data, x0_cols); });
    return tinympc_set_x0(g_solver.get(), x0_data, x0_cols);
}
int set_x_ref(double *x_ref_data, int x_ref_rows, int x_ref_cols, int verbose) {
    (void)verbose;
    if (need_global("set_x_ref")) return -1;
    if (x_ref_rows != g_solver->s.nx) {
        set_error("set_x_ref: wrong number of rows");
        return -1;
    }
    if (g_sharded) return guarded("set_x_ref", [&] { return tinympc_sharded_set_x_ref(g_sharded.get(), x_ref_data, x_ref_cols); });
    return tinympc_set_x_ref(g_solver.get(), x_ref_data, x_ref_cols);
}
int set_u_ref(double *u_ref_data, int u_ref_rows, int u_ref_cols, int verbose) {
    (void)verbose;
    if (need_global("set_u_ref")) return -1;
    if (u_ref_rows != g_solver->s.nu) {
        set_error("set_u_ref: wrong number of rows");
        return -1;
    }
    if (g_sharded) return guarded("set_u_ref", [&] { return tinympc_sharded_set_u_ref(g_sharded.get(), u_ref_data, u_ref_cols); });
    return tinympc_set_u_ref(g_solver.get(), u_ref_data, u_ref_cols);
}

int solve_mpc(int verbose) {
    if (need_global("solve_mpc")) return -1;
    const int st = g_sharded ? guarded("solve_mpc", [&] { return tinympc_sharded_solve(g_sharded.get()); })
                             : tinympc_solve(g_solver.get());
    if (verbose) std::printf("Solve completed with status: %d\n", st);
    return st;
}

int get_states(double *states_buffer, int *rows, int *cols) {
    if (!g_solver || !states_buffer || !rows || !cols) return -1;
    *rows = g_solver->s.nx;
    *cols = g_solver->s.N * global_batch();
    if (g_sharded) return guarded("get_states", [&] { return tinympc_sharded_get_states(g_sharded.get(), states_buffer); });
    return tinympc_get_states(g_solver.get(), states_buffer);
}
int get_controls(double *controls_buffer, int *rows, int *cols) {
    if (!g_solver || !controls_buffer || !rows || !cols) return -1;
    *rows = g_solver->s.nu;
    *cols = (g_solver->s.N - 1) * global_batch();
    if (g_sharded) return guarded("get_controls", [&] { return tinympc_sharded_get_controls(g_sharded.get(), controls_buffer); });
    return tinympc_get_controls(g_solver.get(), controls_buffer);
}

// fp32 forms of the three per-solve transfers for a Julia host that keeps Float32 arrays (not in bindings.cpp: an addition
// next to its names; one device only — a sharded global solver takes the fp64 forms)
int set_x0_f32(float *x0_data, int x0_rows, int x0_cols, int verbose) {
    (void)verbose;
    if (need_global("set_x0_f32")) return -1;
    if (x0_rows != g_solver->s.nx || g_sharded) {
        set_error(g_sharded ? "set_x0_f32: not available on a sharded solver" : "set_x0_f32: x0 is not the correct length");
        return -1;
    }
    return tinympc_set_x0_f32(g_solver.get(), x0_data, x0_cols);
}
int get_states_f32(float *states_buffer, int *rows, int *cols) {
    if (!g_solver || g_sharded || !states_buffer || !rows || !cols) return -1;
    *rows = g_solver->s.nx;
    *cols = g_solver->s.N * global_batch();
    return tinympc_get_states_f32(g_solver.get(), states_buffer);
}
int get_controls_f32(float *controls_buffer, int *rows, int *cols) {
    if (!g_solver || g_sharded || !controls_buffer || !rows || !cols) return -1;
    *rows = g_solver->s.nu;
    *cols = (g_solver->s.N - 1) * global_batch();
    return tinympc_get_controls_f32(g_solver.get(), controls_buffer);
}

// page-lock / release a host array the Julia caller owns and reuses with the fp32 forms (tinympc_pin_host); the caller
// unpins before the array can be collected
int pin_host_buffer(void *ptr, size_t bytes) {
    if (need_global("pin_host_buffer")) return -1;
    if (g_sharded) {
        set_error("pin_host_buffer: not available on a sharded solver");
        return -1;
    }
    return tinympc_pin_host(g_solver.get(), ptr, bytes);
}
int unpin_host_buffer(void *ptr) {
    if (need_global("unpin_host_buffer")) return -1;
    if (g_sharded) {
        set_error("unpin_host_buffer: not available on a sharded solver");
        return -1;
    }
    return tinympc_unpin_host(g_solver.get(), ptr);
}

void cleanup_solver(void) {
    g_sharded.reset();
    g_solver.reset();
}

int update_settings(double abs_pri_tol, double abs_dua_tol, int max_iter, int check_termination,
                    int en_state_bound, int en_input_bound, int en_state_soc, int en_input_soc,
                    int en_state_linear, int en_input_linear, int adaptive_rho,
                    double adaptive_rho_min, double adaptive_rho_max,
                    int adaptive_rho_enable_clipping, int verbose) {
    (void)verbose;
    if (need_global("update_settings")) return -1;
    return each_global([&](tinympc_solver *h) -> int {
        tinympc_enable_cones(h, en_state_soc, en_input_soc);
        tinympc_enable_linear(h, en_state_linear, en_input_linear);
        if (tinympc_set_adaptive_rho(h, adaptive_rho, adaptive_rho_min, adaptive_rho_max, adaptive_rho_enable_clipping))
            return -1;
        return tinympc_update_settings(h, abs_pri_tol, abs_dua_tol, max_iter, check_termination, en_state_bound,
                                       en_input_bound);
    });
}

int set_bound_constraints(double *x_min_data, int x_min_rows, int x_min_cols, double *x_max_data,
                          int x_max_rows, int x_max_cols, double *u_min_data, int u_min_rows,
                          int u_min_cols, double *u_max_data, int u_max_rows, int u_max_cols,
                          int verbose) {
    (void)verbose;
    if (need_global("set_bound_constraints")) return -1;
    const tmpc::Solver &v = g_solver->s;
    if (!dims_ok("x_min", x_min_rows, x_min_cols, v.nx, v.N) ||
        !dims_ok("x_max", x_max_rows, x_max_cols, v.nx, v.N) ||
        !dims_ok("u_min", u_min_rows, u_min_cols, v.nu, v.N - 1) ||
        !dims_ok("u_max", u_max_rows, u_max_cols, v.nu, v.N - 1))
        return -1;
    return each_global(
        [&](tinympc_solver *h) { return tinympc_set_bound_constraints(h, x_min_data, x_max_data, u_min_data, u_max_data); });
}

int set_cache_terms(double *Kinf_data, int Kinf_rows, int Kinf_cols, double *Pinf_data,
                    int Pinf_rows, int Pinf_cols, double *Quu_inv_data, int Quu_inv_rows,
                    int Quu_inv_cols, double *AmBKt_data, int AmBKt_rows, int AmBKt_cols,
                    int verbose) {
    (void)verbose;
    if (need_global("set_cache_terms")) return -1;
    const tmpc::Solver &v = g_solver->s;
    if (!dims_ok("Kinf", Kinf_rows, Kinf_cols, v.nu, v.nx) ||
        !dims_ok("Pinf", Pinf_rows, Pinf_cols, v.nx, v.nx) ||
        !dims_ok("Quu_inv", Quu_inv_rows, Quu_inv_cols, v.nu, v.nu) ||
        !dims_ok("AmBKt", AmBKt_rows, AmBKt_cols, v.nx, v.nx))
        return -1;
    return each_global(
        [&](tinympc_solver *h) { return tinympc_set_cache_terms(h, Kinf_data, Pinf_data, Quu_inv_data, AmBKt_data); });
}

int set_sensitivity(double *dK_data, int dK_rows, int dK_cols, double *dP_data, int dP_rows, int dP_cols,
                    double *dC1_data, int dC1_rows, int dC1_cols, double *dC2_data, int dC2_rows, int dC2_cols,
                    int verbose) {
    (void)verbose;
    if (need_global("set_sensitivity")) return -1;
    const tmpc::Solver &v = g_solver->s;
    if (!dims_ok("dK", dK_rows, dK_cols, v.nu, v.nx) || !dims_ok("dP", dP_rows, dP_cols, v.nx, v.nx) ||
        (dC1_data && !dims_ok("dC1", dC1_rows, dC1_cols, v.nu, v.nu)) ||
        (dC2_data && !dims_ok("dC2", dC2_rows, dC2_cols, v.nx, v.nx)))
        return -1;
    return each_global([&](tinympc_solver *h) { return tinympc_set_sensitivity(h, dK_data, dP_data, dC1_data, dC2_data); });
}

int get_adaptive_rho(double *rho_buffer, int *count) {
    if (need_global("get_adaptive_rho") || !rho_buffer || !count) return -1;
    *count = global_batch();
    if (g_sharded) {
        for (int i = 0; i < tinympc_sharded_n_shards(g_sharded.get()); ++i) {
            int lo = 0;
            tinympc_sharded_shard(g_sharded.get(), i, nullptr, &lo, nullptr, nullptr);
            if (tinympc_get_adaptive_state(global_shard(i), rho_buffer + lo, nullptr, nullptr)) return -1;
        }
        return 0;
    }
    return tinympc_get_adaptive_state(g_solver.get(), rho_buffer, nullptr, nullptr);
}

int print_problem_data(int verbose) {
    if (need_global("print_problem_data")) return -1;
    const tmpc::Solver &v = g_solver->s;
    std::printf("=== TinyMPC Problem Data ===\n");
    std::printf("Problem: nx=%d, nu=%d, N=%d, batch=%d, gpus=%d, kernel=%s\n", v.nx, v.nu, v.N, global_batch(), get_gpus(),
                (g_sharded ? global_shard(0)->s : v).kernel_name.c_str());
    std::printf("Cache: rho=%g\n", v.cache.rho);
    std::printf("Settings: max_iter=%d, abs_pri_tol=%g, abs_dua_tol=%g, check_termination=%d\n",
                v.st.max_iter, v.st.abs_pri_tol, v.st.abs_dua_tol, v.st.check_termination);
    if (verbose) {
        std::printf("Cache Kinf (%d x %d):\n", v.nu, v.nx);
        for (int i = 0; i < v.nu; ++i) {
            for (int j = 0; j < v.nx; ++j) std::printf(" %.10g", v.cache.Kinf(i, j));
            std::printf("\n");
        }
    }
    return 0;
}

int set_linear_constraints(double *Alin_x_data, int Alin_x_rows, int Alin_x_cols, double *blin_x_data,
                           int blin_x_len, double *Alin_u_data, int Alin_u_rows, int Alin_u_cols,
                           double *blin_u_data, int blin_u_len, int verbose) {
    (void)verbose;
    if (need_global("set_linear_constraints")) return -1;
    const tmpc::Solver &v = g_solver->s;
    // an empty side arrives as 0 x n (TinyMPC.jl:264 zeros(0, nu)); a non-empty one must match nx / nu
    const bool has_x = Alin_x_rows > 0 && blin_x_len > 0, has_u = Alin_u_rows > 0 && blin_u_len > 0;
    if ((has_x && (Alin_x_cols != v.nx || blin_x_len != Alin_x_rows)) ||
        (has_u && (Alin_u_cols != v.nu || blin_u_len != Alin_u_rows))) {
        set_error("set_linear_constraints: Alin_x must be (m x nx) with m entries in blin_x, Alin_u (m x nu) likewise");
        return -1;
    }
    return each_global([&](tinympc_solver *h) {
        return tinympc_set_linear_constraints(h, Alin_x_data, has_x ? Alin_x_rows : 0, blin_x_data, Alin_u_data,
                                              has_u ? Alin_u_rows : 0, blin_u_data);
    });
}

int set_cone_constraints(int *Acu_data, int Acu_len, int *qcu_data, int qcu_len, double *cu_data,
                         int cu_len, int *Acx_data, int Acx_len, int *qcx_data, int qcx_len,
                         double *cx_data, int cx_len, int verbose) {
    (void)verbose;
    if (need_global("set_cone_constraints")) return -1;
    if (Acu_len != qcu_len || Acu_len != cu_len || Acx_len != qcx_len || Acx_len != cx_len) {
        set_error("set_cone_constraints: Ac, qc and c of one side must have the same length");
        return -1;
    }
    // bindings.cpp:478-483: the flags of the non-empty halves are switched on
    return each_global([&](tinympc_solver *h) {
        return tinympc_set_cone_constraints(h, Acu_data, qcu_data, cu_data, Acu_len, Acx_data, qcx_data, cx_data, Acx_len);
    });
}

int get_status(int *iter, int *solved, double *residuals4) {
    if (need_global("get_status")) return -1;
    if (g_sharded) return guarded("get_status", [&] { return tinympc_sharded_get_status(g_sharded.get(), iter, solved, residuals4); });
    return tinympc_get_status(g_solver.get(), iter, solved, residuals4);
}
int reset_workspace(void) {
    if (need_global("reset_workspace")) return -1;
    return each_global([&](tinympc_solver *h) { return tinympc_reset(h); });
}
// The fused closed loop on the process-global solver (one device): what examples/cartpole_example_mpc.jl:35-51 and
// examples/rocket_landing_constraints.jl:97-134 do with one solve per host iteration, as one launch.
int set_ref_sequence(double *x_ref_seq, int x_rows, int x_cols, double *u_ref_seq, int u_rows, int u_cols, int steps) {
    if (need_global("set_ref_sequence")) return -1;
    if (g_sharded) {
        set_error("set_ref_sequence: not available on a sharded solver");
        return -1;
    }
    return tinympc_set_ref_sequence(g_solver.get(), x_ref_seq, x_rows, x_cols, u_ref_seq, u_rows, u_cols, steps);
}
int mpc_rollout(int steps, double *x_log, double *u_log, int *iter_log) {
    if (need_global("mpc_rollout")) return -1;
    if (g_sharded) {
        set_error("mpc_rollout: not available on a sharded solver");
        return -1;
    }
    const int st = tinympc_mpc_rollout(g_solver.get(), steps, nullptr);
    if (st < 0) return -1;
    if ((x_log || u_log || iter_log) && tinympc_get_mpc_log(g_solver.get(), x_log, u_log, iter_log)) return -1;
    return st;
}

}  // extern "C"
