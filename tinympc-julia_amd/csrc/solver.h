// Batched TinyMPC solver object behind the C-ABI (include/tinympc_hip.h).
// Host side of the drop-in boundary that replaces the reference's global
// `g_solver` + tiny_* calls (reference: src/bindings.cpp:15-490).
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "admm_params.h"
#include "host_setup.h"

namespace tmpc {

struct Solver;

// One specialised quad-kernel instantiation (admm_quad.hip.h) and its pack builders.
struct KernelEntry {
    int nx, nu, N, G;  // G = lanes per instance
    const char *name;
    void (*build_coef)(const Solver &, std::vector<unsigned char> &);  // typed by Solver::precision
    void (*build_bounds)(const Solver &, std::vector<float> &);
    hipError_t (*launch)(const AdmmParams &, int precision, bool state_bounds_active, hipStream_t);
    bool adp = false;  // the entry also carries the adaptive-rho kernels (QuadShape::ADP_OK)
    bool jit = false;  // specialised at setup (jit.cpp): fp64-recurrence kernels only, no adaptive-rho variants
};
// Specialisation at setup (jit.cpp): compiles / loads the on-chip kernel of a shape the library was not built with and adds
// it to what find_quad_kernel / find_mfma_kernel return; nullptr when nothing could be specialised (no compiler, no
// sources, the state does not fit, TINYMPC_HIP_NO_JIT)
const KernelEntry *jit_kernel_for(int nx, int nu, int N, int verbose);
const KernelEntry *jit_find(bool mfma, int nx, int nu, int N, int group);
// ... and the transposed-sets matrix-core kernel for exactly a solver's constraint layout (two cones per side, linear rows, a
// horizon the library was not built with): the unit already loaded that takes this solver / compile it now
struct ConeEntry;
const ConeEntry *jit_trans_find(const Solver &);
const ConeEntry *jit_trans_for(const Solver &, int verbose);
// group < 0: the shape's default group size; otherwise that exact variant (nullptr if not built)
const KernelEntry *find_quad_kernel(int nx, int nu, int N, int group = -1);
// the variant best suited to `batch` instances (nullptr: no specialised kernel for the shape)
const KernelEntry *select_quad_kernel(int nx, int nu, int N, int batch);
// the matrix-core kernel of the shape, for one-shot solves (nullptr: not instantiated)
const KernelEntry *find_mfma_kernel(int nx, int nu, int N);
// One instantiation of the lean kernel (admm_lean.hip.h): the one-lane-per-instance quad entry's one-shot solves (cold start,
// workspace not kept) with zero or shared references and fp64 recurrences run there — the benchmark's calling pattern.
struct LeanEntry {
    int nx, nu, N;
    const char *name;
    hipError_t (*launch)(const AdmmParams &, bool live, bool knot_bounds, bool state_bounds, hipStream_t);
};
const LeanEntry *find_lean_kernel(int nx, int nu, int N);
// ... or ONE variant of it specialised at the first launch that needs it (jit.cpp; nullptr: the shape does not fit the kernel)
enum { LV_LIVE = 1, LV_UBK = 2, LV_ONE = 4, LV_XB = 8, LV_SHARED = 16, LV_F64 = 32, LV_COUNT = 64 };
const LeanEntry *jit_lean_for(int nx, int nu, int N, int variant, int verbose);
// the lean kernel's fp64 pack (lean_layout); false when the family does not qualify (cache.AmBKt is not (A - B Kinf)')
bool build_lean_pack(const Solver &, std::vector<double> &);
// One (nx, nu) instantiation of the run-time-horizon stream kernel (admm_streamg.hip.h).
struct StreamEntry {
    int nx, nu;
    int lanes;  // lanes per instance
    const char *name;
    void (*build_coef)(const Solver &, std::vector<unsigned char> &);
    void (*build_bounds)(const Solver &, std::vector<float> &);
    size_t (*lds_bytes)(int N, int precision);
    size_t (*scratch_floats)(int N, int sets);  // per instance; sets: 1 box, 2 + cones, 3 + linear
    hipError_t (*launch)(const AdmmParams &, int precision, int ext, bool het, hipStream_t);  // ext: 0 | 1 fdyn, cones | 2 + linear
};
const StreamEntry *find_stream_kernel(int nx, int nu);
// One (nx, nu) instantiation of the LDS-resident matrix-core kernel with a run-time horizon (admm_mfmac.hip.h):
// one-shot solves with box bounds, the affine term and cones.
struct ConeEntry {
    int nx, nu;
    int N;                                      // compile-time horizon of the entry, 0: any (run-time horizon)
    bool (*supports)(const Solver &);           // further conditions of the entry (null: none)
    bool plain;                                 // box-only one-shot solves of the shape run here too (measured faster than the quad kernel)
    const char *name;
    void (*build_coef)(const Solver &, std::vector<unsigned char> &);
    void (*build_bounds)(const Solver &, std::vector<float> &);
    size_t (*lds_bytes)(const Solver &);        // per workgroup (one wavefront = 16 instances)
    size_t (*scratch_floats)(const Solver &);   // for the whole batch
    bool (*bounds_vary)(const Solver &);
    hipError_t (*launch)(const AdmmParams &, bool ext, size_t lds, hipStream_t);
    bool ws = false;                            // takes every kind of solve: warm starts, kept workspace, chunks, the fused closed loop
};
const ConeEntry *find_cone_kernel(int nx, int nu, int N);   // N = 0: the run-time-horizon entry only
const ConeEntry *find_trans_kernel(int nx, int nu, int N);  // the transposed-sets kernel of the shape (admm_mfmat.hip.h), or null
// CU count of the current device (cached per device: a sharded handle launches on several)
int device_cu_count();
hipError_t launch_generic(const AdmmParams &, int precision, hipStream_t);
void build_generic_coef(const Solver &, std::vector<unsigned char> &);
void build_generic_bounds(const Solver &, std::vector<float> &);

// Environment switches (tuning / test aids), read once per solver at creation (read_switches)
struct Switches {
    int group = 0;          // TINYMPC_HIP_GROUP = 1 | 2 | 4: that lanes-per-instance variant, no matrix-core kernels
    bool strict_fp32 = false, no_quad = false, no_quad_adp = false, no_quad_adp1 = false, no_mfma = false, no_mfma_adp = false,
         mfma_oneshot_only = false, no_stream = false, no_stream_adp = false, no_mfmar = false, no_mfmac = false, mfmac_all = false,
         no_mfmat = false, mfmat_all = false, mfmat_ws_only = false, no_lean = false, no_refill = false,
         no_uni = false, no_os = false, lean_one = false,   // lean_one: TINYMPC_HIP_LEAN_ONE — the lean kernel's 512-register variant at any batch
         no_jit = false;                                    // TINYMPC_HIP_NO_JIT: no unit specialised at setup, loaded or not
    int mfmac_debug = 0;    // timing probe builds only
};
Switches read_switches();

struct Settings {
    double abs_pri_tol = 1e-3, abs_dua_tol = 1e-3;  // TinyMPC.jl:57-58
    int max_iter = 100;                              // TinyMPC.jl:59
    int check_termination = 1;                       // TinyMPC.jl:59,202
    int en_state_bound = 0, en_input_bound = 0;      // TinyMPC.jl:94-95
    int en_state_soc = 0, en_input_soc = 0;          // TinyMPC.jl:96-97 (parity unpinned)
    int en_state_linear = 0, en_input_linear = 0;    // TinyMPC.jl:98-99 (parity unpinned)
    int adaptive_rho = 0, adaptive_rho_clip = 1;     // TinyMPC.jl:59-61,100-103
    double adaptive_rho_min = 0.1, adaptive_rho_max = 10.0;
};

struct Solver {
    int nx = 0, nu = 0, N = 0, batch = 1, device = 0;
    int verbose = 0;
    Mat A, B, Q, R;
    Cache cache;
    Settings st;
    // per-knot bounds, column-major fp64 (nx x N, nu x (N-1)); +-1e17 until set
    std::vector<double> x_min, x_max, u_min, u_max;
    // affine dynamics term and cone constraints (parity UNPINNED: they exist only in the absent
    // TinyMPC submodule; run on the stream kernel, or the generic one for shapes outside its grid)
    std::vector<double> fdyn;  // nx, all zero by default
    bool has_fdyn = false;
    int ncx = 0, ncu = 0;
    int Acx[8] = {0}, qcx[8] = {0}, Acu[8] = {0}, qcu[8] = {0};
    double cx[8] = {0}, cu[8] = {0};
    float *d_sgc = nullptr, *d_svc = nullptr, *d_syc = nullptr, *d_szc = nullptr;
    bool cones_active() const { return (st.en_state_soc && ncx > 0) || (st.en_input_soc && ncu > 0); }
    // linear inequalities Alin_x x <= blin_x, Alin_u u <= blin_u at every knot (bindings.cpp:413-450; UNPINNED):
    // rows row-major fp64, at most LIN_MAX_ROWS per side
    int mlx = 0, mlu = 0;
    std::vector<double> lin_Ax, lin_bx, lin_Au, lin_bu;
    float *d_lin = nullptr, *d_sgl = nullptr, *d_svl = nullptr, *d_syl = nullptr, *d_szl = nullptr;
    bool lin_dirty = false;
    bool lin_active() const { return (st.en_state_linear && mlx > 0) || (st.en_input_linear && mlu > 0); }
    // constraint sets whose arrays the stream / generic scratch lays out: box | + cones | + linear
    int constraint_sets() const { return lin_active() ? 3 : (cones_active() ? 2 : 1); }
    // adaptive rho (admm.cpp:147-174): sensitivities d(Kinf | Pinf)/d rho of the family (column-major, set by the
    // caller or computed on first use as TinyMPC.jl:301-352 does) and each instance's own (rho, Kinf, Pinf) on the
    // device; generic kernel only
    std::vector<double> sens;
    bool sens_set = false, sens_dirty = true, adapt_dirty = true;
    // the adaptive state on the device is `family + (rho_b - rho_family) x tables` with the CURRENT tables (what every
    // adaptive kernel leaves; false once the tables change under a live state): the matrix-core variant rebuilds an instance's
    // Kinf / Pinf from rho_b alone
    bool adapt_pure = true;
    double *d_sens = nullptr, *d_adapt = nullptr;
    unsigned char *d_adp_cols = nullptr;  // stream kernel, adaptive rho: per-lane coefficient columns (scratch)
    size_t adp_cols_bytes = 0;
    int set_sensitivity(const double *dK, const double *dP);
    int get_adaptive_state(double *rho, double *Kinf, double *Pinf);
    // one problem family PER INSTANCE: per-instance A, B (column-major, concatenated),
    // Riccati caches and diag/rho scalars; runs on the stream kernel with per-lane coefficient columns
    bool hetero = false;
    bool no_specialise = false;   // init() does not compile an on-chip unit for the shape (jit.cpp)
    bool layout_final = false;    // a constraint layout is specialised (jit_trans_for) from the first solve on, not while the setters run
    std::vector<double> het_A, het_B;
    std::vector<Cache> het_cache;
    float *d_het_aux = nullptr;
    // references as last set by the host API: kind 0 zero / 1 shared / 2 per instance
    std::vector<float> h_xref, h_uref;
    int xref_kind = 0, uref_kind = 0;
    bool refs_dirty = true;
    bool refs_device_owned = false;  // caller writes d_xref/d_uref itself (tinympc_set_ref_mode)
    // the references the next launch will see are per instance ([B][N][nx] / [B][N-1][nu], upload_refs)
    bool refs_per_instance() const { return refs_device_owned ? ref_mode == REF_PER_INSTANCE : (xref_kind >= 2 || uref_kind >= 2); }
    int ref_mode = REF_ZERO;
    bool warm_start = true;
    bool cache_overridden = false;  // set_cache_terms replaced the host Riccati's terms
    bool packs_dirty = true;
    bool state_bounds_active = false;  // any finite (|b| < 1e17) enabled state bound
    const KernelEntry *ke = nullptr;  // specialised quad kernel, or
    const StreamEntry *se = nullptr;  // run-time-horizon stream kernel, or
    const ConeEntry *ce = nullptr;    // LDS-resident matrix-core kernel (one-shot solves), or (all null) the generic kernel
    std::string kernel_name;
    // the lean kernel of the shape (admm_lean.hip.h): takes the one-lane-per-instance quad entry's one-shot solves without an
    // active state bound (launch_pass decides per launch); its pack, and whether the family qualifies (build_lean_pack)
    const LeanEntry *le = nullptr;
    bool lean_jit = false;                // no built-in lean instantiation: single variants specialised at the launches that need them
    const LeanEntry *le_var[LV_COUNT] = {};
    bool le_var_tried[LV_COUNT] = {};
    double *d_lean = nullptr;
    bool lean_ok = false, lean_knot_bounds = false;
    bool lean_enabled = true;             // TINYMPC_HIP_NO_LEAN, read once at creation
    std::string last_launch_name;         // the kernel the most recent launch actually ran (family or its lean variant)
    // device buffers
    unsigned char *d_coef = nullptr;
    float *d_bounds = nullptr;
    float *d_x0 = nullptr, *d_xref = nullptr, *d_uref = nullptr;
    size_t xref_cap = 0, uref_cap = 0;  // floats allocated
    float *d_xout = nullptr, *d_uout = nullptr, *d_res = nullptr;
    int *d_iter = nullptr, *d_solved = nullptr;
    float *d_sd = nullptr, *d_sy = nullptr, *d_sz = nullptr, *d_sg = nullptr, *d_sv = nullptr;
    uint32_t *d_gstat = nullptr;  // [2 * GSTAT_WORDS]: the public status block, then the kernels' accumulator
    uint32_t *h_gstat = nullptr;  // pinned
    // host round trips (fp64 caller buffers <-> fp32 device buffers): pinned staging, copy stream, two chunk slots
    float *h_stage = nullptr;
    size_t stage_cap = 0;  // floats
    hipStream_t s_copy = nullptr;
    hipEvent_t ev_copy[2] = {nullptr, nullptr};
    int d2h_double(const float *d, double *out, size_t n);
    // fp32 host buffers of the *_f32 entry points: page-locked on first sight (at most 8 ranges, of at least 1 MB) so that
    // the copies of a caller who reuses its buffers are direct DMAs; unlocked when the solver is destroyed
    std::vector<std::pair<void *, size_t>> pinned_ranges;
    int pin_host_range(void *p, size_t bytes);
    int unpin_host_range(void *p);
    int h2d_float(float *d, const double *in, size_t n);
    float *d_scratch = nullptr;
    double *d_ws64 = nullptr;    // precision 2: the fp64 workspace block (admm_generic.hip.h, Ws64)
    size_t ws64_cap = 0;
    float *d_mpc_x = nullptr, *d_mpc_u = nullptr;  // fused closed-loop logs
    int *d_mpc_iter = nullptr;
    int mpc_cap = 0, mpc_steps_last = 0;
    size_t scratch_cap = 0;
    bool solved_once = false;
    bool rollout_quad = false;     // the pending closed loop runs fused on the quad kernel
    bool g_maybe_nonzero = false;  // the workspace's state dual may hold non-zeros (see launch_pass)
    bool profiling = false;
    static constexpr int EV_RING = 256;  // event pairs around the most recent launches (profiling mode)
    std::vector<hipEvent_t> ev_ring;     // [2 * EV_RING], created on first use
    long launches = 0;                   // launches recorded since profiling was switched on
    int precision = 0;  // 0: fp64 recurrences, fp32 state (default), 1: all fp32, 2: all fp64 (generic kernel)

    int ex() const { return nx * N; }
    int eu() const { return nu * (N - 1); }

    ~Solver();
    int init(const double *A_, const double *B_, const double *Q_, const double *R_, double rho,
             int nx_, int nu_, int N_, int batch_, int device_, int verbose_);
    int init_families(const double *A_, const double *B_, const double *Q_, const double *R_, const double *rho_,
                      int nx_, int nu_, int N_, int batch_, int device_, int verbose_);
    int alloc_batch(int batch_);
    // family-level state of another solver of the SAME family (settings, bounds, affine term, cones, linear rows, cache,
    // sensitivities, precision / warm-start / compaction switches): what set_gpus() replays onto fresh shards
    int copy_family_state(const Solver &o);
    // last launch of this solver, whatever stream it went to: every getter waits for it first
    hipEvent_t ev_done = nullptr;
    bool ev_done_pending = false;
    int wait_last_launch();
    int select_kernel(bool rollout = false);  // rollout: the next launch is the fused closed loop
    // routing (solver.hip, "kernel routing"): switches, the families' routes, the cached decision
    Switches sw;
    bool strict_precision = false;            // tinympc_set_strict_precision: precision = 1 really means fp32 recurrences
    bool strict_fp32() const { return precision != 0 && (sw.strict_fp32 || strict_precision); }
    bool extensions_active() const { return has_fdyn || cones_active() || lin_active(); }
    bool rollout_on_quad() const { return sw.mfma_oneshot_only || !warm_start; }
    const KernelEntry *route_quad(bool rollout) const;
    const KernelEntry *route_mfma(bool rollout, const KernelEntry *quad) const;
    const StreamEntry *route_stream() const;
    const ConeEntry *route_cone(bool rollout, bool have_quad, bool have_stream) const;
    const ConeEntry *route_trans(bool rollout, const ConeEntry *oneshot) const;
    std::vector<long> routing_key(bool rollout) const;
    bool bounds_vary_by_knot() const;   // do the enabled box bounds depend on the knot?
    std::vector<long> routed_key;
    bool routed = false;
    unsigned route_gen = 0;                   // bumped by setters whose effect on routing the key's scalars do not show
    void free_batch();
    int upload_packs();
    int upload_refs();
    int set_x0(const double *x0, int cols);
    int set_ref(bool is_x, const double *ref, int cols);
    int set_ref_sequence(const double *x_seq, const double *u_seq, int steps);
    float *d_xref_seq = nullptr, *d_uref_seq = nullptr;  // shared references of every step of a fused closed loop
    int ref_seq_steps = 0;
    int set_bounds(const double *xmin, const double *xmax, const double *umin, const double *umax);
    int set_fdyn(const double *f);
    int set_cones(const int *Acu_, const int *qcu_, const double *cu_, int ncu_, const int *Acx_, const int *qcx_,
                  const double *cx_, int ncx_);
    int set_linear(const double *Ax, int mx, const double *bx, const double *Au, int mu, const double *bu);
    int ensure_extension_buffers();
    int reset();
    int solve_async(hipStream_t stream, int mpc_steps = 0);
    // one kernel launch over `n_slots` instances (idx: their ids, NULL = 0..n_slots-1) for at most `max_iter_pass`
    // iterations, the instances having done `iter_offset` already
    int launch_pass(hipStream_t stream, int mpc_steps, const int *idx, int n_slots, int iter_offset, int max_iter_pass,
                    bool cold, bool save);
    // tolerance-terminated solves of big batches in chunks of `chunk_iters` iterations: after each chunk the
    // instances still iterating are compacted, so wavefronts do not idle behind their slowest instance
    int solve_chunked(hipStream_t stream);
    // closed loop on the matrix-core kernel: per step one WS launch and a plant-update kernel, stream-ordered, the plant
    // state kept in fp64 on the device between steps (what the quad kernel's fused loop keeps in registers)
    int rollout_steps(hipStream_t stream, int mpc_steps);
    double *d_plant = nullptr, *d_x0d = nullptr;  // [A | B] column-major fp64; [B][nx] plant state
    const double *x0d_launch = nullptr;           // handed to the next launch_pass
    int chunk_iters = 0;  // 0: off
    int *d_idx[2] = {nullptr, nullptr};
    int *d_count = nullptr;
    int get_mpc_log(double *x, double *u, int *iter);
    int solve_status();
    double kernel_elapsed_ms();
    double kernel_elapsed_mean_ms(int last_n);
    int get_traj(bool states, double *buf);
    int get_status(int *iter, int *solved, double *res4);
    int get_workspace(double *d, double *y, double *g, double *v, double *z);
    int set_workspace(const double *d, const double *y, const double *g, const double *v,
                      const double *z);
};

void set_error(const std::string &msg);
const char *last_error();
bool hip_ok(hipError_t e, const char *what);

}  // namespace tmpc

struct tinympc_solver {
    tmpc::Solver s;
};
