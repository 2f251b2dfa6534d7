// Host side of the batched solver: device memory, pack uploads, launches.
#include "solver.h"
#include "admm_generic.hip.h"   // Ws64 (layout of the precision-2 workspace)

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <cstring>
#include <thread>


namespace tmpc {

static thread_local std::string g_err;
void set_error(const std::string &msg) {
    g_err = msg;
    std::fprintf(stderr, "tinympc_hip: %s\n", msg.c_str());
}
const char *last_error() { return g_err.c_str(); }
bool hip_ok(hipError_t e, const char *what) {
    if (e == hipSuccess) return true;
    set_error(std::string(what) + ": " + hipGetErrorString(e));
    return false;
}
#define HIP_TRY(expr)                       \
    do {                                    \
        if (!hip_ok((expr), #expr)) return -1; \
    } while (0)

// fp32 <-> fp64 conversion of the host-side staging buffers, spread over a few threads: at batch 65 536
// a solution is 6.5 M elements and a single-threaded loop costs more than the kernel that produced it.
// The workers are started once and parked on a condition variable: spawning 16 threads per call costs ~0.5 ms,
// more than the copy they help with.
class HostPool {
public:
    static HostPool &get() {
        static HostPool pool;
        return pool;
    }
    // runs f(lo, hi) over [0, n) in `parts` contiguous ranges, part 0 on the calling thread
    void run(size_t n, size_t parts, const std::function<void(size_t, size_t)> &f) {
        std::unique_lock<std::mutex> outer(call_);  // one parallel region at a time
        parts = std::min(parts, workers_.size() + 1);
        const size_t chunk = (n + parts - 1) / parts;
        {
            std::lock_guard<std::mutex> lk(m_);
            fn_ = &f;
            n_ = n;
            chunk_ = chunk;
            next_ = 1;
            parts_ = parts;
            pending_ = parts - 1;
        }
        cv_.notify_all();
        f(0, std::min(n, chunk));
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [&] { return pending_ == 0; });
        fn_ = nullptr;
    }

private:
    HostPool() {
        const unsigned hw = std::thread::hardware_concurrency();
        const unsigned nt = std::min<unsigned>(hw ? hw : 1, 16);
        for (unsigned i = 1; i < nt; ++i) workers_.emplace_back([this] { loop(); });
    }
    ~HostPool() {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &t : workers_) t.join();
    }
    void loop() {
        for (;;) {
            std::unique_lock<std::mutex> lk(m_);
            cv_.wait(lk, [&] { return stop_ || next_ < parts_; });  // parked until a region has an unclaimed part
            if (stop_) return;
            const size_t part = next_++;
            const auto *f = fn_;
            const size_t lo = part * chunk_, hi = std::min(n_, lo + chunk_);
            lk.unlock();
            if (lo < hi) (*f)(lo, hi);
            lk.lock();
            if (--pending_ == 0) done_.notify_one();
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_, call_;
    std::condition_variable cv_, done_;
    const std::function<void(size_t, size_t)> *fn_ = nullptr;
    size_t n_ = 0, chunk_ = 0, next_ = 0, parts_ = 0, pending_ = 0;
    bool stop_ = false;
};

template <class F>
static void parallel_chunks(size_t n, F &&f, size_t min_parallel = (size_t)1 << 16) {
    const unsigned hw = std::thread::hardware_concurrency();
    const size_t nt = n < min_parallel ? 1 : std::min<size_t>(hw ? hw : 1, 16);
    if (nt <= 1) {
        f((size_t)0, n);
        return;
    }
    HostPool::get().run(n, nt, std::function<void(size_t, size_t)>(f));
}
static void widen(const float *src, double *dst, size_t n) {
    parallel_chunks(n, [=](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) dst[i] = (double)src[i];
    });
}
static void narrow(const double *src, float *dst, size_t n) {
    parallel_chunks(n, [=](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) dst[i] = (float)src[i];
    });
}

template <class T>
static int dev_alloc(T *&p, size_t n) {
    if (p) {
        (void)hipFree(p);
        p = nullptr;
    }
    if (n == 0) n = 1;
    HIP_TRY(hipMalloc((void **)&p, n * sizeof(T)));
    return 0;
}
template <class T>
static void dev_free(T *&p) {
    if (p) (void)hipFree(p);
    p = nullptr;
}

Solver::~Solver() {
    for (const auto &r : pinned_ranges) (void)hipHostUnregister(r.first);
    pinned_ranges.clear();
    free_batch();
    dev_free(d_coef);
    dev_free(d_bounds);
    dev_free(d_gstat);
    dev_free(d_sens);
    dev_free(d_lean);
    dev_free(d_plant);
    if (h_gstat) (void)hipHostFree(h_gstat);
    h_gstat = nullptr;
    for (hipEvent_t e : ev_ring)
        if (e) (void)hipEventDestroy(e);
    if (h_stage) (void)hipHostFree(h_stage);
    for (hipEvent_t e : ev_copy)
        if (e) (void)hipEventDestroy(e);
    if (s_copy) (void)hipStreamDestroy(s_copy);
    if (ev_done) (void)hipEventDestroy(ev_done);
}

// Getters are synchronous, and a solve may have been enqueued on any stream (tinympc_solve_async): they wait for the
// event recorded behind the last launch rather than relying on null-stream ordering, which a non-blocking stream escapes.
int Solver::wait_last_launch() {
    if (ev_done_pending) {
        HIP_TRY(hipEventSynchronize(ev_done));
        ev_done_pending = false;
    }
    return 0;
}

int Solver::copy_family_state(const Solver &o) {
    if (o.nx != nx || o.nu != nu || o.N != N || hetero || o.hetero) {
        set_error("copy_family_state: solvers of different shape");
        return -1;
    }
    st = o.st;
    x_min = o.x_min, x_max = o.x_max, u_min = o.u_min, u_max = o.u_max;
    fdyn = o.fdyn, has_fdyn = o.has_fdyn;
    ncx = o.ncx, ncu = o.ncu;
    for (int i = 0; i < 8; ++i) {
        Acx[i] = o.Acx[i], qcx[i] = o.qcx[i], cx[i] = o.cx[i];
        Acu[i] = o.Acu[i], qcu[i] = o.qcu[i], cu[i] = o.cu[i];
    }
    mlx = o.mlx, mlu = o.mlu;
    lin_Ax = o.lin_Ax, lin_bx = o.lin_bx, lin_Au = o.lin_Au, lin_bu = o.lin_bu;
    lin_dirty = true;
    cache = o.cache;
    sens = o.sens, sens_set = o.sens_set, sens_dirty = true, adapt_dirty = true;
    warm_start = o.warm_start;
    cache_overridden = o.cache_overridden;
    precision = o.precision;
    strict_precision = o.strict_precision;
    chunk_iters = o.chunk_iters;
    packs_dirty = true;
    return select_kernel() || ensure_extension_buffers();
}

void Solver::free_batch() {
    dev_free(d_x0);
    dev_free(d_xref);
    dev_free(d_uref);
    dev_free(d_xout);
    dev_free(d_uout);
    dev_free(d_res);
    dev_free(d_iter);
    dev_free(d_solved);
    dev_free(d_sd);
    dev_free(d_sy);
    dev_free(d_sz);
    dev_free(d_sg);
    dev_free(d_sv);
    dev_free(d_scratch);
    dev_free(d_ws64);
    ws64_cap = 0;
    dev_free(d_het_aux);
    dev_free(d_sgc);
    dev_free(d_svc);
    dev_free(d_syc);
    dev_free(d_szc);
    dev_free(d_idx[0]);
    dev_free(d_idx[1]);
    dev_free(d_count);
    dev_free(d_lin);
    dev_free(d_sgl);
    dev_free(d_svl);
    dev_free(d_syl);
    dev_free(d_szl);
    dev_free(d_adapt);
    dev_free(d_adp_cols);
    adp_cols_bytes = 0;
    dev_free(d_x0d);
    adapt_dirty = true;
    dev_free(d_mpc_x);
    dev_free(d_mpc_u);
    dev_free(d_mpc_iter);
    dev_free(d_xref_seq);
    dev_free(d_uref_seq);
    ref_seq_steps = 0;
    mpc_cap = 0;
    xref_cap = uref_cap = scratch_cap = 0;
}

int Solver::init(const double *A_, const double *B_, const double *Q_, const double *R_, double rho,
                 int nx_, int nu_, int N_, int batch_, int device_, int verbose_) {
    if (nx_ < 1 || nu_ < 1 || N_ < 2 || batch_ < 1) {
        set_error("invalid dimensions (need nx >= 1, nu >= 1, N >= 2, batch >= 1)");
        return -1;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
        set_error("no HIP device available (this library has no CPU fallback)");
        return -1;
    }
    if (device_ < 0) HIP_TRY(hipGetDevice(&device_));
    device = device_;
    HIP_TRY(hipSetDevice(device));
    sw = read_switches();   // the environment, once per solver
    lean_enabled = !sw.no_lean;
    nx = nx_;
    nu = nu_;
    N = N_;
    verbose = verbose_;
    A = Mat(nx, nx, A_);
    B = Mat(nx, nu, B_);
    Q = Mat(nx, nx, Q_);
    R = Mat(nu, nu, R_);
    if (precompute_cache(A, B, Q, R, rho, cache) != 0) {
        set_error("Riccati precompute failed: R + B'PB is singular");
        return -1;
    }
    // a shape without an on-chip kernel in the library: specialise one now (jit.cpp; one-off, cached on disk)
    if (!no_specialise && !find_quad_kernel(nx, nu, N, -1) && !find_mfma_kernel(nx, nu, N) && !find_trans_kernel(nx, nu, N))
        (void)jit_kernel_for(nx, nu, N, verbose);
    if (verbose)
        std::printf("tinympc_hip: setup nx=%d nu=%d N=%d rho=%g batch=%d (Riccati %d sweeps)\n", nx, nu,
                    N, rho, batch_, cache.riccati_iters);
    fdyn.assign((size_t)nx, 0.0);
    x_min.assign((size_t)ex(), -1e17);
    x_max.assign((size_t)ex(), 1e17);
    u_min.assign((size_t)eu(), -1e17);
    u_max.assign((size_t)eu(), 1e17);
    batch = batch_;
    if (select_kernel()) return -1;
    if (dev_alloc(d_gstat, (size_t)2 * GSTAT_WORDS)) return -1;
    HIP_TRY(hipMemset(d_gstat, 0, 2 * GSTAT_WORDS * sizeof(uint32_t)));
    HIP_TRY(hipHostMalloc((void **)&h_gstat, GSTAT_WORDS * sizeof(uint32_t), hipHostMallocDefault));
    std::memset(h_gstat, 0, GSTAT_WORDS * sizeof(uint32_t));
    packs_dirty = true;
    return alloc_batch(batch_);
}

// One family per instance: B independent (A, B, Q, R, rho) sets, each with its own Riccati cache computed on
// the host in fp64 (threaded); bounds, settings and references behave as in the single-family solver.
int Solver::init_families(const double *A_, const double *B_, const double *Q_, const double *R_, const double *rho_,
                          int nx_, int nu_, int N_, int batch_, int device_, int verbose_) {
    no_specialise = true;   // (a per-instance-family solver runs on the stream kernel: nothing to specialise at setup)
    if (init(A_, B_, Q_, R_, rho_[0], nx_, nu_, N_, batch_, device_, verbose_)) return -1;
    hetero = true;
    const size_t Bn = (size_t)batch, nxx = (size_t)nx * nx, nxu = (size_t)nx * nu, nuu = (size_t)nu * nu;
    het_A.assign(A_, A_ + Bn * nxx);
    het_B.assign(B_, B_ + Bn * nxu);
    het_cache.assign(Bn, Cache());
    std::vector<int> bad(1, 0);
    parallel_chunks(Bn, [&](size_t lo, size_t hi) {
        for (size_t b = lo; b < hi; ++b)
            if (precompute_cache(Mat(nx, nx, A_ + b * nxx), Mat(nx, nu, B_ + b * nxu), Mat(nx, nx, Q_ + b * nxx),
                                 Mat(nu, nu, R_ + b * nuu), rho_[b], het_cache[b]))
                bad[0] = 1;
    }, 64);
    if (bad[0]) {
        set_error("Riccati precompute failed for at least one instance (R + B'PB singular)");
        return -1;
    }
    std::vector<float> aux((size_t)(nx + nu + 1) * Bn);
    for (size_t b = 0; b < Bn; ++b) {
        for (int i = 0; i < nx; ++i) aux[(size_t)i * Bn + b] = (float)het_cache[b].Qd[i];
        for (int a = 0; a < nu; ++a) aux[(size_t)(nx + a) * Bn + b] = (float)het_cache[b].Rd[a];
        aux[(size_t)(nx + nu) * Bn + b] = (float)het_cache[b].rho;
    }
    if (dev_alloc(d_het_aux, aux.size())) return -1;
    HIP_TRY(hipMemcpy(d_het_aux, aux.data(), aux.size() * sizeof(float), hipMemcpyHostToDevice));
    packs_dirty = true;
    return select_kernel() || ensure_extension_buffers();
}

// ---- kernel routing ------------------------------------------------------------------------------------------------------
// The environment switches (tuning / test aids) are read ONCE, when the solver is created; nothing on
// the solve path calls getenv.
Switches read_switches() {
    auto on = [](const char *name) { return std::getenv(name) != nullptr; };
    Switches w;
    if (const char *g = std::getenv("TINYMPC_HIP_GROUP")) w.group = std::atoi(g);
    w.strict_fp32 = on("TINYMPC_HIP_STRICT_FP32");
    w.no_quad = on("TINYMPC_HIP_NO_QUAD");
    w.no_quad_adp = on("TINYMPC_HIP_NO_QUAD_ADP");
    w.no_quad_adp1 = on("TINYMPC_HIP_NO_QUAD_ADP1");
    w.no_mfma = on("TINYMPC_HIP_NO_MFMA");
    w.no_mfma_adp = on("TINYMPC_HIP_NO_MFMA_ADP");
    w.mfma_oneshot_only = on("TINYMPC_HIP_MFMA_ONESHOT_ONLY");
    w.no_stream = on("TINYMPC_HIP_NO_STREAM");
    w.no_stream_adp = on("TINYMPC_HIP_NO_STREAM_ADP");
    w.no_mfmar = on("TINYMPC_HIP_NO_MFMAR");
    w.no_mfmac = on("TINYMPC_HIP_NO_MFMAC");
    w.mfmac_all = on("TINYMPC_HIP_MFMAC_ALL");
    w.no_jit = on("TINYMPC_HIP_NO_JIT");
    w.no_mfmat = on("TINYMPC_HIP_NO_MFMAT");
    w.mfmat_all = on("TINYMPC_HIP_MFMAT_ALL");
    w.mfmat_ws_only = on("TINYMPC_HIP_MFMAT_WS_ONLY");
    w.no_lean = on("TINYMPC_HIP_NO_LEAN");
    w.no_refill = on("TINYMPC_HIP_NO_REFILL");
    w.no_uni = on("TINYMPC_HIP_NO_UNI");
    w.no_os = on("TINYMPC_HIP_NO_OS");
    w.lean_one = on("TINYMPC_HIP_LEAN_ONE");
    if (const char *d = std::getenv("TINYMPC_HIP_MFMAC_DEBUG")) w.mfmac_debug = std::atoi(d);
    return w;
}

// Capability table.  A kernel family is a route: `route_*` says whether the family takes the solver's shape, options and
// calling pattern (its "supports"), and — within the family — which instantiation costs least for the batch (lanes per
// instance by batch size: select_quad_kernel).  select_kernel asks the routes in order of preference, last word first:
//   transposed-sets matrix-core kernel (mfmat: every calling pattern of its shapes)
//   > one-shot matrix-core kernels with the affine term / cones (mfmar, compiled horizon; mfmac, LDS, any horizon)
//   > matrix-core kernel of the box-only shapes (mfma, and its adaptive-rho variant)
//   > lanes-per-instance kernels (quad, and their adaptive-rho variants)  > run-time-horizon stream kernel  > generic kernel.
// The result is cached under the key of everything the routes read; a solve re-routes only when that key has changed.

// lanes-per-instance family, its adaptive-rho variants included (precision = 1 — fp32 recurrences, asked for to save time —
// stays here only where no matrix-core kernel exists, or with TINYMPC_HIP_STRICT_FP32: route_mfma)
const KernelEntry *Solver::route_quad(bool rollout) const {
    if (sw.no_quad || extensions_active() || hetero) return nullptr;
    if (!st.adaptive_rho) {
        const KernelEntry *k = sw.group ? find_quad_kernel(nx, nu, N, sw.group) : nullptr;
        if (!k) k = select_quad_kernel(nx, nu, N, batch);
        // (a unit specialised at setup carries fp64 recurrences only; TINYMPC_HIP_NO_JIT on this solver: not a unit another one loaded either)
        return (k && k->jit && (precision != 0 || sw.no_jit)) ? nullptr : k;
    }
    // adaptive rho: the ADP variant where the shape has one (4 lanes per instance, coefficient rows in registers: the
    // cartpole shapes) ...
    const KernelEntry *ka = sw.no_quad_adp ? nullptr : find_quad_kernel(nx, nu, N, 4);
    if (!(ka && ka->adp && chunk_iters == 0 && !rollout && !cache_overridden)) return nullptr;
    // ... with ONE lane per instance — the benched variant of large batches — in the correction form (admm_quad.hip.h:
    // dK = (rho_b - rho_family) dKinf/drho next to the family's wave-uniform coefficients): zero or shared references,
    // fp64 recurrences, the adaptive state in closed form
    if (batch >= 20480 && precision == 0 && !sw.group && !refs_device_owned && xref_kind <= 1 && uref_kind <= 1 &&
        (adapt_pure || adapt_dirty) && !sw.no_quad_adp1)
        if (const KernelEntry *k1 = find_quad_kernel(nx, nu, N, 1))
            if (k1->adp) return k1;
    return ka;
}

// matrix-core kernel of the box-only shapes: plain solves with fp64 recurrences (horizons the shape has no
// lanes-per-instance kernel for — quadrotor N = 10, 15, 25 — included; the fused closed loop of a cold-started solver stays
// on the quad kernel), and the adaptive-rho variant (the instance's own Kinf as a correction to the shared products) where
// the quad family has none
const KernelEntry *Solver::route_mfma(bool rollout, const KernelEntry *quad) const {
    if (strict_fp32() || sw.group || sw.no_mfma || sw.no_quad) return nullptr;
    const KernelEntry *m = find_mfma_kernel(nx, nu, N);
    if (!m || (m->jit && sw.no_jit)) return nullptr;
    if (st.adaptive_rho)
        return (!quad && m->adp && !extensions_active() && chunk_iters == 0 && !rollout && !cache_overridden &&
                (adapt_pure || adapt_dirty) && !sw.no_mfma_adp) ? m : nullptr;
    if (!(quad || !(extensions_active() || hetero))) return nullptr;
    if (rollout && rollout_on_quad()) return nullptr;
    if (!(!sw.mfma_oneshot_only || (!warm_start && chunk_iters == 0))) return nullptr;
    return m;
}

// run-time-horizon stream kernel of (nx, nu): whatever no specialised kernel takes, if its LDS image fits
const StreamEntry *Solver::route_stream() const {
    const bool adp_ok = !extensions_active() && chunk_iters == 0 && !sw.no_stream_adp;
    if ((st.adaptive_rho && !adp_ok) || sw.no_stream) return nullptr;
    const StreamEntry *s2 = find_stream_kernel(nx, nu);
    if (s2 && 16.0 * batch * std::max(nx, nu) >= 4.0e9) s2 = nullptr;   // 32-bit lane byte offsets into one knot's rows
    if (s2 && s2->lds_bytes(N, precision) > 150 * 1024) s2 = nullptr;   // LDS image of coefficients + bounds
    return s2;
}

// one-shot solves (cold start, workspace not kept) with the affine term and / or cones: the register-resident matrix-core
// kernel where the horizon is compiled in (mfmar; box-only solves too where the entry says so: rocket N = 50, 4.3 ms
// against 5.5 on the quad kernel), else the LDS-resident one (mfmac, any horizon).  Only where a workspace-carrying
// kernel exists to fall back to (`fallback`).
const ConeEntry *Solver::route_cone(bool rollout, bool have_quad, bool have_stream) const {
    const ConeEntry *cn = sw.no_mfmar ? nullptr : find_cone_kernel(nx, nu, N);
    if (cn && cn->supports && !cn->supports(*this)) cn = nullptr;
    const bool plain_ok = sw.mfmac_all || (cn != nullptr && cn->plain);
    if (!((have_stream || (have_quad && plain_ok)) && !warm_start && chunk_iters == 0 && !rollout && !strict_fp32() && !hetero &&
          !st.adaptive_rho && (extensions_active() || plain_ok) && xref_kind < 2 && uref_kind < 2 &&
          !(refs_device_owned && ref_mode == REF_PER_INSTANCE) && !sw.no_mfmac && !sw.group && !sw.no_mfma))
        return nullptr;
    const ConeEntry *c2 = cn ? cn : find_cone_kernel(nx, nu, 0);
    if (c2 && c2->lds_bytes(*this) > 160 * 1024 - 1024) c2 = nullptr;   // horizon too long for one tile's LDS
    // (two cones per side / linear rows: the transposed-sets kernel specialised for the layout, route_trans, or the stream kernel)
    if ((st.en_state_soc && ncx > 1) || (st.en_input_soc && ncu > 1) || lin_active()) c2 = nullptr;
    return c2;
}

// transposed-sets matrix-core kernel: every kind of solve (one-shot, warm-started, workspace kept, chunked, the fused closed
// loop) of a shape that has it, with the affine term / at most one cone per side (box-only problems where the entry says so)
const ConeEntry *Solver::route_trans(bool rollout, const ConeEntry *oneshot) const {
    if (sw.no_mfmat || sw.no_mfma || sw.group) return nullptr;
    if (strict_fp32() || hetero || st.adaptive_rho || (refs_per_instance() && ref_seq_steps > 0) || st.max_iter < 1 || (double)batch * ex() >= 2.0e9)
        return nullptr;
    const ConeEntry *ct = find_trans_kernel(nx, nu, N);
    if (ct && (lin_active() || (ct->supports && !ct->supports(*this)))) ct = nullptr;
    if (ct && !(has_fdyn || cones_active() || ct->plain || sw.mfmat_all)) return nullptr;
    // a layout the built-in entries do not have — two cones on a side, linear rows, a horizon the library was not built with:
    // the unit specialised for exactly this layout (jit.cpp; compiled on first use, cached on disk)
    if (!ct && extensions_active() && !sw.no_jit) {
        ct = jit_trans_find(*this);
        if (!ct && !no_specialise && layout_final) ct = jit_trans_for(*this, verbose);   // (not for every intermediate layout of a setter sequence)
    }
    if (!ct || ct->lds_bytes(*this) > 160 * 1024 - 1024) return nullptr;
    if (sw.mfmat_ws_only && !warm_start && chunk_iters == 0 && !rollout && oneshot) return nullptr;   // (tests hold the families against each other)
    return ct;
}

bool Solver::bounds_vary_by_knot() const {
    if (st.en_state_bound)
        for (int k = 1; k < N; ++k)
            for (int r = 0; r < nx; ++r)
                if (x_min[r + (size_t)k * nx] != x_min[r] || x_max[r + (size_t)k * nx] != x_max[r]) return true;
    if (st.en_input_bound)
        for (int k = 1; k < N - 1; ++k)
            for (int a = 0; a < nu; ++a)
                if (u_min[a + (size_t)k * nu] != u_min[a] || u_max[a + (size_t)k * nu] != u_max[a]) return true;
    return false;
}

// everything the routes read, in one comparable value
std::vector<long> Solver::routing_key(bool rollout) const {
    return {nx, nu, N, batch, precision, warm_start, chunk_iters, has_fdyn, cones_active(), lin_active(), hetero, st.adaptive_rho,
            cache_overridden, refs_device_owned, xref_kind, uref_kind, ref_mode, adapt_pure, adapt_dirty, ref_seq_steps,
            st.max_iter < 1, st.en_state_soc, st.en_input_soc, ncx, ncu, Acx[0], qcx[0], Acu[0], qcu[0], mlx, mlu, rollout, strict_precision,
            (long)route_gen, Acx[1], qcx[1], Acu[1], qcu[1], st.en_state_linear, st.en_input_linear,
            extensions_active() && bounds_vary_by_knot(), layout_final};   // (a unit specialised at setup is compiled for one bound kind)
}

int Solver::select_kernel(bool rollout) {
    std::vector<long> key = routing_key(rollout);
    if (routed && key == routed_key) return 0;
    if (st.adaptive_rho && hetero) {
        set_error("adaptive_rho is not available on a per-instance-family solver");
        return -1;
    }
    if (precision == 2) {   // fp64 end to end: the generic kernel's double-state form, whatever the shape
        if (hetero || rollout) {
            set_error(hetero ? "precision 2 is not available on a per-instance-family solver" : "precision 2 has no fused closed loop (step it from the host)");
            return -1;
        }
        if (nx > GEN_MAX_NX || nu > GEN_MAX_NU) {
            set_error("problem shape exceeds the generic kernel limits (nx <= 64, nu <= 32)");
            return -1;
        }
        if (ke || se || ce) packs_dirty = true;
        ke = nullptr, se = nullptr, ce = nullptr;
        rollout_quad = false;
        kernel_name = "generic<f64>";
        routed_key = std::move(key);
        routed = true;
        return 0;
    }
    const KernelEntry *q = route_quad(rollout), *m = route_mfma(rollout, q), *k = m ? m : q;
    if (!k && (nx > GEN_MAX_NX || nu > GEN_MAX_NU)) {
        set_error("problem shape exceeds the generic kernel limits (nx <= 64, nu <= 32)");
        return -1;
    }
    const StreamEntry *s2 = k ? nullptr : route_stream();
    if (hetero && !s2) {
        set_error("per-instance families need a stream-kernel instantiation for (nx, nu) (nx in {2,3,4,6,8,10,12}, nu <= 4)");
        return -1;
    }
    const ConeEntry *c2 = route_cone(rollout, k != nullptr, s2 != nullptr), *ct = route_trans(rollout, c2);
    if (ct) c2 = ct;
    if (c2) k = nullptr, s2 = nullptr;
    rollout_quad = rollout && rollout_on_quad() && !ct;   // else: rollout_steps() on the matrix-core kernel / mfmat's fused loop
    if (k != ke || s2 != se || c2 != ce) packs_dirty = true;
    ke = k, se = s2, ce = c2;
    kernel_name = ke ? ke->name : (se ? se->name : (ce ? ce->name : "generic"));
    routed_key = std::move(key);
    routed = true;
    return 0;
}

int Solver::alloc_batch(int batch_) {
    if (hetero && batch_ != batch) {
        set_error("set_batch_size: a per-instance-family solver has a fixed batch");
        return -1;
    }
    if (batch_ < 1) {
        set_error("batch must be >= 1");
        return -1;
    }
    HIP_TRY(hipSetDevice(device));
    free_batch();
    batch = batch_;
    if (select_kernel()) return -1;
    const size_t Bn = (size_t)batch, EX = (size_t)ex(), EU = (size_t)eu();
    if (dev_alloc(d_x0, Bn * nx) || dev_alloc(d_xout, Bn * EX) || dev_alloc(d_uout, Bn * EU) ||
        dev_alloc(d_res, Bn * 4) || dev_alloc(d_iter, Bn) || dev_alloc(d_solved, Bn) ||
        dev_alloc(d_sd, Bn * EU) || dev_alloc(d_sy, Bn * EU) || dev_alloc(d_sz, Bn * EU) ||
        dev_alloc(d_sg, Bn * EX) || dev_alloc(d_sv, Bn * EX))
        return -1;
    // shared-size reference buffers up front; per-instance ones on demand
    if (dev_alloc(d_xref, EX) || dev_alloc(d_uref, EU)) return -1;
    xref_cap = EX;
    uref_cap = EU;
    HIP_TRY(hipMemset(d_x0, 0, Bn * nx * sizeof(float)));
    HIP_TRY(hipMemset(d_xout, 0, Bn * EX * sizeof(float)));
    HIP_TRY(hipMemset(d_uout, 0, Bn * EU * sizeof(float)));
    HIP_TRY(hipMemset(d_iter, 0, Bn * sizeof(int)));
    HIP_TRY(hipMemset(d_solved, 0, Bn * sizeof(int)));
    HIP_TRY(hipMemset(d_xref, 0, EX * sizeof(float)));
    HIP_TRY(hipMemset(d_uref, 0, EU * sizeof(float)));
    h_xref.clear();
    h_uref.clear();
    xref_kind = uref_kind = 0;
    ref_mode = REF_ZERO;
    refs_dirty = false;
    refs_device_owned = false;
    if (ensure_extension_buffers()) return -1;
    solved_once = false;
    return reset();
}

int Solver::reset() {
    HIP_TRY(hipSetDevice(device));
    if (wait_last_launch()) return -1;
    const size_t Bn = (size_t)batch, EX = (size_t)ex(), EU = (size_t)eu();
    HIP_TRY(hipMemset(d_sd, 0, Bn * EU * sizeof(float)));
    HIP_TRY(hipMemset(d_sy, 0, Bn * EU * sizeof(float)));
    HIP_TRY(hipMemset(d_sz, 0, Bn * EU * sizeof(float)));
    HIP_TRY(hipMemset(d_sg, 0, Bn * EX * sizeof(float)));
    HIP_TRY(hipMemset(d_sv, 0, Bn * EX * sizeof(float)));
    HIP_TRY(hipMemset(d_res, 0, Bn * 4 * sizeof(float)));
    if (d_sgc) {
        HIP_TRY(hipMemset(d_sgc, 0, Bn * EX * sizeof(float)));
        HIP_TRY(hipMemset(d_svc, 0, Bn * EX * sizeof(float)));
        HIP_TRY(hipMemset(d_syc, 0, Bn * EU * sizeof(float)));
        HIP_TRY(hipMemset(d_szc, 0, Bn * EU * sizeof(float)));
    }
    if (d_sgl) {
        HIP_TRY(hipMemset(d_sgl, 0, Bn * EX * sizeof(float)));
        HIP_TRY(hipMemset(d_svl, 0, Bn * EX * sizeof(float)));
        HIP_TRY(hipMemset(d_syl, 0, Bn * EU * sizeof(float)));
        HIP_TRY(hipMemset(d_szl, 0, Bn * EU * sizeof(float)));
    }
    if (d_ws64) HIP_TRY(hipMemset(d_ws64, 0, ws64_cap * sizeof(double)));
    adapt_dirty = true;  // adapted (rho, Kinf, Pinf) go back to the family's cache
    g_maybe_nonzero = false;
    return 0;
}

__global__ void adapt_fill_kernel(double *adapt, const double *family, int n, long batch) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (long)n * batch) adapt[i] = family[i / batch];
}

int Solver::set_sensitivity(const double *dK, const double *dP) {
    sens.assign((size_t)nu * nx + (size_t)nx * nx, 0.0);
    std::copy(dK, dK + (size_t)nu * nx, sens.begin());
    std::copy(dP, dP + (size_t)nx * nx, sens.begin() + (size_t)nu * nx);
    if (d_adapt && !adapt_dirty && sens_set) adapt_pure = false;   // new tables under a live adaptive state
    sens_set = true;
    sens_dirty = true;
    packs_dirty = true;   // (the matrix-core kernel's adaptive variant carries dPinf' as an operand)
    return 0;
}

int Solver::get_adaptive_state(double *rho, double *Kinf, double *Pinf) {
    HIP_TRY(hipSetDevice(device));
    if (wait_last_launch()) return -1;
    const size_t Bn = (size_t)batch, nk = (size_t)nu * nx, np = (size_t)nx * nx;
    if (!d_adapt || adapt_dirty) {  // nothing adapted yet: every instance holds the family's values
        for (size_t b = 0; b < Bn; ++b) {
            if (rho) rho[b] = cache.rho;
            if (Kinf) std::copy(cache.Kinf.a.begin(), cache.Kinf.a.end(), Kinf + b * nk);
            if (Pinf) std::copy(cache.Pinf.a.begin(), cache.Pinf.a.end(), Pinf + b * np);
        }
        return 0;
    }
    std::vector<double> h((1 + nk + np) * Bn);
    HIP_TRY(hipMemcpy(h.data(), d_adapt, h.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (size_t b = 0; b < Bn; ++b) {
        if (rho) rho[b] = h[b];
        if (Kinf)
            for (size_t e = 0; e < nk; ++e) Kinf[b * nk + e] = h[(1 + e) * Bn + b];
        if (Pinf)
            for (size_t e = 0; e < np; ++e) Pinf[b * np + e] = h[(1 + nk + e) * Bn + b];
    }
    return 0;
}

int Solver::upload_packs() {
    if (wait_last_launch()) return -1;   // (a launch still in flight on another stream reads the packs this overwrites)
    state_bounds_active = false;
    if (st.en_state_bound)
        for (size_t i = 0; i < x_min.size(); ++i)
            if (x_min[i] > -1e17 || x_max[i] < 1e17) {
                state_bounds_active = true;
                break;
            }
    std::vector<unsigned char> coef;
    std::vector<float> bnd;
    if (ke) {
        ke->build_coef(*this, coef);
        ke->build_bounds(*this, bnd);
    } else if (ce) {
        ce->build_coef(*this, coef);
        ce->build_bounds(*this, bnd);
    } else if (se) {
        se->build_coef(*this, coef);
        se->build_bounds(*this, bnd);
    } else {
        build_generic_coef(*this, coef);
        build_generic_bounds(*this, bnd);
    }
    if (dev_alloc(d_coef, coef.size()) || dev_alloc(d_bounds, bnd.size())) return -1;
    HIP_TRY(hipMemcpy(d_coef, coef.data(), coef.size(), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_bounds, bnd.data(), bnd.size() * sizeof(float), hipMemcpyHostToDevice));
    // the lean kernel's own pack where the selected entry has one lane per instance and the shape a lean instantiation
    le = (ke && ke->G == 1 && lean_enabled) ? find_lean_kernel(nx, nu, N) : nullptr;
    // (a shape without one — cartpole at another horizon, a unit specialised at setup with four lanes per instance: where one
    // lane per instance is the batch's variant, launch_pass specialises the variant it needs, jit_lean_for)
    lean_jit = !le && ke && lean_enabled && !sw.no_jit && !no_specialise && (ke->G == 1 || (ke->jit && ke->G < 16 && batch >= 20480));
    // precision 2 (the generic kernel's fp64-state form, one lane per instance like this one): one-shot solves of a shape the lean
    // kernel holds run on ITS fp64-state form, specialised on request — the reference's digits at the headline kernel's speed
    if (precision == 2 && !ke && !se && !ce && !hetero && !extensions_active() && lean_enabled && !sw.no_jit && !no_specialise) lean_jit = true;
    std::fill(le_var_tried, le_var_tried + LV_COUNT, false);
    lean_ok = false;
    if (le || lean_jit) {
        std::vector<double> lp;
        if (build_lean_pack(*this, lp)) {
            if (!le && (!ke || ke->G != 1)) {
                // the lean kernel reads the one-lane-per-instance bound pack — [knot][x_min x_max u_min u_max], then diag(Q) + rho,
                // diag(R) + rho (quad_entry.hip.h: build_quad_bounds with G = 1); the selected entry's pack is four lanes per
                // instance: the lean one rides behind the coefficient doubles (launch_pass points P.bounds there)
                constexpr float kInf = std::numeric_limits<float>::infinity();
                const int BW = 2 * nx + 2 * nu;
                std::vector<float> lb((size_t)N * BW + nx + nu + 1, 0.f);
                for (int k = 0; k < N; ++k) {
                    float *p = lb.data() + (size_t)k * BW;
                    for (int r = 0; r < nx; ++r) {
                        p[r] = st.en_state_bound ? (float)x_min[r + (size_t)k * nx] : -kInf;
                        p[nx + r] = st.en_state_bound ? (float)x_max[r + (size_t)k * nx] : kInf;
                    }
                    for (int a = 0; a < nu; ++a) {
                        const bool on = st.en_input_bound && k < N - 1;
                        p[2 * nx + a] = on ? (float)u_min[a + (size_t)k * nu] : -kInf;
                        p[2 * nx + nu + a] = on ? (float)u_max[a + (size_t)k * nu] : kInf;
                    }
                }
                for (int r = 0; r < nx; ++r) lb[(size_t)N * BW + r] = (float)cache.Qd[r];
                for (int a = 0; a < nu; ++a) lb[(size_t)N * BW + nx + a] = (float)cache.Rd[a];
                const size_t at = lp.size();
                lp.resize(at + (lb.size() + 1) / 2, 0.0);
                std::memcpy(lp.data() + at, lb.data(), lb.size() * sizeof(float));
            }
            if (dev_alloc(d_lean, lp.size())) return -1;
            HIP_TRY(hipMemcpy(d_lean, lp.data(), lp.size() * sizeof(double), hipMemcpyHostToDevice));
            lean_ok = true;
            lean_knot_bounds = false;
            if (st.en_input_bound)
                for (int k = 1; k < N - 1 && !lean_knot_bounds; ++k)
                    for (int a = 0; a < nu; ++a)
                        if (u_min[a + (size_t)k * nu] != u_min[a] || u_max[a + (size_t)k * nu] != u_max[a]) lean_knot_bounds = true;
        }
    }
    packs_dirty = false;
    return 0;
}

int Solver::set_x0(const double *x0, int cols) {
    if (cols != 1 && cols != batch) {
        set_error("set_x0: expected nx x 1 or nx x batch");
        return -1;
    }
    HIP_TRY(hipSetDevice(device));
    if (wait_last_launch()) return -1;
    std::vector<float> h((size_t)batch * nx);
    for (int b = 0; b < batch; ++b)
        for (int i = 0; i < nx; ++i) h[(size_t)b * nx + i] = (float)x0[(cols == 1 ? 0 : (size_t)b * nx) + i];
    HIP_TRY(hipMemcpy(d_x0, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}

int Solver::set_ref(bool is_x, const double *ref, int cols) {
    const int kn = is_x ? N : N - 1;       // knots
    const int rows = is_x ? nx : nu;
    int kind;
    if (cols == kn)
        kind = 1;
    else if ((long)cols == (long)kn * batch)
        kind = 2;
    else {
        set_error(is_x ? "set_x_ref: expected nx x N or nx x (N*batch)"
                       : "set_u_ref: expected nu x (N-1) or nu x ((N-1)*batch)");
        return -1;
    }
    std::vector<float> &h = is_x ? h_xref : h_uref;
    const size_t n = (size_t)rows * cols;
    h.resize(n);
    bool all_zero = true;
    for (size_t i = 0; i < n; ++i) {
        h[i] = (float)ref[i];
        all_zero = all_zero && (ref[i] == 0.0);
    }
    if (all_zero) {
        kind = 0;  // identical arithmetic (-(0*Q) - rho(..) == 0 - rho(..)), cheaper kernel
        h.clear();
    }
    (is_x ? xref_kind : uref_kind) = kind;
    ref_seq_steps = 0;  // (a closed loop's per-step references go with the references they replaced)
    refs_dirty = true;
    refs_device_owned = false;
    return 0;
}

// Shared references of every step of the next fused closed loop (rocket_landing_constraints.jl:107-115 shifts x_ref by one
// knot per step): x_seq is nx x (N steps), u_seq nu x ((N-1) steps), column-major like set_x_ref / set_u_ref, step after
// step.  Step 0's become the solver's references; steps = 0 clears the sequence.
int Solver::set_ref_sequence(const double *x_seq, const double *u_seq, int steps) {
    if (steps <= 0) {
        ref_seq_steps = 0;
        return 0;
    }
    if (!x_seq || !u_seq) {
        set_error("set_ref_sequence: null reference sequence");
        return -1;
    }
    HIP_TRY(hipSetDevice(device));
    if (wait_last_launch()) return -1;
    const size_t EX = (size_t)ex(), EU = (size_t)eu();
    std::vector<float> hx(EX * steps), hu(EU * steps);
    for (size_t i = 0; i < hx.size(); ++i) hx[i] = (float)x_seq[i];
    for (size_t i = 0; i < hu.size(); ++i) hu[i] = (float)u_seq[i];
    if (dev_alloc(d_xref_seq, hx.size()) || dev_alloc(d_uref_seq, hu.size())) return -1;
    HIP_TRY(hipMemcpy(d_xref_seq, hx.data(), hx.size() * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_uref_seq, hu.data(), hu.size() * sizeof(float), hipMemcpyHostToDevice));
    // the solver's own references = step 0's, kept in shared mode even when they are all zero (later steps need not be)
    h_xref.assign(hx.begin(), hx.begin() + EX);
    h_uref.assign(hu.begin(), hu.begin() + EU);
    xref_kind = uref_kind = 1;
    refs_dirty = true;
    refs_device_owned = false;
    ref_seq_steps = steps;
    return 0;
}

// Materialise h_xref/h_uref on the device in one common mode for the kernel.
int Solver::upload_refs() {
    if (refs_device_owned || !refs_dirty) return 0;
    HIP_TRY(hipSetDevice(device));
    if (wait_last_launch()) return -1;
    const int mode = xref_kind > uref_kind ? xref_kind : uref_kind;
    const size_t EX = (size_t)ex(), EU = (size_t)eu(), Bn = (size_t)batch;
    auto put = [&](float *&dptr, size_t &cap, const std::vector<float> &h, int kind, size_t E) -> int {
        const size_t need = mode == REF_PER_INSTANCE ? Bn * E : E;
        if (cap < need) {
            if (dev_alloc(dptr, need)) return -1;
            cap = need;
        }
        if (mode == REF_ZERO) return 0;
        std::vector<float> tmp;
        const float *src;
        if (kind == mode) {
            src = h.data();
        } else {
            tmp.assign(need, 0.f);
            if (kind == 1)  // shared -> replicate per instance
                for (size_t b = 0; b < Bn; ++b) std::memcpy(tmp.data() + b * E, h.data(), E * sizeof(float));
            src = tmp.data();
        }
        HIP_TRY(hipMemcpy(dptr, src, need * sizeof(float), hipMemcpyHostToDevice));
        return 0;
    };
    if (put(d_xref, xref_cap, h_xref, xref_kind, EX)) return -1;
    if (put(d_uref, uref_cap, h_uref, uref_kind, EU)) return -1;
    ref_mode = mode;
    refs_dirty = false;
    return 0;
}

int Solver::set_bounds(const double *xmin, const double *xmax, const double *umin, const double *umax) {
    route_gen += 1;   // (array-valued state the routes read: the routing key only carries scalars)
    x_min.assign(xmin, xmin + ex());
    x_max.assign(xmax, xmax + ex());
    u_min.assign(umin, umin + eu());
    u_max.assign(umax, umax + eu());
    st.en_state_bound = 1;  // bindings.cpp:400-404
    st.en_input_bound = 1;
    packs_dirty = true;
    return 0;
}

int Solver::set_fdyn(const double *f) {
    route_gen += 1;   // (array-valued state the routes read: the routing key only carries scalars)
    bool nz = false;
    for (int i = 0; i < nx; ++i) {
        fdyn[i] = f ? f[i] : 0.0;
        nz = nz || fdyn[i] != 0.0;
    }
    has_fdyn = nz;
    packs_dirty = true;
    return select_kernel() || ensure_extension_buffers();
}

int Solver::set_cones(const int *Acu_, const int *qcu_, const double *cu_, int ncu_, const int *Acx_,
                      const int *qcx_, const double *cx_, int ncx_) {
    route_gen += 1;
    if (ncu_ > 8 || ncx_ > 8) {
        set_error("set_cone_constraints: at most 8 cones per knot and side");
        return -1;
    }
    for (int i = 0; i < ncu_; ++i)
        if (Acu_[i] < 0 || qcu_[i] < 2 || Acu_[i] + qcu_[i] > nu || !(cu_[i] > 0.0)) {
            set_error("set_cone_constraints: input cone out of range (need 0 <= Ac, qc >= 2, Ac + qc <= nu, mu > 0)");
            return -1;
        }
    for (int i = 0; i < ncx_; ++i)
        if (Acx_[i] < 0 || qcx_[i] < 2 || Acx_[i] + qcx_[i] > nx || !(cx_[i] > 0.0)) {
            set_error("set_cone_constraints: state cone out of range (need 0 <= Ac, qc >= 2, Ac + qc <= nx, mu > 0)");
            return -1;
        }
    ncu = ncu_;
    ncx = ncx_;
    for (int i = 0; i < ncu; ++i) {
        Acu[i] = Acu_[i];
        qcu[i] = qcu_[i];
        cu[i] = cu_[i];
    }
    for (int i = 0; i < ncx; ++i) {
        Acx[i] = Acx_[i];
        qcx[i] = qcx_[i];
        cx[i] = cx_[i];
    }
    if (ncu > 0) st.en_input_soc = 1;  // bindings.cpp:478-483: only the non-empty halves are enabled
    if (ncx > 0) st.en_state_soc = 1;
    return select_kernel() || ensure_extension_buffers();
}

// UNPINNED (bindings.cpp:413-450): Alin_x (mx x nx), Alin_u (mu x nu) column-major; enables the non-empty halves
int Solver::set_linear(const double *Ax, int mx, const double *bx, const double *Au, int mu, const double *bu) {
    route_gen += 1;   // (array-valued state the routes read: the routing key only carries scalars)
    if (mx < 0 || mu < 0 || mx > LIN_MAX_ROWS || mu > LIN_MAX_ROWS) {
        set_error("set_linear_constraints: at most 8 rows per side");
        return -1;
    }
    if ((mx > 0 && (!Ax || !bx)) || (mu > 0 && (!Au || !bu))) {
        set_error("set_linear_constraints: null matrix or right-hand side");
        return -1;
    }
    auto take = [&](const double *A, const double *b, int m, int n, std::vector<double> &Ar, std::vector<double> &br) {
        Ar.assign((size_t)m * n, 0.0);
        br.assign((size_t)m, 0.0);
        for (int k = 0; k < m; ++k) {
            double n2 = 0.0;
            for (int j = 0; j < n; ++j) {
                Ar[(size_t)k * n + j] = A[k + (size_t)j * m];
                n2 += A[k + (size_t)j * m] * A[k + (size_t)j * m];
            }
            if (!(n2 > 0.0)) return false;
            br[k] = b[k];
        }
        return true;
    };
    std::vector<double> ax, bxv, au, buv;
    if (!take(Ax, bx, mx, nx, ax, bxv) || !take(Au, bu, mu, nu, au, buv)) {
        set_error("set_linear_constraints: a constraint row is all zero");
        return -1;
    }
    mlx = mx;
    mlu = mu;
    lin_Ax.swap(ax);
    lin_bx.swap(bxv);
    lin_Au.swap(au);
    lin_bu.swap(buv);
    lin_dirty = true;
    if (mlx > 0) st.en_state_linear = 1;  // bindings.cpp:438-443: only the non-empty halves are enabled
    if (mlu > 0) st.en_input_linear = 1;
    return select_kernel() || ensure_extension_buffers();
}

// Scratch and cone / linear warm-start buffers of the stream and generic kernels, sized for the current batch / options.
int Solver::ensure_extension_buffers() {
    if (ke && !st.adaptive_rho) return 0;
    HIP_TRY(hipSetDevice(device));
    const size_t Bn = (size_t)batch, EX = (size_t)ex(), EU = (size_t)eu();
    if (st.adaptive_rho) {
        const size_t nk = (size_t)nu * nx, np = (size_t)nx * nx;
        if (!sens_set) {  // what the reference's host would compute and hand over (TinyMPC.jl:301-323)
            Mat dK, dP, dC1, dC2;
            if (compute_sensitivity(A, B, Q, R, cache.rho, dK, dP, dC1, dC2)) {
                set_error("adaptive_rho: sensitivity computation failed (singular R + rho I + B'PB)");
                return -1;
            }
            set_sensitivity(dK.a.data(), dP.a.data());
        }
        if (sens_dirty) {
            if (dev_alloc(d_sens, nk + np)) return -1;
            HIP_TRY(hipMemcpy(d_sens, sens.data(), (nk + np) * sizeof(double), hipMemcpyHostToDevice));
            sens_dirty = false;
        }
        if (!d_adapt && dev_alloc(d_adapt, (1 + nk + np) * Bn)) return -1;
        if (adapt_dirty) {
            std::vector<double> fam(1 + nk + np);
            fam[0] = cache.rho;
            std::copy(cache.Kinf.a.begin(), cache.Kinf.a.end(), fam.begin() + 1);
            std::copy(cache.Pinf.a.begin(), cache.Pinf.a.end(), fam.begin() + 1 + nk);
            double *d_fam = nullptr;
            if (dev_alloc(d_fam, fam.size())) return -1;
            HIP_TRY(hipMemcpy(d_fam, fam.data(), fam.size() * sizeof(double), hipMemcpyHostToDevice));
            const long total = (long)fam.size() * (long)Bn;
            adapt_fill_kernel<<<(unsigned)((total + 255) / 256), 256>>>(d_adapt, d_fam, (int)fam.size(), (long)Bn);
            HIP_TRY(hipDeviceSynchronize());
            dev_free(d_fam);
            adapt_dirty = false;
            adapt_pure = true;
        }
        if (se) {  // stream kernel: the per-lane columns of the rows built from Kinf / Pinf (kernel-local scratch)
            const size_t G = (size_t)se->lanes, rx = (nx + G - 1) / G, ru = (nu + G - 1) / G;
            const size_t len = ru * (G * rx) + rx * (G * ru) + rx * (G * rx);
            const size_t bytes = len * G * Bn * (precision == 0 ? 8 : 4);
            if (bytes > adp_cols_bytes) {
                dev_free(d_adp_cols);
                if (dev_alloc(d_adp_cols, bytes)) return -1;
                adp_cols_bytes = bytes;
            }
        }
    }
    if (ke) return 0;  // (a quad kernel running an adaptive solve: the adaptive state above is all it needs)
    const size_t sets = (size_t)constraint_sets();
    size_t need = Bn * ((2 + 3 * sets) * EX + (3 + 3 * sets) * EU);  // generic kernel: admm_generic.hip.h
    if (precision == 2) {   // fp64 state: the scratch arrays are doubles, and the workspace kept between solves is the fp64 block
        need *= 2;
        const size_t w = (size_t)Ws64::doubles((long)Bn, (long)EX, (long)EU, 3);
        if (ws64_cap < w) {
            dev_free(d_ws64);
            if (dev_alloc(d_ws64, w)) return -1;
            HIP_TRY(hipMemset(d_ws64, 0, w * sizeof(double)));
            ws64_cap = w;
        }
    }
    if (se) need = std::max(need, Bn * se->scratch_floats(N, (int)sets));
    if (ce) need = ce->scratch_floats(*this);
    if (scratch_cap < need) {
        if (dev_alloc(d_scratch, need)) return -1;
        scratch_cap = need;
    }
    if (cones_active() && !d_sgc) {
        if (dev_alloc(d_sgc, Bn * EX) || dev_alloc(d_svc, Bn * EX) || dev_alloc(d_syc, Bn * EU) ||
            dev_alloc(d_szc, Bn * EU))
            return -1;
        HIP_TRY(hipMemset(d_sgc, 0, Bn * EX * sizeof(float)));
        HIP_TRY(hipMemset(d_svc, 0, Bn * EX * sizeof(float)));
        HIP_TRY(hipMemset(d_syc, 0, Bn * EU * sizeof(float)));
        HIP_TRY(hipMemset(d_szc, 0, Bn * EU * sizeof(float)));
    }
    if (lin_active() && !d_sgl) {
        if (dev_alloc(d_sgl, Bn * EX) || dev_alloc(d_svl, Bn * EX) || dev_alloc(d_syl, Bn * EU) ||
            dev_alloc(d_szl, Bn * EU))
            return -1;
        HIP_TRY(hipMemset(d_sgl, 0, Bn * EX * sizeof(float)));
        HIP_TRY(hipMemset(d_svl, 0, Bn * EX * sizeof(float)));
        HIP_TRY(hipMemset(d_syl, 0, Bn * EU * sizeof(float)));
        HIP_TRY(hipMemset(d_szl, 0, Bn * EU * sizeof(float)));
    }
    if (lin_dirty || (lin_active() && !d_lin)) {
        // [mlx][nx] rows | b | |a|^2 | [mlu][nu] rows | b | |a|^2   (fp32)
        std::vector<float> pk;
        auto put = [&](const std::vector<double> &A, const std::vector<double> &b, int m, int n) {
            for (size_t i = 0; i < (size_t)m * n; ++i) pk.push_back((float)A[i]);
            for (int k = 0; k < m; ++k) pk.push_back((float)b[k]);
            for (int k = 0; k < m; ++k) {
                double n2 = 0.0;
                for (int j = 0; j < n; ++j) n2 += A[(size_t)k * n + j] * A[(size_t)k * n + j];
                pk.push_back((float)n2);
            }
        };
        // (a side that has rows but is switched off — tinympc_enable_linear — is left out: the kernels find the input block
        // behind P.mlx state rows)
        put(lin_Ax, lin_bx, st.en_state_linear ? mlx : 0, nx);
        put(lin_Au, lin_bu, st.en_input_linear ? mlu : 0, nu);
        if (pk.empty()) pk.push_back(0.f);
        if (dev_alloc(d_lin, pk.size())) return -1;
        HIP_TRY(hipMemcpy(d_lin, pk.data(), pk.size() * sizeof(float), hipMemcpyHostToDevice));
        lin_dirty = false;
    }
    return 0;
}

// appends the ids of the instances that have not converged to `out` (order unspecified)
__global__ void compact_unsolved_kernel(const int *solved, const int *idx_in, int n_in, int *idx_out, int *count) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_in) return;
    const int b = idx_in ? idx_in[j] : j;
    if (!solved[b]) idx_out[atomicAdd(count, 1)] = b;
}

// status words 0..3 of a chunked solve: the maxima of every instance's FINAL residuals (what one launch would have
// folded), rebuilt from the per-instance residual array — the chunks' own blocks also hold the larger residuals that
// late-converging instances had at earlier chunk ends
__global__ void residual_max_kernel(const float *res, long batch, uint32_t *gstat) {
    float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f;
    for (long b = (long)blockIdx.x * blockDim.x + threadIdx.x; b < batch; b += (long)gridDim.x * blockDim.x) {
        const float4 r = reinterpret_cast<const float4 *>(res)[b];
        m0 = fmaxf(m0, r.x), m1 = fmaxf(m1, r.y), m2 = fmaxf(m2, r.z), m3 = fmaxf(m3, r.w);
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        m0 = fmaxf(m0, __shfl_xor(m0, off, 64));
        m1 = fmaxf(m1, __shfl_xor(m1, off, 64));
        m2 = fmaxf(m2, __shfl_xor(m2, off, 64));
        m3 = fmaxf(m3, __shfl_xor(m3, off, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMax(&gstat[0], __float_as_uint(m0));
        atomicMax(&gstat[1], __float_as_uint(m1));
        atomicMax(&gstat[2], __float_as_uint(m2));
        atomicMax(&gstat[3], __float_as_uint(m3));
    }
}

int Solver::solve_async(hipStream_t stream, int mpc_steps) {
    HIP_TRY(hipSetDevice(device));
    layout_final = true;
    if (select_kernel(mpc_steps > 0) || ensure_extension_buffers()) return -1;
    const bool chunkable = chunk_iters > 0 && mpc_steps == 0 && !hetero && (ke || se || (ce && ce->ws)) && st.check_termination > 0 &&
                           st.abs_pri_tol > 0.0 && st.abs_dua_tol > 0.0 && st.max_iter > chunk_iters;
    if (chunkable) return solve_chunked(stream);
    if (mpc_steps > 0 && ke && ke->G == 16) return rollout_steps(stream, mpc_steps);
    return launch_pass(stream, mpc_steps, nullptr, batch, 0, st.max_iter, !warm_start, warm_start);
}

// Chunks of (a multiple of check_termination) iterations; between chunks the unconverged instances are gathered
// into a dense index list on the device, so the next launch has no idle lanes for the finished ones.  The warm-start
// workspace carries every instance from chunk to chunk exactly (it is what the kernel holds, fp32), so the iterates,
// iteration counts and residuals are those of the single-launch solve.  Synchronises the stream between chunks.
int Solver::solve_chunked(hipStream_t stream) {
    const int ct = st.check_termination;
    const int chunk = std::max(ct, (chunk_iters + ct - 1) / ct * ct);
    if (!d_idx[0]) {
        if (dev_alloc(d_idx[0], (size_t)batch) || dev_alloc(d_idx[1], (size_t)batch) || dev_alloc(d_count, 1)) return -1;
    }
    uint32_t acc[GSTAT_WORDS] = {0};
    const int *idx = nullptr;
    int n = batch, offset = 0, cur = 0;
    for (;;) {
        const int iters = std::min(chunk, st.max_iter - offset);
        if (launch_pass(stream, 0, idx, n, offset, iters, offset == 0 && !warm_start, true)) return -1;
        offset += iters;
        HIP_TRY(hipMemcpyAsync(h_gstat, d_gstat, GSTAT_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        const bool last = offset >= st.max_iter;
        int count = 0;
        if (!last) {
            HIP_TRY(hipMemsetAsync(d_count, 0, sizeof(int), stream));
            hipLaunchKernelGGL(compact_unsolved_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, d_solved, idx, n,
                               d_idx[cur], d_count);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(&count, d_count, sizeof(int), hipMemcpyDeviceToHost, stream));
        }
        HIP_TRY(hipStreamSynchronize(stream));
        acc[4] = h_gstat[4];                                                // unsolved: what the latest chunk left
        if (last || count == 0) {
            if (count == 0 && !last) acc[4] = 0;
            break;
        }
        idx = d_idx[cur];
        n = count;
        cur ^= 1;
    }
    // the public block: unsolved count of the last chunk; residual maxima over every instance's final residuals
    HIP_TRY(hipMemcpyAsync(d_gstat, acc, GSTAT_WORDS * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(residual_max_kernel, dim3((unsigned)std::min<long>(1024, ((long)batch + 255) / 256)), dim3(256), 0,
                       stream, d_res, (long)batch, d_gstat);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(stream));
    return 0;
}

int Solver::launch_pass(hipStream_t stream, int mpc_steps, const int *idx, int n_slots, int iter_offset, int max_iter_pass,
                        bool cold, bool save) {
    if (mpc_steps > 0) {
        if (!ke && !(ce && ce->ws)) {
            set_error("mpc_rollout: this problem shape / option set has no kernel with a fused closed loop");
            return -1;
        }
        if (!warm_start) {
            set_error("mpc_rollout needs the persistent workspace (set_warm_start(1))");
            return -1;
        }
        if (mpc_cap < mpc_steps) {
            const size_t n = (size_t)batch * mpc_steps;
            if (dev_alloc(d_mpc_x, n * nx) || dev_alloc(d_mpc_u, n * nu) || dev_alloc(d_mpc_iter, n)) return -1;
            mpc_cap = mpc_steps;
        }
        mpc_steps_last = mpc_steps;
    }
    if (packs_dirty && upload_packs()) return -1;
    if (upload_refs()) return -1;
    AdmmParams P;
    std::memset(&P, 0, sizeof(P));
    P.coef = reinterpret_cast<const float *>(d_coef);
    P.bounds = d_bounds;
    P.x0 = d_x0;
    P.x0d = x0d_launch;
    P.xref = d_xref;
    P.uref = d_uref;
    P.xout = d_xout;
    P.uout = d_uout;
    P.iter = d_iter;
    P.solved = d_solved;
    P.res = d_res;
    P.sd = d_sd;
    P.sy = d_sy;
    P.sz = d_sz;
    P.sg = d_sg;
    P.sv = d_sv;
    P.gstat = d_gstat;
    P.gacc = d_gstat + GSTAT_WORDS;
    P.scratch = d_scratch;
    P.idx = idx;
    P.iter_offset = iter_offset;
    P.batch = n_slots;
    P.max_iter = max_iter_pass;
    P.check_termination = st.check_termination;
    P.ref_mode = ref_mode;
    P.cold_start = cold ? 1 : 0;
    P.save_state = save ? 1 : 0;
    P.abs_pri_tol = (float)st.abs_pri_tol;
    P.abs_dua_tol = (float)st.abs_dua_tol;
    P.rho = (float)cache.rho;
    P.nx = nx;
    P.nu = nu;
    P.N = N;
    P.mpc_steps = mpc_steps;
    P.mpc_x = d_mpc_x;
    P.mpc_u = d_mpc_u;
    P.mpc_iter = d_mpc_iter;
    P.x0_out = d_x0;
    P.xb_active = state_bounds_active ? 1 : 0;
    P.has_fdyn = has_fdyn ? 1 : 0;
    P.ncx = st.en_state_soc ? ncx : 0;
    P.ncu = st.en_input_soc ? ncu : 0;
    for (int i = 0; i < 8; ++i) {
        P.Acx[i] = Acx[i];
        P.qcx[i] = qcx[i];
        P.cx[i] = (float)cx[i];
        P.Acu[i] = Acu[i];
        P.qcu[i] = qcu[i];
        P.cu[i] = (float)cu[i];
    }
    P.het_aux = d_het_aux;
    P.adaptive_rho = st.adaptive_rho;
    P.rho_clip = st.adaptive_rho_clip;
    P.rho_family = cache.rho;
    P.rho_min = (float)st.adaptive_rho_min;
    P.rho_max = (float)st.adaptive_rho_max;
    P.sens = d_sens;
    P.adapt = d_adapt;
    P.adapt_stride = batch;
    P.adp_cols = d_adp_cols;
    P.sgc = d_sgc;
    P.svc = d_svc;
    P.syc = d_syc;
    P.szc = d_szc;
    P.mlx = st.en_state_linear ? mlx : 0;
    P.mlu = st.en_input_linear ? mlu : 0;
    P.lin = d_lin;
    P.sgl = d_sgl;
    P.svl = d_svl;
    P.syl = d_syl;
    P.szl = d_szl;
    // the quad and stream kernels keep the status block clean themselves (fold_status); the generic kernel
    // accumulates straight into it
    P.bounds_stride = (ce && ce->bounds_vary(*this)) ? 1 : 0;
    if (mpc_steps > 0 && ref_seq_steps > 0) {
        if (!(ce && ce->ws) || ref_seq_steps < mpc_steps || ref_mode != REF_SHARED) {
            set_error("mpc_rollout: per-step references need the transposed-sets kernel (mfmat) and one reference set per step");
            return -1;
        }
        P.xref_seq = d_xref_seq;
        P.uref_seq = d_uref_seq;
    }
    if (!ke && !se && !ce) HIP_TRY(hipMemsetAsync(d_gstat, 0, GSTAT_WORDS * sizeof(uint32_t), stream));
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (profiling) {
        if (ev_ring.empty()) {
            ev_ring.assign(2 * EV_RING, nullptr);
            for (hipEvent_t &e : ev_ring) HIP_TRY(hipEventCreate(&e));
        }
        ev0 = ev_ring[2 * (launches % EV_RING)];
        ev1 = ev_ring[2 * (launches % EV_RING) + 1];
        HIP_TRY(hipEventRecord(ev0, stream));
    }
    // the workspace's state dual is non-zero only if some solve since the last reset had a finite state bound (or the
    // caller wrote one in): the matrix-core kernel (G == 16) carries g only then
    if (state_bounds_active && save) g_maybe_nonzero = true;
    const bool carry_g = state_bounds_active || (ke && ke->G == 16 && g_maybe_nonzero);
    // one-shot solves (cold start, nothing of the workspace kept) of a one-lane-per-instance entry, zero or shared references,
    // fp64 recurrences: the lean kernel (same arithmetic, a third fewer instructions)
    const bool lean_f64 = precision == 2 && !ke && lean_jit;
    const bool lean_call = (ke ? precision == 0 : lean_f64) && lean_ok && cold && !save && mpc_steps == 0 && !idx &&
                           ref_mode != REF_PER_INSTANCE && !st.adaptive_rho && max_iter_pass >= 1;
    const bool lean_live = st.abs_pri_tol > 0.0 && st.abs_dua_tol > 0.0;
    const LeanEntry *lk = lean_call ? le : nullptr;
    if (lean_call && !le && lean_jit) {
        // the one variant this launch needs (lean_entry.hip.h: launch_lean_v's choices), compiled on first use
        bool one = (P.batch + 255) / 256 <= device_cu_count() || lean_live || sw.lean_one || lean_f64;
        if (2 * N * nx + 3 * N * nu + 50 > 250) one = true;   // (the 256-register form does not hold this horizon)
        const int v = (lean_live ? LV_LIVE : 0) | (lean_knot_bounds ? 0 : LV_UBK) | (one ? LV_ONE : 0) | (state_bounds_active ? LV_XB : 0) |
                      (ref_mode == REF_SHARED ? LV_SHARED : 0) | (lean_f64 ? LV_F64 : 0);
        if (!le_var_tried[v]) le_var[v] = jit_lean_for(nx, nu, N, v, verbose), le_var_tried[v] = true;
        lk = le_var[v];
    }
    const bool lean = lk != nullptr;
    if (lean && !le && (!ke || ke->G != 1)) P.bounds = reinterpret_cast<const float *>(d_lean + lean_layout(nx, nu).total);   // (upload_packs)
    P.lean = d_lean;
    P.ws64 = d_ws64;
    P.abs_pri_tol64 = st.abs_pri_tol;
    P.abs_dua_tol64 = st.abs_dua_tol;
    P.host_flags = (sw.no_refill ? HF_NO_REFILL : 0) | (sw.no_uni ? HF_NO_UNI : 0) | (sw.no_os ? HF_NO_OS : 0) | (sw.lean_one ? HF_LEAN_ONE : 0);
    last_launch_name = lean ? lk->name : kernel_name;
    if (lean) {
        HIP_TRY(lk->launch(P, lean_live, lean_knot_bounds, state_bounds_active, stream));
    } else
    HIP_TRY(ke ? ke->launch(P, precision, carry_g, stream)
               : (ce ? ce->launch(P, cones_active(), ce->lds_bytes(*this), stream)
                     : (se ? se->launch(P, precision, lin_active() ? 2 : ((has_fdyn || cones_active()) ? 1 : 0), hetero, stream)
                           : launch_generic(P, precision, stream))));
    if (profiling) {
        HIP_TRY(hipEventRecord(ev1, stream));
        launches += 1;
    }
    // the status block stays on the device; solve_status() fetches it when asked (nothing but the kernel and the
    // 32-byte clear is enqueued per solve)
    solved_once = true;
    if (!ev_done) HIP_TRY(hipEventCreateWithFlags(&ev_done, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(ev_done, stream));
    ev_done_pending = true;
    return 0;
}

// x0d <- x0 (start of a closed loop)
__global__ void plant_init_kernel(double *x0d, const float *x0, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x0d[i] = (double)x0[i];
}

// One closed-loop step after a solve (cartpole_example_mpc.jl:35-51): u0 = first column of the solution,
// x0 <- A x0 + B u0 in fp64, logged like the quad kernel's fused loop does (iteration count signed by the solved flag).
__global__ void plant_step_kernel(double *x0d, float *x0, const float *uout, const int *iter, const int *solved,
                                  const double *AB, float *mpc_x, float *mpc_u, int *mpc_iter, int nx, int nu, int N,
                                  long batch, int steps, int step) {
    const long b = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    const double *A = AB, *Bm = AB + nx * nx;
    const float *u0 = uout + b * (long)nu * (N - 1);
    const long so = b * steps + step;
    double xn[GEN_MAX_NX];
    for (int r = 0; r < nx; ++r) {
        double acc = 0.0;
        for (int j = 0; j < nx; ++j) acc = fma(A[r + j * nx], x0d[b * nx + j], acc);
        for (int a = 0; a < nu; ++a) acc = fma(Bm[r + a * nx], (double)u0[a], acc);
        xn[r] = acc;
    }
    for (int r = 0; r < nx; ++r) {
        x0d[b * nx + r] = xn[r];
        x0[b * nx + r] = (float)xn[r];
        mpc_x[so * nx + r] = (float)xn[r];
    }
    for (int a = 0; a < nu; ++a) mpc_u[so * nu + a] = u0[a];
    mpc_iter[so] = solved[b] ? iter[b] : -iter[b];
}

int Solver::rollout_steps(hipStream_t stream, int mpc_steps) {
    if (!warm_start) {
        set_error("mpc_rollout needs the persistent workspace (set_warm_start(1))");
        return -1;
    }
    const size_t Bn = (size_t)batch;
    if (mpc_cap < mpc_steps) {
        const size_t n = Bn * mpc_steps;
        if (dev_alloc(d_mpc_x, n * nx) || dev_alloc(d_mpc_u, n * nu) || dev_alloc(d_mpc_iter, n)) return -1;
        mpc_cap = mpc_steps;
    }
    if (!d_x0d && dev_alloc(d_x0d, Bn * nx)) return -1;
    if (!d_plant) {
        if (dev_alloc(d_plant, (size_t)nx * nx + (size_t)nx * nu)) return -1;
        std::vector<double> ab(A.a);
        ab.insert(ab.end(), B.a.begin(), B.a.end());
        HIP_TRY(hipMemcpy(d_plant, ab.data(), ab.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    const long n0 = (long)Bn * nx;
    plant_init_kernel<<<(unsigned)((n0 + 255) / 256), 256, 0, stream>>>(d_x0d, d_x0, n0);
    for (int step = 0; step < mpc_steps; ++step) {
        x0d_launch = d_x0d;
        const int rc = launch_pass(stream, 0, nullptr, batch, 0, st.max_iter, false, true);
        x0d_launch = nullptr;
        if (rc) return -1;
        plant_step_kernel<<<(unsigned)((Bn + 255) / 256), 256, 0, stream>>>(d_x0d, d_x0, d_uout, d_iter, d_solved, d_plant,
                                                                        d_mpc_x, d_mpc_u, d_mpc_iter, nx, nu, N,
                                                                        (long)Bn, mpc_steps, step);
    }
    HIP_TRY(hipGetLastError());
    mpc_steps_last = mpc_steps;
    if (!ev_done) HIP_TRY(hipEventCreateWithFlags(&ev_done, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(ev_done, stream));
    ev_done_pending = true;
    return 0;
}

int Solver::solve_status() {
    if (!solved_once) {
        set_error("solve_status before any solve");
        return -1;
    }
    HIP_TRY(hipSetDevice(device));
    if (wait_last_launch()) return -1;
    HIP_TRY(hipMemcpy(h_gstat, d_gstat, GSTAT_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return h_gstat[4] == 0 ? 0 : 1;  // admm.cpp:192 / :206 folded over the batch
}

int Solver::get_mpc_log(double *x, double *u, int *iter) {
    if (mpc_steps_last <= 0) {
        set_error("get_mpc_log: no mpc_rollout has run");
        return -1;
    }
    HIP_TRY(hipSetDevice(device));
    if (wait_last_launch()) return -1;
    const size_t n = (size_t)batch * mpc_steps_last;
    std::vector<float> h;
    if (x) {
        h.resize(n * nx);
        HIP_TRY(hipMemcpy(h.data(), d_mpc_x, h.size() * sizeof(float), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < h.size(); ++i) x[i] = (double)h[i];
    }
    if (u) {
        h.resize(n * nu);
        HIP_TRY(hipMemcpy(h.data(), d_mpc_u, h.size() * sizeof(float), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < h.size(); ++i) u[i] = (double)h[i];
    }
    if (iter) HIP_TRY(hipMemcpy(iter, d_mpc_iter, n * sizeof(int), hipMemcpyDeviceToHost));
    return 0;
}

double Solver::kernel_elapsed_ms() { return kernel_elapsed_mean_ms(1); }

// mean duration of the last `last_n` launches (at most EV_RING), from the event pairs recorded around each
double Solver::kernel_elapsed_mean_ms(int last_n) {
    if (!profiling || ev_ring.empty() || launches == 0 || last_n <= 0) return -1.0;
    const long n = std::min<long>(std::min<long>(last_n, EV_RING), launches);
    double sum = 0.0;
    for (long i = launches - n; i < launches; ++i) {
        float ms = -1.f;
        if (hipEventElapsedTime(&ms, ev_ring[2 * (i % EV_RING)], ev_ring[2 * (i % EV_RING) + 1]) != hipSuccess) return -1.0;
        sum += ms;
    }
    return sum / (double)n;
}

int Solver::get_traj(bool states, double *buf) {
    HIP_TRY(hipSetDevice(device));
    if (wait_last_launch()) return -1;
    return d2h_double(states ? d_xout : d_uout, buf, (size_t)batch * (states ? ex() : eu()));
}

int Solver::get_status(int *iter, int *solved, double *res4) {
    HIP_TRY(hipSetDevice(device));
    if (wait_last_launch()) return -1;
    const size_t Bn = (size_t)batch;
    if (iter) HIP_TRY(hipMemcpy(iter, d_iter, Bn * sizeof(int), hipMemcpyDeviceToHost));
    if (solved) HIP_TRY(hipMemcpy(solved, d_solved, Bn * sizeof(int), hipMemcpyDeviceToHost));
    if (res4) {
        std::vector<float> h(Bn * 4);
        HIP_TRY(hipMemcpy(h.data(), d_res, Bn * 4 * sizeof(float), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < Bn * 4; ++i) res4[i] = (double)h[i];
    }
    return 0;
}

// fp32 device buffer -> fp64 caller buffer.  Chunks travel into pinned staging on a copy stream while the previous
// chunk is widened by the host threads, so the round trip costs about max(PCIe, widening) instead of their sum plus
// a pageable-memory bounce.
int Solver::d2h_double(const float *d, double *out, size_t n) {
    if (!out || n == 0) return 0;
    constexpr size_t CHUNK = (size_t)1 << 21;  // floats per slot (8 MB)
    if (!s_copy) {
        HIP_TRY(hipStreamCreate(&s_copy));  // callers have already waited for the last launch (wait_last_launch)
        for (hipEvent_t &e : ev_copy) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    if (stage_cap < 2 * CHUNK) {
        if (h_stage) (void)hipHostFree(h_stage);
        h_stage = nullptr;
        HIP_TRY(hipHostMalloc((void **)&h_stage, 2 * CHUNK * sizeof(float), hipHostMallocDefault));
        stage_cap = 2 * CHUNK;
    }
    const size_t nchunks = (n + CHUNK - 1) / CHUNK;
    for (size_t c = 0; c <= nchunks; ++c) {
        if (c < nchunks) {
            const size_t off = c * CHUNK, len = std::min(CHUNK, n - off);
            HIP_TRY(hipMemcpyAsync(h_stage + (c & 1) * CHUNK, d + off, len * sizeof(float), hipMemcpyDeviceToHost, s_copy));
            HIP_TRY(hipEventRecord(ev_copy[c & 1], s_copy));
        }
        if (c > 0) {
            const size_t pc = c - 1, off = pc * CHUNK, len = std::min(CHUNK, n - off);
            HIP_TRY(hipEventSynchronize(ev_copy[pc & 1]));
            widen(h_stage + (pc & 1) * CHUNK, out + off, len);
        }
    }
    return 0;
}
// Page-lock a caller-owned host range so that the fp32 transfers DMA straight into it.  Only on request
// (tinympc_pin_host): the library does not own these buffers and cannot know when they are freed, so it never registers
// them on its own — the caller unpins (tinympc_unpin_host) before releasing the memory; whatever is still registered
// when the solver is destroyed is unregistered there.
int Solver::pin_host_range(void *p, size_t bytes) {
    if (!p || bytes == 0) {
        set_error("pin_host: null or empty range");
        return -1;
    }
    for (const auto &r : pinned_ranges)
        if (r.first == p) {
            if (r.second >= bytes) return 0;
            set_error("pin_host: this address is already pinned with a smaller size; unpin it first");
            return -1;
        }
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipHostRegister(p, bytes, hipHostRegisterDefault));
    pinned_ranges.emplace_back(p, bytes);
    return 0;
}
int Solver::unpin_host_range(void *p) {
    for (size_t i = 0; i < pinned_ranges.size(); ++i)
        if (pinned_ranges[i].first == p) {
            HIP_TRY(hipSetDevice(device));
            if (wait_last_launch()) return -1;
            pinned_ranges.erase(pinned_ranges.begin() + i);
            HIP_TRY(hipHostUnregister(p));
            return 0;
        }
    set_error("unpin_host: this address was not pinned through this solver");
    return -1;
}

int Solver::h2d_float(float *d, const double *in, size_t n) {
    if (!in) return 0;
    std::vector<float> h(n);
    narrow(in, h.data(), n);
    HIP_TRY(hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}

int Solver::get_workspace(double *d, double *y, double *g, double *v, double *z) {
    HIP_TRY(hipSetDevice(device));
    if (wait_last_launch()) return -1;
    const size_t Bn = (size_t)batch, EX = (size_t)ex(), EU = (size_t)eu();
    if (precision == 2 && d_ws64) {   // the fp64 block itself: no widening
        const Ws64 w(d_ws64, (long)Bn, (long)EX, (long)EU);
        auto get = [&](double *dst, const double *src, size_t n) { return !dst || hipMemcpy(dst, src, n * sizeof(double), hipMemcpyDeviceToHost) == hipSuccess; };
        if (!(get(d, w.sd, Bn * EU) && get(y, w.sy, Bn * EU) && get(z, w.sz, Bn * EU) && get(g, w.sg, Bn * EX) && get(v, w.sv, Bn * EX))) {
            set_error("get_workspace: copy failed");
            return -1;
        }
        return 0;
    }
    if (d2h_double(d_sd, d, Bn * EU) || d2h_double(d_sy, y, Bn * EU) || d2h_double(d_sz, z, Bn * EU) ||
        d2h_double(d_sg, g, Bn * EX) || d2h_double(d_sv, v, Bn * EX))
        return -1;
    return 0;
}

int Solver::set_workspace(const double *d, const double *y, const double *g, const double *v,
                          const double *z) {
    HIP_TRY(hipSetDevice(device));
    if (wait_last_launch()) return -1;
    const size_t Bn = (size_t)batch, EX = (size_t)ex(), EU = (size_t)eu();
    if (precision == 2) {
        if (select_kernel() || ensure_extension_buffers()) return -1;
        const Ws64 w(d_ws64, (long)Bn, (long)EX, (long)EU);
        auto put = [&](double *dst, const double *src, size_t n) { return !src || hipMemcpy(dst, src, n * sizeof(double), hipMemcpyHostToDevice) == hipSuccess; };
        if (!(put(w.sd, d, Bn * EU) && put(w.sy, y, Bn * EU) && put(w.sz, z, Bn * EU) && put(w.sg, g, Bn * EX) && put(w.sv, v, Bn * EX))) {
            set_error("set_workspace: copy failed");
            return -1;
        }
        return 0;
    }
    if (h2d_float(d_sd, d, Bn * EU) || h2d_float(d_sy, y, Bn * EU) || h2d_float(d_sz, z, Bn * EU) ||
        h2d_float(d_sg, g, Bn * EX) || h2d_float(d_sv, v, Bn * EX))
        return -1;
    g_maybe_nonzero = true;
    return 0;
}

}  // namespace tmpc
