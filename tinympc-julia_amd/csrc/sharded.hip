// Multi-GPU behind the C-ABI (SURVEY.md 8(b) last row, 8(e)): ONE handle drives n_gpus devices of one node from one
// host process — what a Julia `ccall` host needs to reach 8 GPUs without a process launcher.
//
//   * the batch is cut into contiguous shards (shard i = instances [lo_i, hi_i), sizes differing by at most one —
//     the rule of sharding.shard_range), shard i living entirely on device i: its own tinympc_solver, its own
//     non-blocking HIP stream;
//   * per-instance inputs are scattered and outputs gathered by plain offset arithmetic on the caller's
//     column-major buffers (an instance's block is contiguous, so a shard is one contiguous span);
//   * nothing moves between GPUs for a solve.  The path's one exchange is the solve status: after its solve each
//     device copies its 8-word status block (float bits of the 4 residual maxima, unsolved count) into a fold
//     buffer and the buffers are all-reduced (MAX) over RCCL — ncclCommInitAll once, one grouped
//     ncclAllReduce(ncclUint32, ncclMax) per solve, enqueued on the shards' own streams behind the kernels.
//     32 bytes: latency-bound, xGMI bandwidth is irrelevant.
//
// RCCL is bound at run time (dlopen of librccl.so when the first multi-device handle is created), so single-GPU users
// of libtinympc_hip.so do not load it.  A device list that names the same device twice cannot form an RCCL communicator;
// such a handle (the one-GPU rehearsal the tests use: every other part of the multi-shard path runs for real) folds
// the status on the host instead and says so in tinympc_sharded_fold_backend().
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <memory>
#include <set>
#include <string>
#include <vector>

#include "../../include/tinympc_hip.h"
#include <thread>

#include "solver.h"

using tmpc::hip_ok;
using tmpc::set_error;

namespace {

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool load() {
        if (lib) return true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) {
            set_error(std::string("cannot load librccl.so: ") + dlerror());
            return false;
        }
        auto sym = [&](const char *n) { return dlsym(lib, n); };
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
        AllReduce = reinterpret_cast<decltype(AllReduce)>(sym("ncclAllReduce"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
        if (!CommInitAll || !CommDestroy || !AllReduce || !GroupStart || !GroupEnd || !GetErrorString) {
            set_error("librccl.so lacks an expected symbol");
            dlclose(lib);
            lib = nullptr;      // (a later call must not take the half-bound table for a loaded library)
            return false;
        }
        return true;
    }
};
Rccl g_rccl;

bool nccl_ok(ncclResult_t r, const char *what) {
    if (r == ncclSuccess) return true;
    set_error(std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error"));
    return false;
}

}  // namespace

struct tinympc_sharded {
    int nx = 0, nu = 0, N = 0, batch = 0;
    std::vector<tinympc_solver *> shard;
    std::vector<int> dev, lo, hi;
    std::vector<hipStream_t> stream;
    std::vector<uint32_t *> d_fold;   // [GSTAT_WORDS] per device: the all-reduce operand
    std::vector<ncclComm_t> comm;     // empty: host fold (repeated devices)
    uint32_t h_fold[tmpc::GSTAT_WORDS] = {0};
    bool pending = false;

    int n() const { return (int)shard.size(); }
    ~tinympc_sharded() {
        for (int i = 0; i < n(); ++i) {
            (void)hipSetDevice(dev[i]);
            if (i < (int)stream.size() && stream[i]) (void)hipStreamSynchronize(stream[i]);
        }
        for (ncclComm_t c : comm)
            if (c && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c);
        for (int i = 0; i < n(); ++i) {
            (void)hipSetDevice(dev[i]);
            if (i < (int)d_fold.size() && d_fold[i]) (void)hipFree(d_fold[i]);
            if (i < (int)stream.size() && stream[i]) (void)hipStreamDestroy(stream[i]);
            tinympc_destroy(shard[i]);
        }
    }
};

#define SH_TRY(expr)                        \
    do {                                    \
        if (!hip_ok((expr), #expr)) return -1; \
    } while (0)

extern "C" {

void tinympc_shard_range(int batch, int n_shards, int shard, int *lo, int *hi) {
    // sharding.shard_range: contiguous, sizes differ by at most one, the larger shards first
    if (n_shards < 1 || shard < 0 || shard >= n_shards || batch < 0) {   // (reachable from ctypes / Julia with any integers)
        if (lo) *lo = 0;
        if (hi) *hi = 0;
        return;
    }
    const int base = batch / n_shards, rem = batch % n_shards;
    const int l = shard * base + std::min(shard, rem);
    if (lo) *lo = l;
    if (hi) *hi = l + base + (shard < rem ? 1 : 0);
}

int tinympc_create_sharded(tinympc_sharded **out, const double *A, const double *B, const double *Q, const double *R,
                           double rho, int nx, int nu, int N, int batch, int n_gpus, const int *devices, int verbose) {
    if (!out || !A || !B || !Q || !R) {
        set_error("tinympc_create_sharded: null argument");
        return -1;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
        set_error("no HIP device available (this library has no CPU fallback)");
        return -1;
    }
    if (n_gpus < 1 || batch < n_gpus) {
        set_error("tinympc_create_sharded: need 1 <= n_gpus <= batch");
        return -1;
    }
    std::unique_ptr<tinympc_sharded> s(new tinympc_sharded());
    s->nx = nx, s->nu = nu, s->N = N, s->batch = batch;
    std::set<int> distinct;
    for (int i = 0; i < n_gpus; ++i) {
        const int d = devices ? devices[i] : i;
        if (d < 0 || d >= ndev) {
            set_error("tinympc_create_sharded: device " + std::to_string(d) + " does not exist (" + std::to_string(ndev) +
                      " visible)");
            return -1;
        }
        distinct.insert(d);
        s->dev.push_back(d);
    }
    for (int i = 0; i < n_gpus; ++i) {
        int l, h;
        tinympc_shard_range(batch, n_gpus, i, &l, &h);
        tinympc_solver *loc = nullptr;
        if (tinympc_create(&loc, A, B, Q, R, rho, nx, nu, N, h - l, s->dev[i], verbose && i == 0)) return -1;
        s->shard.push_back(loc);
        s->lo.push_back(l);
        s->hi.push_back(h);
        SH_TRY(hipSetDevice(s->dev[i]));
        hipStream_t st = nullptr;
        SH_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        s->stream.push_back(st);
        uint32_t *f = nullptr;
        SH_TRY(hipMalloc((void **)&f, tmpc::GSTAT_WORDS * sizeof(uint32_t)));
        SH_TRY(hipMemset(f, 0, tmpc::GSTAT_WORDS * sizeof(uint32_t)));
        s->d_fold.push_back(f);
    }
    if ((int)distinct.size() == n_gpus) {
        // one communicator per device, all owned by this process (ncclCommInitAll); a single device is a one-rank clique
        if (!g_rccl.load()) return -1;
        s->comm.assign(n_gpus, nullptr);
        if (!nccl_ok(g_rccl.CommInitAll(s->comm.data(), n_gpus, s->dev.data()), "ncclCommInitAll")) {
            s->comm.clear();
            return -1;
        }
    }
    if (verbose)
        std::printf("tinympc_hip: %d instances on %d device(s), status fold: %s\n", batch, n_gpus,
                    s->comm.empty() ? "host (repeated devices)" : "RCCL all-reduce(MAX)");
    *out = s.release();
    return 0;
}

void tinympc_sharded_destroy(tinympc_sharded *s) { delete s; }

int tinympc_sharded_n_shards(tinympc_sharded *s) { return s ? s->n() : -1; }

const char *tinympc_sharded_fold_backend(tinympc_sharded *s) { return !s ? "" : (s->comm.empty() ? "host" : "rccl"); }

int tinympc_sharded_shard(tinympc_sharded *s, int i, int *device, int *lo, int *hi, tinympc_solver **local) {
    if (!s || i < 0 || i >= s->n()) return -1;
    if (device) *device = s->dev[i];
    if (lo) *lo = s->lo[i];
    if (hi) *hi = s->hi[i];
    if (local) *local = s->shard[i];
    return 0;
}

// family-level settings: the same call on every shard
#define SH_EACH(call)                                  \
    do {                                               \
        if (!s) return -1;                             \
        for (int i = 0; i < s->n(); ++i) {             \
            tinympc_solver *h = s->shard[i];           \
            if (call) return -1;                       \
        }                                              \
        return 0;                                      \
    } while (0)

int tinympc_sharded_update_settings(tinympc_sharded *s, double abs_pri_tol, double abs_dua_tol, int max_iter,
                                    int check_termination, int en_state_bound, int en_input_bound) {
    SH_EACH(tinympc_update_settings(h, abs_pri_tol, abs_dua_tol, max_iter, check_termination, en_state_bound, en_input_bound));
}
int tinympc_sharded_set_bound_constraints(tinympc_sharded *s, const double *x_min, const double *x_max, const double *u_min,
                                          const double *u_max) {
    SH_EACH(tinympc_set_bound_constraints(h, x_min, x_max, u_min, u_max));
}
int tinympc_sharded_set_warm_start(tinympc_sharded *s, int warm_start) { SH_EACH(tinympc_set_warm_start(h, warm_start)); }
int tinympc_sharded_reset(tinympc_sharded *s) { SH_EACH(tinympc_reset(h)); }
int tinympc_sharded_set_precision(tinympc_sharded *s, int precision) { SH_EACH(tinympc_set_precision(h, precision)); }
int tinympc_sharded_set_compaction(tinympc_sharded *s, int chunk_iters) { SH_EACH(tinympc_set_compaction(h, chunk_iters)); }

// per-instance inputs: shard i takes columns [lo_i, hi_i) of the caller's instance-major buffer (or the shared one)
int tinympc_sharded_set_x0(tinympc_sharded *s, const double *x0, int cols) {
    if (!s || !x0) return -1;
    if (cols != 1 && cols != s->batch) {
        set_error("set_x0: expected nx x 1 or nx x batch");
        return -1;
    }
    for (int i = 0; i < s->n(); ++i)
        if (tinympc_set_x0(s->shard[i], cols == 1 ? x0 : x0 + (size_t)s->lo[i] * s->nx, cols == 1 ? 1 : s->hi[i] - s->lo[i]))
            return -1;
    return 0;
}
static int sharded_set_ref(tinympc_sharded *s, bool is_x, const double *ref, int cols) {
    if (!s || !ref) return -1;
    const int kn = is_x ? s->N : s->N - 1, rows = is_x ? s->nx : s->nu;
    const bool shared = cols == kn;
    if (!shared && (long)cols != (long)kn * s->batch) {
        set_error(is_x ? "set_x_ref: expected nx x N or nx x (N*batch)" : "set_u_ref: expected nu x (N-1) or nu x ((N-1)*batch)");
        return -1;
    }
    for (int i = 0; i < s->n(); ++i) {
        const double *p = shared ? ref : ref + (size_t)s->lo[i] * kn * rows;
        const int c = shared ? kn : kn * (s->hi[i] - s->lo[i]);
        if (is_x ? tinympc_set_x_ref(s->shard[i], p, c) : tinympc_set_u_ref(s->shard[i], p, c)) return -1;
    }
    return 0;
}
int tinympc_sharded_set_x_ref(tinympc_sharded *s, const double *x_ref, int cols) { return sharded_set_ref(s, true, x_ref, cols); }
int tinympc_sharded_set_u_ref(tinympc_sharded *s, const double *u_ref, int cols) { return sharded_set_ref(s, false, u_ref, cols); }

// Enqueue every shard's solve on its device's stream, then the status fold behind it; returns without waiting.
int tinympc_sharded_solve_async(tinympc_sharded *s) {
    if (!s) return -1;
    if (s->pending) {
        set_error("tinympc_sharded_solve_async: the previous solve has not been waited for (tinympc_sharded_wait)");
        return -1;
    }
    // A solve in chunks with compaction (tinympc_sharded_set_compaction) synchronises its stream between chunks on the
    // host: enqueued shard after shard, shard i would run to completion before shard i + 1 is even launched and n GPUs
    // would work one after another.  Such solves get one host thread per shard, so the devices run side by side; the call
    // then returns when every shard's solve has finished (only the status fold is still in flight).
    bool chunked = false;
    for (int i = 0; i < s->n(); ++i) chunked = chunked || s->shard[i]->s.chunk_iters > 0;
    if (chunked && s->n() > 1) {
        (void)tmpc::device_cu_count();   // (per-device caches are filled by the first caller: not from several threads at once)
        std::vector<int> rc((size_t)s->n(), 0);
        std::vector<std::string> err((size_t)s->n());   // last_error() is per thread: carried back to the caller's
        std::vector<std::thread> th;
        for (int i = 0; i < s->n(); ++i)
            th.emplace_back([s, i, &rc, &err] {
                if (hipSetDevice(s->dev[i]) != hipSuccess) {
                    rc[(size_t)i] = -1;
                    err[(size_t)i] = "hipSetDevice failed on a shard's worker thread";
                    return;
                }
                rc[(size_t)i] = tinympc_solve_async(s->shard[i], s->stream[i]);
                if (rc[(size_t)i]) err[(size_t)i] = tmpc::last_error();
            });
        for (std::thread &t : th) t.join();
        for (int i = 0; i < s->n(); ++i)
            if (rc[(size_t)i]) {
                // the other shards have run to completion: drain every stream so that the handle's state (nothing
                // pending) matches the devices', then report the first failing shard's message on THIS thread
                for (int j = 0; j < s->n(); ++j)
                    if (hipSetDevice(s->dev[j]) == hipSuccess) (void)hipStreamSynchronize(s->stream[j]);
                set_error("shard " + std::to_string(i) + ": " + err[(size_t)i]);
                return -1;
            }
    } else {
        for (int i = 0; i < s->n(); ++i) {
            SH_TRY(hipSetDevice(s->dev[i]));
            if (tinympc_solve_async(s->shard[i], s->stream[i])) return -1;
        }
    }
    for (int i = 0; i < s->n(); ++i) {
        SH_TRY(hipSetDevice(s->dev[i]));
        SH_TRY(hipMemcpyAsync(s->d_fold[i], s->shard[i]->s.d_gstat, tmpc::GSTAT_WORDS * sizeof(uint32_t),
                              hipMemcpyDeviceToDevice, s->stream[i]));
    }
    if (!s->comm.empty()) {
        if (!nccl_ok(g_rccl.GroupStart(), "ncclGroupStart")) return -1;
        for (int i = 0; i < s->n(); ++i)
            if (!nccl_ok(g_rccl.AllReduce(s->d_fold[i], s->d_fold[i], tmpc::GSTAT_WORDS, ncclUint32, ncclMax, s->comm[i],
                                          s->stream[i]),
                         "ncclAllReduce")) {
                (void)g_rccl.GroupEnd();
                return -1;
            }
        if (!nccl_ok(g_rccl.GroupEnd(), "ncclGroupEnd")) return -1;
    }
    s->pending = true;
    return 0;
}

// Waits for every device and returns the GLOBAL solve status: 0 iff every instance on every device converged, else 1.
int tinympc_sharded_wait(tinympc_sharded *s) {
    if (!s) return -1;
    if (!s->pending) {
        set_error("tinympc_sharded_wait: no solve is pending");
        return -1;
    }
    for (int i = 0; i < s->n(); ++i) {
        SH_TRY(hipSetDevice(s->dev[i]));
        SH_TRY(hipStreamSynchronize(s->stream[i]));
    }
    s->pending = false;
    if (!s->comm.empty()) {  // every device holds the folded block; read device 0's
        SH_TRY(hipSetDevice(s->dev[0]));
        SH_TRY(hipMemcpy(s->h_fold, s->d_fold[0], sizeof(s->h_fold), hipMemcpyDeviceToHost));
    } else {
        std::memset(s->h_fold, 0, sizeof(s->h_fold));
        for (int i = 0; i < s->n(); ++i) {
            uint32_t w[tmpc::GSTAT_WORDS];
            SH_TRY(hipSetDevice(s->dev[i]));
            SH_TRY(hipMemcpy(w, s->d_fold[i], sizeof(w), hipMemcpyDeviceToHost));
            for (int k = 0; k < tmpc::GSTAT_WORDS; ++k) s->h_fold[k] = std::max(s->h_fold[k], w[k]);
        }
    }
    return s->h_fold[4] == 0 ? 0 : 1;
}

int tinympc_sharded_solve(tinympc_sharded *s) {
    if (tinympc_sharded_solve_async(s)) return -1;
    return tinympc_sharded_wait(s);
}

// What the fold produced for the last solve: max over ALL instances of (pri_x, dua_x, pri_u, dua_u) and whether any
// device reported unsolved instances (the word is a MAX over devices of their unsolved counts).
int tinympc_sharded_global_status(tinympc_sharded *s, double *residual_maxima4, int *max_unsolved_per_device) {
    if (!s) return -1;
    if (residual_maxima4)
        for (int k = 0; k < 4; ++k) {
            float f;
            std::memcpy(&f, &s->h_fold[k], sizeof f);
            residual_maxima4[k] = (double)f;
        }
    if (max_unsolved_per_device) *max_unsolved_per_device = (int)s->h_fold[4];
    return 0;
}

// outputs: shard i fills its span of the caller's buffers
int tinympc_sharded_get_states(tinympc_sharded *s, double *buf) {
    if (!s || !buf) return -1;
    for (int i = 0; i < s->n(); ++i)
        if (tinympc_get_states(s->shard[i], buf + (size_t)s->lo[i] * s->nx * s->N)) return -1;
    return 0;
}
int tinympc_sharded_get_controls(tinympc_sharded *s, double *buf) {
    if (!s || !buf) return -1;
    for (int i = 0; i < s->n(); ++i)
        if (tinympc_get_controls(s->shard[i], buf + (size_t)s->lo[i] * s->nu * (s->N - 1))) return -1;
    return 0;
}
int tinympc_sharded_get_status(tinympc_sharded *s, int *iter, int *solved, double *residuals4) {
    if (!s) return -1;
    for (int i = 0; i < s->n(); ++i)
        if (tinympc_get_status(s->shard[i], iter ? iter + s->lo[i] : nullptr, solved ? solved + s->lo[i] : nullptr,
                               residuals4 ? residuals4 + (size_t)s->lo[i] * 4 : nullptr))
            return -1;
    return 0;
}
int tinympc_sharded_get_workspace(tinympc_sharded *s, double *d, double *y, double *g, double *v, double *z) {
    if (!s) return -1;
    const size_t ex = (size_t)s->nx * s->N, eu = (size_t)s->nu * (s->N - 1);
    for (int i = 0; i < s->n(); ++i) {
        const size_t l = (size_t)s->lo[i];
        if (tinympc_get_workspace(s->shard[i], d ? d + l * eu : nullptr, y ? y + l * eu : nullptr, g ? g + l * ex : nullptr,
                                  v ? v + l * ex : nullptr, z ? z + l * eu : nullptr))
            return -1;
    }
    return 0;
}

}  // extern "C"
