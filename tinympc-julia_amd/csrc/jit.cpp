// Specialisation at setup: the on-chip kernel of a shape the library was not built with.
//
// The reference accepts any (nx, nu, N) at run time (tiny_setup, tiny_api.cpp:21-71).  The fast kernels here are templates
// on the shape — trajectories in registers, horizon loops unrolled — and the library carries a fixed set of
// instantiations; until round 3 every other shape fell to the HBM-streaming kernel (2.9-6 x slower) and the answer to
// "which shapes?" was another translation unit in a 50 MB library.  Instead: when a solver is created for a shape without
// an on-chip kernel, ONE unit — the instantiation macro of the family that suits the shape, exactly what an inst_*.hip /
// minst_*.hip file of the build contains — is compiled by hipcc (a child process), linked against this library, cached,
// loaded with dlopen, and its entry joins the tables find_quad_kernel / find_mfma_kernel search.  The unit is the
// library's own source (csrc/*.h next to lib/), so pack builders, launchers and kernels are the same code as a built-in
// instantiation's.  Cache: $TINYMPC_HIP_CACHE or ~/.cache/tinympc_hip/<hash of the kernel headers>/<unit>.so — a changed
// header is a new directory.  No compiler, no sources, a failed compile (a shape whose state does not fit the chip), or
// TINYMPC_HIP_NO_JIT=1: the solver runs on the stream / generic kernels as before.  Adaptive-rho and fp32-recurrence
// variants are not part of such a unit (-DTMPC_JIT_UNIT: a third of the compile time); the routes know (KernelEntry::jit).
#include <dlfcn.h>
#include <fcntl.h>
#include <spawn.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <map>
#include <mutex>
#include <sstream>
#include <string>
#include <vector>

#include "solver.h"

extern char **environ;

namespace tmpc {
namespace {

std::mutex g_mu;
std::map<std::string, const KernelEntry *> g_units;   // unit name -> entry (nullptr: tried and failed, do not try again)
std::vector<const KernelEntry *> g_quad, g_mfma;

bool file_exists(const std::string &p) {
    struct stat st;
    return ::stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode);
}
void mkdirs(const std::string &p) {
    for (size_t i = 1; i <= p.size(); ++i)
        if (i == p.size() || p[i] == '/') ::mkdir(p.substr(0, i).c_str(), 0755);
}
std::string lib_dir() {   // .../tinympc-julia_amd/lib (where this library was loaded from)
    Dl_info info;
    if (!dladdr((const void *)&jit_kernel_for, &info) || !info.dli_fname) return "";
    std::string p = info.dli_fname;
    const size_t k = p.rfind('/');
    return k == std::string::npos ? "." : p.substr(0, k);
}
// FNV-1a over the kernel headers: the cache key of everything a unit is compiled from
std::string source_hash(const std::string &csrc) {
    static const char *files[] = {"admm_params.h", "solver.h", "host_setup.h", "admm_quad.hip.h", "quad_entry.hip.h", "admm_mfma.hip.h",
                                  "mfma_entry.hip.h"};
    unsigned long long h = 1469598103934665603ull;
    for (const char *f : files) {
        std::ifstream in(csrc + "/" + f, std::ios::binary);
        if (!in) return "";
        char buf[4096];
        while (in.read(buf, sizeof(buf)) || in.gcount() > 0)
            for (std::streamsize i = 0; i < in.gcount(); ++i) h = (h ^ (unsigned char)buf[i]) * 1099511628211ull;
    }
    char out[32];
    std::snprintf(out, sizeof(out), "%016llx", h);
    return out;
}
int run(const std::vector<std::string> &argv, const std::string &log) {
    std::vector<char *> av;
    for (const std::string &a : argv) av.push_back(const_cast<char *>(a.c_str()));
    av.push_back(nullptr);
    posix_spawn_file_actions_t fa;
    posix_spawn_file_actions_init(&fa);
    posix_spawn_file_actions_addopen(&fa, 1, log.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
    posix_spawn_file_actions_adddup2(&fa, 1, 2);
    pid_t pid = 0;
    const int rc = posix_spawn(&pid, av[0], &fa, nullptr, av.data(), environ);   // a child process; this one carries on
    posix_spawn_file_actions_destroy(&fa);
    if (rc != 0) return -1;
    int status = 0;
    while (waitpid(pid, &status, 0) < 0) {}
    return WIFEXITED(status) ? WEXITSTATUS(status) : -1;
}

const KernelEntry *build_unit(const std::string &unit, const std::string &source, int verbose) {
    const std::string lib = lib_dir();
    if (lib.empty()) return nullptr;
    const std::string csrc = lib + "/../csrc";
    const char *hipcc_env = std::getenv("TINYMPC_HIP_HIPCC");
    const std::string hipcc = hipcc_env ? hipcc_env : "/opt/rocm/bin/hipcc";
    if (!file_exists(csrc + "/admm_quad.hip.h") || !file_exists(hipcc)) return nullptr;   // no sources / no compiler here
    const std::string hash = source_hash(csrc);
    if (hash.empty()) return nullptr;
    std::string cache;
    if (const char *c = std::getenv("TINYMPC_HIP_CACHE")) cache = c;
    else if (const char *h = std::getenv("HOME")) cache = std::string(h) + "/.cache/tinympc_hip";
    else cache = "/tmp/tinympc_hip_cache";
    cache += "/" + hash;
    mkdirs(cache);
    const std::string so = cache + "/" + unit + ".so";
    if (!file_exists(so)) {
        const std::string src = cache + "/" + unit + ".hip", tmp = so + ".tmp." + std::to_string((long)getpid());
        {
            std::ofstream out(src);
            out << source;
        }
        const auto t0 = std::chrono::steady_clock::now();
        const int rc = run({hipcc, "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-fno-honor-nans", "-DTMPC_JIT_UNIT",
                            "-mllvm", "-amdgpu-mfma-vgpr-form", "-I" + csrc, src, "-o", tmp, "-L" + lib, "-ltinympc_hip", "-Wl,-rpath," + lib},
                           cache + "/" + unit + ".log");
        const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (rc != 0 || ::rename(tmp.c_str(), so.c_str()) != 0) {
            ::unlink(tmp.c_str());
            std::fprintf(stderr, "tinympc_hip: specialising %s failed after %.1f s (log: %s/%s.log); the run-time-shape kernels take this solver\n",
                         unit.c_str(), secs, cache.c_str(), unit.c_str());
            return nullptr;
        }
        std::fprintf(stderr, "tinympc_hip: specialised %s in %.1f s (one-off; cached at %s)\n", unit.c_str(), secs, so.c_str());
    } else if (verbose) {
        std::fprintf(stderr, "tinympc_hip: %s from the cache (%s)\n", unit.c_str(), so.c_str());
    }
    void *h = dlopen(so.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!h) {
        std::fprintf(stderr, "tinympc_hip: cannot load %s: %s\n", so.c_str(), dlerror());
        return nullptr;
    }
    using Fn = const void *(*)();
    Fn fn = (Fn)dlsym(h, "tmpc_jit_entry");
    return fn ? static_cast<const KernelEntry *>(fn()) : nullptr;
}

}  // namespace

const KernelEntry *jit_find(bool mfma, int nx, int nu, int N, int group) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (const KernelEntry *e : (mfma ? g_mfma : g_quad))
        if (e->nx == nx && e->nu == nu && e->N == N && (group < 0 || e->G == group)) return e;
    return nullptr;
}

// The family that suits a shape without a built-in on-chip kernel, compiled / loaded once per process:
//   seven or more state rows (nx <= 12, nu <= 4): the matrix-core kernel, else four lanes per instance.  Returns nullptr when
//   nothing could be specialised.
const KernelEntry *jit_kernel_for(int nx, int nu, int N, int verbose) {
    if (std::getenv("TINYMPC_HIP_NO_JIT")) return nullptr;
    if (nx > 12 || nu > 4 || N < 2 || N > 64) return nullptr;   // (beyond: the state does not fit a lane group / tile either way)
    // matrix-core time per knot does not depend on the shape (one 16-row tile per 16 instances: ~128 us per knot and 65 536
    // instances, quadrotor N = 10 .. 30), four-lanes-per-instance time grows with nx^2 (43 us per knot for (4,1), 134-225 for
    // (8,2): scripts/jit_time.py) — from seven state rows on the matrix cores win
    const bool mfma = nx >= 7;
    std::ostringstream name, src;
    if (mfma) {
        name << "mfma_" << nx << "_" << nu << "_" << N;
        src << "// specialised at setup by jit.cpp\n#include \"mfma_entry.hip.h\"\nTMPC_DEFINE_MFMA_JIT_ENTRY(" << nx << ", " << nu << ", " << N << ")\n";
    } else {
        name << "quad_" << nx << "_" << nu << "_" << N << "_g4";
        src << "// specialised at setup by jit.cpp\n#include \"quad_entry.hip.h\"\nTMPC_DEFINE_QUAD_JIT_ENTRY(" << nx << ", " << nu << ", " << N << ", 4)\n";
    }
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_units.find(name.str());
    if (it != g_units.end()) return it->second;
    const KernelEntry *e = build_unit(name.str(), src.str(), verbose);
    g_units[name.str()] = e;
    if (e) (mfma ? g_mfma : g_quad).push_back(e);
    return e;
}

}  // namespace tmpc
