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
//
// Three kinds of unit, all through build_unit():
//   jit_kernel_for   at setup: the quad / mfma entry of a SHAPE the library has no on-chip kernel for;
//   jit_trans_for    at a solver's first solve: ONE admm_mfmat_kernel for exactly its CONSTRAINT LAYOUT — cone lists, linear
//                    rows, other cone rows, another horizon (bindings.cpp:414-490 takes them at run time);
//   jit_lean_for     at the launch that needs it: ONE variant of the lean kernel (the headline's) for a cartpole-class shape
//                    without a built-in lean instantiation, and its fp64-state form for precision 2.
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
std::map<std::string, const void *> g_units;   // unit name -> entry (nullptr: tried and failed, do not try again)
std::vector<const KernelEntry *> g_quad, g_mfma;
std::vector<const ConeEntry *> g_trans;

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
                                  "mfma_entry.hip.h", "admm_mfmac.hip.h", "mfmac_entry.hip.h", "admm_mfmat.hip.h", "mfmat_entry.hip.h",
                                  "admm_lean.hip.h", "lean_entry.hip.h"};
    unsigned long long h = 1469598103934665603ull;
    for (const char *c = "flags: -O3 -DTMPC_JIT_UNIT -DTMPC_MFMAT_HANDOVER=2 -amdgpu-mfma-vgpr-form"; *c; ++c) h = (h ^ (unsigned char)*c) * 1099511628211ull;
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

const void *build_unit(const std::string &unit, const std::string &source, int verbose, const char *extra_flag = nullptr) {
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
        // one process per GPU is the deployment model: eight ranks may specialise the same unit at the same moment — source, log
        // and object of a compile carry the pid, and only the finished object is renamed into place (atomic)
        const std::string pid = std::to_string((long)getpid());
        const std::string src = cache + "/" + unit + "." + pid + ".hip", tmp = so + ".tmp." + pid;
        {
            std::ofstream out(src);
            out << source;
        }
        const auto t0 = std::chrono::steady_clock::now();
        // (-DTMPC_MFMAT_HANDOVER=2: the transposed-sets kernel's hand-over stores as one asm statement each — the default form's
        // store order is held by an assembly test that only sees the built-in instantiations, tests/test_mfmat_asm.py)
        std::vector<std::string> cmd = {hipcc, "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-fno-honor-nans", "-DTMPC_JIT_UNIT",
                                        "-DTMPC_MFMAT_HANDOVER=2", "-mllvm", "-amdgpu-mfma-vgpr-form", "-I" + csrc, src, "-o", tmp, "-L" + lib, "-ltinympc_hip",
                                        "-Wl,-rpath," + lib};
        if (extra_flag) cmd.push_back(extra_flag);
        const int rc = run(cmd, cache + "/" + unit + "." + pid + ".log");
        const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (rc != 0 || ::rename(tmp.c_str(), so.c_str()) != 0) {
            ::unlink(tmp.c_str());
            std::fprintf(stderr, "tinympc_hip: specialising %s failed after %.1f s (log: %s/%s.%s.log); the run-time-shape kernels take this solver\n",
                         unit.c_str(), secs, cache.c_str(), unit.c_str(), pid.c_str());
            return nullptr;
        }
        (void)::rename(src.c_str(), (cache + "/" + unit + ".hip").c_str());                       // (kept for reference)
        (void)::rename((cache + "/" + unit + "." + pid + ".log").c_str(), (cache + "/" + unit + ".log").c_str());
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
    return fn ? fn() : nullptr;
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
    if (it != g_units.end()) return static_cast<const KernelEntry *>(it->second);
    const KernelEntry *e = static_cast<const KernelEntry *>(build_unit(name.str(), src.str(), verbose));
    g_units[name.str()] = e;
    if (e) (mfma ? g_mfma : g_quad).push_back(e);
    return e;
}

const ConeEntry *jit_trans_find(const Solver &sv) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (const ConeEntry *e : g_trans)
        if (e->nx == sv.nx && e->nu == sv.nu && e->N == sv.N && e->supports(sv)) return e;
    return nullptr;
}

// The transposed-sets matrix-core kernel (admm_mfmat.hip.h) for EXACTLY this solver's constraint layout: the affine term, up
// to two disjoint cones per side (ascending rows), linear-inequality rows, its reference mode and bound kind — ONE kernel,
// compiled / loaded once per process and layout.  The reference takes cone lists and row blocks at run time
// (bindings.cpp:414-490); the built-in entries compile one cone per side and no rows.  nullptr: not a layout this kernel
// takes (the stream / generic kernels do), or nothing could be compiled.
const ConeEntry *jit_trans_for(const Solver &sv, int verbose) {
    if (std::getenv("TINYMPC_HIP_NO_JIT")) return nullptr;
    const int nx = sv.nx, nu = sv.nu, N = sv.N;
    if (nx < 4 || nx > 8 || nu < 1 || nu > 4 || N < 3 || N > 64) return nullptr;
    const int ncx = sv.st.en_state_soc ? sv.ncx : 0, ncu = sv.st.en_input_soc ? sv.ncu : 0;
    const int mlx = sv.st.en_state_linear ? sv.mlx : 0, mlu = sv.st.en_input_linear ? sv.mlu : 0;
    if (ncx > 2 || ncu > 2 || mlx > LIN_MAX_ROWS || mlu > LIN_MAX_ROWS) return nullptr;
    if (mlx * (nx + 2) + mlu * (nu + 2) > 48) return nullptr;   // (the rows are scalar registers of the sets phase)
    int c[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};                   // per side: first row, dimension of cone 1, of cone 2
    for (int side = 0; side < 2; ++side) {
        const int n = side ? ncu : ncx, rows = side ? nu : nx;
        const int *A = side ? sv.Acu : sv.Acx, *q = side ? sv.qcu : sv.qcx;
        int behind = 0;
        for (int i = 0; i < n; ++i) {
            if (q[i] < 2 || A[i] < behind || A[i] + q[i] > rows) return nullptr;   // ascending, disjoint, inside the side
            c[side][2 * i] = A[i], c[side][2 * i + 1] = q[i];
            behind = A[i] + q[i];
        }
    }
    // registers of the sets layout (TransShape::state_regs; the last knot group may live in LDS): the 512-entry file must hold them
    const int NG = (N + 3) / 4, qx = c[0][1] + c[0][3], qu = c[1][1] + c[1][3];
    const int gr = (nx + qx + nu + qu) + (nx + (qx ? nx : 0) + nu + (qu ? nu : 0)) + (mlx ? 2 * nx : 0) + (mlu ? 2 * nu : 0);
    int regs = NG * gr, in_lds = 0;                                // (TransShape::spill_groups: up to three groups live in LDS)
    if (NG >= 2 && regs + 70 > 450)
        for (in_lds = 1; in_lds < 3 && in_lds < NG - 1 && (NG - in_lds) * gr + 70 > 470;) ++in_lds;
    regs -= in_lds * gr;
    if (regs + 70 > 470) return nullptr;
    const bool bv = sv.bounds_vary_by_knot();
    const int refs = sv.refs_device_owned ? sv.ref_mode : (sv.xref_kind > sv.uref_kind ? sv.xref_kind : sv.uref_kind);   // (Solver::upload_refs)
    const size_t cells = (size_t)16 * (nx + nu) * N * sizeof(float);
    if ((refs == REF_PER_INSTANCE ? 2 : 1) * cells + (bv ? (size_t)2 * (nx + nu) * N * 4 : 0) + (size_t)256 * in_lds * (gr + 6) > 150 * 1024) return nullptr;
    std::ostringstream name, label, src;
    name << "mfmat_" << nx << "_" << nu << "_" << N << "_r" << refs << (bv ? "_bv" : "") << "_cx" << c[0][0] << "_" << c[0][1] << "_" << c[0][2] << "_"
         << c[0][3] << "_cu" << c[1][0] << "_" << c[1][1] << "_" << c[1][2] << "_" << c[1][3] << "_l" << mlx << "_" << mlu;
    label << "mfmat<" << nx << "," << nu << "," << N << ">";
    if (qx) label << " cx" << c[0][0] << ":" << c[0][1];
    if (c[0][3]) label << "+" << c[0][2] << ":" << c[0][3];
    if (qu) label << " cu" << c[1][0] << ":" << c[1][1];
    if (c[1][3]) label << "+" << c[1][2] << ":" << c[1][3];
    if (mlx || mlu) label << " lin" << mlx << "," << mlu;
    src << "// specialised at setup by jit.cpp\n#include \"mfmat_entry.hip.h\"\nTMPC_DEFINE_MFMAT_JIT_ENTRY(\"" << label.str() << "\", " << nx << ", "
        << nu << ", " << N << ", " << refs << ", " << c[0][0] << ", " << c[0][1] << ", " << c[1][0] << ", " << c[1][1] << ", " << (bv ? "true" : "false")
        << ", " << c[0][2] << ", " << c[0][3] << ", " << c[1][2] << ", " << c[1][3] << ", " << mlx << ", " << mlu << ")\n";
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_units.find(name.str());
    if (it != g_units.end()) return static_cast<const ConeEntry *>(it->second);
    const ConeEntry *e = static_cast<const ConeEntry *>(build_unit(name.str(), src.str(), verbose));
    g_units[name.str()] = e;
    if (e) g_trans.push_back(e);
    return e;
}

// One variant of the lean kernel (admm_lean.hip.h, the headline's) for a shape without a built-in lean instantiation, compiled at
// the first launch that needs it.  variant bits: LV_LIVE tolerance-terminated, LV_UBK the input bounds do not depend on the
// knot, LV_ONE the 512-register form, LV_XB a finite state bound, LV_SHARED shared references, LV_F64 slack / dual state in
// fp64 (precision 2: the reference's arithmetic end to end at this kernel's speed).  The kernel holds one lane's
// whole solve in registers and its coefficients in scalar registers: 32 coefficient doubles and ~490 registers at most —
// cartpole-class systems ((4,1) to N = 36, (3,2), (2,x)); anything else: nullptr, the quad / stream kernels as before.
const LeanEntry *jit_lean_for(int nx, int nu, int N, int variant, int verbose) {
    if (std::getenv("TINYMPC_HIP_NO_JIT")) return nullptr;
    if (nx < 1 || nu < 1 || N < 3 || lean_layout(nx, nu).padded > 32) return nullptr;
    const bool one = (variant & LV_ONE) != 0, f64 = (variant & LV_F64) != 0, xb = (variant & LV_XB) != 0;
    if (f64 && !one) return nullptr;
    const int regs = f64 ? (xb ? 4 : 2) * N * nx + 6 * N * nu + 50 : 2 * N * nx + (one ? 4 : 3) * N * nu + 50;
    if (regs > (f64 ? 450 : (one ? 490 : 250))) return nullptr;   // (fp64 state with a state bound at N = 20: 490 values, 463 of them spilled)
    std::ostringstream name, src;
    name << "lean_" << nx << "_" << nu << "_" << N << "_v" << variant;
    auto tf = [&](int bit) { return (variant & bit) ? "true" : "false"; };
    src << "// specialised at the first solve by jit.cpp\n#include \"lean_entry.hip.h\"\nTMPC_DEFINE_LEAN_JIT_ENTRY(\"lean<" << nx << "," << nu << "," << N
        << (f64 ? ";f64" : "") << ">\", " << nx << ", " << nu << ", " << N << ", " << tf(LV_LIVE) << ", " << tf(LV_UBK) << ", " << tf(LV_ONE) << ", " << tf(LV_XB)
        << ", " << ((variant & LV_SHARED) ? "tmpc::REF_SHARED" : "tmpc::REF_ZERO") << ", " << (f64 ? "double" : "float") << ")\n";
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_units.find(name.str());
    if (it != g_units.end()) return static_cast<const LeanEntry *>(it->second);
    // (-fno-slp-vectorize: the vectoriser's <2 x double> tuples are what makes this kernel spill, csrc/Makefile)
    const LeanEntry *e = static_cast<const LeanEntry *>(build_unit(name.str(), src.str(), verbose, "-fno-slp-vectorize"));
    g_units[name.str()] = e;
    return e;
}

}  // namespace tmpc
