// Kernel table, generic-kernel launcher and its host-side pack builders.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <limits>

#include "admm_generic.hip.h"
#include "solver.h"

namespace tmpc {

// Shapes with a specialised kernel: the BASELINE.json configs plus the shapes the
// reference's own tests/examples use (tests/test_basic.jl N=10, test_settings.jl N=2,
// examples/*: cartpole N=20, quadrotor N=20, rocket N=10).
// Table order = preference: the first entry matching (nx, nu, N) is the shape's default variant.
#define TMPC_ENTRY(NX, NU, NN, GG) const KernelEntry *quad_entry_##NX##_##NU##_##NN##_g##GG();
TMPC_ENTRY(4, 1, 20, 4) TMPC_ENTRY(4, 1, 20, 2) TMPC_ENTRY(4, 1, 20, 1)
TMPC_ENTRY(4, 1, 10, 4) TMPC_ENTRY(4, 1, 10, 2) TMPC_ENTRY(4, 1, 10, 1) TMPC_ENTRY(4, 1, 2, 4)
TMPC_ENTRY(4, 1, 5, 4) TMPC_ENTRY(4, 1, 5, 1) TMPC_ENTRY(4, 1, 15, 4) TMPC_ENTRY(4, 1, 15, 1) TMPC_ENTRY(4, 1, 30, 4) TMPC_ENTRY(4, 1, 30, 1)
TMPC_ENTRY(12, 4, 30, 4) TMPC_ENTRY(12, 4, 20, 4)
TMPC_ENTRY(6, 3, 10, 4) TMPC_ENTRY(6, 3, 10, 2) TMPC_ENTRY(6, 3, 50, 4)
#undef TMPC_ENTRY

const KernelEntry *find_quad_kernel(int nx, int nu, int N, int group) {
    static const KernelEntry *const table[] = {
        quad_entry_4_1_20_g4(),  quad_entry_4_1_20_g2(),  quad_entry_4_1_20_g1(), quad_entry_4_1_10_g4(),
        quad_entry_4_1_10_g2(),  quad_entry_4_1_10_g1(),  quad_entry_4_1_2_g4(),  quad_entry_12_4_30_g4(),
        quad_entry_12_4_20_g4(), quad_entry_6_3_10_g4(),  quad_entry_6_3_10_g2(), quad_entry_6_3_50_g4(),
        // further cartpole horizons (tests/test_codegen.jl:15 uses N = 5): without them these fall to the HBM-streaming kernel
        quad_entry_4_1_5_g4(),   quad_entry_4_1_5_g1(),   quad_entry_4_1_15_g4(), quad_entry_4_1_15_g1(),
        quad_entry_4_1_30_g4(),  quad_entry_4_1_30_g1(),
    };
    for (const KernelEntry *e : table)
        if (e->nx == nx && e->nu == nu && e->N == N && (group < 0 || e->G == group)) return e;
    return jit_find(false, nx, nu, N, group);   // ... or one specialised at setup (jit.cpp)
}

const KernelEntry *mfma_entry_12_4_30();
const KernelEntry *mfma_entry_12_4_25();
const KernelEntry *mfma_entry_12_4_20();
const KernelEntry *mfma_entry_12_4_15();
const KernelEntry *mfma_entry_12_4_10();

// matrix-core kernels (admm_mfma.hip.h): one-shot solves of the shapes instantiated.  The kernel is fully unrolled with its
// state in registers: rocket N=50 (347 state floats per lane with finite state bounds) spills ~1 000 registers and runs
// 18 ms against the quad kernel's 5.5, and rocket N=10 fills only 9 of a tile's 16 rows (1.13 ms against 0.74):
// neither is instantiated.  Quadrotor ([A; -Kinf] is a full 16 x 12): N=30 4.13 against 11.6 ms, N=20 2.7 against 5.9
const KernelEntry *find_mfma_kernel(int nx, int nu, int N) {
    static const KernelEntry *const table[] = {mfma_entry_12_4_30(), mfma_entry_12_4_25(), mfma_entry_12_4_20(), mfma_entry_12_4_15(),
                                               mfma_entry_12_4_10()};
    for (const KernelEntry *e : table)
        if (e->nx == nx && e->nu == nu && e->N == N) return e;
    return jit_find(true, nx, nu, N, -1);   // ... or one specialised at setup (jit.cpp)
}

const LeanEntry *lean_entry_4_1_20();
const LeanEntry *lean_entry_4_1_15();
const LeanEntry *lean_entry_4_1_10();
const LeanEntry *lean_entry_4_1_5();

const LeanEntry *find_lean_kernel(int nx, int nu, int N) {
    static const LeanEntry *const table[] = {lean_entry_4_1_20(), lean_entry_4_1_15(), lean_entry_4_1_10(), lean_entry_4_1_5()};
    for (const LeanEntry *e : table)
        if (e->nx == nx && e->nu == nu && e->N == N) return e;
    return nullptr;
}

bool build_lean_pack(const Solver &sv, std::vector<double> &out) {
    const int nx = sv.nx, nu = sv.nu;
    const LeanLayout L = lean_layout(nx, nu);
    out.assign((size_t)L.total, 0.0);
    const Cache &c = sv.cache;
    // the kernel reads ONE matrix as A - B Kinf (rollout) and, transposed, as AmBKt (gradient recursion): only valid while
    // the cache's AmBKt is that transpose (tiny_api.cpp:170; set_cache_terms can install anything)
    double scale = 0.0, diff = 0.0;
    for (int i = 0; i < nx; ++i)
        for (int j = 0; j < nx; ++j) {
            double m = sv.A(i, j);
            for (int a = 0; a < nu; ++a) m -= sv.B(i, a) * c.Kinf(a, j);
            out[L.oM + i * nx + j] = m;
            scale = std::max(scale, std::fabs(m));
            diff = std::max(diff, std::fabs(m - c.AmBKt(j, i)));
        }
    if (!(diff <= 1e-12 * std::max(scale, 1.0))) return false;
    for (int i = 0; i < nx; ++i)      // (bit for bit the cache's own values where the two agree to rounding)
        for (int j = 0; j < nx; ++j) out[L.oM + i * nx + j] = c.AmBKt(j, i);
    for (int a = 0; a < nu; ++a)
        for (int j = 0; j < nx; ++j) out[L.oK + a * nx + j] = c.Kinf(a, j);
    for (int i = 0; i < nx; ++i)
        for (int a = 0; a < nu; ++a) out[L.oB + i * nu + a] = sv.B(i, a);
    for (int a = 0; a < nu; ++a)
        for (int b2 = 0; b2 < nu; ++b2) out[L.oC + a * nu + b2] = -c.rho * c.Quu_inv(a, b2);
    for (int i = 0; i < nx; ++i)
        for (int j = 0; j < nx; ++j) out[L.oP + i * nx + j] = c.Pinf(i, j);
    return true;
}

// Lanes per instance for a batch size.  Fewer lanes per instance means fewer cross-lane moves and no redundant
// work, but also fewer wavefronts.  A launch with at most one wavefront per SIMD takes about the same time
// whatever the batch (the instances' serial chains run side by side), so one lane per instance wins as soon as four
// lanes would need more than 1 024 wavefronts.  Measured on cartpole N=20, MI355X (scripts/group_sweep.sh), kernel ms
// for 1 / 2 / 4 lanes: batch 16 384: 0.326 / 0.331 / 0.272; 24 576: 0.331 / 0.348 / 0.383; 65 536: 0.370 / 0.512 / 0.748.
const KernelEntry *select_quad_kernel(int nx, int nu, int N, int batch) {
    const int pref[2][3] = {{1, 2, 4}, {4, 2, 1}};
    const int *order = batch >= 20480 ? pref[0] : pref[1];
    for (int i = 0; i < 3; ++i)
        if (const KernelEntry *e = find_quad_kernel(nx, nu, N, order[i])) return e;
    return nullptr;
}

const ConeEntry *mfmac_entry_6_3();
const ConeEntry *mfmar_entry_6_3_50();
const ConeEntry *mfmar_entry_6_3_10();
const ConeEntry *mfmar_entry_6_3_20();
const ConeEntry *mfmar_entry_6_3_30();

// matrix-core kernels for one-shot solves with the affine term / cones, instantiated for the rocket's shape: the
// register-resident one where the horizon is compiled in (admm_mfmar.hip.h), else the LDS-resident one with a run-time
// horizon (admm_mfmac.hip.h)
const ConeEntry *find_cone_kernel(int nx, int nu, int N) {
    static const ConeEntry *const table[] = {mfmar_entry_6_3_50(), mfmar_entry_6_3_30(), mfmar_entry_6_3_20(), mfmar_entry_6_3_10(), mfmac_entry_6_3()};
    for (const ConeEntry *e : table)
        if (e->nx == nx && e->nu == nu && e->N == N) return e;
    return nullptr;
}

const ConeEntry *mfmat_entry_6_3_50();
const ConeEntry *mfmat_entry_6_3_30();
const ConeEntry *mfmat_entry_6_3_20();
const ConeEntry *mfmat_entry_6_3_10();

// the transposed-sets matrix-core kernel (admm_mfmat.hip.h): every kind of solve of the shapes instantiated — one-shot,
// warm-started, workspace-keeping, chunked, closed loop — with box bounds, the affine term and one cone per side
const ConeEntry *find_trans_kernel(int nx, int nu, int N) {
    static const ConeEntry *const table[] = {mfmat_entry_6_3_50(), mfmat_entry_6_3_30(), mfmat_entry_6_3_20(), mfmat_entry_6_3_10()};
    for (const ConeEntry *e : table)
        if (e->nx == nx && e->nu == nu && e->N == N) return e;
    return nullptr;
}

int device_cu_count() {
    static std::atomic<int> cus[64];   // (a sharded handle's worker threads may ask for several devices at once)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    int c = cus[dev & 63].load(std::memory_order_relaxed);
    if (c <= 0) {
        hipDeviceProp_t prop;
        c = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        cus[dev & 63].store(c, std::memory_order_relaxed);
    }
    return c;
}

const StreamEntry *stream4_entry_2_1();
const StreamEntry *stream4_entry_2_2();
const StreamEntry *stream4_entry_3_1();
const StreamEntry *stream4_entry_3_2();
const StreamEntry *stream4_entry_3_3();
const StreamEntry *stream4_entry_4_1();
const StreamEntry *stream4_entry_4_2();
const StreamEntry *stream4_entry_4_3();
const StreamEntry *stream4_entry_4_4();
const StreamEntry *stream4_entry_6_1();
const StreamEntry *stream4_entry_6_2();
const StreamEntry *stream4_entry_6_3();
const StreamEntry *stream4_entry_6_4();
const StreamEntry *stream4_entry_8_1();
const StreamEntry *stream4_entry_8_2();
const StreamEntry *stream4_entry_8_3();
const StreamEntry *stream4_entry_8_4();
const StreamEntry *stream4_entry_10_1();
const StreamEntry *stream4_entry_10_2();
const StreamEntry *stream4_entry_10_3();
const StreamEntry *stream4_entry_10_4();
const StreamEntry *stream4_entry_12_1();
const StreamEntry *stream4_entry_12_2();
const StreamEntry *stream4_entry_12_3();
const StreamEntry *stream4_entry_12_4();

// The run-time-horizon kernel (admm_streamg.hip.h) is instantiated with 4 lanes per instance: measured on MI355X
// (rocket N=50 with cones, batch 4 096 .. 65 536) 1 and 2 lanes per instance are no faster at any batch size - all
// three are bound by the scratch traffic from ~32 768 instances up - and slower below.
const StreamEntry *find_stream_kernel(int nx, int nu) {
    static const StreamEntry *const table[] = {stream4_entry_2_1(), stream4_entry_2_2(), stream4_entry_3_1(), stream4_entry_3_2(), stream4_entry_3_3(), stream4_entry_4_1(), stream4_entry_4_2(), stream4_entry_4_3(), stream4_entry_4_4(), stream4_entry_6_1(), stream4_entry_6_2(), stream4_entry_6_3(), stream4_entry_6_4(), stream4_entry_8_1(), stream4_entry_8_2(), stream4_entry_8_3(), stream4_entry_8_4(), stream4_entry_10_1(), stream4_entry_10_2(), stream4_entry_10_3(), stream4_entry_10_4(), stream4_entry_12_1(), stream4_entry_12_2(), stream4_entry_12_3(), stream4_entry_12_4()};
    for (const StreamEntry *e : table)
        if (e->nx == nx && e->nu == nu) return e;
    return nullptr;
}

hipError_t launch_generic(const AdmmParams &P, int precision, hipStream_t stream) {
    const int threads = 256;
    const int grid = (P.batch + threads - 1) / threads;
    if (precision == 2)
        hipLaunchKernelGGL((admm_generic_kernel<double, double>), dim3(grid), dim3(threads), 0, stream, P);
    else if (precision == 0)
        hipLaunchKernelGGL((admm_generic_kernel<double, float>), dim3(grid), dim3(threads), 0, stream, P);
    else
        hipLaunchKernelGGL((admm_generic_kernel<float, float>), dim3(grid), dim3(threads), 0, stream, P);
    return hipGetLastError();
}

template <class RT>
static void fill_generic_coef(const Solver &sv, std::vector<unsigned char> &out) {
    const int nx = sv.nx, nu = sv.nu;
    const GenericPack pk(nx, nu);
    out.assign((size_t)pk.len * sizeof(RT), 0);
    auto put = [&](size_t idx, double val) {
        const RT v = (RT)val;
        std::memcpy(out.data() + idx * sizeof(RT), &v, sizeof(RT));
    };
    const Cache &c = sv.cache;
    for (int j = 0; j < nx; ++j)
        for (int i = 0; i < nx; ++i) {
            put(pk.oA + i + j * nx, sv.A(i, j));
            put(pk.oP + i + j * nx, c.Pinf(i, j));
            put(pk.oAt + i + j * nx, c.AmBKt(i, j));
        }
    for (int a = 0; a < nu; ++a)
        for (int i = 0; i < nx; ++i) {
            put(pk.oB + i + a * nx, sv.B(i, a));
            put(pk.oK + a + i * nu, c.Kinf(a, i));
        }
    for (int a = 0; a < nu; ++a)
        for (int b2 = 0; b2 < nu; ++b2) put(pk.oQi + a + b2 * nu, c.Quu_inv(a, b2));
    // affine dynamics (UNPINNED): f, APf = AmBKt Pinf f, BPf = B^T Pinf f, all formed in fp64
    std::vector<double> Pf(nx, 0.0);
    for (int i = 0; i < nx; ++i)
        for (int l = 0; l < nx; ++l) Pf[i] += c.Pinf(i, l) * sv.fdyn[l];
    for (int i = 0; i < nx; ++i) {
        double apf = 0.0;
        for (int j = 0; j < nx; ++j) apf += c.AmBKt(i, j) * Pf[j];
        put(pk.oF + i, sv.fdyn[i]);
        put(pk.oAPf + i, apf);
    }
    for (int a = 0; a < nu; ++a) {
        double bpf = 0.0;
        for (int j = 0; j < nx; ++j) bpf += sv.B(j, a) * Pf[j];
        put(pk.oBPf + a, bpf);
    }
}

void build_generic_coef(const Solver &sv, std::vector<unsigned char> &out) {
    if (sv.precision != 1)
        fill_generic_coef<double>(sv, out);
    else
        fill_generic_coef<float>(sv, out);
}

void build_generic_bounds(const Solver &sv, std::vector<float> &out) {
    constexpr float kInf = std::numeric_limits<float>::infinity();
    const int EX = sv.ex(), EU = sv.eu();
    out.assign((size_t)2 * EX + 2 * EU + sv.nx + sv.nu, 0.f);
    for (int e = 0; e < EX; ++e) {
        out[e] = sv.st.en_state_bound ? (float)sv.x_min[e] : -kInf;
        out[EX + e] = sv.st.en_state_bound ? (float)sv.x_max[e] : kInf;
    }
    for (int e = 0; e < EU; ++e) {
        out[2 * EX + e] = sv.st.en_input_bound ? (float)sv.u_min[e] : -kInf;
        out[2 * EX + EU + e] = sv.st.en_input_bound ? (float)sv.u_max[e] : kInf;
    }
    for (int i = 0; i < sv.nx; ++i) out[2 * EX + 2 * EU + i] = (float)sv.cache.Qd[i];
    for (int a = 0; a < sv.nu; ++a) out[2 * EX + 2 * EU + sv.nx + a] = (float)sv.cache.Rd[a];
}

}  // namespace tmpc
