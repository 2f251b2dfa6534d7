// Kernel instantiations, launchers and host-side pack builders.
#include <cmath>
#include <limits>

#include "admm_generic.hip.h"
#include "admm_quad.hip.h"
#include "solver.h"

namespace tmpc {
namespace {

constexpr float kInf = std::numeric_limits<float>::infinity();

// ---- quad kernel packs (layout: QuadShape in admm_quad.hip.h) ----
template <class S>
void build_quad_coef(const Solver &sv, std::vector<float> &out) {
    constexpr int NX = S::NX, NU = S::NU, RX = S::RX, RU = S::RU, NXP = S::NXP, NUP = S::NUP;
    out.assign((size_t)4 * S::CP, 0.f);
    const Cache &c = sv.cache;
    for (int q = 0; q < 4; ++q) {
        float *p = out.data() + (size_t)q * S::CP;
        for (int m = 0; m < RX; ++m) {
            const int row = q * RX + m;
            if (row >= NX) continue;
            for (int j = 0; j < NX; ++j) {
                p[S::O_A + m * NXP + j] = (float)sv.A(row, j);
                p[S::O_AT + m * NXP + j] = (float)c.AmBKt(row, j);
                p[S::O_PT + m * NXP + j] = (float)c.Pinf(j, row);  // (Pinf^T)[row][j]
            }
            for (int a = 0; a < NU; ++a) {
                p[S::O_B + m * NUP + a] = (float)sv.B(row, a);
                p[S::O_KT + m * NUP + a] = (float)c.Kinf(a, row);  // (Kinf^T)[row][a]
            }
            p[S::O_QD + m] = (float)c.Qd[row];
        }
        for (int m = 0; m < RU; ++m) {
            const int row = q * RU + m;
            if (row >= NU) continue;
            for (int j = 0; j < NX; ++j) {
                p[S::O_K + m * NXP + j] = (float)c.Kinf(row, j);
                p[S::O_BT + m * NXP + j] = (float)sv.B(j, row);  // (B^T)[row][j]
            }
            for (int a = 0; a < NU; ++a) p[S::O_QI + m * NUP + a] = (float)c.Quu_inv(row, a);
            p[S::O_RD + m] = (float)c.Rd[row];
        }
    }
}

template <class S>
void build_quad_bounds(const Solver &sv, std::vector<float> &out) {
    constexpr int NX = S::NX, NU = S::NU, N = S::N, RX = S::RX, RU = S::RU, BW = S::BW;
    out.assign((size_t)S::BOUNDS_LEN, 0.f);
    for (int k = 0; k < N; ++k)
        for (int q = 0; q < 4; ++q) {
            float *p = out.data() + ((size_t)k * 4 + q) * BW;
            for (int m = 0; m < RX; ++m) {
                const int row = q * RX + m;
                const bool on = sv.st.en_state_bound && row < NX;
                p[m] = on ? (float)sv.x_min[row + (size_t)k * NX] : -kInf;
                p[RX + m] = on ? (float)sv.x_max[row + (size_t)k * NX] : kInf;
            }
            for (int m = 0; m < RU; ++m) {
                const int row = q * RU + m;
                const bool on = sv.st.en_input_bound && row < NU && k < N - 1;
                p[2 * RX + m] = on ? (float)sv.u_min[row + (size_t)k * NU] : -kInf;
                p[2 * RX + RU + m] = on ? (float)sv.u_max[row + (size_t)k * NU] : kInf;
            }
        }
}

template <class S>
hipError_t launch_quad(const AdmmParams &P, hipStream_t stream) {
    const int grid = (P.batch + S::INST_PER_BLOCK - 1) / S::INST_PER_BLOCK;
    switch (P.ref_mode) {
        case REF_ZERO:
            hipLaunchKernelGGL((admm_quad_kernel<S, REF_ZERO>), dim3(grid), dim3(S::THREADS), 0, stream, P);
            break;
        case REF_SHARED:
            hipLaunchKernelGGL((admm_quad_kernel<S, REF_SHARED>), dim3(grid), dim3(S::THREADS), 0, stream, P);
            break;
        default:
            hipLaunchKernelGGL((admm_quad_kernel<S, REF_PER_INSTANCE>), dim3(grid), dim3(S::THREADS), 0, stream, P);
            break;
    }
    return hipGetLastError();
}

#define TMPC_QUAD_ENTRY(NX, NU, NN)                                                              \
    {                                                                                            \
        NX, NU, NN, "quad<" #NX "," #NU "," #NN ">", 4 * QuadShape<NX, NU, NN>::CP,              \
            QuadShape<NX, NU, NN>::BOUNDS_LEN, &build_quad_coef<QuadShape<NX, NU, NN>>,          \
            &build_quad_bounds<QuadShape<NX, NU, NN>>, &launch_quad<QuadShape<NX, NU, NN>>       \
    }

// Shapes with a specialised kernel: the BASELINE.json configs plus the shapes the
// reference's own tests/examples use (tests/test_basic.jl N=10, test_settings.jl N=2,
// examples/*: cartpole N=20, quadrotor N=20, rocket N=10).
const KernelEntry kTable[] = {
    TMPC_QUAD_ENTRY(4, 1, 20),  TMPC_QUAD_ENTRY(4, 1, 10), TMPC_QUAD_ENTRY(4, 1, 2),
    TMPC_QUAD_ENTRY(12, 4, 30), TMPC_QUAD_ENTRY(12, 4, 20), TMPC_QUAD_ENTRY(6, 3, 10),
};

}  // namespace

const KernelEntry *find_quad_kernel(int nx, int nu, int N) {
    for (const KernelEntry &e : kTable)
        if (e.nx == nx && e.nu == nu && e.N == N) return &e;
    return nullptr;
}

hipError_t launch_generic(const AdmmParams &P, hipStream_t stream) {
    const int threads = 256;
    const int grid = (P.batch + threads - 1) / threads;
    hipLaunchKernelGGL(admm_generic_kernel, dim3(grid), dim3(threads), 0, stream, P);
    return hipGetLastError();
}

void build_generic_coef(const Solver &sv, std::vector<float> &out) {
    const int nx = sv.nx, nu = sv.nu;
    const GenericPack pk(nx, nu);
    out.assign(pk.len, 0.f);
    const Cache &c = sv.cache;
    for (int j = 0; j < nx; ++j)
        for (int i = 0; i < nx; ++i) {
            out[pk.oA + i + j * nx] = (float)sv.A(i, j);
            out[pk.oP + i + j * nx] = (float)c.Pinf(i, j);
            out[pk.oAt + i + j * nx] = (float)c.AmBKt(i, j);
        }
    for (int a = 0; a < nu; ++a)
        for (int i = 0; i < nx; ++i) {
            out[pk.oB + i + a * nx] = (float)sv.B(i, a);
            out[pk.oK + a + i * nu] = (float)c.Kinf(a, i);
        }
    for (int a = 0; a < nu; ++a)
        for (int b2 = 0; b2 < nu; ++b2) out[pk.oQi + a + b2 * nu] = (float)c.Quu_inv(a, b2);
    for (int i = 0; i < nx; ++i) out[pk.oQd + i] = (float)c.Qd[i];
    for (int a = 0; a < nu; ++a) out[pk.oRd + a] = (float)c.Rd[a];
}

void build_generic_bounds(const Solver &sv, std::vector<float> &out) {
    const int EX = sv.ex(), EU = sv.eu();
    out.assign((size_t)2 * EX + 2 * EU, 0.f);
    for (int e = 0; e < EX; ++e) {
        out[e] = sv.st.en_state_bound ? (float)sv.x_min[e] : -kInf;
        out[EX + e] = sv.st.en_state_bound ? (float)sv.x_max[e] : kInf;
    }
    for (int e = 0; e < EU; ++e) {
        out[2 * EX + e] = sv.st.en_input_bound ? (float)sv.u_min[e] : -kInf;
        out[2 * EX + EU + e] = sv.st.en_input_bound ? (float)sv.u_max[e] : kInf;
    }
}

}  // namespace tmpc
