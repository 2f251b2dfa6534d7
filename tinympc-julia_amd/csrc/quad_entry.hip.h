// Host-side pack builders + launcher for one QuadShape; each shape is instantiated in its
// own translation unit (inst_*.hip) so the unrolled kernels compile in parallel.
#pragma once
#include <cstdlib>
#include <cstring>
#include <limits>

#include "admm_quad.hip.h"
#include "solver.h"

namespace tmpc {

template <class RT>
inline void put_coef(std::vector<unsigned char> &out, size_t idx, double val) {
    const RT v = (RT)val;
    std::memcpy(out.data() + idx * sizeof(RT), &v, sizeof(RT));
}

// layout: QuadShape in admm_quad.hip.h
template <class S, class RT>
void fill_quad_coef(const Solver &sv, std::vector<unsigned char> &out) {
    constexpr int NX = S::NX, NU = S::NU, RX = S::RX, RU = S::RU, NXP = S::NXP, NUP = S::NUP;
    out.assign((size_t)S::G * S::CP * sizeof(RT), 0);
    const Cache &c = sv.cache;
    for (int q = 0; q < S::G; ++q) {
        const size_t o = (size_t)q * S::CP;
        for (int m = 0; m < RX; ++m) {
            const int row = q * RX + m;
            if (row >= NX) continue;
            for (int j = 0; j < NX; ++j) {
                put_coef<RT>(out, o + S::O_A + m * NXP + j, sv.A(row, j));
                put_coef<RT>(out, o + S::O_AT + m * NXP + j, c.AmBKt(row, j));
                put_coef<RT>(out, o + S::O_PT + m * NXP + j, c.Pinf(j, row));  // (Pinf^T)[row][j]
            }
            for (int a = 0; a < NU; ++a) {
                put_coef<RT>(out, o + S::O_B + m * NUP + a, sv.B(row, a));
                put_coef<RT>(out, o + S::O_KT + m * NUP + a, c.Kinf(a, row));  // (Kinf^T)[row][a]
            }
        }
        for (int m = 0; m < RU; ++m) {
            const int row = S::UREP ? m : q * RU + m;  // UREP: every role carries input row 0
            if (row >= NU) continue;
            for (int j = 0; j < NX; ++j) {
                put_coef<RT>(out, o + S::O_K + m * NXP + j, c.Kinf(row, j));
                put_coef<RT>(out, o + S::O_BT + m * NXP + j, sv.B(j, row));  // (B^T)[row][j]
            }
            for (int a = 0; a < NU; ++a) put_coef<RT>(out, o + S::O_QI + m * NUP + a, c.Quu_inv(row, a));
        }
    }
}

template <class S>
void build_quad_coef(const Solver &sv, std::vector<unsigned char> &out) {
    if (sv.precision == 0)
        fill_quad_coef<S, double>(sv, out);
    else
        fill_quad_coef<S, float>(sv, out);
}

template <class S>
void build_quad_bounds(const Solver &sv, std::vector<float> &out) {
    constexpr float kInf = std::numeric_limits<float>::infinity();
    constexpr int NX = S::NX, NU = S::NU, N = S::N, RX = S::RX, RU = S::RU, BW = S::BW;
    out.assign((size_t)S::BOUNDS_LEN, 0.f);
    for (int k = 0; k < N; ++k)
        for (int q = 0; q < S::G; ++q) {
            float *p = out.data() + ((size_t)k * S::G + q) * BW;
            for (int m = 0; m < RX; ++m) {
                const int row = q * RX + m;
                const bool on = sv.st.en_state_bound && row < NX;
                p[m] = on ? (float)sv.x_min[row + (size_t)k * NX] : -kInf;
                p[RX + m] = on ? (float)sv.x_max[row + (size_t)k * NX] : kInf;
            }
            for (int m = 0; m < RU; ++m) {
                const int row = S::UREP ? m : q * RU + m;
                const bool on = sv.st.en_input_bound && row < NU && k < N - 1;
                p[2 * RX + m] = on ? (float)sv.u_min[row + (size_t)k * NU] : -kInf;
                p[2 * RX + RU + m] = on ? (float)sv.u_max[row + (size_t)k * NU] : kInf;
            }
        }
    // diag(Q)+rho, diag(R)+rho per role (tiny_api.cpp:90-91)
    float *dg = out.data() + (size_t)N * S::G * BW;
    for (int q = 0; q < S::G; ++q) {
        for (int m = 0; m < RX; ++m) {
            const int row = q * RX + m;
            dg[q * S::DW + m] = row < NX ? (float)sv.cache.Qd[row] : 0.f;
        }
        for (int m = 0; m < RU; ++m) {
            const int row = S::UREP ? m : q * RU + m;
            dg[q * S::DW + RX + m] = row < NU ? (float)sv.cache.Rd[row] : 0.f;
        }
    }
}

template <class S, class RT, bool XB, bool OS, bool UNI>
hipError_t launch_quad_os(const AdmmParams &P, hipStream_t stream) {
    const int grid = (P.batch + S::INST_PER_BLOCK - 1) / S::INST_PER_BLOCK;
    switch (P.ref_mode) {
        case REF_ZERO:
            hipLaunchKernelGGL((admm_quad_kernel<S, REF_ZERO, RT, XB, OS, UNI>), dim3(grid), dim3(S::THREADS), 0, stream,
                               P);
            break;
        case REF_SHARED:
            hipLaunchKernelGGL((admm_quad_kernel<S, REF_SHARED, RT, XB, OS, UNI>), dim3(grid), dim3(S::THREADS), 0,
                               stream, P);
            break;
        default:
            hipLaunchKernelGGL((admm_quad_kernel<S, REF_PER_INSTANCE, RT, XB, OS, UNI>), dim3(grid), dim3(S::THREADS), 0,
                               stream, P);
            break;
    }
    return hipGetLastError();
}

// One-shot (cold start, nothing saved, no closed loop) solves need less state, and fixed-iteration ones
// no per-lane guard: see admm_quad_kernel.  Measured (MI355X, batch 65 536 / 32 768): a clear win where the
// state does not fit the register file (rocket N=50: 7.8 -> 5.9 ms), neutral to slightly negative for the
// small shapes (cartpole: 0.42 -> 0.45 ms) and for quadrotor (11.6 -> 11.4 / 14.7 ms), so the variants are
// built per shape (QuadShape::LOOPV).
// adaptive rho: the ADP kernels (per-lane guard; one-shot or workspace-carrying)
template <class S, class RT, bool XB, bool OS>
hipError_t launch_quad_adp(const AdmmParams &P, hipStream_t stream) {
    const int grid = (P.batch + S::INST_PER_BLOCK - 1) / S::INST_PER_BLOCK;
    switch (P.ref_mode) {
        case REF_ZERO:
            hipLaunchKernelGGL((admm_quad_kernel<S, REF_ZERO, RT, XB, OS, false, true>), dim3(grid), dim3(S::THREADS), 0,
                               stream, P);
            break;
        case REF_SHARED:
            hipLaunchKernelGGL((admm_quad_kernel<S, REF_SHARED, RT, XB, OS, false, true>), dim3(grid), dim3(S::THREADS), 0,
                               stream, P);
            break;
        default:
            hipLaunchKernelGGL((admm_quad_kernel<S, REF_PER_INSTANCE, RT, XB, OS, false, true>), dim3(grid),
                               dim3(S::THREADS), 0, stream, P);
            break;
    }
    return hipGetLastError();
}

template <class S, class RT, bool XB>
hipError_t launch_quad_rt(const AdmmParams &P, hipStream_t stream) {
#ifdef TMPC_JIT_UNIT
    if (P.adaptive_rho) return hipErrorInvalidValue;   // (no adaptive-rho variants in a unit specialised at setup)
#else
    if (P.adaptive_rho) {
        if constexpr (S::G == 1) {
            // one lane per instance: the correction form — zero or shared references, fp64 recurrences (the solver selects
            // the entry only then)
            if constexpr (sizeof(RT) == 8) {
                if (P.ref_mode == REF_PER_INSTANCE) return hipErrorInvalidValue;
                const int grid = (P.batch + S::INST_PER_BLOCK - 1) / S::INST_PER_BLOCK;
                const bool oneshot = P.cold_start && !P.save_state && P.mpc_steps == 0;
#define TMPC_QUAD_ADP1(REFS_, OS_) \
    hipLaunchKernelGGL((admm_quad_kernel<S, REFS_, RT, XB, OS_, false, true>), dim3(grid), dim3(S::THREADS), 0, stream, P)
                if (P.ref_mode == REF_ZERO) {
                    if (oneshot) TMPC_QUAD_ADP1(REF_ZERO, true); else TMPC_QUAD_ADP1(REF_ZERO, false);
                } else {
                    if (oneshot) TMPC_QUAD_ADP1(REF_SHARED, true); else TMPC_QUAD_ADP1(REF_SHARED, false);
                }
#undef TMPC_QUAD_ADP1
                return hipGetLastError();
            } else {
                return hipErrorInvalidValue;
            }
        } else if constexpr (S::ADP_OK) {
            const bool oneshot = P.cold_start && !P.save_state && P.mpc_steps == 0;
            return oneshot ? launch_quad_adp<S, RT, XB, true>(P, stream) : launch_quad_adp<S, RT, XB, false>(P, stream);
        } else {
            return hipErrorInvalidValue;   // (the solver never selects such an entry for an adaptive solve)
        }
    }
#endif
    if constexpr (S::LOOPV != 0) {
        const bool oneshot = P.cold_start && !P.save_state && P.mpc_steps == 0;
        const bool uniform = oneshot && !(P.abs_pri_tol > 0.f && P.abs_dua_tol > 0.f);  // nobody can converge
        if constexpr ((S::LOOPV & 2) != 0)
            if (uniform && !(P.host_flags & HF_NO_UNI)) return launch_quad_os<S, RT, XB, true, true>(P, stream);
        if constexpr ((S::LOOPV & 1) != 0)
            if (oneshot && !(P.host_flags & HF_NO_OS)) return launch_quad_os<S, RT, XB, true, false>(P, stream);
    }
    return launch_quad_os<S, RT, XB, false, false>(P, stream);
}

// state_bounds_active: false when every state bound is disabled or at/beyond +-1e17 (what
// set_bound_constraints callers pass for "unbounded", e.g. rocket_landing_constraints.jl:36-37);
// the projection is then the identity for any iterate below 1e17 and is compiled out.
template <class S>
hipError_t launch_quad(const AdmmParams &P, int precision, bool state_bounds_active, hipStream_t stream) {
    if (precision == 0)
        return state_bounds_active ? launch_quad_rt<S, double, true>(P, stream)
                                   : launch_quad_rt<S, double, false>(P, stream);
#ifdef TMPC_JIT_UNIT   // a unit specialised at setup (jit.cpp) carries the fp64-recurrence kernels only (the solver's routing knows)
    return hipErrorInvalidValue;
#endif
    return state_bounds_active ? launch_quad_rt<S, float, true>(P, stream)
                               : launch_quad_rt<S, float, false>(P, stream);
}

// Big shapes: the four (RT, XB) launchers are compiled in four translation units (TMPC_QUAD_PART) and only
// declared here, so the unrolled kernels build in parallel.
#define TMPC_QUAD_EXTERN(...)                                                                                \
    extern template hipError_t launch_quad_rt<QuadShape<__VA_ARGS__>, double, true>(const AdmmParams &, hipStream_t);  \
    extern template hipError_t launch_quad_rt<QuadShape<__VA_ARGS__>, double, false>(const AdmmParams &, hipStream_t); \
    extern template hipError_t launch_quad_rt<QuadShape<__VA_ARGS__>, float, true>(const AdmmParams &, hipStream_t);   \
    extern template hipError_t launch_quad_rt<QuadShape<__VA_ARGS__>, float, false>(const AdmmParams &, hipStream_t);
#define TMPC_QUAD_PART(RT_, XB_, ...) \
    template hipError_t launch_quad_rt<QuadShape<__VA_ARGS__>, RT_, XB_>(const AdmmParams &, hipStream_t);

#define TMPC_DEFINE_QUAD_ENTRY(NX, NU, NN, GG, ...)                                            \
    const KernelEntry *quad_entry_##NX##_##NU##_##NN##_g##GG() {                               \
        using S = QuadShape<NX, NU, NN, GG, ##__VA_ARGS__>;                                    \
        static const KernelEntry e = {NX, NU, NN, GG, "quad<" #NX "," #NU "," #NN ",g" #GG ">", \
                                      &build_quad_coef<S>, &build_quad_bounds<S>, &launch_quad<S>, S::ADP_OK}; \
        return &e;                                                                             \
    }

// the same entry under a fixed C name: what a unit specialised at setup exports (jit.cpp); no adaptive-rho variants
#define TMPC_DEFINE_QUAD_JIT_ENTRY(NX, NU, NN, GG)                                                           \
    extern "C" const void *tmpc_jit_entry() {                                                                \
        using S = tmpc::QuadShape<NX, NU, NN, GG>;                                                           \
        static const tmpc::KernelEntry e = {NX, NU, NN, GG, "quad<" #NX "," #NU "," #NN ",g" #GG ">",        \
                                            &tmpc::build_quad_coef<S>, &tmpc::build_quad_bounds<S>, &tmpc::launch_quad<S>, false, true}; \
        return &e;                                                                                           \
    }

}  // namespace tmpc
