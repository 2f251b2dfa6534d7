// Host-side pack builders + launcher for one (nx, nu, N) instantiation of the matrix-core kernel (admm_mfma.hip.h).
#pragma once
#include <cstdlib>
#include <cstring>
#include <limits>

#include "admm_mfma.hip.h"
#include "solver.h"

namespace tmpc {

// operand doubles of every lane: [field][64]; lane l supplies row l % 16, K-column l / 16 of each 16 x 4 operand
template <int NX, int NU, int N>
void build_mfma_coef(const Solver &sv, std::vector<unsigned char> &out) {
    using S = MfmaShape<NX, NU, N>;
    out.assign((size_t)S::COEF_DOUBLES * sizeof(double), 0);
    double *o = reinterpret_cast<double *>(out.data());
    const Cache &c = sv.cache;
    for (int l = 0; l < 64; ++l) {
        const int i = l % 16, kq = l / 16;          // result row (= 4 v + g), K-column within a slice
        const bool xrow = i < NX;                   // rows 0..11: state rows
        const int a = i - 12;                       // rows 12..15: input rows
        const bool urow = a >= 0 && a < NU;
        for (int s = 0; s < S::VX; ++s) {
            const int col = 4 * s + kq;             // state index the slice's column stands for
            double mf = 0.0, mb = 0.0, pt = 0.0, at = 0.0, sp = 0.0;
            if (col < NX) {
                // (dPinf/drho)^T [i][col]: row i of the product = sum_col dPinf[col][i] x_col, like Pinf^T above
                if (xrow && sv.sens.size() == (size_t)NU * NX + (size_t)NX * NX) sp = sv.sens[(size_t)NU * NX + col + (size_t)i * NX];
                at = xrow ? sv.A(col, i) : (urow ? sv.B(col, a) : 0.0);   // ([A'; B'])[i][col]: A'g, B'g of the adaptive-rho norms
                if (xrow) {
                    mf = sv.A(i, col);              // (A - B Kinf)[i][col], from A, B, Kinf themselves (set_cache_terms may
                    for (int a2 = 0; a2 < NU; ++a2) mf -= sv.B(i, a2) * c.Kinf(a2, col);   // hand in an AmBKt that differs)
                    mb = c.AmBKt(i, col);
                    pt = c.Pinf(col, i);            // (Pinf^T)[i][col]
                } else if (urow) {
                    mf = -c.Kinf(a, col);           // u = -d - Kinf x
                    mb = sv.B(col, a);              // (B^T)[a][col]
                }
            }
            o[(S::O_MF + s) * 64 + l] = mf;
            o[(S::O_MB + s) * 64 + l] = mb;
            o[(S::O_PT + s) * 64 + l] = pt;
            o[(S::O_AT + s) * 64 + l] = at;
            o[(S::O_SP + s) * 64 + l] = sp;
        }
        o[S::O_BF * 64 + l] = (xrow && kq < NU) ? sv.B(i, kq) : 0.0;
        o[S::O_KT * 64 + l] = (xrow && kq < NU) ? -c.Kinf(kq, i) : 0.0;   // -(Kinf^T)[i][kq]
        o[S::O_QI * 64 + l] = (urow && kq < NU) ? c.Quu_inv(a, kq) : 0.0;
    }
    for (int a = 0; a < NU; ++a)
        for (int r = 0; r < NX; ++r) o[S::O_K0 + a * NX + r] = c.Kinf(a, r);
    for (int jj = 0; jj < NX; ++jj)
        for (int r = 0; r < NX; ++r) o[S::O_P0 + jj * NX + r] = c.Pinf(jj, r);
    o[S::O_RHO0] = c.rho;
}

template <int NX, int NU, int N>
void build_mfma_bounds(const Solver &sv, std::vector<float> &out) {
    using S = MfmaShape<NX, NU, N>;
    constexpr float kInf = std::numeric_limits<float>::infinity();
    out.assign((size_t)S::BOUNDS_LEN, 0.f);
    for (int k = 0; k < N; ++k)
        for (int r = 0; r < NX; ++r) {
            out[S::B_XMIN + k * NX + r] = sv.st.en_state_bound ? (float)sv.x_min[r + (size_t)k * NX] : -kInf;
            out[S::B_XMAX + k * NX + r] = sv.st.en_state_bound ? (float)sv.x_max[r + (size_t)k * NX] : kInf;
        }
    for (int k = 0; k < N - 1; ++k)
        for (int a = 0; a < NU; ++a) {
            out[S::B_UMIN + k * NU + a] = sv.st.en_input_bound ? (float)sv.u_min[a + (size_t)k * NU] : -kInf;
            out[S::B_UMAX + k * NU + a] = sv.st.en_input_bound ? (float)sv.u_max[a + (size_t)k * NU] : kInf;
        }
    for (int r = 0; r < NX; ++r) out[S::B_QD + r] = (float)sv.cache.Qd[r];
    for (int a = 0; a < NU; ++a) out[S::B_RD + a] = (float)sv.cache.Rd[a];
}

template <int NX, int NU, int N, bool XB, bool WS>
hipError_t launch_mfma_xb(const AdmmParams &P, hipStream_t stream) {
    const int grid = (P.batch + 63) / 64;
    const size_t lds = WS ? mfma_ws_lds_bytes<NX, NU, N>() : 0;
#define TMPC_MFMA_LAUNCH(REFS_)                                                                                     \
    do {                                                                                                            \
        if (lds > 48 * 1024)                                                                                        \
            (void)hipFuncSetAttribute((const void *)admm_mfma_kernel<NX, NU, N, REFS_, XB, WS>,                     \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                        \
        hipLaunchKernelGGL((admm_mfma_kernel<NX, NU, N, REFS_, XB, WS>), dim3(grid), dim3(256), lds, stream, P);    \
    } while (0)
    switch (P.ref_mode) {
        case REF_ZERO: TMPC_MFMA_LAUNCH(REF_ZERO); break;
        case REF_SHARED: TMPC_MFMA_LAUNCH(REF_SHARED); break;
        default: TMPC_MFMA_LAUNCH(REF_PER_INSTANCE); break;
    }
#undef TMPC_MFMA_LAUNCH
    return hipGetLastError();
}

// Tolerance-terminated one-shot solves of more instances than the chip holds at once: the refill variant (admm_mfma.hip.h,
// RF) on as many workgroups as are resident together.  Returns hipErrorNotReady when the launch is not such a case.
template <int NX, int NU, int N, bool XB>
hipError_t launch_mfma_refill(const AdmmParams &P, hipStream_t stream) {
    const int tiles = (P.batch + 63) / 64, ct = P.check_termination;
    if (!(P.abs_pri_tol > 0.f && P.abs_dua_tol > 0.f) || ct <= 0 || P.max_iter % ct != 0 || P.idx != nullptr ||
        P.ref_mode == REF_PER_INSTANCE || P.x0d != nullptr || P.batch % 64 != 0 || (P.host_flags & HF_NO_REFILL))
        return hipErrorNotReady;
    const int cus = device_cu_count();   // (per device: a sharded handle launches on several)
#define TMPC_MFMA_RF(REFS_)                                                                                          \
    do {                                                                                                             \
        int per_cu = 0;                                                                                              \
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, admm_mfma_kernel<NX, NU, N, REFS_, XB, false, true>, \
                                                         256, 0) != hipSuccess || per_cu <= 0)                       \
            per_cu = 1;                                                                                              \
        if (tiles < 2 * per_cu * cus) return hipErrorNotReady;   /* fewer than two rounds: nothing to refill */       \
        hipLaunchKernelGGL((admm_mfma_kernel<NX, NU, N, REFS_, XB, false, true>), dim3(per_cu * cus), dim3(256), 0,   \
                           stream, P);                                                                               \
    } while (0)
    if (P.ref_mode == REF_ZERO) TMPC_MFMA_RF(REF_ZERO); else TMPC_MFMA_RF(REF_SHARED);
#undef TMPC_MFMA_RF
    return hipGetLastError();
}

// precision is ignored: the matrix cores run the recurrences in fp64 (precision = 1 solves of the shape run here too unless the
// caller asked for strict precision: tinympc_set_strict_precision; tinympc_effective_precision reports what runs).
// A launch that reads or keeps the workspace takes the WS variant (old slack parked in LDS).  `state_bounds_active`
// here also covers "the workspace's state dual may be non-zero" (Solver::launch_pass) — only then is g carried.
// adaptive rho (admm.cpp:147-174): the ADP variants — one-shot or workspace-carrying, no refill
template <int NX, int NU, int N, bool XB, bool WS>
hipError_t launch_mfma_adp(const AdmmParams &P, hipStream_t stream) {
    const int grid = (P.batch + 63) / 64;
    const size_t lds = WS ? mfma_ws_lds_bytes<NX, NU, N>() : 0;
#define TMPC_MFMA_LAUNCH_ADP(REFS_)                                                                                 \
    do {                                                                                                            \
        if (lds > 48 * 1024)                                                                                        \
            (void)hipFuncSetAttribute((const void *)admm_mfma_kernel<NX, NU, N, REFS_, XB, WS, false, true>,        \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                        \
        hipLaunchKernelGGL((admm_mfma_kernel<NX, NU, N, REFS_, XB, WS, false, true>), dim3(grid), dim3(256), lds, stream, P); \
    } while (0)
    switch (P.ref_mode) {
        case REF_ZERO: TMPC_MFMA_LAUNCH_ADP(REF_ZERO); break;
        case REF_SHARED: TMPC_MFMA_LAUNCH_ADP(REF_SHARED); break;
        default: TMPC_MFMA_LAUNCH_ADP(REF_PER_INSTANCE); break;
    }
#undef TMPC_MFMA_LAUNCH_ADP
    return hipGetLastError();
}

template <int NX, int NU, int N>
hipError_t launch_mfma(const AdmmParams &P, int /*precision*/, bool state_bounds_active, hipStream_t stream) {
    if (P.adaptive_rho) {
#ifdef TMPC_JIT_UNIT   // a unit specialised at setup (jit.cpp) carries no adaptive-rho variants: the solver never routes such a solve here
        return hipErrorInvalidValue;
#else
        const bool ws = !P.cold_start || P.save_state;
        if (state_bounds_active) return ws ? launch_mfma_adp<NX, NU, N, true, true>(P, stream) : launch_mfma_adp<NX, NU, N, true, false>(P, stream);
        return ws ? launch_mfma_adp<NX, NU, N, false, true>(P, stream) : launch_mfma_adp<NX, NU, N, false, false>(P, stream);
#endif
    }
    if (!P.cold_start || P.save_state)
        return state_bounds_active ? launch_mfma_xb<NX, NU, N, true, true>(P, stream)
                                   : launch_mfma_xb<NX, NU, N, false, true>(P, stream);
    const hipError_t rf = state_bounds_active ? launch_mfma_refill<NX, NU, N, true>(P, stream)
                                              : launch_mfma_refill<NX, NU, N, false>(P, stream);
    if (rf != hipErrorNotReady) return rf;
    return state_bounds_active ? launch_mfma_xb<NX, NU, N, true, false>(P, stream)
                               : launch_mfma_xb<NX, NU, N, false, false>(P, stream);
}

// The twelve adaptive-rho kernels of a shape (reference mode x state bounds x workspace), as explicit instantiation
// definitions (EXT empty) or declarations (EXT = extern): they are compiled in a translation unit of their own
// (minst_*_adp.hip) — with them the N = 30 unit took six minutes.
#define TMPC_MFMA_ADP_KERNELS_R(EXT, NX, NU, NN, REFS)                                                              \
    EXT template __global__ void admm_mfma_kernel<NX, NU, NN, REFS, false, false, false, true>(const AdmmParams);  \
    EXT template __global__ void admm_mfma_kernel<NX, NU, NN, REFS, false, true, false, true>(const AdmmParams);   \
    EXT template __global__ void admm_mfma_kernel<NX, NU, NN, REFS, true, false, false, true>(const AdmmParams);   \
    EXT template __global__ void admm_mfma_kernel<NX, NU, NN, REFS, true, true, false, true>(const AdmmParams);
#define TMPC_MFMA_ADP_KERNELS(EXT, NX, NU, NN)                    \
    TMPC_MFMA_ADP_KERNELS_R(EXT, NX, NU, NN, REF_ZERO)            \
    TMPC_MFMA_ADP_KERNELS_R(EXT, NX, NU, NN, REF_SHARED)          \
    TMPC_MFMA_ADP_KERNELS_R(EXT, NX, NU, NN, REF_PER_INSTANCE)

#define TMPC_DEFINE_MFMA_ENTRY(NX, NU, NN)                                                                 \
    const KernelEntry *mfma_entry_##NX##_##NU##_##NN() {                                                   \
        static const KernelEntry e = {NX, NU, NN, 16, "mfma<" #NX "," #NU "," #NN ">", &build_mfma_coef<NX, NU, NN>, \
                                      &build_mfma_bounds<NX, NU, NN>, &launch_mfma<NX, NU, NN>, true};     \
        return &e;                                                                                         \
    }

// the same entry without adaptive-rho variants, under a fixed C name: what a unit specialised at setup exports (jit.cpp)
#define TMPC_DEFINE_MFMA_JIT_ENTRY(NX, NU, NN)                                                                           \
    extern "C" const void *tmpc_jit_entry() {                                                                            \
        static const tmpc::KernelEntry e = {NX, NU, NN, 16, "mfma<" #NX "," #NU "," #NN ">", &tmpc::build_mfma_coef<NX, NU, NN>, \
                                            &tmpc::build_mfma_bounds<NX, NU, NN>, &tmpc::launch_mfma<NX, NU, NN>, false, true}; \
        return &e;                                                                                                       \
    }

}  // namespace tmpc
