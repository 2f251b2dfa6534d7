// Host-side pack builder + launcher for one (nx, nu, N) instantiation of the transposed-sets matrix-core kernel
// (admm_mfmat.hip.h); the lane fields of the operand pack and the bound pack are the LDS kernel's (mfmac_entry.hip.h).
#pragma once
#include <atomic>
#include "admm_mfmat.hip.h"
#include "mfmac_entry.hip.h"

namespace tmpc {

// The operand pack of the transposed-sets kernel (layout: TransShape).  Lane l supplies, for the 16 x 4 A operand of a
// slot, tile row l % 16 and K column l / 16 of the slot; tile rows 0..7 are state rows, 8 + a input rows (as everywhere in
// the matrix-core kernels); K columns 0..3 of slot 0 are x_0..x_3, those of slot 1 x_4..x_{nx-1} followed by the first MU
// input components.  Behind the lane fields: the per-lane-group constants (affine terms; the columns of the VU input
// components that are multiplied on the VALU), then what the sets layout needs for the workspace's feed-forward term
// (t = Quu d, d = Quu_inv t, d = -Kinf x - u) and the closed loop's plant step.
template <int NX, int NU, int N>
void build_mfmat_coef(const Solver &sv, std::vector<unsigned char> &out) {
    using T = TransShape<NX, NU, N>;
    constexpr int MU = T::MU, VU = T::VU;
    out.assign((size_t)T::COEF_DOUBLES * sizeof(double), 0);
    double *o = reinterpret_cast<double *>(out.data());
    const Cache &c = sv.cache;
    double Pf[NX], APf[NX], BPf[NU], ABK[NX][NX], BQ[NX][NU];
    for (int i = 0; i < NX; ++i) {
        Pf[i] = 0.0;
        for (int k = 0; k < NX; ++k) Pf[i] += c.Pinf(i, k) * sv.fdyn[k];
    }
    for (int i = 0; i < NX; ++i) {
        APf[i] = 0.0;
        for (int k = 0; k < NX; ++k) APf[i] += c.AmBKt(i, k) * Pf[k];
    }
    for (int a = 0; a < NU; ++a) {
        BPf[a] = 0.0;
        for (int k = 0; k < NX; ++k) BPf[a] += sv.B(k, a) * Pf[k];
    }
    for (int i = 0; i < NX; ++i) {
        for (int k = 0; k < NX; ++k) {
            ABK[i][k] = sv.A(i, k);                    // A - B Kinf from A, B, Kinf themselves (set_cache_terms may hand in an
            for (int a = 0; a < NU; ++a) ABK[i][k] -= sv.B(i, a) * c.Kinf(a, k);   // AmBKt that differs)
        }
        for (int a = 0; a < NU; ++a) {
            BQ[i][a] = 0.0;
            for (int b = 0; b < NU; ++b) BQ[i][a] += sv.B(i, b) * c.Quu_inv(b, a);   // B Quu_inv
        }
    }
    const bool fd = sv.has_fdyn;
    for (int l = 0; l < 64; ++l) {
        const int m = l % 16, kq = l / 16;
        const bool mx = m < 8 && m < NX, mu = m >= 8 && m - 8 < NU;
        // slot 0: x_kq
        if (kq < NX) {
            o[T::L_MF0 * 64 + l] = mx ? ABK[m][kq] : (mu ? -c.Kinf(m - 8, kq) : 0.0);
            o[T::L_MB0 * 64 + l] = mx ? c.AmBKt(m, kq) : (mu ? sv.B(kq, m - 8) : 0.0);
        }
        // slot 1: x_{4+kq}, or input component 4 + kq - NX
        const int col = 4 + kq;
        if (col < NX) {
            o[T::L_MF1 * 64 + l] = mx ? ABK[m][col] : (mu ? -c.Kinf(m - 8, col) : 0.0);
            o[T::L_MB1 * 64 + l] = mx ? c.AmBKt(m, col) : (mu ? sv.B(col, m - 8) : 0.0);
        } else if (col - NX < MU) {
            const int a = col - NX;
            o[T::L_MF1 * 64 + l] = mx ? -BQ[m][a] : (mu ? -c.Quu_inv(m - 8, a) : 0.0);
            o[T::L_MB1 * 64 + l] = mx ? -c.Kinf(a, m) : 0.0;
        }
    }
    double *kc = o + T::O_KC;
    for (int g = 0; g < 4; ++g) {
        const bool x1 = 4 + g < NX, ug = g < NU;
        kc[T::K_FD0 * 4 + g] = fd ? sv.fdyn[g] : 0.0;
        kc[T::K_FD1 * 4 + g] = fd && x1 ? sv.fdyn[4 + g] : 0.0;
        kc[T::K_APF0 * 4 + g] = fd ? APf[g] : 0.0;
        kc[T::K_APF1 * 4 + g] = fd && x1 ? APf[4 + g] : 0.0;
        kc[T::K_BPF * 4 + g] = fd && ug ? BPf[g] : 0.0;
        for (int v = 0; v < VU; ++v) {
            const int a = MU + v;
            kc[(T::K_GF0 + v) * 4 + g] = -BQ[g][a];
            kc[(T::K_GF1 + v) * 4 + g] = x1 ? -BQ[4 + g][a] : 0.0;
            kc[(T::K_GF2 + v) * 4 + g] = ug ? -c.Quu_inv(g, a) : 0.0;
            kc[(T::K_GB0 + v) * 4 + g] = -c.Kinf(a, g);
            kc[(T::K_GB1 + v) * 4 + g] = x1 ? -c.Kinf(a, 4 + g) : 0.0;
        }
    }
    for (int i = 0; i < NX; ++i)
        for (int k = 0; k < NX; ++k) o[T::O_PINF + i * NX + k] = c.Pinf(i, k);
    // Quu = Quu_inv^-1 (Gauss-Jordan with partial pivoting, nu <= 4)
    double M[NU][2 * NU];
    for (int a = 0; a < NU; ++a)
        for (int b = 0; b < NU; ++b) M[a][b] = c.Quu_inv(a, b), M[a][NU + b] = a == b ? 1.0 : 0.0;
    for (int col = 0; col < NU; ++col) {
        int piv = col;
        for (int a = col + 1; a < NU; ++a)
            if (std::fabs(M[a][col]) > std::fabs(M[piv][col])) piv = a;
        for (int b = 0; b < 2 * NU; ++b) std::swap(M[col][b], M[piv][b]);
        const double d = M[col][col];
        for (int b = 0; b < 2 * NU; ++b) M[col][b] /= d;
        for (int a = 0; a < NU; ++a)
            if (a != col) {
                const double f = M[a][col];
                for (int b = 0; b < 2 * NU; ++b) M[a][b] -= f * M[col][b];
            }
    }
    for (int a = 0; a < NU; ++a)
        for (int b = 0; b < NU; ++b) {
            o[T::O_QUU + a * NU + b] = M[a][NU + b];
            o[T::O_QUI + a * NU + b] = c.Quu_inv(a, b);
        }
    for (int a = 0; a < NU; ++a)
        for (int r = 0; r < NX; ++r) o[T::O_KINF + a * NX + r] = c.Kinf(a, r);
    for (int r = 0; r < NX; ++r) {
        for (int k = 0; k < NX; ++k) o[T::O_A + r * NX + k] = sv.A(r, k);
        for (int a = 0; a < NU; ++a) o[T::O_B + r * NU + a] = sv.B(r, a);
        o[T::O_F + r] = sv.has_fdyn ? sv.fdyn[r] : 0.0;
    }
}

template <int NX, int NU, int N, int CXQ, int CUQ>
size_t mfmat_lds_bytes(const Solver &sv) {
    const bool cx = sv.st.en_state_soc && sv.ncx > 0, cu = sv.st.en_input_soc && sv.ncu > 0;
    return TransShape<NX, NU, N>::lds_bytes(mfmac_bounds_vary(sv) ? N : 1, cx ? CXQ : 0, cu ? CUQ : 0, sv.refs_per_instance());
}

inline size_t mfmat_scratch_floats(const Solver &) { return 1; }   // nothing of the iterated state goes through HBM

// an enabled cone must be the one the entry is compiled for (its rows are registers of a lane)
template <int CXA, int CXQ, int CUA, int CUQ>
bool mfmat_supports(const Solver &sv) {
    if (sv.st.en_state_soc && sv.ncx > 0 && (sv.ncx > 1 || CXQ == 0 || sv.Acx[0] != CXA || sv.qcx[0] != CXQ)) return false;
    if (sv.st.en_input_soc && sv.ncu > 0 && (sv.ncu > 1 || CUQ == 0 || sv.Acu[0] != CUA || sv.qcu[0] != CUQ)) return false;
    return true;
}

template <int NX, int NU, int N, int CXA, int CXQ, int CUA, int CUQ>
hipError_t launch_mfmat(const AdmmParams &P, bool ext, size_t lds, hipStream_t stream) {
    const int tiles = (P.batch + 15) / 16;
    const int cus = device_cu_count();
    // persistent workgroups (the kernel takes tiles off a counter): as many as fit on the chip at once
#define TMPC_MFMAT_LAUNCH(REFS_, CXQ_, CUQ_, BV_)                                                                                    \
    do {                                                                                                                             \
        auto kfn = admm_mfmat_kernel<NX, NU, N, REFS_, (CXQ_) ? CXA : 0, CXQ_, (CUQ_) ? CUA : 0, CUQ_, BV_>;                             \
        static std::atomic<int> per_cu_dev[64];                                                                                                   \
        int dev = 0;                                                                                                                 \
        (void)hipGetDevice(&dev);                                                                                                    \
        int per_cu = per_cu_dev[dev & 63].load(std::memory_order_relaxed);                                                                                          \
        if (per_cu <= 0) {                                                                                                           \
            (void)hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);              \
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, 64, lds) != hipSuccess || per_cu <= 0) per_cu = 1;       \
            per_cu_dev[dev & 63].store(per_cu, std::memory_order_relaxed);                                                            \
        }                                                                                                                            \
        const int grid = tiles < per_cu * cus ? tiles : per_cu * cus;                                                                \
        hipLaunchKernelGGL(kfn, dim3(grid), dim3(64), lds, stream, P);                                                               \
    } while (0)
#define TMPC_MFMAT_LAUNCH_BV(REFS_, CXQ_, CUQ_)                                                                      \
    do {                                                                                                            \
        if (P.bounds_stride) TMPC_MFMAT_LAUNCH(REFS_, CXQ_, CUQ_, true); else TMPC_MFMAT_LAUNCH(REFS_, CXQ_, CUQ_, false); \
    } while (0)
#define TMPC_MFMAT_LAUNCH_C(REFS_)                                                              \
    do {                                                                                        \
        if (P.ncx > 0 && P.ncu > 0) { if constexpr (CXQ > 0 && CUQ > 0) TMPC_MFMAT_LAUNCH_BV(REFS_, CXQ, CUQ); }    \
        else if (P.ncx > 0) { if constexpr (CXQ > 0) TMPC_MFMAT_LAUNCH_BV(REFS_, CXQ, 0); }                  \
        else if (P.ncu > 0) { if constexpr (CUQ > 0) TMPC_MFMAT_LAUNCH_BV(REFS_, 0, CUQ); }                  \
        else TMPC_MFMAT_LAUNCH_BV(REFS_, 0, 0);                                                 \
    } while (0)
    (void)ext;
    if (P.ref_mode == REF_ZERO) TMPC_MFMAT_LAUNCH_C(REF_ZERO);
    else if (P.ref_mode == REF_PER_INSTANCE) TMPC_MFMAT_LAUNCH_C(REF_PER_INSTANCE);
    else TMPC_MFMAT_LAUNCH_C(REF_SHARED);
#undef TMPC_MFMAT_LAUNCH_C
#undef TMPC_MFMAT_LAUNCH_BV
#undef TMPC_MFMAT_LAUNCH
    return hipGetLastError();
}

// The eight kernels of one reference mode (cones on both / one / no side x bounds per knot or not), as explicit
// instantiation definitions (EXT empty) or declarations (EXT = extern): an entry's 24 kernels are compiled in three
// translation units, one per reference mode, and the entry's own unit only launches them — the N = 50 family alone is
// eight minutes of compiler time in one unit.
#define TMPC_MFMAT_KERNELS(EXT, NX, NU, N, REFS, CXA, CXQ, CUA, CUQ)                                              \
    EXT template __global__ void admm_mfmat_kernel<NX, NU, N, REFS, CXA, CXQ, CUA, CUQ, false>(const AdmmParams); \
    EXT template __global__ void admm_mfmat_kernel<NX, NU, N, REFS, CXA, CXQ, CUA, CUQ, true>(const AdmmParams);  \
    EXT template __global__ void admm_mfmat_kernel<NX, NU, N, REFS, CXA, CXQ, 0, 0, false>(const AdmmParams);     \
    EXT template __global__ void admm_mfmat_kernel<NX, NU, N, REFS, CXA, CXQ, 0, 0, true>(const AdmmParams);      \
    EXT template __global__ void admm_mfmat_kernel<NX, NU, N, REFS, 0, 0, CUA, CUQ, false>(const AdmmParams);     \
    EXT template __global__ void admm_mfmat_kernel<NX, NU, N, REFS, 0, 0, CUA, CUQ, true>(const AdmmParams);      \
    EXT template __global__ void admm_mfmat_kernel<NX, NU, N, REFS, 0, 0, 0, 0, false>(const AdmmParams);         \
    EXT template __global__ void admm_mfmat_kernel<NX, NU, N, REFS, 0, 0, 0, 0, true>(const AdmmParams);
#define TMPC_MFMAT_KERNELS_EXTERN(NX, NU, N, CXA, CXQ, CUA, CUQ)                  \
    TMPC_MFMAT_KERNELS(extern, NX, NU, N, REF_ZERO, CXA, CXQ, CUA, CUQ)           \
    TMPC_MFMAT_KERNELS(extern, NX, NU, N, REF_SHARED, CXA, CXQ, CUA, CUQ)         \
    TMPC_MFMAT_KERNELS(extern, NX, NU, N, REF_PER_INSTANCE, CXA, CXQ, CUA, CUQ)

// CXA, CXQ / CUA, CUQ: the state / input cone the entry is compiled for (first row, dimension; dimension 0: none);
// problems without a cone on a side use the same entry
#define TMPC_DEFINE_MFMAT_ENTRY(NX, NU, N, CXA, CXQ, CUA, CUQ, PLAIN)                                                            \
    const ConeEntry *mfmat_entry_##NX##_##NU##_##N() {                                                                           \
        static const ConeEntry e = {NX, NU, N, &mfmat_supports<CXA, CXQ, CUA, CUQ>, PLAIN, "mfmat<" #NX "," #NU "," #N ">",          \
                                    &build_mfmat_coef<NX, NU, N>, &build_mfmac_bounds<NX, NU>, &mfmat_lds_bytes<NX, NU, N, CXQ, CUQ>,         \
                                    [](const Solver &s) { return mfmat_scratch_floats(s); }, &mfmac_bounds_vary,                   \
                                    &launch_mfmat<NX, NU, N, CXA, CXQ, CUA, CUQ>, true};                                          \
        return &e;                                                                                                               \
    }

// ---- one kernel specialised at setup (jit.cpp): exactly the solver's constraint layout, reference mode and bound kind ----
// (the built-in entries compile one cone per side; cone LISTS and linear rows — bindings.cpp:414-490 — come this way, and so
// does the affine term / cone layout of a horizon the library was not built with)
template <int NX, int NU, int N, int REFS, int CXA, int CXQ, int CUA, int CUQ, bool BV, class GX>
bool mfmat_exact_supports(const Solver &sv) {
    const int ncx = sv.st.en_state_soc ? sv.ncx : 0, ncu = sv.st.en_input_soc ? sv.ncu : 0;
    if (ncx != (CXQ > 0) + (GX::CXQ2 > 0) || ncu != (CUQ > 0) + (GX::CUQ2 > 0)) return false;
    if (CXQ > 0 && (sv.Acx[0] != CXA || sv.qcx[0] != CXQ)) return false;
    if (GX::CXQ2 > 0 && (sv.Acx[1] != GX::CXA2 || sv.qcx[1] != GX::CXQ2)) return false;
    if (CUQ > 0 && (sv.Acu[0] != CUA || sv.qcu[0] != CUQ)) return false;
    if (GX::CUQ2 > 0 && (sv.Acu[1] != GX::CUA2 || sv.qcu[1] != GX::CUQ2)) return false;
    if ((sv.st.en_state_linear ? sv.mlx : 0) != GX::MLX || (sv.st.en_input_linear ? sv.mlu : 0) != GX::MLU) return false;
    // (the mode the next launch uploads the references in, Solver::upload_refs — ref_mode itself is the last launch's)
    const int refs = sv.refs_device_owned ? sv.ref_mode : (sv.xref_kind > sv.uref_kind ? sv.xref_kind : sv.uref_kind);
    return refs == REFS && mfmac_bounds_vary(sv) == BV;
}
template <int NX, int NU, int N, int REFS, int CXQ, int CUQ, bool BV, class GX>
size_t mfmat_exact_lds_bytes(const Solver &) {
    return TransShape<NX, NU, N>::lds_bytes(BV ? N : 1, CXQ + GX::CXQ2, CUQ + GX::CUQ2, REFS == REF_PER_INSTANCE, GX::MLX > 0, GX::MLU > 0);
}
template <int NX, int NU, int N, int REFS, int CXA, int CXQ, int CUA, int CUQ, bool BV, class GX>
hipError_t launch_mfmat_exact(const AdmmParams &P, bool, size_t lds, hipStream_t stream) {
    auto kfn = admm_mfmat_kernel<NX, NU, N, REFS, CXA, CXQ, CUA, CUQ, BV, GX>;
    static std::atomic<int> per_cu_dev[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    int per_cu = per_cu_dev[dev & 63].load(std::memory_order_relaxed);
    if (per_cu <= 0) {
        (void)hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, 64, lds) != hipSuccess || per_cu <= 0) per_cu = 1;
        per_cu_dev[dev & 63].store(per_cu, std::memory_order_relaxed);
    }
    const int tiles = (P.batch + 15) / 16, slots = per_cu * device_cu_count();
    hipLaunchKernelGGL(kfn, dim3(tiles < slots ? tiles : slots), dim3(64), lds, stream, P);
    return hipGetLastError();
}
#define TMPC_DEFINE_MFMAT_JIT_ENTRY(NAME, NX, NU, N, REFS, CXA, CXQ, CUA, CUQ, BV, CXA2, CXQ2, CUA2, CUQ2, MLX, MLU)                  \
    namespace tmpc {                                                                                                                \
    using JitExtra = TransExtra<CXA2, CXQ2, CUA2, CUQ2, MLX, MLU>;                                                                   \
    const ConeEntry *mfmat_jit_entry() {                                                                                            \
        static const ConeEntry e = {NX, NU, N, &mfmat_exact_supports<NX, NU, N, REFS, CXA, CXQ, CUA, CUQ, BV, JitExtra>, true, NAME, \
                                    &build_mfmat_coef<NX, NU, N>, &build_mfmac_bounds<NX, NU>,                                       \
                                    &mfmat_exact_lds_bytes<NX, NU, N, REFS, CXQ, CUQ, BV, JitExtra>,                                 \
                                    [](const Solver &s) { return mfmat_scratch_floats(s); }, &mfmac_bounds_vary,                    \
                                    &launch_mfmat_exact<NX, NU, N, REFS, CXA, CXQ, CUA, CUQ, BV, JitExtra>, true};                   \
        return &e;                                                                                                                  \
    }                                                                                                                               \
    }                                                                                                                               \
    extern "C" const void *tmpc_jit_entry() { return tmpc::mfmat_jit_entry(); }

}  // namespace tmpc
