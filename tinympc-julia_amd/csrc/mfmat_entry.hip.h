// Host-side pack builder + launcher for one (nx, nu, N) instantiation of the transposed-sets matrix-core kernel
// (admm_mfmat.hip.h); the lane fields of the operand pack and the bound pack are the LDS kernel's (mfmac_entry.hip.h).
#pragma once
#include "admm_mfmat.hip.h"
#include "mfmac_entry.hip.h"

namespace tmpc {

// lane fields + Pinf as mfmac, then Quu, Quu_inv, Kinf, A, B (row-major) and f: what the sets layout needs for the
// workspace's feed-forward term (t = Quu d, d = Quu_inv t, d = -Kinf x - u) and the closed loop's plant step
template <int NX, int NU, int N>
void build_mfmat_coef(const Solver &sv, std::vector<unsigned char> &out) {
    using T = TransShape<NX, NU, N>;
    build_mfmac_coef<NX, NU>(sv, out);
    out.resize((size_t)T::COEF_DOUBLES * sizeof(double), 0);
    double *o = reinterpret_cast<double *>(out.data());
    const Cache &c = sv.cache;
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

template <int NX, int NU, int N>
size_t mfmat_lds_bytes(const Solver &sv) {
    return TransShape<NX, NU, N>::lds_bytes(mfmac_bounds_vary(sv) ? N : 1);
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
        static int per_cu_dev[64];                                                                                                   \
        int dev = 0;                                                                                                                 \
        (void)hipGetDevice(&dev);                                                                                                    \
        int &per_cu = per_cu_dev[dev & 63];                                                                                          \
        if (per_cu <= 0) {                                                                                                           \
            (void)hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);              \
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, 64, lds) != hipSuccess || per_cu <= 0) per_cu = 1;       \
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
    if (P.ref_mode == REF_ZERO) TMPC_MFMAT_LAUNCH_C(REF_ZERO); else TMPC_MFMAT_LAUNCH_C(REF_SHARED);
#undef TMPC_MFMAT_LAUNCH_C
#undef TMPC_MFMAT_LAUNCH_BV
#undef TMPC_MFMAT_LAUNCH
    return hipGetLastError();
}

// CXA, CXQ / CUA, CUQ: the state / input cone the entry is compiled for (first row, dimension; dimension 0: none);
// problems without a cone on a side use the same entry
#define TMPC_DEFINE_MFMAT_ENTRY(NX, NU, N, CXA, CXQ, CUA, CUQ, PLAIN)                                                            \
    const ConeEntry *mfmat_entry_##NX##_##NU##_##N() {                                                                           \
        static const ConeEntry e = {NX, NU, N, &mfmat_supports<CXA, CXQ, CUA, CUQ>, PLAIN, "mfmat<" #NX "," #NU "," #N ">",          \
                                    &build_mfmat_coef<NX, NU, N>, &build_mfmac_bounds<NX, NU>, &mfmat_lds_bytes<NX, NU, N>,         \
                                    [](const Solver &s) { return mfmat_scratch_floats(s); }, &mfmac_bounds_vary,                   \
                                    &launch_mfmat<NX, NU, N, CXA, CXQ, CUA, CUQ>, true};                                          \
        return &e;                                                                                                               \
    }

}  // namespace tmpc
