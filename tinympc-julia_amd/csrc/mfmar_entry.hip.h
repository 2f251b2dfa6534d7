// Host-side launcher for one (nx, nu, N) instantiation of the register-resident matrix-core kernel (admm_mfmar.hip.h);
// the operand and bound packs are the LDS kernel's (mfmac_entry.hip.h).
#pragma once
#include "admm_mfmar.hip.h"
#include "mfmac_entry.hip.h"

namespace tmpc {

template <int NX, int NU, int N>
size_t mfmar_lds_bytes(const Solver &sv) {
    return RegShape<NX, NU, N>::lds_bytes(mfmac_bounds_vary(sv) ? N : 1);
}

// the state cone's rows must be rows of slot 0 (0 .. 3): wave 1 owns them together with their box set
inline bool mfmar_supports(const Solver &sv) {
    if (sv.st.en_state_soc && sv.ncx > 0 && sv.Acx[0] + sv.qcx[0] > 4) return false;
    if ((sv.st.en_state_soc && sv.ncx > 1) || (sv.st.en_input_soc && sv.ncu > 1)) return false;   // one cone per side
    if (sv.lin_active()) return false;                                                            // no linear rows
    return true;
}

template <int NX, int NU, int N>
hipError_t launch_mfmar(const AdmmParams &P_, bool ext, size_t lds, hipStream_t stream) {
    AdmmParams P = P_;
    P.mpc_steps = 0;
#ifdef TMPC_MFMAC_PROBE
    if (std::getenv("TINYMPC_HIP_MFMAC_DEBUG")) P.mpc_steps = std::atoi(std::getenv("TINYMPC_HIP_MFMAC_DEBUG")) & 40;   // timing probe build only
#endif
    const int tiles = (P.batch + 15) / 16;
    // the termination check can end instances: every checking iteration reads the previous slack back (kernel variant PF)
    const bool live_check = P.abs_pri_tol > 0.f && P.abs_dua_tol > 0.f && P.check_termination > 0;
    // persistent workgroups (the kernel takes tiles off a counter): as many as fit on the chip at once
    const int cus = device_cu_count();   // (per device: a sharded handle launches on several)
#define TMPC_MFMAR_LAUNCH(REFS_, CX_, CU_, BV_, PF_)                                                                          \
    do {                                                                                                                 \
        (void)hipFuncSetAttribute((const void *)admm_mfmar_kernel<NX, NU, N, REFS_, CX_, CU_, BV_, PF_>,                      \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                 \
        int per_cu = 0;                                                                                                  \
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, admm_mfmar_kernel<NX, NU, N, REFS_, CX_, CU_, BV_, PF_>,    \
                                                         192, lds) != hipSuccess || per_cu <= 0)                         \
            per_cu = 1;                                                                                                  \
        const int grid = tiles < per_cu * cus ? tiles : per_cu * cus;                                                    \
        hipLaunchKernelGGL((admm_mfmar_kernel<NX, NU, N, REFS_, CX_, CU_, BV_, PF_>), dim3(grid), dim3(192), lds, stream, P); \
    } while (0)
#define TMPC_MFMAR_LAUNCH_PF(REFS_, CX_, CU_, BV_)                                                                      \
    do {                                                                                                               \
        if (live_check) TMPC_MFMAR_LAUNCH(REFS_, CX_, CU_, BV_, true); else TMPC_MFMAR_LAUNCH(REFS_, CX_, CU_, BV_, false); \
    } while (0)
#define TMPC_MFMAR_LAUNCH_BV(REFS_, CX_, CU_)                                                                          \
    do {                                                                                                               \
        if (P.bounds_stride) TMPC_MFMAR_LAUNCH_PF(REFS_, CX_, CU_, true); else TMPC_MFMAR_LAUNCH_PF(REFS_, CX_, CU_, false); \
    } while (0)
#define TMPC_MFMAR_LAUNCH_C(REFS_)                                                         \
    do {                                                                                   \
        if (P.ncx > 0 && P.ncu > 0) TMPC_MFMAR_LAUNCH_BV(REFS_, 1, 1);                     \
        else if (P.ncx > 0) TMPC_MFMAR_LAUNCH_BV(REFS_, 1, 0);                             \
        else if (P.ncu > 0) TMPC_MFMAR_LAUNCH_BV(REFS_, 0, 1);                             \
        else TMPC_MFMAR_LAUNCH_BV(REFS_, 0, 0);                                            \
    } while (0)
    (void)ext;
    if (P.ref_mode == REF_ZERO) TMPC_MFMAR_LAUNCH_C(REF_ZERO); else TMPC_MFMAR_LAUNCH_C(REF_SHARED);
#undef TMPC_MFMAR_LAUNCH_C
#undef TMPC_MFMAR_LAUNCH_BV
#undef TMPC_MFMAR_LAUNCH_PF
#undef TMPC_MFMAR_LAUNCH
    return hipGetLastError();
}

#define TMPC_DEFINE_MFMAR_ENTRY(NX, NU, N, PLAIN)                                                                         \
    const ConeEntry *mfmar_entry_##NX##_##NU##_##N() {                                                              \
        static const ConeEntry e = {NX, NU, N, &mfmar_supports, PLAIN, "mfmar<" #NX "," #NU "," #N ">",                    \
                                    &build_mfmac_coef<NX, NU>, &build_mfmac_bounds<NX, NU>, &mfmar_lds_bytes<NX, NU, N>, \
                                    &mfmac_scratch_floats<NX, NU>, &mfmac_bounds_vary, &launch_mfmar<NX, NU, N>};   \
        return &e;                                                                                                  \
    }

}  // namespace tmpc
