// Launcher of one lean-kernel instantiation (admm_lean.hip.h); one translation unit per shape (linst_*.hip).
#pragma once
#include "admm_lean.hip.h"
#include "solver.h"

namespace tmpc {

template <int NX, int NU, int N>
hipError_t launch_lean(const AdmmParams &P, bool live, bool knot_bounds, hipStream_t stream) {
    const int grid = (P.batch + 255) / 256;
    const bool one = grid <= device_cu_count();   // at most one workgroup per CU = one wavefront per SIMD
#define TMPC_LEAN_LAUNCH(LIVE_, UBK_, ONE_) \
    hipLaunchKernelGGL((admm_lean_kernel<NX, NU, N, LIVE_, UBK_, ONE_>), dim3(grid), dim3(256), 0, stream, P)
#define TMPC_LEAN_LAUNCH2(LIVE_, UBK_) \
    do { if (one) TMPC_LEAN_LAUNCH(LIVE_, UBK_, true); else TMPC_LEAN_LAUNCH(LIVE_, UBK_, false); } while (0)
    if (live) {
        if (knot_bounds) TMPC_LEAN_LAUNCH2(true, false); else TMPC_LEAN_LAUNCH2(true, true);
    } else {
        if (knot_bounds) TMPC_LEAN_LAUNCH2(false, false); else TMPC_LEAN_LAUNCH2(false, true);
    }
#undef TMPC_LEAN_LAUNCH2
#undef TMPC_LEAN_LAUNCH
    return hipGetLastError();
}

#define TMPC_DEFINE_LEAN_ENTRY(NX, NU, NN)                                                          \
    const LeanEntry *lean_entry_##NX##_##NU##_##NN() {                                              \
        static const LeanEntry e = {NX, NU, NN, "lean<" #NX "," #NU "," #NN ">", &launch_lean<NX, NU, NN>}; \
        return &e;                                                                                  \
    }

}  // namespace tmpc
