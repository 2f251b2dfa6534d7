// Launcher of one lean-kernel instantiation (admm_lean.hip.h); one translation unit per shape (linst_*.hip).
#pragma once
#include "admm_lean.hip.h"
#include "solver.h"

namespace tmpc {

template <int NX, int NU, int N, bool XB, int REFS>
hipError_t launch_lean_v(const AdmmParams &P, bool live, bool knot_bounds, hipStream_t stream) {
    const int grid = (P.batch + 255) / 256;
    // The 512-register variant when the launch has at most one workgroup per CU (= one wavefront per SIMD), and — at any batch —
    // for tolerance-terminated solves: held to 256 registers the LIVE variants spill (73-187 registers) and lose to 512-register
    // wavefronts taking turns (batch 131 072, check live: 0.92 against 0.69 ms; with a state bound 2.95 against 1.05;
    // fixed-iteration solves: 0.47 / 0.61 against 0.46 / 0.69 — scripts/lean_time.py "big").  TINYMPC_HIP_LEAN_ONE: always (tuning aid).
    const bool one = grid <= device_cu_count() || live || (P.host_flags & HF_LEAN_ONE);
#define TMPC_LEAN_LAUNCH(LIVE_, UBK_, ONE_) \
    hipLaunchKernelGGL((admm_lean_kernel<NX, NU, N, LIVE_, UBK_, ONE_, XB, REFS>), dim3(grid), dim3(256), 0, stream, P)
#define TMPC_LEAN_LAUNCH2(LIVE_, UBK_) \
    do { if (one) TMPC_LEAN_LAUNCH(LIVE_, UBK_, true); else TMPC_LEAN_LAUNCH(LIVE_, UBK_, false); } while (0)
    if (live) {   // (always the 512-register variant: the 256-register LIVE kernels are not even built)
        if (knot_bounds) TMPC_LEAN_LAUNCH(true, false, true); else TMPC_LEAN_LAUNCH(true, true, true);
    } else {
        if (knot_bounds) TMPC_LEAN_LAUNCH2(false, false); else TMPC_LEAN_LAUNCH2(false, true);
    }
#undef TMPC_LEAN_LAUNCH2
#undef TMPC_LEAN_LAUNCH
    return hipGetLastError();
}

// live: positive tolerances (residuals at every check, per-instance exits); knot_bounds: the input bounds depend on the knot;
// state_bounds: some enabled state bound is finite; P.ref_mode: REF_ZERO or REF_SHARED
template <int NX, int NU, int N>
hipError_t launch_lean(const AdmmParams &P, bool live, bool knot_bounds, bool state_bounds, hipStream_t stream) {
    if (P.ref_mode == REF_SHARED)
        return state_bounds ? launch_lean_v<NX, NU, N, true, REF_SHARED>(P, live, knot_bounds, stream)
                            : launch_lean_v<NX, NU, N, false, REF_SHARED>(P, live, knot_bounds, stream);
    return state_bounds ? launch_lean_v<NX, NU, N, true, REF_ZERO>(P, live, knot_bounds, stream)
                        : launch_lean_v<NX, NU, N, false, REF_ZERO>(P, live, knot_bounds, stream);
}

// ---- one variant specialised at the first solve that needs it (jit.cpp: jit_lean_for) ----
// The headline kernel for a shape the library has no lean instantiation of (cartpole at another horizon, a smaller system):
// the reference accepts any (nx, nu, N) at run time (tiny_api.cpp:21-71).  A whole entry is 24 kernels and 45-90 s of compiler;
// one variant — the (LIVE, UBK, ONE, XB, REFS) the launch in hand needs — is a few seconds, so a unit carries exactly one.
// (ST = double: the fp64-state form, precision 2 — only ever built this way)
template <int NX, int NU, int N, bool LIVE, bool UBK, bool ONE, bool XB, int REFS, class ST>
hipError_t launch_lean_exact(const AdmmParams &P, bool, bool, bool, hipStream_t stream) {
    hipLaunchKernelGGL((admm_lean_kernel<NX, NU, N, LIVE, UBK, ONE, XB, REFS, ST>), dim3((P.batch + 255) / 256), dim3(256), 0, stream, P);
    return hipGetLastError();
}
#define TMPC_DEFINE_LEAN_JIT_ENTRY(NAME, NX, NU, NN, LIVE, UBK, ONE, XB, REFS, ST)                                    \
    namespace tmpc {                                                                                                \
    const LeanEntry *lean_jit_entry() {                                                                             \
        static const LeanEntry e = {NX, NU, NN, NAME, &launch_lean_exact<NX, NU, NN, LIVE, UBK, ONE, XB, REFS, ST>}; \
        return &e;                                                                                                  \
    }                                                                                                               \
    }                                                                                                               \
    extern "C" const void *tmpc_jit_entry() { return tmpc::lean_jit_entry(); }

#define TMPC_DEFINE_LEAN_ENTRY(NX, NU, NN)                                                          \
    const LeanEntry *lean_entry_##NX##_##NU##_##NN() {                                              \
        static const LeanEntry e = {NX, NU, NN, "lean<" #NX "," #NU "," #NN ">", &launch_lean<NX, NU, NN>}; \
        return &e;                                                                                  \
    }

}  // namespace tmpc
