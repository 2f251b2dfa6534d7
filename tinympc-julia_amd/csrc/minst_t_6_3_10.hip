// Transposed-sets matrix-core kernel for nx=6 nu=3 N=10: the horizon of the reference's own rocket example
// (examples/rocket_landing_constraints.jl:14) — its warm-started closed loop runs here; box-only solves too: 0.60 ms
// against the quad kernel's 0.73 (32 768 instances, scripts/mfmat_scan.py)
// (this unit: the entry and its launch code; the kernels are compiled in minst_t_6_3_10_r{0,1,2}.hip)
#include "mfmat_entry.hip.h"
namespace tmpc {
TMPC_MFMAT_KERNELS_EXTERN(6, 3, 10, 0, 3, 0, 3)
TMPC_DEFINE_MFMAT_ENTRY(6, 3, 10, 0, 3, 0, 3, true)
}
