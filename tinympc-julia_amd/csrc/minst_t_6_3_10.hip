// Transposed-sets matrix-core kernel for nx=6 nu=3 N=10: the horizon of the reference's own rocket example
// (examples/rocket_landing_constraints.jl:14) — its warm-started closed loop runs here
#include "mfmat_entry.hip.h"
namespace tmpc {
TMPC_DEFINE_MFMAT_ENTRY(6, 3, 10, 0, 3, 0, 3, true)   // box-only solves too: 0.60 ms against the quad kernel's 0.73 (32 768 instances, scripts/mfmat_scan.py)
}
