// Register-resident matrix-core kernel (compile-time horizon, cones + affine term) for nx=6 nu=3 N=10: the horizon of the
// reference's own rocket example (examples/rocket_landing_constraints.jl:14)
#include "mfmar_entry.hip.h"
namespace tmpc {
TMPC_DEFINE_MFMAR_ENTRY(6, 3, 10, false)   // box-only solves too: no — the quad kernel is faster there (0.73 against 0.90 ms)
}
