// Register-resident matrix-core kernel (compile-time horizon, cones + affine term) for nx=6 nu=3 N=50: BASELINE config 4
#include "mfmar_entry.hip.h"
namespace tmpc {
TMPC_DEFINE_MFMAR_ENTRY(6, 3, 50, true)   // box-only solves too: 4.3 ms against the quad kernel's 5.5
}
