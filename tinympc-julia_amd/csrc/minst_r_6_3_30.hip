// Register-resident matrix-core kernel (compile-time horizon, cones + affine term) for nx=6 nu=3 N=30
#include "mfmar_entry.hip.h"
namespace tmpc {
TMPC_DEFINE_MFMAR_ENTRY(6, 3, 30, true)    // box-only solves too: no quad kernel is instantiated for this horizon (stream kernel otherwise)
}
