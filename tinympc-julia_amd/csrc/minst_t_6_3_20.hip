// Transposed-sets matrix-core kernel (all of a solve's state on chip; warm starts, closed loop) for nx=6 nu=3 N=20
#include "mfmat_entry.hip.h"
namespace tmpc {
TMPC_DEFINE_MFMAT_ENTRY(6, 3, 20, 0, 3, 0, 3, true)
}
