// matrix-core kernel instantiation for nx=12 nu=4 N=25 (quadrotor horizons other than the examples' 20 / BASELINE's 30: the
// shape has no lanes-per-instance kernel at these horizons, the matrix-core kernel takes its fp64 box-constrained solves)
#include "mfma_entry.hip.h"
namespace tmpc {
TMPC_DEFINE_MFMA_ENTRY(12, 4, 25)
}
