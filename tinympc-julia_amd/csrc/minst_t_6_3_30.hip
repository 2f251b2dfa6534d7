// Transposed-sets matrix-core kernel (all of a solve's state on chip; warm starts, closed loop) for nx=6 nu=3 N=30
// (this unit: the entry and its launch code; the kernels are compiled in minst_t_6_3_30_r{0,1,2}.hip)
#include "mfmat_entry.hip.h"
namespace tmpc {
TMPC_MFMAT_KERNELS_EXTERN(6, 3, 30, 0, 3, 0, 3)
TMPC_DEFINE_MFMAT_ENTRY(6, 3, 30, 0, 3, 0, 3, true)
}
