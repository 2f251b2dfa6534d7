// Transposed-sets matrix-core kernels for nx=6 nu=3 N=20, reference mode REF_ZERO (entry: minst_t_6_3_20.hip)
#include "mfmat_entry.hip.h"
namespace tmpc {
TMPC_MFMAT_KERNELS(, 6, 3, 20, REF_ZERO, 0, 3, 0, 3)
}
