// Transposed-sets matrix-core kernels for nx=6 nu=3 N=50, reference mode REF_SHARED (entry: minst_t_6_3_50.hip)
#include "mfmat_entry.hip.h"
namespace tmpc {
TMPC_MFMAT_KERNELS(, 6, 3, 50, REF_SHARED, 0, 3, 0, 3)
}
