// matrix-core kernel for nx=12 nu=4 N=30: the adaptive-rho variants (entry: minst_12_4_30.hip)
#include "mfma_entry.hip.h"
namespace tmpc {
TMPC_MFMA_ADP_KERNELS(, 12, 4, 30)
}
