// matrix-core kernel instantiation for nx=12 nu=4 N=30 (the adaptive-rho variants: minst_12_4_30_adp.hip)
#include "mfma_entry.hip.h"
namespace tmpc {
TMPC_MFMA_ADP_KERNELS(extern, 12, 4, 30)
TMPC_DEFINE_MFMA_ENTRY(12, 4, 30)
}
