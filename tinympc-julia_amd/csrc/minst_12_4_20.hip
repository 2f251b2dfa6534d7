// matrix-core kernel instantiation for nx=12 nu=4 N=20
#include "mfma_entry.hip.h"
namespace tmpc {
TMPC_DEFINE_MFMA_ENTRY(12, 4, 20)
}
