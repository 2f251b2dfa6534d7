// 4-lane stream kernel instantiations (run-time horizon) for (nx, nu) in [(6, 4), (8, 1), (8, 2)]
#include "stream4_entry.hip.h"
namespace tmpc {
TMPC_DEFINE_STREAM4_ENTRY(6, 4)
TMPC_DEFINE_STREAM4_ENTRY(8, 1)
TMPC_DEFINE_STREAM4_ENTRY(8, 2)
}
