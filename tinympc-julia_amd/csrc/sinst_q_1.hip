// 4-lane stream kernel instantiations (run-time horizon) for (nx, nu) in [(3, 2), (3, 3), (4, 1)]
#include "stream4_entry.hip.h"
namespace tmpc {
TMPC_DEFINE_STREAM4_ENTRY(3, 2)
TMPC_DEFINE_STREAM4_ENTRY(3, 3)
TMPC_DEFINE_STREAM4_ENTRY(4, 1)
}
