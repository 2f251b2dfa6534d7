// 4-lane stream kernel instantiations (run-time horizon) for (nx, nu) in [(4, 2), (4, 3), (4, 4)]
#include "stream4_entry.hip.h"
namespace tmpc {
TMPC_DEFINE_STREAM4_ENTRY(4, 2)
TMPC_DEFINE_STREAM4_ENTRY(4, 3)
TMPC_DEFINE_STREAM4_ENTRY(4, 4)
}
