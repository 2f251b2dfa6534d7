// 4-lane stream kernel instantiations (run-time horizon) for (nx, nu) in [(8, 3), (8, 4), (10, 1)]
#include "stream4_entry.hip.h"
namespace tmpc {
TMPC_DEFINE_STREAM4_ENTRY(8, 3)
TMPC_DEFINE_STREAM4_ENTRY(8, 4)
TMPC_DEFINE_STREAM4_ENTRY(10, 1)
}
