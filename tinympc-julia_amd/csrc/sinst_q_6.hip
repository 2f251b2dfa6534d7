// 4-lane stream kernel instantiations (run-time horizon) for (nx, nu) in [(10, 2), (10, 3), (10, 4)]
#include "stream4_entry.hip.h"
namespace tmpc {
TMPC_DEFINE_STREAM4_ENTRY(10, 2)
TMPC_DEFINE_STREAM4_ENTRY(10, 3)
TMPC_DEFINE_STREAM4_ENTRY(10, 4)
}
