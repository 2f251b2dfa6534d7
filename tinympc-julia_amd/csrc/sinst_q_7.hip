// 4-lane stream kernel instantiations (run-time horizon) for (nx, nu) in [(12, 1), (12, 2), (12, 3)]
#include "stream4_entry.hip.h"
namespace tmpc {
TMPC_DEFINE_STREAM4_ENTRY(12, 1)
TMPC_DEFINE_STREAM4_ENTRY(12, 2)
TMPC_DEFINE_STREAM4_ENTRY(12, 3)
}
