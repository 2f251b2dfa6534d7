// 4-lane stream kernel instantiations (run-time horizon) for (nx, nu) in [(2, 1), (2, 2), (3, 1)]
#include "stream4_entry.hip.h"
namespace tmpc {
TMPC_DEFINE_STREAM4_ENTRY(2, 1)
TMPC_DEFINE_STREAM4_ENTRY(2, 2)
TMPC_DEFINE_STREAM4_ENTRY(3, 1)
}
