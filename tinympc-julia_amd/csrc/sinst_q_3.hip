// 4-lane stream kernel instantiations (run-time horizon) for (nx, nu) in [(6, 1), (6, 2), (6, 3)]
#include "stream4_entry.hip.h"
namespace tmpc {
TMPC_DEFINE_STREAM4_ENTRY(6, 1)
TMPC_DEFINE_STREAM4_ENTRY(6, 2)
TMPC_DEFINE_STREAM4_ENTRY(6, 3)
}
