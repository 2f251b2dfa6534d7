// 4-lane stream kernel instantiations (run-time horizon) for (nx, nu) in [(12, 4)]
#include "stream4_entry.hip.h"
namespace tmpc {
TMPC_DEFINE_STREAM4_ENTRY(12, 4)
}
