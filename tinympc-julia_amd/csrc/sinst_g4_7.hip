// stream kernel instantiations (run-time horizon), 4 lanes per instance, for (nx, nu) in [(12, 1), (12, 2), (12, 3)]
#include "streamg_entry.hip.h"
namespace tmpc {
TMPC_DEFINE_STREAMG_ENTRY(12, 1, 4)
TMPC_DEFINE_STREAMG_ENTRY(12, 2, 4)
TMPC_DEFINE_STREAMG_ENTRY(12, 3, 4)
}
