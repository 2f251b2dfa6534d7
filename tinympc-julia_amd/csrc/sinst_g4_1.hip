// stream kernel instantiations (run-time horizon), 4 lanes per instance, for (nx, nu) in [(3, 2), (3, 3), (4, 1)]
#include "streamg_entry.hip.h"
namespace tmpc {
TMPC_DEFINE_STREAMG_ENTRY(3, 2, 4)
TMPC_DEFINE_STREAMG_ENTRY(3, 3, 4)
TMPC_DEFINE_STREAMG_ENTRY(4, 1, 4)
}
