// stream kernel instantiations (run-time horizon), 4 lanes per instance, for (nx, nu) in [(4, 2), (4, 3), (4, 4)]
#include "streamg_entry.hip.h"
namespace tmpc {
TMPC_DEFINE_STREAMG_ENTRY(4, 2, 4)
TMPC_DEFINE_STREAMG_ENTRY(4, 3, 4)
TMPC_DEFINE_STREAMG_ENTRY(4, 4, 4)
}
