// stream kernel instantiations (run-time horizon), 4 lanes per instance, for (nx, nu) in [(8, 3), (8, 4), (10, 1)]
#include "streamg_entry.hip.h"
namespace tmpc {
TMPC_DEFINE_STREAMG_ENTRY(8, 3, 4)
TMPC_DEFINE_STREAMG_ENTRY(8, 4, 4)
TMPC_DEFINE_STREAMG_ENTRY(10, 1, 4)
}
