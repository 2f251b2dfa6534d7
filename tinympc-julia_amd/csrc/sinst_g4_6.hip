// stream kernel instantiations (run-time horizon), 4 lanes per instance, for (nx, nu) in [(10, 2), (10, 3), (10, 4)]
#include "streamg_entry.hip.h"
namespace tmpc {
TMPC_DEFINE_STREAMG_ENTRY(10, 2, 4)
TMPC_DEFINE_STREAMG_ENTRY(10, 3, 4)
TMPC_DEFINE_STREAMG_ENTRY(10, 4, 4)
}
