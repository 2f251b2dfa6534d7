// stream kernel instantiations (run-time horizon), 4 lanes per instance, for (nx, nu) in [(2, 1), (2, 2), (3, 1)]
#include "streamg_entry.hip.h"
namespace tmpc {
TMPC_DEFINE_STREAMG_ENTRY(2, 1, 4)
TMPC_DEFINE_STREAMG_ENTRY(2, 2, 4)
TMPC_DEFINE_STREAMG_ENTRY(3, 1, 4)
}
