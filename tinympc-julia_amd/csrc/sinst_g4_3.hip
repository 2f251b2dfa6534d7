// stream kernel instantiations (run-time horizon), 4 lanes per instance, for (nx, nu) in [(6, 1), (6, 2), (6, 3)]
#include "streamg_entry.hip.h"
namespace tmpc {
TMPC_DEFINE_STREAMG_ENTRY(6, 1, 4)
TMPC_DEFINE_STREAMG_ENTRY(6, 2, 4)
TMPC_DEFINE_STREAMG_ENTRY(6, 3, 4)
}
