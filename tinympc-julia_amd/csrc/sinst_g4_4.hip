// stream kernel instantiations (run-time horizon), 4 lanes per instance, for (nx, nu) in [(6, 4), (8, 1), (8, 2)]
#include "streamg_entry.hip.h"
namespace tmpc {
TMPC_DEFINE_STREAMG_ENTRY(6, 4, 4)
TMPC_DEFINE_STREAMG_ENTRY(8, 1, 4)
TMPC_DEFINE_STREAMG_ENTRY(8, 2, 4)
}
