// stream kernel instantiations (run-time horizon), 4 lanes per instance, for (nx, nu) in [(12, 4)]
#include "streamg_entry.hip.h"
namespace tmpc {
TMPC_DEFINE_STREAMG_ENTRY(12, 4, 4)
}
