// stream kernel instantiations (run-time horizon) for (nx, nu) in [(12, 4)]
#include "stream_entry.hip.h"
namespace tmpc {
TMPC_DEFINE_STREAM_ENTRY(12, 4)
}
