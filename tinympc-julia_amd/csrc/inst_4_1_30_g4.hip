// quad kernel instantiation for nx=4 nu=1 N=30, 4 lane(s) per instance
#include "quad_entry.hip.h"
namespace tmpc {
TMPC_DEFINE_QUAD_ENTRY(4, 1, 30, 4)
}
