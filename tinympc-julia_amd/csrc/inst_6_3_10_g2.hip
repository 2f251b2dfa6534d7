// quad kernel instantiation for nx=6 nu=3 N=10, 2 lane(s) per instance
#include "quad_entry.hip.h"
namespace tmpc {
TMPC_DEFINE_QUAD_ENTRY(6, 3, 10, 2)
}
