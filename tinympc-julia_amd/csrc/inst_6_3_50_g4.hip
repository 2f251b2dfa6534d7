// quad kernel instantiation for nx=6 nu=3 N=50, 4 lane(s) per instance
#include "quad_entry.hip.h"
namespace tmpc {
TMPC_QUAD_EXTERN(6, 3, 50, 4, 470, 470, 7)
TMPC_DEFINE_QUAD_ENTRY(6, 3, 50, 4, 470, 470, 7)
}
