// quad kernel instantiation for nx=12 nu=4 N=30, 4 lane(s) per instance
#include "quad_entry.hip.h"
namespace tmpc {
TMPC_QUAD_EXTERN(12, 4, 30, 4, 380, 470)
TMPC_DEFINE_QUAD_ENTRY(12, 4, 30, 4, 380, 470)
}
