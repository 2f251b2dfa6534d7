// part 1 of inst_6_3_50_g4: the (double, state bounds false) kernels
#include "quad_entry.hip.h"
namespace tmpc {
TMPC_QUAD_PART(double, false, 6, 3, 50, 4, 470, 470, 7)
}
