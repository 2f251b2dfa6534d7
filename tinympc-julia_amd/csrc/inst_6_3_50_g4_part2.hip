// part 2 of inst_6_3_50_g4: the (float, state bounds true) kernels
#include "quad_entry.hip.h"
namespace tmpc {
TMPC_QUAD_PART(float, true, 6, 3, 50, 4, 470, 470, 7)
}
