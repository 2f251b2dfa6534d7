// part 0 of inst_12_4_30_g4: the (double, state bounds true) kernels
#include "quad_entry.hip.h"
namespace tmpc {
TMPC_QUAD_PART(double, true, 12, 4, 30, 4, 380, 470)
}
