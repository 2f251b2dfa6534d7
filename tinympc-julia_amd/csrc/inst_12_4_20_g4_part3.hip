// part 3 of inst_12_4_20_g4: the (float, state bounds false) kernels
#include "quad_entry.hip.h"
namespace tmpc {
TMPC_QUAD_PART(float, false, 12, 4, 20, 4, 380, 470)
}
