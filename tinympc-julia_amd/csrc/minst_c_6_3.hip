// LDS-resident matrix-core kernel (rolled knot loops, cones + affine term) for nx=6 nu=3: BASELINE config 4
#include "mfmac_entry.hip.h"
namespace tmpc {
TMPC_DEFINE_MFMAC_ENTRY(6, 3)
}
