// LDS-resident matrix-core kernel (rolled knot loops; cones, linear rows, affine term) for nx=6 nu=4: four input rows take two
// input cones (a cone has at least two rows), and the affine term's constants are added on the VALU (no spare K index)
#include "mfmac_entry.hip.h"
namespace tmpc {
TMPC_DEFINE_MFMAC_ENTRY(6, 4)
}
