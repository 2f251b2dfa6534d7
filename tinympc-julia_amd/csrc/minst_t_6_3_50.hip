// Transposed-sets matrix-core kernel (all of a solve's state on chip; warm starts, closed loop) for nx=6 nu=3 N=50: BASELINE config 4
// (this unit: the entry and its launch code; the kernels are compiled in minst_t_6_3_50_r{0,1,2}.hip)
#include "mfmat_entry.hip.h"
namespace tmpc {
TMPC_MFMAT_KERNELS_EXTERN(6, 3, 50, 0, 3, 0, 3)
TMPC_DEFINE_MFMAT_ENTRY(6, 3, 50, 0, 3, 0, 3, true)
// config 4 (cones on both sides, shared references, constant bounds): four tiles per CU — one per SIMD — need a tile's LDS
// image within a quarter of the CU's 160 KB; it is 572 bytes inside, and one CU with three tiles costs the launch a third
static_assert(TransShape<6, 3, 50>::lds_bytes(1, 3, 3) <= 40 * 1024, "config 4: four tiles per CU");
}
