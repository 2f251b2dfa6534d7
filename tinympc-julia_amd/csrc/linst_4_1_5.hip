// lean kernel instantiation for nx=4 nu=1 N=5 (one lane per instance, one-shot solves without an active state bound)
#include "lean_entry.hip.h"
namespace tmpc {
TMPC_DEFINE_LEAN_ENTRY(4, 1, 5)
}
