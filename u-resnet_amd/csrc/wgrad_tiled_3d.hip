// 3-D instantiations of the tiled weight gradient (see wgrad_tiled_kernel.h).
#include "wgrad_tiled_kernel.h"

int twgrad_dispatch_3d(const TWPlan& p, const TWgradArgs& a, hipStream_t s) {
  constexpr int MODE = 3;
  URSN_TW(8, 8) URSN_TW(16, 8) URSN_TW(8, 16) URSN_TW(16, 16) URSN_TW(8, 4) URSN_TW(1, 8)
  ursn_set_error("tiled wgrad 3d: no instantiation for %d->%d", p.cin, p.cout);
  return 3;
}
