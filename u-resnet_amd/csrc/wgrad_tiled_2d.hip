// 2-D instantiations of the tiled weight gradient (see wgrad_tiled_kernel.h).
#include "wgrad_tiled_kernel.h"

int twgrad_dispatch_2d(const TWPlan& p, const TWgradArgs& a, hipStream_t s) {
  constexpr int MODE = 2;
  URSN_TW(8, 8) URSN_TW(16, 8) URSN_TW(8, 16) URSN_TW(16, 16) URSN_TW(16, 32) URSN_TW(16, 4) URSN_TW(1, 16)
  ursn_set_error("tiled wgrad 2d: no instantiation for %d->%d", p.cin, p.cout);
  return 3;
}
