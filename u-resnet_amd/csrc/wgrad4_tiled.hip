// Instantiations of the 4x4x1-MFMA weight gradient for Cout <= 8 (see wgrad4_tiled_kernel.h).
#include "wgrad4_tiled_kernel.h"

int twgrad4_dispatch_3d(const TWPlan& p, const TWgradArgs& a, hipStream_t s) {
  constexpr int MODE = 3;
  URSN_TW4(8, 8) URSN_TW4(16, 8) URSN_TW4(8, 4)
  ursn_set_error("tiled wgrad4 3d: no instantiation for %d->%d", p.cin, p.cout);
  return 3;
}

int twgrad4_dispatch_2d(const TWPlan& p, const TWgradArgs& a, hipStream_t s) {
  constexpr int MODE = 2;
  URSN_TW4(8, 8) URSN_TW4(16, 8) URSN_TW4(8, 4) URSN_TW4(16, 4)
  ursn_set_error("tiled wgrad4 2d: no instantiation for %d->%d", p.cin, p.cout);
  return 3;
}
