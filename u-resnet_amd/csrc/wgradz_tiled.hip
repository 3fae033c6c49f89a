// Instantiations of the plane-pair weight gradient for Cout = 8 (see wgradz_tiled_kernel.h).
#include "wgradz_tiled_kernel.h"

int twgradz_dispatch(const TWPlan& p, const TWgradArgs& a, hipStream_t s) {
  ursn_note_kernel("twgradz<8,8>");
  return p.mode == 3 ? launch_twz<3>(p, a, s) : launch_twz<2>(p, a, s);
}
