// Instantiation of the 4x4-block weight gradient for the 3-D full-resolution Cout = 8 layers (see wgradq_tiled_kernel.h).
#include "wgradq_tiled_kernel.h"

int twgradq_dispatch(const TWPlan& p, const TWgradArgs& a, hipStream_t s) {
  ursn_note_kernel("twgradq<8,8>");
  return launch_twq(p, a, s);
}
