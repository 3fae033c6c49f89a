// 2-D instantiations of the tiled small-channel convolution (see conv_tiled_kernel.h).
#include "conv_tiled_kernel.h"

int tconv_dispatch_2d(const TPlan& p, const TConvArgs& a, hipStream_t s) {
  constexpr int MODE = 2;
  const bool flip = p.flip;
  URSN_TC(8, 8) URSN_TC(16, 8) URSN_TC(8, 16) URSN_TC(16, 16) URSN_TC(32, 16) URSN_TC(16, 32) URSN_TC(16, 4) URSN_TC(4, 16) URSN_TC(4, 8)
  ursn_set_error("tiled conv 2d: no instantiation for %d->%d", p.cin, p.cout);
  return 3;
}
