// LDS-tiled small-channel convolution kernels (family I).  Placeholder: reports "unsupported" so every
// layer takes the generic path until the tiled kernels land.
#include "ursn_common.h"

int tiled_conv_supported(const ursn_conv_desc&, ConvPass) { return 0; }
int launch_tiled_conv(const ursn_conv_desc&, ConvPass, const float*, const float*, float*, int, hipStream_t) {
  ursn_set_error("tiled conv: not built");
  return 3;
}
int tiled_wgrad_supported(const ursn_conv_desc&) { return 0; }
size_t tiled_wgrad_scratch_bytes(const ursn_conv_desc&) { return 0; }
int launch_tiled_wgrad(const ursn_conv_desc&, const float*, const float*, float*, void*, size_t, hipStream_t) {
  ursn_set_error("tiled wgrad: not built");
  return 3;
}
