// Host side of the LDS-tiled small-channel convolution kernels (kernel: conv_tiled_kernel.h).
#include <stdlib.h>

#include "conv_tiled_kernel.h"
#include "wgrad_tiled_kernel.h"

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
static bool tiled_shape_ok(int cin, int cout, int mode) {
  if (cin % 4 || cout % 4 || cin < 4 || cout < 4) return false;
  // instantiated shapes (weights + accumulators must fit 256 VGPRs for two waves per SIMD)
  if ((cin == 8 || cin == 16) && (cout == 8 || cout == 16)) return true;
  if (mode == 3) return (cin == 8 && cout == 4) || (cin == 4 && cout == 8);     // conv2 (8 -> 3 classes) and its dgrad
  return (cin == 32 && cout == 16) || (cin == 16 && cout == 32) || (cin == 16 && cout == 4) || (cin == 4 && cout == 16);
}

static bool tiled_disabled() {  // URSN_DISABLE_TILED=1: route everything through the generic kernels (A/B debugging)
  static int v = -1;
  if (v < 0) { const char* e = getenv("URSN_DISABLE_TILED"); v = (e && e[0] == '1') ? 1 : 0; }
  return v == 1;
}

static bool make_plan(const ursn_conv_desc& d, ConvPass pass, TPlan& p) {
  if (tiled_disabled() && d.algo != 3) return false;
  if (d.transposed || d.k != 3 || d.stride != 1) return false;
  if (pass != PASS_FWD && pass != PASS_DGRAD) return false;
  p.mode = d.ndim;
  p.flip = (pass == PASS_DGRAD);
  p.cin = ((p.flip ? d.cout : d.cin) + 3) & ~3;   // kernel view: channel counts padded to 4 (buffers are padded too)
  p.cout = ((p.flip ? d.cin : d.cout) + 3) & ~3;
  if (!tiled_shape_ok(p.cin, p.cout, p.mode)) return false;
  const int ics = d.in_cstride > 0 ? d.in_cstride : d.cin, ocs = d.out_cstride > 0 ? d.out_cstride : d.cout;
  if ((ics & 3) || (ocs & 3)) return false;
  if (d.ndim == 3) { p.Z = d.in_sp[0]; p.Y = d.in_sp[1]; p.X = d.in_sp[2]; }
  else { p.Z = d.in_sp[0]; p.Y = 1; p.X = d.in_sp[1]; }
  const int TX = p.mode == 3 ? 32 : 256, TY = p.mode == 3 ? 8 : 1;
  // only worth it when tiles are reasonably full
  if (p.X < TX / 2 || p.Y < TY || p.Z < 8) return false;
  p.ntx = (p.X + TX - 1) / TX;
  p.nty = (p.Y + TY - 1) / TY;
  // enough workgroups to fill 256 CUs x 2, but long marches to amortise the prologue and weight load
  int64_t base = (int64_t)d.n * p.ntx * p.nty;
  int nz = 1;
  while (base * nz < 1024 && p.Z / (nz * 2) >= 8) nz *= 2;
  p.zseg = (p.Z + nz - 1) / nz;
  p.nzseg = (p.Z + p.zseg - 1) / p.zseg;
  const int PX = TX + 2, PY = TY + (p.mode == 3 ? 2 : 0);
  p.lds = (size_t)4 * (p.cin / 4) * PX * PY * 16;
  if (p.lds > 160 * 1024) return false;
  p.grid = (int)((int64_t)d.n * p.nzseg * p.nty * p.ntx);
  return true;
}

int tiled_conv_supported(const ursn_conv_desc& d, ConvPass pass) {
  TPlan p;
  return make_plan(d, pass, p) ? 1 : 0;
}

int tiled_conv_stats_blocks(const ursn_conv_desc& d) {
  TPlan p;
  return make_plan(d, PASS_FWD, p) ? p.grid : 0;
}

int launch_tiled_conv_stats(const ursn_conv_desc& d, ConvPass pass, const float* in, const float* w, float* out,
                            int accumulate, double* stats_partial, hipStream_t s) {
  TPlan p;
  URSN_REQUIRE(make_plan(d, pass, p), "tiled conv: unsupported shape");
  TConvArgs a;
  a.in = in; a.w = w; a.out = out; a.stats_partial = stats_partial;
  a.N = d.n; a.Z = p.Z; a.Y = p.Y; a.X = p.X;
  const int ics = d.in_cstride > 0 ? d.in_cstride : d.cin, ocs = d.out_cstride > 0 ? d.out_cstride : d.cout;
  a.in_cs = p.flip ? ocs : ics;
  a.out_cs = p.flip ? ics : ocs;
  a.zseg = p.zseg; a.nzseg = p.nzseg; a.nty = p.nty; a.ntx = p.ntx;
  a.accumulate = accumulate;
  a.cin_w = d.cin; a.cout_w = d.cout;
  return p.mode == 3 ? tconv_dispatch_3d(p, a, s) : tconv_dispatch_2d(p, a, s);
}

int launch_tiled_conv(const ursn_conv_desc& d, ConvPass pass, const float* in, const float* w, float* out,
                      int accumulate, hipStream_t s) {
  return launch_tiled_conv_stats(d, pass, in, w, out, accumulate, nullptr, s);
}

// ---------------------------------------------------------------------------------------------------------
// weight gradient
// ---------------------------------------------------------------------------------------------------------
static bool make_wplan(const ursn_conv_desc& d, TWPlan& p) {
  if (tiled_disabled() && d.algo != 3) return false;
  if (d.transposed || d.k != 3 || d.stride != 1) return false;
  p.mode = d.ndim; p.cin = d.cin; p.cout = (d.cout + 3) & ~3;
  bool c816 = (d.cin == 8 || d.cin == 16) && (p.cout == 8 || p.cout == 16);
  bool extra = d.ndim == 3 ? (d.cin == 8 && p.cout == 4)
                           : ((d.cin == 16 && p.cout == 32) || (d.cin == 16 && p.cout == 4));
  if (!(c816 || extra)) return false;
  const int ics = d.in_cstride > 0 ? d.in_cstride : d.cin, ocs = d.out_cstride > 0 ? d.out_cstride : d.cout;
  if ((ics & 3) || (ocs & 3)) return false;
  if (d.ndim == 3) { p.Z = d.in_sp[0]; p.Y = d.in_sp[1]; p.X = d.in_sp[2]; }
  else { p.Z = d.in_sp[0]; p.Y = 1; p.X = d.in_sp[1]; }
  const int TX = p.mode == 3 ? 32 : 256, TY = p.mode == 3 ? 8 : 1;
  if (p.X < TX / 2 || p.Y < TY || p.Z < 8) return false;
  p.ntx = (p.X + TX - 1) / TX;
  p.nty = (p.Y + TY - 1) / TY;
  int64_t base = (int64_t)d.n * p.ntx * p.nty;
  int nz = 1;
  while (base * nz < 512 && p.Z / (nz * 2) >= 8) nz *= 2;
  p.zseg = (p.Z + nz - 1) / nz;
  p.nzseg = (p.Z + p.zseg - 1) / p.zseg;
  const int PX = TX + 2, PY = TY + (p.mode == 3 ? 2 : 0);
  p.lds = ((size_t)4 * PX * PY * d.cin + (size_t)2 * TX * TY * p.cout) * sizeof(float) + 256;
  if (p.lds > 160 * 1024) return false;
  p.grid = (int)((int64_t)d.n * p.nzseg * p.nty * p.ntx);
  return true;
}

int tiled_wgrad_supported(const ursn_conv_desc& d) {
  TWPlan p;
  return make_wplan(d, p) ? 1 : 0;
}

size_t tiled_wgrad_scratch_bytes(const ursn_conv_desc& d) {
  TWPlan p;
  if (!make_wplan(d, p)) return 0;
  int taps = d.ndim == 3 ? 27 : 9;
  return (size_t)p.grid * 4 * taps * d.cin * d.cout * sizeof(float);
}

int launch_tiled_wgrad(const ursn_conv_desc& d, const float* x, const float* dy, float* dw, void* scratch,
                       size_t scratch_bytes, hipStream_t s) {
  TWPlan p;
  URSN_REQUIRE(make_wplan(d, p), "tiled wgrad: unsupported shape");
  size_t need = tiled_wgrad_scratch_bytes(d);
  URSN_REQUIRE(scratch && scratch_bytes >= need, "tiled wgrad: scratch too small (%zu < %zu)", scratch_bytes, need);
  TWgradArgs a;
  a.x = x; a.dz = dy; a.slab = (float*)scratch;
  a.N = d.n; a.Z = p.Z; a.Y = p.Y; a.X = p.X;
  a.x_cs = d.in_cstride > 0 ? d.in_cstride : d.cin;
  a.dz_cs = d.out_cstride > 0 ? d.out_cstride : d.cout;
  a.zseg = p.zseg; a.nzseg = p.nzseg; a.nty = p.nty; a.ntx = p.ntx;
  a.cout_w = d.cout;
  URSN_TRY(p.mode == 3 ? twgrad_dispatch_3d(p, a, s) : twgrad_dispatch_2d(p, a, s));
  int taps = d.ndim == 3 ? 27 : 9;
  return launch_reduce_accum(dw, (const float*)scratch, (int64_t)taps * d.cin * d.cout, p.grid * 4, s);
}
