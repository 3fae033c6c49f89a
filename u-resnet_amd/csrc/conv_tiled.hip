// Host side of the LDS-tiled small-channel convolution kernels (kernels: conv_tiled_kernel.h,
// wgrad_tiled_kernel.h).
//
// Shapes the kernels are instantiated for run as one launch.  Wider layers (kernel-view channel counts that
// are multiples of 16, up to 64) run as a grid of 16x16 channel blocks: each launch reads a 16-channel slice
// of the input (channel stride = the tensor's) and accumulates into a 16-channel slice of the output, so a
// 32->16 conv is two launches, 32->32 four.  The extra read-modify-write of the output is cheap next to
// falling back to the gather kernel (25-35 TFLOP/s vs ~90).
#include <stdlib.h>

#include "conv_tiled_kernel.h"
#include "wgrad_tiled_kernel.h"

int twgradq_dispatch(const TWPlan& p, const TWgradArgs& a, hipStream_t s);
int twgrad4_dispatch_3d(const TWPlan& p, const TWgradArgs& a, hipStream_t s);
int twgrad4_dispatch_2d(const TWPlan& p, const TWgradArgs& a, hipStream_t s);
int twgradz_dispatch(const TWPlan& p, const TWgradArgs& a, hipStream_t s);

static bool tiled_shape_ok(int cin, int cout, int mode) {
  if (cin % 4 || cout % 4 || cin < 4 || cout < 4) return false;
  // instantiated shapes (weights + accumulators must fit 256 VGPRs for two waves per SIMD)
  if ((cin == 8 || cin == 16) && (cout == 8 || cout == 16)) return true;
  if (mode == 3) return (cin == 8 && cout == 4) || (cin == 4 && cout == 8);     // conv2 (8 -> 3 classes), its dgrad, conv0
  return (cin == 32 && cout == 16) || (cin == 16 && cout == 32) || (cin == 16 && cout == 4) || (cin == 4 && cout == 16) ||
         (cin == 4 && cout == 8);
}

static bool tiled_disabled() {  // URSN_DISABLE_TILED=1: route everything through the generic kernels (A/B debugging)
  static int v = -1;
  if (v < 0) { const char* e = getenv("URSN_DISABLE_TILED"); v = (e && e[0] == '1') ? 1 : 0; }
  return v == 1;
}

struct Blocking {
  int nbi = 1, nbo = 1;  // channel blocks of the contraction / produced dims (kernel view)
  int bsz = 16;          // block size when > 1 block (16; in_split for a split input)
  bool split = false;    // the blocked side is a split input: block 1 lives in x2 / dx2
};

static int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

// z-segments per (image, xy tile) for the marching kernels.  All workgroups of a launch take about the same time
// (~ zseg + prologue planes), so the launch costs rounds x (zseg + 3) with rounds = ceil(workgroups / resident slots):
// pick the split that minimises it (2.25 rounds executed as 3 was a 25 % loss on the level-0 weight gradient).
int ursn_cu_count() {
  static int n = 0;
  if (!n) {
    hipDeviceProp_t pr;
    int dev = 0;
    n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0)
            ? pr.multiProcessorCount : 256;
  }
  return n;
}
void ursn_pick_zseg(int64_t base, int Z, int occ, int min_seg, int& zseg, int& nzseg) {
  static const int legacy = getenv("URSN_ZSEG_LEGACY") ? atoi(getenv("URSN_ZSEG_LEGACY")) : 0;
  const int64_t slots = (int64_t)(occ > 0 ? occ : 1) * ursn_cu_count();
  int best_seg = Z;
  int64_t best_cost = -1;
  if (legacy) {   // powers of two up to `legacy` workgroups (A/B)
    int nz = 1;
    while (base * nz < legacy && Z / (nz * 2) >= min_seg) nz *= 2;
    best_seg = (Z + nz - 1) / nz;
  } else {
    for (int nz = 1; nz <= Z; ++nz) {
      const int seg = (Z + nz - 1) / nz;
      if (seg < min_seg && nz > 1) break;
      const int ns = (Z + seg - 1) / seg;
      const int64_t rounds = (base * ns + slots - 1) / slots;
      const int64_t cost = rounds * (seg + 3);
      if (best_cost < 0 || cost < best_cost) { best_cost = cost; best_seg = seg; }
    }
  }
  zseg = best_seg;
  nzseg = (Z + zseg - 1) / zseg;
}

// resident workgroups per CU of the 3-D tiled kernels: min(VGPR limit from the compiler's resource report, LDS limit)
static int occ_limit(int vgpr_waves, size_t lds) {
  int by_lds = lds ? (int)((160 * 1024) / lds) : 8;
  int o = vgpr_waves < by_lds ? vgpr_waves : by_lds;
  return o < 1 ? 1 : o;
}
static int tconv_vgpr_waves(int mode, int cin, int cout) {
  if (mode != 3) return 2;
  if (cin == 8 && cout == 8) return 4;
  if (cin == 16 && cout == 16) return 2;
  if (cin <= 8 && cout <= 8) return 6;
  return 3;
}
static int twgrad_vgpr_waves(int mode, int cin, int cout) {
  if (mode != 3) return 2;
  if (cin == 16) return 1;
  if (cin == 1) return 8;
  return 2;
}

static bool spatial_tiles(const ursn_conv_desc& d, int& Z, int& Y, int& X, int& ntx, int& nty) {
  if (d.ndim == 3) { Z = d.in_sp[0]; Y = d.in_sp[1]; X = d.in_sp[2]; }
  else { Z = d.in_sp[0]; Y = 1; X = d.in_sp[1]; }
  {   // buffer-path staging (buffer_stage.h): a z plane (2-D: a row) must stay below the out-of-range marker
    // ... of EVERY tensor the launch addresses that way: the second input / gradient tensor of a split input, the fused
    // shortcut gradient and the fused BatchNorm-backward operands as well as x / y
    int cs = d.in_cstride > 0 ? d.in_cstride : d.cin;
    const int ocs = d.out_cstride > 0 ? d.out_cstride : d.cout;
    if (ocs > cs) cs = ocs;
    if (d.in_split > 0) { const int c2 = d.in2_cstride > 0 ? d.in2_cstride : d.cin - d.in_split; if (c2 > cs) cs = c2; }
    if (d.pw_dy) { const int c2 = d.pw_dy_cstride > 0 ? d.pw_dy_cstride : d.cout; if (c2 > cs) cs = c2; }
    if (d.bs_partial) {
      if (d.bs_z_cstride > cs) cs = d.bs_z_cstride;
      if (d.bs_z2 && d.bs_z2_cstride > cs) cs = d.bs_z2_cstride;
    }
    if ((int64_t)Y * X * cs * 4 >= (int64_t)0x80000000ll) return false;
  }
  const int TX = d.ndim == 3 ? 32 : 256, TY = d.ndim == 3 ? 8 : 1;
  if (X < TX / 2 || Y < TY || Z < 8) return false;  // only worth it when tiles are reasonably full
  ntx = (X + TX - 1) / TX;
  nty = (Y + TY - 1) / TY;
  return true;
}

static bool make_plan(const ursn_conv_desc& d, ConvPass pass, TPlan& p, Blocking& b) {
  if (tiled_disabled() && d.algo != 3) return false;
  if (d.transposed || d.k != 3 || d.stride != 1) return false;
  if (pass != PASS_FWD && pass != PASS_DGRAD) return false;
  p.mode = d.ndim;
  p.flip = (pass == PASS_DGRAD);
  p.cin = ((p.flip ? d.cout : d.cin) + 3) & ~3;   // kernel view: channel counts padded to 4 (buffers are padded too)
  p.cout = ((p.flip ? d.cin : d.cout) + 3) & ~3;
  const int ics = d.in_cstride > 0 ? d.in_cstride : d.cin, ocs = d.out_cstride > 0 ? d.out_cstride : d.cout;
  const bool single_in = (d.cin == 1 && ics == 1 && !p.flip);   // conv0: scalar input fetch, kernel view Cin = 4
  if (((ics & 3) && !single_in) || (ocs & 3)) return false;
  if (d.cin == 1 && p.flip) return false;
  if (!spatial_tiles(d, p.Z, p.Y, p.X, p.ntx, p.nty)) return false;
  p.grid = (int)((int64_t)d.n * p.nty * p.ntx);   // x nzseg below, once the kernel shape (-> occupancy) is known
  b = Blocking();
  if (d.in_split) {
    // never-materialised concat: the two halves of the layer input are separate tensors -> two channel blocks on
    // the input side (forward: contraction blocks accumulate into y; data gradient: one produced block per tensor)
    const int half = d.in_split, i2 = d.in2_cstride > 0 ? d.in2_cstride : d.cin - d.in_split;
    if (2 * half != d.cin || (half & 3) || (i2 & 3)) return false;
    const int kc = p.flip ? p.cin : half, kp = p.flip ? half : p.cout;
    if (!tiled_shape_ok(kc, kp, p.mode)) return false;
    b.split = true; b.bsz = half;
    if (p.flip) { b.nbo = 2; p.cout = half; } else { b.nbi = 2; p.cin = half; }
  } else if (!tiled_shape_ok(p.cin, p.cout, p.mode)) {
    if ((p.cin % 16) || (p.cout % 16) || p.cin > 64 || p.cout > 64 || (p.grid * (p.Z / 12) < 192 && d.algo != 3)) return false;
    b.nbi = p.cin / 16;
    b.nbo = p.cout / 16;
    p.cin = p.cout = 16;
  }
  const int TX = p.mode == 3 ? 32 : 256, TY = p.mode == 3 ? 8 : 1;
  const int PX = TX + 2, PY = TY + (p.mode == 3 ? 2 : 0);
  p.lds = (size_t)4 * (p.cin / 4) * PX * PY * 16;
  if (p.lds > 160 * 1024) return false;
  ursn_pick_zseg(p.grid, p.Z, occ_limit(tconv_vgpr_waves(p.mode, p.cin, p.cout), p.lds), 8, p.zseg, p.nzseg);
  p.grid *= p.nzseg;
  return true;
}

int tiled_conv_supported(const ursn_conv_desc& d, ConvPass pass) {
  TPlan p;
  Blocking b;
  if (!make_plan(d, pass, p, b)) return 0;
  // fused shortcut term: the kernel contracts ALL channels of the shortcut gradient in one pass (<= 16), so the
  // contraction side must not be split into channel blocks (found with URSN_IGEMM=0: 32 -> 16 ran as two blocks and
  // dropped half of the term)
  if (d.pw_dy && !(pass == PASS_DGRAD && p.cin <= 16 && b.nbi == 1)) return 0;
  if (d.vdz_z && !(pass == PASS_DGRAD && p.mode == 3 && p.cin == 8 && p.cout == 8 && d.cout == 8 && d.cin == 8 && b.nbi == 1 && b.nbo == 1 && !b.split)) return 0;
  if (d.in_mean && pass == PASS_FWD && (b.nbi > 1 || b.nbo > 1 || p.cin != p.cout || !(p.cin == 8 || p.cin == 16))) return 0;   // normalise-on-load: 8->8 / 16->16
  // fused BatchNorm-backward reductions: 3-D data gradients producing 8 channels (per tensor of a split input) from 4 | 8
  if (d.bs_partial && !(pass == PASS_DGRAD && p.mode == 3 && p.cout == 8 && (p.cin == 8 || p.cin == 4) && b.nbi == 1 &&
                        (b.nbo == 1 || b.split))) return 0;
  return 1;
}

int tiled_conv_bs_blocks(const ursn_conv_desc& d) {
  ursn_conv_desc t = d;
  t.bs_partial = (double*)1;   // only "non-null" matters for the support check
  TPlan p;
  Blocking b;
  if (!make_plan(t, PASS_DGRAD, p, b) || !tiled_conv_supported(t, PASS_DGRAD)) return 0;
  return p.grid;
}

// doubles of scratch the fused-statistics forward needs
size_t tiled_conv_stats_scratch_doubles(const ursn_conv_desc& d) {
  TPlan p;
  Blocking b;
  if (!make_plan(d, PASS_FWD, p, b)) return 0;
  return (size_t)p.grid * 2 * p.cout;
}

static int launch_blocks(const ursn_conv_desc& d, const TPlan& p, const Blocking& b, const float* in, const float* w,
                         float* out, int accumulate, double* stats_partial, float eps, float* mean, float* rstd,
                         hipStream_t s) {
  const int ics = d.in_cstride > 0 ? d.in_cstride : d.cin, ocs = d.out_cstride > 0 ? d.out_cstride : d.cout;
  const int i2cs = d.in2_cstride > 0 ? d.in2_cstride : d.cin - d.in_split;
  if (b.split) URSN_REQUIRE(p.flip ? d.dx2 != nullptr : d.x2 != nullptr, "tiled conv: split input without x2 / dx2");
  TConvArgs a;
  a.N = d.n; a.Z = p.Z; a.Y = p.Y; a.X = p.X;
  a.in_cs = p.flip ? ocs : ics;
  a.out_cs = p.flip ? ics : ocs;
  a.zseg = p.zseg; a.nzseg = p.nzseg; a.nty = p.nty; a.ntx = p.ntx;
  a.cin_w = d.cin; a.cout_w = d.cout;
  a.pw_in = nullptr; a.pw_w = nullptr; a.pw_in_cs = d.pw_dy_cstride > 0 ? d.pw_dy_cstride : d.cout; a.pw_ws = d.cout;
  a.aff_mean = a.aff_rstd = a.aff_beta = nullptr;
  a.bs_z = a.bs_mean = a.bs_rstd = a.bs_beta = a.bs_z2 = a.bs_mean2 = a.bs_rstd2 = nullptr;
  a.bs_mask = nullptr; a.bs_partial = nullptr; a.bs_z_cs = a.bs_z2_cs = 0; a.bs_relu = 0;
  a.dz_z = nullptr; a.dz_coef = nullptr; a.dz_out = nullptr; a.dz_relu = 0;
  if (d.vdz_z) {
    URSN_REQUIRE(p.flip && d.vdz_coef && d.vdz_out && b.nbi == 1 && b.nbo == 1 && !b.split && p.mode == 3 && p.cin == 8 && p.cout == 8 && d.cout == 8,
                 "tiled conv: BatchNorm-backward apply on load (vdz_z) needs the 3-D 8 -> 8 data gradient with coefficients and an output tensor");
    a.dz_z = d.vdz_z; a.dz_coef = d.vdz_coef; a.dz_out = d.vdz_out; a.dz_relu = d.vdz_relu;
  }
  if (d.bs_partial) {
    URSN_REQUIRE(p.flip && d.bs_z && d.bs_mean && d.bs_rstd && (d.bs_relu != 1 || d.bs_beta) && (d.bs_relu != 2 || d.bs_mask) &&
                 (!d.bs_z2 || (d.bs_mean2 && d.bs_rstd2)), "tiled conv: incomplete fused BatchNorm-backward arguments");
  }
  if (d.in_mean && !p.flip) {
    URSN_REQUIRE(b.nbi == 1 && b.nbo == 1 && d.in_rstd && d.in_beta, "tiled conv: normalise-on-load needs a native 8 / 16 channel shape");
    a.aff_mean = d.in_mean; a.aff_rstd = d.in_rstd; a.aff_beta = d.in_beta;
  }
  if (d.pw_dy) URSN_REQUIRE(p.flip && d.pw_w && p.cin <= 16 && (a.pw_in_cs & 3) == 0, "tiled conv: fused pointwise term needs the data-gradient pass with <= 16 contraction channels");
  const int64_t V = (int64_t)d.n * p.Z * p.Y * p.X;
  const int real_out = p.flip ? d.cin : d.cout;
  for (int bo = 0; bo < b.nbo; ++bo)
    for (int bi = 0; bi < b.nbi; ++bi) {
      const bool last = (bi == b.nbi - 1);
      a.in = in + b.bsz * bi;
      a.out = out + b.bsz * bo;
      if (b.split) {   // block 1 of the split side is the second tensor
        if (!p.flip) { a.in = bi ? d.x2 : in; a.in_cs = bi ? i2cs : ics; }
        else { a.out = bo ? d.dx2 : out; a.out_cs = bo ? i2cs : ics; }
      }
      // W[t][ci][co]: forward contracts ci (block bi) and produces co (block bo); the data gradient contracts co
      // (block bi) and produces ci (block bo)
      a.w = p.flip ? w + (size_t)b.bsz * bo * d.cout + b.bsz * bi : w + (size_t)b.bsz * bi * d.cout + b.bsz * bo;
      a.accumulate = (accumulate || bi > 0) ? 1 : 0;
      if (d.pw_dy && bi == 0) { a.pw_in = d.pw_dy; a.pw_w = d.pw_w + (size_t)b.bsz * bo * d.cout; }   // once per produced block
      else a.pw_in = nullptr;
      a.stats_partial = (stats_partial && last) ? stats_partial : nullptr;
      a.bs_partial = nullptr;
      if (d.bs_partial && bo == 0 && last) {   // the reductions belong to the first produced tensor, once its value is final
        a.bs_partial = d.bs_partial; a.bs_z = d.bs_z; a.bs_mean = d.bs_mean; a.bs_rstd = d.bs_rstd; a.bs_beta = d.bs_beta;
        a.bs_z2 = d.bs_z2; a.bs_mean2 = d.bs_mean2; a.bs_rstd2 = d.bs_rstd2; a.bs_mask = (const unsigned long long*)d.bs_mask;
        a.bs_z_cs = d.bs_z_cstride > 0 ? d.bs_z_cstride : 8; a.bs_z2_cs = d.bs_z2_cstride > 0 ? d.bs_z2_cstride : 8;
        a.bs_relu = d.bs_relu;
      }
      URSN_TRY(p.mode == 3 ? tconv_dispatch_3d(p, a, s) : tconv_dispatch_2d(p, a, s));
      if (stats_partial && last) {
        int cb = (b.nbo > 1) ? b.bsz : real_out;  // real channels produced by this block
        URSN_TRY(launch_bn_stats_final(stats_partial, p.grid, cb, p.cout, V, eps, mean + b.bsz * bo, rstd + b.bsz * bo, s));
      }
    }
  if (d.vdz_z) {
    ursn_relabel_kernel(d.bs_partial ? (d.bs_z2 ? "tconv_dgrad<8,8>+dz+bnred2" : "tconv_dgrad<8,8>+dz+bnred") : "tconv_dgrad<8,8>+dz");
  } else if (d.bs_partial) {   // own lines in the per-kernel breakdown: the epilogue is not free
    if (b.split) ursn_relabel_kernel("tconv_dgrad x2(split)+bnred");
    else if (p.cin == 4) ursn_relabel_kernel("tconv_dgrad<4,8>+bnred");
    else ursn_relabel_kernel(d.bs_z2 ? "tconv_dgrad<8,8>+bnred2" : (d.bs_relu == 2 ? "tconv_dgrad<8,8>+bnred(mask)" : "tconv_dgrad<8,8>+bnred"));
  } else if (b.split) ursn_relabel_kernel(p.flip ? "tconv_dgrad x2(split)" : "tconv x2(split)");
  else if (b.nbi > 1 || b.nbo > 1) ursn_relabel_kernel(p.flip ? "tconv_dgrad<16,16>xB" : "tconv<16,16>xB");
  return 0;
}

// forward conv + batch statistics (mean, rstd) of its output, statistics fused into the conv epilogue
int launch_tiled_conv_bn(const ursn_conv_desc& d, const float* in, const float* w, float* out, double* scratch,
                         float eps, float* mean, float* rstd, hipStream_t s) {
  TPlan p;
  Blocking b;
  URSN_REQUIRE(make_plan(d, PASS_FWD, p, b), "tiled conv: unsupported shape");
  return launch_blocks(d, p, b, in, w, out, 0, scratch, eps, mean, rstd, s);
}

int launch_tiled_conv(const ursn_conv_desc& d, ConvPass pass, const float* in, const float* w, float* out,
                      int accumulate, hipStream_t s) {
  TPlan p;
  Blocking b;
  URSN_REQUIRE(make_plan(d, pass, p, b), "tiled conv: unsupported shape");
  return launch_blocks(d, p, b, in, w, out, accumulate, nullptr, 0.f, nullptr, nullptr, s);
}

// ---------------------------------------------------------------------------------------------------------
// weight gradient
// ---------------------------------------------------------------------------------------------------------
static bool wgrad4_enabled() {  // URSN_WGRAD4=0: keep the 16x16x4 kernel for Cout <= 8 too (A/B)
  static int v = -1;
  if (v < 0) { const char* e = getenv("URSN_WGRAD4"); v = (e && e[0] == '0') ? 0 : 1; }
  return v == 1;
}
static bool use_wgrad4(const ursn_conv_desc& d, const TWPlan& p) {
  // measured (profiles/r01): the 4x4x1 form needs one LDS operand per 8-cycle MFMA and loses to the 16x16x4 kernel
  // at Cout = 8 (37 vs 42 TFLOP/s) but wins for the 3|4-channel logits layer (1.4 vs 2.0 ms); URSN_WGRAD4=8 forces it
  static int force8 = -1;
  if (force8 < 0) { const char* e = getenv("URSN_WGRAD4"); force8 = (e && e[0] == '8') ? 1 : 0; }
  if (!wgrad4_enabled() || !(d.cin == 8 || d.cin == 16)) return false;
  if (d.ndim == 3 && d.cin == 16 && p.cout == 4) return false;
  return p.cout == 4 || (force8 && p.cout == 8);
}

static bool use_wgradz(const ursn_conv_desc& d) {  // Cout == 8: plane-pair kernel (URSN_WGRADZ=0 disables, A/B)
  static int v = -1;
  if (v < 0) { const char* e = getenv("URSN_WGRADZ"); v = (e && e[0] == '0') ? 0 : 1; }
  return v == 1 && d.cout == 8 && (d.cin == 8 || d.cin == 16);
}

// 3-D Cout == 8: 4x4-block kernel, no padded MFMA rows (wgradq_tiled_kernel.h).  Measured equal to the plane-pair kernel
// (1.24 vs 1.225 ms at 192^3 x 4), so it is opt-in: URSN_WGRADQ=1
static bool use_wgradq(const ursn_conv_desc& d) {
  static int v = -1;
  if (v < 0) { const char* e = getenv("URSN_WGRADQ"); v = (e && e[0] == '1') ? 1 : 0; }
  return v == 1 && d.ndim == 3 && use_wgradz(d);
}

static bool make_wplan(const ursn_conv_desc& d, TWPlan& p, Blocking& b) {
  if (tiled_disabled() && d.algo != 3) return false;
  if (d.transposed || d.k != 3 || d.stride != 1) return false;
  p.mode = d.ndim; p.cin = d.cin; p.cout = (d.cout + 3) & ~3;
  const int ics = d.in_cstride > 0 ? d.in_cstride : d.cin, ocs = d.out_cstride > 0 ? d.out_cstride : d.cout;
  if (((ics & 3) && !(d.cin == 1 && ics == 1)) || (ocs & 3)) return false;
  if (!spatial_tiles(d, p.Z, p.Y, p.X, p.ntx, p.nty)) return false;
  p.grid = (int)((int64_t)d.n * p.nty * p.ntx);   // x nzseg below
  b = Blocking();
  bool blocked_shape = false;
  if (d.in_mean && !(d.cin == 8 || d.cin == 16)) return false;
  if (d.in_split && !(use_wgradz(d) && d.in_split == 8 && d.cin == 16)) return false;   // split input: plane-pair kernel only
  bool c816 = (d.cin == 8 || d.cin == 16) && (p.cout == 8 || p.cout == 16);
  bool extra = d.ndim == 3 ? ((d.cin == 8 && p.cout == 4) || (d.cin == 1 && p.cout == 8))
                           : ((d.cin == 16 && p.cout == 32) || (d.cin == 16 && p.cout == 4) || (d.cin == 1 && p.cout == 16) ||
                              (d.cin == 8 && p.cout == 4));
  if (!(c816 || extra)) {
    if ((d.cin % 16) || (d.cout % 16) || d.cin > 64 || d.cout > 64 || (p.grid * (p.Z / 12) < 96 && d.algo != 3)) return false;
    b.nbi = d.cin / 16;
    b.nbo = d.cout / 16;
    p.cin = p.cout = 16;
    blocked_shape = true;
  }
  const int TX = p.mode == 3 ? 32 : 256, TY = p.mode == 3 ? 8 : 1;
  const int PX = TX + 2, PY = TY + (p.mode == 3 ? 2 : 0);
  p.lds = ((size_t)4 * PX * PY * p.cin + (size_t)2 * TX * TY * p.cout) * sizeof(float) + 256;
  int vg = twgrad_vgpr_waves(p.mode, p.cin, p.cout);
  if (use_wgradz(d)) {
    int ty = TY;
    if (p.mode == 3) {   // plane-pair kernel: 32 x 4 tiles, 4 waves (wgradz_tiled_kernel.h ZTile<3>)
      ty = 4;
      p.nty = (p.Y + ty - 1) / ty;
      p.grid = (int)((int64_t)d.n * p.nty * p.ntx);
    }
    const int py = ty + (p.mode == 3 ? 2 : 0);
    p.lds = ((size_t)6 * PX * py * 8 + (size_t)2 * TX * ty * 8) * sizeof(float) + 256;   // 6 x planes, 2 dz planes
    if (use_wgradq(d)) p.lds = ((size_t)4 * 6 * 8 * 48 + (size_t)2 * 4 * 8 * 48) * sizeof(float);   // QTile::LDS: 4 x planes, 2 dz planes, channel-major rows of 48
    // 3-D, plain input: 159 VGPRs and 46.5 KB would run THREE workgroups per CU (URSN_WGRADZ_OCC3=1).  Measured (round 4, three
    // A/B pairs on one box): the kernel alone 1.125 -> 1.08 ms per launch (0.552 -> 0.576 of the MFMA peak), but the step with the
    // weight gradients beside the main stream 58.48 -> 58.70 ms -- three resident workgroups hold 140 of the CU's 160 KB of LDS
    // and the main stream's kernels (43.5 KB per workgroup) wait for a slot.  Default: LDS padded to two workgroups per CU.
    static const bool occ3 = getenv("URSN_WGRADZ_OCC3") && getenv("URSN_WGRADZ_OCC3")[0] == '1';
    vg = (p.mode == 3 && !d.in_mean && !use_wgradq(d) && occ3) ? 3 : 2;
    // (A/B of the other direction, URSN_WGRADZ_LDSPAD=40 = ONE workgroup per CU: the kernel alone 7.0 -> 8.2 ms per step, the
    // overlapped step 58.86 -> 58.74 ms, two pairs on one box -- the main stream gains what the weight gradients lose; not the
    // default: 0.2 % against a fifth of the dominant kernel's rate when it runs alone)
    static const int pad_kb = getenv("URSN_WGRADZ_LDSPAD") ? atoi(getenv("URSN_WGRADZ_LDSPAD")) : 8;
    if (p.mode == 3 && !use_wgradq(d) && !occ3) p.lds += (size_t)pad_kb * 1024;
  } else if (!blocked_shape && use_wgrad4(d, p)) vg = 2;
  {  // the per-wave accumulator copies of the final cross-wave sum reuse the plane rings
    const int taps = p.mode == 3 ? 27 : 9;
    size_t red = use_wgradq(d) ? 0
                 : use_wgradz(d) ? (size_t)(p.mode == 3 ? 2 : 8) * 2 * taps * 64 * sizeof(float)
                                 : (size_t)4 * taps * p.cin * p.cout * sizeof(float);
    if (red > p.lds) p.lds = red;
  }
  if (p.lds > 160 * 1024) return false;
  if (d.in_mean && (blocked_shape || (!use_wgradz(d) && use_wgrad4(d, p)))) return false;   // kernels without the staging affine
  ursn_pick_zseg(p.grid, p.Z, occ_limit(vg, p.lds), 8, p.zseg, p.nzseg);
  if (use_wgradz(d) && !use_wgradq(d)) {
    static const int force_nz = getenv("URSN_WGRADZ_NZ") ? atoi(getenv("URSN_WGRADZ_NZ")) : 0;   // A/B
    if (force_nz > 0) { p.zseg = (p.Z + force_nz - 1) / force_nz; p.nzseg = (p.Z + p.zseg - 1) / p.zseg; }
    if ((p.zseg & 1) && p.nzseg > 1) { p.zseg += 1; p.nzseg = (p.Z + p.zseg - 1) / p.zseg; }   // plane pairs
  }
  p.grid *= p.nzseg;
  return true;
}

int tiled_wgrad_supported(const ursn_conv_desc& d) {
  TWPlan p;
  Blocking b;
  return make_wplan(d, p, b) ? 1 : 0;
}

size_t tiled_wgrad_scratch_bytes(const ursn_conv_desc& d) {
  TWPlan p;
  Blocking b;
  if (!make_wplan(d, p, b)) return 0;
  int taps = d.ndim == 3 ? 27 : 9;
  int ci = (b.nbi > 1 || b.nbo > 1) ? 16 : d.cin, co = (b.nbi > 1 || b.nbo > 1) ? 16 : d.cout;
  if (use_wgradq(d)) return (size_t)p.grid * taps * 8 * 8 * sizeof(float);
  if (use_wgradz(d)) return (size_t)p.grid * 2 * taps * 8 * 8 * sizeof(float);   // two slabs per workgroup
  return (size_t)p.grid * taps * ci * co * sizeof(float);   // one slab per workgroup
}

int launch_tiled_wgrad(const ursn_conv_desc& d, const float* x, const float* dy, float* dw, void* scratch,
                       size_t scratch_bytes, hipStream_t s) {
  TWPlan p;
  Blocking b;
  URSN_REQUIRE(make_wplan(d, p, b), "tiled wgrad: unsupported shape");
  size_t need = tiled_wgrad_scratch_bytes(d);
  URSN_REQUIRE(scratch && scratch_bytes >= need, "tiled wgrad: scratch too small (%zu < %zu)", scratch_bytes, need);
  const bool blocked = b.nbi > 1 || b.nbo > 1;
  TWgradArgs a;
  a.slab = (float*)scratch;
  a.N = d.n; a.Z = p.Z; a.Y = p.Y; a.X = p.X;
  a.x_cs = d.in_cstride > 0 ? d.in_cstride : d.cin;
  a.dz_cs = d.out_cstride > 0 ? d.out_cstride : d.cout;
  a.zseg = p.zseg; a.nzseg = p.nzseg; a.nty = p.nty; a.ntx = p.ntx;
  a.cout_w = blocked ? 16 : d.cout;
  a.aff_mean = d.in_mean; a.aff_rstd = d.in_rstd; a.aff_beta = d.in_beta;
  URSN_REQUIRE(!d.in_mean || (d.in_rstd && d.in_beta), "tiled wgrad: normalise-on-load needs in_rstd and in_beta");
  const int taps = d.ndim == 3 ? 27 : 9;
  if (!blocked && use_wgradz(d)) {  // Cout == 8: two output planes share each MFMA; Cin = 16 as two 8-channel slices
    a.dz = dy;
    a.cout_w = 8;
    for (int bi = 0; bi < d.cin / 8; ++bi) {
      a.x = x + 8 * bi;
      if (d.in_mean) { a.aff_mean = d.in_mean + 8 * bi; a.aff_rstd = d.in_rstd + 8 * bi; a.aff_beta = d.in_beta + 8 * bi; }
      if (d.in_split && bi) {   // second half of a split input
        URSN_REQUIRE(d.x2, "tiled wgrad: split input without x2");
        a.x = d.x2;
        a.x_cs = d.in2_cstride > 0 ? d.in2_cstride : d.cin - d.in_split;
      }
      const bool quad = use_wgradq(d);
      const int nslab = quad ? p.grid : p.grid * 2;
      URSN_TRY(quad ? twgradq_dispatch(p, a, s) : twgradz_dispatch(p, a, s));
      if (d.cin == 8) return launch_reduce_accum(dw, (const float*)scratch, (int64_t)taps * 64, nslab, s);
      URSN_TRY(launch_reduce_accum_blocked(dw + (size_t)8 * bi * 8, (const float*)scratch, taps, 8, 8, (int64_t)d.cin * 8,
                                           8, nslab, s));
    }
    return 0;
  }
  if (!blocked && use_wgrad4(d, p)) {  // Cout <= 8: 4x4x1 blocks, one slab per workgroup
    a.x = x;
    a.dz = dy;
    URSN_TRY(p.mode == 3 ? twgrad4_dispatch_3d(p, a, s) : twgrad4_dispatch_2d(p, a, s));
    return launch_reduce_accum(dw, (const float*)scratch, (int64_t)taps * d.cin * d.cout, p.grid, s);
  }
  for (int bi = 0; bi < b.nbi; ++bi)
    for (int bo = 0; bo < b.nbo; ++bo) {
      a.x = x + 16 * bi;
      a.dz = dy + 16 * bo;
      URSN_TRY(p.mode == 3 ? twgrad_dispatch_3d(p, a, s) : twgrad_dispatch_2d(p, a, s));
      if (!blocked) return launch_reduce_accum(dw, (const float*)scratch, (int64_t)taps * d.cin * d.cout, p.grid, s);
      // slab [tap][16][16] -> dw[tap][16*bi + r][16*bo + c]
      URSN_TRY(launch_reduce_accum_blocked(dw + (size_t)16 * bi * d.cout + 16 * bo, (const float*)scratch, taps, 16, 16,
                                           (int64_t)d.cin * d.cout, d.cout, p.grid, s));
    }
  ursn_relabel_kernel("twgrad<16,16>xB");
  return 0;
}
