// Instantiations + host side of the tiled stride-2 scatter-type convolution (see deconv_tiled_kernel.h).
#include <stdlib.h>

#include "deconv_tiled_kernel.h"

int tdeconv_dispatch_3d(const TDPlan& p, const TDeconvArgs& a, hipStream_t s) {
  constexpr int MODE = 3;
  URSN_TD(16, 8) URSN_TD(16, 16)
  ursn_set_error("tiled deconv 3d: no instantiation for %d->%d", p.ck, p.cp);
  return 3;
}

int tdeconv_dispatch_2d(const TDPlan& p, const TDeconvArgs& a, hipStream_t s) {
  constexpr int MODE = 2;
  URSN_TD(16, 8) URSN_TD(16, 16)
  ursn_set_error("tiled deconv 2d: no instantiation for %d->%d", p.ck, p.cp);
  return 3;
}

struct DBlocking { int nbk = 1, nbp = 1, pb = 0; };  // contraction blocks of 16, produced blocks of pb channels

static bool make_dplan(const ursn_conv_desc& d, ConvPass pass, TDPlan& p, DBlocking& b) {
  {
    static int off = -1;
    if (off < 0) { const char* e = getenv("URSN_DISABLE_TILED"); off = (e && e[0] == '1') ? 1 : 0; }
    if (off && d.algo != 3) return false;
  }
  // transposed conv forward, or data gradient of a k3 stride-2 conv
  if (d.in_split || d.in_mean) return false;
  if (d.pw_dy && (pass != PASS_DGRAD || d.transposed || !d.pw_w)) return false;
  const bool fwd_t = d.transposed && pass == PASS_FWD;
  const bool dgrad_s2 = !d.transposed && d.k == 3 && d.stride == 2 && pass == PASS_DGRAD;
  if (!fwd_t && !dgrad_s2) return false;
  const int ck = fwd_t ? d.cin : d.cout, cp = fwd_t ? d.cout : d.cin;
  const int ics = d.in_cstride > 0 ? d.in_cstride : d.cin, ocs = d.out_cstride > 0 ? d.out_cstride : d.cout;
  if ((ics & 3) || (ocs & 3)) return false;
  // low-res grid = the tensor being read: x for the transposed forward, dy for the stride-2 data gradient
  int lo[3];
  for (int j = 0; j < d.ndim; ++j) {
    if (fwd_t) lo[j] = d.in_sp[j];
    else {
      if (d.in_sp[j] & 1) return false;  // TF SAME pads 0 before only for even sizes
      lo[j] = d.in_sp[j] / 2;
    }
  }
  p.mode = d.ndim;
  if (d.ndim == 3) { p.Z = lo[0]; p.Y = lo[1]; p.X = lo[2]; }
  else { p.Z = lo[0]; p.Y = 1; p.X = lo[1]; }
  const int TX = p.mode == 3 ? 32 : 256, TY = p.mode == 3 ? 8 : 1;
  if (p.X < TX / 2 || p.Y < TY || p.Z < 8) return false;
  // buffer path (staging loads, output stores): a low-res plane and a high-res plane stay below the out-of-range marker
  const int lo_cs = fwd_t ? ics : ocs, hi_cs = fwd_t ? ocs : ics;
  if ((int64_t)p.Y * p.X * lo_cs * 4 >= (int64_t)0x80000000ll) return false;
  if ((int64_t)(p.mode == 3 ? 2 * p.Y : 1) * 2 * p.X * hi_cs * 4 >= (int64_t)0x80000000ll) return false;
  p.ntx = (p.X + TX - 1) / TX;
  p.nty = (p.Y + TY - 1) / TY;
  int64_t base = (int64_t)d.n * p.ntx * p.nty;
  // resident workgroups per CU: 2-D 2; 3-D 1 at 16 produced channels (256 VGPRs), 2 at 8 (<= 192 VGPRs since round 4)
  ursn_pick_zseg(base, p.Z, p.mode == 3 ? (cp <= 8 ? 2 : 1) : 2, 6, p.zseg, p.nzseg);
  p.grid = (int)((int64_t)d.n * p.nzseg * p.nty * p.ntx);
  b = DBlocking();
  if (ck % 16 || ck > 64) return false;
  if (cp == 8 || cp == 16) b.pb = cp;
  else if (cp % 16 == 0 && cp <= 32) b.pb = 16;
  else return false;
  b.nbk = ck / 16;
  b.nbp = cp / b.pb;
  if ((b.nbk > 1 || b.nbp > 1) && p.grid < 128 && d.algo != 3) return false;
  // fused shortcut term: one launch (16 contracted, 8 | 16 produced channels), 3-D, compact-enough gradient tensor
  if (d.pw_dy && (b.nbk != 1 || b.nbp != 1 || p.mode != 3 || (((d.pw_dy_cstride > 0 ? d.pw_dy_cstride : d.cout)) & 3) ||
                  (int64_t)p.Y * p.X * (d.pw_dy_cstride > 0 ? d.pw_dy_cstride : d.cout) * 4 >= (int64_t)0x80000000ll)) return false;
  p.ck = 16;
  p.cp = b.pb;
  const int PX = TX + 1, PY = TY + (p.mode == 3 ? 1 : 0);
  p.lds = (size_t)3 * (p.ck / 4) * PX * PY * 16;
  return p.lds <= 160 * 1024;
}

int tiled_deconv_blocks(const ursn_conv_desc& d, ConvPass pass) {  // launches the tiled kernel needs (0 = unsupported)
  TDPlan p;
  DBlocking b;
  return make_dplan(d, pass, p, b) ? b.nbk * b.nbp : 0;
}

int tiled_deconv_supported(const ursn_conv_desc& d, ConvPass pass) {
  TDPlan p;
  DBlocking b;
  return make_dplan(d, pass, p, b) ? 1 : 0;
}

size_t tiled_deconv_stats_scratch_doubles(const ursn_conv_desc& d) {
  TDPlan p;
  DBlocking b;
  if (!make_dplan(d, PASS_FWD, p, b)) return 0;
  return (size_t)p.grid * 2 * p.cp;
}

// stats_partial != nullptr: also finalise BN statistics of the produced tensor into mean/rstd
int launch_tiled_deconv(const ursn_conv_desc& d, ConvPass pass, const float* in, const float* w, float* out,
                        int accumulate, double* stats_partial, float eps, float* mean, float* rstd, hipStream_t s) {
  TDPlan p;
  DBlocking b;
  URSN_REQUIRE(make_dplan(d, pass, p, b), "tiled deconv: unsupported shape");
  const bool fwd_t = d.transposed != 0;
  const int ics = d.in_cstride > 0 ? d.in_cstride : d.cin, ocs = d.out_cstride > 0 ? d.out_cstride : d.cout;
  TDeconvArgs a;
  a.N = d.n; a.Z = p.Z; a.Y = p.Y; a.X = p.X;
  a.in_cs = fwd_t ? ics : ocs;
  a.out_cs = fwd_t ? ocs : ics;
  a.zseg = p.zseg; a.nzseg = p.nzseg; a.nty = p.nty; a.ntx = p.ntx;
  // W[k][produced][contracted]: transposed conv [k][cout][cin]; conv [k][cin][cout]
  a.cp_w = fwd_t ? d.cout : d.cin;
  a.ck_w = fwd_t ? d.cin : d.cout;
  a.pw_in = d.pw_dy; a.pw_w = d.pw_w; a.pw_in_cs = d.pw_dy_cstride > 0 ? d.pw_dy_cstride : d.cout; a.pw_ws = d.cout;
  URSN_REQUIRE(!d.pw_dy || !stats_partial, "tiled deconv: the fused shortcut term belongs to a plain data gradient");
  const int64_t V = (int64_t)d.n * p.Z * p.Y * p.X * (d.ndim == 3 ? 8 : 4);
  for (int bp = 0; bp < b.nbp; ++bp)
    for (int bk = 0; bk < b.nbk; ++bk) {
      const bool last = (bk == b.nbk - 1);
      a.in = in + 16 * bk;
      a.out = out + b.pb * bp;
      a.w = w + (size_t)b.pb * bp * a.ck_w + 16 * bk;
      a.accumulate = (accumulate || bk > 0) ? 1 : 0;
      a.stats_partial = (stats_partial && last) ? stats_partial : nullptr;
      URSN_TRY(p.mode == 3 ? tdeconv_dispatch_3d(p, a, s) : tdeconv_dispatch_2d(p, a, s));
      if (stats_partial && last)
        URSN_TRY(launch_bn_stats_final(stats_partial, p.grid, b.pb, b.pb, V, eps, mean + b.pb * bp, rstd + b.pb * bp, s));
    }
  if (b.nbk > 1 || b.nbp > 1) ursn_relabel_kernel("tdeconv xB");
  return 0;
}
