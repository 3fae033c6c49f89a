// Host side + instantiations of the LDS-staged implicit-GEMM convolution (conv_igemm_kernel.h).
#include <stdlib.h>

#include "conv_igemm_kernel.h"

static int igemm_mode() {  // URSN_IGEMM: 0 = off, 1 = on (default)
  static int v = -1;
  if (v < 0) { const char* e = getenv("URSN_IGEMM"); v = e ? atoi(e) : 1; }
  return v;
}

static bool make_igplan(const ursn_conv_desc& d, ConvPass pass, IGPlan& p, int& kcin, int& kcout) {
  {
    static int off = -1;
    if (off < 0) { const char* e = getenv("URSN_DISABLE_TILED"); off = (e && e[0] == '1') ? 1 : 0; }
    if ((off || igemm_mode() == 0) && d.algo != 4) return false;
  }
  if (d.transposed || d.k != 3 || d.stride != 1 || d.in_split || d.in_mean) return false;
  if (pass != PASS_FWD && pass != PASS_DGRAD) return false;
  p.flip = (pass == PASS_DGRAD);
  kcin = p.flip ? d.cout : d.cin;
  kcout = p.flip ? d.cin : d.cout;
  if ((kcin % 16) || (kcout % 16)) return false;
  if (kcin <= 16 && kcout <= 16 && d.algo != 4) return false;   // register-resident weights win there
  const int ics = d.in_cstride > 0 ? d.in_cstride : d.cin, ocs = d.out_cstride > 0 ? d.out_cstride : d.cout;
  if ((ics & 3) || (ocs & 3)) return false;
  p.mode = d.ndim;
  if (d.ndim == 3) { p.Z = d.in_sp[0]; p.Y = d.in_sp[1]; p.X = d.in_sp[2]; }
  else { p.Z = 1; p.Y = d.in_sp[0]; p.X = d.in_sp[1]; }
  static const int min_x = getenv("URSN_IGEMM_MINX") ? atoi(getenv("URSN_IGEMM_MINX")) : 5;
  if (p.X < min_x && d.algo != 4) return false;  // below that the gather kernel wastes less
  static const int at = getenv("URSN_IGEMM_ALLTAPS") ? atoi(getenv("URSN_IGEMM_ALLTAPS")) : 1;
  // small deep levels (3-D): boxes of 4x4x12 / 3x6x6 voxels with a per-lane voxel table (all-taps kernel, BM = 16)
  p.var = 0;
  if (at && p.mode == 3 && p.X <= 12) p.var = (p.X <= 6) ? 2 : 1;
  const bool at32 = at >= 1 && at != 16;   // 32-wide co tiles also have the all-taps kernel
  if (!p.var && at32 && p.mode == 3 && (p.X % 16) != 0 && (p.X % 16) <= 8 && (p.X % 12) == 0) p.var = 1;   // e.g. 24: 2 x 12 instead of 16 + 8
  if (p.X < 12 && !p.var && d.algo != 4) return false;
  const int BZ = p.mode == 3 ? (p.var == 2 ? 3 : 4) : 1, BY = p.mode == 3 ? (p.var == 2 ? 6 : 4) : 16,
            BX = p.var == 2 ? 6 : (p.var == 1 ? 12 : 16);
  p.nbz = (p.Z + BZ - 1) / BZ;
  p.nby = (p.Y + BY - 1) / BY;
  p.nbx = (p.X + BX - 1) / BX;
  p.gridx = d.n * p.nbz * p.nby * p.nbx;
  // wide cout tiles reuse the staged input more, narrow ones give the small deep levels enough workgroups
  if (p.var == 2 || (p.var == 1 && !(at32 && kcout >= 32 && (int64_t)p.gridx * (kcout / 32) >= 384))) p.bm = 16;
  else if (p.var == 1) p.bm = 32;
  else if (kcout >= 64 && p.gridx >= 512) p.bm = 64;
  else if (kcout >= 32 && (int64_t)p.gridx * (kcout / 32) >= 384) p.bm = 32;
  else p.bm = 16;
  p.gridy = (kcout + p.bm - 1) / p.bm;
  const int HZ = p.mode == 3 ? BZ + 2 : 1, HY = BY + 2, HX = BX + 2;
  p.lds = ((size_t)4 * HZ * HY * HX * 4 + (size_t)2 * 16 * (p.bm + 16)) * sizeof(float);
  p.alltaps = (at && p.bm == 16) || (at >= 1 && at != 16 && p.bm == 32);   // URSN_IGEMM_ALLTAPS: 0 off, 16 only BM=16, 1 both
  if (d.pw_dy && !(p.flip && p.alltaps)) return false;   // the fused shortcut term lives in the all-taps data-gradient kernel
  if (p.alltaps) {
    {   // the all-taps kernel stages through buffer loads with 32-bit offsets inside one image (buffer_stage.h)
      const int ics = d.in_cstride > 0 ? d.in_cstride : d.cin, ocs = d.out_cstride > 0 ? d.out_cstride : d.cout;
      if ((int64_t)p.Z * p.Y * p.X * (p.flip ? ocs : ics) * 4 >= (int64_t)0x80000000ll) return false;
    }
    const int kc = p.bm == 16 ? 16 : 8;
    p.lds = ((size_t)(kc / 4) * HZ * HY * HX * 4 + (size_t)(p.mode == 3 ? 27 : 9) * kc * p.bm) * sizeof(float);
    if (d.pw_dy) {   // fused shortcut term: the box's voxels of the shortcut gradient + the KC x BM slab of its weights
      const int nv = BZ * BY * BX;
      p.lds += ((size_t)(kc / 4) * nv * 4 + (size_t)p.bm * kc) * sizeof(float);
    }
    static const int pad_kb = getenv("URSN_IGEMM_LDS_KB") ? atoi(getenv("URSN_IGEMM_LDS_KB")) : 0;   // A/B: caps the occupancy
    if (pad_kb > 0 && p.lds < (size_t)pad_kb * 1024) p.lds = (size_t)pad_kb * 1024;
  }
  return true;
}

int igemm_conv_supported(const ursn_conv_desc& d, ConvPass pass) {
  IGPlan p;
  int a, b;
  return make_igplan(d, pass, p, a, b) ? 1 : 0;
}

size_t igemm_stats_scratch_doubles(const ursn_conv_desc& d) {
  IGPlan p;
  int a, b;
  if (!make_igplan(d, PASS_FWD, p, a, b)) return 0;
  return (size_t)p.gridx * p.gridy * 2 * p.bm;
}

template <int MODE, int BM>
static int dispatch_flags(const IGPlan& p, const IGemmArgs& a, hipStream_t s) {
  if (p.flip) return launch_ig<MODE, BM, true, false>(p, a, s);
  if (a.stats_partial) return launch_ig<MODE, BM, false, true>(p, a, s);
  return launch_ig<MODE, BM, false, false>(p, a, s);
}

template <int MODE>
static int dispatch_bm(const IGPlan& p, const IGemmArgs& a, hipStream_t s) {
  if (p.bm == 64) { ursn_note_kernel(p.flip ? "igemm_dgrad<64>" : "igemm<64>"); return dispatch_flags<MODE, 64>(p, a, s); }
  if (p.bm == 32 && p.alltaps) {
    ursn_note_kernel(p.flip ? "igemm_at_dgrad<32>" : "igemm_at<32>");
    if constexpr (MODE == 3) {
      if (p.var == 1) {
        if (p.flip) return launch_ig_at<3, 32, 8, true, false, 1>(p, a, s);
        if (a.stats_partial) return launch_ig_at<3, 32, 8, false, true, 1>(p, a, s);
        return launch_ig_at<3, 32, 8, false, false, 1>(p, a, s);
      }
    }
    if (p.flip) return launch_ig_at<MODE, 32, 8, true, false>(p, a, s);
    if (a.stats_partial) return launch_ig_at<MODE, 32, 8, false, true>(p, a, s);
    return launch_ig_at<MODE, 32, 8, false, false>(p, a, s);
  }
  if (p.bm == 32) { ursn_note_kernel(p.flip ? "igemm_dgrad<32>" : "igemm<32>"); return dispatch_flags<MODE, 32>(p, a, s); }
  if (p.alltaps) {
    ursn_note_kernel(p.flip ? "igemm_at_dgrad<16>" : "igemm_at<16>");
    if constexpr (MODE == 3) {
      if (p.var == 1) {
        if (p.flip) return launch_ig_at<3, 16, 16, true, false, 1>(p, a, s);
        if (a.stats_partial) return launch_ig_at<3, 16, 16, false, true, 1>(p, a, s);
        return launch_ig_at<3, 16, 16, false, false, 1>(p, a, s);
      }
      if (p.var == 2) {
        if (p.flip) return launch_ig_at<3, 16, 16, true, false, 2>(p, a, s);
        if (a.stats_partial) return launch_ig_at<3, 16, 16, false, true, 2>(p, a, s);
        return launch_ig_at<3, 16, 16, false, false, 2>(p, a, s);
      }
    }
    if (p.flip) return launch_ig_at<MODE, 16, 16, true, false>(p, a, s);
    if (a.stats_partial) return launch_ig_at<MODE, 16, 16, false, true>(p, a, s);
    return launch_ig_at<MODE, 16, 16, false, false>(p, a, s);
  }
  ursn_note_kernel(p.flip ? "igemm_dgrad<16>" : "igemm<16>");
  return dispatch_flags<MODE, 16>(p, a, s);
}

int launch_igemm_conv(const ursn_conv_desc& d, ConvPass pass, const float* in, const float* w, float* out,
                      int accumulate, double* stats_partial, float eps, float* mean, float* rstd, hipStream_t s) {
  IGPlan p;
  int kcin, kcout;
  URSN_REQUIRE(make_igplan(d, pass, p, kcin, kcout), "igemm conv: unsupported shape");
  const int ics = d.in_cstride > 0 ? d.in_cstride : d.cin, ocs = d.out_cstride > 0 ? d.out_cstride : d.cout;
  IGemmArgs a;
  a.in = in; a.w = w; a.out = out; a.stats_partial = stats_partial;
  a.N = d.n; a.Z = p.Z; a.Y = p.Y; a.X = p.X;
  a.cin = kcin; a.cout = kcout;
  a.in_cs = p.flip ? ocs : ics;
  a.out_cs = p.flip ? ics : ocs;
  a.cin_w = d.cin; a.cout_w = d.cout;
  a.nbz = p.nbz; a.nby = p.nby; a.nbx = p.nbx;
  a.accumulate = accumulate;
  a.pw_in = nullptr; a.pw_w = nullptr; a.pw_cs = 0; a.pw_ws = 0;
  if (d.pw_dy) {
    URSN_REQUIRE(p.flip && p.alltaps && d.pw_w, "igemm conv: fused pointwise term needs the all-taps data-gradient kernel");
    a.pw_in = d.pw_dy; a.pw_w = d.pw_w;
    a.pw_cs = d.pw_dy_cstride > 0 ? d.pw_dy_cstride : d.cout;
    a.pw_ws = d.cout;   // shortcut weights [cin][cout]
    URSN_REQUIRE((a.pw_cs & 3) == 0, "igemm conv: pw_dy channel stride must be a multiple of 4");
  }
  URSN_TRY(p.mode == 3 ? dispatch_bm<3>(p, a, s) : dispatch_bm<2>(p, a, s));
  if (stats_partial) {
    const int64_t V = (int64_t)d.n * p.Z * p.Y * p.X;
    URSN_TRY(launch_bn_stats_final_blocked(stats_partial, p.gridx, kcout, p.bm, p.bm, (size_t)p.gridx * 2 * p.bm, V, eps, mean, rstd, s));
  }
  return 0;
}
