// Geometry construction (TF SAME semantics) and the op-level C-ABI.
#include <dlfcn.h>
#include <stdarg.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

#include "bf16_common.h"

static thread_local char g_err[1024] = "";
void ursn_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* ursn_last_error(void) { return g_err; }
static thread_local const char* g_kernel = "";
static long g_kernel_launches = 0;
void ursn_note_kernel(const char* name) { g_kernel = name; ++g_kernel_launches; }   // one call per main-kernel launch
void ursn_relabel_kernel(const char* name) { g_kernel = name; }
long ursn_kernel_launch_count() { return g_kernel_launches; }
extern "C" const char* ursn_last_kernel_name() { return g_kernel; }
extern "C" int ursn_abi_version(void) { return URSN_ABI_VERSION; }

// ---- roctx ranges ---------------------------------------------------------------------------------------------------
namespace {
typedef int (*roctx_push_fn)(const char*);
typedef int (*roctx_pop_fn)(void);
struct Roctx {
  roctx_push_fn push = nullptr;
  roctx_pop_fn pop = nullptr;
  Roctx() {
    const char* e = getenv("URSN_ROCTX");
    if (!(e && e[0] == '1')) return;
    void* h = dlopen("librocprofiler-sdk-roctx.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("libroctx64.so.4", RTLD_NOW | RTLD_GLOBAL);
    if (!h) { fprintf(stderr, "URSN_ROCTX=1 but no roctx library could be opened: %s\n", dlerror()); return; }
    push = (roctx_push_fn)dlsym(h, "roctxRangePushA");
    pop = (roctx_pop_fn)dlsym(h, "roctxRangePop");
    if (!push || !pop) { push = nullptr; pop = nullptr; fprintf(stderr, "URSN_ROCTX=1: roctxRangePushA / roctxRangePop not found\n"); }
  }
};
Roctx& roctx() { static Roctx r; return r; }
}  // namespace
bool ursn_roctx_on() { return roctx().push != nullptr; }
void ursn_roctx_push(const char* scope, int pass) {
  if (!roctx().push) return;
  static const char* pn[] = {"fwd", "dgrad", "wgrad", "bn_stats", "bn_act", "bn_bwd", "head"};
  char buf[160];
  snprintf(buf, sizeof(buf), "%s:%s", scope, (pass >= 0 && pass < 7) ? pn[pass] : "?");
  roctx().push(buf);
}
void ursn_roctx_pop() { if (roctx().pop) roctx().pop(); }

// ---------------------------------------------------------------------------------------------
// Spatial axes are padded to 3 with a leading unit axis for 2-D problems.
// ---------------------------------------------------------------------------------------------
struct Dims3 {
  int in[3], out[3], kd[3], pb[3];
};

static int make_dims(const ursn_conv_desc& d, Dims3& D) {
  URSN_REQUIRE(d.ndim == 2 || d.ndim == 3, "conv: ndim %d not in {2,3}", d.ndim);
  URSN_REQUIRE(d.k == 1 || d.k == 3, "conv: kernel %d not in {1,3}", d.k);
  URSN_REQUIRE(d.stride == 1 || d.stride == 2, "conv: stride %d not in {1,2}", d.stride);
  URSN_REQUIRE(!d.transposed || (d.k == 3 && d.stride == 2), "conv_transpose: only k3 s2 supported");
  URSN_REQUIRE(d.cin >= 1 && d.cout >= 1 && d.n >= 1, "conv: bad channel/batch counts");
  int lead = 3 - d.ndim;
  for (int j = 0; j < 3; ++j) {
    if (j < lead) {
      D.in[j] = D.out[j] = 1;
      D.kd[j] = 1;
      D.pb[j] = 0;
      continue;
    }
    int sz = d.in_sp[j - lead];
    URSN_REQUIRE(sz >= 1, "conv: bad spatial size");
    D.in[j] = sz;
    D.kd[j] = d.k;
    if (d.transposed) {
      D.out[j] = 2 * sz;
      D.pb[j] = 0;
    } else {
      int o = (sz + d.stride - 1) / d.stride;
      int tot = (o - 1) * d.stride + d.k - sz;
      if (tot < 0) tot = 0;
      D.out[j] = o;
      D.pb[j] = tot / 2;
    }
  }
  return 0;
}

static void geom_common(GatherGeom& g, const ursn_conv_desc& d) {
  memset(&g, 0, sizeof(g));
  g.N = d.n;
  for (int j = 0; j < 3; ++j) {
    g.so[j] = 1;
    g.si[j] = 1;
  }
}

int build_geoms(const ursn_conv_desc& d, ConvPass pass, GatherGeom* out8) {
  Dims3 D;
  if (make_dims(d, D)) return -1;
  const int ics = d.in_cstride > 0 ? d.in_cstride : d.cin;
  const int ocs = d.out_cstride > 0 ? d.out_cstride : d.cout;
  const int ntap_all = D.kd[0] * D.kd[1] * D.kd[2];
  const bool gather_type = (!d.transposed && pass == PASS_FWD) || (d.transposed && pass == PASS_DGRAD);

  if (pass == PASS_WGRAD) {
    GatherGeom& g = out8[0];
    geom_common(g, d);
    for (int j = 0; j < 3; ++j) {
      // S tensor: x (conv) or dy (transposed);  C tensor: dy (conv) or x (transposed)
      g.in_d[j] = d.transposed ? D.out[j] : D.in[j];
      g.q_d[j] = d.transposed ? D.in[j] : D.out[j];
      g.out_d[j] = g.q_d[j];
      g.si[j] = d.stride;
    }
    g.in_cs = d.transposed ? ocs : ics;
    g.out_cs = d.transposed ? ics : ocs;
    g.K = d.transposed ? d.cout : d.cin;
    g.Nn = d.transposed ? d.cin : d.cout;
    g.ntaps = ntap_all;
    int t = 0;
    for (int t0 = 0; t0 < D.kd[0]; ++t0)
      for (int t1 = 0; t1 < D.kd[1]; ++t1)
        for (int t2 = 0; t2 < D.kd[2]; ++t2, ++t) {
          g.tap_d[t][0] = t0 - D.pb[0];
          g.tap_d[t][1] = t1 - D.pb[1];
          g.tap_d[t][2] = t2 - D.pb[2];
          g.tap_w[t] = t;
        }
    g.w_tap_stride = g.K * g.Nn;
    g.w_sk = g.Nn;
    g.w_sn = 1;
    return 1;
  }

  if (gather_type) {
    // conv forward:        y[o] = sum_t x[o*s + t - pb] . W[t][ci][co]
    // transposed dgrad:    dx[i] = sum_t dy[2i + t] . Wd[t][co][ci]
    GatherGeom& g = out8[0];
    geom_common(g, d);
    const bool fwd = (pass == PASS_FWD);
    for (int j = 0; j < 3; ++j) {
      g.in_d[j] = fwd ? D.in[j] : D.out[j];
      g.out_d[j] = fwd ? D.out[j] : D.in[j];
      g.q_d[j] = g.out_d[j];
      g.si[j] = d.stride;
    }
    g.in_cs = fwd ? ics : ocs;
    g.out_cs = fwd ? ocs : ics;
    g.K = fwd ? d.cin : d.cout;
    g.Nn = fwd ? d.cout : d.cin;
    g.ntaps = ntap_all;
    int t = 0;
    for (int t0 = 0; t0 < D.kd[0]; ++t0)
      for (int t1 = 0; t1 < D.kd[1]; ++t1)
        for (int t2 = 0; t2 < D.kd[2]; ++t2, ++t) {
          g.tap_d[t][0] = t0 - D.pb[0];
          g.tap_d[t][1] = t1 - D.pb[1];
          g.tap_d[t][2] = t2 - D.pb[2];
          g.tap_w[t] = t;
        }
    g.w_tap_stride = g.K * g.Nn;  // [t][K][N] natural in both cases
    g.w_sk = g.Nn;
    g.w_sn = 1;
    return 1;
  }

  // scatter-type passes, evaluated as gathers per output-parity class:
  //   conv dgrad:         dx[p] = sum_{o,t: o*s + t - pb = p} dy[o] . W[t][ci][co]     (contract co)
  //   transposed forward: y[o]  = sum_{i,t: 2i + t = o}       x[i]  . Wd[t][co][ci]    (contract ci)
  const bool dgrad = (pass == PASS_DGRAD);
  const int s = d.stride;
  int ncls = 0;
  int npar[3];
  for (int j = 0; j < 3; ++j) npar[j] = (s == 2 && D.kd[j] > 0 && (dgrad ? D.in[j] : D.out[j]) > 1) ? 2 : 1;
  for (int c0 = 0; c0 < npar[0]; ++c0)
    for (int c1 = 0; c1 < npar[1]; ++c1)
      for (int c2 = 0; c2 < npar[2]; ++c2) {
        GatherGeom& g = out8[ncls++];
        geom_common(g, d);
        int par[3] = {c0, c1, c2};
        for (int j = 0; j < 3; ++j) {
          int tgt = dgrad ? D.in[j] : D.out[j];  // dims of the tensor being produced
          g.in_d[j] = dgrad ? D.out[j] : D.in[j];
          g.out_d[j] = tgt;
          g.so[j] = npar[j];
          g.po[j] = par[j];
          g.q_d[j] = (tgt - par[j] + npar[j] - 1) / npar[j];
          g.si[j] = 1;
        }
        g.in_cs = dgrad ? ocs : ics;
        g.out_cs = dgrad ? ics : ocs;
        g.K = dgrad ? d.cout : d.cin;
        g.Nn = dgrad ? d.cin : d.cout;
        // weights: conv W[t][ci][co] with n=ci,k=co ; transposed Wd[t][co][ci] with n=co,k=ci
        g.w_tap_stride = d.cin * d.cout;
        g.w_sk = 1;
        g.w_sn = g.K;
        int nt = 0, t = 0;
        for (int t0 = 0; t0 < D.kd[0]; ++t0)
          for (int t1 = 0; t1 < D.kd[1]; ++t1)
            for (int t2 = 0; t2 < D.kd[2]; ++t2, ++t) {
              int tt[3] = {t0, t1, t2};
              bool okt = true;
              int dd[3];
              for (int j = 0; j < 3; ++j) {
                // target position p = so*q + par ; source index = (p + pb - t) / s must be integral
                int num = par[j] + D.pb[j] - tt[j];
                if (s == 2 && npar[j] == 2) {
                  if (num & 1) { okt = false; break; }
                  dd[j] = num / 2;  // exact (num even); C++ division truncates toward zero, fine for even
                } else if (s == 2) {
                  // size-1 axis with stride 2 (only q=0, p=0): source = (pb - t)/2 if integral
                  if (num & 1) { okt = false; break; }
                  dd[j] = num / 2;
                } else {
                  dd[j] = num;
                }
              }
              if (!okt) continue;
              g.tap_d[nt][0] = dd[0];
              g.tap_d[nt][1] = dd[1];
              g.tap_d[nt][2] = dd[2];
              g.tap_w[nt] = t;
              ++nt;
            }
        g.ntaps = nt;
      }
  return ncls;
}

// ---------------------------------------------------------------------------------------------
// op-level C-ABI
// ---------------------------------------------------------------------------------------------
int conv_dispatch(const ursn_conv_desc& d, ConvPass pass, const float* in, const float* w, float* out, int accumulate,
                  hipStream_t s);  // conv_tiled.hip may override for small-C layers

static int run_gather(const ursn_conv_desc& d, ConvPass pass, const float* in, const float* w, float* out,
                      int accumulate, hipStream_t s) {
  GatherGeom g[8];
  int n = build_geoms(d, pass, g);
  if (n < 0) return 2;
  for (int i = 0; i < n; ++i) {
    g[i].accumulate = accumulate;
    if (g[i].ntaps == 0 && accumulate) continue;
    if (d.algo == 1) URSN_TRY(launch_gconv_naive(g[i], in, w, out, s));
    else URSN_TRY(launch_gconv_mfma(g[i], in, w, out, s));
  }
  return 0;
}

int tiled_conv_supported(const ursn_conv_desc& d, ConvPass pass);
int launch_tiled_conv(const ursn_conv_desc& d, ConvPass pass, const float* in, const float* w, float* out,
                      int accumulate, hipStream_t s);

// scratch of the op-level entry points for kernels that need packed weights / slabs (net-level calls pass the caller's
// workspace instead): a small library-owned device buffer, grown on demand
static float* op_scratch_floats(size_t n) {
  static float* buf = nullptr;
  static size_t cap = 0;
  if (n > cap) {
    if (buf) (void)hipFree(buf);
    buf = nullptr; cap = 0;
    if (hipMalloc((void**)&buf, n * sizeof(float)) != hipSuccess) return nullptr;
    cap = n;
  }
  return buf;
}

int conv_dispatch(const ursn_conv_desc& d, ConvPass pass, const float* in, const float* w, float* out, int accumulate,
                  hipStream_t s) {
  if (d.vdz_z) {   // BatchNorm-backward apply on load: tiled 3-D 8 -> 8 data gradient only
    URSN_REQUIRE(pass == PASS_DGRAD && d.vdz_coef && d.vdz_out && tiled_conv_supported(d, pass), "conv: BatchNorm-backward apply on load (vdz_z) not supported for this shape / pass");
    return launch_tiled_conv(d, pass, in, w, out, accumulate, s);
  }
  if (d.bs_partial) {   // fused BatchNorm-backward reductions: tiled data-gradient kernels only
    URSN_REQUIRE(pass == PASS_DGRAD && tiled_conv_supported(d, pass), "conv: fused BatchNorm-backward reductions (bs_partial) not supported for this shape / pass");
    return launch_tiled_conv(d, pass, in, w, out, accumulate, s);
  }
  if (d.in_mean && pass != PASS_DGRAD) {   // normalise-on-load: tiled kernels only (the data gradient does not read x)
    URSN_REQUIRE(d.in_rstd && d.in_beta && tiled_conv_supported(d, pass), "conv: normalise-on-load (in_mean) not supported for this shape / pass");
    return launch_tiled_conv(d, pass, in, w, out, accumulate, s);
  }
  if (d.pw_dy) {      // fused shortcut data gradient: all-taps implicit GEMM or tiled kernels
    URSN_REQUIRE(pass == PASS_DGRAD, "conv: fused pointwise term (pw_dy) applies to the data gradient only");
    if (tiled_deconv_supported(d, pass))   // stride-2 resnet_conv1 + its stride-2 shortcut (level 0 / 1 of an F = 8 network)
      return launch_tiled_deconv(d, pass, in, w, out, accumulate, nullptr, 0.f, nullptr, nullptr, s);
    if (!d.in_split && igemm_conv_supported(d, pass))
      return launch_igemm_conv(d, pass, in, w, out, accumulate, nullptr, 0.f, nullptr, nullptr, s);
    URSN_REQUIRE(tiled_conv_supported(d, pass), "conv: fused pointwise term (pw_dy) not supported for this shape");
    return launch_tiled_conv(d, pass, in, w, out, accumulate, s);
  }
  if (d.in_split) {   // never-materialised concat: only the kernels that take two input tensors
    if (pointwise_conv_supported(d, pass, accumulate))
      return launch_pointwise_conv(d, pass, in, w, out, accumulate, nullptr, 0.f, nullptr, nullptr, s);
    URSN_REQUIRE(tiled_conv_supported(d, pass), "conv: split input (in_split=%d) not supported for this shape", d.in_split);
    return launch_tiled_conv(d, pass, in, w, out, accumulate, s);
  }
  if (d.algo == 3) {
    if (tiled_deconv_supported(d, pass))
      return launch_tiled_deconv(d, pass, in, w, out, accumulate, nullptr, 0.f, nullptr, nullptr, s);
    URSN_REQUIRE(tiled_conv_supported(d, pass), "tiled conv kernel does not support this shape");
    return launch_tiled_conv(d, pass, in, w, out, accumulate, s);
  }
  if (d.algo == 4) {
    URSN_REQUIRE(igemm_conv_supported(d, pass), "igemm conv kernel does not support this shape");
    return launch_igemm_conv(d, pass, in, w, out, accumulate, nullptr, 0.f, nullptr, nullptr, s);
  }
  if (d.algo == 7) {
    URSN_REQUIRE(lds_scatter_supported(d, pass), "LDS scatter conv kernel does not support this shape");
    return launch_lds_scatter(d, pass, in, w, out, accumulate, nullptr, 0.f, nullptr, nullptr, s);
  }
  if (d.algo == 6) {
    URSN_REQUIRE(stride2_conv_supported(d, pass), "stride-2 conv kernel does not support this shape");
    return launch_stride2_conv(d, pass, in, w, out, accumulate, nullptr, 0.f, nullptr, nullptr, s);
  }
  if ((d.algo == 0 || d.algo == 5) && pointwise_conv_supported(d, pass, accumulate))
    return launch_pointwise_conv(d, pass, in, w, out, accumulate, nullptr, 0.f, nullptr, nullptr, s);
  URSN_REQUIRE(d.algo != 5, "pointwise conv kernel does not support this shape");
  if (d.algo == 0 && deep_conv_supported(d, pass)) {   // deepest levels (conv_deep.hip)
    float* sc = op_scratch_floats(deep_conv_scratch_floats(d, pass));
    URSN_REQUIRE(sc, "conv: no memory for the deep-level kernel's packed weights");
    return launch_deep_conv(d, pass, in, w, out, accumulate, sc, nullptr, 0.f, nullptr, nullptr, s);
  }
  if (d.algo == 0 && igemm_conv_supported(d, pass))
    return launch_igemm_conv(d, pass, in, w, out, accumulate, nullptr, 0.f, nullptr, nullptr, s);
  if (d.algo == 0 && tiled_conv_supported(d, pass)) return launch_tiled_conv(d, pass, in, w, out, accumulate, s);
  if (d.algo == 0 && prefer_lds_scatter(d, pass))
    return launch_lds_scatter(d, pass, in, w, out, accumulate, nullptr, 0.f, nullptr, nullptr, s);
  if (d.algo == 0 && tiled_deconv_supported(d, pass))
    return launch_tiled_deconv(d, pass, in, w, out, accumulate, nullptr, 0.f, nullptr, nullptr, s);
  if (d.algo == 0 && stride2_conv_supported(d, pass))
    return launch_stride2_conv(d, pass, in, w, out, accumulate, nullptr, 0.f, nullptr, nullptr, s);
  return run_gather(d, pass, in, w, out, accumulate, s);
}

// ---- bf16 tensors at op level (desc.dtype == 1): the two kernels of bf16_conv.hip, one launch per output-parity class.
// The packed-weight scratch of these test entry points is a small library-owned device buffer (net-level calls pack into
// the caller's workspace instead).
static bf16_t* op_wpack(size_t elems) {
  static bf16_t* buf = nullptr;
  static size_t cap = 0;
  if (elems > cap) {
    if (buf) (void)hipFree(buf);
    buf = nullptr; cap = 0;
    if (hipMalloc((void**)&buf, elems * sizeof(bf16_t)) != hipSuccess) return nullptr;
    cap = elems;
  }
  return buf;
}
// the stored weight tensor of a layer whose kernel-view channel counts are padded (cin = 1 read as an 8-channel piece)
static void stored_weight_strides(GatherGeom& g, int Kr, int Nr) { g.w_tap_stride = Kr * Nr; g.w_sk = Nr; g.w_sn = 1; }

static int bf16_conv_op(const ursn_conv_desc& d0, ConvPass pass, const void* in, const float* w, void* out, int accumulate,
                        double* stats_partial, size_t stats_bytes, float eps, float* mean, float* rstd, hipStream_t s) {
  ursn_conv_desc d = d0;
  const bool scalar_in = d.cin == 1;   // conv0 of the plan: x is one fp32 channel per voxel
  if (scalar_in) {
    URSN_REQUIRE(pass == PASS_FWD && !d.transposed && d.k == 3 && d.stride == 1 && !d.in_mean && !d.in_split,
                 "bf16 conv: the scalar fp32 input form (cin = 1) is the k3 s1 forward pass / weight gradient only");
    d.cin = 8; d.in_cstride = 8;
  }
  if (d.in_split) {
    URSN_REQUIRE(pass == PASS_DGRAD && d.in_split == 8 && d.cin == 16 && d.dx2 && d.pw_dy,
                 "bf16 conv: a split tensor is supported as the two-tensor output (dx, dx2) of the 16 -> 8 data gradient with the fused shortcut term");
    if (d.in_cstride <= 0) d.in_cstride = 8;
  }
  GatherGeom g[8];
  const int n = build_geoms(d, pass, g);
  URSN_REQUIRE(n >= 1, "bf16 conv: bad descriptor");
  int64_t V = (int64_t)d.n;
  for (int j = 0; j < 3; ++j) V *= g[0].out_d[j];
  URSN_REQUIRE(!stats_partial || n == 1 || pass == PASS_FWD, "bf16 conv: statistics only on forward passes");
  if (d.pw_dy && n == 1 && !b3conv_ok(g[0]) && bcbconv_pw_ok(g[0])) {   // fused shortcut term on the channel-block kernel (levels 1-2)
    URSN_REQUIRE(pass == PASS_DGRAD && d.pw_w && !d.in_split && !d.in_mean && !stats_partial, "bf16 conv: the fused shortcut term belongs to a plain data gradient");
    bf16_t* wp = op_wpack(bcbconv_pack_elems(g[0]));
    URSN_REQUIRE(wp, "bf16 conv: no memory for the packed weights");
    g[0].accumulate = accumulate;
    return launch_bcbconv(g[0], (const bf16_t*)in, w, 0, 0, wp, (bf16_t*)out, nullptr, s, (const bf16_t*)d.pw_dy,
                          d.pw_dy_cstride > 0 ? d.pw_dy_cstride : d.cout, d.pw_w);
  }
  if (scalar_in || d.in_mean || d.pw_dy) {   // the fused forms of the input-stationary kernel (bf16_conv3.hip)
    URSN_REQUIRE(n == 1 && b3conv_ok(g[0]), "bf16 conv: fused forms (cin = 1 / in_mean / pw_dy) need a 3-D k3 s1 layer with 8 / 16 channels");
    bf16_t* wp = op_wpack(b3conv_pack_elems());
    URSN_REQUIRE(wp, "bf16 conv: no memory for the packed weights");
    g[0].accumulate = accumulate;
    const int blocks = bconv_grid_blocks(g[0]);
    URSN_REQUIRE(!stats_partial || bconv_stats_scratch_doubles(g[0]) * sizeof(double) <= stats_bytes, "bf16 conv: statistics scratch too small");
    if (scalar_in && (stored_weight_strides(g[0], 1, d.cout), b0conv_ok(g[0])) && !accumulate) {   // the taps are the contraction (bf16_conv0.hip)
      bf16_t* wp0 = op_wpack(b0conv_pack_elems());
      URSN_REQUIRE(wp0, "bf16 conv: no memory for the packed weights");
      const int blocks0 = b0conv_grid_blocks(g[0]);
      URSN_REQUIRE(!stats_partial || (size_t)blocks0 * 32 * sizeof(double) <= stats_bytes, "bf16 conv: statistics scratch too small");
      URSN_TRY(launch_b0conv(g[0], (const float*)in, w, d.cout, wp0, (bf16_t*)out, stats_partial, s));
      if (stats_partial) URSN_TRY(launch_bn_stats_final(stats_partial, blocks0, g[0].Nn, 16, V, eps, mean, rstd, s));
      return 0;
    }
    if (scalar_in) {
      URSN_REQUIRE(g[0].K == 8, "bf16 conv: the scalar input form needs the 8-channel kernel view");
      stored_weight_strides(g[0], 1, d.cout);
      URSN_TRY(launch_b3conv(g[0], nullptr, w, 1, d.cout, wp, (bf16_t*)out, stats_partial, 0, blocks, s, nullptr, 0, nullptr, nullptr, nullptr,
                             nullptr, 0, (const float*)in));
    } else if (d.in_mean) {
      URSN_REQUIRE(pass == PASS_FWD && d.in_rstd && d.in_beta && b3conv_aff_ok(g[0]) && !d.pw_dy,
                   "bf16 conv: normalise-on-load needs a C -> C forward pass");
      B3Affine af = {d.in_mean, d.in_rstd, d.in_beta, d.in_relu};
      URSN_TRY(launch_b3conv(g[0], (const bf16_t*)in, w, 0, 0, wp, (bf16_t*)out, stats_partial, 0, blocks, s, nullptr, 0, nullptr, nullptr, &af));
    } else {
      URSN_REQUIRE(pass == PASS_DGRAD && d.pw_w && b3conv_pw_ok(g[0]), "bf16 conv: the fused shortcut term needs the data gradient of a 16 -> 8 layer");
      const int pcs = d.pw_dy_cstride > 0 ? d.pw_dy_cstride : d.cout;
      bf16_t* out2 = d.in_split ? (bf16_t*)d.dx2 : nullptr;
      URSN_REQUIRE(!out2 || !accumulate, "bf16 conv: the two-tensor output overwrites");
      URSN_TRY(launch_b3conv(g[0], (const bf16_t*)in, w, 0, 0, wp, (bf16_t*)out, nullptr, 0, 0, s, (const bf16_t*)d.pw_dy, pcs, d.pw_w, nullptr, nullptr,
                             out2, out2 ? (d.in2_cstride > 0 ? d.in2_cstride : 8) : 0));
    }
    if (stats_partial) URSN_TRY(bconv_stats_finalize(g[0], stats_partial, blocks, V, eps, mean, rstd, s));
    return 0;
  }
  bool empty = false;
  for (int i = 0; i < n; ++i) empty = empty || g[i].ntaps == 0;
  if (empty && !accumulate) {   // 1x1 stride-2 data gradient: the voxels no output reads get a zero gradient
    URSN_REQUIRE(g[0].out_cs == g[0].Nn, "bf16 conv: overwrite with empty parity classes needs a compact output tensor");
    int64_t e = (int64_t)d.n * g[0].Nn;
    for (int j = 0; j < 3; ++j) e *= g[0].out_d[j];
    URSN_HIP(hipMemsetAsync(out, 0, (size_t)e * sizeof(bf16_t), s));
  }
  if (!stats_partial && bdeconv_ok(g, n)) {   // all parity classes in one launch
    bf16_t* wp = op_wpack(bdeconv_pack_elems());
    URSN_REQUIRE(wp, "bf16 conv: no memory for the packed weights");
    return launch_bdeconv(g, n, (const bf16_t*)in, w, 0, 0, wp, (bf16_t*)out, nullptr, accumulate, s);
  }
  if (n == 1 && bs2k8_ok(g[0])) {   // stride-2 gather 8 -> 16 (forward of the stride-2 conv, data gradient of the transposed conv)
    bf16_t* wp = op_wpack(bs2k8_pack_elems());
    URSN_REQUIRE(wp, "bf16 conv: no memory for the packed weights");
    const int blocks = bs2k8_grid_blocks(g[0]);
    URSN_REQUIRE(!stats_partial || (size_t)blocks * 32 * sizeof(double) <= stats_bytes, "bf16 conv: statistics scratch too small");
    URSN_TRY(launch_bs2k8(g[0], (const bf16_t*)in, w, 0, 0, wp, (bf16_t*)out, stats_partial, accumulate, nullptr, nullptr, 0, s));
    if (stats_partial) URSN_TRY(launch_bn_stats_final(stats_partial, blocks, g[0].Nn, 16, V, eps, mean, rstd, s));
    return 0;
  }
  if (bsconv_ok(g, n)) {   // ... of the deeper levels (forward with BatchNorm moments too)
    bf16_t* wp = op_wpack(bsconv_pack_elems(g, n));
    URSN_REQUIRE(wp, "bf16 conv: no memory for the packed weights");
    URSN_REQUIRE(!stats_partial || bsconv_stats_scratch_doubles(g, n) * sizeof(double) <= stats_bytes, "bf16 conv: statistics scratch too small");
    URSN_TRY(launch_bsconv(g, n, (const bf16_t*)in, w, 0, 0, wp, (bf16_t*)out, stats_partial, accumulate, s));
    if (stats_partial) URSN_TRY(bsconv_stats_finalize(g, n, stats_partial, V, eps, mean, rstd, s));
    return 0;
  }
  for (int i = 0; i < n; ++i) {
    g[i].accumulate = accumulate;
    if (g[i].ntaps == 0) continue;
    bf16_t* wp = op_wpack(bconv_pack_elems(g[i]));
    URSN_REQUIRE(wp, "bf16 conv: unsupported geometry or no memory for the packed weights");
    // transposed forward: each parity class writes its own voxels; statistics partials are summed over the classes below
    URSN_REQUIRE(!stats_partial || bconv_stats_scratch_doubles(g[i]) * sizeof(double) * n <= stats_bytes, "bf16 conv: statistics scratch too small");
    URSN_TRY(launch_bconv(g[i], (const bf16_t*)in, w, 0, 0, wp, (bf16_t*)out, (stats_partial && n == 1) ? stats_partial : nullptr, 0, 0, s));
    if (stats_partial && n == 1) URSN_TRY(bconv_stats_finalize(g[i], stats_partial, bconv_grid_blocks(g[i]), V, eps, mean, rstd, s));
  }
  return 0;
}

extern "C" int ursn_conv_forward(const ursn_conv_desc* d, const float* x, const float* w, float* y, void* stream) {
  URSN_REQUIRE(d && x && w && y, "conv_forward: null argument");
  if (d->dtype == 1) return bf16_conv_op(*d, PASS_FWD, x, w, y, 0, nullptr, 0, 0.f, nullptr, nullptr, (hipStream_t)stream);
  return conv_dispatch(*d, PASS_FWD, x, w, y, 0, (hipStream_t)stream);
}

// conv + batch statistics of its output (mean, rstd = rsqrt(var+eps)); the tiled kernels fuse the
// statistics partials into the conv epilogue, other shapes run the separate reduction.
extern "C" int ursn_conv_forward_stats(const ursn_conv_desc* d, const float* x, const float* w, float* y, float* mean,
                                       float* rstd, float eps, void* scratch, size_t scratch_bytes, void* stream) {
  URSN_REQUIRE(d && x && w && y && mean && rstd && scratch, "conv_forward_stats: null argument");
  hipStream_t s = (hipStream_t)stream;
  if (d->dtype == 1) {
    URSN_REQUIRE(!d->transposed, "conv_forward_stats: bf16 transposed convs take their statistics at net level");
    return bf16_conv_op(*d, PASS_FWD, x, w, y, 0, (double*)scratch, scratch_bytes, eps, mean, rstd, s);
  }
  GatherGeom g[8];
  URSN_REQUIRE(build_geoms(*d, PASS_FWD, g) >= 1, "conv_forward_stats: bad descriptor");
  int64_t V = (int64_t)d->n;
  for (int j = 0; j < 3; ++j) V *= g[0].out_d[j];
  if (d->transposed) { V = (int64_t)d->n; for (int j = 0; j < d->ndim; ++j) V *= 2 * d->in_sp[j]; }
  const int ocs = d->out_cstride > 0 ? d->out_cstride : d->cout;
  if (d->in_mean)
    URSN_REQUIRE(d->in_rstd && d->in_beta && tiled_conv_supported(*d, PASS_FWD), "conv_forward_stats: normalise-on-load (in_mean) not supported for this shape");
  if (d->in_split)
    URSN_REQUIRE(pointwise_conv_supported(*d, PASS_FWD, 0) || tiled_conv_supported(*d, PASS_FWD),
                 "conv_forward_stats: split input (in_split=%d) not supported for this shape", d->in_split);
  if ((d->algo == 0 || d->algo == 5 || d->in_split) && pointwise_conv_supported(*d, PASS_FWD, 0)) {
    URSN_REQUIRE(pointwise_stats_scratch_doubles(*d) * sizeof(double) <= scratch_bytes,
                 "conv_forward_stats: scratch too small");
    return launch_pointwise_conv(*d, PASS_FWD, x, w, y, 0, (double*)scratch, eps, mean, rstd, s);
  }
  if (d->algo == 0 && deep_conv_supported(*d, PASS_FWD)) {
    URSN_REQUIRE(deep_conv_stats_scratch_doubles(*d) * sizeof(double) <= scratch_bytes, "conv_forward_stats: scratch too small");
    float* sc = op_scratch_floats(deep_conv_scratch_floats(*d, PASS_FWD));
    URSN_REQUIRE(sc, "conv_forward_stats: no memory for the deep-level kernel's packed weights");
    return launch_deep_conv(*d, PASS_FWD, x, w, y, 0, sc, (double*)scratch, eps, mean, rstd, s);
  }
  if ((d->algo == 0 || d->algo == 4) && igemm_conv_supported(*d, PASS_FWD)) {
    URSN_REQUIRE(igemm_stats_scratch_doubles(*d) * sizeof(double) <= scratch_bytes,
                 "conv_forward_stats: scratch too small");
    return launch_igemm_conv(*d, PASS_FWD, x, w, y, 0, (double*)scratch, eps, mean, rstd, s);
  }
  if ((d->algo == 0 || d->algo == 6) && stride2_conv_supported(*d, PASS_FWD)) {
    URSN_REQUIRE(stride2_stats_scratch_doubles(*d) * sizeof(double) <= scratch_bytes,
                 "conv_forward_stats: scratch too small");
    return launch_stride2_conv(*d, PASS_FWD, x, w, y, 0, (double*)scratch, eps, mean, rstd, s);
  }
  if ((d->algo == 0 && prefer_lds_scatter(*d, PASS_FWD)) || (d->algo == 7 && lds_scatter_supported(*d, PASS_FWD))) {
    URSN_REQUIRE(lds_scatter_stats_scratch_doubles(*d) * sizeof(double) <= scratch_bytes,
                 "conv_forward_stats: scratch too small");
    return launch_lds_scatter(*d, PASS_FWD, x, w, y, 0, (double*)scratch, eps, mean, rstd, s);
  }
  if ((d->algo == 0 || d->algo == 3) && tiled_deconv_supported(*d, PASS_FWD)) {
    URSN_REQUIRE(tiled_deconv_stats_scratch_doubles(*d) * sizeof(double) <= scratch_bytes,
                 "conv_forward_stats: scratch too small");
    return launch_tiled_deconv(*d, PASS_FWD, x, w, y, 0, (double*)scratch, eps, mean, rstd, s);
  }
  if ((d->algo == 0 || d->algo == 3 || d->in_split || d->in_mean) && tiled_conv_supported(*d, PASS_FWD)) {
    URSN_REQUIRE(tiled_conv_stats_scratch_doubles(*d) * sizeof(double) <= scratch_bytes,
                 "conv_forward_stats: scratch too small");
    return launch_tiled_conv_bn(*d, x, w, y, (double*)scratch, eps, mean, rstd, s);
  }
  URSN_REQUIRE(reduce_scratch_bytes(V, d->cout, 2) <= scratch_bytes, "conv_forward_stats: scratch too small");
  URSN_TRY(conv_dispatch(*d, PASS_FWD, x, w, y, 0, s));
  return launch_bn_stats(y, ocs, V, d->cout, eps, mean, rstd, scratch, s);
}

extern "C" int ursn_conv_backward_data(const ursn_conv_desc* d, const float* dy, const float* w, float* dx,
                                       int32_t accumulate, void* stream) {
  URSN_REQUIRE(d && dy && w && dx, "conv_backward_data: null argument");
  if (d->dtype == 1) return bf16_conv_op(*d, PASS_DGRAD, dy, w, dx, accumulate, nullptr, 0, 0.f, nullptr, nullptr, (hipStream_t)stream);
  return conv_dispatch(*d, PASS_DGRAD, dy, w, dx, accumulate, (hipStream_t)stream);
}

int tiled_wgrad_supported(const ursn_conv_desc& d);
size_t tiled_wgrad_scratch_bytes(const ursn_conv_desc& d);
int valu_wgrad_supported(const ursn_conv_desc& d);
size_t valu_wgrad_scratch_bytes(const ursn_conv_desc& d);
int launch_valu_wgrad(const ursn_conv_desc& d, const float* x, const float* dy, float* dw, void* scratch, size_t scratch_bytes,
                      hipStream_t s);
int launch_tiled_wgrad(const ursn_conv_desc& d, const float* x, const float* dy, float* dw, void* scratch,
                       size_t scratch_bytes, hipStream_t s);

extern "C" int32_t ursn_conv_bs_blocks(const ursn_conv_desc* d) { return d ? tiled_conv_bs_blocks(*d) : 0; }

// "Explain plan": the kernel family the dispatcher would hand this descriptor / pass to -- host logic only, no device access
// (CPU-side tests of the plan guards; pass: 0 forward, 1 data gradient, 2 weight gradient).  Follows conv_dispatch /
// wgrad_dispatch / launch_bconv / launch_bwgrad predicate by predicate.
extern "C" int ursn_conv_plan(const ursn_conv_desc* d0, int32_t pass_, char* out, size_t cap) {
  URSN_REQUIRE(d0 && out && cap > 0 && pass_ >= 0 && pass_ <= 2, "conv_plan: bad argument");
  const ConvPass pass = (ConvPass)pass_;
  ursn_conv_desc d = *d0;
  const char* name = "gather";
  GatherGeom g[8];
  if (d.dtype == 1) {
    if (d.cin == 1) { d.cin = 8; d.in_cstride = 8; }
    const int n = build_geoms(d, pass, g);
    URSN_REQUIRE(n >= 1, "conv_plan: bad descriptor");
    int first = 0;
    while (first < n - 1 && g[first].ntaps == 0) ++first;
    if (pass == PASS_WGRAD) name = (d0->cin == 1 && b0wgrad_ok(g[0])) ? "b0wgrad" : b3wgrad_ok(g[0]) ? "b3wgrad" : bdwgrad_ok(g[0]) ? "bdwgrad" : bs2k8w_ok(g[0]) ? "bs2k8w" : (bwgrad_scratch_bytes(g[0]) ? "bwgrad" : "none");
    else if (d0->cin == 1 && pass == PASS_FWD && n == 1 && b0conv_ok(g[0])) name = "b0conv";
    else if (n == 1 && bs2k8_ok(g[0])) name = "bs2k8";
    else if (bdeconv_ok(g, n)) name = "bdeconv";
    else if (bsconv_ok(g, n)) name = "bsconv";
    else if (bpw_ok(g[first])) name = "bpw";
    else if (b3conv_ok(g[first])) name = "b3conv";
    else if (bcbconv_ok(g[first])) name = "bcbconv";
    else if (bdconv_ok(g[first])) name = "bdconv";
    else name = bconv_pack_elems(g[first]) ? "bconv" : "none";
    snprintf(out, cap, "%s", name);
    return 0;
  }
  URSN_REQUIRE(build_geoms(d, pass, g) >= 1, "conv_plan: bad descriptor");
  if (pass == PASS_WGRAD) {
    if (d.in_mean || (d.in_split && !pointwise_wgrad_supported(d))) name = tiled_wgrad_supported(d) ? "twgrad" : "none";
    else if (pointwise_wgrad_supported(d)) name = "pwgrad";
    else if (deep_wgrad_supported(d)) name = "dwgrad";
    else if (igemm_wgrad_supported(d)) name = "igemm_wgrad";
    else if (valu_wgrad_supported(d)) name = "vwgrad";
    else if (tiled_wgrad_supported(d)) name = "twgrad";
    else if (stride2_wgrad_supported(d)) name = "s2wgrad";
    else name = "wgrad_mfma";
  } else {
    if (d.vdz_z || d.bs_partial || (d.in_mean && pass != PASS_DGRAD)) name = tiled_conv_supported(d, pass) ? "tconv" : "none";
    else if (d.pw_dy && tiled_deconv_supported(d, pass)) name = "tdeconv+pw";
    else if (d.pw_dy) name = (!d.in_split && igemm_conv_supported(d, pass)) ? "igemm" : (tiled_conv_supported(d, pass) ? "tconv" : "none");
    else if (d.in_split) name = pointwise_conv_supported(d, pass, 0) ? "pconv" : (tiled_conv_supported(d, pass) ? "tconv" : "none");
    else if (pointwise_conv_supported(d, pass, 0)) name = "pconv";
    else if (deep_conv_supported(d, pass)) name = "dconv";
    else if (igemm_conv_supported(d, pass)) name = "igemm";
    else if (tiled_conv_supported(d, pass)) name = "tconv";
    else if (prefer_lds_scatter(d, pass)) name = "s2scatter";
    else if (tiled_deconv_supported(d, pass)) name = "tdeconv";
    else if (stride2_conv_supported(d, pass)) name = "s2conv";
    else name = "gconv_mfma";
  }
  snprintf(out, cap, "%s", name);
  return 0;
}

extern "C" size_t ursn_conv_wgrad_scratch_bytes(const ursn_conv_desc* d) {
  if (!d) return 0;
  GatherGeom g[8];
  if (d->dtype == 1) {
    ursn_conv_desc dd = *d;
    if (dd.cin == 1) { dd.cin = 8; dd.in_cstride = 8; }
    if (build_geoms(dd, PASS_WGRAD, g) != 1) return 0;
    if (d->cin == 1 && b0wgrad_ok(g[0]) && b0wgrad_scratch_bytes(g[0]) > bwgrad_scratch_bytes(g[0])) return b0wgrad_scratch_bytes(g[0]) + 256;
    return bwgrad_scratch_bytes(g[0]) + 256;
  }
  if (build_geoms(*d, PASS_WGRAD, g) != 1) return 0;
  size_t a = wgrad_plan(g[0]).scratch_bytes;
  size_t b = tiled_wgrad_supported(*d) ? tiled_wgrad_scratch_bytes(*d) : 0;
  size_t c = igemm_wgrad_scratch_bytes(*d);
  size_t e = pointwise_wgrad_scratch_bytes(*d);
  size_t f = stride2_wgrad_scratch_bytes(*d);
  size_t v = valu_wgrad_scratch_bytes(*d);
  { const size_t dwb = deep_wgrad_scratch_bytes(*d); if (dwb > a) a = dwb; }
  if (v > a) a = v;
  if (f > a) a = f;
  if (b > a) a = b;
  if (c > a) a = c;
  if (e > a) a = e;
  return a + 256;
}

int wgrad_dispatch(const ursn_conv_desc& d, const float* x, const float* dy, float* dw, void* scratch,
                   size_t scratch_bytes, hipStream_t s) {
  if (d.in_mean) {
    URSN_REQUIRE(d.in_rstd && d.in_beta && tiled_wgrad_supported(d), "conv wgrad: normalise-on-load (in_mean) not supported for this shape");
    return launch_tiled_wgrad(d, x, dy, dw, scratch, scratch_bytes, s);
  }
  if (d.in_split) {
    if (pointwise_wgrad_supported(d)) return launch_pointwise_wgrad(d, x, dy, dw, scratch, scratch_bytes, s);
    URSN_REQUIRE(tiled_wgrad_supported(d), "conv wgrad: split input (in_split=%d) not supported for this shape", d.in_split);
    return launch_tiled_wgrad(d, x, dy, dw, scratch, scratch_bytes, s);
  }
  if ((d.algo == 0 || d.algo == 5) && pointwise_wgrad_supported(d))
    return launch_pointwise_wgrad(d, x, dy, dw, scratch, scratch_bytes, s);
  URSN_REQUIRE(d.algo != 5, "pointwise wgrad kernel does not support this shape");
  if (d.algo == 0 && deep_wgrad_supported(d))   // deepest levels: operands straight from L2 (wgrad_deep.hip)
    return launch_deep_wgrad(d, x, dy, dw, scratch, scratch_bytes, s);
  if ((d.algo == 0 || d.algo == 4) && igemm_wgrad_supported(d))
    return launch_igemm_wgrad(d, x, dy, dw, scratch, scratch_bytes, s);
  URSN_REQUIRE(d.algo != 4, "igemm wgrad kernel does not support this shape");
  if (d.algo == 0 && valu_wgrad_supported(d))   // the 8 -> 3 logits layer: vector pipe, no padding of 3 channels to MFMA columns
    return launch_valu_wgrad(d, x, dy, dw, scratch, scratch_bytes, s);
  if ((d.algo == 0 || d.algo == 3) && tiled_wgrad_supported(d))
    return launch_tiled_wgrad(d, x, dy, dw, scratch, scratch_bytes, s);
  URSN_REQUIRE(d.algo != 3, "tiled wgrad kernel does not support this shape");
  if ((d.algo == 0 || d.algo == 6) && stride2_wgrad_supported(d))
    return launch_stride2_wgrad(d, x, dy, dw, scratch, scratch_bytes, s);
  URSN_REQUIRE(d.algo != 6, "stride-2 wgrad kernel does not support this shape");
  GatherGeom g[8];
  if (build_geoms(d, PASS_WGRAD, g) != 1) return 2;
  const float* S = d.transposed ? dy : x;
  const float* C = d.transposed ? x : dy;
  if (d.algo == 1) return launch_wgrad_naive(g[0], S, C, dw, s);
  return launch_wgrad_mfma(g[0], S, C, dw, scratch, scratch_bytes, s);
}

extern "C" int ursn_conv_backward_weight(const ursn_conv_desc* d, const float* x, const float* dy, float* dw,
                                         void* scratch, size_t scratch_bytes, void* stream) {
  URSN_REQUIRE(d && x && dy && dw, "conv_backward_weight: null argument");
  if (d->dtype == 1) {
    ursn_conv_desc dd = *d;
    const bool scalar_in = dd.cin == 1;
    if (scalar_in) { dd.cin = 8; dd.in_cstride = 8; }
    GatherGeom g[8];
    URSN_REQUIRE(build_geoms(dd, PASS_WGRAD, g) == 1, "conv_backward_weight: bad descriptor");
    if (scalar_in || dd.in_mean) {   // the fused forms of the z-marching kernel (bf16_wgrad3.hip)
      URSN_REQUIRE(!dd.transposed && b3wgrad_ok(g[0]) && (!scalar_in || (b3wgrad_scalar_ok(g[0]) && !dd.in_mean)),
                   "conv_backward_weight: bf16 fused forms (cin = 1 / in_mean) need a 3-D k3 s1 layer with 8 / 16 channels");
      if (scalar_in && b0wgrad_ok(g[0])) {
        stored_weight_strides(g[0], 1, dd.cout);
        return launch_b0wgrad(g[0], x, (const bf16_t*)dy, dw, dd.cout, scratch, scratch_bytes, (hipStream_t)stream);
      }
      if (scalar_in)
        return launch_b3wgrad(g[0], nullptr, (const bf16_t*)dy, dw, 1, dd.cout, scratch, scratch_bytes, (hipStream_t)stream, nullptr, x);
      URSN_REQUIRE(dd.in_rstd && dd.in_beta, "conv_backward_weight: incomplete normalise-on-load arguments");
      B3Affine af = {dd.in_mean, dd.in_rstd, dd.in_beta, dd.in_relu};
      return launch_b3wgrad(g[0], (const bf16_t*)x, (const bf16_t*)dy, dw, 0, 0, scratch, scratch_bytes, (hipStream_t)stream, &af);
    }
    const void* S = d->transposed ? (const void*)dy : (const void*)x;
    const void* Cq = d->transposed ? (const void*)x : (const void*)dy;
    return launch_bwgrad(g[0], (const bf16_t*)S, (const bf16_t*)Cq, dw, 0, 0, scratch, scratch_bytes, (hipStream_t)stream);
  }
  return wgrad_dispatch(*d, x, dy, dw, scratch, scratch_bytes, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------
// BN / head / adam op-level wrappers
// ---------------------------------------------------------------------------------------------
extern "C" size_t ursn_bn_scratch_bytes(int64_t voxels, int32_t channels) {
  return reduce_scratch_bytes(voxels, channels, 3) + 2 * (size_t)channels * sizeof(float) + 512;
}

static float* carve_floats(void* scratch, size_t scratch_bytes, int64_t V, int C) {
  size_t off = (reduce_scratch_bytes(V, C, 3) + 255) & ~(size_t)255;
  if (off + 2 * (size_t)C * sizeof(float) > scratch_bytes) return nullptr;
  return (float*)((char*)scratch + off);
}

extern "C" int ursn_bn_forward(const float* z, const float* beta, const float* res, float* y, int64_t voxels,
                               int32_t channels, float eps, int32_t relu, float* stats_out, void* scratch,
                               size_t scratch_bytes, void* stream) {
  URSN_REQUIRE(z && beta && y && scratch, "bn_forward: null argument");
  float* ms = carve_floats(scratch, scratch_bytes, voxels, channels);
  URSN_REQUIRE(ms, "bn_forward: scratch too small");
  hipStream_t s = (hipStream_t)stream;
  URSN_TRY(launch_bn_stats(z, channels, voxels, channels, eps, ms, ms + channels, scratch, s));
  BnActArgs a;
  memset(&a, 0, sizeof(a));
  a.z = z; a.zcs = channels; a.mean = ms; a.rstd = ms + channels; a.beta = beta;
  a.res = res; a.rescs = channels; a.y = y; a.ycs = channels; a.V = voxels; a.C = channels; a.relu = relu;
  URSN_TRY(launch_bn_act(a, s));
  if (stats_out) URSN_HIP(hipMemcpyAsync(stats_out, ms, 2 * (size_t)channels * sizeof(float), hipMemcpyDeviceToDevice, s));
  return 0;
}

extern "C" int ursn_bn_backward(const float* dy, const float* y, const float* z, float* dz, float* dbeta,
                                int64_t voxels, int32_t channels, float eps, int32_t relu, void* scratch,
                                size_t scratch_bytes, void* stream) {
  URSN_REQUIRE(dy && z && dz && scratch && (!relu || y), "bn_backward: null argument");
  float* ms = carve_floats(scratch, scratch_bytes, voxels, channels);
  URSN_REQUIRE(ms, "bn_backward: scratch too small");
  hipStream_t s = (hipStream_t)stream;
  URSN_TRY(launch_bn_stats(z, channels, voxels, channels, eps, ms, ms + channels, scratch, s));
  BnBwdArgs a;
  memset(&a, 0, sizeof(a));
  a.dy = dy; a.dycs = channels; a.y = y; a.ycs = channels; a.z = z; a.zcs = channels;
  a.mean = ms; a.rstd = ms + channels; a.dz = dz; a.dzcs = channels; a.dbeta = dbeta;
  a.beta = nullptr;  // op-level API: mask from y
  a.V = voxels; a.C = channels; a.relu = relu; a.scratch = scratch;
  return launch_bn_bwd(a, s);
}

// ---- bf16 BatchNorm passes at op level ------------------------------------------------------------------------------
extern "C" size_t ursn_bn_bf16_scratch_bytes(int64_t voxels, int32_t channels) { return bbn_scratch_bytes(voxels, channels) + 256; }

extern "C" int ursn_bn_bf16_forward(const ursn_bn_bf16_desc* d, void* stream) {
  URSN_REQUIRE(d && d->z && d->mean && d->rstd && d->beta && d->y, "bn_bf16_forward: null argument");
  BBnActArgs a;
  memset(&a, 0, sizeof(a));
  const int C = d->channels;
  a.z = (const bf16_t*)d->z; a.zcs = d->z_cstride > 0 ? d->z_cstride : C; a.mean = d->mean; a.rstd = d->rstd; a.beta = d->beta;
  if (d->z2) {
    URSN_REQUIRE(d->mean2 && d->rstd2 && d->beta2, "bn_bf16_forward: the second BatchNorm needs its statistics and beta");
    a.z2 = (const bf16_t*)d->z2; a.z2cs = d->z2_cstride > 0 ? d->z2_cstride : C; a.mean2 = d->mean2; a.rstd2 = d->rstd2; a.beta2 = d->beta2;
  }
  if (d->res) { a.res = (const bf16_t*)d->res; a.rescs = d->res_cstride > 0 ? d->res_cstride : C; }
  a.y = (bf16_t*)d->y; a.ycs = d->y_cstride > 0 ? d->y_cstride : (d->cat ? 2 * C : C);
  a.V = d->voxels; a.C = C; a.relu = d->relu; a.mask_out = d->relu ? d->mask_out : nullptr; a.cat = d->cat;
  return launch_bbn_act(a, (hipStream_t)stream);
}

extern "C" int ursn_bn_bf16_backward(const ursn_bn_bf16_desc* d, void* scratch, size_t scratch_bytes, void* stream) {
  URSN_REQUIRE(d && d->dy && d->z && d->mean && d->rstd && d->dz && d->dbeta && scratch, "bn_bf16_backward: null argument");
  URSN_REQUIRE(scratch_bytes >= bbn_scratch_bytes(d->voxels, d->channels), "bn_bf16_backward: scratch too small");
  BBnBwdArgs a;
  memset(&a, 0, sizeof(a));
  const int C = d->channels;
  a.dy = (const bf16_t*)d->dy; a.dycs = d->dy_cstride > 0 ? d->dy_cstride : C;
  if (d->dy2) { a.dy2 = (const bf16_t*)d->dy2; a.dy2cs = d->dy2_cstride > 0 ? d->dy2_cstride : C; }
  a.mask = d->relu ? d->mask : nullptr;
  if (d->relu && !a.mask && d->y) { a.y = (const bf16_t*)d->y; a.ycs = d->y_cstride > 0 ? d->y_cstride : C; }
  a.z = (const bf16_t*)d->z; a.zcs = d->z_cstride > 0 ? d->z_cstride : C; a.mean = d->mean; a.rstd = d->rstd; a.beta = d->beta;
  a.dz = (bf16_t*)d->dz; a.dzcs = d->dz_cstride > 0 ? d->dz_cstride : C; a.dbeta = d->dbeta;
  if (d->z2) {
    URSN_REQUIRE(d->mean2 && d->rstd2 && d->dz2 && d->dbeta2, "bn_bf16_backward: the second BatchNorm needs statistics, dz2 and dbeta2");
    a.z2 = (const bf16_t*)d->z2; a.z2cs = d->z2_cstride > 0 ? d->z2_cstride : C; a.mean2 = d->mean2; a.rstd2 = d->rstd2;
    a.dz2 = (bf16_t*)d->dz2; a.dz2cs = d->dz2_cstride > 0 ? d->dz2_cstride : C; a.dbeta2 = d->dbeta2;
  }
  if (d->dres) { a.dres = (bf16_t*)d->dres; a.drescs = d->dres_cstride > 0 ? d->dres_cstride : C; a.dres_accumulate = d->dres_accumulate; }
  a.V = d->voxels; a.C = C; a.Cw = C; a.relu = d->relu; a.scratch = scratch;
  return launch_bbn_bwd(a, (hipStream_t)stream);
}

extern "C" int ursn_softmax_ce(const float* logits, const float* data, const float* label, const float* weight,
                               int32_t n, int64_t pix, int32_t ncls, float* softmax_out, float* dlogits,
                               float* out3, void* scratch, size_t scratch_bytes, void* stream) {
  URSN_REQUIRE(logits && scratch, "softmax_ce: null argument");
  size_t need = head_scratch_bytes(n, pix) + 64;
  URSN_REQUIRE(scratch_bytes >= need, "softmax_ce: scratch too small (%zu < %zu)", scratch_bytes, need);
  hipStream_t s = (hipStream_t)stream;
  HeadArgs a;
  memset(&a, 0, sizeof(a));
  a.z = logits; a.z_cs = ncls; a.data = data; a.data_cs = 1; a.label = label; a.weight = weight;
  a.n = n; a.pix = pix; a.ncls = ncls; a.softmax_out = softmax_out; a.dlogits = dlogits; a.ana_out = nullptr;
  a.scratch = scratch;
  a.metrics = (float*)((char*)scratch + ((head_scratch_bytes(n, pix) + 15) & ~(size_t)15));
  URSN_TRY(launch_head(a, s));
  if (out3) {
    URSN_HIP(hipMemcpyAsync(out3, a.metrics, 3 * sizeof(float), hipMemcpyDeviceToHost, s));
    URSN_HIP(hipStreamSynchronize(s));
  }
  return 0;
}

extern "C" int ursn_adam(float* p, const float* g, float* m, float* v, int64_t nelem, float lr, float b1, float b2,
                         float eps, int64_t t, void* stream) {
  URSN_REQUIRE(p && g && m && v && t >= 1, "adam: bad argument");
  double lr_t = (double)lr * sqrt(1.0 - pow((double)b2, (double)t)) / (1.0 - pow((double)b1, (double)t));
  return launch_adam(p, g, m, v, nelem, (float)lr_t, b1, b2, eps, (hipStream_t)stream);
}

extern "C" int ursn_mfma_probe(int32_t which, float* out, void* stream) {
  URSN_REQUIRE(out, "mfma_probe: null output");
  return launch_mfma_probe(which, out, (hipStream_t)stream);
}
