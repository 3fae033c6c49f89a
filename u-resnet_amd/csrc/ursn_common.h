// Internal declarations shared by the HIP translation units of liburesnet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/uresnet_hip.h"

void ursn_set_error(const char* fmt, ...);
void ursn_note_kernel(const char* name);  // remembered per thread for the profiling log
void ursn_relabel_kernel(const char* name);
#ifdef __HIPCC__
// Workgroup i runs on XCD i % 8 (each XCD has its own L2).  Maps the hardware workgroup id to a logical tile index so
// that every XCD works through one contiguous eighth of the tile list: spatial neighbours (shared halos) meet in the
// same L2.  URSN_XCD_REMAP=0 at build time keeps the identity (A/B).
#ifndef URSN_XCD_REMAP
#define URSN_XCD_REMAP 1
#endif
__device__ __forceinline__ int ursn_xcd_block(int b, int G) {
#if URSN_XCD_REMAP
  const int q = G >> 3, r = G & 7, x = b & 7, i = b >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
#else
  return b;
#endif
}

// ---- shifted one-pass moments ---------------------------------------------------------------------------------------
// BatchNorm statistics ride in the conv epilogues as one-pass sums.  TensorFlow's moments are two-pass (SURVEY.md
// Appendix B-3d); a plain one-pass  E[z^2] - E[z]^2  on fp32 partial sums loses var/mean^2 digits (measured: rstd off by
// 1e-2 at |mean|/std = 1e3).  In the implicit-GEMM / stride-2 kernels every lane accumulates  sum(z - K),
// sum((z - K)^2)  around a pivot K (the first value the lane sees of that channel), which keeps the fp32 partials at the
// scale of the variance, and the partials are re-centred to K = 0 in fp64 BEFORE they meet partials with another pivot:
//     sum z = s1 + n K        sum z^2 = s2 + 2 K s1 + n K^2
// The lane-per-voxel kernels (tconv, tdeconv, pconv) have no registers for pivots (measured: +17..36 VGPRs, spills in
// the 16-channel shapes, +2 ms per cfg3 step): they keep plain sums and bn_stats_final_kernel recomputes an
// ill-conditioned channel two-pass from z (elementwise.hip).
__device__ __forceinline__ void ursn_sacc(float piv, float& s1, float& s2, float v) {
  const float d = v - piv;
  s1 += d;
  s2 = fmaf(d, d, s2);
}
__device__ __forceinline__ void ursn_sacc_final(float piv, float s1, float s2, float n, double& S1, double& S2) {
  const double K = (double)piv, a = (double)s1, m = (double)n;
  S1 = a + m * K;
  S2 = (double)s2 + 2.0 * K * a + m * K * K;
}
#endif   // rename the last dispatch without counting a launch
long ursn_kernel_launch_count();
// roctx ranges (URSN_ROCTX=1): "UResNet/<scope>:<pass>" around every launch group, so rocprofv3 --marker-trace
// --kernel-trace attributes kernels to layers (stands in for the per-layer report of lib/ssnet_trainval.py:207-227).
// librocprofiler-sdk-roctx is opened at run time; without the switch these are two untaken branches.
bool ursn_roctx_on();
void ursn_roctx_push(const char* scope, int pass);
void ursn_roctx_pop();

#define URSN_HIP(expr)                                                                   \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess) {                                                              \
      ursn_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
      return 1;                                                                          \
    }                                                                                    \
  } while (0)

#define URSN_REQUIRE(cond, ...)     \
  do {                              \
    if (!(cond)) {                  \
      ursn_set_error(__VA_ARGS__);  \
      return 2;                     \
    }                               \
  } while (0)

#define URSN_TRY(expr)        \
  do {                        \
    int rc_ = (expr);         \
    if (rc_ != 0) return rc_; \
  } while (0)

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------------------------------
// Gather-convolution geometry.  Every conv-like op of the path (conv s1/s2, 1x1, transposed
// conv by output-parity class, and all data gradients) is
//     out[n, q*so+po, :] (+)= sum_t  in[n, q*si + d_t, :] . W_t        (q over q_d)
// with W_t addressed as w[tap_w[t]*w_tap_stride + k*w_sk + n*w_sn] (k = contraction index).
// Weight gradients are  dW_t[m][n] = sum_q S[q*si + d_t][m] * C[q][n].
// Spatial axes are (d0,d1,d2), d2 fastest; 2-D problems use d0 == 1.
// ---------------------------------------------------------------------------------------
#define URSN_MAX_TAPS 27

struct GatherGeom {
  int N;
  int in_d[3];   // spatial dims of the gathered ("shifted") tensor
  int out_d[3];  // spatial dims of the output tensor
  int q_d[3];    // iteration grid
  int so[3], po[3];
  int si[3];
  int ntaps;
  int tap_d[URSN_MAX_TAPS][3];
  int tap_w[URSN_MAX_TAPS];
  int K;   // contraction channels
  int Nn;  // output channels
  int in_cs, out_cs;
  int w_tap_stride, w_sk, w_sn;
  int accumulate;
};

// conv kernels (conv_generic.hip)
int launch_gconv_naive(const GatherGeom& g, const float* in, const float* w, float* out, hipStream_t s);
int launch_gconv_mfma(const GatherGeom& g, const float* in, const float* w, float* out, hipStream_t s);

struct WgradPlan {
  int RT, BN, nchunks;
  int rows;          // T*M
  size_t scratch_bytes;
};
// dW[t][m][n] += sum_q S[q*si+d_t][m] * C[q][n]; S has (in_d, in_cs), C has (q_d, out_cs); M = g.K, N = g.Nn.
WgradPlan wgrad_plan(const GatherGeom& g);
int launch_wgrad_naive(const GatherGeom& g, const float* S, const float* C, float* dw, hipStream_t s);
int launch_wgrad_mfma(const GatherGeom& g, const float* S, const float* C, float* dw, void* scratch,
                      size_t scratch_bytes, hipStream_t s);

// Geometry builders (conv_geom.cpp part of api)
enum ConvPass { PASS_FWD = 0, PASS_DGRAD = 1, PASS_WGRAD = 2 };
// Fills one or more GatherGeom (transposed-type passes need one per output-parity class).
// Returns number of geoms written (<= 8).
int build_geoms(const ursn_conv_desc& d, ConvPass pass, GatherGeom* out8);

// ---------------------------------------------------------------------------------------
// Elementwise / reduction kernels (elementwise.hip)
// ---------------------------------------------------------------------------------------
struct BnStats {     // per-layer device block, all arrays [C]
  double* sum;       // partial-reduced: sum z, sum z^2  (2*C doubles) then mean / var
  float* mean;
  float* rstd;
};

// Reduction scratch: doubles [nblocks][nsums][C]
size_t reduce_scratch_bytes(int64_t voxels, int channels, int nsums);
int reduce_nblocks(int64_t voxels, int channels);

// sum z, sum z^2 over voxels -> mean[C], rstd[C] (fp32), two kernels.
int launch_bn_stats(const float* z, int zcs, int64_t V, int C, float eps, float* mean, float* rstd,
                    void* scratch, hipStream_t s);
// finalise [nblocks][2][C] double partials (as written by the conv epilogue) into mean/rstd
// PC = channel stride of the partials (C padded to 4).  Producers sum around pivots (wave_pivot.h / shifted moments below)
int launch_bn_stats_final(const double* partial, int nblocks, int C, int PC, int64_t V, float eps, float* mean,
                          float* rstd, hipStream_t s);
// partials of channel block ct (CB channels each, PC columns per partial row) start at partial + ct * blk_stride: one launch for all blocks
int launch_bn_stats_final_blocked(const double* partial, int nblocks, int C, int CB, int PC, size_t blk_stride, int64_t V, float eps,
                                  float* mean, float* rstd, hipStream_t s);
// tiled small-channel conv (conv_tiled.hip): forward with fused BN-statistics partials
int tiled_conv_supported(const ursn_conv_desc& d, ConvPass pass);
size_t tiled_conv_stats_scratch_doubles(const ursn_conv_desc& d);
int launch_tiled_conv_bn(const ursn_conv_desc& d, const float* in, const float* w, float* out, double* scratch,
                         float eps, float* mean, float* rstd, hipStream_t s);
int tiled_wgrad_supported(const ursn_conv_desc& d);
int tiled_conv_bs_blocks(const ursn_conv_desc& d);   // partial blocks of a data gradient with fused BatchNorm-backward sums (0: unsupported)
// LDS-staged implicit GEMM for k3 s1 layers with >= 32 channels (conv_igemm.hip)
int igemm_conv_supported(const ursn_conv_desc& d, ConvPass pass);
size_t igemm_stats_scratch_doubles(const ursn_conv_desc& d);
int launch_igemm_conv(const ursn_conv_desc& d, ConvPass pass, const float* in, const float* w, float* out,
                      int accumulate, double* stats_partial, float eps, float* mean, float* rstd, hipStream_t s);
int igemm_wgrad_supported(const ursn_conv_desc& d);
size_t igemm_wgrad_scratch_bytes(const ursn_conv_desc& d);
int launch_igemm_wgrad(const ursn_conv_desc& d, const float* x, const float* dy, float* dw, void* scratch,
                       size_t scratch_bytes, hipStream_t s);
// weight-streaming fp32 kernel for the deepest levels (>= 128 channels, <= 16384 voxels; conv_deep.hip): forward (+ fused
// BatchNorm moments) and data gradient of plain 3-D k3 s1 layers; scratch = packed weights (+ split-K slabs) in floats
int deep_conv_supported(const ursn_conv_desc& d, ConvPass pass);
size_t deep_conv_scratch_floats(const ursn_conv_desc& d, ConvPass pass);
size_t deep_conv_stats_scratch_doubles(const ursn_conv_desc& d);
int launch_deep_conv(const ursn_conv_desc& d, ConvPass pass, const float* in, const float* w, float* out, int accumulate,
                     float* scratch, double* stats_partial, float eps, float* mean, float* rstd, hipStream_t s);
// weight gradient of the same deep-level layers with both operands straight from L2 (wgrad_deep.hip)
int deep_wgrad_supported(const ursn_conv_desc& d);
size_t deep_wgrad_scratch_bytes(const ursn_conv_desc& d);
int launch_deep_wgrad(const ursn_conv_desc& d, const float* x, const float* dy, float* dw, void* scratch, size_t scratch_bytes,
                      hipStream_t s);
// pointwise (1x1) shortcut convolutions (conv_pointwise.hip)
int pointwise_conv_supported(const ursn_conv_desc& d, ConvPass pass, int accumulate);
size_t pointwise_stats_scratch_doubles(const ursn_conv_desc& d);
int launch_pointwise_conv(const ursn_conv_desc& d, ConvPass pass, const float* in, const float* w, float* out,
                          int accumulate, double* stats_partial, float eps, float* mean, float* rstd, hipStream_t s);
int pointwise_wgrad_supported(const ursn_conv_desc& d);
size_t pointwise_wgrad_scratch_bytes(const ursn_conv_desc& d);
int launch_pointwise_wgrad(const ursn_conv_desc& d, const float* x, const float* dy, float* dw, void* scratch,
                           size_t scratch_bytes, hipStream_t s);
// LDS-staged stride-2 gather-type conv (conv_stride2.hip): conv k3 s2 forward / transposed-conv data gradient + wgrads
int stride2_conv_supported(const ursn_conv_desc& d, ConvPass pass);
size_t stride2_stats_scratch_doubles(const ursn_conv_desc& d);
int launch_stride2_conv(const ursn_conv_desc& d, ConvPass pass, const float* in, const float* w, float* out,
                        int accumulate, double* stats_partial, float eps, float* mean, float* rstd, hipStream_t s);
int stride2_wgrad_supported(const ursn_conv_desc& d);
size_t stride2_wgrad_scratch_bytes(const ursn_conv_desc& d);
int launch_stride2_wgrad(const ursn_conv_desc& d, const float* x, const float* dy, float* dw, void* scratch,
                         size_t scratch_bytes, hipStream_t s);
// tiled stride-2 scatter-type conv (deconv_tiled.hip): transposed-conv forward / stride-2 conv data gradient
int tiled_deconv_supported(const ursn_conv_desc& d, ConvPass pass);
size_t tiled_deconv_stats_scratch_doubles(const ursn_conv_desc& d);
int launch_tiled_deconv(const ursn_conv_desc& d, ConvPass pass, const float* in, const float* w, float* out,
                        int accumulate, double* stats_partial, float eps, float* mean, float* rstd, hipStream_t s);
int tiled_deconv_blocks(const ursn_conv_desc& d, ConvPass pass);
// z-segment choice of the marching kernels (conv_tiled.hip): minimises rounds x (segment + prologue)
void ursn_pick_zseg(int64_t base, int Z, int occ, int min_seg, int& zseg, int& nzseg);
int ursn_cu_count();
// LDS-staged stride-2 scatter-type conv for channel counts that are multiples of 16 (deconv_lds.hip)
int lds_scatter_supported(const ursn_conv_desc& d, ConvPass pass);
size_t lds_scatter_stats_scratch_doubles(const ursn_conv_desc& d);
int launch_lds_scatter(const ursn_conv_desc& d, ConvPass pass, const float* in, const float* w, float* out,
                       int accumulate, double* stats_partial, float eps, float* mean, float* rstd, hipStream_t s);
// the LDS kernel takes over where the lane-per-voxel kernel would need several channel-block launches or does not apply
static inline bool prefer_lds_scatter(const ursn_conv_desc& d, ConvPass pass) {
  return lds_scatter_supported(d, pass) && tiled_deconv_blocks(d, pass) != 1;
}
int launch_reduce_accum_blocked(float* dst, const float* src, int taps, int rows, int cols, int64_t dst_tap_stride,
                                int dst_row_stride, int nchunks, hipStream_t s);
// y = act(bn(z) [+ bn(z2) | + res]); any of z2/res may be null.
struct BnActArgs {
  const float* z; int zcs; const float* mean; const float* rstd; const float* beta;
  const float* z2; int z2cs; const float* mean2; const float* rstd2; const float* beta2;
  const float* res; int rescs;
  float* y; int ycs;
  int64_t V; int C; int relu;
  // optional (relu, float4 path, C/4 a power of two): one bit per element, y > 0, for the backward pass -- stored per
  // wave as 4 ballots (component j of lane l = element 4l + j of the wave's 256 consecutive elements); size
  // bn_mask_words(V, C) 64-bit words
  unsigned long long* mask_out;
};
int launch_bn_act(const BnActArgs& a, hipStream_t s);
static inline size_t bn_mask_words(int64_t V, int C) { return (size_t)((V * C + 255) / 256) * 4; }
static inline bool bn_mask_ok(int C) { int q = C / 4; return (C % 4) == 0 && q >= 1 && q <= 64 && (q & (q - 1)) == 0; }

// Backward of the above.  g = dy * (relu ? y > 0 : 1).
//   dz  = rstd  * (g - mean(g) - xhat  * mean(g*xhat))      dbeta  += sum g
//   dz2 = rstd2 * (g - mean(g) - xhat2 * mean(g*xhat2))     dbeta2 += sum g     (if z2)
//   dres (=|+=) g                                                                (if dres)
struct BnBwdArgs {
  const float* dy; int dycs; const float* y; int ycs;
  const float* z; int zcs; const float* mean; const float* rstd; float* dz; int dzcs; float* dbeta;
  const float* beta;  // with relu and y == nullptr the mask is recomputed as bn(z) > 0 (saves reading y twice)
  const float* z2; int z2cs; const float* mean2; const float* rstd2; float* dz2; int dz2cs; float* dbeta2;
  float* dres; int drescs; int dres_accumulate;
  int64_t V; int C; int relu;
  void* scratch;
  const unsigned long long* mask;  // relu mask bits written by launch_bn_act (mask_out); replaces the y reads
  // the reductions were already taken by the kernel that produced dy (ursn_conv_desc.bs_partial): [pre_nblocks][3][C]
  // doubles; the reduce pass is skipped
  const double* pre_partial; int pre_nblocks;
  int Cw;  // channels that own a dbeta entry (0 = C); C may be the 4-padded count of the logits layer (pad: mean = rstd = 0)
  // != nullptr (single z, no dres; relu none or bn(z) > 0): the apply pass is SKIPPED; the finalise writes [6][C] floats
  // {A, B, C, mu, S, T} with dz = A g' + B (z - mu) + C, g' = g (* (fma(z, S, T) > 0)), and the layer's data-gradient kernel
  // forms and stores dz while it stages its operand (ursn_conv_desc.vdz_*)
  float* coef_out;
};
int launch_bn_bwd(const BnBwdArgs& a, hipStream_t s);

// Head: logits = bn(z) (mean/rstd/beta may be null => z are logits already).
struct HeadArgs {
  const float* z; int z_cs; const float* mean; const float* rstd; const float* beta;
  const float* data; int data_cs; const float* label; const float* weight;
  int n; int64_t pix; int ncls;
  float* softmax_out; float* dlogits;  // nullable
  int dl_cs;                           // channel stride of dlogits (0 = ncls); 4 with ncls <= 4: one 16-byte store per pixel
  float* ana_out;                      // nullable: ssnet label volume (lib/ssnet_trainval.py:285-287), [n*pix]
  double* partial;                     // [nblocks][4]
  float* metrics;                      // device [3 + 1]
  void* scratch;
  // with dlogits in the 4-padded layout (dl_cs = z_cs = 4, <= 4 classes): the BatchNorm-backward reductions of the logits layer,
  // [head_blocks][3][4] doubles = sum g, sum g * xhat(z) (BnBwdArgs.pre_partial), or null
  double* bs_partial;
};
size_t head_scratch_bytes(int n, int64_t pix);
int head_blocks(int n, int64_t pix);
int launch_head_final(const double* partial, int nblocks, int n, int64_t pix, float* metrics, hipStream_t s);
// finals [3][C] = mean(g), mean(g xhat), mean(g xhat2) from [nblocks][3][C] partials; dbeta(2)[c < Cw] += sum g
int launch_bn_bwd_final(const double* partial, int nblocks, int C, int64_t V, double* finals, float* dbeta, float* dbeta2,
                        int Cw, hipStream_t s, float* coef_out = nullptr, const float* mean = nullptr, const float* rstd = nullptr,
                        const float* beta = nullptr);
int launch_head(const HeadArgs& a, hipStream_t s);

int launch_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr_t, float b1, float b2,
                float eps, hipStream_t s);
int launch_fill(float* p, float value, int64_t n, hipStream_t s);
// dst[i] += sum_c src[c*n + i]
int launch_reduce_accum(float* dst, const float* src, int64_t n, int nchunks, hipStream_t s);
int launch_mfma_probe(int which, float* out, hipStream_t s);
