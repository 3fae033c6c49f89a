// bf16 mixed-precision path (BASELINE.json configs[4]: 3-D 256^3 bf16 with MFMA channel-block tiles) -- shared
// declarations.  Activations, raw conv outputs z, and every gradient tensor live in HBM as bf16 (channel counts padded to
// multiples of 8, so a voxel is a whole number of 16-byte pieces); weights stay fp32 masters and are re-packed into bf16
// MFMA operand order at the start of every pass; all accumulation (conv, BatchNorm moments, weight gradients, Adam) is
// fp32 / fp64.  gfx950 only.
#pragma once
#include "ursn_common.h"

typedef unsigned short bf16_t;   // storage type (bit pattern)
typedef __bf16 bfx8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float bf_f32x4 __attribute__((ext_vector_type(4)));

#ifdef __HIPCC__
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((unsigned)h) << 16); }
// round-to-nearest-even, NaN stays NaN (plain cast: v_cvt_pk_bf16_f32, MI355X_MICROARCH.md correctness boundaries)
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
typedef float bf_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bfx2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {   // one v_cvt_pk_bf16_f32
  const bf_f32x2 f = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bfx2));
}
// 8 bf16 (one 16-byte piece) <-> 8 floats
__device__ __forceinline__ void unpack8(const u32x4& p, float (&f)[8]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = __uint_as_float(p[i] << 16);
    f[2 * i + 1] = __uint_as_float(p[i] & 0xffff0000u);
  }
}
__device__ __forceinline__ u32x4 pack8(const float (&f)[8]) {
  u32x4 p;
#pragma unroll
  for (int i = 0; i < 4; ++i) p[i] = pack_bf2(f[2 * i], f[2 * i + 1]);
  return p;
}
#endif

// ---- conv-like ops on bf16 tensors (bf16_conv.hip): every layer type runs through two kernels driven by the GatherGeom of
// conv_api.hip::build_geoms ------------------------------------------------------------------------------------------
// packed-weight elements (bf16) one geometry needs; the pack kernel re-orders the fp32 master weights into MFMA A-operand
// order [co block][ci chunk][k step][co tile][lane][8]
size_t bconv_pack_elems(const GatherGeom& g);
// out[n, q*so+po, :] (=|+=) sum_t in[n, q*si+d_t, :] . W_t.  Kw / Nw (0 = g.K / g.Nn): extents of the stored weight tensor when
// the kernel-view channel counts are padded (conv0's single input channel, the 3|5-class logits layer); the weight strides in
// g describe the STORED tensor.  wpack: scratch of bconv_pack_elems(g) bf16.
// stats_partial != nullptr (forward): BatchNorm moment partials of the produced tensor, block stats_off + i of stats_total
// (a transposed conv runs one launch per output-parity class into one partial array); finalise with bconv_stats_finalize.
int launch_bconv(const GatherGeom& g, const bf16_t* in, const float* w, int Kw, int Nw, bf16_t* wpack, bf16_t* out,
                 double* stats_partial, int stats_off, int stats_total, hipStream_t s);
int bconv_grid_blocks(const GatherGeom& g);
size_t bconv_stats_scratch_doubles(const GatherGeom& g);   // for ONE launch; x classes for a transposed conv
int bconv_stats_finalize(const GatherGeom& g, const double* partial, int total_blocks, int64_t V, float eps, float* mean,
                         float* rstd, hipStream_t s);
// Buffer-path addressing (buffer_stage.h): EVERY tensor a launch builds a z-plane resource / 32-bit byte offset for -- the
// auxiliary operands (fused shortcut term, second output, fused BatchNorm-backward operands) as well as in / out -- must
// keep a z plane below the out-of-range marker.  cs = channel stride in bf16 elements; 0 = operand absent.
static inline bool ursn_bf16_plane_ok(const GatherGeom& g, int cs) {
  const int64_t pv = (int64_t)g.in_d[1] * g.in_d[2], qv = (int64_t)g.out_d[1] * g.out_d[2];
  return (pv > qv ? pv : qv) * (int64_t)cs * 2 < (int64_t)0x40000000;
}
// 8 -> 8 channel 3x3x3 stride-1 layers run on the input-stationary kernel of bf16_conv3.hip; the entry points above dispatch
bool b3conv_ok(const GatherGeom& g);
size_t b3conv_pack_elems();
int b3conv_grid_blocks(const GatherGeom& g);
// pw (data gradient 8 -> 16 only, b3conv_pw_ok): + pw[v] . pw_w^T, the data gradient of the module's 1x1 shortcut
// (pw = the shortcut's dz, 8 channels; pw_w = its weights [16][8])
bool b3conv_pw_ok(const GatherGeom& g);
// bs (data gradients 8 -> 8, b3conv_bs_ok): the BatchNorm-backward reductions of the layer(s) consuming the produced gradient,
// taken in the epilogue: partial[b3conv_grid_blocks(g)][3][8] doubles for launch_bbn_bwd (BBnBwdArgs.pre_partial)
struct B3BnRed {
  const bf16_t* z; int z_cs; const float* mean; const float* rstd; const float* beta;   // beta: mode 2
  const bf16_t* y; int y_cs;                                                           // mode 1: mask = y > 0
  const unsigned char* maskb;                                                          // mode 3: relu mask bytes (bit j = channel j)
  const bf16_t* z2; int z2_cs; const float* mean2; const float* rstd2;                 // second BatchNorm of a join (optional)
  int mode;                                                                            // 0 no mask, 1 y > 0, 2 bn(z) > 0, 3 mask bytes
  double* partial;
};
// normalise-on-load (forward C -> C and the weight gradient): `in` / `S` is the raw z of the preceding conv, its BatchNorm
// (+ ReLU) is applied while planes are staged
struct B3Affine { const float* mean; const float* rstd; const float* beta; int relu; };
// data gradient of resnet_conv1 behind an identity shortcut: + g (gradient of the unit's output, channel stride cs) where the join's
// relu mask byte says so -- the residual branch's share of d(input)
struct B3Residual { const bf16_t* g; int cs; const unsigned char* mask; };
bool b3conv_aff_ok(const GatherGeom& g);
bool b3conv_bs_ok(const GatherGeom& g);
int launch_b3conv(const GatherGeom& g, const bf16_t* in, const float* w, int Kw, int Nw, bf16_t* wpack, bf16_t* out,
                  double* stats_partial, int stats_off, int stats_total, hipStream_t s, const bf16_t* pw = nullptr,
                  int pw_cs = 0, const float* pw_w = nullptr, const B3BnRed* bs = nullptr, const B3Affine* aff = nullptr,
                  bf16_t* out2 = nullptr, int out2_cs = 0,    // out2: produced channels 8..15 of a 16-channel result
                  const float* in_f32 = nullptr,              // in_f32 (K = 8): the input is one fp32 channel per voxel (`in` unused)
                  const B3Residual* res = nullptr);           // out = conv + res->g * relu mask (plain C -> C data gradients)
// 3x3x3 stride-1 layers with 16 / 32 contraction channels (levels 1 / 2 of an F = 8 network): z-marching channel-block kernel,
// weights resident in LDS (bf16_convcb.hip); launch_bconv and the bconv_* helpers dispatch
bool bcbconv_ok(const GatherGeom& g);
size_t bcbconv_pack_elems(const GatherGeom& g);
int bcbconv_grid_blocks(const GatherGeom& g);
size_t bcbconv_stats_scratch_doubles(const GatherGeom& g);
// pw (data gradient, bcbconv_pw_ok): + pw[v] . pw_w^T, the data gradient of the unit's 1x1 stride-1 shortcut (pw = its dz with
// g.K channels, pw_w = its weights [g.Nn][g.K])
bool bcbconv_pw_ok(const GatherGeom& g);
int launch_bcbconv(const GatherGeom& g, const bf16_t* in, const float* w, int Kw, int Nw, bf16_t* wpack, bf16_t* out,
                   double* stats_partial, hipStream_t s, const bf16_t* pw = nullptr, int pw_cs = 0, const float* pw_w = nullptr);
int bcbconv_stats_finalize(const GatherGeom& g, const double* partial, int64_t V, float eps, float* mean, float* rstd, hipStream_t s);
// 3x3x3 stride-1 layers with >= 64 contraction channels (levels 3-5 of an F = 8 network): weight-streaming kernel with the
// contraction split inside the workgroup, over workgroups too where voxels are few (bf16_convdeep.hip); launch_bconv and the
// bconv_* helpers dispatch.  Its packed-weight scratch (bdconv_pack_elems) includes the fp32 slabs of a split-K launch.
bool bdconv_ok(const GatherGeom& g);
size_t bdconv_pack_elems(const GatherGeom& g);
int bdconv_grid_blocks(const GatherGeom& g);
size_t bdconv_stats_scratch_doubles(const GatherGeom& g);
int launch_bdconv(const GatherGeom& g, const bf16_t* in, const float* w, int Kw, int Nw, bf16_t* wpack, bf16_t* out,
                  double* stats_partial, hipStream_t s);
int bdconv_stats_finalize(const GatherGeom& g, const double* partial, int64_t V, float eps, float* mean, float* rstd, hipStream_t s);
// 1x1 stride-1 conv between 8 / 16 channel tensors, operands straight from global memory (bf16_pointwise.hip)
bool bpw_ok(const GatherGeom& g);
int bpw_grid_blocks(const GatherGeom& g);
int launch_bpw(const GatherGeom& g, const bf16_t* in, const float* w, int Kw, int Nw, bf16_t* out, double* stats_partial,
               hipStream_t s);
// the 8 output-parity classes of a stride-2 scatter-type pass (16 -> 8 channels, even extents) in one launch (bf16_deconv3.hip)
bool bdeconv_ok(const GatherGeom* g, int cnt);
int bdeconv_grid_blocks(const GatherGeom* g, int cnt);   // = rows of its statistics partials ([grid][2][16] doubles)
size_t bdeconv_pack_elems();
int launch_bdeconv(const GatherGeom* g, int cnt, const bf16_t* in, const float* w, int Kw, int Nw, bf16_t* wpack, bf16_t* out,
                   double* stats_partial, int accumulate, hipStream_t s);
// ... and of the deeper levels (>= 32 contraction channels, >= 16 produced): weight-streaming, classes dealt to the waves
// (bf16_scatter.hip).  Statistics partials: [block of produced channels][bsconv_grid_blocks][2][64] doubles.
bool bsconv_ok(const GatherGeom* g, int cnt);
int bsconv_grid_blocks(const GatherGeom* g, int cnt);
size_t bsconv_pack_elems(const GatherGeom* g, int cnt);
size_t bsconv_stats_scratch_doubles(const GatherGeom* g, int cnt);
int launch_bsconv(const GatherGeom* g, int cnt, const bf16_t* in, const float* w, int Kw, int Nw, bf16_t* wpack, bf16_t* out,
                  double* stats_partial, int accumulate, hipStream_t s);
int bsconv_stats_finalize(const GatherGeom* g, int cnt, const double* partial, int64_t V, float eps, float* mean, float* rstd,
                          hipStream_t s);
// conv0: one fp32 input channel per voxel -> 8 channels, 3x3x3 stride 1 (bf16_conv0.hip): the taps are the contraction.  g = the
// layer's geometry in its 8-channel kernel view (K = 8).  Statistics partials: [b0conv_grid_blocks][2][16] doubles.
bool b0conv_ok(const GatherGeom& g);
int b0conv_grid_blocks(const GatherGeom& g);
size_t b0conv_pack_elems();
int launch_b0conv(const GatherGeom& g, const float* x, const float* w, int Nw, bf16_t* wpack, bf16_t* out, double* stats_partial,
                  hipStream_t s);
bool b0wgrad_ok(const GatherGeom& g);
size_t b0wgrad_scratch_bytes(const GatherGeom& g);
int launch_b0wgrad(const GatherGeom& g, const float* x, const bf16_t* dz, float* dw, int Nw, void* scratch, size_t scratch_bytes,
                   hipStream_t s);
// stride-2 gather 8 -> 16 channels, z-marching (bf16_s2k8.hip): forward of the first stride-2 conv (+ its unit's 1x1 stride-2
// shortcut conv in the same pass: sc_w [8][16], second output, second set of moments) and data gradient of the last transposed conv
bool bs2k8_ok(const GatherGeom& g);
int bs2k8_grid_blocks(const GatherGeom& g);   // statistics partials: [blocks][2][16] doubles, the shortcut's behind the conv's
size_t bs2k8_pack_elems();
int launch_bs2k8(const GatherGeom& g, const bf16_t* in, const float* w, int Kw, int Nw, bf16_t* wpack, bf16_t* out, double* stats_partial,
                 int accumulate, const float* sc_w, bf16_t* out2, int out2_cs, hipStream_t s);
// ... and their weight gradients (bf16_s2k8w.hip): S = the fine tensor (8 channels), C = the coarse one (16); C2 / dw2: the dz of the
// unit's 1x1 stride-2 shortcut (coarse, 16 channels, stride c2_cs) and its weight gradient [8][16] (+=), taken in the same pass
bool bs2k8w_ok(const GatherGeom& g);
bool bs2k8w_sc_ok(const GatherGeom& g);
size_t bs2k8w_scratch_bytes(const GatherGeom& g);
int launch_bs2k8w(const GatherGeom& g, const bf16_t* S, const bf16_t* C, float* dw, int Kw, int Nw, void* scratch, size_t scratch_bytes,
                  const bf16_t* C2, int c2_cs, float* dw2, hipStream_t s);
bool b3wgrad_ok(const GatherGeom& g);   // z-marching weight gradient of the same layers (bf16_wgrad3.hip)
bool b3wgrad_scalar_ok(const GatherGeom& g);   // S may be one fp32 channel per voxel (S_f32)
size_t b3wgrad_scratch_bytes(const GatherGeom& g);
int launch_b3wgrad(const GatherGeom& g, const bf16_t* S, const bf16_t* C, float* dw, int Kw, int Nw, void* scratch,
                   size_t scratch_bytes, hipStream_t s, const B3Affine* aff = nullptr, const float* S_f32 = nullptr);
// dW[t][m][n] += sum_q S[q*si+d_t][m] * C[q][n]   (fp32 accumulation, fp32 dW [t][Kw][Nw])
// ... of the deep levels (3x3x3 stride 1, contraction channels a multiple of 32, produced channels a multiple of 64):
// 32 x 32 x 16 tiles, taps split over seven waves, double-buffered boxes (bf16_wgraddeep.hip); launch_bwgrad dispatches
bool bdwgrad_ok(const GatherGeom& g);
size_t bdwgrad_scratch_bytes(const GatherGeom& g);
int launch_bdwgrad(const GatherGeom& g, const bf16_t* S, const bf16_t* C, float* dw, int Kw, int Nw, void* scratch, size_t scratch_bytes,
                   hipStream_t s);
size_t bwgrad_scratch_bytes(const GatherGeom& g);
int launch_bwgrad(const GatherGeom& g, const bf16_t* S, const bf16_t* C, float* dw, int Kw, int Nw, void* scratch,
                  size_t scratch_bytes, hipStream_t s);

// ---- elementwise (bf16_elementwise.hip) ----------------------------------------------------------------------------------
struct BBnActArgs {   // y = act(bn(z) [+ bn(z2) | + res]); all tensors bf16, channel counts multiples of 8
  const bf16_t* z; int zcs; const float* mean; const float* rstd; const float* beta;
  const bf16_t* z2; int z2cs; const float* mean2; const float* rstd2; const float* beta2;
  const bf16_t* res; int rescs;
  bf16_t* y; int ycs;
  int64_t V; int C; int relu;
  unsigned char* mask_out;   // optional (relu): one byte per 16-byte piece of y, bit j = (channel j of the piece > 0)
  int cat;   // C = 8 only: y[v][0:8] = act(bn(z)), y[v][8:16] = act(bn2(z2)) -- both halves of a concat voxel in one 32-byte store
};
int launch_bbn_act(const BBnActArgs& a, hipStream_t s);
struct BBnBwdArgs {   // as BnBwdArgs (ursn_common.h) on bf16 tensors; the relu mask is y > 0 (y given) or bn(z) > 0 (beta given)
  const bf16_t* dy; int dycs; const bf16_t* y; int ycs;
  const bf16_t* z; int zcs; const float* mean; const float* rstd; bf16_t* dz; int dzcs; float* dbeta;
  const float* beta;
  const bf16_t* z2; int z2cs; const float* mean2; const float* rstd2; bf16_t* dz2; int dz2cs; float* dbeta2;
  bf16_t* dres; int drescs; int dres_accumulate;
  int64_t V; int C; int relu;
  void* scratch;
  int Cw;
  const double* pre_partial; int pre_nblocks;   // the reductions came out of the producing kernel's epilogue: [pre_nblocks][3][C]
  const unsigned char* mask;                    // relu mask bytes written by launch_bbn_act (mask_out); replaces the y reads
  const bf16_t* dy2; int dy2cs;                 // second contribution to the output gradient (C = 8, mask = bn(z) > 0 only): g = dy + dy2
};
int launch_bbn_bwd(const BBnBwdArgs& a, hipStream_t s);
size_t bbn_scratch_bytes(int64_t V, int C);
struct BHeadArgs {    // logits = bn(z); z, dlogits bf16 with channel stride 8
  const bf16_t* z; int z_cs; const float* mean; const float* rstd; const float* beta;
  const float* data; int data_cs; const float* label; const float* weight;
  int n; int64_t pix; int ncls;
  float* softmax_out; bf16_t* dlogits; int dl_cs;
  float* ana_out;
  float* metrics; void* scratch;
  // want dlogits: the BatchNorm-backward reductions of the logits layer, taken from the STORED (rounded) dlogits while they are
  // in registers: [head blocks][3][8] doubles = sum g, sum g * xhat(z) (the layout launch_bbn_bwd's pre_partial reads), or null
  double* bs_partial;
};
int bhead_blocks(int n, int64_t pix);   // rows of bs_partial
int launch_bhead(const BHeadArgs& a, hipStream_t s);
// data [V] fp32 (one input channel) -> [V][8] bf16, channels 1..7 zero
int launch_bf16_input(const float* data, bf16_t* out, int64_t V, hipStream_t s);
int launch_f32_to_bf16(const float* src, bf16_t* dst, int64_t n, hipStream_t s);
int launch_bf16_to_f32(const bf16_t* src, float* dst, int64_t n, hipStream_t s);
