/*
 * uresnet_hip.h -- C-ABI of the MI355X-native U-ResNet forward/backward hot path.
 *
 * The reference (DeepLearnPhysics/u-resnet) has no FFI: the path sits behind a Python
 * class protocol (lib/ssnet.py ssnet_base) whose methods each issue one tf.Session.run
 * fetch-set.  This header declares one entry point per fetch-set plus lifecycle, so a
 * Python 3 `ssnet_base` (u-resnet_amd/ssnet.py) binds them with ctypes and keeps the
 * reference surface.  Citations are file:line relative to the reference tree.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on error; ursn_last_error() gives text;
 *   - all tensor pointers are DEVICE pointers (hipMalloc'd or torch tensor.data_ptr());
 *     the caller owns every buffer it passes in, the library owns only the host-side handle;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *   - layouts follow the reference: activations N(D)HWC row-major (lib/ssnet.py:34-40),
 *     conv filters [k..,Cin,Cout], transposed-conv filters [k..,Cout,Cin];
 *   - one handle per GPU; calls on a handle are serialised by the caller (the reference
 *     issues all sess.run calls from one thread, lib/ssnet_trainval.py:156-233).
 */
#ifndef URESNET_HIP_H
#define URESNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define URSN_ABI_VERSION 7

typedef struct ursn_net ursn_net; /* opaque */

/* Mirrors uresnet.__init__ (lib/uresnet.py:15-20) + ssnet_base.construct (lib/ssnet.py:20-24). */
typedef struct ursn_config {
  int32_t ndim;          /* 2 (H,W,C) or 3 (H,W,D,C): len(dims)-1, lib/ssnet.py:11-14            */
  int32_t spatial[3];    /* dims[:-1]; spatial[2] unused for ndim==2; each divisible by 2^strides */
  int32_t cin;           /* dims[-1]                                                              */
  int32_t base_filters;  /* base_num_outputs, lib/uresnet.py:18                                   */
  int32_t num_class;     /* lib/ssnet.py:15                                                       */
  int32_t num_strides;   /* lib/uresnet.py:19 (reference fixes 5)                                 */
  int32_t max_batch;     /* largest N any call will pass (placeholder dim is None in the ref)     */
  int32_t trainable;     /* construct(trainable=...): allocate the backward workspace             */
  int32_t use_weight;    /* construct(use_weight=...), lib/ssnet.py:68-69                         */
  float bn_eps;          /* slim.batch_norm epsilon (default 1e-3)                                */
  int32_t act_dtype;     /* 0: fp32 everywhere (the reference's precision).  1: bf16 mixed precision
                          * (BASELINE.json configs[4]): activations, raw conv outputs and gradient tensors
                          * live in HBM as bf16, convolutions run on bf16 MFMA with fp32 accumulation;
                          * parameters, BatchNorm statistics, accumulated gradients and Adam stay fp32.
                          * Needs cin == 1 and base_filters % 8 == 0.                               */
} ursn_config;

typedef struct ursn_sizes {
  int64_t n_params;        /* trainable floats (weights + BN beta), TF variable order             */
  int64_t n_tensors;       /* number of trainable variables (2 per conv-like layer)               */
  int64_t n_layers;        /* conv-like layers (58 for num_strides=5)                             */
  int64_t workspace_bytes; /* device bytes for activations / stashes / scratch at max_batch       */
} ursn_sizes;

typedef struct ursn_param_info {
  char name[128];      /* TF variable name, e.g. "UResNet/conv0/weights"                          */
  int64_t offset;      /* float offset inside the flat parameter / gradient buffers              */
  int64_t nelem;
  int32_t rank;
  int32_t shape[5];
} ursn_param_info;

/* ---- lifecycle ------------------------------------------------------------------------ */
int ursn_abi_version(void);
const char* ursn_last_error(void);

/* Size query; no device access. */
int ursn_query(const ursn_config* cfg, ursn_sizes* out);

/* The plan a configuration compiles to, without device access: conv-like layer `index` in TF variable order
 * (lib/uresnet.py:37-121, lib/resnet_module.py:25-66) and the operand order of decoder step `step`'s tf.concat
 * (lib/uresnet.py:81: [deconv_i, skip]).  ssnet_base.construct checks what _build recorded against these. */
typedef struct ursn_layer_info {
  char name[96];       /* TF scope, e.g. "UResNet/resnet_module0/module1/shortcut"                 */
  int32_t transposed;  /* 0 slim.conv{2,3}d, 1 slim.conv{2,3}d_transpose                            */
  int32_t k, stride, cin, cout;
  int32_t relu;        /* activation_fn: 1 = tf.nn.relu (conv0, deconv*, conv1), 0 = None            */
  int64_t w_offset;    /* float offsets of `weights` and `BatchNorm/beta` in the flat buffers       */
  int64_t beta_offset;
} ursn_layer_info;
int ursn_query_layer(const ursn_config* cfg, int64_t index, ursn_layer_info* out);
int ursn_query_concat(const ursn_config* cfg, int32_t step, char* first, char* second, size_t cap);

/* Replaces graph construction (lib/ssnet.py:20-89 + lib/uresnet.py:22-123).  The four flat
 * fp32 buffers (n_params floats each) and the workspace are caller-owned device memory;
 * `grads` is the accum_vars set (lib/ssnet.py:53-55), adam_m/adam_v the Adam slots.
 * grads/adam_m/adam_v may be NULL when cfg->trainable == 0. */
int ursn_create(const ursn_config* cfg, float* params, float* grads, float* adam_m, float* adam_v,
                void* workspace, size_t workspace_bytes, ursn_net** out);
int ursn_destroy(ursn_net* net);
int ursn_get_sizes(const ursn_net* net, ursn_sizes* out);
int ursn_param(const ursn_net* net, int64_t index, ursn_param_info* out);

/* ---- the fetch-sets of lib/ssnet.py:91-139 --------------------------------------------- */
/* zero_gradients (lib/ssnet.py:99-101). */
int ursn_zero_grad(ursn_net* net, void* stream);

/* accum_gradients (lib/ssnet.py:103-115): forward + loss + backward, gradients ADDED into
 * `grads` (assign_add, :77).  data [N,prod(dims)], label/weight [N,prod(dims[:-1])] fp32 (label
 * values are class indices stored as float, :32,40).  weight may be NULL iff use_weight==0.
 * If out3 != NULL the call synchronises the stream and writes {loss, acc_all, acc_nonzero};
 * if NULL it only enqueues (metrics stay on the device, see ursn_read_metrics). */
int ursn_accum_step(ursn_net* net, const float* data, const float* label, const float* weight,
                    int32_t n, float* out3, void* stream);

/* apply_gradients (lib/ssnet.py:117-119): TF-form Adam on the accumulated gradients.
 * lr<=0 selects the TF default 1e-3 (lib/ssnet.py:72-75). The step counter is kept in the handle. */
int ursn_apply_adam(ursn_net* net, float lr, void* stream);

/* run_test (lib/ssnet.py:121-128): forward + {loss, acc_all, acc_nonzero}, no gradients. */
int ursn_eval(ursn_net* net, const float* data, const float* label, const float* weight, int32_t n,
              float* out3, void* stream);

/* inference (lib/ssnet.py:130-139): softmax_out [N,*spatial,num_class]; if label != NULL also
 * out2 = {acc_all, acc_nonzero}.  Always synchronises. */
int ursn_infer(ursn_net* net, const float* data, const float* label, int32_t n, float* softmax_out,
               float* out2, void* stream);

/* ana_step's label rule on the device (lib/ssnet_trainval.py:285-287): labels_out [N,*spatial] =
 * ((p[1] > p[2]) * 1 + (p[2] >= p[1]) * 2) * (data > 1.0), from the same forward pass as the optional softmax_out
 * (NULL: the softmax never leaves the kernel) and, if label != NULL, out2 = {acc_all, acc_nonzero}
 * (the reference prints acc_nonzero per entry, lib/ssnet_trainval.py:262).  Synchronises. */
int ursn_infer_labels(ursn_net* net, const float* data, const float* label, int32_t n, float* labels_out,
                      float* softmax_out, float* out2, void* stream);

/* Metrics of the last accum/eval call: synchronises, writes {loss, acc_all, acc_nonzero}. */
int ursn_read_metrics(ursn_net* net, float* out3, void* stream);

/* Adam step counter access (checkpoint / tests). */
int ursn_get_adam_step(const ursn_net* net, int64_t* t);
int ursn_set_adam_step(ursn_net* net, int64_t t);

/* Debug/parity: device pointer + channel stride of a named internal tensor of the last forward.
 * name = TF scope ("UResNet/conv0", ".../module1") optionally suffixed ":z" (raw conv output). */
int ursn_tensor(const ursn_net* net, const char* name, float** ptr, int64_t* voxels, int32_t* channels,
                int32_t* cstride);

/* Per-launch timing with HIP events recorded on the launch stream (bench.py roofline leg; the
 * reference has no counterpart: lib/ssnet_trainval.py:48-49 only reports peak bytes).
 * pass: 0 conv fwd, 1 conv dgrad, 2 conv wgrad, 3 bn stats, 4 bn apply, 5 bn backward, 6 head. */
typedef struct ursn_prof_rec {
  char kernel[48];
  char layer[96];
  int32_t pass;
  float ms;
  double flops; /* algorithmic: 2*MACs of the layer */
  double bytes; /* algorithmic: x + y + w (fwd), dy + w + dx (dgrad), x + dy + dw (wgrad) */
  int32_t launches; /* launches of the named kernel inside this record (channel-block / split-input passes) */
  int32_t reserved_;
} ursn_prof_rec;
int ursn_profile_enable(ursn_net* net, int32_t on);
/* out == NULL: only counts.  Otherwise fills up to max_recs records and clears the log. */
int ursn_profile_read(ursn_net* net, ursn_prof_rec* out, int64_t max_recs, int64_t* n_out);

/* Weight gradients normally run on a second internal stream, overlapped with the data-gradient / BN-backward chain.
 * on = 0 serialises them on the caller's stream (per-kernel timing: concurrent kernels time-slice, so event intervals
 * would include the partner's work). */
int ursn_set_wgrad_overlap(ursn_net* net, int32_t on);
/* Name of the kernel the calling thread's last op-level / net-level dispatch chose (e.g. "tconv<8,8>", "bcbconv_bf16+pw"):
 * what ursn_prof_rec.kernel is filled from; the dispatch tests assert it.  Static storage, never NULL. */
const char* ursn_last_kernel_name(void);

/* ---- op-level entry points (unit parity tests; same kernels the net-level calls use) ----- */
typedef struct ursn_conv_desc {
  int32_t ndim;        /* 2 or 3                                                                   */
  int32_t n;           /* batch                                                                    */
  int32_t in_sp[3];    /* input spatial dims                                                       */
  int32_t cin, cout;
  int32_t k;           /* 1 or 3                                                                   */
  int32_t stride;      /* 1 or 2                                                                   */
  int32_t transposed;  /* 0: slim.conv{2,3}d SAME; 1: slim.conv{2,3}d_transpose k3 s2 SAME          */
  int32_t in_cstride;  /* channel stride (floats per voxel) of x / dx; 0 = compact (= cin)         */
  int32_t out_cstride; /* channel stride of y / dy; 0 = compact (= cout)                           */
  int32_t algo;        /* 0 auto, 1 naive reference, 2 gather MFMA, 3 tiled small-C, 4 LDS implicit GEMM, 5 pointwise, 6 LDS stride-2 gather, 7 LDS stride-2 scatter */
  /* Split input (a tf.concat that is never materialised, lib/uresnet.py:81): channels [0,in_split) of the layer input
   * live in x / dx, channels [in_split,cin) in x2 / dx2.  in_split = 0: single tensor.  Only k3 s1 and k1 s1 layers with
   * in_split = cin/2 on the tiled / pointwise kernels; other shapes return an error.                                   */
  int32_t in_split;
  int32_t in2_cstride; /* channel stride of x2 / dx2; 0 = compact (= cin - in_split)                                   */
  const float* x2;     /* forward and weight gradient: second input tensor                                            */
  float* dx2;          /* data gradient: second output tensor                                                         */
  /* Data gradient only: fused term of a parallel 1x1 conv with the same cin / cout and stride (the residual unit's shortcut,
   * lib/resnet_module.py:25-33): dx (+)= conv^T(dy, w) + pw_dy . pw_w^T.  pw_dy = NULL: none.  k3 s1 layers on the tiled /
   * all-taps kernels; fp32 k3 s2 layers 8 | 16 <- 16 channels on the lane-per-low-res-voxel kernel (the 1x1 stride-2 shortcut
   * touches the even-even-even voxels only; deconv_tiled_kernel.h).  Any other shape with pw_dy set is refused. */
  const float* pw_dy;  /* [voxels][cout] gradient at the shortcut conv's output                                       */
  const float* pw_w;   /* [cin][cout] shortcut weights                                                                 */
  int32_t pw_dy_cstride; /* 0 = compact (= cout)                                                                       */
  int32_t dtype;         /* 0: fp32 tensors.  1: x / y / dx / dy (and pw_dy, dx2) are bf16 (uint16 bit patterns), channel counts and
                          * strides multiples of 8, weights and dw stay fp32 (BASELINE configs[4] mixed precision).  The fused forms
                          * of the bf16 plan are reachable here on the shapes its dedicated kernels take (3-D k3 s1, 8 / 16 channels):
                          * in_mean / in_rstd / in_beta / in_relu (forward C -> C and weight gradient), pw_dy / pw_w (data gradient of a
                          * 16 -> 8 layer), in_split = 8 with dx2 (that data gradient written as two 8-channel tensors), and cin = 1:
                          * x is ONE fp32 channel per voxel (the network input read by conv0's forward and weight gradient)         */
  /* Normalise-on-load (forward and weight gradient): x is the RAW output z of the preceding conv whose BatchNorm has no
   * activation (resnet_conv1 inside a residual unit, lib/resnet_module.py:43-51); the kernel stages
   * (z - in_mean) * in_rstd + in_beta per input channel (zero padding stays zero), so that activation is never
   * written.  NULL: x is used as is.  k3 s1 tiled kernels with 8 / 16 input channels only.                          */
  const float* in_mean;
  const float* in_rstd;
  const float* in_beta;
  /* Data gradient only: the BatchNorm-backward reductions of the layer(s) whose INPUT gradient this call finishes, fused
   * into its epilogue (slim.batch_norm backward, lib/resnet_module.py:31,49,64: dz = r (g - mean g - xhat mean(g xhat))):
   * with g = dx * mask, bs_partial[block][0][c] = sum g, [1][c] = sum g * xhat(bs_z), [2][c] = sum g * xhat(bs_z2) over the
   * voxels of the block, ursn_conv_bs_blocks(d) blocks of 3 * cin doubles.  bs_relu: 0 no mask, 1 mask = bn(bs_z) > 0 (needs
   * bs_beta), 2 mask = the bit mask a residual join's forward wrote (bs_mask).  bs_z2 (optional): the second BatchNorm'd
   * branch of a residual join.  With a split input the reductions cover dx (the first tensor) only.
   * bs_partial = NULL: none.  3-D k3 s1 tiled kernels with 8 input channels per tensor only.                              */
  const float* bs_z;
  const float* bs_mean;
  const float* bs_rstd;
  const float* bs_beta;
  const float* bs_z2;
  const float* bs_mean2;
  const float* bs_rstd2;
  const void* bs_mask;
  double* bs_partial;
  int32_t bs_z_cstride;  /* 0 = compact */
  int32_t bs_z2_cstride;
  int32_t bs_relu;
  int32_t in_relu;       /* normalise-on-load with the producer's ReLU: stages max(bn(z), 0) (dtype 1 only: conv1 -> conv2,
                          * lib/uresnet.py:103-121)                                                                              */
  /* Data gradient only (ABI 7): BatchNorm-backward APPLY on load (slim.batch_norm backward of THIS layer, lib/resnet_module.py:49,
   * lib/uresnet.py:109).  The `dy` argument of ursn_conv_backward_data is then g, the gradient at the BatchNorm's OUTPUT, and
   * the kernel forms, per channel c, while it stages its operand
   *     dz = A g' + B (z - mu) + C,    g' = g * (fma(z, S, T) > 0)  if vdz_relu  else g
   * with vdz_coef = [6][cout] floats {A, B, C, mu, S, T} (A = S = rstd, B = -rstd^2 mean(g' xhat), C = -rstd mean(g'),
   * T = beta - mu rstd: what ursn_net's BatchNorm-backward finalise writes), uses it for dx and STORES it to vdz_out (same
   * layout / channel stride as dy): the weight gradient reads dz there and the separate apply pass (read g, z; write dz)
   * disappears.  3-D k3 s1 layers with cin = cout = 8 on the tiled kernels; z, dy and vdz_out share out_cstride.  NULL: off. */
  const float* vdz_z;
  const float* vdz_coef;
  float* vdz_out;
  int32_t vdz_relu;
} ursn_conv_desc;

/* y = conv(x, w).  w layout [k..,Cin,Cout] (transposed: [k..,Cout,Cin]). */
int ursn_conv_forward(const ursn_conv_desc* d, const float* x, const float* w, float* y, void* stream);
/* y = conv(x, w) plus the batch statistics BatchNorm needs: mean[cout], rstd[cout] = rsqrt(var+eps)
 * (slim.batch_norm as normalizer_fn, lib/uresnet.py:42).  scratch >= ursn_bn_scratch_bytes(out voxels, cout). */
int ursn_conv_forward_stats(const ursn_conv_desc* d, const float* x, const float* w, float* y, float* mean,
                            float* rstd, float eps, void* scratch, size_t scratch_bytes, void* stream);
/* dx (=|+=) conv^T(dy, w); accumulate != 0 adds into dx. */
int ursn_conv_backward_data(const ursn_conv_desc* d, const float* dy, const float* w, float* dx,
                            int32_t accumulate, void* stream);
/* dw += x (*) dy.  scratch: device buffer of ursn_conv_wgrad_scratch_bytes(d) bytes. */
int ursn_conv_backward_weight(const ursn_conv_desc* d, const float* x, const float* dy, float* dw,
                              void* scratch, size_t scratch_bytes, void* stream);
size_t ursn_conv_wgrad_scratch_bytes(const ursn_conv_desc* d);
/* Blocks of bs_partial a data-gradient call with fused BatchNorm-backward reductions writes (0: shape not supported). */
int32_t ursn_conv_bs_blocks(const ursn_conv_desc* d);

/* "Explain plan": name of the kernel family the dispatcher hands this descriptor to (pass: 0 forward, 1 data gradient,
 * 2 weight gradient), e.g. "tconv", "igemm", "bdconv", "gconv_mfma".  Host logic only -- no device access, nothing runs:
 * what the plan-guard tests and tools/ query. */
int ursn_conv_plan(const ursn_conv_desc* d, int32_t pass, char* out, size_t cap);

/* Batch-statistics BatchNorm (beta only) + optional residual + optional ReLU:
 * y = act((z-mu)*rsqrt(var+eps)+beta [+ res]); stats_out (optional) gets {mean[C], rstd[C]} as fp32. */
int ursn_bn_forward(const float* z, const float* beta, const float* res, float* y, int64_t voxels,
                    int32_t channels, float eps, int32_t relu, float* stats_out, void* scratch,
                    size_t scratch_bytes, void* stream);
/* dz = BN backward of g = dy * (relu ? y>0 : 1); dbeta += sum g. */
int ursn_bn_backward(const float* dy, const float* y, const float* z, float* dz, float* dbeta,
                     int64_t voxels, int32_t channels, float eps, int32_t relu, void* scratch,
                     size_t scratch_bytes, void* stream);
size_t ursn_bn_scratch_bytes(int64_t voxels, int32_t channels);

/* BatchNorm passes of the bf16 plan at op level (slim.batch_norm forward / backward at lib/resnet_module.py:31,49,64 and
 * lib/uresnet.py:42,75,85,109,119 on bf16 tensors; bf16_elementwise.hip) with every optional operand the plan uses.  Tensors
 * are bf16 bit patterns with channel counts / strides multiples of 8; statistics, beta and d(beta) are fp32.
 *   forward : y = act(bn(z) [+ bn2(z2) | + res]); mask_out (relu): one byte per 16-byte piece of y, bit j = (channel j > 0);
 *             cat != 0 (8 channels): y[v] = [act(bn(z)) | act(bn2(z2))], both halves of a concat voxel in one store
 *   backward: g = (dy [+ dy2]) * mask, mask = the bytes `mask` | y > 0 | bn(z) > 0 (first that is given; relu != 0);
 *             dz = r (g - mean g - xhat mean(g xhat)), d(beta) += sum g; with z2: the join's second BatchNorm gets dz2 /
 *             dbeta2 from the same g; dres (=|+=) g is the identity shortcut's share (lib/resnet_module.py:22-23,68)     */
typedef struct ursn_bn_bf16_desc {
  int64_t voxels;
  int32_t channels, relu;
  const void* z;   int32_t z_cstride;   const float *mean, *rstd, *beta;
  const void* z2;  int32_t z2_cstride;  const float *mean2, *rstd2, *beta2;
  const void* res; int32_t res_cstride;
  void* y;         int32_t y_cstride;   /* forward output; backward: optional mask source (y > 0) */
  uint8_t* mask_out;
  int32_t cat;
  const void* dy;  int32_t dy_cstride;
  const void* dy2; int32_t dy2_cstride;
  const uint8_t* mask;
  void* dz;        int32_t dz_cstride;
  void* dz2;       int32_t dz2_cstride;
  float *dbeta, *dbeta2;
  void* dres;      int32_t dres_cstride; int32_t dres_accumulate;
} ursn_bn_bf16_desc;
int ursn_bn_bf16_forward(const ursn_bn_bf16_desc* d, void* stream);
int ursn_bn_bf16_backward(const ursn_bn_bf16_desc* d, void* scratch, size_t scratch_bytes, void* stream);
size_t ursn_bn_bf16_scratch_bytes(int64_t voxels, int32_t channels);

/* Fused head (lib/ssnet.py:57-71): softmax / weighted CE / accuracies / dlogits.
 * logits [n*pix, ncls]; data [n*pix] (cin==1); out3 = {loss, acc_all, acc_nonzero};
 * softmax_out, dlogits may be NULL. Synchronises. */
int ursn_softmax_ce(const float* logits, const float* data, const float* label, const float* weight,
                    int32_t n, int64_t pix, int32_t ncls, float* softmax_out, float* dlogits,
                    float* out3, void* scratch, size_t scratch_bytes, void* stream);

/* TF-form Adam on a flat buffer: lr_t = lr*sqrt(1-b2^t)/(1-b1^t); p -= lr_t*m/(sqrt(v)+eps). */
int ursn_adam(float* p, const float* g, float* m, float* v, int64_t nelem, float lr, float b1, float b2,
              float eps, int64_t t, void* stream);

/* MFMA lane-layout probe used by tests (writes 64*16 floats). */
int ursn_mfma_probe(int32_t which, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* URESNET_HIP_H */
