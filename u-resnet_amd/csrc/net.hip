// Net-level C-ABI: the U-ResNet of lib/uresnet.py:22-123 + lib/resnet_module.py:10-87 as a fixed
// launch plan over caller-owned device memory, one entry point per sess.run fetch-set of
// lib/ssnet.py:91-139.  Host-side orchestration only; all arithmetic is in the HIP kernels.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "ursn_common.h"
#include "net_bf16.h"

int conv_dispatch(const ursn_conv_desc& d, ConvPass pass, const float* in, const float* w, float* out, int accumulate,
                  hipStream_t s);
int wgrad_dispatch(const ursn_conv_desc& d, const float* x, const float* dy, float* dw, void* scratch,
                   size_t scratch_bytes, hipStream_t s);

namespace {

struct Arena {
  char* base = nullptr;
  size_t off = 0;
  void* take(size_t bytes) {
    off = (off + 255) & ~(size_t)255;
    void* p = base ? (void*)(base + off) : nullptr;
    off += bytes;
    return p;
  }
  float* floats(int64_t n) { return (float*)take((size_t)n * sizeof(float)); }
};

const char* g_last_kernel_dummy = "";

struct Act {       // activation tensor view (+ gradient view with identical layout)
  float* p = nullptr;
  float* g = nullptr;
  int C = 0, cs = 0, lvl = 0;
  int flag = -1;   // index into ursn_net::ginit (gradient-initialised state shared between views)
};

struct Layer {
  std::string name;
  int kind = 0;  // 0 conv, 1 deconv
  int k = 3, stride = 1, cin = 0, cout = 0, lin = 0, lout = 0;
  int relu = 0;  // activation_fn of the slim layer call (conv0 / deconv / conv1: ReLU; all others None)
  int64_t w_off = 0, b_off = 0, w_n = 0;
  ursn_conv_desc desc;
  float *z = nullptr, *dz = nullptr, *mean = nullptr, *rstd = nullptr;
  float* coef = nullptr;   // [6][zcs]: BatchNorm-backward apply coefficients for the data-gradient kernel (ursn_conv_desc.vdz_coef)
  int zcs = 0;  // channel stride of z / dz: cout rounded up to 4 (only conv2's 3|5 classes differ), pad lanes stay 0
};

struct Unit {
  std::string scope;
  int sc = -1, c1 = -1, c2 = -1;  // layer indices
  Act in, a1, out;
  Act in2;   // C > 0: the unit input is the never-materialised concat [in | in2] (decoder, narrow levels)
  bool a1_virtual = false;   // a1 = BN(z1) is never written: conv2 and its weight gradient normalise z1 while staging
  unsigned long long* jmask = nullptr;   // bit mask (out > 0) of the join ReLU: the two BN-backward passes read it instead of out
};

}  // namespace

struct ursn_net {
  ursn_bnet* bf = nullptr;   // act_dtype == 1: the bf16 mixed-precision plan (net_bf16.hip) executes the fetch-sets
  ursn_config cfg;
  ursn_sizes sizes;
  int nlev = 0;
  int ldim[8][3];        // spatial dims per level (3 entries, leading 1 for 2-D)
  int64_t lvox[8];       // voxels per image per level
  std::vector<Layer> layers;
  std::vector<Unit> units;           // encoder units then decoder units, execution order
  std::vector<int> deconv;           // layer index per decoder step
  std::vector<Act> deconv_in, deconv_out, cat;  // per decoder step
  int conv0 = -1, conv1 = -1, conv2 = -1;
  Act a_data, a_conv0, a_conv1, a_pre1;  // a_pre1: input of conv1 (last decoder unit output)
  std::vector<int> ginit;
  float *params = nullptr, *grads = nullptr, *adam_m = nullptr, *adam_v = nullptr;
  float* dlog = nullptr;     // d loss / d logits  [V0, ncls]
  float* metrics = nullptr;  // device [4]
  void* red_scratch = nullptr;
  void* red_scratch2 = nullptr;   // statistics partials of the shortcut convs running on the second stream (forward)
  void* head_scratch = nullptr;
  void* wg_scratch = nullptr;
  size_t wg_scratch_bytes = 0;
  float* dc_scratch = nullptr;   // packed weights (+ split-K slabs) of the deep-level kernel (conv_deep.hip), main stream only
  // BatchNorm-backward reductions taken in the epilogue of the data-gradient kernel that finished a layer's output gradient
  double* bs_scratch = nullptr;
  int bs_layer = -1, bs_blocks = 0;   // layer whose (first) reductions are waiting in bs_scratch
  int bs_C = 8;                       // ... and their channel count (8; the logits layer's, from the head: 4)
  // BatchNorm-backward apply on load: bn_back left this layer's dz to its data-gradient kernel (coefficients in Layer::coef)
  int vdz_layer = -1, vdz_relu = 0, vdz_cs = 0;
  const float* vdz_g = nullptr;
  int64_t adam_t = 0;
  int last_n = 0;
  // optional per-launch timing with HIP events on the launch stream (bench.py roofline leg)
  bool profile = false;
  struct ProfRec { int layer; int pass; const char* kernel; double flops; double bytes; hipEvent_t e0, e1; long l0; int launches; };
  std::vector<ProfRec> prof;
  std::vector<hipEvent_t> ev_pool;
  size_t ev_used = 0;
  // weight gradients run on a second stream: they only feed the optimiser, so they overlap the HBM-bound
  // BN-backward passes and the low-occupancy deep-level kernels of the main chain (URSN_WGRAD_STREAM=0 disables)
  hipStream_t s2 = nullptr;
  hipStream_t s2_owned = nullptr;   // s2 == s2_owned when the overlap is switched on
  // hipGraph replay of the accumulate step for launch-bound workloads (ursn_accum_step): one captured graph per (data, label,
  // weight, batch) of the last few calls
  struct StepGraph { const float* data; const float* label; const float* weight; int n, seen; hipGraphExec_t exec; };
  std::vector<StepGraph> graphs;
  bool graph_broken = false;
  hipStream_t gstream = nullptr;   // the graphs are captured on / launched into a stream of their own (the caller's may be the legacy
  hipEvent_t g_in = nullptr, g_out = nullptr;   // default stream, which cannot be captured), ordered against the caller's by two events
  std::vector<hipEvent_t> sync_pool;
  size_t sync_used = 0;
  std::vector<hipEvent_t> fwd_pool;   // forward: fork / join events of the side-stream shortcut convs
  size_t fwd_used = 0;
  hipEvent_t s2_done = nullptr;
  std::vector<std::pair<std::string, std::string>> cat_names;   // per decoder step: producers of [first | second] channel halves
  std::map<std::string, Act> named;       // debug lookup: activations
  std::map<std::string, int> named_z;     // layer name -> layer index
};

namespace {

int new_flag(ursn_net* n) {
  n->ginit.push_back(0);
  return (int)n->ginit.size() - 1;
}

Act make_act(ursn_net* n, Arena& A, int lvl, int C, bool grad, bool grad_only = false) {
  Act a;
  a.C = C;
  a.cs = C;
  a.lvl = lvl;
  int64_t e = (int64_t)n->cfg.max_batch * n->lvox[lvl] * C;
  a.p = grad_only ? nullptr : A.floats(e);
  a.g = grad ? A.floats(e) : nullptr;
  a.flag = new_flag(n);
  return a;
}

Act sub_act(const Act& full, int c0, int C) {
  Act a = full;
  a.p = full.p ? full.p + c0 : nullptr;
  a.g = full.g ? full.g + c0 : nullptr;
  a.C = C;
  return a;  // same cs, same flag
}

int add_layer(ursn_net* n, Arena& A, const std::string& name, int kind, int k, int s, int ci, int co, int lin, int lout,
              int64_t& poff, int relu = 0) {
  Layer L;
  L.name = "UResNet/" + name;
  L.relu = relu;
  L.kind = kind; L.k = k; L.stride = s; L.cin = ci; L.cout = co; L.lin = lin; L.lout = lout;
  int64_t taps = 1;
  for (int j = 0; j < n->cfg.ndim; ++j) taps *= k;
  L.w_n = taps * ci * co;
  L.w_off = poff; poff += L.w_n;
  L.b_off = poff; poff += co;
  memset(&L.desc, 0, sizeof(L.desc));
  L.desc.ndim = n->cfg.ndim;
  L.desc.n = n->cfg.max_batch;
  for (int j = 0; j < n->cfg.ndim; ++j) L.desc.in_sp[j] = n->ldim[lin][3 - n->cfg.ndim + j];
  L.desc.cin = ci; L.desc.cout = co; L.desc.k = k; L.desc.stride = s; L.desc.transposed = kind;
  L.zcs = (co + 3) & ~3;
  L.desc.out_cstride = L.zcs;
  int64_t e = (int64_t)n->cfg.max_batch * n->lvox[lout] * L.zcs;
  L.z = A.floats(e);
  L.dz = n->cfg.trainable ? A.floats(e) : nullptr;
  L.mean = A.floats(L.zcs);   // padded like z: pad entries stay 0 (a BN over the padded channel count yields dz = 0 there)
  L.rstd = A.floats(L.zcs);
  L.coef = n->cfg.trainable ? A.floats(6 * L.zcs) : nullptr;
  n->layers.push_back(L);
  n->named_z[L.name] = (int)n->layers.size() - 1;
  return (int)n->layers.size() - 1;
}

// Builds the whole plan; with A.base == nullptr only sizes are computed.
int plan(ursn_net* n, Arena& A) {
  const ursn_config& c = n->cfg;
  URSN_REQUIRE(c.ndim == 2 || c.ndim == 3, "len(dims) must be 3 (H,W,C) or 4 (H,W,D,C)");
  // the decoder scopes are the reference's literal 'resnet_module%d' % (step + 5) (lib/uresnet.py:100): with more than 5
  // strides they collide with the encoder's names (TensorFlow raises on the duplicate variable scope as well)
  URSN_REQUIRE(c.num_strides >= 1 && c.num_strides <= 5, "num_strides %d out of range [1,5]", c.num_strides);
  URSN_REQUIRE(c.cin >= 1 && c.base_filters >= 1 && c.num_class >= 1 && c.num_class <= 8 && c.max_batch >= 1,
               "bad channel / class / batch configuration");
  const int ns = c.num_strides;
  n->nlev = ns + 1;
  for (int l = 0; l <= ns; ++l) {
    n->lvox[l] = 1;
    for (int j = 0; j < 3; ++j) {
      int lead = 3 - c.ndim;
      if (j < lead) { n->ldim[l][j] = 1; continue; }
      int s0 = c.spatial[j - lead];
      URSN_REQUIRE(s0 > 0 && s0 % (1 << ns) == 0, "spatial size %d not divisible by 2^%d (deconv/skip shapes would differ)", s0, ns);
      n->ldim[l][j] = s0 >> l;
      n->lvox[l] *= n->ldim[l][j];
    }
  }
  const bool tr = c.trainable != 0;
  const int F = c.base_filters;
  int64_t poff = 0;
  n->layers.clear(); n->units.clear(); n->deconv.clear(); n->cat.clear(); n->deconv_in.clear(); n->deconv_out.clear();
  n->ginit.clear(); n->named.clear(); n->named_z.clear(); n->cat_names.clear();

  // concat buffers first (decoder step i lives at level ns-1-i with F*2^(ns-i) channels).  Where a half is narrower
  // than a 64-byte line (<= 8 channels) and the tiled / pointwise kernels take the two halves as separate tensors,
  // the concat is never materialised: half-line strided accesses cost the BN kernels ~2.5x (profiles/r01).
  n->cat.resize(ns);
  std::vector<char> split(ns, 0);
  std::vector<Act> lone_fmap(ns);
  for (int i = 0; i < ns; ++i) {
    const int lvl = ns - 1 - i, co = F << (ns - 1 - i);
    {
      const char* e = getenv("URSN_SPLIT_CAT");
      ursn_conv_desc d3;
      memset(&d3, 0, sizeof(d3));
      d3.ndim = c.ndim; d3.n = c.max_batch;
      for (int j = 0; j < c.ndim; ++j) d3.in_sp[j] = n->ldim[lvl][3 - c.ndim + j];
      d3.cin = 2 * co; d3.cout = co; d3.k = 3; d3.stride = 1; d3.in_split = co;
      ursn_conv_desc d1 = d3;
      d1.k = 1;
      split[i] = !(e && e[0] == '0') && co <= 8 && tiled_conv_supported(d3, PASS_FWD) && tiled_conv_supported(d3, PASS_DGRAD) &&
                 pointwise_conv_supported(d1, PASS_FWD, 0) && pointwise_conv_supported(d1, PASS_DGRAD, 1) &&
                 (!tr || (tiled_wgrad_supported(d3) && pointwise_wgrad_supported(d1)));
    }
    if (split[i]) lone_fmap[i] = make_act(n, A, lvl, co, tr);
    else n->cat[i] = make_act(n, A, lvl, 2 * co, tr);
  }
  auto fmap_view = [&](int lvl) {  // encoder feature map at level lvl (< ns) = second half of its concat buffer
    if (split[ns - 1 - lvl]) return lone_fmap[ns - 1 - lvl];
    const Act& full = n->cat[ns - 1 - lvl];
    return sub_act(full, full.C / 2, full.C / 2);
  };

  n->a_data = Act();
  n->a_data.C = c.cin; n->a_data.cs = c.cin; n->a_data.lvl = 0;

  n->conv0 = add_layer(n, A, "conv0", 0, 3, 1, c.cin, F, 0, 0, poff, 1);
  n->a_conv0 = fmap_view(0);
  n->named["UResNet/conv0"] = n->a_conv0;

  auto add_unit = [&](const std::string& scope, const Act& in, int co, int s, int lout, const Act* out_view,
                      const Act* in2 = nullptr) {
    Unit u;
    u.scope = "UResNet/" + scope;
    u.in = in;
    if (in2) u.in2 = *in2;
    const int cin = in.C + (in2 ? in2->C : 0);
    if (!(cin == co && s == 1)) u.sc = add_layer(n, A, scope + "/shortcut", 0, 1, s, cin, co, in.lvl, lout, poff);
    u.c1 = add_layer(n, A, scope + "/resnet_conv1", 0, 3, s, cin, co, in.lvl, lout, poff);
    {  // normalise-on-load (URSN_NORM_ON_LOAD=0 turns it off, =2 extends it to 16 channels): resnet_conv1's BatchNorm has no
       // activation, so where conv2 (forward + weight gradient) runs on the tiled kernels they apply it while staging and a1
       // is never written.  Rounds 1-2 measured it neutral (-0.7 ms of bn_act passes, +0.3..0.5 ms in the conv kernels) and
       // left it off; with the buffer-path staging of round 3 the affine instantiations lost their spills and gained a
       // workgroup per CU: 65.2 -> 65.9 images/s at cfg3, on by default.
      const char* e = getenv("URSN_NORM_ON_LOAD");
      const int nol = e ? atoi(e) : 1;
      ursn_conv_desc d2;
      memset(&d2, 0, sizeof(d2));
      d2.ndim = c.ndim; d2.n = c.max_batch;
      for (int j = 0; j < c.ndim; ++j) d2.in_sp[j] = n->ldim[lout][3 - c.ndim + j];
      d2.cin = co; d2.cout = co; d2.k = 3; d2.stride = 1;
      d2.in_mean = d2.in_rstd = d2.in_beta = n->layers[u.c1].mean;   // placeholders: only "non-null" matters here
      ursn_conv_desc d2p = d2;
      d2p.in_mean = d2p.in_rstd = d2p.in_beta = nullptr;
      // 8-channel layers by default: the 16 -> 16 kernel has no registers to spare (246 VGPRs with the affine)
      u.a1_virtual = nol >= 1 && (co == 8 || (nol >= 2 && co == 16)) && tiled_conv_supported(d2, PASS_FWD) &&
                     !igemm_conv_supported(d2p, PASS_FWD) && (!tr || tiled_wgrad_supported(d2));
    }
    u.a1 = make_act(n, A, lout, co, tr, u.a1_virtual);
    u.c2 = add_layer(n, A, scope + "/resnet_conv2", 0, 3, 1, co, co, lout, lout, poff);
    u.out = out_view ? *out_view : make_act(n, A, lout, co, tr);
    {
      const char* e = getenv("URSN_RELU_MASK");
      if (tr && bn_mask_ok(co) && !(e && e[0] == '0'))
        u.jmask = (unsigned long long*)A.take(bn_mask_words((int64_t)c.max_batch * n->lvox[lout], co) * sizeof(unsigned long long));
    }
    n->units.push_back(u);
    n->named[u.scope] = u.out;
    n->named[n->layers[u.c1].name] = u.a1;
    return u.out;
  };

  Act net = n->a_conv0;
  std::vector<std::string> skip_name(ns + 1);   // producer of the encoder feature map at each level
  skip_name[0] = "UResNet/conv0";
  for (int step = 0; step < ns; ++step) {
    char sc[64];
    int co = net.C * 2;
    snprintf(sc, sizeof(sc), "resnet_module%d/module1", step);
    Act u1 = add_unit(sc, net, co, 2, step + 1, nullptr);
    snprintf(sc, sizeof(sc), "resnet_module%d/module2", step);
    if (step + 1 < ns) {
      Act view = fmap_view(step + 1);
      net = add_unit(sc, u1, co, 1, step + 1, &view);
    } else {
      net = add_unit(sc, u1, co, 1, step + 1, nullptr);
    }
    skip_name[step + 1] = std::string("UResNet/") + sc;
  }
  for (int i = 0; i < ns; ++i) {
    char sc[64];
    int co = net.C / 2;
    int lvl = ns - 1 - i;
    snprintf(sc, sizeof(sc), "deconv%d", i);
    int li = add_layer(n, A, sc, 1, 3, 2, net.C, co, net.lvl, lvl, poff, 1);
    n->deconv.push_back(li);
    n->deconv_in.push_back(net);
    Act dout = split[i] ? make_act(n, A, lvl, co, tr) : sub_act(n->cat[i], 0, co);
    n->deconv_out.push_back(dout);
    n->named[n->layers[li].name] = dout;
    n->cat_names.push_back({n->layers[li].name, skip_name[lvl]});   // tf.concat([deconv_i, skip]) (lib/uresnet.py:81)
    snprintf(sc, sizeof(sc), "resnet_module%d/module1", i + 5);
    Act u1 = split[i] ? add_unit(sc, dout, co, 1, lvl, nullptr, &lone_fmap[i]) : add_unit(sc, n->cat[i], co, 1, lvl, nullptr);
    snprintf(sc, sizeof(sc), "resnet_module%d/module2", i + 5);
    net = add_unit(sc, u1, co, 1, lvl, nullptr);
  }
  n->a_pre1 = net;
  n->conv1 = add_layer(n, A, "conv1", 0, 3, 1, net.C, F, 0, 0, poff, 1);
  n->a_conv1 = make_act(n, A, 0, F, tr);
  n->named["UResNet/conv1"] = n->a_conv1;
  n->conv2 = add_layer(n, A, "conv2", 0, 3, 1, F, c.num_class, 0, 0, poff);

  const int64_t V0 = (int64_t)c.max_batch * n->lvox[0];
  n->dlog = tr ? A.floats(V0 * n->layers[n->conv2].zcs) : nullptr;   // channel stride = conv2's padded stride
  n->metrics = A.floats(8);
  n->head_scratch = A.take(head_scratch_bytes(c.max_batch, n->lvox[0]) + 64);
  size_t red = 0, wg = 0, dcf = 0;
  for (const Layer& L : n->layers) {
    size_t r = reduce_scratch_bytes((int64_t)c.max_batch * n->lvox[L.lout], L.cout, 3);
    if (r > red) red = r;
    // the launch geometry (z segments, box groups) is chosen per call from the batch actually fed, and a smaller batch
    // can split finer than the planned one: size the partial-sum and slab scratch for every batch up to max_batch
    for (int nb = 1; nb <= c.max_batch; ++nb) {
      ursn_conv_desc d = L.desc;
      d.n = nb;
      size_t rt = tiled_conv_stats_scratch_doubles(d) * sizeof(double);
      if (rt > red) red = rt;
      rt = tiled_deconv_stats_scratch_doubles(d) * sizeof(double);
      if (rt > red) red = rt;
      rt = igemm_stats_scratch_doubles(d) * sizeof(double);
      if (rt > red) red = rt;
      rt = pointwise_stats_scratch_doubles(d) * sizeof(double);
      if (rt > red) red = rt;
      rt = stride2_stats_scratch_doubles(d) * sizeof(double);
      if (rt > red) red = rt;
      rt = lds_scatter_stats_scratch_doubles(d) * sizeof(double);
      if (rt > red) red = rt;
      rt = deep_conv_stats_scratch_doubles(d) * sizeof(double);
      if (rt > red) red = rt;
      {
        size_t f = deep_conv_scratch_floats(d, PASS_FWD);
        if (f > dcf) dcf = f;
        f = deep_conv_scratch_floats(d, PASS_DGRAD);
        if (f > dcf) dcf = f;
      }
      if (tr) {
        size_t w = ursn_conv_wgrad_scratch_bytes(&d);
        if (w > wg) wg = w;
      }
    }
  }
  n->red_scratch = A.take(red + 256);
  n->dc_scratch = dcf ? A.floats((int64_t)dcf + 64) : nullptr;
  {
    size_t red2 = 0;
    for (const Unit& u : n->units) {
      if (u.sc < 0) continue;
      const Layer& L = n->layers[u.sc];
      size_t r = reduce_scratch_bytes((int64_t)c.max_batch * n->lvox[L.lout], L.cout, 3);
      if (r > red2) red2 = r;
      for (int nb = 1; nb <= c.max_batch; ++nb) {
        ursn_conv_desc d = L.desc;
        d.n = nb;
        r = pointwise_stats_scratch_doubles(d) * sizeof(double);
        if (r > red2) red2 = r;
      }
    }
    n->red_scratch2 = A.take(red2 + 256);
  }
  n->wg_scratch_bytes = wg;
  n->wg_scratch = tr ? A.take(wg + 256) : nullptr;
  n->bs_scratch = tr ? (double*)A.take((size_t)16384 * 3 * 8 * sizeof(double)) : nullptr;

  n->sizes.n_params = poff;
  n->sizes.n_layers = (int64_t)n->layers.size();
  n->sizes.n_tensors = 2 * (int64_t)n->layers.size();
  n->sizes.workspace_bytes = (int64_t)((A.off + 255) & ~(size_t)255);
  return 0;
}

// ---- profiling ------------------------------------------------------------------------------
extern "C" const char* ursn_last_kernel_name();
hipEvent_t prof_event(ursn_net* n) {
  if (n->ev_used == n->ev_pool.size()) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    n->ev_pool.push_back(e);
  }
  return n->ev_pool[n->ev_used++];
}
struct ProfScope {
  ursn_net* n; hipStream_t s; int idx = -1; bool range = false;
  ProfScope(ursn_net* n_, hipStream_t s_, int layer, int pass, double flops, double bytes) : n(n_), s(s_) {
    if (ursn_roctx_on()) { ursn_roctx_push(n->layers[layer].name.c_str(), pass); range = true; }
    if (!n->profile || n->prof.size() > 200000) return;
    ursn_net::ProfRec r{layer, pass, "", flops, bytes, prof_event(n), prof_event(n), ursn_kernel_launch_count(), 1};
    if (!r.e0 || !r.e1) return;
    hipEventRecord(r.e0, s);
    n->prof.push_back(r);
    idx = (int)n->prof.size() - 1;
  }
  ~ProfScope() { if (range) ursn_roctx_pop(); }
  void done(const char* kernel) {
    if (idx < 0) return;
    n->prof[idx].kernel = kernel;
    long nl = ursn_kernel_launch_count() - n->prof[idx].l0;
    n->prof[idx].launches = nl > 0 ? (int)nl : 1;   // elementwise scopes do not note their kernels: one launch
    hipEventRecord(n->prof[idx].e1, s);
  }
};
double layer_macs(const ursn_net* n, const Layer& L, int N) {
  double taps = 1;
  for (int j = 0; j < n->cfg.ndim; ++j) taps *= L.k;
  double vox = (double)N * (L.kind ? n->lvox[L.lin] : n->lvox[L.lout]);
  return vox * taps * L.cin * L.cout;
}
double layer_bytes(const ursn_net* n, const Layer& L, int N) {  // x + y + w  (== dy + w + dx == x + dy + dw)
  return 4.0 * ((double)N * n->lvox[L.lin] * L.cin + (double)N * n->lvox[L.lout] * L.cout + (double)L.w_n);
}

// ---- forward pieces -----------------------------------------------------------------------
int conv_stats(ursn_net* n, int li, const Act& in, int N, hipStream_t s, const Act* in2 = nullptr, int aff = -1,
               void* red_scratch = nullptr) {
  if (!red_scratch) red_scratch = n->red_scratch;
  Layer& L = n->layers[li];
  ursn_conv_desc d = L.desc;
  d.n = N;
  d.in_cstride = in.cs;
  d.out_cstride = L.zcs;
  if (in2) { d.in_split = in.C; d.in2_cstride = in2->cs; d.x2 = in2->p; }
  if (aff >= 0) {   // `in` is the raw z of layer aff: normalise on load
    Layer& P = n->layers[aff];
    d.in_mean = P.mean; d.in_rstd = P.rstd; d.in_beta = n->params + P.b_off;
  }
  if (pointwise_conv_supported(d, PASS_FWD, 0)) {  // 1x1 shortcut + BN-statistics partials in one pass
    ProfScope ps(n, s, li, 0, 2.0 * layer_macs(n, L, N), layer_bytes(n, L, N));
    URSN_TRY(launch_pointwise_conv(d, PASS_FWD, in.p, n->params + L.w_off, L.z, 0, (double*)red_scratch,
                                   n->cfg.bn_eps, L.mean, L.rstd, s));
    ps.done(ursn_last_kernel_name());
    return 0;
  }
  if (n->dc_scratch && red_scratch == n->red_scratch && deep_conv_supported(d, PASS_FWD)) {  // deepest levels: weight-streaming kernel + moments
    ProfScope ps(n, s, li, 0, 2.0 * layer_macs(n, L, N), layer_bytes(n, L, N));
    URSN_TRY(launch_deep_conv(d, PASS_FWD, in.p, n->params + L.w_off, L.z, 0, n->dc_scratch, (double*)red_scratch, n->cfg.bn_eps,
                              L.mean, L.rstd, s));
    ps.done(ursn_last_kernel_name());
    return 0;
  }
  if (igemm_conv_supported(d, PASS_FWD)) {  // LDS-staged implicit GEMM + BN-statistics partials in one pass
    ProfScope ps(n, s, li, 0, 2.0 * layer_macs(n, L, N), layer_bytes(n, L, N));
    URSN_TRY(launch_igemm_conv(d, PASS_FWD, in.p, n->params + L.w_off, L.z, 0, (double*)red_scratch,
                               n->cfg.bn_eps, L.mean, L.rstd, s));
    ps.done(ursn_last_kernel_name());
    return 0;
  }
  if (stride2_conv_supported(d, PASS_FWD)) {  // LDS-staged stride-2 conv + BN-statistics partials in one pass
    ProfScope ps(n, s, li, 0, 2.0 * layer_macs(n, L, N), layer_bytes(n, L, N));
    URSN_TRY(launch_stride2_conv(d, PASS_FWD, in.p, n->params + L.w_off, L.z, 0, (double*)red_scratch,
                                 n->cfg.bn_eps, L.mean, L.rstd, s));
    ps.done(ursn_last_kernel_name());
    return 0;
  }
  if (prefer_lds_scatter(d, PASS_FWD)) {  // LDS-staged transposed conv + BN-statistics partials in one pass
    ProfScope ps(n, s, li, 0, 2.0 * layer_macs(n, L, N), layer_bytes(n, L, N));
    URSN_TRY(launch_lds_scatter(d, PASS_FWD, in.p, n->params + L.w_off, L.z, 0, (double*)red_scratch,
                                n->cfg.bn_eps, L.mean, L.rstd, s));
    ps.done(ursn_last_kernel_name());
    return 0;
  }
  if (tiled_deconv_supported(d, PASS_FWD)) {  // transposed conv + BN-statistics partials in one pass
    ProfScope ps(n, s, li, 0, 2.0 * layer_macs(n, L, N), layer_bytes(n, L, N));
    URSN_TRY(launch_tiled_deconv(d, PASS_FWD, in.p, n->params + L.w_off, L.z, 0, (double*)red_scratch,
                                 n->cfg.bn_eps, L.mean, L.rstd, s));
    ps.done(ursn_last_kernel_name());
    return 0;
  }
  if (tiled_conv_supported(d, PASS_FWD)) {  // conv + BN-statistics partials in one pass
    ProfScope ps(n, s, li, 0, 2.0 * layer_macs(n, L, N), layer_bytes(n, L, N));
    URSN_TRY(launch_tiled_conv_bn(d, in.p, n->params + L.w_off, L.z, (double*)red_scratch, n->cfg.bn_eps, L.mean,
                                  L.rstd, s));
    ps.done(ursn_last_kernel_name());
    return 0;
  }
  {
    ProfScope ps(n, s, li, 0, 2.0 * layer_macs(n, L, N), layer_bytes(n, L, N));
    URSN_TRY(conv_dispatch(d, PASS_FWD, in.p, n->params + L.w_off, L.z, 0, s));
    ps.done(ursn_last_kernel_name());
  }
  {
    ProfScope ps(n, s, li, 3, 0.0, 4.0 * N * n->lvox[L.lout] * L.cout);
    URSN_TRY(launch_bn_stats(L.z, L.zcs, (int64_t)N * n->lvox[L.lout], L.cout, n->cfg.bn_eps, L.mean, L.rstd,
                             red_scratch, s));
    ps.done("bn_stats");
  }
  return 0;
}

int bn_out(ursn_net* n, int li, const Act& out, int relu, int N, int li2, const float* res, int rescs, hipStream_t s,
           unsigned long long* mask_out = nullptr) {
  Layer& L = n->layers[li];
  BnActArgs a;
  memset(&a, 0, sizeof(a));
  a.z = L.z; a.zcs = L.zcs; a.mean = L.mean; a.rstd = L.rstd; a.beta = n->params + L.b_off;
  if (li2 >= 0) {
    Layer& L2 = n->layers[li2];
    a.z2 = L2.z; a.z2cs = L2.cout; a.mean2 = L2.mean; a.rstd2 = L2.rstd; a.beta2 = n->params + L2.b_off;
  }
  a.res = res; a.rescs = rescs;
  a.y = out.p; a.ycs = out.cs; a.V = (int64_t)N * n->lvox[L.lout]; a.C = L.cout; a.relu = relu;
  a.mask_out = mask_out;
  ProfScope ps(n, s, li, 4, 0.0, 4.0 * a.V * a.C * (2 + (li2 >= 0) + (res != nullptr)));
  URSN_TRY(launch_bn_act(a, s));
  ps.done("bn_act");
  return 0;
}

hipEvent_t fwd_event(ursn_net* n) {
  if (n->fwd_used == n->fwd_pool.size()) {
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
    n->fwd_pool.push_back(e);
  }
  return n->fwd_pool[n->fwd_used++];
}

int unit_fwd(ursn_net* n, Unit& u, int N, hipStream_t s) {
  const Act* in2 = u.in2.C ? &u.in2 : nullptr;
  // the 1x1 shortcut is HBM-bound and independent of conv1 / conv2 until the join: it runs beside them on the second stream
  hipEvent_t side_done = nullptr;
  if (u.sc >= 0) {
    static const bool side_off = getenv("URSN_FWD_SIDE_STREAM") && getenv("URSN_FWD_SIDE_STREAM")[0] == '0';
    hipEvent_t fork = (n->s2 && !side_off) ? fwd_event(n) : nullptr;
    hipEvent_t join = fork ? fwd_event(n) : nullptr;
    if (fork && join) {
      URSN_HIP(hipEventRecord(fork, s));
      URSN_HIP(hipStreamWaitEvent(n->s2, fork, 0));
      URSN_TRY(conv_stats(n, u.sc, u.in, N, n->s2, in2, -1, n->red_scratch2));
      URSN_HIP(hipEventRecord(join, n->s2));
      side_done = join;
    } else {
      URSN_TRY(conv_stats(n, u.sc, u.in, N, s, in2));
    }
  }
  URSN_TRY(conv_stats(n, u.c1, u.in, N, s, in2));
  if (u.a1_virtual) {
    Act z1 = u.a1;
    z1.p = n->layers[u.c1].z; z1.cs = n->layers[u.c1].zcs;
    URSN_TRY(conv_stats(n, u.c2, z1, N, s, nullptr, u.c1));
  } else {
    URSN_TRY(bn_out(n, u.c1, u.a1, 0, N, -1, nullptr, 0, s));
    URSN_TRY(conv_stats(n, u.c2, u.a1, N, s));
  }
  unsigned long long* jm = n->cfg.trainable ? u.jmask : nullptr;
  if (side_done) URSN_HIP(hipStreamWaitEvent(s, side_done, 0));
  if (u.sc >= 0) URSN_TRY(bn_out(n, u.c2, u.out, 1, N, u.sc, nullptr, 0, s, jm));
  else URSN_TRY(bn_out(n, u.c2, u.out, 1, N, -1, u.in.p, u.in.cs, s, jm));
  return 0;
}

int forward(ursn_net* n, const float* data, int N, hipStream_t s) {
  const int ns = n->cfg.num_strides;
  n->fwd_used = 0;
  Act din = n->a_data;
  din.p = const_cast<float*>(data);
  URSN_TRY(conv_stats(n, n->conv0, din, N, s));
  URSN_TRY(bn_out(n, n->conv0, n->a_conv0, 1, N, -1, nullptr, 0, s));
  size_t ui = 0;
  for (int step = 0; step < ns; ++step) {
    URSN_TRY(unit_fwd(n, n->units[ui++], N, s));
    URSN_TRY(unit_fwd(n, n->units[ui++], N, s));
  }
  for (int i = 0; i < ns; ++i) {
    URSN_TRY(conv_stats(n, n->deconv[i], n->deconv_in[i], N, s));
    URSN_TRY(bn_out(n, n->deconv[i], n->deconv_out[i], 1, N, -1, nullptr, 0, s));
    URSN_TRY(unit_fwd(n, n->units[ui++], N, s));
    URSN_TRY(unit_fwd(n, n->units[ui++], N, s));
  }
  URSN_TRY(conv_stats(n, n->conv1, n->a_pre1, N, s));
  URSN_TRY(bn_out(n, n->conv1, n->a_conv1, 1, N, -1, nullptr, 0, s));
  URSN_TRY(conv_stats(n, n->conv2, n->a_conv1, N, s));
  return 0;
}

int head(ursn_net* n, const float* data, const float* label, const float* weight, int N, float* softmax_out,
         bool want_grad, hipStream_t s, float* ana_out = nullptr) {
  Layer& L = n->layers[n->conv2];
  HeadArgs a;
  memset(&a, 0, sizeof(a));
  a.z = L.z; a.z_cs = L.zcs; a.mean = L.mean; a.rstd = L.rstd; a.beta = n->params + L.b_off;
  a.data = (n->cfg.cin == 1) ? data : nullptr;  // acc_nonzero needs one input channel (lib/ssnet.py:59)
  a.data_cs = n->cfg.cin;
  a.label = label; a.weight = weight; a.n = N; a.pix = n->lvox[0]; a.ncls = n->cfg.num_class;
  a.softmax_out = softmax_out; a.dlogits = want_grad ? n->dlog : nullptr;
  a.dl_cs = L.zcs;
  a.ana_out = ana_out;
  a.scratch = n->head_scratch; a.metrics = n->metrics;
  // the logits layer's BatchNorm-backward sums ride in the head (dlogits and z are in its registers): one pass of two tensors less
  static const bool fuse = !(getenv("URSN_HEAD_BN_BWD") && getenv("URSN_HEAD_BN_BWD")[0] == '0');
  n->bs_layer = -1;
  if (want_grad && fuse && n->bs_scratch && L.zcs == 4 && a.ncls <= 4 && head_blocks(N, n->lvox[0]) <= 16384) {
    a.bs_partial = n->bs_scratch;
    n->bs_layer = n->conv2; n->bs_blocks = head_blocks(N, n->lvox[0]); n->bs_C = 4;
  }
  ProfScope ps(n, s, n->conv2, 6, 0.0, 4.0 * N * n->lvox[0] * (2.0 * a.ncls + 3));
  URSN_TRY(launch_head(a, s));
  ps.done("head");
  return 0;
}

// ---- backward pieces ----------------------------------------------------------------------
bool take_flag(ursn_net* n, const Act& a) {  // returns "accumulate?" and marks the gradient as initialised
  bool acc = n->ginit[a.flag] != 0;
  n->ginit[a.flag] = 1;
  return acc;
}

// desc of layer li for the backward passes; sc >= 0: with the fused data gradient of the unit's 1x1 shortcut;
// aff >= 0 (weight gradient only): x is the raw z of layer aff, normalised on load
ursn_conv_desc bwd_desc(ursn_net* n, int li, const Act& in, int N, const Act* in2, int sc, int aff = -1) {
  Layer& L = n->layers[li];
  ursn_conv_desc d = L.desc;
  d.n = N;
  d.in_cstride = in.cs;
  d.out_cstride = L.zcs;
  if (aff >= 0) {
    Layer& P = n->layers[aff];
    d.in_mean = P.mean; d.in_rstd = P.rstd; d.in_beta = n->params + P.b_off;
  }
  if (in2) { d.in_split = in.C; d.in2_cstride = in2->cs; d.x2 = in2->p; d.dx2 = in2->g; }
  if (sc >= 0) {
    Layer& S = n->layers[sc];
    d.pw_dy = S.dz; d.pw_w = n->params + S.w_off; d.pw_dy_cstride = S.zcs;
  }
  return d;
}

// The layer(s) whose BatchNorm backward consumes the gradient a data-gradient launch finishes: their reductions ride in
// that launch's epilogue where the kernel supports it (ursn_conv_desc.bs_partial), which removes one bn_bwd_reduce pass
struct BsTarget {
  int li = -1, li2 = -1, relu = 0;             // relu: 0 none, 1 bn(z) > 0, 2 the join's bit mask
  const unsigned long long* mask = nullptr;
};

int conv_bwd(ursn_net* n, int li, const Act& in, bool need_dgrad, int N, hipStream_t s, const Act* in2 = nullptr,
             int fused_sc = -1, bool dgrad_done_elsewhere = false, int aff = -1, const BsTarget* bs = nullptr) {
  Layer& L = n->layers[li];
  ursn_conv_desc d = bwd_desc(n, li, in, N, in2, -1, aff);
  const bool vdz = n->vdz_layer == li;   // dz does not exist yet: this layer's data-gradient kernel forms and stores it
  n->vdz_layer = -1;
  if (vdz) {
    URSN_REQUIRE(need_dgrad && !in2 && fused_sc < 0 && !dgrad_done_elsewhere, "BatchNorm-backward apply on load: %s has no plain data gradient", L.name.c_str());
    bool acc = take_flag(n, in);
    ursn_conv_desc t = bwd_desc(n, li, in, N, nullptr, -1);
    t.vdz_z = L.z; t.vdz_coef = L.coef; t.vdz_out = L.dz; t.vdz_relu = n->vdz_relu;
    n->bs_layer = -1;
    static const bool bs_off = getenv("URSN_FUSE_BN_BWD_REDUCE") && getenv("URSN_FUSE_BN_BWD_REDUCE")[0] == '0';
    if (bs && bs->li >= 0 && !bs_off) {
      const Layer& T = n->layers[bs->li];
      ursn_conv_desc u = t;
      u.bs_z = T.z; u.bs_z_cstride = T.zcs; u.bs_mean = T.mean; u.bs_rstd = T.rstd; u.bs_beta = n->params + T.b_off;
      u.bs_relu = bs->relu; u.bs_mask = bs->mask;
      if (bs->li2 >= 0) {
        const Layer& T2 = n->layers[bs->li2];
        u.bs_z2 = T2.z; u.bs_z2_cstride = T2.zcs; u.bs_mean2 = T2.mean; u.bs_rstd2 = T2.rstd;
      }
      u.bs_partial = n->bs_scratch;
      const bool fits = T.cout == 8 && T.zcs == 8 && (bs->li2 < 0 || n->layers[bs->li2].zcs == 8) && (bs->relu != 2 || bs->mask);
      const int blocks = fits ? tiled_conv_bs_blocks(u) : 0;
      if (blocks > 0 && blocks <= 16384 && tiled_conv_supported(u, PASS_DGRAD)) { t = u; n->bs_layer = bs->li; n->bs_blocks = blocks; }
    }
    {
      ProfScope pd(n, s, li, 1, 2.0 * layer_macs(n, L, N), layer_bytes(n, L, N));
      URSN_TRY(conv_dispatch(t, PASS_DGRAD, n->vdz_g, n->params + L.w_off, in.g, acc ? 1 : 0, s));
      pd.done(ursn_last_kernel_name());
    }
    need_dgrad = false;   // done; the weight gradient below is ordered behind it (it reads the dz just stored)
  }
  hipStream_t ws = s;
  if (n->s2) {  // dz is final once the kernels queued so far on the main stream are done
    if (n->sync_used == n->sync_pool.size()) {
      hipEvent_t e;
      URSN_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      n->sync_pool.push_back(e);
    }
    hipEvent_t e = n->sync_pool[n->sync_used++];
    URSN_HIP(hipEventRecord(e, s));
    URSN_HIP(hipStreamWaitEvent(n->s2, e, 0));
    ws = n->s2;
  }
  {
    ProfScope ps(n, ws, li, 2, 2.0 * layer_macs(n, L, N), layer_bytes(n, L, N));
    URSN_TRY(wgrad_dispatch(d, in.p, L.dz, n->grads + L.w_off, n->wg_scratch, n->wg_scratch_bytes, ws));
    ps.done(ursn_last_kernel_name());
  }
  if (dgrad_done_elsewhere) {  // keeps the per-layer roofline accounting (bench.py) complete: the work ran inside conv1's kernel
    ProfScope pd(n, s, li, 1, 2.0 * layer_macs(n, L, N), layer_bytes(n, L, N));
    pd.done("(fused into conv1 dgrad)");
  }
  if (need_dgrad) {
    bool acc = take_flag(n, in);
    if (in2) URSN_REQUIRE(take_flag(n, *in2) == acc, "split input: the two halves disagree on gradient initialisation");
    d = bwd_desc(n, li, in, N, in2, fused_sc);   // the data gradient does not read x: no normalise-on-load fields
    n->bs_layer = -1;
    static const bool bs_off = getenv("URSN_FUSE_BN_BWD_REDUCE") && getenv("URSN_FUSE_BN_BWD_REDUCE")[0] == '0';
    if (bs && bs->li >= 0 && !bs_off) {
      const Layer& T = n->layers[bs->li];
      ursn_conv_desc t = d;
      t.bs_z = T.z; t.bs_z_cstride = T.zcs; t.bs_mean = T.mean; t.bs_rstd = T.rstd; t.bs_beta = n->params + T.b_off;
      t.bs_relu = bs->relu; t.bs_mask = bs->mask;
      if (bs->li2 >= 0) {
        const Layer& T2 = n->layers[bs->li2];
        t.bs_z2 = T2.z; t.bs_z2_cstride = T2.zcs; t.bs_mean2 = T2.mean; t.bs_rstd2 = T2.rstd;
      }
      t.bs_partial = n->bs_scratch;
      // the target's channels are the channels this launch writes (block 0 of a split input)
      // (a split input is left to the separate pass: measured 2.15 vs 1.77 ms for the pair of launches against 0.33 ms saved)
      const bool fits = T.cout == 8 && T.zcs == 8 && in.C == 8 && in.cs == 8 && !in2 &&
                        (bs->li2 < 0 || n->layers[bs->li2].zcs == 8) && (bs->relu != 2 || bs->mask);
      const int blocks = fits ? tiled_conv_bs_blocks(t) : 0;
      if (blocks > 0 && blocks <= 16384) { d = t; n->bs_layer = bs->li; n->bs_blocks = blocks; }
    }
    ProfScope pd(n, s, li, 1, 2.0 * layer_macs(n, L, N), layer_bytes(n, L, N));
    if (n->dc_scratch && deep_conv_supported(d, PASS_DGRAD))
      URSN_TRY(launch_deep_conv(d, PASS_DGRAD, L.dz, n->params + L.w_off, in.g, acc ? 1 : 0, n->dc_scratch, nullptr, 0.f, nullptr, nullptr, s));
    else
      URSN_TRY(conv_dispatch(d, PASS_DGRAD, L.dz, n->params + L.w_off, in.g, acc ? 1 : 0, s));
    pd.done(ursn_last_kernel_name());
  }
  return 0;
}

// May the apply pass of layer li's BatchNorm backward ride in the layer's own data-gradient kernel?  3-D k3 s1 8 -> 8 layers
// on the tiled kernels whose input is ONE compact 8-channel tensor (conv1, the resnet_conv1 of an 8-channel unit): the kernel
// forms dz = A g' + B (z - mu) + C while it stages g and z and stores it for the weight gradient (URSN_FUSE_BN_BWD_APPLY=0: off)
// Measured (round 4, cfg3): the data gradient pays for the extra staging in matrix-pipe issue slots -- conv1's launch 0.97 ->
// 1.29 ms for 0.47 ms of apply pass removed (net -0.15 ms), but the launch that ALSO carries the two-z reductions of a
// residual join below drops from 3 to 2 waves per SIMD (190 VGPRs) and goes 1.18 -> 1.84 ms: by default only launches
// without a second z take the fusion (URSN_FUSE_BN_BWD_APPLY=2: all, 0: none).
bool vdz_ok(ursn_net* n, int li, const Act& in, int N, bool two_z_target) {
  static const int mode = getenv("URSN_FUSE_BN_BWD_APPLY") ? atoi(getenv("URSN_FUSE_BN_BWD_APPLY")) : 1;
  const bool off = mode == 0 || (mode == 1 && two_z_target);
  const Layer& L = n->layers[li];
  if (off || n->cfg.ndim != 3 || L.kind || L.k != 3 || L.stride != 1 || L.cin != 8 || L.cout != 8 || L.zcs != 8 || in.C != 8 || in.cs != 8 || !in.g)
    return false;
  ursn_conv_desc d = L.desc;
  d.n = N; d.in_cstride = in.cs; d.out_cstride = L.zcs;
  d.vdz_z = L.z; d.vdz_coef = L.coef; d.vdz_out = L.dz;
  return tiled_conv_supported(d, PASS_DGRAD) != 0;
}

int bn_back(ursn_net* n, int li, const float* dy, int dycs, const float* y, int ycs, int relu, int li2, float* dres,
            int drescs, int dres_acc, int N, hipStream_t s, const unsigned long long* mask = nullptr, const Act* fuse_in = nullptr,
            bool fuse_two_z = false) {
  Layer& L = n->layers[li];
  BnBwdArgs a;
  memset(&a, 0, sizeof(a));
  a.dy = dy; a.dycs = dycs; a.y = mask ? nullptr : y; a.ycs = ycs; a.mask = mask;
  a.z = L.z; a.zcs = L.zcs; a.mean = L.mean; a.rstd = L.rstd; a.dz = L.dz; a.dzcs = L.zcs;
  a.dbeta = n->grads + L.b_off;
  a.beta = n->params + L.b_off;
  if (li2 >= 0) {
    Layer& L2 = n->layers[li2];
    a.z2 = L2.z; a.z2cs = L2.cout; a.mean2 = L2.mean; a.rstd2 = L2.rstd; a.dz2 = L2.dz; a.dz2cs = L2.cout;
    a.dbeta2 = n->grads + L2.b_off;
  }
  a.dres = dres; a.drescs = drescs; a.dres_accumulate = dres_acc;
  a.V = (int64_t)N * n->lvox[L.lout]; a.C = L.cout; a.relu = relu; a.scratch = n->red_scratch;
  if (L.zcs != L.cout && li2 < 0 && !relu && dycs == L.zcs && !dres) { a.C = L.zcs; a.Cw = L.cout; }   // logits layer: float4 path
  if (n->bs_layer == li && a.C == n->bs_C) { a.pre_partial = n->bs_scratch; a.pre_nblocks = n->bs_blocks; }
  n->bs_layer = -1;
  n->bs_C = 8;
  n->vdz_layer = -1;
  if (fuse_in && li2 < 0 && !dres && !mask && !y && dycs == L.zcs && vdz_ok(n, li, *fuse_in, N, fuse_two_z)) {
    a.coef_out = L.coef;
    n->vdz_layer = li; n->vdz_relu = relu; n->vdz_g = dy; n->vdz_cs = dycs;
  }
  ProfScope ps(n, s, li, 5, 0.0, 4.0 * a.V * a.C * (2.0 * (2 + (relu && !mask) + (li2 >= 0)) + 1 + (li2 >= 0) + (dres != nullptr)));
  URSN_TRY(launch_bn_bwd(a, s));
  ps.done("bn_bwd");
  return 0;
}

int unit_bwd(ursn_net* n, Unit& u, int N, hipStream_t s, const BsTarget* in_target = nullptr) {
  // join: g = dout * (out > 0); BN2 (and shortcut BN) backward; identity shortcut adds g into d(in)
  if (u.sc >= 0) {
    URSN_TRY(bn_back(n, u.c2, u.out.g, u.out.cs, u.out.p, u.out.cs, 1, u.sc, nullptr, 0, 0, N, s, u.jmask));
  } else {
    bool acc = take_flag(n, u.in);
    URSN_TRY(bn_back(n, u.c2, u.out.g, u.out.cs, u.out.p, u.out.cs, 1, -1, u.in.g, u.in.cs, acc ? 1 : 0, N, s, u.jmask));
  }
  BsTarget t1;   // conv2's data gradient IS d(a1): resnet_conv1's BatchNorm (no activation) consumes it
  t1.li = u.c1;
  if (u.a1_virtual) {
    Act z1 = u.a1;   // x = z1 normalised on load, dx -> a1.g
    z1.p = n->layers[u.c1].z; z1.cs = n->layers[u.c1].zcs;
    URSN_TRY(conv_bwd(n, u.c2, z1, true, N, s, nullptr, -1, false, u.c1, &t1));
  } else {
    URSN_TRY(conv_bwd(n, u.c2, u.a1, true, N, s, nullptr, -1, false, -1, &t1));
  }
  URSN_TRY(bn_back(n, u.c1, u.a1.g, u.a1.cs, nullptr, 0, 0, -1, nullptr, 0, 0, N, s, nullptr, (u.in2.C || u.sc >= 0) ? nullptr : &u.in,
                   in_target && in_target->li2 >= 0));
  const Act* in2 = u.in2.C ? &u.in2 : nullptr;
  // stride-1 shortcut next to a tiled conv1: its data gradient rides in conv1's data-gradient kernel
  bool fuse = false;
  if (u.sc >= 0 && n->layers[u.sc].stride == 1) {
    static const bool off = getenv("URSN_FUSE_SHORTCUT_DGRAD") && getenv("URSN_FUSE_SHORTCUT_DGRAD")[0] == '0';
    ursn_conv_desc d = bwd_desc(n, u.c1, u.in, N, in2, u.sc), d0 = bwd_desc(n, u.c1, u.in, N, in2, -1);
    // where conv1's dgrad runs on the all-taps implicit GEMM it carries the term itself; else only where it is tiled anyway
    fuse = !off && ((!in2 && igemm_conv_supported(d, PASS_DGRAD)) ||
                    (tiled_conv_supported(d, PASS_DGRAD) && !igemm_conv_supported(d0, PASS_DGRAD)));
  }
  // stride-2 unit on the lane-per-low-res-voxel kernel: the shortcut's data gradient (it touches the even-even-even voxels only)
  // rides in conv1's scatter-type data gradient (deconv_tiled_kernel.h, PW); URSN_FUSE_SHORTCUT_DGRAD_S2=0: the separate pass
  if (u.sc >= 0 && n->layers[u.sc].stride == 2 && !in2) {
    static const bool off2 = getenv("URSN_FUSE_SHORTCUT_DGRAD_S2") && getenv("URSN_FUSE_SHORTCUT_DGRAD_S2")[0] == '0';
    ursn_conv_desc d = bwd_desc(n, u.c1, u.in, N, nullptr, u.sc);
    fuse = !off2 && n->layers[u.sc].zcs == n->layers[u.sc].cout && tiled_deconv_supported(d, PASS_DGRAD);
  }
  // conv1's data gradient is the LAST contribution to d(in) when the shortcut is the identity (the join wrote the first) or
  // rides in the same kernel: the consumer of d(in) named by the caller gets its reductions from this launch
  const BsTarget* tin = (in_target && (u.sc < 0 || fuse)) ? in_target : nullptr;
  URSN_TRY(conv_bwd(n, u.c1, u.in, true, N, s, in2, fuse ? u.sc : -1, false, -1, tin));
  const int keep_layer = n->bs_layer, keep_blocks = n->bs_blocks;
  if (u.sc >= 0) URSN_TRY(conv_bwd(n, u.sc, u.in, !fuse, N, s, in2, -1, fuse));   // weight gradient (+ data gradient when not fused)
  if (fuse) { n->bs_layer = keep_layer; n->bs_blocks = keep_blocks; }
  return 0;
}

int backward(ursn_net* n, const float* data, int N, hipStream_t s) {
  const int ns = n->cfg.num_strides;
  for (size_t i = 0; i < n->ginit.size(); ++i) n->ginit[i] = 0;
  n->sync_used = 0;
  Layer& L2 = n->layers[n->conv2];
  URSN_TRY(bn_back(n, n->conv2, n->dlog, L2.zcs, nullptr, 0, 0, -1, nullptr, 0, 0, N, s));   // over the padded channels
  BsTarget tc;   // consumer of the gradient each data-gradient launch below completes (see BsTarget)
  tc.li = n->conv1; tc.relu = 1;
  URSN_TRY(conv_bwd(n, n->conv2, n->a_conv1, true, N, s, nullptr, -1, false, -1, &tc));
  URSN_TRY(bn_back(n, n->conv1, n->a_conv1.g, n->a_conv1.cs, nullptr, 0, 1, -1, nullptr, 0, 0, N, s, nullptr, &n->a_pre1));
  size_t ui = n->units.size();
  auto join_of = [&](const Unit& u) {
    BsTarget t;
    if (u.jmask) { t.li = u.c2; t.li2 = u.sc; t.relu = 2; t.mask = u.jmask; }
    return t;
  };
  auto act_of = [&](int li) { BsTarget t; t.li = li; t.relu = 1; return t; };
  tc = join_of(n->units[ui - 1]);
  URSN_TRY(conv_bwd(n, n->conv1, n->a_pre1, true, N, s, nullptr, -1, false, -1, &tc));
  for (int i = ns - 1; i >= 0; --i) {
    tc = join_of(n->units[ui - 2]);
    URSN_TRY(unit_bwd(n, n->units[--ui], N, s, &tc));
    tc = act_of(n->deconv[i]);   // block 0 of module1's input is the transposed conv's activation
    URSN_TRY(unit_bwd(n, n->units[--ui], N, s, &tc));
    const Act& dout = n->deconv_out[i];
    URSN_TRY(bn_back(n, n->deconv[i], dout.g, dout.cs, nullptr, 0, 1, -1, nullptr, 0, 0, N, s));
    tc = join_of(n->units[ui - 1]);   // its input: the output of the unit below (or of the bottom of the encoder)
    URSN_TRY(conv_bwd(n, n->deconv[i], n->deconv_in[i], true, N, s, nullptr, -1, false, -1, &tc));
  }
  for (int step = ns - 1; step >= 0; --step) {
    tc = join_of(n->units[ui - 2]);
    URSN_TRY(unit_bwd(n, n->units[--ui], N, s, &tc));
    tc = ui >= 2 ? join_of(n->units[ui - 2]) : act_of(n->conv0);
    URSN_TRY(unit_bwd(n, n->units[--ui], N, s, &tc));
  }
  const Act& a0 = n->a_conv0;
  URSN_TRY(bn_back(n, n->conv0, a0.g, a0.cs, nullptr, 0, 1, -1, nullptr, 0, 0, N, s));
  Act din = n->a_data;
  din.p = const_cast<float*>(data);
  URSN_TRY(conv_bwd(n, n->conv0, din, false, N, s));
  if (n->s2) {  // everything after this call on the caller's stream (Adam, the next step) sees finished gradients
    URSN_HIP(hipEventRecord(n->s2_done, n->s2));
    URSN_HIP(hipStreamWaitEvent(s, n->s2_done, 0));
  }
  return 0;
}

int read_metrics(ursn_net* n, float* out, int cnt, hipStream_t s) {
  float h[4];
  URSN_HIP(hipMemcpyAsync(h, n->metrics, sizeof(h), hipMemcpyDeviceToHost, s));
  URSN_HIP(hipStreamSynchronize(s));
  if (cnt == 3) { out[0] = h[0]; out[1] = h[1]; out[2] = h[2]; }
  else { out[0] = h[1]; out[1] = h[2]; }
  return 0;
}

int check_call(ursn_net* n, const float* data, int N) {
  URSN_REQUIRE(n, "null handle");
  URSN_REQUIRE(data, "input_data is null");
  URSN_REQUIRE(N >= 1 && N <= n->cfg.max_batch, "batch %d outside [1,%d]", N, n->cfg.max_batch);
  return 0;
}

}  // namespace

// ---- C-ABI ----------------------------------------------------------------------------------
extern "C" int ursn_query(const ursn_config* cfg, ursn_sizes* out) {
  URSN_REQUIRE(cfg && out, "query: null argument");
  URSN_REQUIRE(cfg->act_dtype == 0 || cfg->act_dtype == 1, "query: act_dtype %d not in {0 fp32, 1 bf16}", cfg->act_dtype);
  if (cfg->act_dtype == 1) return bnet_query(cfg, out);
  ursn_net tmp;
  tmp.cfg = *cfg;
  if (tmp.cfg.bn_eps <= 0.f) tmp.cfg.bn_eps = 1e-3f;
  Arena A;
  URSN_TRY(plan(&tmp, A));
  *out = tmp.sizes;
  return 0;
}

// Layer / concat tables of the plan a configuration compiles to (no device access): what ssnet_base.construct checks
// the topology recorded by _build against.
extern "C" int ursn_query_layer(const ursn_config* cfg, int64_t index, ursn_layer_info* out) {
  URSN_REQUIRE(cfg && out, "query_layer: null argument");
  if (cfg->act_dtype == 1) return bnet_layer(cfg, index, out, nullptr);
  ursn_net tmp;
  tmp.cfg = *cfg;
  Arena A;
  URSN_TRY(plan(&tmp, A));
  URSN_REQUIRE(index >= 0 && index < (int64_t)tmp.layers.size(), "query_layer: index %lld out of range", (long long)index);
  const Layer& L = tmp.layers[index];
  memset(out, 0, sizeof(*out));
  snprintf(out->name, sizeof(out->name), "%s", L.name.c_str());
  out->transposed = L.kind; out->k = L.k; out->stride = L.stride; out->cin = L.cin; out->cout = L.cout; out->relu = L.relu;
  out->w_offset = L.w_off; out->beta_offset = L.b_off;
  return 0;
}

extern "C" int ursn_query_concat(const ursn_config* cfg, int32_t step, char* first, char* second, size_t cap) {
  URSN_REQUIRE(cfg && first && second && cap > 0, "query_concat: null argument");
  ursn_net tmp;
  tmp.cfg = *cfg;
  tmp.cfg.act_dtype = 0;   // same topology and scope names in both precisions
  Arena A;
  URSN_TRY(plan(&tmp, A));
  URSN_REQUIRE(step >= 0 && step < (int)tmp.cat_names.size(), "query_concat: step %d out of range", step);
  snprintf(first, cap, "%s", tmp.cat_names[step].first.c_str());
  snprintf(second, cap, "%s", tmp.cat_names[step].second.c_str());
  return 0;
}

extern "C" int ursn_create(const ursn_config* cfg, float* params, float* grads, float* adam_m, float* adam_v,
                           void* workspace, size_t workspace_bytes, ursn_net** out) {
  URSN_REQUIRE(cfg && params && workspace && out, "create: null argument");
  URSN_REQUIRE(!cfg->trainable || (grads && adam_m && adam_v), "create: trainable net needs grads/adam buffers");
  URSN_REQUIRE((((uintptr_t)workspace) & 255) == 0, "create: workspace must be 256-byte aligned");
  URSN_REQUIRE(cfg->act_dtype == 0 || cfg->act_dtype == 1, "create: act_dtype %d not in {0 fp32, 1 bf16}", cfg->act_dtype);
  ursn_net* n = new ursn_net();
  n->cfg = *cfg;
  if (cfg->act_dtype == 1) {
    int rc = bnet_create(cfg, params, grads, workspace, workspace_bytes, &n->bf);
    if (rc) { delete n; return rc; }
    n->sizes = *bnet_sizes(n->bf);
    n->params = params; n->grads = grads; n->adam_m = adam_m; n->adam_v = adam_v;
    n->metrics = bnet_metrics(n->bf);
    *out = n;
    return 0;
  }
  if (n->cfg.bn_eps <= 0.f) n->cfg.bn_eps = 1e-3f;
  Arena A;
  A.base = (char*)workspace;
  int rc = plan(n, A);
  if (rc == 0 && (size_t)n->sizes.workspace_bytes > workspace_bytes) {
    ursn_set_error("create: workspace too small: need %lld bytes, got %zu", (long long)n->sizes.workspace_bytes,
                   workspace_bytes);
    rc = 2;
  }
  if (rc) { delete n; return rc; }
  n->params = params; n->grads = grads; n->adam_m = adam_m; n->adam_v = adam_v;
  {
    const char* e = getenv("URSN_WGRAD_STREAM");
    if (!(e && e[0] == '0')) {   // also for inference-only nets: the forward pass runs the shortcut convs on it
      int prio_lo = 0, prio_hi = 0;   // lowest priority: the dgrad / BN chain on the caller's stream is the critical path
      (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
      const char* pe = getenv("URSN_WGRAD_PRIO");
      int prio = pe ? atoi(pe) : prio_lo;
      if (hipStreamCreateWithPriority(&n->s2_owned, hipStreamNonBlocking, prio) != hipSuccess ||
          hipEventCreateWithFlags(&n->s2_done, hipEventDisableTiming) != hipSuccess) {
        ursn_set_error("create: could not create the weight-gradient stream");
        delete n;
        return 1;
      }
      n->s2 = n->s2_owned;
    }
  }
  for (const Layer& L : n->layers) {  // pad channels of z/dz are never written afterwards: keep them 0 (not NaN)
    if (L.zcs == L.cout) continue;
    size_t bytes = (size_t)n->cfg.max_batch * n->lvox[L.lout] * L.zcs * sizeof(float);
    if (hipMemset(L.mean, 0, L.zcs * sizeof(float)) != hipSuccess || hipMemset(L.rstd, 0, L.zcs * sizeof(float)) != hipSuccess ||
        hipMemset(L.z, 0, bytes) != hipSuccess || (L.dz && hipMemset(L.dz, 0, bytes) != hipSuccess)) {
      ursn_set_error("create: hipMemset of padded logits buffers failed");
      delete n;
      return 1;
    }
  }
  if (n->dlog) {  // pad lanes of the logits gradient (3|5 classes in 4|8 channels) must be finite: the BN backward runs over them
    const Layer& LL = n->layers[n->conv2];
    if (hipMemset(n->dlog, 0, (size_t)n->cfg.max_batch * n->lvox[0] * LL.zcs * sizeof(float)) != hipSuccess) {
      ursn_set_error("create: hipMemset of the logits gradient failed");
      delete n;
      return 1;
    }
  }
  *out = n;
  return 0;
}

extern "C" int ursn_destroy(ursn_net* net) {
  if (net) for (auto& gph : net->graphs) if (gph.exec) (void)hipGraphExecDestroy(gph.exec);
  if (net && net->gstream) { (void)hipStreamSynchronize(net->gstream); (void)hipStreamDestroy(net->gstream); }
  if (net && net->g_in) (void)hipEventDestroy(net->g_in);
  if (net && net->g_out) (void)hipEventDestroy(net->g_out);
  if (!net) return 0;
  if (net->bf) { bnet_destroy(net->bf); delete net; return 0; }
  if (net->s2_owned) { (void)hipStreamSynchronize(net->s2_owned); (void)hipStreamDestroy(net->s2_owned); }
  if (net->s2_done) (void)hipEventDestroy(net->s2_done);
  for (hipEvent_t e : net->sync_pool) (void)hipEventDestroy(e);
  for (hipEvent_t e : net->fwd_pool) (void)hipEventDestroy(e);
  for (hipEvent_t e : net->ev_pool) (void)hipEventDestroy(e);
  delete net;
  return 0;
}

extern "C" int ursn_get_sizes(const ursn_net* net, ursn_sizes* out) {
  URSN_REQUIRE(net && out, "get_sizes: null argument");
  *out = net->sizes;
  return 0;
}

extern "C" int ursn_param(const ursn_net* net, int64_t index, ursn_param_info* out) {
  URSN_REQUIRE(net && out, "param: null argument");
  if (net->bf) return bnet_param(net->bf, index, out);
  URSN_REQUIRE(index >= 0 && index < net->sizes.n_tensors, "param: index %lld out of range", (long long)index);
  const Layer& L = net->layers[index / 2];
  memset(out, 0, sizeof(*out));
  if (index % 2 == 0) {
    snprintf(out->name, sizeof(out->name), "%s/weights", L.name.c_str());
    out->offset = L.w_off; out->nelem = L.w_n; out->rank = net->cfg.ndim + 2;
    for (int j = 0; j < net->cfg.ndim; ++j) out->shape[j] = L.k;
    out->shape[net->cfg.ndim] = L.kind ? L.cout : L.cin;
    out->shape[net->cfg.ndim + 1] = L.kind ? L.cin : L.cout;
  } else {
    snprintf(out->name, sizeof(out->name), "%s/BatchNorm/beta", L.name.c_str());
    out->offset = L.b_off; out->nelem = L.cout; out->rank = 1; out->shape[0] = L.cout;
  }
  return 0;
}

extern "C" int ursn_zero_grad(ursn_net* net, void* stream) {
  URSN_REQUIRE(net && net->grads, "zero_grad: net is not trainable");
  URSN_HIP(hipMemsetAsync(net->grads, 0, (size_t)net->sizes.n_params * sizeof(float), (hipStream_t)stream));
  return 0;
}

extern "C" int ursn_accum_step(ursn_net* net, const float* data, const float* label, const float* weight, int32_t n,
                               float* out3, void* stream) {
  URSN_TRY(check_call(net, data, n));
  URSN_REQUIRE(net->cfg.trainable && net->grads, "accum_step: net constructed with trainable=False");
  URSN_REQUIRE(label, "accum_step: input_label is null");
  URSN_REQUIRE(!net->cfg.use_weight || weight, "Network configured to use loss pixel-weighting. Cannot run w/ input_weight=None");
  hipStream_t s = (hipStream_t)stream;
  net->last_n = n;
  if (net->bf) {
    URSN_TRY(bnet_step(net->bf, data, label, weight, n, 0, nullptr, nullptr, s));
    if (out3) URSN_TRY(read_metrics(net, out3, 3, s));
    return 0;
  }
  // hipGraph replay (opt-in experiment).  cfg1 (2-D 256^2 x 4) issues ~560 launches per 5.7 ms step and the host needs 5.5 ms to
  // enqueue them (tools/host_enqueue_probe.py), so the step looked launch-bound.  With URSN_GRAPH >= 1 the third call with the same
  // buffers and batch captures the step's launches -- both streams, the events between them -- into a hipGraph on a stream of the
  // library's own, and later calls are ONE graph launch.  Measured on ROCm 7.2: the replay takes 10.7 ms per step against 5.7 ms
  // of ordinary launches (the runtime orders the nodes more strictly than the two streams do), so it is OFF by default.
  // URSN_GRAPH=1: when the batch holds <= 2^20 voxels, 2: always.  Never while profiling (events per launch).
  static const int gmode = getenv("URSN_GRAPH") ? atoi(getenv("URSN_GRAPH")) : 0;
  bool geligible = gmode > 0 && !net->profile && !net->graph_broken && !ursn_roctx_on() &&
                   (gmode > 1 || (int64_t)n * net->lvox[0] <= ((int64_t)1 << 20));
  auto run_on = [&](hipStream_t rs) -> int {
    URSN_TRY(forward(net, data, n, rs));
    URSN_TRY(head(net, data, label, net->cfg.use_weight ? weight : nullptr, n, nullptr, true, rs));
    URSN_TRY(backward(net, data, n, rs));
    return 0;
  };
  auto run = [&]() -> int { return run_on(s); };
  if (geligible && !net->gstream) {
    if (hipStreamCreateWithFlags(&net->gstream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&net->g_in, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&net->g_out, hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();
      net->graph_broken = true;
    }
  }
  auto graph_launch = [&](hipGraphExec_t ex) -> int {   // caller's stream -> graph stream -> caller's stream
    URSN_HIP(hipEventRecord(net->g_in, s));
    URSN_HIP(hipStreamWaitEvent(net->gstream, net->g_in, 0));
    URSN_HIP(hipGraphLaunch(ex, net->gstream));
    URSN_HIP(hipEventRecord(net->g_out, net->gstream));
    URSN_HIP(hipStreamWaitEvent(s, net->g_out, 0));
    return 0;
  };
  geligible = geligible && !net->graph_broken;
  if (geligible) {
    ursn_net::StepGraph* sg = nullptr;
    for (auto& gph : net->graphs)
      if (gph.data == data && gph.label == label && gph.weight == weight && gph.n == n) sg = &gph;
    if (!sg) {
      if (net->graphs.size() >= 4) {   // forget the oldest
        if (net->graphs[0].exec) (void)hipGraphExecDestroy(net->graphs[0].exec);
        net->graphs.erase(net->graphs.begin());
      }
      net->graphs.push_back({data, label, weight, n, 0, nullptr});
      sg = &net->graphs.back();
    }
    if (sg->exec) {
      URSN_TRY(graph_launch(sg->exec));
    } else if (++sg->seen >= 3) {
      hipGraph_t graph = nullptr;
      bool ok = hipStreamBeginCapture(net->gstream, hipStreamCaptureModeRelaxed) == hipSuccess;
      int rc = ok ? run_on(net->gstream) : 1;
      if (ok) ok = hipStreamEndCapture(net->gstream, &graph) == hipSuccess && graph != nullptr;
      if (ok && rc == 0) ok = hipGraphInstantiate(&sg->exec, graph, nullptr, nullptr, 0) == hipSuccess;
      if (graph) (void)hipGraphDestroy(graph);
      if (getenv("URSN_GRAPH_DEBUG")) fprintf(stderr, "ursn graph capture: ok=%d rc=%d\n", (int)ok, rc);
      if (ok && rc == 0) {
        URSN_TRY(graph_launch(sg->exec));
      } else {   // nothing of the captured step ran: give up on graphs and run it the ordinary way
        (void)hipGetLastError();
        sg->exec = nullptr;
        net->graph_broken = true;
        URSN_TRY(run());
      }
    } else {
      URSN_TRY(run());
    }
  } else {
    URSN_TRY(run());
  }
  if (out3) URSN_TRY(read_metrics(net, out3, 3, s));
  return 0;
}

extern "C" int ursn_apply_adam(ursn_net* net, float lr, void* stream) {
  URSN_REQUIRE(net && net->grads, "apply_adam: net is not trainable");
  const double b1 = 0.9, b2 = 0.999;
  if (lr <= 0.f) lr = 1e-3f;  // tf.train.AdamOptimizer() default (lib/ssnet.py:72-73)
  net->adam_t += 1;
  double lr_t = (double)lr * sqrt(1.0 - pow(b2, (double)net->adam_t)) / (1.0 - pow(b1, (double)net->adam_t));
  return launch_adam(net->params, net->grads, net->adam_m, net->adam_v, net->sizes.n_params, (float)lr_t, (float)b1,
                     (float)b2, 1e-8f, (hipStream_t)stream);
}

extern "C" int ursn_eval(ursn_net* net, const float* data, const float* label, const float* weight, int32_t n,
                         float* out3, void* stream) {
  URSN_TRY(check_call(net, data, n));
  URSN_REQUIRE(label, "eval: input_label is null");
  URSN_REQUIRE(!net->cfg.use_weight || weight, "Network configured to use loss pixel-weighting. Cannot run w/ input_weight=None");
  hipStream_t s = (hipStream_t)stream;
  net->last_n = n;
  if (net->bf) URSN_TRY(bnet_step(net->bf, data, label, weight, n, 1, nullptr, nullptr, s));
  else {
    URSN_TRY(forward(net, data, n, s));
    URSN_TRY(head(net, data, label, net->cfg.use_weight ? weight : nullptr, n, nullptr, false, s));
  }
  if (out3) URSN_TRY(read_metrics(net, out3, 3, s));
  return 0;
}

extern "C" int ursn_infer(ursn_net* net, const float* data, const float* label, int32_t n, float* softmax_out,
                          float* out2, void* stream) {
  URSN_TRY(check_call(net, data, n));
  URSN_REQUIRE(softmax_out, "infer: softmax_out is null");
  hipStream_t s = (hipStream_t)stream;
  net->last_n = n;
  if (net->bf) URSN_TRY(bnet_step(net->bf, data, label, nullptr, n, 2, softmax_out, nullptr, s));
  else {
    URSN_TRY(forward(net, data, n, s));
    URSN_TRY(head(net, data, label, nullptr, n, softmax_out, false, s));
  }
  if (label && out2) URSN_TRY(read_metrics(net, out2, 2, s));
  else URSN_HIP(hipStreamSynchronize(s));
  return 0;
}

extern "C" int ursn_infer_labels(ursn_net* net, const float* data, const float* label, int32_t n, float* labels_out,
                                 float* softmax_out, float* out2, void* stream) {
  URSN_TRY(check_call(net, data, n));
  URSN_REQUIRE(labels_out, "infer_labels: labels_out is null");
  URSN_REQUIRE(net->cfg.num_class >= 3 && net->cfg.cin == 1, "infer_labels: needs >= 3 classes and one input channel");
  hipStream_t s = (hipStream_t)stream;
  net->last_n = n;
  if (net->bf) URSN_TRY(bnet_step(net->bf, data, label, nullptr, n, 2, softmax_out, labels_out, s));
  else {
    URSN_TRY(forward(net, data, n, s));
    URSN_TRY(head(net, data, label, nullptr, n, softmax_out, false, s, labels_out));
  }
  if (label && out2) URSN_TRY(read_metrics(net, out2, 2, s));
  else URSN_HIP(hipStreamSynchronize(s));
  return 0;
}

extern "C" int ursn_read_metrics(ursn_net* net, float* out3, void* stream) {
  URSN_REQUIRE(net && out3, "read_metrics: null argument");
  return read_metrics(net, out3, 3, (hipStream_t)stream);
}

extern "C" int ursn_get_adam_step(const ursn_net* net, int64_t* t) {
  URSN_REQUIRE(net && t, "null argument");
  *t = net->adam_t;
  return 0;
}
extern "C" int ursn_set_adam_step(ursn_net* net, int64_t t) {
  URSN_REQUIRE(net && t >= 0, "bad argument");
  net->adam_t = t;
  return 0;
}

extern "C" int ursn_tensor(const ursn_net* net, const char* name, float** ptr, int64_t* voxels, int32_t* channels,
                           int32_t* cstride) {
  URSN_REQUIRE(net && name && ptr && voxels && channels && cstride, "tensor: null argument");
  if (net->bf) return bnet_tensor(net->bf, name, (void**)ptr, voxels, channels, cstride);
  std::string s(name);
  bool want_z = false, want_g = false, want_dz = false, want_mean = false, want_rstd = false;
  auto strip = [&](const char* suf, bool& f) {
    size_t L = strlen(suf);
    if (s.size() > L && s.compare(s.size() - L, L, suf) == 0) { f = true; s = s.substr(0, s.size() - L); }
  };
  strip(":dz", want_dz);
  strip(":z", want_z);
  strip(":grad", want_g);
  strip(":mean", want_mean);
  strip(":rstd", want_rstd);
  if (want_g && s == "logits") {   // d loss / d logits as the head wrote it (input of conv2's BatchNorm backward)
    URSN_REQUIRE(net->dlog, "tensor: net is not trainable");
    const Layer& L = net->layers[net->conv2];
    *ptr = net->dlog; *voxels = net->lvox[0]; *channels = L.cout; *cstride = L.zcs;
    return 0;
  }
  if (want_z || want_dz || want_mean || want_rstd) {
    auto it = net->named_z.find(s);
    URSN_REQUIRE(it != net->named_z.end(), "tensor: no layer named %s", s.c_str());
    const Layer& L = net->layers[it->second];
    *ptr = want_z ? L.z : want_dz ? L.dz : want_mean ? L.mean : L.rstd;
    URSN_REQUIRE(*ptr, "tensor: %s does not exist (net is not trainable)", name);
    *voxels = (want_mean || want_rstd) ? 0 : net->lvox[L.lout];   // 0: a per-channel fp32 vector, not a per-voxel tensor
    *channels = L.cout;
    *cstride = L.zcs;
    return 0;
  }
  auto it = net->named.find(s);
  URSN_REQUIRE(it != net->named.end(), "tensor: no activation named %s", s.c_str());
  *ptr = want_g ? it->second.g : it->second.p;
  URSN_REQUIRE(*ptr, "tensor: %s is not materialised (normalised on load inside the consuming convolution, URSN_NORM_ON_LOAD=1)", name);
  *voxels = net->lvox[it->second.lvl];
  *channels = it->second.C;
  *cstride = it->second.cs;
  return 0;
}

// ---- profiling C-ABI --------------------------------------------------------------------------
extern "C" int ursn_profile_enable(ursn_net* net, int32_t on) {
  URSN_REQUIRE(net, "null handle");
  if (net->bf) return bnet_profile_enable(net->bf, on);
  net->profile = on != 0;
  net->prof.clear();
  net->ev_used = 0;
  return 0;
}

extern "C" int ursn_profile_read(ursn_net* net, ursn_prof_rec* out, int64_t max_recs, int64_t* n_out) {
  URSN_REQUIRE(net && n_out, "null argument");
  if (net->bf) return bnet_profile_read(net->bf, out, max_recs, n_out);
  int64_t cnt = 0;
  for (size_t i = 0; i < net->prof.size() && (!out || cnt < max_recs); ++i) {   // out == NULL: only counts, all of them
    const ursn_net::ProfRec& r = net->prof[i];
    float ms = 0.f;
    if (hipEventSynchronize(r.e1) != hipSuccess || hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) continue;
    if (out) {
      ursn_prof_rec& o = out[cnt];
      memset(&o, 0, sizeof(o));
      snprintf(o.kernel, sizeof(o.kernel), "%s", r.kernel);
      snprintf(o.layer, sizeof(o.layer), "%s", net->layers[r.layer].name.c_str());
      o.pass = r.pass; o.ms = ms; o.flops = r.flops; o.bytes = r.bytes; o.launches = r.launches;
    }
    ++cnt;
  }
  *n_out = cnt;
  if (out) { net->prof.clear(); net->ev_used = 0; }
  return 0;
}

extern "C" int ursn_set_wgrad_overlap(ursn_net* net, int32_t on) {
  URSN_REQUIRE(net, "null handle");
  if (net->bf) return bnet_set_wgrad_overlap(net->bf, on);
  if (net->s2_owned) (void)hipStreamSynchronize(net->s2_owned);
  net->s2 = on ? net->s2_owned : nullptr;
  return 0;
}
