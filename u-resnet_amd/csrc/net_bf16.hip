// bf16 mixed-precision plan of the U-ResNet of lib/uresnet.py:22-123 + lib/resnet_module.py:10-87 (BASELINE.json configs[4]):
// activations, raw conv outputs z and all gradient tensors in HBM as bf16, fp32 master weights / BatchNorm statistics /
// gradient accumulators / Adam.  Same topology, variable layout and fetch-set semantics as net.hip; every conv-like pass runs
// on the two kernels of bf16_conv.hip, BatchNorm / join / head on bf16_elementwise.hip.
//   * channel counts are padded to multiples of 8 in memory (one voxel = whole 16-byte pieces): the single input channel
//     is expanded to 8 (7 zero channels) by launch_bf16_input, the 3|5-class logits live in 8-channel pieces; the padded
//     rows / columns of those two layers' weights do not exist in the parameter buffer (Kw / Nw of launch_bconv);
//   * tf.concat([deconv_i, skip]) is never executed: both producers write channel slices of one buffer;
//   * weight gradients run on a second stream beside the data-gradient / BatchNorm chain (they only feed the optimiser).
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "bf16_common.h"
#include "bf16_pack.h"
#include <functional>
#include "net_bf16.h"

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
};

struct BAct {              // activation view + its gradient view (same layout)
  bf16_t* p = nullptr;
  bf16_t* g = nullptr;
  int C = 0, cs = 0, lvl = 0, flag = -1;
  // virtual activation (never written): value = [relu] bn(z of layer aff_layer), applied by the consuming kernels while they
  // stage it (B3Affine); p stays null
  int aff_layer = -1, aff_relu = 0;
  const float* in_f32 = nullptr;   // the network input, read as one fp32 channel per voxel by conv0's kernels (set per call)
};

struct BLayer {
  std::string name;
  int kind = 0, k = 3, stride = 1, cin = 0, cout = 0, lin = 0, lout = 0;   // real channel counts (parameter tensor)
  int kin = 0, kout = 0;                                                   // channel counts in memory (multiples of 8)
  int64_t w_off = 0, b_off = 0, w_n = 0;
  ursn_conv_desc desc;                                                     // memory-view channels
  bf16_t *z = nullptr, *dz = nullptr;
  float *mean = nullptr, *rstd = nullptr;
  // packed weights of the forward pass [0] and the data gradient [1]: the layer's OWN buffers (one per parity class where the
  // pass still runs as several launches), so that one launch at the start of a step can fill them all (bf16_pack.h)
  bf16_t* wp[2] = {nullptr, nullptr};
  size_t wp_elems[2] = {0, 0}, wp_stride[2] = {0, 0};
};

struct BUnit {
  int sc = -1, c1 = -1, c2 = -1;
  BAct in, a1, out;
  unsigned char* jmask = nullptr;   // relu mask of the join (one byte per 16-byte piece): both BatchNorm-backward passes read it instead of out
};

}  // namespace

struct BProfRec { int layer, pass; const char* kernel; double flops, bytes; hipEvent_t e0, e1; long l0; int launches; };

struct ursn_bnet {
  bool profile = false, s2_on = true;
  std::vector<BProfRec> prof;
  std::vector<hipEvent_t> pev;
  size_t pev_used = 0;
  // BatchNorm-backward reductions taken in the epilogue of the data gradient that completed the layer's output gradient
  double* bs_scratch = nullptr;
  int bs_layer = -1, bs_blocks = 0;
  ursn_config cfg;
  ursn_sizes sizes;
  int nlev = 0;
  int ldim[8][3];
  int64_t lvox[8];
  std::vector<BLayer> layers;
  std::vector<BUnit> units;
  std::vector<int> deconv;
  std::vector<BAct> deconv_in, deconv_out, cat;
  int conv0 = -1, conv1 = -1, conv2 = -1;
  BAct a_data, a_conv0, a_conv1, a_pre1;
  // F = 8: conv0's activation lives in its own contiguous tensor; the level-0 concat voxel (16 channels) is written whole
  // when the last transposed conv's BatchNorm runs (its half from z, the skip half recomputed from conv0's z): 16-byte
  // halves written one tensor pass apart were partial-sector writes (1.0 ms instead of 0.36 per pass at 256^3 x 4)
  bool skip0_own = false;
  bool scalar_in = false;   // conv0 (forward and weight gradient) reads the fp32 input directly: no 8-channel bf16 copy of it
  // ... and the gradient of that concat voxel is produced as two 8-channel tensors (the transposed conv's half and the skip's
  // half have different consumers; read as 16-byte halves of a 32-byte voxel each cost a full pass of the other half)
  bf16_t* dec0_g = nullptr; bf16_t* skip0_g = nullptr;
  bool skip0_merge = false;
  bool split0_done = false;   // this backward pass produced the two tensors (the fused 8 -> 16 data gradient ran)
  std::vector<int> ginit;
  float *params = nullptr, *grads = nullptr;
  bf16_t* dlog = nullptr;
  float* metrics = nullptr;
  float* beta_pad = nullptr;     // conv2's beta padded to 8 (the head and the BatchNorm kernels index 8 channels)
  BPackCtx pack; int pack_N = -1;   // the step's weight packing in one launch (bf16_pack.h)
  bf16_t* wpack2 = nullptr;      // packed weights of the launches on the second stream (none today: wgrad packs nothing)
  double* stats = nullptr; size_t stats_doubles = 0;
  void* bn_scratch = nullptr;
  void* head_scratch = nullptr;
  void* wg_scratch = nullptr; size_t wg_bytes = 0;
  double* stats2 = nullptr;
  std::vector<std::function<int()>> deferred;   // weight-gradient launches held back while the decoder is at level 0
  bool defer_on = true, defer_open = false;
  hipStream_t s2 = nullptr;
  hipEvent_t s2_done = nullptr;
  std::vector<hipEvent_t> evs;
  size_t ev_used = 0;
  std::map<std::string, BAct> named;   // debug lookup (ursn_tensor): materialised activations and their gradient tensors
};

namespace {

int new_flag(ursn_bnet* n) { n->ginit.push_back(0); return (int)n->ginit.size() - 1; }

BAct make_act(ursn_bnet* n, Arena& A, int lvl, int C, bool grad, bool value = true) {
  BAct a;
  a.C = C; a.cs = C; a.lvl = lvl;
  const size_t bytes = (size_t)n->cfg.max_batch * n->lvox[lvl] * C * sizeof(bf16_t);
  a.p = value ? (bf16_t*)A.take(bytes) : nullptr;
  a.g = grad ? (bf16_t*)A.take(bytes) : nullptr;
  a.flag = new_flag(n);
  return a;
}
BAct sub_act(const BAct& full, int c0, int C) {
  BAct a = full;
  a.p = full.p ? full.p + c0 : nullptr;
  a.g = full.g ? full.g + c0 : nullptr;
  a.C = C;
  return a;
}
int pad8(int c) { return (c + 7) & ~7; }

int add_layer(ursn_bnet* n, Arena& A, const std::string& name, int kind, int k, int s, int ci, int co, int lin, int lout,
              int64_t& poff) {
  BLayer L;
  L.name = "UResNet/" + name;
  L.kind = kind; L.k = k; L.stride = s; L.cin = ci; L.cout = co; L.lin = lin; L.lout = lout;
  L.kin = pad8(ci); L.kout = pad8(co);
  int64_t taps = 1;
  for (int j = 0; j < n->cfg.ndim; ++j) taps *= k;
  L.w_n = taps * ci * co;
  L.w_off = poff; poff += L.w_n;
  L.b_off = poff; poff += co;
  memset(&L.desc, 0, sizeof(L.desc));
  L.desc.ndim = n->cfg.ndim;
  L.desc.n = n->cfg.max_batch;
  for (int j = 0; j < n->cfg.ndim; ++j) L.desc.in_sp[j] = n->ldim[lin][3 - n->cfg.ndim + j];
  L.desc.cin = L.kin; L.desc.cout = L.kout; L.desc.k = k; L.desc.stride = s; L.desc.transposed = kind; L.desc.dtype = 1;
  const size_t bytes = (size_t)n->cfg.max_batch * n->lvox[lout] * L.kout * sizeof(bf16_t);
  L.z = (bf16_t*)A.take(bytes);
  L.dz = n->cfg.trainable ? (bf16_t*)A.take(bytes) : nullptr;
  L.mean = (float*)A.take(L.kout * sizeof(float));
  L.rstd = (float*)A.take(L.kout * sizeof(float));
  n->layers.push_back(L);
  return (int)n->layers.size() - 1;
}

// geometry of one pass of a layer with the weight strides of the STORED (unpadded) parameter tensor
int layer_geoms(const ursn_bnet* n, const BLayer& L, ConvPass pass, int N, int in_cs, int out_cs, GatherGeom* g8) {
  ursn_conv_desc d = L.desc;
  d.n = N; d.in_cstride = in_cs; d.out_cstride = out_cs;
  const int cnt = build_geoms(d, pass, g8);
  const bool gather = (!L.kind && pass == PASS_FWD) || (L.kind && pass == PASS_DGRAD);
  for (int i = 0; i < cnt; ++i) {
    GatherGeom& g = g8[i];
    if (pass == PASS_WGRAD || gather) {   // [t][K][N] natural
      const int Kr = pass == PASS_WGRAD ? (L.kind ? L.cout : L.cin) : (pass == PASS_FWD ? L.cin : L.cout);
      const int Nr = pass == PASS_WGRAD ? (L.kind ? L.cin : L.cout) : (pass == PASS_FWD ? L.cout : L.cin);
      g.w_tap_stride = Kr * Nr; g.w_sk = Nr; g.w_sn = 1;
    } else {                              // contraction along the last stored axis
      g.w_tap_stride = L.cin * L.cout; g.w_sk = 1;
      g.w_sn = pass == PASS_DGRAD ? L.cout : L.cin;   // conv dgrad contracts cout, transposed forward contracts cin
    }
  }
  return cnt;
}
void real_extents(const BLayer& L, ConvPass pass, int& Kw, int& Nw) {
  if (pass == PASS_WGRAD) { Kw = L.kind ? L.cout : L.cin; Nw = L.kind ? L.cin : L.cout; }
  else if (pass == PASS_FWD) { Kw = L.cin; Nw = L.cout; }
  else { Kw = L.cout; Nw = L.cin; }
}

int plan(ursn_bnet* n, Arena& A) {
  const ursn_config& c = n->cfg;
  URSN_REQUIRE(c.ndim == 2 || c.ndim == 3, "len(dims) must be 3 (H,W,C) or 4 (H,W,D,C)");
  URSN_REQUIRE(c.num_strides >= 1 && c.num_strides <= 5, "num_strides %d out of range [1,5]", c.num_strides);
  URSN_REQUIRE(c.cin == 1, "bf16 path: one input channel (the reference's data), got %d", c.cin);
  URSN_REQUIRE(c.base_filters >= 8 && c.base_filters % 8 == 0, "bf16 path: base_num_outputs must be a multiple of 8, got %d", c.base_filters);
  URSN_REQUIRE(c.num_class >= 1 && c.num_class <= 8 && c.max_batch >= 1, "bad class / batch configuration");
  const int ns = c.num_strides, F = c.base_filters;
  const bool tr = c.trainable != 0;
  n->nlev = ns + 1;
  for (int l = 0; l <= ns; ++l) {
    n->lvox[l] = 1;
    for (int j = 0; j < 3; ++j) {
      const int lead = 3 - c.ndim;
      if (j < lead) { n->ldim[l][j] = 1; continue; }
      const int s0 = c.spatial[j - lead];
      URSN_REQUIRE(s0 > 0 && s0 % (1 << ns) == 0, "spatial size %d not divisible by 2^%d (deconv/skip shapes would differ)", s0, ns);
      n->ldim[l][j] = s0 >> l;
      n->lvox[l] *= n->ldim[l][j];
    }
  }
  int64_t poff = 0;
  n->layers.clear(); n->units.clear(); n->deconv.clear(); n->cat.clear(); n->deconv_in.clear(); n->deconv_out.clear();
  n->ginit.clear(); n->named.clear();
  n->cat.resize(ns);
  for (int i = 0; i < ns; ++i) n->cat[i] = make_act(n, A, ns - 1 - i, 2 * (F << (ns - 1 - i)), tr);
  auto fmap_view = [&](int lvl) { const BAct& full = n->cat[ns - 1 - lvl]; return sub_act(full, full.C / 2, full.C / 2); };

  n->a_data = make_act(n, A, 0, 8, false);
  n->conv0 = add_layer(n, A, "conv0", 0, 3, 1, c.cin, F, 0, 0, poff);
  {
    GatherGeom g[8];
    const BLayer& L0 = n->layers[n->conv0];
    n->scalar_in = c.cin == 1 && !(getenv("URSN_BF16_SCALAR_IN") && getenv("URSN_BF16_SCALAR_IN")[0] == '0') &&
                   layer_geoms(n, L0, PASS_FWD, c.max_batch, L0.kin, L0.kout, g) == 1 && b3conv_ok(g[0]) && g[0].K == 8 &&
                   (!tr || (layer_geoms(n, L0, PASS_WGRAD, c.max_batch, L0.kin, L0.kout, g) == 1 && b3wgrad_scalar_ok(g[0])));
  }
  n->skip0_own = F == 8 && !(getenv("URSN_BF16_SKIP0_OWN") && getenv("URSN_BF16_SKIP0_OWN")[0] == '0');
  // ... and the skip's half IS the gradient tensor of conv0's activation: the encoder's data gradients (module0, run later in
  // the backward pass) accumulate into it, so conv0's BatchNorm backward reads one gradient tensor in each of its two passes
  // instead of two (URSN_BF16_SKIP0_MERGE=0: a tensor of its own for the encoder's share)
  n->skip0_merge = n->skip0_own && tr && !(getenv("URSN_BF16_SKIP0_MERGE") && getenv("URSN_BF16_SKIP0_MERGE")[0] == '0');
  n->a_conv0 = n->skip0_own ? make_act(n, A, 0, F, tr && !n->skip0_merge) : fmap_view(0);
  if (n->skip0_own && tr) {
    const size_t bytes = (size_t)c.max_batch * n->lvox[0] * 8 * sizeof(bf16_t);
    n->dec0_g = (bf16_t*)A.take(bytes);
    n->skip0_g = (bf16_t*)A.take(bytes);
    if (n->skip0_merge) n->a_conv0.g = n->skip0_g;
  }
  // can the consumer layer L (k3 s1, C -> C) apply its producer's BatchNorm while staging (forward and weight gradient)?
  auto virtual_ok = [&](const BLayer& L) {
    static const bool off = getenv("URSN_BF16_NORM_ON_LOAD") && getenv("URSN_BF16_NORM_ON_LOAD")[0] == '0';
    if (off) return false;
    GatherGeom g[8];
    if (layer_geoms(n, L, PASS_FWD, c.max_batch, L.kin, L.kout, g) != 1 || !b3conv_aff_ok(g[0])) return false;
    if (tr && (layer_geoms(n, L, PASS_WGRAD, c.max_batch, L.kin, L.kout, g) != 1 || !b3wgrad_ok(g[0]))) return false;
    return true;
  };
  auto add_unit = [&](const std::string& scope, const BAct& in, int co, int s, int lout, const BAct* out_view) {
    BUnit u;
    u.in = in;
    if (!(in.C == co && s == 1)) u.sc = add_layer(n, A, scope + "/shortcut", 0, 1, s, in.C, co, in.lvl, lout, poff);
    u.c1 = add_layer(n, A, scope + "/resnet_conv1", 0, 3, s, in.C, co, in.lvl, lout, poff);
    u.c2 = add_layer(n, A, scope + "/resnet_conv2", 0, 3, 1, co, co, lout, lout, poff);
    const bool virt = virtual_ok(n->layers[u.c2]);   // resnet_conv1's BatchNorm output feeds resnet_conv2 only
    u.a1 = make_act(n, A, lout, co, tr, !virt);
    if (virt) { u.a1.aff_layer = u.c1; u.a1.aff_relu = 0; }
    u.out = out_view ? *out_view : make_act(n, A, lout, co, tr);
    if (tr && !(getenv("URSN_BF16_RELU_MASK") && getenv("URSN_BF16_RELU_MASK")[0] == '0'))
      u.jmask = (unsigned char*)A.take((size_t)c.max_batch * n->lvox[lout] * (pad8(co) / 8) + 256);
    n->units.push_back(u);
    n->named["UResNet/" + scope] = u.out;
    n->named[n->layers[u.c1].name] = u.a1;
    return u.out;
  };
  n->named["UResNet/conv0"] = n->a_conv0;
  BAct net = n->a_conv0;
  char sc[64];
  for (int step = 0; step < ns; ++step) {
    const int co = net.C * 2;
    snprintf(sc, sizeof(sc), "resnet_module%d/module1", step);
    BAct u1 = add_unit(sc, net, co, 2, step + 1, nullptr);
    snprintf(sc, sizeof(sc), "resnet_module%d/module2", step);
    if (step + 1 < ns) { BAct view = fmap_view(step + 1); net = add_unit(sc, u1, co, 1, step + 1, &view); }
    else net = add_unit(sc, u1, co, 1, step + 1, nullptr);
  }
  for (int i = 0; i < ns; ++i) {
    const int co = net.C / 2, lvl = ns - 1 - i;
    snprintf(sc, sizeof(sc), "deconv%d", i);
    const int li = add_layer(n, A, sc, 1, 3, 2, net.C, co, net.lvl, lvl, poff);
    n->deconv.push_back(li);
    n->deconv_in.push_back(net);
    n->deconv_out.push_back(sub_act(n->cat[i], 0, co));
    n->named[n->layers[li].name] = n->deconv_out.back();
    snprintf(sc, sizeof(sc), "resnet_module%d/module1", i + 5);
    BAct u1 = add_unit(sc, n->cat[i], co, 1, lvl, nullptr);
    snprintf(sc, sizeof(sc), "resnet_module%d/module2", i + 5);
    net = add_unit(sc, u1, co, 1, lvl, nullptr);
  }
  n->a_pre1 = net;
  n->conv1 = add_layer(n, A, "conv1", 0, 3, 1, net.C, F, 0, 0, poff);
  n->conv2 = add_layer(n, A, "conv2", 0, 3, 1, F, c.num_class, 0, 0, poff);
  {
    const bool virt = virtual_ok(n->layers[n->conv2]);   // conv1's activation feeds conv2 only (lib/uresnet.py:84-100)
    n->a_conv1 = make_act(n, A, 0, F, tr, !virt);
    if (virt) { n->a_conv1.aff_layer = n->conv1; n->a_conv1.aff_relu = 1; }
    n->named["UResNet/conv1"] = n->a_conv1;
  }

  const int64_t V0 = (int64_t)c.max_batch * n->lvox[0];
  n->dlog = tr ? (bf16_t*)A.take((size_t)V0 * 8 * sizeof(bf16_t)) : nullptr;
  n->metrics = (float*)A.take(8 * sizeof(float));
  n->beta_pad = (float*)A.take(8 * sizeof(float));
  n->head_scratch = A.take(head_scratch_bytes(c.max_batch, n->lvox[0]) + 64);
  size_t st = 0, bn = 0, wg = 0;
  for (BLayer& L : n->layers) {
    const size_t b = bbn_scratch_bytes((int64_t)c.max_batch * n->lvox[L.lout], L.kout);
    if (b > bn) bn = b;
    for (int nb = 1; nb <= c.max_batch; ++nb) {   // the launch geometry is chosen per call from the batch actually fed
      GatherGeom g[8];
      for (int pass = 0; pass < (tr ? 3 : 1); ++pass) {
        const int cnt = layer_geoms(n, L, (ConvPass)pass, nb, L.kin, L.kout, g);
        URSN_REQUIRE(cnt >= 1, "bf16 plan: bad geometry for %s", L.name.c_str());
        size_t stl = 0, wpl = 0, wps = 0;   // this pass: statistics doubles, packed elements (all classes), class stride
        if (pass != PASS_WGRAD && bdeconv_ok(g, cnt)) {
          wpl = bdeconv_pack_elems();
          if (pass == PASS_FWD) stl = (size_t)bdeconv_grid_blocks(g, cnt) * 32;
        }
        if (pass != PASS_WGRAD && bsconv_ok(g, cnt)) {
          wpl = bsconv_pack_elems(g, cnt);
          if (pass == PASS_FWD) stl = bsconv_stats_scratch_doubles(g, cnt);
        }
        for (int i = 0; i < cnt; ++i) {
          if (g[i].ntaps == 0) continue;
          if (pass == PASS_WGRAD) {
            size_t w = bwgrad_scratch_bytes(g[i]);
            if (&L == &n->layers[n->conv0] && b0wgrad_ok(g[i]) && b0wgrad_scratch_bytes(g[i]) > w) w = b0wgrad_scratch_bytes(g[i]);
            URSN_REQUIRE(w > 0, "bf16 plan: no weight-gradient kernel for %s", L.name.c_str());
            if (w > wg) wg = w;
          } else {
            const size_t e = (bconv_pack_elems(g[i]) + 16 + 127) & ~(size_t)127;   // (+ the zero piece; 256-byte aligned classes)
            URSN_REQUIRE(e > 16, "bf16 plan: no conv kernel for %s (pass %d)", L.name.c_str(), pass);
            if (e > wps) wps = e;
            if (pass == PASS_FWD) stl += bconv_stats_scratch_doubles(g[i]);
            if (cnt == 1 && bs2k8_ok(g[i])) {
              if (pass == PASS_FWD && (size_t)bs2k8_grid_blocks(g[i]) * 64 > stl) stl = (size_t)bs2k8_grid_blocks(g[i]) * 64;
              if (((bs2k8_pack_elems() + 127) & ~(size_t)127) > wps) wps = (bs2k8_pack_elems() + 127) & ~(size_t)127;
            }
            if (pass == PASS_FWD && &L == &n->layers[n->conv0] && b0conv_ok(g[i])) {
              if ((size_t)b0conv_grid_blocks(g[i]) * 32 > stl) stl = (size_t)b0conv_grid_blocks(g[i]) * 32;
              if (b0conv_pack_elems() > wps) wps = (b0conv_pack_elems() + 127) & ~(size_t)127;
            }
          }
        }
        if (stl > st) st = stl;
        if (pass != PASS_WGRAD) {
          if (wps * cnt > wpl) wpl = wps * cnt;
          if (wpl > L.wp_elems[pass]) L.wp_elems[pass] = wpl;
          if (wps > L.wp_stride[pass]) L.wp_stride[pass] = wps;
        }
      }
    }
  }
  for (BLayer& L : n->layers)
    for (int pass = 0; pass < (tr ? 2 : 1); ++pass) {
      // a class stride chosen at one batch size must leave room for every class at any other: stride * 8 classes at most
      if (L.wp_stride[pass] * 8 > L.wp_elems[pass] && L.wp_stride[pass] > 0) {
        GatherGeom g[8];
        int cmax = 1;
        for (int nb = 1; nb <= c.max_batch; ++nb) { const int cc = layer_geoms(n, L, (ConvPass)pass, nb, L.kin, L.kout, g); if (cc > cmax) cmax = cc; }
        if (L.wp_stride[pass] * cmax > L.wp_elems[pass]) L.wp_elems[pass] = L.wp_stride[pass] * cmax;
      }
      L.wp[pass] = (bf16_t*)A.take(L.wp_elems[pass] * sizeof(bf16_t) + 256);
    }
  n->pack.cap = 1024;
  n->pack.d_jobs = (BPackJob*)A.take((size_t)n->pack.cap * sizeof(BPackJob));
  n->pack.d_first = (int*)A.take((size_t)(n->pack.cap + 1) * sizeof(int));
  n->stats_doubles = st;
  n->stats = (double*)A.take(st * sizeof(double) + 256);
  n->stats2 = (double*)A.take(st * sizeof(double) + 256);   // the shortcut convs' partials: they run beside conv1 / conv2 on the second stream
  n->bn_scratch = A.take(bn + 256);
  n->wg_bytes = wg;
  n->wg_scratch = tr ? A.take(wg + 256) : nullptr;
  n->bs_scratch = tr ? (double*)A.take((size_t)16384 * 24 * sizeof(double)) : nullptr;

  n->sizes.n_params = poff;
  n->sizes.n_layers = (int64_t)n->layers.size();
  n->sizes.n_tensors = 2 * (int64_t)n->layers.size();
  n->sizes.workspace_bytes = (int64_t)((A.off + 255) & ~(size_t)255);
  return 0;
}

// ---- forward -------------------------------------------------------------------------------------------------------------
// ---- per-launch event records (bench.py --breakdown / --layers, roofline of the dominant kernel) ------------------------
extern "C" const char* ursn_last_kernel_name();
hipEvent_t bprof_event(ursn_bnet* n) {
  if (n->pev_used == n->pev.size()) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    n->pev.push_back(e);
  }
  return n->pev[n->pev_used++];
}
struct BProf {   // RAII: the conv paths return from several places
  ursn_bnet* n; hipStream_t s; int idx = -1; const char* fixed; bool range = false;
  BProf(ursn_bnet* n_, hipStream_t s_, int layer, int pass, double flops, double bytes, const char* name = nullptr)
      : n(n_), s(s_), fixed(name) {
    if (ursn_roctx_on()) { ursn_roctx_push(n->layers[layer].name.c_str(), pass); range = true; }
    if (!n->profile || n->prof.size() > 200000) return;
    BProfRec r{layer, pass, "", flops, bytes, bprof_event(n), bprof_event(n), ursn_kernel_launch_count(), 1};
    if (!r.e0 || !r.e1) return;
    hipEventRecord(r.e0, s);
    n->prof.push_back(r);
    idx = (int)n->prof.size() - 1;
  }
  ~BProf() {
    if (range) ursn_roctx_pop();
    if (idx < 0) return;
    BProfRec& r = n->prof[idx];
    r.kernel = fixed ? fixed : ursn_last_kernel_name();
    const long nl = ursn_kernel_launch_count() - r.l0;
    r.launches = nl > 0 ? (int)nl : 1;
    hipEventRecord(r.e1, s);
  }
};
// algorithmic work of a conv-like layer (SURVEY.md 8d conventions; bf16 tensors, fp32 weights)
double blayer_flops(const ursn_bnet* n, const BLayer& L, int N) {
  double taps = 1;
  for (int j = 0; j < n->cfg.ndim; ++j) taps *= L.k;
  return 2.0 * (double)N * (L.kind ? n->lvox[L.lin] : n->lvox[L.lout]) * taps * L.cin * L.cout;
}
double blayer_bytes(const ursn_bnet* n, const BLayer& L, int N) {
  double taps = 1;
  for (int j = 0; j < n->cfg.ndim; ++j) taps *= L.k;
  return 2.0 * ((double)N * n->lvox[L.lin] * L.cin + (double)N * n->lvox[L.lout] * L.cout) + 4.0 * taps * L.cin * L.cout;
}

const float* beta_of(ursn_bnet* n, const BLayer& L);
int conv_stats(ursn_bnet* n, int li, const BAct& in, int N, hipStream_t s, double* stats = nullptr) {
  if (!stats) stats = n->stats;
  BLayer& L = n->layers[li];
  GatherGeom g[8];
  const int cnt = layer_geoms(n, L, PASS_FWD, N, in.cs, L.kout, g);
  int Kw, Nw;
  real_extents(L, PASS_FWD, Kw, Nw);
  BProf ps(n, s, li, 0, blayer_flops(n, L, N), blayer_bytes(n, L, N));
  int total = 0, off = 0;
  if (bdeconv_ok(g, cnt)) {   // transposed conv 16 -> 8: the eight parity classes in one launch
    total = bdeconv_grid_blocks(g, cnt);
    URSN_TRY(launch_bdeconv(g, cnt, in.p, n->params + L.w_off, Kw, Nw, L.wp[0], L.z, stats, 0, s));
    return bconv_stats_finalize(g[0], stats, total, (int64_t)N * n->lvox[L.lout], n->cfg.bn_eps, L.mean, L.rstd, s);
  }
  if (bsconv_ok(g, cnt)) {   // transposed convs of the deeper levels: the eight parity classes in one launch
    URSN_TRY(launch_bsconv(g, cnt, in.p, n->params + L.w_off, Kw, Nw, L.wp[0], L.z, stats, 0, s));
    return bsconv_stats_finalize(g, cnt, stats, (int64_t)N * n->lvox[L.lout], n->cfg.bn_eps, L.mean, L.rstd, s);
  }
  if (cnt == 1 && !in.in_f32 && in.aff_layer < 0 && bs2k8_ok(g[0])) {   // stride-2 gather 8 -> 16 (without a shortcut beside it)
    total = bs2k8_grid_blocks(g[0]);
    URSN_TRY(launch_bs2k8(g[0], in.p, n->params + L.w_off, Kw, Nw, L.wp[0], L.z, stats, 0, nullptr, nullptr, 0, s));
    return launch_bn_stats_final(stats, total, g[0].Nn, 16, (int64_t)N * n->lvox[L.lout], n->cfg.bn_eps, L.mean, L.rstd, s);
  }
  if (in.in_f32 && cnt == 1 && b0conv_ok(g[0])) {   // conv0 on the raw fp32 input: the taps are the contraction (bf16_conv0.hip)
    total = b0conv_grid_blocks(g[0]);
    URSN_TRY(launch_b0conv(g[0], in.in_f32, n->params + L.w_off, Nw, L.wp[0], L.z, stats, s));
    return launch_bn_stats_final(stats, total, g[0].Nn, 16, (int64_t)N * n->lvox[L.lout], n->cfg.bn_eps, L.mean, L.rstd, s);
  }
  if (in.in_f32) {   // ... or as an 8-channel layer with seven absent channels
    URSN_REQUIRE(cnt == 1 && b3conv_ok(g[0]) && g[0].K == 8, "bf16 forward: %s cannot read a scalar fp32 input", L.name.c_str());
    total = bconv_grid_blocks(g[0]);
    g[0].accumulate = 0;
    URSN_TRY(launch_b3conv(g[0], nullptr, n->params + L.w_off, Kw, Nw, L.wp[0], L.z, stats, 0, total, s, nullptr, 0, nullptr, nullptr,
                           nullptr, nullptr, 0, in.in_f32));
    return bconv_stats_finalize(g[0], stats, total, (int64_t)N * n->lvox[L.lout], n->cfg.bn_eps, L.mean, L.rstd, s);
  }
  if (in.aff_layer >= 0) {   // virtual input: BatchNorm of the producer applied while staging
    const BLayer& P = n->layers[in.aff_layer];
    const int cnt2 = layer_geoms(n, L, PASS_FWD, N, P.kout, L.kout, g);
    URSN_REQUIRE(cnt2 == 1 && b3conv_aff_ok(g[0]), "bf16 forward: %s cannot normalise its input on load", L.name.c_str());
    B3Affine af = {P.mean, P.rstd, beta_of(n, P), in.aff_relu};
    total = bconv_grid_blocks(g[0]);
    g[0].accumulate = 0;
    URSN_TRY(launch_b3conv(g[0], P.z, n->params + L.w_off, Kw, Nw, L.wp[0], L.z, stats, 0, total, s, nullptr, 0, nullptr, nullptr, &af));
    return bconv_stats_finalize(g[0], stats, total, (int64_t)N * n->lvox[L.lout], n->cfg.bn_eps, L.mean, L.rstd, s);
  }
  for (int i = 0; i < cnt; ++i) total += bconv_grid_blocks(g[i]);
  URSN_REQUIRE(total > 0, "bf16 forward: no kernel for %s", L.name.c_str());
  for (int i = 0; i < cnt; ++i) {
    g[i].accumulate = 0;
    URSN_TRY(launch_bconv(g[i], in.p, n->params + L.w_off, Kw, Nw, L.wp[0] + (size_t)i * L.wp_stride[0], L.z, stats, off, total, s));
    off += bconv_grid_blocks(g[i]);
  }
  return bconv_stats_finalize(g[0], stats, total, (int64_t)N * n->lvox[L.lout], n->cfg.bn_eps, L.mean, L.rstd, s);
}

const float* beta_of(ursn_bnet* n, const BLayer& L) { return L.cout == L.kout ? n->params + L.b_off : n->beta_pad; }

int bn_out(ursn_bnet* n, int li, const BAct& out, int relu, int N, int li2, const BAct* res, hipStream_t s,
           unsigned char* mask_out = nullptr) {
  BLayer& L = n->layers[li];
  BBnActArgs a;
  memset(&a, 0, sizeof(a));
  a.z = L.z; a.zcs = L.kout; a.mean = L.mean; a.rstd = L.rstd; a.beta = beta_of(n, L);
  if (li2 >= 0) {
    BLayer& L2 = n->layers[li2];
    a.z2 = L2.z; a.z2cs = L2.kout; a.mean2 = L2.mean; a.rstd2 = L2.rstd; a.beta2 = beta_of(n, L2);
  }
  if (res) { a.res = res->p; a.rescs = res->cs; }
  a.y = out.p; a.ycs = out.cs; a.V = (int64_t)N * n->lvox[L.lout]; a.C = L.kout; a.relu = relu;
  a.mask_out = relu ? mask_out : nullptr;
  BProf ps(n, s, li, 4, 0.0, 2.0 * a.V * a.C * (2.0 + (li2 >= 0) + (res != nullptr)), "bbn_act");
  return launch_bbn_act(a, s);
}

hipEvent_t next_event(ursn_bnet* n) {
  if (n->ev_used == n->evs.size()) {
    hipEvent_t e;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
    n->evs.push_back(e);
  }
  return n->evs[n->ev_used++];
}

int unit_fwd(ursn_bnet* n, BUnit& u, int N, hipStream_t s) {
  // The 1x1 shortcut conv reads the unit's input like conv1 and is needed at the join only: it runs on the second stream beside
  // conv1 -> BatchNorm -> conv2 (these passes sit at 4-5 TB/s each, latency-bound; together they use more of the HBM).
  // URSN_BF16_FWD_OVERLAP=0: in line.
  static const bool overlap = !(getenv("URSN_BF16_FWD_OVERLAP") && getenv("URSN_BF16_FWD_OVERLAP")[0] == '0');
  hipEvent_t joined = nullptr;
  bool fused_sc = false;
  if (u.sc >= 0) {   // stride-2 conv 8 -> 16 with its 1x1 stride-2 shortcut: ONE pass over the fine tensor (bf16_s2k8.hip)
    BLayer& L = n->layers[u.c1];
    BLayer& S = n->layers[u.sc];
    GatherGeom g[8];
    if (u.in.aff_layer < 0 && !u.in.in_f32 && L.stride == 2 && S.stride == 2 && S.k == 1 && S.kin == 8 && S.cin == 8 && S.kout == 16 && S.cout == 16 &&
        layer_geoms(n, L, PASS_FWD, N, u.in.cs, L.kout, g) == 1 && bs2k8_ok(g[0])) {
      int Kw, Nw;
      real_extents(L, PASS_FWD, Kw, Nw);
      BProf ps(n, s, u.c1, 0, blayer_flops(n, L, N) + blayer_flops(n, S, N), blayer_bytes(n, L, N) + 2.0 * N * n->lvox[S.lout] * S.cout);
      const int blocks = bs2k8_grid_blocks(g[0]);
      URSN_TRY(launch_bs2k8(g[0], u.in.p, n->params + L.w_off, Kw, Nw, L.wp[0], L.z, n->stats, 0, n->params + S.w_off, S.z, S.kout, s));
      const int64_t V = (int64_t)N * n->lvox[L.lout];
      URSN_TRY(launch_bn_stats_final(n->stats, blocks, 16, 16, V, n->cfg.bn_eps, L.mean, L.rstd, s));
      URSN_TRY(launch_bn_stats_final(n->stats + (size_t)blocks * 32, blocks, 16, 16, V, n->cfg.bn_eps, S.mean, S.rstd, s));
      fused_sc = true;
    }
  }
  if (fused_sc) {
  } else if (u.sc >= 0 && overlap && n->s2 && n->s2_on) {
    hipEvent_t e0 = next_event(n);
    joined = next_event(n);
    URSN_REQUIRE(e0 && joined, "bf16 forward: no event for the shortcut stream");
    URSN_HIP(hipEventRecord(e0, s));
    URSN_HIP(hipStreamWaitEvent(n->s2, e0, 0));
    URSN_TRY(conv_stats(n, u.sc, u.in, N, n->s2, n->stats2));
    URSN_HIP(hipEventRecord(joined, n->s2));
  } else if (u.sc >= 0) {
    URSN_TRY(conv_stats(n, u.sc, u.in, N, s));
  }
  if (!fused_sc) URSN_TRY(conv_stats(n, u.c1, u.in, N, s));
  if (u.a1.aff_layer < 0) URSN_TRY(bn_out(n, u.c1, u.a1, 0, N, -1, nullptr, s));
  URSN_TRY(conv_stats(n, u.c2, u.a1, N, s));
  if (joined) URSN_HIP(hipStreamWaitEvent(s, joined, 0));
  if (u.sc >= 0) return bn_out(n, u.c2, u.out, 1, N, u.sc, nullptr, s, u.jmask);
  return bn_out(n, u.c2, u.out, 1, N, -1, &u.in, s, u.jmask);
}

int forward(ursn_bnet* n, const float* data, int N, hipStream_t s) {
  const int ns = n->cfg.num_strides;
  // Weight packing: every (layer, pass) has its own packed buffer; from the second step at a batch size on, ONE launch here
  // fills them all and the launchers find their job done (bf16_pack.h).  URSN_BF16_PREPACK=0: every launcher packs for itself.
  n->ev_used = 0;
  static const bool prepack = !(getenv("URSN_BF16_PREPACK") && getenv("URSN_BF16_PREPACK")[0] == '0');
  if (prepack && n->pack.d_jobs) {
    if (n->pack_N != N) { n->pack.clear(); n->pack_N = N; }
    bpack_set_ctx(&n->pack);
    URSN_TRY(bpack_replay(n->pack, s));
  }
  {  // conv2's beta, padded to the 8-channel piece (a packing job of its own kind: rides in the same launch)
    const BLayer& L2 = n->layers[n->conv2];
    BPackJob k = bpack_job(BPK_PAD8);
    k.w = n->params + L2.b_off; k.wp = (bf16_t*)n->beta_pad; k.p[0] = L2.cout; k.blocks = 1;
    URSN_TRY(bpack_submit(k, s));
  }
  n->a_data.in_f32 = n->scalar_in ? data : nullptr;
  if (!n->scalar_in) URSN_TRY(launch_bf16_input(data, n->a_data.p, (int64_t)N * n->lvox[0], s));
  URSN_TRY(conv_stats(n, n->conv0, n->a_data, N, s));
  URSN_TRY(bn_out(n, n->conv0, n->a_conv0, 1, N, -1, nullptr, s));
  size_t ui = 0;
  for (int step = 0; step < ns; ++step) {
    URSN_TRY(unit_fwd(n, n->units[ui++], N, s));
    URSN_TRY(unit_fwd(n, n->units[ui++], N, s));
  }
  for (int i = 0; i < ns; ++i) {
    URSN_TRY(conv_stats(n, n->deconv[i], n->deconv_in[i], N, s));
    if (n->skip0_own && i == ns - 1) {   // both halves of the level-0 concat voxel in one pass
      const BLayer& L = n->layers[n->deconv[i]];
      const BLayer& L0 = n->layers[n->conv0];
      BBnActArgs a;
      memset(&a, 0, sizeof(a));
      a.z = L.z; a.zcs = L.kout; a.mean = L.mean; a.rstd = L.rstd; a.beta = beta_of(n, L);
      a.z2 = L0.z; a.z2cs = L0.kout; a.mean2 = L0.mean; a.rstd2 = L0.rstd; a.beta2 = beta_of(n, L0);
      a.y = n->cat[i].p; a.ycs = n->cat[i].cs; a.V = (int64_t)N * n->lvox[0]; a.C = 8; a.relu = 1; a.cat = 1;
      BProf ps(n, s, n->deconv[i], 4, 0.0, 2.0 * a.V * 8 * 4.0, "bbn_act(concat)");
      URSN_TRY(launch_bbn_act(a, s));
    } else {
      URSN_TRY(bn_out(n, n->deconv[i], n->deconv_out[i], 1, N, -1, nullptr, s));
    }
    URSN_TRY(unit_fwd(n, n->units[ui++], N, s));
    URSN_TRY(unit_fwd(n, n->units[ui++], N, s));
  }
  URSN_TRY(conv_stats(n, n->conv1, n->a_pre1, N, s));
  if (n->a_conv1.aff_layer < 0) URSN_TRY(bn_out(n, n->conv1, n->a_conv1, 1, N, -1, nullptr, s));
  return conv_stats(n, n->conv2, n->a_conv1, N, s);
}

int head(ursn_bnet* n, const float* data, const float* label, const float* weight, int N, float* softmax_out, bool want_grad,
         hipStream_t s, float* ana_out = nullptr) {
  BLayer& L = n->layers[n->conv2];
  BHeadArgs a;
  memset(&a, 0, sizeof(a));
  a.z = L.z; a.z_cs = L.kout; a.mean = L.mean; a.rstd = L.rstd; a.beta = n->beta_pad;
  a.data = data; a.data_cs = 1; a.label = label; a.weight = weight; a.n = N; a.pix = n->lvox[0]; a.ncls = n->cfg.num_class;
  a.softmax_out = softmax_out; a.dlogits = want_grad ? n->dlog : nullptr; a.dl_cs = 8; a.ana_out = ana_out;
  a.metrics = n->metrics; a.scratch = n->head_scratch;
  // the logits layer's BatchNorm-backward sums ride in the head (dlogits and z are in its registers): one pass of two tensors less
  static const bool fuse = !(getenv("URSN_BF16_HEAD_BN_BWD") && getenv("URSN_BF16_HEAD_BN_BWD")[0] == '0');
  n->bs_layer = -1;
  if (want_grad && fuse && n->bs_scratch && L.kout == 8 && bhead_blocks(N, n->lvox[0]) <= 16384) {
    a.bs_partial = n->bs_scratch;
    n->bs_layer = n->conv2; n->bs_blocks = bhead_blocks(N, n->lvox[0]);
  }
  BProf ps(n, s, n->conv2, 6, 0.0, (double)N * n->lvox[0] * (16.0 + 12.0 + (want_grad ? 16.0 : 0.0)), "bhead");
  return launch_bhead(a, s);
}

// ---- backward ------------------------------------------------------------------------------------------------------------
bool take_flag(ursn_bnet* n, const BAct& a) {
  const bool acc = n->ginit[a.flag] != 0;
  n->ginit[a.flag] = 1;
  return acc;
}

// The BatchNorm backward that consumes the gradient a data-gradient launch completes (as net.hip's BsTarget)
struct BBsTarget {
  int li = -1, li2 = -1, mode = 0;   // mode: 0 no activation, 1 mask = y > 0, 2 mask = bn(z) > 0, 3 mask bytes
  const bf16_t* y = nullptr; int ycs = 0;
  const unsigned char* maskb = nullptr;
};

// fused_sc >= 0: the data gradient also carries the term of that (1x1, stride-1) shortcut layer
// wgrad_sc >= 0: the weight gradient launch also takes that (1x1 stride-2) shortcut layer's (bf16_s2k8w.hip); skip_wgrad: it was taken there
int conv_bwd(ursn_bnet* n, int li, const BAct& in, bool need_dgrad, int N, hipStream_t s, int fused_sc = -1,
             const BBsTarget* bs = nullptr, const B3Residual* res = nullptr, int wgrad_sc = -1, bool skip_wgrad = false) {
  BLayer& L = n->layers[li];
  GatherGeom g[8];
  int Kw, Nw;
  if (!skip_wgrad) {  // weight gradient on the second stream, ordered after the dz it reads
    const BAct inw = in;   // (by value: a deferred launch outlives the caller's reference)
    auto wgrad = [n, li, inw, N, s, wgrad_sc]() -> int {
      BLayer& L = n->layers[li];
      const BAct& in = inw;
      GatherGeom g[8];
      int Kw, Nw;
      URSN_REQUIRE(layer_geoms(n, L, PASS_WGRAD, N, in.cs, L.kout, g) == 1, "bf16 backward: bad weight-gradient geometry");
      real_extents(L, PASS_WGRAD, Kw, Nw);
      hipStream_t ws = s;
      if (n->s2 && n->s2_on) {
        hipEvent_t e = next_event(n);
        URSN_REQUIRE(e, "bf16 backward: no event for the weight-gradient stream");
        URSN_HIP(hipEventRecord(e, s));
        URSN_HIP(hipStreamWaitEvent(n->s2, e, 0));
        ws = n->s2;
      }
      BProf pw(n, ws, li, 2, blayer_flops(n, L, N), blayer_bytes(n, L, N));
      if (in.in_f32 && !L.kind && b0wgrad_ok(g[0])) {
        URSN_TRY(launch_b0wgrad(g[0], in.in_f32, L.dz, n->grads + L.w_off, Nw, n->wg_scratch, n->wg_bytes, ws));
      } else if (in.in_f32) {
        URSN_REQUIRE(!L.kind && b3wgrad_scalar_ok(g[0]), "bf16 backward: %s cannot read a scalar fp32 input", L.name.c_str());
        URSN_TRY(launch_b3wgrad(g[0], nullptr, L.dz, n->grads + L.w_off, Kw, Nw, n->wg_scratch, n->wg_bytes, ws, nullptr, in.in_f32));
      } else if (in.aff_layer >= 0) {
        const BLayer& P = n->layers[in.aff_layer];
        URSN_REQUIRE(!L.kind && layer_geoms(n, L, PASS_WGRAD, N, P.kout, L.kout, g) == 1 && b3wgrad_ok(g[0]), "bf16 backward: %s cannot normalise its input on load", L.name.c_str());
        B3Affine af = {P.mean, P.rstd, beta_of(n, P), in.aff_relu};
        URSN_TRY(launch_b3wgrad(g[0], P.z, L.dz, n->grads + L.w_off, Kw, Nw, n->wg_scratch, n->wg_bytes, ws, &af));
      } else if (wgrad_sc >= 0) {
        const BLayer& SL = n->layers[wgrad_sc];
        URSN_REQUIRE(!L.kind && bs2k8w_sc_ok(g[0]), "bf16 backward: %s cannot take its shortcut's weight gradient along", L.name.c_str());
        URSN_TRY(launch_bs2k8w(g[0], in.p, L.dz, n->grads + L.w_off, Kw, Nw, n->wg_scratch, n->wg_bytes, SL.dz, SL.kout, n->grads + SL.w_off, ws));
      } else {
        const bf16_t* S = L.kind ? L.dz : in.p;
        const bf16_t* Cq = L.kind ? in.p : L.dz;
        URSN_TRY(launch_bwgrad(g[0], S, Cq, n->grads + L.w_off, Kw, Nw, n->wg_scratch, n->wg_bytes, ws));
      }
      return 0;
    };
    // The level-0 weight gradients are HBM-heavy (2-3 GB each) and so is everything the main stream does at level 0: launched
    // at once they run BESIDE it and both slow down (32.4 ms of main-stream kernels + 9.9 ms of weight gradients take 39.6 ms).
    // Opt-in experiment (URSN_BF16_DEFER_WGRAD=L): queue them while the decoder is at level 0 and release them when the main stream
    // reaches decoder level L (small latency-bound kernels that leave the HBM idle).  Measured at cfg5: L = 1 / 2 / 3 / 4: 98.0 /
    // 98.5 / 98.7 / 98.5 images/s against 101.1 launched at once -- the deep levels are too short (5 ms) to absorb 4.8 ms of level-0
    // weight gradients, which then land beside the encoder's level-1 / level-0 passes instead.  Off.
    if (n->defer_open && n->s2 && n->s2_on && L.lout == 0) n->deferred.push_back(wgrad);
    else URSN_TRY(wgrad());
  }
  if (!need_dgrad) return 0;
  BProf pd(n, s, li, 1, blayer_flops(n, L, N), blayer_bytes(n, L, N));
  const bool acc = take_flag(n, in);
  const int cnt = layer_geoms(n, L, PASS_DGRAD, N, in.cs, L.kout, g);
  real_extents(L, PASS_DGRAD, Kw, Nw);
  bool empty = false;
  for (int i = 0; i < cnt; ++i) empty = empty || g[i].ntaps == 0;
  URSN_REQUIRE(!empty || acc, "bf16 backward: %s would leave voxels of its input gradient unwritten", L.name.c_str());
  n->bs_layer = -1;
  if (res) {   // identity unit: the residual branch's share rides in this launch (unit_bwd checked the shape)
    URSN_REQUIRE(cnt == 1 && !acc && fused_sc < 0 && b3conv_ok(g[0]) && g[0].K == g[0].Nn, "bf16 backward: %s cannot carry the residual term", L.name.c_str());
    g[0].accumulate = 0;
    return launch_b3conv(g[0], L.dz, n->params + L.w_off, Kw, Nw, L.wp[1], in.g, nullptr, 0, 0, s, nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0,
                         nullptr, res);
  }
  if (bs && bs->li >= 0 && bs->mode != 1 && fused_sc < 0 && cnt == 1 && b3conv_bs_ok(g[0]) && in.C == 8 && in.cs == 8) {
    const BLayer& T = n->layers[bs->li];
    const int blocks = bconv_grid_blocks(g[0]);
    if (T.kout == 8 && blocks <= 16384 && (bs->li2 < 0 || n->layers[bs->li2].kout == 8)) {
      B3BnRed r;
      memset(&r, 0, sizeof(r));
      r.z = T.z; r.z_cs = T.kout; r.mean = T.mean; r.rstd = T.rstd; r.beta = beta_of(n, T);
      r.y = bs->y; r.y_cs = bs->ycs; r.mode = bs->mode; r.maskb = bs->maskb;
      if (bs->li2 >= 0) {
        const BLayer& T2 = n->layers[bs->li2];
        r.z2 = T2.z; r.z2_cs = T2.kout; r.mean2 = T2.mean; r.rstd2 = T2.rstd;
      }
      r.partial = n->bs_scratch;
      g[0].accumulate = acc ? 1 : 0;
      URSN_TRY(launch_b3conv(g[0], L.dz, n->params + L.w_off, Kw, Nw, L.wp[1], in.g, nullptr, 0, 0, s, nullptr, 0, nullptr, &r));
      n->bs_layer = bs->li; n->bs_blocks = blocks;
      return 0;
    }
  }
  if (fused_sc >= 0 && cnt == 1 && !b3conv_ok(g[0]) && bcbconv_pw_ok(g[0])) {   // levels 1-2: the term rides in the channel-block kernel
    const BLayer& S = n->layers[fused_sc];
    URSN_REQUIRE(S.k == 1 && S.stride == 1 && S.kout == g[0].K && S.cout == S.kout && S.kin == g[0].Nn && S.cin == S.kin,
                 "bf16 backward: shortcut of %s does not match its data gradient", L.name.c_str());
    g[0].accumulate = acc ? 1 : 0;
    return launch_bcbconv(g[0], L.dz, n->params + L.w_off, Kw, Nw, L.wp[1], in.g, nullptr, s, S.dz, S.kout, n->params + S.w_off);
  }
  if (fused_sc >= 0) {
    const BLayer& S = n->layers[fused_sc];
    URSN_REQUIRE(cnt == 1 && b3conv_pw_ok(g[0]) && S.kin == 16 && S.cin == 16 && S.cout == 8 && S.kout == 8, "bf16 backward: no fused shortcut term for %s", L.name.c_str());
    g[0].accumulate = acc ? 1 : 0;
    if (n->dec0_g && in.g == n->cat[n->cfg.num_strides - 1].g && !acc) {   // level-0 concat gradient: two 8-channel tensors
      g[0].out_cs = 8;
      n->split0_done = true;
      if (n->skip0_merge) n->ginit[n->a_conv0.flag] = 1;   // the skip's share is in place: the encoder's data gradients accumulate
      return launch_b3conv(g[0], L.dz, n->params + L.w_off, Kw, Nw, L.wp[1], n->dec0_g, nullptr, 0, 0, s, S.dz, S.kout, n->params + S.w_off,
                           nullptr, nullptr, n->skip0_g, 8);
    }
    return launch_b3conv(g[0], L.dz, n->params + L.w_off, Kw, Nw, L.wp[1], in.g, nullptr, 0, 0, s, S.dz, S.kout, n->params + S.w_off);
  }
  if (bdeconv_ok(g, cnt))   // stride-2 conv 8 -> 16: the eight parity classes of its data gradient in one launch
    return launch_bdeconv(g, cnt, L.dz, n->params + L.w_off, Kw, Nw, L.wp[1], in.g, nullptr, acc ? 1 : 0, s);
  if (cnt == 1 && fused_sc < 0 && bs2k8_ok(g[0]))   // transposed conv 16 -> 8: its data gradient is a stride-2 gather 8 -> 16
    return launch_bs2k8(g[0], L.dz, n->params + L.w_off, Kw, Nw, L.wp[1], in.g, nullptr, acc ? 1 : 0, nullptr, nullptr, 0, s);
  if (bsconv_ok(g, cnt))   // stride-2 convs of the deeper levels: likewise
    return launch_bsconv(g, cnt, L.dz, n->params + L.w_off, Kw, Nw, L.wp[1], in.g, nullptr, acc ? 1 : 0, s);
  for (int i = 0; i < cnt; ++i) {
    if (g[i].ntaps == 0) continue;
    g[i].accumulate = acc ? 1 : 0;
    URSN_TRY(launch_bconv(g[i], L.dz, n->params + L.w_off, Kw, Nw, L.wp[1] + (size_t)i * L.wp_stride[1], in.g, nullptr, 0, 0, s));
  }
  return 0;
}

int bn_back(ursn_bnet* n, int li, const bf16_t* dy, int dycs, const bf16_t* y, int ycs, int relu, int li2, bf16_t* dres,
            int drescs, int dres_acc, int N, hipStream_t s, const bf16_t* dy2 = nullptr, int dy2cs = 0,
            const unsigned char* mask = nullptr) {
  BLayer& L = n->layers[li];
  BBnBwdArgs a;
  memset(&a, 0, sizeof(a));
  a.dy = dy; a.dycs = dycs; a.y = y; a.ycs = ycs; a.dy2 = dy2; a.dy2cs = dy2cs; a.mask = mask;
  a.z = L.z; a.zcs = L.kout; a.mean = L.mean; a.rstd = L.rstd; a.dz = L.dz; a.dzcs = L.kout;
  a.dbeta = n->grads + L.b_off; a.beta = beta_of(n, L);
  if (li2 >= 0) {
    BLayer& L2 = n->layers[li2];
    a.z2 = L2.z; a.z2cs = L2.kout; a.mean2 = L2.mean; a.rstd2 = L2.rstd; a.dz2 = L2.dz; a.dz2cs = L2.kout;
    a.dbeta2 = n->grads + L2.b_off;
  }
  a.dres = dres; a.drescs = drescs; a.dres_accumulate = dres_acc;
  a.V = (int64_t)N * n->lvox[L.lout]; a.C = L.kout; a.Cw = L.cout; a.relu = relu; a.scratch = n->bn_scratch;
  if (n->bs_layer == li && a.C == 8) { a.pre_partial = n->bs_scratch; a.pre_nblocks = n->bs_blocks; }
  n->bs_layer = -1;
  // reduce: dy, z (+ y | z2); apply: the same again + dz (+ dz2 | dres); the mask bytes are 1/16 of a tensor
  BProf ps(n, s, li, 5, 0.0, 2.0 * a.V * a.C * (2.0 * (2 + (relu && y && !mask) + (li2 >= 0)) + 1 + (li2 >= 0) + (dres != nullptr)), "bbn_bwd");
  return launch_bbn_bwd(a, s);
}

int unit_bwd(ursn_bnet* n, BUnit& u, int N, hipStream_t s, const BBsTarget* in_target = nullptr) {
  bool res_in_dgrad = false;
  if (u.sc >= 0) {
    URSN_TRY(bn_back(n, u.c2, u.out.g, u.out.cs, u.out.p, u.out.cs, 1, u.sc, nullptr, 0, 0, N, s, nullptr, 0, u.jmask));
  } else {
    // identity shortcut: d(in) = conv1's data gradient + g(out) * relu mask.  Where conv1's data gradient runs on the
    // input-stationary kernel and nothing has been written to d(in) yet, that kernel adds the second term itself (it reads g(out)
    // and the mask bytes instead of old values) and the join's BatchNorm backward writes one tensor less
    // (URSN_BF16_RESIDUAL_IN_DGRAD=0: written here, accumulated there)
    static const bool fuse_res = !(getenv("URSN_BF16_RESIDUAL_IN_DGRAD") && getenv("URSN_BF16_RESIDUAL_IN_DGRAD")[0] == '0');
    GatherGeom gg[8];
    const BLayer& C1 = n->layers[u.c1];
    const int gc = layer_geoms(n, C1, PASS_DGRAD, N, u.in.cs, C1.kout, gg);
    res_in_dgrad = fuse_res && u.jmask && !n->ginit[u.in.flag] && gc == 1 && b3conv_ok(gg[0]) && gg[0].K == gg[0].Nn &&
                   (in_target == nullptr || !b3conv_bs_ok(gg[0])) && ursn_bf16_plane_ok(gg[0], u.out.cs);
    if (res_in_dgrad) {
      URSN_TRY(bn_back(n, u.c2, u.out.g, u.out.cs, u.out.p, u.out.cs, 1, -1, nullptr, 0, 0, N, s, nullptr, 0, u.jmask));
    } else {
      const bool acc = take_flag(n, u.in);
      URSN_TRY(bn_back(n, u.c2, u.out.g, u.out.cs, u.out.p, u.out.cs, 1, -1, u.in.g, u.in.cs, acc ? 1 : 0, N, s, nullptr, 0, u.jmask));
    }
  }
  BBsTarget t1;   // conv2's data gradient IS d(a1): resnet_conv1's BatchNorm (no activation) consumes it
  t1.li = u.c1;
  URSN_TRY(conv_bwd(n, u.c2, u.a1, true, N, s, -1, &t1));
  URSN_TRY(bn_back(n, u.c1, u.a1.g, u.a1.cs, nullptr, 0, 0, -1, nullptr, 0, 0, N, s));
  bool fuse = false;   // stride-1 shortcut next to an 8 -> 16 data gradient: its term rides in that kernel's idle k slot
  if (u.sc >= 0 && n->layers[u.sc].stride == 1) {
    GatherGeom g[8];
    const BLayer& S = n->layers[u.sc];
    const bool one = layer_geoms(n, n->layers[u.c1], PASS_DGRAD, N, u.in.cs, n->layers[u.c1].kout, g) == 1;
    fuse = one && b3conv_pw_ok(g[0]) && S.kin == 16 && S.cin == 16 && S.cout == 8 && S.kout == 8;
    // ... or, at levels 1-2, as KS more k steps of the channel-block kernel's centre tap
    fuse = fuse || (one && !b3conv_ok(g[0]) && bcbconv_pw_ok(g[0]) && S.kout == g[0].K && S.cout == S.kout && S.kin == g[0].Nn && S.cin == S.kin);
  }
  // with an identity shortcut conv1's data gradient is the last contribution to d(in): its consumer's reductions ride along
  if (res_in_dgrad) {
    const B3Residual rs = {u.out.g, u.out.cs, u.jmask};
    URSN_TRY(conv_bwd(n, u.c1, u.in, true, N, s, -1, nullptr, &rs));
    return 0;
  }
  // stride-2 unit between the two finest levels: the shortcut's weight gradient rides in conv1's launch (one pass over the fine tensor)
  bool wsc = false;
  if (u.sc >= 0 && u.in.aff_layer < 0 && !u.in.in_f32) {
    GatherGeom gw[8];
    const BLayer& C1 = n->layers[u.c1];
    const BLayer& S = n->layers[u.sc];
    wsc = C1.stride == 2 && S.stride == 2 && S.k == 1 && S.kin == 8 && S.cin == 8 && S.kout == 16 && S.cout == 16 &&
          layer_geoms(n, C1, PASS_WGRAD, N, u.in.cs, C1.kout, gw) == 1 && bs2k8w_sc_ok(gw[0]);
  }
  URSN_TRY(conv_bwd(n, u.c1, u.in, true, N, s, fuse ? u.sc : -1, u.sc < 0 ? in_target : nullptr, nullptr, wsc ? u.sc : -1));   // k3 (s1 | s2): writes every voxel of d(in)
  if (u.sc >= 0) URSN_TRY(conv_bwd(n, u.sc, u.in, !fuse, N, s, -1, nullptr, nullptr, -1, wsc));   // 1x1 (s1 | s2): weight gradient (+ accumulated data gradient)
  return 0;
}

int flush_deferred(ursn_bnet* n) {
  n->defer_open = false;
  for (auto& f : n->deferred) URSN_TRY(f());
  n->deferred.clear();
  return 0;
}

int backward(ursn_bnet* n, int N, hipStream_t s) {
  const int ns = n->cfg.num_strides;
  for (size_t i = 0; i < n->ginit.size(); ++i) n->ginit[i] = 0;
  n->ev_used = 0;
  n->split0_done = false;
  // URSN_BF16_DEFER_WGRAD=L: the level-0 weight gradients of the decoder are released when the main stream starts decoder level L
  // (default 0: launched at once -- measured faster, see conv_bwd)
  static const int defer_level = getenv("URSN_BF16_DEFER_WGRAD") ? atoi(getenv("URSN_BF16_DEFER_WGRAD")) : 0;
  n->deferred.clear();
  n->defer_open = defer_level > 0 && n->s2 && n->s2_on;
  URSN_TRY(bn_back(n, n->conv2, n->dlog, 8, nullptr, 0, 0, -1, nullptr, 0, 0, N, s));
  BBsTarget tc;
  tc.li = n->conv1; tc.mode = 2;
  URSN_TRY(conv_bwd(n, n->conv2, n->a_conv1, true, N, s, -1, &tc));
  URSN_TRY(bn_back(n, n->conv1, n->a_conv1.g, n->a_conv1.cs, nullptr, 0, 1, -1, nullptr, 0, 0, N, s));
  size_t ui = n->units.size();
  auto join_of = [&](const BUnit& u) {
    BBsTarget t;
    t.li = u.c2; t.li2 = u.sc; t.mode = u.jmask ? 3 : 1; t.y = u.out.p; t.ycs = u.out.cs; t.maskb = u.jmask;
    return t;
  };
  tc = join_of(n->units[ui - 1]);
  URSN_TRY(conv_bwd(n, n->conv1, n->a_pre1, true, N, s, -1, &tc));
  for (int i = ns - 1; i >= 0; --i) {
    if (ns - 1 - i == defer_level) URSN_TRY(flush_deferred(n));
    tc = join_of(n->units[ui - 2]);
    URSN_TRY(unit_bwd(n, n->units[--ui], N, s, &tc));
    URSN_TRY(unit_bwd(n, n->units[--ui], N, s));
    const BAct& dout = n->deconv_out[i];
    if (n->dec0_g && i == ns - 1 && n->split0_done) URSN_TRY(bn_back(n, n->deconv[i], n->dec0_g, 8, nullptr, 0, 1, -1, nullptr, 0, 0, N, s));
    else URSN_TRY(bn_back(n, n->deconv[i], dout.g, dout.cs, nullptr, 0, 1, -1, nullptr, 0, 0, N, s));
    URSN_TRY(conv_bwd(n, n->deconv[i], n->deconv_in[i], true, N, s));
  }
  URSN_TRY(flush_deferred(n));   // (a shallow network never reached the level)
  for (int step = ns - 1; step >= 0; --step) {
    tc = join_of(n->units[ui - 2]);
    URSN_TRY(unit_bwd(n, n->units[--ui], N, s, &tc));
    URSN_TRY(unit_bwd(n, n->units[--ui], N, s));
  }
  const BAct& a0 = n->a_conv0;
  if (n->skip0_own) {   // d(conv0 activation) = the encoder's share (own tensor) + the skip's share (second half of the concat gradient)
    const BAct& cg = n->cat[ns - 1];
    if (n->split0_done && n->skip0_merge) URSN_TRY(bn_back(n, n->conv0, a0.g, a0.cs, nullptr, 0, 1, -1, nullptr, 0, 0, N, s));
    else if (n->split0_done) URSN_TRY(bn_back(n, n->conv0, a0.g, a0.cs, nullptr, 0, 1, -1, nullptr, 0, 0, N, s, n->skip0_g, 8));
    else URSN_TRY(bn_back(n, n->conv0, a0.g, a0.cs, nullptr, 0, 1, -1, nullptr, 0, 0, N, s, cg.g + cg.C / 2, cg.cs));
  } else {
    URSN_TRY(bn_back(n, n->conv0, a0.g, a0.cs, nullptr, 0, 1, -1, nullptr, 0, 0, N, s));
  }
  URSN_TRY(conv_bwd(n, n->conv0, n->a_data, false, N, s));
  if (n->s2) {
    URSN_HIP(hipEventRecord(n->s2_done, n->s2));
    URSN_HIP(hipStreamWaitEvent(s, n->s2_done, 0));
  }
  return 0;
}

}  // namespace

// ---- entry points used by net.hip's C-ABI dispatch -----------------------------------------------------------------------
int bnet_query(const ursn_config* cfg, ursn_sizes* out) {
  ursn_bnet tmp;
  tmp.cfg = *cfg;
  if (tmp.cfg.bn_eps <= 0.f) tmp.cfg.bn_eps = 1e-3f;
  Arena A;
  URSN_TRY(plan(&tmp, A));
  *out = tmp.sizes;
  return 0;
}

int bnet_layer(const ursn_config* cfg, int64_t index, ursn_layer_info* out, int* n_layers) {
  ursn_bnet tmp;
  tmp.cfg = *cfg;
  Arena A;
  URSN_TRY(plan(&tmp, A));
  if (n_layers) *n_layers = (int)tmp.layers.size();
  if (!out) return 0;
  URSN_REQUIRE(index >= 0 && index < (int64_t)tmp.layers.size(), "query_layer: index %lld out of range", (long long)index);
  const BLayer& L = tmp.layers[index];
  memset(out, 0, sizeof(*out));
  snprintf(out->name, sizeof(out->name), "%s", L.name.c_str());
  out->transposed = L.kind; out->k = L.k; out->stride = L.stride; out->cin = L.cin; out->cout = L.cout;
  out->relu = (L.kind == 1 || L.name == "UResNet/conv0" || L.name == "UResNet/conv1") ? 1 : 0;
  out->w_offset = L.w_off; out->beta_offset = L.b_off;
  return 0;
}

int bnet_create(const ursn_config* cfg, float* params, float* grads, void* workspace, size_t workspace_bytes, ursn_bnet** out) {
  ursn_bnet* n = new ursn_bnet();
  n->cfg = *cfg;
  if (n->cfg.bn_eps <= 0.f) n->cfg.bn_eps = 1e-3f;
  Arena A;
  A.base = (char*)workspace;
  int rc = plan(n, A);
  if (rc == 0 && (size_t)n->sizes.workspace_bytes > workspace_bytes) {
    ursn_set_error("create: workspace too small: need %lld bytes, got %zu", (long long)n->sizes.workspace_bytes, workspace_bytes);
    rc = 2;
  }
  if (rc) { delete n; return rc; }
  n->params = params; n->grads = grads;
  int lo = 0, hi = 0;
  (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
  const char* e2 = getenv("URSN_WGRAD_STREAM");
  if (e2 && e2[0] == '0') { *out = n; return 0; }   // weight gradients on the caller's stream (per-kernel profiling)
  if (hipStreamCreateWithPriority(&n->s2, hipStreamNonBlocking, lo) != hipSuccess ||
      hipEventCreateWithFlags(&n->s2_done, hipEventDisableTiming) != hipSuccess) {
    ursn_set_error("create: could not create the weight-gradient stream");
    delete n;
    return 1;
  }
  // pad channels of the logits buffers / input expansion are written by the kernels themselves (zero weights -> zeros)
  *out = n;
  return 0;
}

void bnet_destroy(ursn_bnet* n) {
  if (!n) return;
  if (n->s2) { (void)hipStreamSynchronize(n->s2); (void)hipStreamDestroy(n->s2); }
  if (n->s2_done) (void)hipEventDestroy(n->s2_done);
  for (hipEvent_t e : n->evs) (void)hipEventDestroy(e);
  for (hipEvent_t e : n->pev) (void)hipEventDestroy(e);
  delete n;
}

int bnet_profile_enable(ursn_bnet* n, int on) {
  n->profile = on != 0;
  n->prof.clear();
  n->pev_used = 0;
  return 0;
}
int bnet_profile_read(ursn_bnet* n, ursn_prof_rec* out, int64_t max_recs, int64_t* n_out) {
  int64_t cnt = 0;
  for (size_t i = 0; i < n->prof.size() && (!out || cnt < max_recs); ++i) {   // out == NULL: only counts, all of them
    const BProfRec& r = n->prof[i];
    float ms = 0.f;
    if (hipEventSynchronize(r.e1) != hipSuccess || hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) continue;
    if (out) {
      ursn_prof_rec& o = out[cnt];
      memset(&o, 0, sizeof(o));
      snprintf(o.kernel, sizeof(o.kernel), "%s", r.kernel);
      snprintf(o.layer, sizeof(o.layer), "%s", n->layers[r.layer].name.c_str());
      o.pass = r.pass; o.ms = ms; o.flops = r.flops; o.bytes = r.bytes; o.launches = r.launches;
    }
    ++cnt;
  }
  *n_out = cnt;
  if (out) { n->prof.clear(); n->pev_used = 0; }
  return 0;
}
int bnet_set_wgrad_overlap(ursn_bnet* n, int on) {
  if (n->s2) (void)hipStreamSynchronize(n->s2);
  n->s2_on = on != 0;
  return 0;
}

const ursn_sizes* bnet_sizes(const ursn_bnet* n) { return &n->sizes; }
float* bnet_metrics(ursn_bnet* n) { return n->metrics; }

int bnet_param(const ursn_bnet* n, int64_t index, ursn_param_info* out) {
  URSN_REQUIRE(index >= 0 && index < n->sizes.n_tensors, "param: index %lld out of range", (long long)index);
  const BLayer& L = n->layers[index / 2];
  memset(out, 0, sizeof(*out));
  if (index % 2 == 0) {
    snprintf(out->name, sizeof(out->name), "%s/weights", L.name.c_str());
    out->offset = L.w_off; out->nelem = L.w_n; out->rank = n->cfg.ndim + 2;
    for (int j = 0; j < n->cfg.ndim; ++j) out->shape[j] = L.k;
    out->shape[n->cfg.ndim] = L.kind ? L.cout : L.cin;
    out->shape[n->cfg.ndim + 1] = L.kind ? L.cin : L.cout;
  } else {
    snprintf(out->name, sizeof(out->name), "%s/BatchNorm/beta", L.name.c_str());
    out->offset = L.b_off; out->nelem = L.cout; out->rank = 1; out->shape[0] = L.cout;
  }
  return 0;
}

int bnet_step(ursn_bnet* n, const float* data, const float* label, const float* weight, int N, int mode, float* softmax_out,
              float* labels_out, hipStream_t s) {
  // mode 0: accumulate gradients (forward + loss + backward); 1: evaluate (forward + loss); 2: inference
  struct PackScope { ~PackScope() { bpack_set_ctx(nullptr); } } pack_scope;   // forward() installs the network's packing context
  URSN_TRY(forward(n, data, N, s));
  const float* w = n->cfg.use_weight ? weight : nullptr;
  if (mode == 0) {
    URSN_TRY(head(n, data, label, w, N, nullptr, true, s));
    return backward(n, N, s);
  }
  if (mode == 1) return head(n, data, label, w, N, nullptr, false, s);
  return head(n, data, label, nullptr, N, softmax_out, false, s, labels_out);
}

int bnet_tensor(const ursn_bnet* n, const char* name, void** ptr, int64_t* voxels, int32_t* channels, int32_t* cstride) {
  // <scope>:z / :dz raw conv output and its gradient (bf16), <scope>:mean / :rstd the BatchNorm statistics (fp32, voxels = 0),
  // <scope> / <scope>:grad a materialised activation and the gradient tensor of the same layout (bf16), logits:grad the head's
  // output; at F = 8 the level-0 concat gradient lives in two 8-channel tensors: UResNet/deconv<last>:grad and
  // UResNet/conv0:grad2 (the skip's share; UResNet/conv0:grad is the encoder's share)
  std::string s(name);
  bool want_z = false, want_dz = false, want_g = false, want_g2 = false, want_mean = false, want_rstd = false;
  auto strip = [&](const char* suf, bool& f) {
    const size_t L = strlen(suf);
    if (s.size() > L && s.compare(s.size() - L, L, suf) == 0) { f = true; s = s.substr(0, s.size() - L); }
  };
  strip(":dz", want_dz);
  strip(":z", want_z);
  strip(":grad2", want_g2);
  strip(":grad", want_g);
  strip(":mean", want_mean);
  strip(":rstd", want_rstd);
  const int ns = n->cfg.num_strides;
  if (want_g && s == "logits") {
    URSN_REQUIRE(n->dlog, "tensor: net is not trainable");
    *ptr = n->dlog; *voxels = n->lvox[0]; *channels = n->cfg.num_class; *cstride = 8;
    return 0;
  }
  if (want_z || want_dz || want_mean || want_rstd) {
    for (const BLayer& L : n->layers)
      if (L.name == s) {
        *ptr = want_z ? (void*)L.z : want_dz ? (void*)L.dz : want_mean ? (void*)L.mean : (void*)L.rstd;
        URSN_REQUIRE(*ptr, "tensor: %s does not exist (net is not trainable)", name);
        *voxels = (want_mean || want_rstd) ? 0 : n->lvox[L.lout]; *channels = L.cout; *cstride = L.kout;
        return 0;
      }
    ursn_set_error("tensor: no layer named %s", s.c_str());
    return 2;
  }
  if (want_g2) {
    URSN_REQUIRE(s == "UResNet/conv0" && n->skip0_own && n->cfg.trainable && !(n->skip0_merge && n->split0_done),
                 "tensor: %s has no second gradient tensor", s.c_str());
    const BAct& cg = n->cat[ns - 1];
    if (n->split0_done) { *ptr = n->skip0_g; *cstride = 8; }
    else { *ptr = cg.g + cg.C / 2; *cstride = cg.cs; }
    *voxels = n->lvox[0]; *channels = 8;
    return 0;
  }
  auto it = n->named.find(s);
  URSN_REQUIRE(it != n->named.end(), "tensor: no activation named %s", s.c_str());
  const BAct& a = it->second;
  *voxels = n->lvox[a.lvl]; *channels = a.C; *cstride = a.cs;
  if (want_g && n->split0_done && n->dec0_g && s == n->layers[n->deconv[ns - 1]].name) {
    *ptr = n->dec0_g; *cstride = 8;
    return 0;
  }
  *ptr = want_g ? (void*)a.g : (void*)a.p;
  URSN_REQUIRE(*ptr, "tensor: %s is not materialised (its BatchNorm is applied while the consuming convolution stages it)", name);
  return 0;
}
