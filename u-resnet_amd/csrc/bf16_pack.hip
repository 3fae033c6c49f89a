// bf16 plan: the weight packing kernels of every conv kernel family behind one job description (bf16_pack.h) -- gfx950.
// fp32 master W[t][k][n] (strides w_tap_stride / w_sk / w_sn: the data-gradient passes read the same tensor transposed) -> the
// bf16 A-fragment order of the family; layouts are documented beside the kernels that read them.
#include "bf16_pack.h"

namespace {

__device__ __forceinline__ float pk_w(const BPackJob& k, int tw, int ci, int co) {
  return k.w[(int64_t)tw * k.w_tap_stride + (int64_t)ci * k.w_sk + (int64_t)co * k.w_sn];
}

// BPK_GENERIC (bf16_conv.hip: box kernel): p = {ntaps, cinc, nchunks, nj, cot, ncob}; tap[t] = stored tap index
// [co block][chunk][j][co tile][lane = 16 g + m][8] + the 8-element zero piece
__device__ void pk_generic(const BPackJob& a, int vb, int tid) {
  const int ntaps = a.p[0], cinc = a.p[1], nchunks = a.p[2], nj = a.p[3], cot = a.p[4], ncob = a.p[5];
  const int cpb = cinc >> 3;
  const int64_t total = (int64_t)ncob * nchunks * nj * cot * 64 * 8;
  for (int64_t e = (int64_t)vb * 256 + tid; e < total; e += (int64_t)a.blocks * 256) {
    const int i = (int)(e & 7);
    int64_t r = e >> 3;
    const int lane = (int)(r & 63); r >>= 6;
    const int c = (int)(r % cot); r /= cot;
    const int j = (int)(r % nj); r /= nj;
    const int ch = (int)(r % nchunks);
    const int cb_ = (int)(r / nchunks);
    const int m = lane & 15, g = lane >> 4;
    const int s = 4 * j + g, t = s / cpb, cb = s - t * cpb;
    const int k = ch * cinc + cb * 8 + i, nn = (cb_ * cot + c) * 16 + m;
    float v = 0.f;
    if (t < ntaps && k < a.Kw && nn < a.Nw) v = pk_w(a, a.tap[t], k, nn);
    a.wp[e] = f2bf(v);
  }
  if (vb == 0 && tid < 8) a.wp[total + tid] = 0;   // the zero piece the conv's LDS-DMA pads with
}

// BPK_B3 (bf16_conv3.hip: input-stationary 8 / 16 channel kernel): p = {CI, CO, WPACK, MT}; tap[(dz+1)*9 + (dy+1)*3 + (dx+1)]
// lane (row = l & 31, k half h = l >> 5) of k step m, row tile mt holds 8 contraction channels of in-plane tap t (CI = 8:
// t = 2 m + h, channels 0..7; CI = 16: t = m, channels 8 h ..) for row (tap plane rg, produced channel co): CO = 8:
// rg = row >> 3 (rows 24..31 zero); CO = 16: tile 0 rg = row >> 4, tile 1 rows 0..15 rg = 2
__device__ void pk_b3(const BPackJob& k, int vb, int tid) {
  const int CI = k.p[0], CO = k.p[1], WPACK = k.p[2], MT = k.p[3];
  const int e = vb * 256 + tid;
  if (e < 8) k.wp[WPACK + e] = 0;   // the zero piece the LDS-DMA staging pads with
  if (e >= WPACK) return;
  const int j = e & 7, lane = (e >> 3) & 63, mm = e >> 9, mt = mm % MT, m = mm / MT;
  const int row = lane & 31, h = lane >> 5;
  const int t = CI == 8 ? 2 * m + h : m, ci = CI == 8 ? j : 8 * h + j;
  int rg, co;
  if (CO == 8) { rg = row >> 3; co = row & 7; }
  else { rg = mt == 0 ? (row >> 4) : (row < 16 ? 2 : 3); co = row & 15; }
  float v = 0.f;
  if (t < 9 && rg < 3 && ci < k.Kw && co < k.Nw) {
    const int tw = k.tap[rg * 9 + t];
    if (tw >= 0) v = pk_w(k, tw, ci, co);
  }
  // the idle slot (tap "9") of the centre tap plane carries the shortcut: dx[co] += sum_j pw[j] * Wsc[co][j]
  if (CI == 8 && k.pw_w && t == 9 && rg == 1 && co < k.Nw) v = k.pw_w[(size_t)co * 8 + j];
  k.wp[e] = f2bf(v);
}

// BPK_CB (bf16_convcb.hip: channel-block kernel): p = {KS, ncob}; tap as BPK_B3
// [cout block][tap plane tz][in-plane tap][k step][lane][8] (+ KS k steps of the fused shortcut) + the zero piece
__device__ void pk_cb(const BPackJob& k, int vb, int tid) {
  const int KS = k.p[0], ncob = k.p[1];
  const int nk = 27 * KS + (k.pw_w ? KS : 0);
  const int total = ncob * nk * 512;
  const int e = vb * 256 + tid;
  if (e < 8) k.wp[total + e] = 0;
  if (e >= total) return;
  const int j = e & 7, lane = (e >> 3) & 63;
  const int r = e >> 9;
  const int slot = r % nk, cob = r / nk;
  const int co = cob * 32 + (lane & 31);
  float v = 0.f;
  if (slot < 27 * KS) {
    const int ks = slot % KS, tt = slot / KS, t = tt % 9, tz = tt / 9;
    const int ci = 16 * ks + 8 * (lane >> 5) + j;
    if (ci < k.Kw && co < k.Nw) {
      const int tw = k.tap[tz * 9 + t];
      if (tw >= 0) v = pk_w(k, tw, ci, co);
    }
  } else {
    const int ci = 16 * (slot - 27 * KS) + 8 * (lane >> 5) + j;
    if (ci < k.Kw && co < k.Nw) v = k.pw_w[(size_t)co * k.Kw + ci];
  }
  k.wp[e] = f2bf(v);
}

// BPK_D3 (bf16_deconv3.hip: 16 -> 8 stride-2 scatter passes): p = {WPACK}; tap[class * 8 + neighbour], -1 = not read
__device__ void pk_d3(const BPackJob& k, int vb, int tid) {
  const int x = vb * 256 + tid;
  if (x >= k.p[0]) return;
  const int j = x & 7, lane = (x >> 3) & 63, mt = (x >> 9) & 1, e = x >> 10;
  const int row = lane & 31, h = lane >> 5, cl = 4 * mt + (row >> 3), co = row & 7, ci = 8 * h + j;
  float v = 0.f;
  const int tw = k.tap[cl * 8 + e];
  if (tw >= 0 && ci < k.Kw && co < k.Nw) v = pk_w(k, tw, ci, co);
  k.wp[x] = f2bf(v);
}

// BPK_DEEP (bf16_convdeep.hip): p = {nchunks, ncob}; [cout block][chunk][tap][co tile][lane = 16 g + m][8]
__device__ void pk_deep(const BPackJob& k, int vb, int tid) {
  const int nchunks = k.p[0], ncob = k.p[1];
  const int64_t total = (int64_t)ncob * nchunks * 27 * 2048;
  for (int64_t e = (int64_t)vb * 256 + tid; e < total; e += (int64_t)k.blocks * 256) {
    const int i = (int)(e & 7), lane = (int)((e >> 3) & 63), mt = (int)((e >> 9) & 3);
    int64_t r = e >> 11;
    const int t = (int)(r % 27); r /= 27;
    const int ch = (int)(r % nchunks), cob = (int)(r / nchunks);
    const int ci = ch * 32 + 8 * (lane >> 4) + i, co = cob * 64 + mt * 16 + (lane & 15);
    float v = 0.f;
    if (ci < k.Kw && co < k.Nw) v = pk_w(k, k.tap[t], ci, co);
    k.wp[e] = f2bf(v);
  }
}

// BPK_SCATTER (bf16_scatter.hip): p = {nchunks, ncob, mt}; [cout block][item][chunk][mt][lane = 16 g + m][8]; tap[item]
__device__ void pk_scatter(const BPackJob& k, int vb, int tid) {
  const int nchunks = k.p[0], ncob = k.p[1], MT = k.p[2];
  const int64_t total = (int64_t)ncob * 27 * nchunks * MT * 512;
  for (int64_t e = (int64_t)vb * 256 + tid; e < total; e += (int64_t)k.blocks * 256) {
    const int i = (int)(e & 7), lane = (int)((e >> 3) & 63);
    int64_t r = e >> 9;
    const int mt = (int)(r % MT); r /= MT;
    const int ch = (int)(r % nchunks); r /= nchunks;
    const int it = (int)(r % 27), cob = (int)(r / 27);
    const int ci = ch * 32 + 8 * (lane >> 4) + i, co = (cob * MT + mt) * 16 + (lane & 15);
    float v = 0.f;
    if (ci < k.Kw && co < k.Nw) v = pk_w(k, k.tap[it], ci, co);
    k.wp[e] = f2bf(v);
  }
}

// BPK_PAD8: p = {count}: 8 fp32 values at wp = the first `count` values of w, zero beyond (conv2's beta as an 8-channel piece)
__device__ void pk_pad8(const BPackJob& k, int vb, int tid) {
  if (vb == 0 && tid < 8) ((float*)k.wp)[tid] = tid < k.p[0] ? k.w[tid] : 0.f;
}

// BPK_C0 (bf16_conv0.hip: one input channel): [tap plane dz][lane = 16 g + m][4]: k = 4 g + j is the tap (dy = g - 1, dx = j - 1)
// of plane dz for produced channel m; k group 3 and j = 3 are zero.  tap[(dz+1)*9 + (dy+1)*3 + (dx+1)] = stored tap index
__device__ void pk_c0(const BPackJob& k, int vb, int tid) {
  const int e = vb * 256 + tid;
  if (e >= 768) return;
  const int j = e & 3, lane = (e >> 2) & 63, d = e >> 8;
  const int m = lane & 15, g = lane >> 4;
  float v = 0.f;
  if (g < 3 && j < 3 && m < k.Nw) v = pk_w(k, k.tap[d * 9 + g * 3 + j], 0, m);
  k.wp[e] = f2bf(v);
}

// BPK_S2K8 (bf16_s2k8.hip: stride-2 gather 8 -> 16): [fragment 0..7][lane = 16 g + m][8]: fragments 0..6: k slot 4 j + g = tap (slot 27
// zero), 8 channels, produced channel m; fragment 7: the 1x1 shortcut's weights pw_w[ci][16] in k slot 27 only (or zeros)
__device__ void pk_s2k8(const BPackJob& k, int vb, int tid) {
  const int e = vb * 256 + tid;
  if (e >= 4096) return;
  const int i = e & 7, lane = (e >> 3) & 63, j = e >> 9;
  const int m = lane & 15, g = lane >> 4;
  float v = 0.f;
  if (j < 7) {
    const int t = 4 * j + g;
    if (t < 27 && i < k.Kw && m < k.Nw) v = pk_w(k, k.tap[t], i, m);
  } else if (g == 3 && k.pw_w && i < k.Kw && m < k.Nw) {
    v = k.pw_w[i * 16 + m];
  }
  k.wp[e] = f2bf(v);
}

__device__ __forceinline__ void pk_run(const BPackJob& j, int vb, int tid) {
  switch (j.type) {
    case BPK_GENERIC: pk_generic(j, vb, tid); break;
    case BPK_B3: pk_b3(j, vb, tid); break;
    case BPK_CB: pk_cb(j, vb, tid); break;
    case BPK_D3: pk_d3(j, vb, tid); break;
    case BPK_DEEP: pk_deep(j, vb, tid); break;
    case BPK_PAD8: pk_pad8(j, vb, tid); break;
    case BPK_C0: pk_c0(j, vb, tid); break;
    case BPK_S2K8: pk_s2k8(j, vb, tid); break;
    default: pk_scatter(j, vb, tid); break;
  }
}

__global__ __launch_bounds__(256) void bpack_one_kernel(BPackJob j) { pk_run(j, blockIdx.x, threadIdx.x); }

// block b belongs to the job whose [first[i], first[i + 1]) holds it
__global__ __launch_bounds__(256) void bpack_multi_kernel(const BPackJob* __restrict__ jobs, const int* __restrict__ first, int njobs) {
  const int b = blockIdx.x;
  int lo = 0, hi = njobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (first[mid] <= b) lo = mid; else hi = mid - 1;
  }
  pk_run(jobs[lo], b - first[lo], threadIdx.x);
}

thread_local BPackCtx* g_ctx = nullptr;

int launch_now(const BPackJob& j, hipStream_t s) {
  hipLaunchKernelGGL(bpack_one_kernel, dim3(j.blocks), dim3(256), 0, s, j);
  URSN_HIP(hipGetLastError());
  return 0;
}

}  // namespace

void bpack_set_ctx(BPackCtx* c) { g_ctx = c; }

int bpack_submit(const BPackJob& j, hipStream_t s) {
  URSN_REQUIRE(j.blocks > 0 && j.w && j.wp, "bf16 weight packing: empty job");
  BPackCtx* c = g_ctx;
  if (!c || !c->d_jobs) return launch_now(j, s);
  auto it = c->by_dest.find(j.wp);
  if (it != c->by_dest.end()) {
    BPackJob& old = c->jobs[it->second];
    const bool same = memcmp(&old, &j, sizeof(BPackJob)) == 0;
    if (same && c->mode == 1 && it->second < c->uploaded) { ++c->launches_saved; return 0; }   // packed by this step's replay
    if (!same) { old = j; c->dirty = true; }
    return launch_now(j, s);
  }
  if ((int)c->jobs.size() < c->cap) {   // (a full table: the job simply stays a launch of its own)
    c->by_dest[j.wp] = (int)c->jobs.size();
    c->jobs.push_back(j);
    c->dirty = true;
  }
  return launch_now(j, s);
}

int bpack_replay(BPackCtx& c, hipStream_t s) {
  if (!c.d_jobs || c.jobs.empty()) { c.mode = 0; return 0; }
  if (c.dirty) {
    std::vector<int> first(c.jobs.size() + 1);
    int b = 0;
    for (size_t i = 0; i < c.jobs.size(); ++i) { first[i] = b; b += c.jobs[i].blocks; }
    first[c.jobs.size()] = b;
    // rare (second step at a batch size, or a job seen for the first time): kernels of the previous replay may still read the table
    URSN_HIP(hipStreamSynchronize(s));
    URSN_HIP(hipMemcpy(c.d_jobs, c.jobs.data(), c.jobs.size() * sizeof(BPackJob), hipMemcpyHostToDevice));
    URSN_HIP(hipMemcpy(c.d_first, first.data(), first.size() * sizeof(int), hipMemcpyHostToDevice));
    c.uploaded = (int)c.jobs.size();
    c.total_blocks = b;
    c.dirty = false;
  }
  hipLaunchKernelGGL(bpack_multi_kernel, dim3(c.total_blocks), dim3(256), 0, s, c.d_jobs, c.d_first, c.uploaded);
  URSN_HIP(hipGetLastError());
  c.mode = 1;
  return 0;
}
