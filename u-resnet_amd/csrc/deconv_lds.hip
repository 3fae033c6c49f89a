// LDS-staged stride-2 SCATTER-type convolutions for the mid levels (channels multiples of 16) -- gfx950.
//
//     transposed conv k3 s2 forward:  y[o]  = sum_{i,t: 2i + t = o} x[i]  . Wd[t][co][ci]      (lo-res x  -> hi-res y)
//     conv k3 s2 data gradient:       dx[p] = sum_{o,t: 2o + t = p} dy[o] . W [t][ci][co]      (lo-res dy -> hi-res dx)
// Both read the weights as [t][produced][contraction].  Per axis an even output 2j takes tap 0 from input j and tap 2
// from input j-1, an odd output 2j+1 takes tap 1 from input j: the 27 (9) taps fall into 8 (4) output-parity classes.
//
// Workgroup = 4 waves, a box of lo-res voxels (3-D 2z x 4y x 16x, 2-D 16y x 16x) x 16 produced channels, i.e. a
// 4 x 8 x 32 (32 x 32) block of the hi-res output.  Per 16-channel chunk of the contraction the lo-res box (+1 halo
// towards -1) and ALL taps of the 16 x 16 weight slab are staged once; every tap is one v_mfma_f32_16x16x4_f32 stream
// into the accumulator tile of its parity class (16 tiles per wave), no barrier inside a chunk, next chunk prefetched
// into registers.  The tiled lane-per-voxel kernel (deconv_tiled_kernel.h) keeps the narrow level-0 layers.
#include <stdlib.h>
#include <utility>

#include "ursn_common.h"

typedef float sc_f32x4 __attribute__((ext_vector_type(4)));

template <int MODE> struct ScBox;
template <> struct ScBox<3> { static constexpr int BZ = 2, BY = 4, BX = 16, NT = 27, KZ = 3, NCLS = 8; };
template <> struct ScBox<2> { static constexpr int BZ = 1, BY = 16, BX = 16, NT = 9, KZ = 1, NCLS = 4; };

struct ScArgs {
  const float* in;        // lo-res tensor
  const float* w;         // [t][M][K]
  float* out;             // hi-res tensor (dims 2x lo-res)
  double* stats_partial;  // [grid.y][grid.x][2][16] or null
  int N, IZ, IY, IX;      // lo-res dims (2-D: IZ = 1)
  int K, M;               // contraction / produced channels
  int in_cs, out_cs;
  int nbz, nby, nbx;
  int accumulate;
};

template <int N, class F, int... I>
__device__ __forceinline__ void sc_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void sc_static_for(F&& f) {
  sc_static_for_impl<N>(f, std::make_integer_sequence<int, N>{});
}

template <int MODE, bool STATS>
__global__ __launch_bounds__(256, 2) void s2scatter_kernel(ScArgs a) {
  using B = ScBox<MODE>;
  constexpr int BZ = B::BZ, BY = B::BY, BX = B::BX, NT = B::NT, KZ = B::KZ, NCLS = B::NCLS;
  constexpr int HZ = (KZ == 3) ? BZ + 1 : 1, HY = BY + 1, HX = BX + 1, PS = HZ * HY * HX;
  constexpr int NH = (4 * PS + 255) / 256;
  constexpr int NW = (NT * 64 + 255) / 256;      // float4 of the [NT][16][16] weight slab per thread
  constexpr int NR = (BZ * BY) / 4;              // 16-voxel rows per wave
  extern __shared__ __attribute__((aligned(16))) float scl[];  // [4][PS][4] lo-res box, then [NT][16 k][16 m] weights
  float* hal = scl;
  float* wl = scl + 4 * PS * 4;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int il = lane & 15, kl = lane >> 4;
  int bid = ursn_xcd_block(blockIdx.x, gridDim.x);
  const int bx = bid % a.nbx; bid /= a.nbx;
  const int by = bid % a.nby; bid /= a.nby;
  const int bz = bid % a.nbz;
  const int n = bid / a.nbz;
  const int x0 = bx * BX, y0 = by * BY, z0 = bz * BZ;
  const int m0 = blockIdx.y * 16;

  sc_f32x4 acc[NCLS][NR];
#pragma unroll
  for (int c = 0; c < NCLS; ++c)
#pragma unroll
    for (int r = 0; r < NR; ++r) acc[c][r] = (sc_f32x4){0.f, 0.f, 0.f, 0.f};

  // box slot (hz, hy, hx) <-> lo-res voxel (z0 - 1 + hz, y0 - 1 + hy, x0 - 1 + hx)
  auto load_h = [&](int k0, sc_f32x4 (&hv)[NH]) {
#pragma unroll
    for (int i = 0; i < NH; ++i) {
      const int idx = tid + i * 256;
      const int s = idx >> 2, q = idx & 3;
      const int hx = s % HX, r = s / HX;
      const int hy = r % HY, hz = r / HY;
      const int pz = (KZ == 3) ? z0 - 1 + hz : 0, py = y0 - 1 + hy, px = x0 - 1 + hx;
      sc_f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (idx < 4 * PS && pz >= 0 && pz < a.IZ && py >= 0 && py < a.IY && px >= 0 && px < a.IX)
        v = *(const sc_f32x4*)(a.in + ((((size_t)n * a.IZ + pz) * a.IY + py) * a.IX + px) * a.in_cs + k0 + 4 * q);
      hv[i] = v;
    }
  };
  auto store_h = [&](const sc_f32x4 (&hv)[NH]) {
#pragma unroll
    for (int i = 0; i < NH; ++i) {
      const int idx = tid + i * 256;
      if (idx < 4 * PS) *(sc_f32x4*)(hal + ((size_t)(idx & 3) * PS + (idx >> 2)) * 4) = hv[i];
    }
  };
  // weights [t][m][k]: float4 along k (contiguous), transposed into wl[t][k][m] while storing
  auto load_w = [&](int k0, sc_f32x4 (&wv)[NW]) {
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int idx = tid + i * 256;
      sc_f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (idx < NT * 64) {
        const int t = idx >> 6, rem = idx & 63;
        const int m = rem >> 2, k4 = (rem & 3) * 4;
        v = *(const sc_f32x4*)(a.w + ((size_t)t * a.M + m0 + m) * a.K + k0 + k4);
      }
      wv[i] = v;
    }
  };
  auto store_w = [&](const sc_f32x4 (&wv)[NW]) {
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int idx = tid + i * 256;
      if (idx < NT * 64) {
        const int t = idx >> 6, rem = idx & 63;
        const int m = rem >> 2, k4 = (rem & 3) * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) wl[((size_t)t * 16 + k4 + j) * 16 + m] = wv[i][j];
      }
    }
  };

  // slot of this lane's lo-res voxel (shift 0) per row
  int hb[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int row = NR * wave + r;
    const int lz = (MODE == 3) ? row / BY : 0, ly = (MODE == 3) ? row % BY : row;
    hb[r] = ((((KZ == 3) ? lz + 1 : 0) * HY + ly + 1) * HX + il + 1) * 4 + kl;
  }
  const int wb = kl * 16 + il;

  sc_f32x4 hv[NH], wv[NW];
  const int nchunks = a.K / 16;
  load_h(0, hv);
  load_w(0, wv);
  for (int ch = 0; ch < nchunks; ++ch) {
    if (ch) __syncthreads();
    store_h(hv);
    store_w(wv);
    __syncthreads();
    if (ch + 1 < nchunks) {
      load_h(16 * (ch + 1), hv);
      load_w(16 * (ch + 1), wv);
    }
    sc_static_for<NT>([&](auto T) {
      constexpr int t = decltype(T)::value;
      constexpr int tz = (KZ == 3) ? t / 9 : 0, ty = (t / 3) % 3, tx = t % 3;
      // per axis: tap 0 -> even outputs from input j, tap 1 -> odd outputs from input j, tap 2 -> even from j - 1
      constexpr int cls = ((KZ == 3 && tz == 1) ? 4 : 0) + (ty == 1 ? 2 : 0) + (tx == 1 ? 1 : 0);
      constexpr int doff = (((KZ == 3 && tz == 2) ? -1 : 0) * HY + (ty == 2 ? -1 : 0)) * HX + (tx == 2 ? -1 : 0);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float av = wl[wb + (t * 16 + 4 * s) * 16];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          const float bv = hal[hb[r] + (s * PS + doff) * 4];
          acc[cls][r] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[cls][r], 0, 0, 0);
        }
      }
    });
  }

  // epilogue: lane (il, kl) holds channels m0 + 4kl + j of the hi-res voxels 2*(lo voxel) + class offset
  // BatchNorm moments around a per-lane pivot (ursn_common.h: shifted one-pass moments); re-centred in fp64 below
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f}, piv[4] = {0.f, 0.f, 0.f, 0.f}, nacc = 0.f;
  const int OZ = (KZ == 3) ? 2 * a.IZ : 1, OY = 2 * a.IY, OX = 2 * a.IX;
  const int lx = x0 + il;
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int row = NR * wave + r;
    const int lz = (MODE == 3) ? z0 + row / BY : 0, ly = (MODE == 3) ? y0 + row % BY : y0 + row;
    if (!(lz < a.IZ && ly < a.IY && lx < a.IX)) continue;
#pragma unroll
    for (int c = 0; c < NCLS; ++c) {
      const int oz = (KZ == 3) ? 2 * lz + ((c >> 2) & 1) : 0, oy = 2 * ly + ((c >> 1) & 1), ox = 2 * lx + (c & 1);
      float* op = a.out + ((((size_t)n * OZ + oz) * OY + oy) * OX + ox) * a.out_cs + m0 + 4 * kl;
      sc_f32x4 val = acc[c][r];
      if (a.accumulate) val += *(sc_f32x4*)op;
      *(sc_f32x4*)op = val;
      if constexpr (STATS) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (nacc == 0.f) piv[j] = val[j];
          ursn_sacc(piv[j], s1[j], s2[j], val[j]);
        }
        nacc += 1.f;
      }
    }
  }
  if constexpr (STATS) {
    __shared__ double red[4][32];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      double u, w2;
      ursn_sacc_final(piv[j], s1[j], s2[j], nacc, u, w2);
#pragma unroll
      for (int o = 8; o >= 1; o >>= 1) { u += __shfl_xor(u, o); w2 += __shfl_xor(w2, o); }
      if (il == 0) {
        red[wave][4 * kl + j] = u;
        red[wave][16 + 4 * kl + j] = w2;
      }
    }
    __syncthreads();
    if (tid < 32)
      a.stats_partial[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 32 + tid] =
          (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
  }
}

// ------------------------------------------------------------------------------------------------------------
struct ScPlan {
  int mode, K, M, in_cs, out_cs;
  int IZ, IY, IX, nbz, nby, nbx, gridx;
  size_t lds;
};

static bool make_scplan(const ursn_conv_desc& d, ConvPass pass, ScPlan& p) {
  {
    static int off = -1;
    if (off < 0) {
      const char* e = getenv("URSN_DISABLE_TILED");
      const char* f = getenv("URSN_SCATTER_LDS");
      off = ((e && e[0] == '1') || (f && f[0] == '0')) ? 1 : 0;
    }
    if (off && d.algo != 7) return false;
  }
  if (d.in_split || d.pw_dy || d.in_mean) return false;
  const bool fwd_t = d.transposed && pass == PASS_FWD;
  const bool dgrad_s2 = !d.transposed && d.k == 3 && d.stride == 2 && pass == PASS_DGRAD;
  if (!fwd_t && !dgrad_s2) return false;
  if (d.ndim != 2 && d.ndim != 3) return false;
  const int ics = d.in_cstride > 0 ? d.in_cstride : d.cin, ocs = d.out_cstride > 0 ? d.out_cstride : d.cout;
  p.mode = d.ndim;
  p.K = fwd_t ? d.cin : d.cout;        // contraction: channels of the lo-res tensor being read
  p.M = fwd_t ? d.cout : d.cin;        // produced: channels of the hi-res tensor
  p.in_cs = fwd_t ? ics : ocs;
  p.out_cs = fwd_t ? ocs : ics;
  if ((p.K % 16) || (p.M % 16) || (p.in_cs & 3) || (p.out_cs & 3)) return false;
  int lo[3] = {1, 1, 1};
  for (int j = 0; j < d.ndim; ++j) {
    if (fwd_t) lo[3 - d.ndim + j] = d.in_sp[j];
    else {
      if (d.in_sp[j] & 1) return false;   // TF SAME pads 0 before only for even sizes
      lo[3 - d.ndim + j] = d.in_sp[j] / 2;
    }
  }
  p.IZ = lo[0]; p.IY = lo[1]; p.IX = lo[2];
  if (p.IX < 8 && d.algo != 7) return false;
  const int BZ = p.mode == 3 ? 2 : 1, BY = p.mode == 3 ? 4 : 16, BX = 16;
  p.nbz = (p.IZ + BZ - 1) / BZ;
  p.nby = (p.IY + BY - 1) / BY;
  p.nbx = (p.IX + BX - 1) / BX;
  const int64_t nb = (int64_t)d.n * p.nbz * p.nby * p.nbx;
  if (nb > (1 << 30)) return false;
  p.gridx = (int)nb;
  const int HZ = p.mode == 3 ? BZ + 1 : 1, PS = HZ * (BY + 1) * (BX + 1);
  p.lds = ((size_t)16 * PS + (size_t)(p.mode == 3 ? 27 : 9) * 256) * sizeof(float);
  return true;
}

int lds_scatter_supported(const ursn_conv_desc& d, ConvPass pass) {
  ScPlan p;
  return make_scplan(d, pass, p) ? 1 : 0;
}

size_t lds_scatter_stats_scratch_doubles(const ursn_conv_desc& d) {
  ScPlan p;
  if (!make_scplan(d, PASS_FWD, p)) return 0;
  return (size_t)p.gridx * (p.M / 16) * 32;
}

template <int MODE, bool STATS>
static int launch_sc(const ScPlan& p, const ScArgs& a, hipStream_t s) {
  auto kern = s2scatter_kernel<MODE, STATS>;
  static size_t attr_lds = 48 * 1024;
  if (p.lds > attr_lds) {
    URSN_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds));
    attr_lds = p.lds;
  }
  hipLaunchKernelGGL(kern, dim3(p.gridx, p.M / 16), dim3(256), p.lds, s, a);
  URSN_HIP(hipGetLastError());
  return 0;
}

int launch_lds_scatter(const ursn_conv_desc& d, ConvPass pass, const float* in, const float* w, float* out,
                       int accumulate, double* stats_partial, float eps, float* mean, float* rstd, hipStream_t s) {
  ScPlan p;
  URSN_REQUIRE(make_scplan(d, pass, p), "LDS scatter conv: unsupported shape");
  ScArgs a;
  a.in = in; a.w = w; a.out = out; a.stats_partial = stats_partial;
  a.N = d.n; a.IZ = p.IZ; a.IY = p.IY; a.IX = p.IX;
  a.K = p.K; a.M = p.M; a.in_cs = p.in_cs; a.out_cs = p.out_cs;
  a.nbz = p.nbz; a.nby = p.nby; a.nbx = p.nbx;
  a.accumulate = accumulate;
  ursn_note_kernel(d.transposed ? "s2scatter(deconv)" : "s2scatter(s2 dgrad)");
  int rc;
  if (p.mode == 3) rc = stats_partial ? launch_sc<3, true>(p, a, s) : launch_sc<3, false>(p, a, s);
  else rc = stats_partial ? launch_sc<2, true>(p, a, s) : launch_sc<2, false>(p, a, s);
  if (rc) return rc;
  if (stats_partial) {
    const int64_t V = (int64_t)d.n * p.IZ * p.IY * p.IX * (p.mode == 3 ? 8 : 4);
    URSN_TRY(launch_bn_stats_final_blocked(stats_partial, p.gridx, p.M, 16, 16, (size_t)p.gridx * 32, V, eps, mean, rstd, s));
  }
  return 0;
}
