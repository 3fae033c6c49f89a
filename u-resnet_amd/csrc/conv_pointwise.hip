// Pointwise (1x1) convolutions of the residual shortcuts (lib/resnet_module.py:25-33): stride 1 (decoder,
// 2C -> C) and stride 2 (encoder, C -> 2C on the even voxels) -- forward, data gradient, weight gradient.
// HBM-bound layers (AI 0.8-42); the gather kernel spends them on 16-byte scattered loads and half-empty
// 16-wide MFMA tiles.
//
//  * pconv_kernel: lane = output voxel, no LDS.  The lane fetches its voxel's Ck channels with Ck/4 float4
//    loads (all in flight at once), multiplies by the register-resident weight matrix with
//    v_mfma_f32_4x4x1_16b (cbsz/abid A-broadcast as in conv_tiled_kernel.h) and stores Cp channels;
//    BN statistics in the epilogue.  FLIP = data gradient (weights transposed at load time).  Stride 2 reads
//    (forward) or read-modify-writes (data gradient) the even voxels of the high-resolution tensor.
//  * pwgrad_kernel: dW[ci][co] = sum_v x[v][ci] dz[v][co] on v_mfma_f32_16x16x4_f32; 256 voxels per step staged
//    into LDS with full 64-byte rows ([quad][voxel] float4), 4 waves split the voxel quads, fixed-order LDS sum.
#include <stdlib.h>
#include <utility>

#include "ursn_common.h"
#include "wave_pivot.h"

typedef float pw_f32x4 __attribute__((ext_vector_type(4)));

template <int... Is, class F>
__device__ __forceinline__ void pw_static_for_impl(std::integer_sequence<int, Is...>, F&& f) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void pw_static_for(F&& f) {
  pw_static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

struct PConvArgs {
  const float* in;
  const float* w;      // stored [cin_w][cout_w]
  float* out;
  double* stats_partial;
  int64_t nvox;        // voxels of the LOW-resolution grid (= output grid for forward, = dy grid for dgrad)
  int lo[3];           // low-res dims (z,y,x)
  int sm[3];           // per-axis stride of the high-res tensor (1 on the unit axis of 2-D problems)
  int stride;          // 1 | 2
  int in_cs, out_cs;
  int cin_w, cout_w;
  int accumulate;
  // split tensor on the contracted (forward: in) or produced (data gradient: out) side: channels >= split live in *2
  int split;             // 0 = none
  const float* in2;
  float* out2;
  int in2_cs, out2_cs;
};

// CK contracted, CP produced.  FLIP=false: in = x (hi grid when stride 2, sampled at even voxels), out = y (lo grid).
// FLIP=true : in = dy (lo grid), out = dx (hi grid, even voxels only when stride 2).
template <int CK, int CP, bool FLIP, bool STATS>
__global__ __launch_bounds__(256) void pconv_kernel(PConvArgs a) {
  constexpr int NQ = CK / 4, CQ = CP / 4, R = (CK + 15) / 16;
  const int tid = threadIdx.x, lane = tid & 63;
  float wreg[CQ][R];
  {
    const int kl = lane >> 2, cl = lane & 3;
#pragma unroll
    for (int cq = 0; cq < CQ; ++cq)
#pragma unroll
      for (int r = 0; r < R; ++r) {
        int k = 16 * r + kl, p = 4 * cq + cl;
        float v = 0.f;
        if (k < CK) v = FLIP ? a.w[(size_t)p * a.cout_w + k] : a.w[(size_t)k * a.cout_w + p];
        wreg[cq][r] = v;
      }
  }
  // BatchNorm moments around a wave-uniform pivot (wave_pivot.h): piv[] lives in SGPRs
  float s1[STATS ? CP : 1], s2[STATS ? CP : 1], piv[STATS ? CP : 1], nsum = 0.f;
  bool have_piv = false;
#pragma unroll
  for (int c = 0; c < (STATS ? CP : 1); ++c) s1[c] = s2[c] = piv[c] = 0.f;

  const int64_t hi_z = (int64_t)a.lo[0] * a.sm[0], hi_y = (int64_t)a.lo[1] * a.sm[1], hi_x = (int64_t)a.lo[2] * a.sm[2];
  // The loop is wave-uniform: v_mfma with the cbsz/abid A-broadcast must run with all 64 lanes active (a masked-off
  // source lane would feed garbage to the whole wave), so tail lanes compute on a clamped voxel and skip the store.
  for (int64_t vb = (int64_t)blockIdx.x * 256 + (tid & ~63); vb < a.nvox; vb += (int64_t)gridDim.x * 256) {
    const bool ok = vb + lane < a.nvox;
    const int64_t v = ok ? vb + lane : a.nvox - 1;
    // low-res voxel v = ((n*Z + z)*Y + y)*X + x  ->  high-res voxel ((n*sZ + sz)*sY + sy)*sX + sx
    int64_t hv = v;
    if (a.stride == 2) {
      int x = (int)(v % a.lo[2]);
      int64_t r = v / a.lo[2];
      int y = (int)(r % a.lo[1]);
      r /= a.lo[1];
      int z = (int)(r % a.lo[0]);
      int64_t n = r / a.lo[0];
      hv = ((n * hi_z + a.sm[0] * z) * hi_y + a.sm[1] * y) * hi_x + a.sm[2] * x;
    }
    const float* ip = a.in + (FLIP ? v : hv) * a.in_cs;
    float* op = a.out + (FLIP ? hv : v) * a.out_cs;
    const int sq = a.split >> 2;   // first channel quad held by the second tensor (split on in: forward, on out: dgrad)
    const float* ip2 = (!FLIP && a.split) ? a.in2 + hv * a.in2_cs - a.split : ip;
    float* op2 = (FLIP && a.split) ? a.out2 + hv * a.out2_cs - a.split : op;
    pw_f32x4 xv[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) xv[q] = *(const pw_f32x4*)(((!FLIP && a.split && q >= sq) ? ip2 : ip) + 4 * q);
    pw_f32x4 acc[CQ];
#pragma unroll
    for (int cq = 0; cq < CQ; ++cq) acc[cq] = (pw_f32x4){0.f, 0.f, 0.f, 0.f};
    pw_static_for<CK>([&](auto K) {
      constexpr int k = decltype(K)::value;
      pw_static_for<CQ>([&](auto C) {
        constexpr int cq = decltype(C)::value;
        acc[cq] = __builtin_amdgcn_mfma_f32_4x4x1f32(wreg[cq][k / 16], xv[k / 4][k % 4], acc[cq], 4, k % 16, 0);
      });
    });
    if constexpr (STATS) {
      if (!have_piv && a.stats_partial) {   // wave-uniform, once: lane 0 of a wave's first iteration is always a real voxel
#pragma unroll
        for (int cq = 0; cq < CQ; ++cq) {
          pw_f32x4 val = acc[cq];
          if (a.accumulate) val += *(const pw_f32x4*)(op + 4 * cq);   // tail lanes sit on a clamped, readable voxel
#pragma unroll
          for (int j = 0; j < 4; ++j) piv[4 * cq + j] = wave_lane_value(val[j], 0);
        }
        have_piv = true;
      }
    }
    if (!ok) continue;
    nsum += 1.f;
#pragma unroll
    for (int cq = 0; cq < CQ; ++cq) {
      pw_f32x4 val = acc[cq];
      float* o = ((FLIP && a.split && cq >= sq) ? op2 : op) + 4 * cq;
      if (a.accumulate) val += *(pw_f32x4*)o;
      *(pw_f32x4*)o = val;
      if constexpr (STATS) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float d = val[j] - piv[4 * cq + j];
          s1[4 * cq + j] += d;
          s2[4 * cq + j] = __builtin_fmaf(d, d, s2[4 * cq + j]);
        }
      }
    }
  }
  if constexpr (STATS) if (a.stats_partial) {
    __shared__ double red[4][2 * CP];
    const float nw = wave_sum(nsum);
#pragma unroll
    for (int c = 0; c < CP; ++c) {
      const float u = wave_sum(s1[c]), w2 = wave_sum(s2[c]);
      if (lane == 0) wave_unpivot(u, w2, nw, piv[c], red[tid >> 6][c], red[tid >> 6][CP + c]);
    }
    __syncthreads();
    if (tid < 2 * CP) a.stats_partial[(size_t)blockIdx.x * 2 * CP + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
  }
}

// ---- weight gradient ------------------------------------------------------------------------------------------
struct PWgradArgs {
  const float* x;   // hi grid when stride 2
  const float* dz;  // lo grid
  float* slab;      // [grid][cin][cout]
  int64_t nvox;     // lo voxels
  int lo[3];
  int sm[3];
  int stride;
  int x_cs, dz_cs;
  int64_t vox_per_block;
  int split;        // channels >= split of x live in x2 (0 = none)
  const float* x2;
  int x2_cs;
};

template <int CIN, int COUT>
__global__ __launch_bounds__(256) void pwgrad_kernel(PWgradArgs a) {
  constexpr int XQ = CIN / 4, DQ = COUT / 4, SP = 258;       // plane stride == 2 (mod 8): conflict-free operand reads
  constexpr int MT = (CIN + 15) / 16, NTl = (COUT + 15) / 16;   // 8-channel sides use half of a 16-wide tile
  constexpr int NSX = (XQ * 256 + 255) / 256, NSD = (DQ * 256 + 255) / 256;
  __shared__ __attribute__((aligned(16))) float xl[XQ * SP * 4];
  __shared__ __attribute__((aligned(16))) float dl[DQ * SP * 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int il = lane & 15, kl = lane >> 4;
  pw_f32x4 acc[MT][NTl];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int c = 0; c < NTl; ++c) acc[m][c] = (pw_f32x4){0.f, 0.f, 0.f, 0.f};
  const int64_t hi_z = (int64_t)a.lo[0] * a.sm[0], hi_y = (int64_t)a.lo[1] * a.sm[1], hi_x = (int64_t)a.lo[2] * a.sm[2];
  const int64_t v_begin = (int64_t)blockIdx.x * a.vox_per_block;
  int64_t v_end = v_begin + a.vox_per_block;
  if (v_end > a.nvox) v_end = a.nvox;
  const int a_lane = ((il >> 2) * SP + kl) * 4 + (il & 3);
  // the next step's operands are requested before this step's MFMAs (round 4: loaded and stored to LDS back to back the kernel
  // paid a full memory round trip per 256-voxel step: 4.4 TB/s on the level-0 shortcut, whose operands are 2.7 GB)
  pw_f32x4 xr[NSX], dr[NSD];
  auto load_step = [&](int64_t v0) {
#pragma unroll
    for (int i = 0; i < NSX; ++i) {
      int idx = tid + i * 256;
      int s = idx / XQ, q = idx - s * XQ;
      int64_t v = v0 + s;
      pw_f32x4 val = {0.f, 0.f, 0.f, 0.f};
      if (v < v_end) {
        int64_t hv = v;
        if (a.stride == 2) {
          int x = (int)(v % a.lo[2]);
          int64_t r = v / a.lo[2];
          int y = (int)(r % a.lo[1]);
          r /= a.lo[1];
          int z = (int)(r % a.lo[0]);
          int64_t n = r / a.lo[0];
          hv = ((n * hi_z + a.sm[0] * z) * hi_y + a.sm[1] * y) * hi_x + a.sm[2] * x;
        }
        val = __builtin_nontemporal_load((a.split && 4 * q >= a.split) ? (const pw_f32x4*)(a.x2 + hv * a.x2_cs + 4 * q - a.split)
                                                                        : (const pw_f32x4*)(a.x + hv * a.x_cs + 4 * q));
      }
      xr[i] = val;
    }
#pragma unroll
    for (int i = 0; i < NSD; ++i) {
      int idx = tid + i * 256;
      int s = idx / DQ, q = idx - s * DQ;
      int64_t v = v0 + s;
      pw_f32x4 val = {0.f, 0.f, 0.f, 0.f};
      if (v < v_end) val = __builtin_nontemporal_load((const pw_f32x4*)(a.dz + v * a.dz_cs + 4 * q));
      dr[i] = val;
    }
  };
  if (v_begin < v_end) load_step(v_begin);
  for (int64_t v0 = v_begin; v0 < v_end; v0 += 256) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NSX; ++i) {
      int idx = tid + i * 256;
      int s = idx / XQ, q = idx - s * XQ;
      *(pw_f32x4*)(xl + ((size_t)q * SP + s) * 4) = xr[i];
    }
#pragma unroll
    for (int i = 0; i < NSD; ++i) {
      int idx = tid + i * 256;
      int s = idx / DQ, q = idx - s * DQ;
      *(pw_f32x4*)(dl + ((size_t)q * SP + s) * 4) = dr[i];
    }
    if (v0 + 256 < v_end) load_step(v0 + 256);
    __syncthreads();
#pragma unroll 4
    for (int ks = wave; ks < 64; ks += 4) {   // voxel quads of this step, interleaved over the 4 waves
      float av[MT], bv[NTl];
#pragma unroll
      for (int m = 0; m < MT; ++m) av[m] = (16 * m + il < CIN) ? xl[a_lane + ((size_t)m * 4 * SP + ks * 4) * 4] : 0.f;
#pragma unroll
      for (int c = 0; c < NTl; ++c) bv[c] = (16 * c + il < COUT) ? dl[a_lane + ((size_t)c * 4 * SP + ks * 4) * 4] : 0.f;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int c = 0; c < NTl; ++c) acc[m][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], bv[c], acc[m][c], 0, 0, 0);
    }
  }
  // fixed-order sum of the 4 waves in LDS, one slab per workgroup
  __syncthreads();
  float* red = xl;  // CIN*COUT floats <= XQ*SP*4 (CIN*COUT <= 1032*CIN/4 holds for COUT <= 256)
  for (int i = tid; i < CIN * COUT; i += 256) red[i] = 0.f;
  __syncthreads();
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int c = 0; c < NTl; ++c)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (16 * m + 4 * kl + r < CIN && 16 * c + il < COUT) red[(16 * m + 4 * kl + r) * COUT + 16 * c + il] += acc[m][c][r];
    }
    __syncthreads();
  }
  float* slab = a.slab + (size_t)blockIdx.x * CIN * COUT;
  for (int i = tid; i < CIN * COUT; i += 256) slab[i] = red[i];
}

// ---- host --------------------------------------------------------------------------------------------------------
static bool pw_enabled() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("URSN_DISABLE_TILED");
    const char* f = getenv("URSN_POINTWISE");
    v = ((e && e[0] == '1') || (f && f[0] == '0')) ? 0 : 1;
  }
  return v == 1;
}

static bool pw_geometry(const ursn_conv_desc& d, int lo[3], int sm[3], int64_t& nvox) {
  if (d.transposed || d.k != 1 || d.in_mean) return false;
  nvox = d.n;
  for (int j = 0; j < 3; ++j) { lo[j] = 1; sm[j] = 1; }
  for (int j = 0; j < d.ndim; ++j) {
    int sz = d.in_sp[j];
    if (d.stride == 2 && (sz & 1)) return false;
    lo[3 - d.ndim + j] = sz / d.stride;
    sm[3 - d.ndim + j] = d.stride;
    nvox *= sz / d.stride;
  }
  const int ics = d.in_cstride > 0 ? d.in_cstride : d.cin, ocs = d.out_cstride > 0 ? d.out_cstride : d.cout;
  if (d.in_split) {
    const int i2 = d.in2_cstride > 0 ? d.in2_cstride : d.cin - d.in_split;
    if (d.stride != 1 || (d.in_split & 3) || d.in_split >= d.cin || (i2 & 3)) return false;
  }
  return !((ics & 3) || (ocs & 3));
}

static bool pw_shape(int ck, int cp) {
  return (ck == 8 && cp == 16) || (ck == 16 && cp == 8) || (ck == 16 && cp == 32) || (ck == 32 && cp == 16) ||
         (ck == 32 && cp == 64) || (ck == 64 && cp == 32);
}

int pointwise_conv_supported(const ursn_conv_desc& d, ConvPass pass, int accumulate) {
  if (!pw_enabled() && d.algo != 5) return 0;
  int lo[3], sm[3];
  int64_t nvox;
  if (!pw_geometry(d, lo, sm, nvox)) return 0;
  if (pass == PASS_FWD) return pw_shape(d.cin, d.cout) ? 1 : 0;
  if (pass == PASS_DGRAD) {
    if (d.stride == 2 && !accumulate) return 0;   // odd voxels would have to be zero-filled
    return pw_shape(d.cout, d.cin) ? 1 : 0;
  }
  return 0;
}

static int pconv_grid_cap() {   // workgroups of the forward / data-gradient kernel (URSN_PCONV_GRID, A/B)
  static const int cap = getenv("URSN_PCONV_GRID") ? atoi(getenv("URSN_PCONV_GRID")) : 1024;
  return cap < 1 ? 1 : cap;
}

size_t pointwise_stats_scratch_doubles(const ursn_conv_desc& d) {
  if (!pointwise_conv_supported(d, PASS_FWD, 0)) return 0;
  return (size_t)pconv_grid_cap() * 2 * d.cout;
}

template <int CK, int CP>
static int pconv_launch(const PConvArgs& a, bool flip, int grid, hipStream_t s) {
  if (flip) hipLaunchKernelGGL((pconv_kernel<CK, CP, true, false>), dim3(grid), dim3(256), 0, s, a);
  else if (a.stats_partial) hipLaunchKernelGGL((pconv_kernel<CK, CP, false, true>), dim3(grid), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((pconv_kernel<CK, CP, false, false>), dim3(grid), dim3(256), 0, s, a);
  URSN_HIP(hipGetLastError());
  return 0;
}

int launch_pointwise_conv(const ursn_conv_desc& d, ConvPass pass, const float* in, const float* w, float* out,
                          int accumulate, double* stats_partial, float eps, float* mean, float* rstd, hipStream_t s) {
  PConvArgs a;
  URSN_REQUIRE(pw_geometry(d, a.lo, a.sm, a.nvox), "pointwise conv: unsupported geometry");
  const bool flip = pass == PASS_DGRAD;
  const int ics = d.in_cstride > 0 ? d.in_cstride : d.cin, ocs = d.out_cstride > 0 ? d.out_cstride : d.cout;
  a.in = in; a.w = w; a.out = out; a.stats_partial = stats_partial;
  a.stride = d.stride;
  a.in_cs = flip ? ocs : ics;
  a.out_cs = flip ? ics : ocs;
  a.cin_w = d.cin; a.cout_w = d.cout;
  a.accumulate = accumulate;
  a.split = d.in_split; a.in2 = nullptr; a.out2 = nullptr; a.in2_cs = a.out2_cs = 0;
  if (d.in_split) {
    const int i2 = d.in2_cstride > 0 ? d.in2_cstride : d.cin - d.in_split;
    if (flip) { URSN_REQUIRE(d.dx2, "pointwise conv: split input without dx2"); a.out2 = d.dx2; a.out2_cs = i2; }
    else { URSN_REQUIRE(d.x2, "pointwise conv: split input without x2"); a.in2 = d.x2; a.in2_cs = i2; }
  }
  const int ck = flip ? d.cout : d.cin, cp = flip ? d.cin : d.cout;
  const int cap = pconv_grid_cap();
  int64_t blocks = cdiv64(a.nvox, 256 * 4);
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  const int grid = (int)blocks;
  ursn_note_kernel(flip ? "pconv_dgrad" : "pconv");
  int rc = 3;
  if (ck == 8 && cp == 16) rc = pconv_launch<8, 16>(a, flip, grid, s);
  else if (ck == 16 && cp == 8) rc = pconv_launch<16, 8>(a, flip, grid, s);
  else if (ck == 16 && cp == 32) rc = pconv_launch<16, 32>(a, flip, grid, s);
  else if (ck == 32 && cp == 16) rc = pconv_launch<32, 16>(a, flip, grid, s);
  else if (ck == 32 && cp == 64) rc = pconv_launch<32, 64>(a, flip, grid, s);
  else if (ck == 64 && cp == 32) rc = pconv_launch<64, 32>(a, flip, grid, s);
  else ursn_set_error("pointwise conv: no instantiation for %d->%d", ck, cp);
  if (rc) return rc;
  if (stats_partial) return launch_bn_stats_final(stats_partial, grid, cp, cp, a.nvox, eps, mean, rstd, s);
  return 0;
}

int pointwise_wgrad_supported(const ursn_conv_desc& d) {
  if (!pw_enabled() && d.algo != 5) return 0;
  int lo[3], sm[3];
  int64_t nvox;
  if (!pw_geometry(d, lo, sm, nvox)) return 0;
  return ((d.cin % 16) == 0 && (d.cout % 16) == 0 && d.cin <= 64 && d.cout <= 64 && d.cin * d.cout <= 2048) ||
         (d.cin == 16 && d.cout == 8) || (d.cin == 8 && d.cout == 16);
}

static int pw_blocks(int64_t nvox) {
  int64_t b = cdiv64(nvox, 256 * 8);
  if (b > 1024) b = 1024;
  if (b < 1) b = 1;
  return (int)b;
}
size_t pointwise_wgrad_scratch_bytes(const ursn_conv_desc& d) {
  int lo[3], sm[3];
  int64_t nvox;
  if (!pointwise_wgrad_supported(d) || !pw_geometry(d, lo, sm, nvox)) return 0;
  return (size_t)pw_blocks(nvox) * 32 * 64 * sizeof(float);
}

template <int CIN, int COUT>
static int pwgrad_launch(const PWgradArgs& a, int grid, hipStream_t s) {
  hipLaunchKernelGGL((pwgrad_kernel<CIN, COUT>), dim3(grid), dim3(256), 0, s, a);
  URSN_HIP(hipGetLastError());
  return 0;
}

int launch_pointwise_wgrad(const ursn_conv_desc& d, const float* x, const float* dy, float* dw, void* scratch,
                           size_t scratch_bytes, hipStream_t s) {
  PWgradArgs a;
  URSN_REQUIRE(pw_geometry(d, a.lo, a.sm, a.nvox), "pointwise wgrad: unsupported geometry");
  const int grid = pw_blocks(a.nvox);
  URSN_REQUIRE(scratch && scratch_bytes >= (size_t)grid * d.cin * d.cout * sizeof(float), "pointwise wgrad: scratch too small");
  a.x = x; a.dz = dy; a.slab = (float*)scratch;
  a.stride = d.stride;
  a.x_cs = d.in_cstride > 0 ? d.in_cstride : d.cin;
  a.dz_cs = d.out_cstride > 0 ? d.out_cstride : d.cout;
  a.vox_per_block = cdiv64(cdiv64(a.nvox, grid), 256) * 256;
  a.split = d.in_split; a.x2 = d.x2; a.x2_cs = d.in2_cstride > 0 ? d.in2_cstride : d.cin - d.in_split;
  URSN_REQUIRE(!d.in_split || d.x2, "pointwise wgrad: split input without x2");
  ursn_note_kernel("pwgrad");
  int rc = 3;
  if (d.cin == 16 && d.cout == 8) rc = pwgrad_launch<16, 8>(a, grid, s);
  else if (d.cin == 8 && d.cout == 16) rc = pwgrad_launch<8, 16>(a, grid, s);
  else if (d.cin == 16 && d.cout == 16) rc = pwgrad_launch<16, 16>(a, grid, s);
  else if (d.cin == 16 && d.cout == 32) rc = pwgrad_launch<16, 32>(a, grid, s);
  else if (d.cin == 32 && d.cout == 16) rc = pwgrad_launch<32, 16>(a, grid, s);
  else if (d.cin == 32 && d.cout == 32) rc = pwgrad_launch<32, 32>(a, grid, s);
  else if (d.cin == 32 && d.cout == 64) rc = pwgrad_launch<32, 64>(a, grid, s);
  else if (d.cin == 64 && d.cout == 32) rc = pwgrad_launch<64, 32>(a, grid, s);
  else ursn_set_error("pointwise wgrad: no instantiation for %d->%d", d.cin, d.cout);
  if (rc) return rc;
  return launch_reduce_accum(dw, (const float*)scratch, (int64_t)d.cin * d.cout, grid, s);
}
