// LDS-tiled weight gradient for the full-resolution small-channel k3 s1 layers -- gfx950.
//
//   dW[t][ci][co] = sum_{n,v} x[n, v + t - 1][ci] * dz[n, v][co]
//
// v_mfma_f32_16x16x4_f32 with M = (tap, ci) rows, N = co columns, K = 4 consecutive voxels:
//   A (16 rows x 4 vox): lane l -> row (l&15), voxel (l>>4): x[v0 + (l>>4) + tap(row)][ci(row)]
//   B (4 vox x 16 cols): lane l -> voxel (l>>4), col (l&15):  dz[v0 + (l>>4)][co]
//   D (16 x 16)        : lane l, reg r -> row 4*(l>>4)+r, col l&15
// One wave keeps the accumulators of ALL taps (27*Cin/16 tiles, or two taps per tile when Cin = 8), so
// every x plane and dz plane is read from HBM exactly once.  A workgroup owns a TYxTX tile and marches
// along the slowest axis with a ring of 4 x planes (halo included) + a double-buffered dz plane in LDS,
// voxel-major ([slot][channel], a straight copy of the NDHWC rows).  Each wave takes a quarter of the
// tile's voxels; the waves are summed in LDS in fixed order, one slab per workgroup goes to scratch and the slabs
// are summed (and ADDED into the gradient buffer, the assign_add of lib/ssnet.py:77) by a deterministic second kernel.
#pragma once
#include <utility>

#include "ursn_common.h"
#include "buffer_stage.h"

typedef float wg_f32x4 __attribute__((ext_vector_type(4)));

struct TWgradArgs {
  const float* x;
  const float* dz;
  float* slab;  // [grid*4][taps][cin][cout]
  int N, Z, Y, X;
  int x_cs, dz_cs;
  int zseg, nzseg, nty, ntx;
  int cout_w;  // stored output channels (<= COUT, which is padded to 4)
  // normalise-on-load of x (x = raw z of the preceding conv): staged value = (z - mean) * rstd + beta; null = none
  const float* aff_mean;
  const float* aff_rstd;
  const float* aff_beta;
};

template <int MODE> struct WTile;
template <> struct WTile<3> { static constexpr int TX = 32, TY = 8, NTY = 3, NT = 27; };
template <> struct WTile<2> { static constexpr int TX = 256, TY = 1, NTY = 1, NT = 9; };

template <int... Is, class F>
__device__ __forceinline__ void wg_static_for_impl(std::integer_sequence<int, Is...>, F&& f) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void wg_static_for(F&& f) {
  wg_static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

template <int CIN, int COUT, int MODE>
__global__ __launch_bounds__(256, 1) void twgrad_kernel(TWgradArgs a) {
  using TL = WTile<MODE>;
  constexpr int TX = TL::TX, TY = TL::TY, NTY = TL::NTY, NT = TL::NT;
  constexpr int PX = TX + 2, PY = TY + (NTY == 3 ? 2 : 0), PS = PX * PY;
  constexpr int MT = CIN >= 16 ? CIN / 16 : 1;                 // row tiles per tap
  // A operands (= 16-row tiles) per voxel group: Cin>=16: Cin/16 per tap; Cin=8: two taps per tile; Cin=1: 16 taps
  constexpr int NA = CIN >= 16 ? NT * MT : (CIN == 1 ? (NT + 15) / 16 : (NT + 1) / 2);
  constexpr int CT = (COUT + 15) / 16;                         // col tiles
  constexpr int XQ = CIN >= 4 ? CIN / 4 : 1, DQ = COUT / 4;    // float4 per voxel (Cin = 1: one scalar)
  constexpr int NSX = (XQ * PS + 255) / 256;                   // staging elements per thread (x plane)
  constexpr int NSD = (DQ * TX * TY + 255) / 256;              // staging float4 per thread (dz plane)
  constexpr int XPLANE = PS * CIN, DPLANE = TX * TY * COUT;    // floats
  constexpr int GPR = TX / 4 / (MODE == 3 ? 1 : 4);            // voxel groups per tile row handled by one wave
  extern __shared__ __attribute__((aligned(16))) float wlds[];  // [4][XPLANE] then [2][DPLANE]
  float* xr = wlds;
  float* dr = wlds + 4 * XPLANE;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int il = lane & 15, kl = lane >> 4;
  int bid = ursn_xcd_block(blockIdx.x, gridDim.x);
  const int xt = bid % a.ntx; bid /= a.ntx;
  const int yt = bid % a.nty; bid /= a.nty;
  const int zs = bid % a.nzseg;
  const int n = bid / a.nzseg;
  const int x0 = xt * TX, y0 = yt * TY;
  const int z0 = zs * a.zseg;
  const int z1 = (z0 + a.zseg < a.Z) ? z0 + a.zseg : a.Z;

  // per-lane operand geometry
  int a_off[NA], a_tz[NA];  // float offset inside an x plane (tap shift + channel), plane selector
#pragma unroll
  for (int m = 0; m < NA; ++m) {
    int tap, ci;
    if (CIN >= 16) { tap = m / MT; ci = (m % MT) * 16 + il; }
    else if (CIN == 1) { tap = 16 * m + il; ci = 0; }
    else { tap = 2 * m + (il >> 3); ci = il & 7; }
    bool ok = tap < NT;
    if (!ok) tap = 0;
    int tz = tap / (NTY * 3), tyy = (tap / 3) % NTY, txx = tap % 3;
    a_tz[m] = ok ? tz : -1;
    a_off[m] = (tyy * PX + txx) * CIN + ci;
  }
  // this wave's first voxel: 3-D: rows 2*wave, 2*wave+1 of the tile; 2-D: x in [64*wave, 64*wave+64)
  const int wrow = (MODE == 3) ? 2 * wave : 0;
  const int wcol = (MODE == 3) ? 0 : 64 * wave;
  const int a_lane = ((wrow * PX) + wcol + kl) * CIN;
  const int b_lane = ((wrow * TX) + wcol + kl) * COUT + il;

  wg_f32x4 acc[NA][CT];
#pragma unroll
  for (int m = 0; m < NA; ++m)
#pragma unroll
    for (int c = 0; c < CT; ++c) acc[m][c] = (wg_f32x4){0.f, 0.f, 0.f, 0.f};

  __shared__ wg_f32x4 aff_s[XQ], aff_t[XQ];
  const bool aff = a.aff_mean != nullptr && CIN >= 4;
  if (aff) {
    if (tid < CIN) {
      const float r = a.aff_rstd[tid];
      ((float*)aff_s)[tid] = r;
      ((float*)aff_t)[tid] = a.aff_beta[tid] - a.aff_mean[tid] * r;
    }
    __syncthreads();
  }
  // staging through buffer loads (buffer_stage.h): tabulated byte offsets, out-of-range elements read as 0
  wg_f32x4 sx[NSX], sd[NSD];
  unsigned sx_inb = 0;   // staged x elements that are real voxels
  unsigned xgo[NSX], dgo[NSD], xin_mask = 0;
#pragma unroll
  for (int i = 0; i < NSX; ++i) {
    const int idx = tid + i * 256;
    const int s = idx / XQ, q = idx - s * XQ;
    const int yy = s / PX, xx = s - yy * PX;
    const int py = y0 + yy - (NTY == 3 ? 1 : 0), px = x0 + xx - 1;
    const bool ok = idx < XQ * PS && py >= 0 && py < a.Y && px >= 0 && px < a.X;
    xgo[i] = ok ? (unsigned)((py * a.X + px) * a.x_cs + 4 * q) * 4u : URSN_OOB_OFFSET;
    if (ok) xin_mask |= 1u << i;
    asm volatile("" : "+v"(xgo[i]));
  }
  asm volatile("" : "+v"(xin_mask));
#pragma unroll
  for (int i = 0; i < NSD; ++i) {
    const int idx = tid + i * 256;
    const int s = idx / DQ, q = idx - s * DQ;
    const int yy = s / TX, xx = s - yy * TX;
    const int py = y0 + yy, px = x0 + xx;
    const bool ok = idx < DQ * TX * TY && py < a.Y && px < a.X;
    dgo[i] = ok ? (unsigned)((py * a.X + px) * a.dz_cs + 4 * q) * 4u : URSN_OOB_OFFSET;
    asm volatile("" : "+v"(dgo[i]));
  }
  const ptrdiff_t xplane_f = (ptrdiff_t)a.Y * a.X * a.x_cs, dplane_f = (ptrdiff_t)a.Y * a.X * a.dz_cs;
  const unsigned xplane_b = (unsigned)xplane_f * 4u, dplane_b = (unsigned)dplane_f * 4u;
  const float* ximg = a.x + (size_t)n * a.Z * xplane_f;
  const float* dimg = a.dz + (size_t)n * a.Z * dplane_f;
  auto load_x = [&](int zin) {
    const bool zok = zin >= 0 && zin < a.Z;
    sx_inb = zok ? xin_mask : 0u;
    const __amdgpu_buffer_rsrc_t r = ursn_plane_rsrc(ximg + (ptrdiff_t)zin * xplane_f, zok ? xplane_b : 0u);
#pragma unroll
    for (int i = 0; i < NSX; ++i) {
      if constexpr (CIN == 1) sx[i] = (wg_f32x4){ursn_buffer_load_f1(r, xgo[i]), 0.f, 0.f, 0.f};
      else sx[i] = ursn_buffer_load_f4(r, xgo[i]);
    }
  };
  auto store_x = [&](int slot) {
#pragma unroll
    for (int i = 0; i < NSX; ++i) {
      int idx = tid + i * 256;
      if (idx < XQ * PS) {
        if (CIN == 1) xr[(size_t)slot * XPLANE + idx] = sx[i][0];
        else {
          wg_f32x4 v = sx[i];
          if (aff && ((sx_inb >> i) & 1u)) { const int q = idx % XQ; v = v * aff_s[q] + aff_t[q]; }   // at the store: loads stay in flight
          *(wg_f32x4*)(xr + (size_t)slot * XPLANE + idx * 4) = v;
        }
      }
    }
  };
  auto load_d = [&](int zin) {
    const bool zok = zin < z1;
    const __amdgpu_buffer_rsrc_t r = ursn_plane_rsrc(dimg + (ptrdiff_t)zin * dplane_f, zok ? dplane_b : 0u);
#pragma unroll
    for (int i = 0; i < NSD; ++i) sd[i] = ursn_buffer_load_f4(r, dgo[i]);
  };
  auto store_d = [&](int slot) {
#pragma unroll
    for (int i = 0; i < NSD; ++i) {
      int idx = tid + i * 256;
      if (idx < DQ * TX * TY) *(wg_f32x4*)(dr + (size_t)slot * DPLANE + idx * 4) = sd[i];
    }
  };

  for (int p = -1; p <= 1; ++p) {
    load_x(z0 + p);
    store_x((z0 + p) & 3);
  }
  load_d(z0);
  store_d(z0 & 1);
  __syncthreads();

  for (int z = z0; z < z1; ++z) {
    load_x(z + 2);
    load_d(z + 1);
    int abase[NA];
#pragma unroll
    for (int m = 0; m < NA; ++m) {
      int tz = a_tz[m] < 0 ? 0 : a_tz[m];
      abase[m] = ((z - 1 + tz) & 3) * XPLANE + a_lane + a_off[m];
    }
    const float* dcur = dr + (size_t)(z & 1) * DPLANE + b_lane;
    wg_static_for<16>([&](auto G) {
      constexpr int g = decltype(G)::value;
      constexpr int grow = (MODE == 3) ? g / 8 : 0;
      constexpr int gcol = (MODE == 3) ? (g % 8) * 4 : g * 4;
      float b[CT];
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        float v = dcur[(grow * TX + gcol) * COUT + c * 16];
        b[c] = (c * 16 + il < a.cout_w) ? v : 0.f;
      }
#pragma unroll
      for (int m = 0; m < NA; ++m) {
        float av = xr[abase[m] + (grow * PX + gcol) * CIN];
        if (a_tz[m] < 0) av = 0.f;
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[m][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[c], acc[m][c], 0, 0, 0);
      }
    });
    store_x((z + 2) & 3);
    store_d((z + 1) & 1);
    __syncthreads();
  }

  // Sum the 4 waves in fixed order -> one slab per workgroup (reproducible).  Every wave stores its accumulators to its
  // own copy in LDS (plain stores, the plane ring is free now; a read-modify-write chain through one copy serialises on
  // LDS latency: ~15 us per workgroup), then all threads add the four copies.
  const int nred = NT * CIN * a.cout_w;
  float* part = wlds + (size_t)wave * nred;
#pragma unroll
  for (int m = 0; m < NA; ++m)
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      int col = c * 16 + il;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int row = 4 * kl + r;
        int tap, ci;
        if (CIN >= 16) { tap = m / MT; ci = (m % MT) * 16 + row; }
        else if (CIN == 1) { tap = 16 * m + row; ci = 0; }
        else { tap = 2 * m + (row >> 3); ci = row & 7; }
        if (tap < NT && col < a.cout_w) part[(tap * CIN + ci) * a.cout_w + col] = acc[m][c][r];
      }
    }
  __syncthreads();
  float* slab = a.slab + (size_t)blockIdx.x * (size_t)nred;
  for (int i = tid; i < nred; i += 256)
    slab[i] = ((wlds[i] + wlds[nred + i]) + wlds[2 * nred + i]) + wlds[3 * nred + i];
}

struct TWPlan {
  int mode, cin, cout;
  int Z, Y, X, zseg, nzseg, nty, ntx;
  size_t lds;
  int grid;
};

template <int CIN, int COUT, int MODE>
static int launch_tw(const TWPlan& p, const TWgradArgs& a, hipStream_t s) {
  auto kern = twgrad_kernel<CIN, COUT, MODE>;
  static size_t attr_lds = 48 * 1024;
  if (p.lds > attr_lds) {
    URSN_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds));
    attr_lds = p.lds;
  }
  hipLaunchKernelGGL(kern, dim3(p.grid), dim3(256), p.lds, s, a);
  URSN_HIP(hipGetLastError());
  return 0;
}

#define URSN_TW(ci, co)                              \
  if (p.cin == ci && p.cout == co) {                 \
    ursn_note_kernel("twgrad<" #ci "," #co ">");     \
    return launch_tw<ci, co, MODE>(p, a, s);         \
  }

int twgrad_dispatch_3d(const TWPlan& p, const TWgradArgs& a, hipStream_t s);
int twgrad_dispatch_2d(const TWPlan& p, const TWgradArgs& a, hipStream_t s);
