// LDS-tiled weight gradient for layers with Cout <= 8 (the full-resolution level: 8->8, 16->8, 8->3) -- gfx950.
//
// With 8 output channels a 16-wide MFMA column tile is half padding (twgrad<*,8> runs at ~40 TFLOP/s).
// v_mfma_f32_4x4x1_16b_f32 has no such waste: each of its 16 blocks is an independent 4x4 outer product
//     D_b[ci][co] += x[v_b + tap][4q + ci] * dz[v_b][4c + co]              (one voxel v_b per block)
// so one instruction retires 16 voxels x 16 MACs at the full fp32 rate.  The price is accumulator
// replication (every (tap, ci-quad, co-quad) tile lives 16x across the blocks), so the 27 taps are split over
// the 8 waves of a workgroup (4+4+4+4+4+4+3+0) and every wave sweeps all 256 voxels of the plane tile for its
// taps: 8 + 2 ds_read_b32 per 16 MFMAs (conflict-free in the [quad][voxel] float4 layout); two waves per SIMD
// keep the matrix pipe fed while the partner waits on LDS.
// Blocks are summed with 4 xor-shuffles per value once per workgroup; one slab per workgroup.
#pragma once
#include "wgrad_tiled_kernel.h"

template <int CIN, int COUT, int MODE>
__global__ __launch_bounds__(512, 2) void twgrad4_kernel(TWgradArgs a) {
  constexpr int NW = 8, NTHR = NW * 64;                  // 8 waves: 2 per SIMD hide the LDS latency of the 8-cycle MFMAs
  using TL = WTile<MODE>;
  constexpr int TX = TL::TX, TY = TL::TY, NTY = TL::NTY, NT = TL::NT;
  constexpr int PX = TX + 2, PY = TY + (NTY == 3 ? 2 : 0), PS = PX * PY, TS = TX * TY;
  constexpr int NQ = CIN / 4, CQ = COUT / 4;
  constexpr int TPW = (NT + NW - 1) / NW;                // taps per wave
  constexpr int NSX = (NQ * PS + NTHR - 1) / NTHR, NSD = (CQ * TS + NTHR - 1) / NTHR;
  constexpr int XPLANE = NQ * PS * 4, DPLANE = CQ * TS * 4;  // floats
  extern __shared__ __attribute__((aligned(16))) float wlds4[];  // [4][XPLANE] then [2][DPLANE]
  float* xr = wlds4;
  float* dr = wlds4 + 4 * XPLANE;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bl = lane >> 2, il = lane & 3;
  int bid = ursn_xcd_block(blockIdx.x, gridDim.x);
  const int xt = bid % a.ntx; bid /= a.ntx;
  const int yt = bid % a.nty; bid /= a.nty;
  const int zs = bid % a.nzseg;
  const int n = bid / a.nzseg;
  const int x0 = xt * TX, y0 = yt * TY;
  const int z0 = zs * a.zseg;
  const int z1 = (z0 + a.zseg < a.Z) ? z0 + a.zseg : a.Z;

  // this wave's taps (wave-uniform): plane selector and float offset of the tap shift inside a plane
  int t_tz[TPW], t_off[TPW];
  bool t_ok[TPW];
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    int t = wave * TPW + i;
    t_ok[i] = t < NT;
    if (!t_ok[i]) t = 0;
    t_tz[i] = t / (NTY * 3);
    t_off[i] = (((t / 3) % NTY) * PX + (t % 3)) * 4;
  }
  const int a_lane = bl * 4 + il;  // + (q*PS + row*PX + x)*4 per group
  const int b_lane = bl * 4 + il;

  wg_f32x4 acc[TPW][NQ][CQ];
#pragma unroll
  for (int i = 0; i < TPW; ++i)
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int c = 0; c < CQ; ++c) acc[i][q][c] = (wg_f32x4){0.f, 0.f, 0.f, 0.f};

  wg_f32x4 sx[NSX], sd[NSD];
  auto load_x = [&](int zin) {
#pragma unroll
    for (int i = 0; i < NSX; ++i) {
      int idx = tid + i * NTHR;
      wg_f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (idx < NQ * PS) {
        int q = idx / PS, s = idx - q * PS;
        int yy = s / PX, xx = s - yy * PX;
        int py = y0 + yy - (NTY == 3 ? 1 : 0), px = x0 + xx - 1;
        if (zin >= 0 && zin < a.Z && py >= 0 && py < a.Y && px >= 0 && px < a.X)
          v = *(const wg_f32x4*)(a.x + ((((size_t)n * a.Z + zin) * a.Y + py) * a.X + px) * a.x_cs + 4 * q);
      }
      sx[i] = v;
    }
  };
  auto store_x = [&](int slot) {
#pragma unroll
    for (int i = 0; i < NSX; ++i) {
      int idx = tid + i * NTHR;
      if (idx < NQ * PS) *(wg_f32x4*)(xr + (size_t)slot * XPLANE + idx * 4) = sx[i];
    }
  };
  auto load_d = [&](int zin) {
#pragma unroll
    for (int i = 0; i < NSD; ++i) {
      int idx = tid + i * NTHR;
      wg_f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (idx < CQ * TS) {
        int q = idx / TS, s = idx - q * TS;
        int yy = s / TX, xx = s - yy * TX;
        int py = y0 + yy, px = x0 + xx;
        if (zin < z1 && py < a.Y && px < a.X) {
          const float* src = a.dz + ((((size_t)n * a.Z + zin) * a.Y + py) * a.X + px) * a.dz_cs + 4 * q;
          v = *(const wg_f32x4*)src;
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (4 * q + j >= a.cout_w) v[j] = 0.f;   // padded logits channels carry no gradient
        }
      }
      sd[i] = v;
    }
  };
  auto store_d = [&](int slot) {
#pragma unroll
    for (int i = 0; i < NSD; ++i) {
      int idx = tid + i * NTHR;
      if (idx < CQ * TS) *(wg_f32x4*)(dr + (size_t)slot * DPLANE + idx * 4) = sd[i];
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
    int abase[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) abase[i] = ((z - 1 + t_tz[i]) & 3) * XPLANE + t_off[i] + a_lane;
    const float* dcur = dr + (size_t)(z & 1) * DPLANE + b_lane;
    wg_static_for<16>([&](auto G) {
      constexpr int g = decltype(G)::value;
      // 16 consecutive x per group: 3-D tile rows hold 2 groups, the 2-D strip 16
      constexpr int grow = (MODE == 3) ? g / 2 : 0;
      constexpr int gcol = (MODE == 3) ? (g % 2) * 16 : g * 16;
      float b[CQ];
#pragma unroll
      for (int c = 0; c < CQ; ++c) b[c] = dcur[(c * TS + grow * TX + gcol) * 4];
#pragma unroll
      for (int i = 0; i < TPW; ++i) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          float av = xr[abase[i] + (q * PS + grow * PX + gcol) * 4];
          if (!t_ok[i]) av = 0.f;
#pragma unroll
          for (int c = 0; c < CQ; ++c)
            acc[i][q][c] = __builtin_amdgcn_mfma_f32_4x4x1f32(av, b[c], acc[i][q][c], 0, 0, 0);
        }
      }
    });
    store_x((z + 2) & 3);
    store_d((z + 1) & 1);
    __syncthreads();
  }

  // sum the 16 blocks (lanes l, l+4, ..., l+60), then lanes 0..3 write this wave's taps of the workgroup slab
  float* slab = a.slab + (size_t)blockIdx.x * (size_t)(NT * CIN * a.cout_w);
#pragma unroll
  for (int i = 0; i < TPW; ++i)
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int c = 0; c < CQ; ++c) {
        wg_f32x4 v = acc[i][q][c];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float u = v[r];
          u += __shfl_xor(u, 4);
          u += __shfl_xor(u, 8);
          u += __shfl_xor(u, 16);
          u += __shfl_xor(u, 32);
          v[r] = u;
        }
        const int tap = wave * TPW + i;
        const int co = 4 * c + il;
        if (bl == 0 && tap < NT && co < a.cout_w) {
#pragma unroll
          for (int r = 0; r < 4; ++r) slab[((size_t)tap * CIN + 4 * q + r) * a.cout_w + co] = v[r];
        }
      }
}

template <int CIN, int COUT, int MODE>
static int launch_tw4(const TWPlan& p, const TWgradArgs& a, hipStream_t s) {
  auto kern = twgrad4_kernel<CIN, COUT, MODE>;
  static size_t attr_lds = 48 * 1024;
  if (p.lds > attr_lds) {
    URSN_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds));
    attr_lds = p.lds;
  }
  hipLaunchKernelGGL(kern, dim3(p.grid), dim3(512), p.lds, s, a);
  URSN_HIP(hipGetLastError());
  return 0;
}

#define URSN_TW4(ci, co)                              \
  if (p.cin == ci && p.cout == co) {                  \
    ursn_note_kernel("twgrad4<" #ci "," #co ">");     \
    return launch_tw4<ci, co, MODE>(p, a, s);         \
  }

int twgrad4_dispatch_3d(const TWPlan& p, const TWgradArgs& a, hipStream_t s);
int twgrad4_dispatch_2d(const TWPlan& p, const TWgradArgs& a, hipStream_t s);
