// 3-D weight gradient of the full-resolution 8 -> 8 (16 -> 8 as two slices) layers on 4x4 MFMA blocks -- gfx950.
//
// twgradz (wgradz_tiled_kernel.h) feeds v_mfma_f32_16x16x4_f32 with 36 tap rows for 2 x 27 taps: 25 % of the matrix pipe's
// cycles multiply padding, which caps it at 0.75 x pipe-busy (0.50 of the fp32 peak measured).  v_mfma_f32_4x4x1_16b_f32 has
// no padding at 8 channels: each of its 16 blocks is an independent 4 x 4 outer product of ONE voxel,
//     D_b[ci][co] += x[v_b + tap][4q + ci] * dz[v_b][4c + co]
// so one 8-cycle instruction retires 16 voxels x 16 MACs, all useful.  Two things decide whether that pays:
//  * accumulator replication (every (tap, q, c) tile lives once per block, 4 VGPRs): the 27 taps are dealt to the 4 waves
//    of a workgroup (7 + 7 + 7 + 6), 112 accumulator VGPRs per wave held for the whole z column;
//  * LDS INSTRUCTIONS, not bytes: tools/micro/mfma_rate.hip measures ~15 cycles of lost MFMA issue per ds_read of any
//    width on the issuing SIMD, so a ds_read_b32 per two 8-cycle MFMAs (the first version of this kernel, and twgrad4)
//    halves the pipe rate whatever the occupancy.  The planes are therefore kept CHANNEL-MAJOR in LDS, [row][channel][x],
//    and block b stands for the voxel QUAD (row yb = b / 8, x = 4 (b % 8) .. + 3): one aligned ds_read_b128 hands lane
//    (b, i) channel i of four x-consecutive voxels, a second one the next quad, and those 8 registers are the A operands
//    of all 3 x-taps of the 4 voxel "phases" m = 0..3 of the quad (operand = register m + tx).  dz likewise: one b128 per
//    channel quad serves the 4 phases.  Per 64 voxels a wave issues <= 12 + 2 ds_read_b128 for 112 MFMAs.
// Row stride 48 floats: the 16 lanes of a b128 phase (4 quads x 4 channels) fall on 64 distinct banks.
// MEASURED (192^3 x 4, 8 -> 8, same box): 1.24-1.26 ms against twgradz 1.225 ms -- no gain, so this kernel is OFF by default
// (URSN_WGRADQ=1 selects it).  Why: the MFMA loop alone (staging and barriers removed) runs at 0.95-0.99 ms = 100 TFLOP/s,
// i.e. 28 ds_read_b128 per 224 MFMAs still cost ~25 % of the pipe, and the staging (3 global loads, 6 ds_write2_b32 and
// their address / mask code per thread and plane) another 0.2 ms.  The first version (ds_read2st64_b32 operands in the
// [quad][voxel] layout, 80 LDS instructions per 224 MFMAs) measured 1.26 ms with the pipe 61 % busy; this one 55-65 %.
// Priority by wave slot (URSN_WGRADQ_PRIO) to de-phase the two workgroups of a CU changed nothing.
// z-march: ring of 4 x planes (3 live + 1 in flight through registers), 2 dz planes; the staging transposes voxel-major
// global float4 into the channel-major rows with 2 ds_write2_b32 each.  The 16 blocks are summed with xor-shuffles once
// per workgroup; each wave writes its own taps of the workgroup's slab (every element exactly once: the two-stage reduce
// stays bitwise reproducible).
#pragma once
#include "wgrad_tiled_kernel.h"

struct QTile {
  static constexpr int TX = 32, TY = 4, NW = 4, NT = 27, TPW = 7;
  static constexpr int RS = 48;                                  // floats per (row, channel) of a plane
  static constexpr int XPLANE = (TY + 2) * 8 * RS, DPLANE = TY * 8 * RS;   // floats
  static constexpr size_t LDS = (size_t)(4 * XPLANE + 2 * DPLANE) * sizeof(float);
};

// One z plane of one wave: W = wave index (compile time, so that the operand register of a tap is a constant).
template <int W>
__device__ __forceinline__ void twq_plane(wg_f32x4 (&acc)[7][2][2], const float* __restrict__ xr, const float* __restrict__ dcur,
                                          const int (&xb)[3], int lane_off) {
  constexpr int RS = QTile::RS, TPW = QTile::TPW, NT = QTile::NT;
  constexpr int T0 = W * TPW;
  constexpr int ROW0 = T0 / 3;                                   // first (tz, ty) tap row this wave touches
  constexpr int TL = (T0 + TPW - 1 < NT ? T0 + TPW - 1 : NT - 1);
  constexpr int NR = TL / 3 - ROW0 + 1;                          // tap rows touched (<= 3)
  // Software pipeline over the 2 NR (super-group, tap row) steps of the plane: the operands of step k + 1 are requested
  // before the MFMAs of step k; a scheduling barrier per step keeps the compiler from hoisting every read of the plane to
  // the top (it does: 56 operand registers on top of the 112 accumulators, 88 spills).
  constexpr int NS = 2 * NR;
  wg_f32x4 ra[2][2][2], rb[2][2];                                // [step parity][q][quad], [super-group][c]
  auto load_step = [&](auto K) {
    constexpr int k = decltype(K)::value;
    constexpr int sg = k / NR, r = k % NR;                       // super-group: tile rows 2 sg, 2 sg + 1
    constexpr int row = ROW0 + r, tz = row / 3, ty = row % 3;
    if constexpr (r == 0) {
#pragma unroll
      for (int c = 0; c < 2; ++c) rb[sg][c] = *(const wg_f32x4*)(dcur + ((2 * sg) * 8 + 4 * c) * RS);
    }
    const float* p = xr + xb[tz] + lane_off + ((2 * sg + ty) * 8) * RS;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      ra[k & 1][q][0] = *(const wg_f32x4*)(p + (4 * q) * RS);
      ra[k & 1][q][1] = *(const wg_f32x4*)(p + (4 * q) * RS + 4);
    }
  };
  load_step(std::integral_constant<int, 0>{});
  wg_static_for<NS>([&](auto K) {
    constexpr int k = decltype(K)::value;
    constexpr int sg = k / NR, r = k % NR;
    if constexpr (k + 1 < NS) load_step(std::integral_constant<int, k + 1>{});
    wg_static_for<TPW>([&](auto TI) {
      constexpr int ti = decltype(TI)::value;
      constexpr int t = (T0 + ti < NT) ? T0 + ti : NT - 1;       // wave 3: the 7th slot recomputes tap 26, never written out
      constexpr int tx = t % 3;
      if constexpr (t / 3 - ROW0 == r) {
        wg_static_for<4>([&](auto M) {
          constexpr int m = decltype(M)::value;
          constexpr int n = m + tx;                              // 0..5: element n of the two quads
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const float av = ra[k & 1][q][n / 4][n % 4];
#pragma unroll
            for (int c = 0; c < 2; ++c)
              acc[ti][q][c] = __builtin_amdgcn_mfma_f32_4x4x1f32(av, rb[sg][c][m], acc[ti][q][c], 0, 0, 0);
          }
        });
      }
    });
    __builtin_amdgcn_sched_barrier(0);
  });
}

// The whole z column of wave W.  One instantiation per wave, selected by a SCALAR branch in the kernel: four variants of the
// plane sharing one accumulator array through phis cost 200 spills, and the barriers below need wave-uniform control flow.
template <int W>
__device__ __forceinline__ void twq_body(const TWgradArgs& a, float* wldq, const wg_f32x4* aff_st) {
  constexpr int TX = QTile::TX, TY = QTile::TY, NW = QTile::NW, NTHR = 64 * NW, NT = QTile::NT, TPW = QTile::TPW;
  constexpr int CIN = 8, COUT = 8, NQ = 2, CQ = 2, RS = QTile::RS;
  constexpr int PX = TX + 2, PY = TY + 2, PS = PX * PY;      // 34 x 6 halo plane
  constexpr int TS = TX * TY;
  constexpr int XPLANE = QTile::XPLANE, DPLANE = QTile::DPLANE;
  constexpr int NSX = (NQ * PS + NTHR - 1) / NTHR;
  static_assert(CQ * TS == NTHR, "one dz float4 per thread and plane");
  float* xr = wldq;                                             // [4][XPLANE] then [2][DPLANE]
  float* dr = wldq + 4 * XPLANE;

  const int tid = threadIdx.x, lane = tid & 63;
  constexpr int wave = W;
  int bid = ursn_xcd_block(blockIdx.x, gridDim.x);
  const int xt = bid % a.ntx; bid /= a.ntx;
  const int yt = bid % a.nty; bid /= a.nty;
  const int zs = bid % a.nzseg;
  const int n = bid / a.nzseg;
  const int x0 = xt * TX, y0 = yt * TY;
  const int z0 = zs * a.zseg;
  const int z1 = (z0 + a.zseg < a.Z) ? z0 + a.zseg : a.Z;

  // lane (b, i): block b = voxel quad (row yb, x quad xq), i = channel within the quad of channels
  const int bl = lane >> 2, il = lane & 3;
  const int lane_off = ((bl >> 3) * 8 + il) * RS + 4 * (bl & 7);

  const bool aff = a.aff_mean != nullptr;   // normalise-on-load of x: aff_st = scale / shift per channel quad

  // staging tables: the tile's (y, x) footprint is the same for every plane
  int xgo[NSX], xlo[NSX], dgo, dlo;
  bool xok[NSX], dok;
#pragma unroll
  for (int i = 0; i < NSX; ++i) {
    const int idx = tid + i * NTHR;
    const int q = idx / PS, sl = idx - q * PS;
    const int yy = sl / PX, xx = sl - yy * PX;
    const int py = y0 + yy - 1, px = x0 + xx - 1;
    xok[i] = idx < NQ * PS && py >= 0 && py < a.Y && px >= 0 && px < a.X;
    xgo[i] = (py * a.X + px) * a.x_cs + 4 * q;
    xlo[i] = idx < NQ * PS ? (yy * 8 + 4 * q) * RS + xx : -1;
  }
  {
    const int q = tid / TS, sl = tid - q * TS;
    const int yy = sl / TX, xx = sl - yy * TX;
    const int py = y0 + yy, px = x0 + xx;
    dok = py < a.Y && px < a.X;
    dgo = (py * a.X + px) * a.dz_cs + 4 * q;
    dlo = (yy * 8 + 4 * q) * RS + xx;
  }

  wg_f32x4 acc[TPW][NQ][CQ];
#pragma unroll
  for (int i = 0; i < TPW; ++i)
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int c = 0; c < CQ; ++c) acc[i][q][c] = (wg_f32x4){0.f, 0.f, 0.f, 0.f};

  wg_f32x4 sx[NSX], sd;
  unsigned inb = 0;
  auto load_x = [&](int zin) {
    const bool zok = zin >= 0 && zin < a.Z;
    inb = 0;
    const float* base = a.x + ((size_t)n * a.Z + (zok ? zin : 0)) * a.Y * a.X * a.x_cs;
#pragma unroll
    for (int i = 0; i < NSX; ++i) {
      wg_f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (zok && xok[i]) {
        v = *(const wg_f32x4*)(base + xgo[i]);
        inb |= 1u << i;
      }
      sx[i] = v;
    }
  };
  auto store_x = [&](int zin) {   // voxel-major float4 -> channel-major rows
#pragma unroll
    for (int i = 0; i < NSX; ++i) {
      if (xlo[i] >= 0) {
        wg_f32x4 v = sx[i];
        if (aff && ((inb >> i) & 1u)) {   // applied at the store: the loads stay in flight over the MFMA block
          const int q = (tid + i * NTHR) / PS;
          v = v * aff_st[q] + aff_st[NQ + q];
        }
        float* p = xr + (size_t)(zin & 3) * XPLANE + xlo[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) p[j * RS] = v[j];
      }
    }
  };
  auto load_d = [&](int zin) {
    wg_f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (zin < z1 && dok) v = *(const wg_f32x4*)(a.dz + ((size_t)n * a.Z + zin) * a.Y * a.X * a.dz_cs + dgo);
    sd = v;
  };
  auto store_d = [&](int zin) {
    float* p = dr + (size_t)(zin & 1) * DPLANE + dlo;
#pragma unroll
    for (int j = 0; j < 4; ++j) p[j * RS] = sd[j];
  };

  // columns 34..47 of a row are padding: the second quad of x quad 7 loads columns 32..35, operands reach column 33 only
  for (int p = -1; p <= 1; ++p) {
    load_x(z0 + p);
    store_x(z0 + p);
  }
  load_d(z0);
  store_d(z0);
  __syncthreads();

  for (int z = z0; z < z1; ++z) {
    load_x(z + 2);
    load_d(z + 1);
    int xb[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) xb[j] = ((z - 1 + j) & 3) * XPLANE;
    const float* dcur = dr + (size_t)(z & 1) * DPLANE + lane_off;
    twq_plane<W>(acc, xr, dcur, xb, lane_off);
    store_x(z + 2);
    store_d(z + 1);
    __syncthreads();
  }

  // sum the 16 blocks (lanes l, l + 4, ..., l + 60); lanes 0..3 write this wave's taps of the workgroup slab
  float* slab = a.slab + (size_t)blockIdx.x * (size_t)(NT * CIN * COUT);
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
        if (lane < 4 && tap < NT) {
#pragma unroll
          for (int r = 0; r < 4; ++r) slab[((size_t)tap * CIN + 4 * q + r) * COUT + 4 * c + il] = v[r];
        }
      }
}

__global__ __launch_bounds__(256, 2) void twgradq_kernel(TWgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) float wldq[];
  // normalise-on-load of x (x = raw z of the preceding conv): staged value = z * rstd + (beta - mean * rstd)
  __shared__ wg_f32x4 aff_st[4];
  if (a.aff_mean != nullptr) {
    if (threadIdx.x < 8) {
      const float r = a.aff_rstd[threadIdx.x];
      ((float*)aff_st)[threadIdx.x] = r;
      ((float*)aff_st)[8 + threadIdx.x] = a.aff_beta[threadIdx.x] - a.aff_mean[threadIdx.x] * r;
    }
    __syncthreads();
  }
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // SGPR: the branch below is scalar
#if URSN_WGRADQ_PRIO
  // Two workgroups share a CU, one wave of each per SIMD.  With equal priority the arbiter alternates their MFMAs, both
  // advance in lockstep and reach their plane barrier / staging bubble TOGETHER (matrix pipe 55 % busy).  Priority by wave
  // slot (HW_ID.wave_id, the same on all four SIMDs for the waves of one workgroup) lets one stream its plane while the
  // other takes what is left, so the bubbles of one fall into the MFMA phase of the other.
  if (__builtin_amdgcn_s_getreg((3 << 11) | 4) & 1) __builtin_amdgcn_s_setprio(1);
#endif
  if (wave == 0) twq_body<0>(a, wldq, aff_st);
  else if (wave == 1) twq_body<1>(a, wldq, aff_st);
  else if (wave == 2) twq_body<2>(a, wldq, aff_st);
  else twq_body<3>(a, wldq, aff_st);
}

static int launch_twq(const TWPlan& p, const TWgradArgs& a, hipStream_t s) {
  static bool attr = false;
  if (!attr) {
    URSN_HIP(hipFuncSetAttribute((const void*)twgradq_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)QTile::LDS));
    attr = true;
  }
  URSN_REQUIRE(p.lds == QTile::LDS, "twgradq: plan / kernel LDS mismatch");
  hipLaunchKernelGGL(twgradq_kernel, dim3(p.grid), dim3(QTile::NW * 64), p.lds, s, a);
  URSN_HIP(hipGetLastError());
  return 0;
}
