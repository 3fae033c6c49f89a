// Weight gradient for Cout = 8 (the full-resolution 8->8 / 16->8 layers) with both halves of the 16-wide MFMA
// column tile doing useful work -- gfx950.
//
// twgrad<*,8> leaves columns 8..15 of every v_mfma_f32_16x16x4_f32 empty (Cout = 8).  Here two OUTPUT planes
// are processed per step: columns 0..7 carry dz of plane z, columns 8..15 dz of plane z+1.  The A operand
// (x at plane z-1+tz4, tz4 = 0..3) is shared, so
//     D[(tz4,ty,tx,ci)][co]     accumulates tap (tz4,   ty, tx) for plane z      (valid tz4 <= 2)
//     D[(tz4,ty,tx,ci)][8 + co] accumulates tap (tz4-1, ty, tx) for plane z + 1  (valid tz4 >= 1)
// i.e. 36 tap rows serve 2 x 27 taps: 18 MFMAs per 4-voxel group and plane pair instead of 2 x 14 (Cin = 8).
// Both column halves cover exactly the workgroup's voxels of their own plane, so image borders need no
// special case.  Ring of 6 x planes (4 live + 2 in flight), 2 dz planes (a wave stages and reads only its OWN 32 voxels of a dz
// plane, so the two incoming planes overwrite the two live ones once the wave's MFMAs of the step are issued -- no barrier
// involved; round 4: with 4 dz planes the ring was 55.5 KB, 0.9 KB too much for three workgroups per CU); the waves are summed in LDS in fixed order,
// two slabs per workgroup (one per column half) go to the deterministic reduce.  Cin = 16 runs as two 8-channel slices.
#pragma once
#include "wgrad_tiled_kernel.h"
#ifndef URSN_SCHED_PIPELINE
#define URSN_SCHED_PIPELINE 1
#endif

// 3-D: 4 waves on a 32 x 4 tile (one row per wave), three workgroups per CU (two with normalise-on-load: 174 VGPRs) so that
// others compute while one sits in its plane barrier (46.5 KB LDS each); 2-D: 8 waves x 32 voxels of a 256-wide row.
template <int MODE> struct ZTile;
template <> struct ZTile<3> { static constexpr int TX = 32, TY = 4, NTY = 3, NT = 27, NW = 4; };
template <> struct ZTile<2> { static constexpr int TX = 256, TY = 1, NTY = 1, NT = 9, NW = 8; };

// AFF: normalise-on-load of x compiled in (a.aff_* set); the plain instantiation carries none of its selects and branches
template <int MODE, bool AFF>
__global__ __launch_bounds__(ZTile<MODE>::NW * 64, (MODE == 3 && !AFF) ? 3 : 2) void twgradz_kernel(TWgradArgs a) {
  constexpr int CIN = 8, COUT = 8;
  using TL = ZTile<MODE>;
  constexpr int NW = TL::NW, NTHR = 64 * NW, NG = 8;            // 32 voxels = 8 groups of 4 per wave and plane
  constexpr int TX = TL::TX, TY = TL::TY, NTY = TL::NTY, NT = TL::NT;
  constexpr int NT4 = 4 * NTY * 3;                              // tap rows incl. the 4th z plane
  constexpr int PX = TX + 2, PY = TY + (NTY == 3 ? 2 : 0), PS = PX * PY;
  constexpr int NA = NT4 / 2;                                  // two taps (8 ci each) per 16-row tile
  constexpr int XQ = CIN / 4, DQ = COUT / 4;
  constexpr int NSX = (XQ * PS + NTHR - 1) / NTHR, NSD = (DQ * TX * TY + NTHR - 1) / NTHR;
  constexpr int XPLANE = PS * CIN, DPLANE = TX * TY * COUT;
  extern __shared__ __attribute__((aligned(16))) float wldz[];  // [6][XPLANE] then [2][DPLANE]
  float* xr = wldz;
  float* dr = wldz + 6 * XPLANE;

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

  const int wrow = (MODE == 3) ? wave : 0;          // 3-D: one 32-wide tile row per wave; 2-D: 32 x per wave
  const int wcol = (MODE == 3) ? 0 : 32 * wave;
  const int a_lane = ((wrow * PX) + wcol + kl) * CIN;
  const int b_lane = ((wrow * TX) + wcol + kl) * COUT + (il & 7);
  const bool hi = il >= 8;
  int a_cm[NA];   // lane constant of tile m: in-plane offset of its tap (lanes 0-7: tap 2m, lanes 8-15: tap 2m+1) + ci
#pragma unroll
  for (int m = 0; m < NA; ++m) {
    int tap = 2 * m + (il >> 3), ci = il & 7;
    a_cm[m] = a_lane + (((tap / 3) % NTY) * PX + (tap % 3)) * CIN + ci;
  }

  // normalise-on-load of x: this thread always stages channel quad (tid & 1)
  __shared__ wg_f32x4 aff_st[2 * XQ];
  constexpr bool aff = AFF;
  wg_f32x4 aff_s4 = {1.f, 1.f, 1.f, 1.f}, aff_t4 = {0.f, 0.f, 0.f, 0.f};
  if constexpr (aff) {
    if (tid < CIN) {
      const float r = a.aff_rstd[tid];
      ((float*)aff_st)[tid] = r;
      ((float*)aff_st)[CIN + tid] = a.aff_beta[tid] - a.aff_mean[tid] * r;
    }
    __syncthreads();
    static_assert(XQ == 2 && (NTHR % XQ) == 0, "channel quad of a staging thread must not depend on the iteration");
    aff_s4 = aff_st[tid & 1];
    aff_t4 = aff_st[XQ + (tid & 1)];
  }
  // staging tables: the tile's (y, x) footprint is the same for every plane, only the plane base moves
  // byte offsets inside a z plane; elements outside the image carry URSN_OOB_OFFSET and read as 0 (wgrad_tiled_kernel.h)
  unsigned xgo[NSX], dgo[NSD], xin = 0;
#pragma unroll
  for (int i = 0; i < NSX; ++i) {
    int idx = tid + i * NTHR;
    int sl = idx / XQ, q = idx - sl * XQ;
    int yy = sl / PX, xx = sl - yy * PX;
    int py = y0 + yy - (NTY == 3 ? 1 : 0), px = x0 + xx - 1;
    const bool ok = idx < XQ * PS && py >= 0 && py < a.Y && px >= 0 && px < a.X;
    xgo[i] = ok ? (unsigned)((py * a.X + px) * a.x_cs + 4 * q) * 4u : URSN_OOB_OFFSET;
    if (ok) xin |= 1u << i;
  }
#pragma unroll
  for (int i = 0; i < NSD; ++i) {
    int idx = tid + i * NTHR;
    int sl = idx / DQ, q = idx - sl * DQ;
    int yy = sl / TX, xx = sl - yy * TX;
    int py = y0 + yy, px = x0 + xx;
    const bool ok = idx < DQ * TX * TY && py < a.Y && px < a.X;
    dgo[i] = ok ? (unsigned)((py * a.X + px) * a.dz_cs + 4 * q) * 4u : URSN_OOB_OFFSET;
  }
  // pin the tables in registers: as plain arithmetic on tid the compiler re-derives them inside the z loop (+200 instructions)
#pragma unroll
  for (int i = 0; i < NSX; ++i) asm volatile("" : "+v"(xgo[i]));
#pragma unroll
  for (int i = 0; i < NSD; ++i) asm volatile("" : "+v"(dgo[i]));
  asm volatile("" : "+v"(xin));
  // normalise-on-load: an element outside the image arrives as 0 and must stay 0 -- its shift is zeroed here, once, so the
  // store applies v * s + t_i without a per-element test (a plane outside the volume skips the affine by a scalar branch)
  wg_f32x4 aff_ti[aff ? NSX : 1];
  if constexpr (aff) {
#pragma unroll
    for (int i = 0; i < NSX; ++i) aff_ti[i] = ((xin >> i) & 1u) ? aff_t4 : (wg_f32x4){0.f, 0.f, 0.f, 0.f};
  }
  // one resource per z plane (num_records = 0 for a plane outside the volume); inside the loop the plane pointers advance by
  // two planes per step instead of being rebuilt from (n, z) with 64-bit multiplies
  const unsigned xplane_bytes = (unsigned)a.Y * a.X * a.x_cs * 4u, dplane_bytes = (unsigned)a.Y * a.X * a.dz_cs * 4u;
  const ptrdiff_t xplane_floats = (ptrdiff_t)a.Y * a.X * a.x_cs, dplane_floats = (ptrdiff_t)a.Y * a.X * a.dz_cs;
  const float* ximg = a.x + (size_t)n * a.Z * a.Y * a.X * a.x_cs;
  const float* dimg = a.dz + (size_t)n * a.Z * a.Y * a.X * a.dz_cs;

  wg_f32x4 acc[NA];
#pragma unroll
  for (int m = 0; m < NA; ++m) acc[m] = (wg_f32x4){0.f, 0.f, 0.f, 0.f};

  // x ring: plane p sits at float offset ((p + 6) % 6) * XPLANE.  The mod is taken once (z0 - 1); inside the loop the four
  // live offsets rotate through scalar registers (add, compare, select) instead of a divide-by-6 chain per plane
  auto xnext = [](int off) { return off + XPLANE == 6 * XPLANE ? 0 : off + XPLANE; };
  auto load_x = [&](int zin, wg_f32x4 (&sx)[NSX], unsigned& inb) {
    const bool zok = zin >= 0 && zin < a.Z;
    inb = zok ? 1u : 0u;
    const __amdgpu_buffer_rsrc_t r = ursn_plane_rsrc(ximg + (ptrdiff_t)zin * xplane_floats, zok ? xplane_bytes : 0u);
#pragma unroll
    for (int i = 0; i < NSX; ++i) sx[i] = ursn_buffer_load_f4(r, xgo[i]);
  };
  auto store_x = [&](int slot_off, const wg_f32x4 (&sx)[NSX], unsigned inb) {
#pragma unroll
    for (int i = 0; i < NSX; ++i) {
      int idx = tid + i * NTHR;
      if (idx < XQ * PS) {
        wg_f32x4 v = sx[i];
        if constexpr (aff) if (inb) v = v * aff_s4 + aff_ti[i];   // normalise-on-load, applied at the store: loads stay in flight (inb: wave-uniform)
        *(wg_f32x4*)(xr + slot_off + idx * 4) = v;
      }
    }
  };

  auto load_d = [&](int zin, wg_f32x4 (&sd)[NSD]) {
    const bool zok = zin < z1;
    const __amdgpu_buffer_rsrc_t r = ursn_plane_rsrc(dimg + (ptrdiff_t)zin * dplane_floats, zok ? dplane_bytes : 0u);
#pragma unroll
    for (int i = 0; i < NSD; ++i) sd[i] = ursn_buffer_load_f4(r, dgo[i]);
  };
  auto store_d = [&](int zin, const wg_f32x4 (&sd)[NSD]) {
#pragma unroll
    for (int i = 0; i < NSD; ++i) {
      int idx = tid + i * NTHR;
      if (idx < DQ * TX * TY) *(wg_f32x4*)(dr + (size_t)(zin & 1) * DPLANE + idx * 4) = sd[i];
    }
  };

  static_assert(NSD == 1 && DQ * 32 == 64, "dz staging is wave-private: thread t stages quad t % 2 of voxel t / 2 = the voxels its own wave reads");
  wg_f32x4 sxa[NSX], sxb[NSX], sda[NSD], sdb[NSD];
  unsigned inba = 0, inbb = 0;
  int sb[4];   // offsets of planes z - 1 .. z + 2
  sb[0] = ((z0 + 5) % 6) * XPLANE;
#pragma unroll
  for (int j = 1; j < 4; ++j) sb[j] = xnext(sb[j - 1]);
#pragma unroll
  for (int p = -1; p <= 2; ++p) {
    load_x(z0 + p, sxa, inba);
    store_x(sb[p + 1], sxa, inba);
  }
  for (int p = 0; p <= 1; ++p) {
    load_d(z0 + p, sda);
    store_d(z0 + p, sda);
  }
  __syncthreads();

  for (int z = z0; z < z1; z += 2) {
    load_x(z + 3, sxa, inba);
    load_x(z + 4, sxb, inbb);
    load_d(z + 2, sda);
    load_d(z + 3, sdb);
    // ring slot bases of the 4 x planes of this step (uniform); tile m's two taps sit in plane (2m)/(3 NTY) and
    // (2m+1)/(3 NTY) -- known at compile time, so a tile costs one add (or select + add where the pair straddles)
    const int sb4 = xnext(sb[3]), sb5 = xnext(sb4);   // where planes z + 3, z + 4 go
    int abase[NA];
    wg_static_for<NA>([&](auto M) {
      constexpr int m = decltype(M)::value;
      constexpr int tzA = (2 * m) / (NTY * 3), tzB = (2 * m + 1) / (NTY * 3);
      if constexpr (tzA == tzB) abase[m] = sb[tzA] + a_cm[m];
      else abase[m] = (hi ? sb[tzB] : sb[tzA]) + a_cm[m];
    });
    const float* dcur = dr + (size_t)((hi ? z + 1 : z) & 1) * DPLANE + b_lane;
    wg_static_for<NG>([&](auto G) {
      constexpr int g = decltype(G)::value;
      constexpr int grow = 0;
      constexpr int gcol = g * 4;
      const float b = dcur[(grow * TX + gcol) * COUT];
#pragma unroll
      for (int m = 0; m < NA; ++m) {
        float av = xr[abase[m] + (grow * PX + gcol) * CIN];
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b, acc[m], 0, 0, 0);
      }
    });
#if URSN_SCHED_PIPELINE
    // software pipelining hint: keep LDS operand reads a few MFMAs ahead instead of read-batch / MFMA-batch phases
    __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
    for (int i = 0; i < NG * NA / 2; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
#endif
    store_x(sb4, sxa, inba);
    store_x(sb5, sxb, inbb);
    store_d(z + 2, sda);
    store_d(z + 3, sdb);
    sb[0] = sb[2]; sb[1] = sb[3]; sb[2] = sb4; sb[3] = sb5;
    __syncthreads();
  }

  // Sum the waves in fixed order -> ONE pair of slabs per workgroup, bitwise reproducible.  Each wave stores its
  // accumulators to its own copy (plain stores into the free plane ring; a read-modify-write chain through one copy
  // serialises on LDS latency), then all threads add the copies.  Half 0: columns 0..7 (plane z, tap tz4); half 1:
  // columns 8..15 (plane z+1, tap tz4-1).
  constexpr int NRED = 2 * NT * CIN * COUT;
  // 3-D: TWO copies (waves 0, 1 store, waves 2, 3 add theirs on top: (w0 + w2) + (w1 + w3)) -- four would need 54 KB, more than
  // the ring; 2-D: one copy per wave (the ring is larger there anyway)
  constexpr int NCOPY = (MODE == 3) ? 2 : NW;
  __syncthreads();   // every wave is done with the rings
  float* part = wldz + (size_t)(wave % NCOPY) * NRED;
  for (int ph = 0; ph < NW / NCOPY; ++ph) {
    if (wave / NCOPY == ph) {
#pragma unroll
      for (int m = 0; m < NA; ++m) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int row = 4 * kl + r;
          int tap4 = 2 * m + (row >> 3), ci = row & 7;
          int tz4 = tap4 / (NTY * 3), rest = tap4 - tz4 * (NTY * 3);
          int tz = hi ? tz4 - 1 : tz4;
          if (tz >= 0 && tz <= 2) {
            float* q = part + (hi ? NT * CIN * COUT : 0) + ((tz * (NTY * 3) + rest) * CIN + ci) * COUT + (il & 7);
            *q = ph ? *q + acc[m][r] : acc[m][r];
          }
        }
      }
    }
    __syncthreads();
  }
  float* slab = a.slab + (size_t)blockIdx.x * (size_t)NRED;
  for (int i = tid; i < NRED; i += NTHR) {
    float v = wldz[i];
#pragma unroll
    for (int w = 1; w < NCOPY; ++w) v += wldz[(size_t)w * NRED + i];
    slab[i] = v;
  }
}

template <int MODE>
static int launch_twz(const TWPlan& p, const TWgradArgs& a, hipStream_t s) {
  auto kern = a.aff_mean ? twgradz_kernel<MODE, true> : twgradz_kernel<MODE, false>;
  static size_t attr_lds[2] = {48 * 1024, 48 * 1024};   // one opt-in per instantiation (plain / normalise-on-load): both run in every training process
  size_t& attr = attr_lds[a.aff_mean ? 1 : 0];
  if (p.lds > attr) {
    URSN_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds));
    attr = p.lds;
  }
  hipLaunchKernelGGL(kern, dim3(p.grid), dim3(ZTile<MODE>::NW * 64), p.lds, s, a);
  URSN_HIP(hipGetLastError());
  return 0;
}

int twgradz_dispatch(const TWPlan& p, const TWgradArgs& a, hipStream_t s);
