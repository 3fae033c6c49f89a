// bf16 weight gradient of the 3x3x3 stride-1 layers of the 8/16-channel levels (lib/uresnet.py:31-36, :84-100,
// lib/resnet_module.py:43-51 under tf.gradients): dW[t][ci][co] = sum_v x[v + d_t][ci] * dz[v][co], fp32 accumulation on
// v_mfma_f32_16x16x32_bf16, z-MARCHING.
//
// The box kernel of bf16_conv.hip stages a (2+2) x (8+2) x (32+2) halo box of x per 2 x 8 x 32 voxels of dz: 2.7 x the
// bytes of x, which alone put the 256^3 level at 0.8 ms of its 1.3 (HBM time of the layer: 0.27 ms).  Here a workgroup walks
// a 32 x TY voxel column through z with a ring of four x planes (three in use, one arriving) and two dz planes in LDS, so
// x is read 1.2 x and dz once, and the 27 tap products stay in the accumulators for the whole column: the fp32 slab is
// written once per workgroup.  MFMA view (as the box kernel): rows = (tap, ci) 16 per tile -- two taps x 8 channels or one
// tap x 16 -- columns = co, contraction = 32 voxels along x, both operands read with ds_read_b64_tr_b16; the tiles are
// dealt round-robin to the four waves (every wave sees every voxel, a tile lives in exactly one wave: no cross-wave sum).
#include <stdlib.h>

#include "bf16_common.h"
#include "buffer_stage.h"

namespace {

template <int K, int NN>
struct W3 {
  static constexpr int CPV = K / 8, DPV = NN / 8;
  static constexpr int TY = K == 8 ? 16 : 8;
  static constexpr int PX = 34, PY = TY + 2;
  static constexpr int XPIECES = PX * PY * CPV, XPLANE = XPIECES * 16;
  static constexpr int DPIECES = 32 * TY * DPV, DPLANE = DPIECES * 16;
  static constexpr int NXS = (XPIECES + 255) / 256, NDS = (DPIECES + 255) / 256;
  static constexpr int TPD = K == 8 ? 5 : 9;          // 16-row tiles per tap plane (K = 8: 4 tap pairs + 1 single)
  static constexpr int NT = 3 * TPD;                  // 15 | 27
  static constexpr int MTW = (NT + 3) / 4;            // tiles per wave
  static constexpr int LDS = 4 * XPLANE + 2 * DPLANE;
};

struct W3Args {
  const bf16_t* S;    // x: (N, Z, Y, X, in_cs)
  const bf16_t* C;    // dz: (N, Z, Y, X, out_cs)
  float* slab;        // [grid][NT][16][16] fp32
  int N, Z, Y, X, in_cs, out_cs;
  int zseg, nzseg, nty, ntx;
  const float* aff_mean; const float* aff_rstd; const float* aff_beta;   // AFF: S is a raw conv output, normalised while staged
  int aff_relu;
  const float* S_f32;   // pair kernel only: S is one fp32 channel per voxel (the network input), staged as (bf16(value), 0 x 7)
};

__device__ __forceinline__ bfx8 w3_tr_pair(const unsigned char* p0, const unsigned char* p1) {
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)p0);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)p1);
  s16x8 v;
  v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
  v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
  return __builtin_bit_cast(bfx8, v);
}

template <int K, int NN, bool AFF = false>
__global__ __launch_bounds__(256, 2) void b3wgrad_kernel(W3Args a) {
  using G = W3<K, NN>;
  constexpr int CPV = G::CPV, DPV = G::DPV, PX = G::PX, TY = G::TY, MTW = G::MTW, TPD = G::TPD;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* xring = smem;
  unsigned char* dbuf = smem + 4 * G::XPLANE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Gk = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;   // transposed read: lane supplies (k row tq, column chunk tp)
  int bid = blockIdx.x;
  const int nblk = gridDim.x;
  if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);
  const int tx = bid % a.ntx;
  int r_ = bid / a.ntx;
  const int ty = r_ % a.nty;
  r_ /= a.nty;
  const int zs = r_ % a.nzseg, n = r_ / a.nzseg;
  const int x0 = tx * 32, y0 = ty * TY, z0 = zs * a.zseg;
  const int z1 = z0 + a.zseg < a.Z ? z0 + a.zseg : a.Z;

  // staging geometry (fixed over z)
  // byte offsets inside a z plane; a piece outside the image carries URSN_OOB_BYTES and reads as zeros through the buffer
  // bounds check (buffer_stage.h): no zero fill, no branch, no 64-bit address per piece
  unsigned xrel[G::NXS], drel[G::NDS];
  unsigned xval = 0;
#pragma unroll
  for (int i = 0; i < G::NXS; ++i) {
    const int idx = tid + 256 * i;
    xrel[i] = URSN_OOB_BYTES;
    if (idx < G::XPIECES) {
      const int vi = idx / CPV, hp = idx - vi * CPV;
      const int yy = vi / PX, xx = vi - yy * PX;
      const int gy = y0 + yy - 1, gx = x0 + xx - 1;
      if (gy >= 0 && gy < a.Y && gx >= 0 && gx < a.X) { xval |= 1u << i; xrel[i] = (unsigned)((gy * a.X + gx) * a.in_cs + hp * 8) * 2u; }
    }
    asm volatile("" : "+v"(xrel[i]));
  }
#pragma unroll
  for (int i = 0; i < G::NDS; ++i) {
    const int idx = tid + 256 * i;
    drel[i] = URSN_OOB_BYTES;
    if (idx < G::DPIECES) {
      const int vi = idx / DPV, hp = idx - vi * DPV;
      const int yy = vi >> 5, xx = vi & 31;
      const int gy = y0 + yy, gx = x0 + xx;
      if (gy < a.Y && gx < a.X) drel[i] = (unsigned)((gy * a.X + gx) * a.out_cs + hp * 8) * 2u;
    }
    asm volatile("" : "+v"(drel[i]));
  }
  const size_t x_plane = (size_t)a.Y * a.X * a.in_cs, d_plane = (size_t)a.Y * a.X * a.out_cs;
  const bf16_t* x_img = a.S + (size_t)n * a.Z * x_plane;
  const bf16_t* d_img = a.C + (size_t)n * a.Z * d_plane;
  u32x4 xs[G::NXS], ds[G::NDS];
  unsigned xin = 0;
  float asc[8], ash[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    asc[j] = 1.f; ash[j] = 0.f;
    if constexpr (AFF) {
      const int ch = (K == 16 ? (tid & 1) * 8 : 0) + j;
      asc[j] = a.aff_rstd[ch];
      ash[j] = fmaf(-a.aff_mean[ch], asc[j], a.aff_beta[ch]);   // as b3conv: bf16(fma(z, r, fma(-mu, r, beta)))
    }
  }
  auto load_x = [&](int p) {
    const bool pz = p >= 0 && p < a.Z;
    const __amdgpu_buffer_rsrc_t r = ursn_rsrc(x_img + (ptrdiff_t)p * (ptrdiff_t)x_plane, pz ? (unsigned)x_plane * 2u : 0u);
#pragma unroll
    for (int i = 0; i < G::NXS; ++i) xs[i] = ursn_bload_b128(r, xrel[i]);
    if constexpr (AFF) xin = pz ? xval : 0u;
  };
  auto store_x = [&](int p) {   // plane p lives in ring slot (p + 1) & 3
    unsigned char* dst = xring + ((p + 1) & 3) * G::XPLANE;
#pragma unroll
    for (int i = 0; i < G::NXS; ++i) {
      const int idx = tid + 256 * i;
      if constexpr (AFF) {
        if ((xin >> i) & 1u) {
          float f[8];
          unpack8(xs[i], f);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            f[j] = fmaf(f[j], asc[j], ash[j]);
            if (a.aff_relu) f[j] = fmaxf(f[j], 0.f);
          }
          xs[i] = pack8(f);
        }
      }
      if (idx < G::XPIECES) *(u32x4*)(dst + idx * 16) = xs[i];
    }
  };
  auto load_d = [&](int q) {
    const bool qz = q < z1;
    const __amdgpu_buffer_rsrc_t r = ursn_rsrc(d_img + (ptrdiff_t)q * (ptrdiff_t)d_plane, qz ? (unsigned)d_plane * 2u : 0u);
#pragma unroll
    for (int i = 0; i < G::NDS; ++i) ds[i] = ursn_bload_b128(r, drel[i]);
  };
  auto store_d = [&](int q) {
    unsigned char* dst = dbuf + (q & 1) * G::DPLANE;
#pragma unroll
    for (int i = 0; i < G::NDS; ++i) {
      const int idx = tid + 256 * i;
      if (idx < G::DPIECES) *(u32x4*)(dst + idx * 16) = ds[i];
    }
  };

  // this wave's tiles: u = wave + 4 m  ->  tap plane dzm, in-plane tap(s); aoff: byte offset of the lane's A piece in a plane
  int dzm[MTW];
  unsigned aoff[MTW];
#pragma unroll
  for (int m = 0; m < MTW; ++m) {
    const int u = wave + 4 * m;
    const int uu = u < G::NT ? u : G::NT - 1;
    dzm[m] = uu / TPD;
    const int i = uu - dzm[m] * TPD;
    int j = K == 8 ? 2 * i + (tp >> 1) : i;   // in-plane tap of this lane's column chunk
    if (j > 8) j = 8;                         // the unpaired slot of a tap plane: duplicate, dropped by the reduce
    const int chunk = K == 8 ? (tp & 1) : tp; // 4-channel chunk of the voxel
    aoff[m] = (unsigned)((((j / 3) * PX + (j % 3) + 8 * Gk + tq) * CPV) * 16 + chunk * 8);
  }
  const unsigned boff = (unsigned)(((8 * Gk + tq) * DPV) * 16 + (NN == 8 ? (tp & 1) : tp) * 8);

  bf_f32x4 acc[MTW];
#pragma unroll
  for (int m = 0; m < MTW; ++m) acc[m] = (bf_f32x4){0.f, 0.f, 0.f, 0.f};

  // prologue: x planes z0-1, z0, z0+1 and dz plane z0
  for (int p = z0 - 1; p <= z0 + 1; ++p) { load_x(p); store_x(p); }
  load_d(z0);
  store_d(z0);
  __syncthreads();
  for (int q = z0; q < z1; ++q) {
    load_x(q + 2);
    load_d(q + 1);
    const unsigned char* db = dbuf + (q & 1) * G::DPLANE;
    unsigned sb[MTW];
#pragma unroll
    for (int m = 0; m < MTW; ++m) sb[m] = (unsigned)(((q + dzm[m]) & 3) * G::XPLANE) + aoff[m];   // plane q - 1 + dz -> slot (q + dz) & 3
#pragma unroll   // all rows: every LDS offset an immediate
    for (int r = 0; r < TY; ++r) {
      const unsigned char* bp = db + boff + r * (32 * DPV * 16);
      const bfx8 B = w3_tr_pair(bp, bp + 4 * DPV * 16);
#pragma unroll
      for (int m = 0; m < MTW; ++m) {
        const unsigned char* ap = xring + sb[m] + r * (PX * CPV * 16);
        const bfx8 A = w3_tr_pair(ap, ap + 4 * CPV * 16);
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, acc[m], 0, 0, 0);
      }
    }
    store_x(q + 2);
    store_d(q + 1);
    __syncthreads();
  }
  float* sl = a.slab + (size_t)blockIdx.x * (G::NT * 256);
#pragma unroll
  for (int m = 0; m < MTW; ++m) {
    const int u = wave + 4 * m;
    if (u >= G::NT) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) sl[(size_t)u * 256 + (4 * Gk + r) * 16 + li] = acc[m][r];
  }
}

struct W3ReduceArgs {
  const float* slab; float* dw;
  int nslabs, Kw, Nw;
  int tapw[27];   // stored tap index by (dz * 9 + dy * 3 + dx), -1 = none
  int stride;     // in slabs: > 1 after slab_fold_kernel summed groups of `stride` slabs into the first slab of each group
};
// Stage 1 of the slab sum, in place: slab[g * group] += slab[g * group + 1 .. (g + 1) * group - 1], element-wise in that fixed
// order, on a grid that fills the chip.  The per-tile reduce kernels below (15 .. 27 workgroups) then walk nslabs / group
// slabs instead of 1024: walking all of them took 0.11 ms per layer of pure load latency (rocprofv3, cfg5).
__global__ __launch_bounds__(256) void slab_fold_kernel(float* __restrict__ slab, int nslabs, size_t per4, int group) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= per4) return;
  const int k0 = blockIdx.y * group, k1 = k0 + group < nslabs ? k0 + group : nslabs;
  bf_f32x4* base = (bf_f32x4*)slab;
  bf_f32x4 acc = base[(size_t)k0 * per4 + i];
  for (int k = k0 + 1; k < k1; ++k) acc += base[(size_t)k * per4 + i];
  base[(size_t)k0 * per4 + i] = acc;
}
static int slab_fold(float* slab, int& nslabs, int& stride, size_t per, hipStream_t s) {
  stride = 1;
  static const int group = getenv("URSN_SLAB_FOLD") ? atoi(getenv("URSN_SLAB_FOLD")) : 16;   // A/B: 0 | 1 = off
  if (group < 2 || nslabs < 4 * group || (per & 3)) return 0;
  const int ng = (nslabs + group - 1) / group;
  hipLaunchKernelGGL(slab_fold_kernel, dim3((unsigned)((per / 4 + 255) / 256), ng), dim3(256), 0, s, slab, nslabs, per / 4, group);
  URSN_HIP(hipGetLastError());
  nslabs = ng; stride = group;
  return 0;
}
// dw[tap][ci][co] += sum over workgroup slabs in a fixed order: a block owns one 16 x 16 tile, its four y-slices each sum a
// quarter of the slabs (one thread per element walking 1024 slabs was 0.11 ms of pure load latency per layer), then add up
// in slice order
template <int K, int NN>
__global__ __launch_bounds__(1024) void b3wgrad_reduce_kernel(W3ReduceArgs a) {
  using G = W3<K, NN>;
  __shared__ float part[4][256];
  const int u = blockIdx.x, el = threadIdx.x, sl = threadIdx.y;
  const size_t per = (size_t)G::NT * 256;
  const int q = (a.nslabs + 3) / 4, k0 = sl * q, k1 = k0 + q < a.nslabs ? k0 + q : a.nslabs;
  const float* p = a.slab + (size_t)u * 256 + el;
  const size_t pstep = per * a.stride;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int k = k0;
  for (; k + 3 < k1; k += 4) {
    s0 += p[(size_t)k * pstep]; s1 += p[(size_t)(k + 1) * pstep]; s2 += p[(size_t)(k + 2) * pstep]; s3 += p[(size_t)(k + 3) * pstep];
  }
  for (; k < k1; ++k) s0 += p[(size_t)k * pstep];
  part[sl][el] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (sl != 0) return;
  const int row = el >> 4, col = el & 15;
  const int dz = u / G::TPD, i = u - dz * G::TPD;
  const int j = K == 8 ? 2 * i + (row >> 3) : i, ci = K == 8 ? (row & 7) : row;
  if (j > 8 || ci >= a.Kw || col >= a.Nw || col >= NN) return;
  const int tw = a.tapw[dz * 9 + j];
  if (tw < 0) return;
  a.dw[((size_t)tw * a.Kw + ci) * a.Nw + col] += (part[0][el] + part[1][el]) + (part[2][el] + part[3][el]);
}

// ---- 8 -> 8: plane-PAIR form ---------------------------------------------------------------------------------------------
// With 8 produced channels the 16 MFMA columns are half empty.  As the fp32 twgradz kernel does, columns 0..7 take dz of plane
// q and columns 8..15 dz of plane q + 1 while the A operand (a tap pair of x plane q - 1 + s, s = 0..3) is shared: rows of x
// plane slot s are tap plane s for the first half and tap plane s - 1 for the second.  4 x 5 = 20 tiles serve two planes instead
// of 2 x 15, i.e. 2/3 of the LDS reads and MFMAs per voxel (the kernel above is bound by its transposing LDS reads).
// 32 x 8 tiles, ring of six x planes (four in use, two arriving) and four dz planes: 49 KB, three workgroups per CU.
struct WZ {
  static constexpr int TY = 8, PX = 34, PY = TY + 2;
  static constexpr int XPIECES = PX * PY, XPLANE = XPIECES * 16;        // 340 pieces
  static constexpr int DPIECES = 32 * TY, DPLANE = DPIECES * 16;         // 256 pieces
  static constexpr int NXS = (XPIECES + 255) / 256;                      // 2
  static constexpr int TPD = 5, NT = 4 * TPD, MTW = NT / 4;              // 20 tiles, 5 per wave
  static constexpr int LDS = 6 * XPLANE + 4 * DPLANE;
};

template <bool AFF>
__global__ __launch_bounds__(256, 3) void b3wgradz_kernel(W3Args a) {
  using G = WZ;
  constexpr int PX = G::PX, TY = G::TY, MTW = G::MTW, TPD = G::TPD;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* xring = smem;
  unsigned char* dbuf = smem + 6 * G::XPLANE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Gk = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
  int bid = blockIdx.x;
  const int nblk = gridDim.x;
  if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);
  const int tx = bid % a.ntx;
  int r_ = bid / a.ntx;
  const int ty = r_ % a.nty;
  r_ /= a.nty;
  const int zs = r_ % a.nzseg, n = r_ / a.nzseg;
  const int x0 = tx * 32, y0 = ty * TY, z0 = zs * a.zseg;
  const int z1 = z0 + a.zseg < a.Z ? z0 + a.zseg : a.Z;

  // byte offsets inside a z plane (fp32 scalar input: half the byte offset); URSN_OOB_BYTES outside the image (buffer_stage.h)
  unsigned xrel[G::NXS], drel;
  unsigned xval = 0;
#pragma unroll
  for (int i = 0; i < G::NXS; ++i) {
    const int idx = tid + 256 * i;
    xrel[i] = URSN_OOB_BYTES;
    if (idx < G::XPIECES) {
      const int yy = idx / PX, xx = idx - yy * PX;
      const int gy = y0 + yy - 1, gx = x0 + xx - 1;
      if (gy >= 0 && gy < a.Y && gx >= 0 && gx < a.X) { xval |= 1u << i; xrel[i] = (unsigned)((gy * a.X + gx) * a.in_cs) * 2u; }
    }
    asm volatile("" : "+v"(xrel[i]));
  }
  {
    const int gy = y0 + (tid >> 5), gx = x0 + (tid & 31);
    drel = (gy < a.Y && gx < a.X) ? (unsigned)((gy * a.X + gx) * a.out_cs) * 2u : URSN_OOB_BYTES;
  }
  const size_t x_plane = (size_t)a.Y * a.X * a.in_cs, d_plane = (size_t)a.Y * a.X * a.out_cs;
  const bf16_t* x_img = a.S + (size_t)n * a.Z * x_plane;
  const bf16_t* d_img = a.C + (size_t)n * a.Z * d_plane;
  float asc[8], ash[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    asc[j] = 1.f; ash[j] = 0.f;
    if constexpr (AFF) { asc[j] = a.aff_rstd[j]; ash[j] = fmaf(-a.aff_mean[j], asc[j], a.aff_beta[j]); }
  }
  u32x4 xs[2][G::NXS], ds[2];
  unsigned xin[2] = {0u, 0u};
  auto load_x = [&](int p, int k) {
    const bool pz = p >= 0 && p < a.Z;
    if (a.S_f32) {   // in_cs = 1
      const __amdgpu_buffer_rsrc_t r = ursn_rsrc(a.S_f32 + ((ptrdiff_t)n * a.Z + p) * (ptrdiff_t)a.Y * a.X, pz ? (unsigned)a.Y * a.X * 4u : 0u);
#pragma unroll
      for (int i = 0; i < G::NXS; ++i) {
        unsigned cv = (unsigned)f2bf(__uint_as_float(ursn_bload_b32(r, xrel[i] * 2u)));
        asm volatile("" : "+v"(cv));   // see b3conv: keeps the vectoriser from pairing the conversions
        xs[k][i] = (u32x4){cv, 0u, 0u, 0u};
      }
    } else {
      const __amdgpu_buffer_rsrc_t r = ursn_rsrc(x_img + (ptrdiff_t)p * (ptrdiff_t)x_plane, pz ? (unsigned)x_plane * 2u : 0u);
#pragma unroll
      for (int i = 0; i < G::NXS; ++i) xs[k][i] = ursn_bload_b128(r, xrel[i]);
    }
    xin[k] = pz ? xval : 0u;
  };
  auto store_x = [&](int p, int k) {   // plane p lives in ring slot (p + 1) % 6
    unsigned char* dst = xring + ((p + 7) % 6) * G::XPLANE;
#pragma unroll
    for (int i = 0; i < G::NXS; ++i) {
      const int idx = tid + 256 * i;
      if constexpr (AFF) {
        if ((xin[k] >> i) & 1u) {
          float f[8];
          unpack8(xs[k][i], f);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            f[j] = fmaf(f[j], asc[j], ash[j]);
            if (a.aff_relu) f[j] = fmaxf(f[j], 0.f);
          }
          xs[k][i] = pack8(f);
        }
      }
      if (idx < G::XPIECES) *(u32x4*)(dst + idx * 16) = xs[k][i];
    }
  };
  auto load_d = [&](int q, int k) {
    ds[k] = ursn_bload_b128(ursn_rsrc(d_img + (ptrdiff_t)q * (ptrdiff_t)d_plane, q < z1 ? (unsigned)d_plane * 2u : 0u), drel);
  };
  auto store_d = [&](int q, int k) { *(u32x4*)(dbuf + (q & 3) * G::DPLANE + tid * 16) = ds[k]; };

  // this wave's tiles u = wave + 4 m: x plane slot sm = u / 5, tap pair i = u % 5 of that plane
  int sm[MTW];
  unsigned aoff[MTW];
#pragma unroll
  for (int m = 0; m < MTW; ++m) {
    const int u = wave + 4 * m;
    sm[m] = u / TPD;
    const int i = u - sm[m] * TPD;
    int j = 2 * i + (tp >> 1);
    if (j > 8) j = 8;
    aoff[m] = (unsigned)(((j / 3) * PX + (j % 3) + 8 * Gk + tq) * 16 + (tp & 1) * 8);
  }
  const unsigned boff = (unsigned)((8 * Gk + tq) * 16 + (tp & 1) * 8);
  const int bhalf = tp >> 1;   // column chunks 0, 1: plane q; 2, 3: plane q + 1

  bf_f32x4 acc[MTW];
#pragma unroll
  for (int m = 0; m < MTW; ++m) acc[m] = (bf_f32x4){0.f, 0.f, 0.f, 0.f};

  // prologue: x planes z0-1 .. z0+2, dz planes z0, z0+1
  for (int p = z0 - 1; p <= z0 + 2; p += 2) { load_x(p, 0); load_x(p + 1, 1); store_x(p, 0); store_x(p + 1, 1); }
  load_d(z0, 0); load_d(z0 + 1, 1); store_d(z0, 0); store_d(z0 + 1, 1);
  __syncthreads();
  for (int q = z0; q < z1; q += 2) {
    load_x(q + 3, 0); load_x(q + 4, 1);
    load_d(q + 2, 0); load_d(q + 3, 1);
    const unsigned char* db = dbuf + ((q + bhalf) & 3) * G::DPLANE + boff;
    unsigned sb[MTW];
    // plane q - 1 + s -> slot (q + s) % 6: the four live slot offsets, the remainder taken once per step (was once per tile)
    const int q6 = q % 6;
    unsigned so[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) so[j] = (unsigned)((q6 + j >= 6 ? q6 + j - 6 : q6 + j) * G::XPLANE);
#pragma unroll
    for (int m = 0; m < MTW; ++m) sb[m] = (sm[m] == 0 ? so[0] : sm[m] == 1 ? so[1] : sm[m] == 2 ? so[2] : so[3]) + aoff[m];
    // rows fully unrolled: every LDS offset of the 96 transposing reads is an immediate
#pragma unroll
    for (int r = 0; r < TY; ++r) {
      const unsigned char* bp = db + r * (32 * 16);
      const bfx8 B = w3_tr_pair(bp, bp + 4 * 16);
#pragma unroll
      for (int m = 0; m < MTW; ++m) {
        const unsigned char* ap = xring + sb[m] + r * (PX * 16);
        const bfx8 A = w3_tr_pair(ap, ap + 4 * 16);
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, acc[m], 0, 0, 0);
      }
    }
    store_x(q + 3, 0); store_x(q + 4, 1);
    store_d(q + 2, 0); store_d(q + 3, 1);
    __syncthreads();
  }
  float* sl = a.slab + (size_t)blockIdx.x * (G::NT * 256);
#pragma unroll
  for (int m = 0; m < MTW; ++m) {
    const int u = wave + 4 * m;
#pragma unroll
    for (int r = 0; r < 4; ++r) sl[(size_t)u * 256 + (4 * Gk + r) * 16 + li] = acc[m][r];
  }
}

// dw element (tap plane dzt, in-plane tap j, ci, co) = slab tile (s = dzt, i)[row][co] + tile (s = dzt + 1, i)[row][8 + co]
__global__ __launch_bounds__(1024) void b3wgradz_reduce_kernel(W3ReduceArgs a) {
  __shared__ float part[4][128];
  const int blk = blockIdx.x;               // (dzt, i): 3 x 5
  const int dzt = blk / WZ::TPD, i = blk - dzt * WZ::TPD;
  const int el = threadIdx.x, sl = threadIdx.y;   // el: (row 0..15, co 0..7)
  const int row = el >> 3, co = el & 7;
  const size_t per = (size_t)WZ::NT * 256 * a.stride;
  const float* p0 = a.slab + (size_t)(dzt * WZ::TPD + i) * 256 + row * 16 + co;
  const float* p1 = a.slab + (size_t)((dzt + 1) * WZ::TPD + i) * 256 + row * 16 + 8 + co;
  const int qn = (a.nslabs + 3) / 4, k0 = sl * qn, k1 = k0 + qn < a.nslabs ? k0 + qn : a.nslabs;
  float s0 = 0.f, s1 = 0.f;
  for (int k = k0; k < k1; ++k) { s0 += p0[(size_t)k * per]; s1 += p1[(size_t)k * per]; }
  part[sl][el] = s0 + s1;
  __syncthreads();
  if (sl != 0) return;
  const int j = 2 * i + (row >> 3), ci = row & 7;
  if (j > 8 || ci >= a.Kw || co >= a.Nw) return;
  const int tw = a.tapw[dzt * 9 + j];
  if (tw < 0) return;
  a.dw[((size_t)tw * a.Kw + ci) * a.Nw + co] += (part[0][el] + part[1][el]) + (part[2][el] + part[3][el]);
}

struct W3Plan { int zseg, nzseg, nty, ntx, grid; };
static bool w3_pair(const GatherGeom& g) {
  static const bool off = getenv("URSN_B3WGRAD_PAIR") && getenv("URSN_B3WGRAD_PAIR")[0] == '0';
  return !off && g.K == 8 && g.Nn == 8;
}
W3Plan w3_plan(const GatherGeom& g) {
  W3Plan p;
  const int Z = g.in_d[0], Y = g.in_d[1], X = g.in_d[2];
  const int TY = (g.K == 8 && !w3_pair(g)) ? 16 : 8;
  p.ntx = (X + 31) / 32;
  p.nty = (Y + TY - 1) / TY;
  const int64_t tiles = (int64_t)g.N * p.nty * p.ntx;
  // long columns (three prologue planes each, one slab each), but at least ~2 workgroups per CU slot
  static const int64_t minwg = getenv("URSN_B3W_MINWG") ? atoi(getenv("URSN_B3W_MINWG")) : 1024;   // A/B
  int nz = (int)((minwg + tiles - 1) / tiles);
  if (nz < 1) nz = 1;
  if (nz > (Z + 7) / 8) nz = (Z + 7) / 8;
  p.zseg = (Z + nz - 1) / nz;
  p.nzseg = (Z + p.zseg - 1) / p.zseg;
  p.grid = (int)(tiles * p.nzseg);
  return p;
}

}  // namespace

bool b3wgrad_ok(const GatherGeom& g) {
  static const bool off = getenv("URSN_B3WGRAD") && getenv("URSN_B3WGRAD")[0] == '0';
  if (off) return false;
  {   // buffer-path staging (buffer_stage.h): a z plane of either tensor must stay below the out-of-range marker
    const int64_t pv = (int64_t)g.in_d[1] * g.in_d[2], qv = (int64_t)g.out_d[1] * g.out_d[2];
    const int64_t cs = g.in_cs > g.out_cs ? g.in_cs : g.out_cs;
    if ((pv > qv ? pv : qv) * cs * 2 >= (int64_t)0x40000000) return false;
  }
  if (!((g.K == 8 && g.Nn == 8) || (g.K == 16 && (g.Nn == 8 || g.Nn == 16)))) return false;
  if (g.ntaps != 27 || (g.in_cs & 7) || (g.out_cs & 7)) return false;
  for (int j = 0; j < 3; ++j)
    if (g.si[j] != 1 || g.in_d[j] != g.q_d[j]) return false;
  for (int t = 0; t < 27; ++t)
    for (int j = 0; j < 3; ++j)
      if (g.tap_d[t][j] < -1 || g.tap_d[t][j] > 1) return false;
  if ((int64_t)g.in_d[1] * g.in_d[2] * (g.in_cs > g.out_cs ? g.in_cs : g.out_cs) >= ((int64_t)1 << 31)) return false;
  return w3_plan(g).grid <= (1 << 20);
}

bool b3wgrad_scalar_ok(const GatherGeom& g) { return b3wgrad_ok(g) && w3_pair(g); }
size_t b3wgrad_scratch_bytes(const GatherGeom& g) {
  const int nt = g.K == 8 ? (w3_pair(g) ? WZ::NT : 15) : 27;
  return (size_t)w3_plan(g).grid * nt * 256 * sizeof(float) + 256;
}

template <int K, int NN, bool AFF>
static int w3_launch(const W3Plan& p, const W3Args& a, const W3ReduceArgs& r, hipStream_t s) {
  using G = W3<K, NN>;
  auto kern = b3wgrad_kernel<K, NN, AFF>;
  static bool attr = false;
  if (!attr) {
    URSN_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS));
    attr = true;
  }
  hipLaunchKernelGGL(kern, dim3(p.grid), dim3(256), G::LDS, s, a);
  URSN_HIP(hipGetLastError());
  W3ReduceArgs r2 = r;
  URSN_TRY(slab_fold(a.slab, r2.nslabs, r2.stride, (size_t)G::NT * 256, s));
  hipLaunchKernelGGL((b3wgrad_reduce_kernel<K, NN>), dim3(G::NT), dim3(256, 4), 0, s, r2);
  URSN_HIP(hipGetLastError());
  return 0;
}

int launch_b3wgrad(const GatherGeom& g, const bf16_t* S, const bf16_t* C, float* dw, int Kw, int Nw, void* scratch,
                   size_t scratch_bytes, hipStream_t s, const B3Affine* aff, const float* S_f32) {
  URSN_REQUIRE(b3wgrad_ok(g), "bf16 3x3x3 wgrad: unsupported geometry");
  URSN_REQUIRE(scratch && scratch_bytes >= b3wgrad_scratch_bytes(g), "bf16 3x3x3 wgrad: scratch too small");
  const W3Plan p = w3_plan(g);
  W3Args a;
  a.S = S; a.C = C; a.slab = (float*)scratch;
  a.N = g.N; a.Z = g.in_d[0]; a.Y = g.in_d[1]; a.X = g.in_d[2]; a.in_cs = g.in_cs; a.out_cs = g.out_cs;
  a.zseg = p.zseg; a.nzseg = p.nzseg; a.nty = p.nty; a.ntx = p.ntx;
  a.aff_mean = a.aff_rstd = a.aff_beta = nullptr; a.aff_relu = 0;
  URSN_REQUIRE(!S_f32 || (w3_pair(g) && !aff), "bf16 3x3x3 wgrad: the scalar fp32 input form needs the plane-pair kernel");
  a.S_f32 = S_f32;
  if (S_f32) a.in_cs = 1;
  if (aff) {
    URSN_REQUIRE(aff->mean && aff->rstd && aff->beta, "bf16 3x3x3 wgrad: incomplete normalise-on-load arguments");
    a.aff_mean = aff->mean; a.aff_rstd = aff->rstd; a.aff_beta = aff->beta; a.aff_relu = aff->relu;
  }
  W3ReduceArgs r;
  r.stride = 1;
  r.slab = a.slab; r.dw = dw; r.nslabs = p.grid; r.Kw = Kw > 0 ? Kw : g.K; r.Nw = Nw > 0 ? Nw : g.Nn;
  for (int i = 0; i < 27; ++i) r.tapw[i] = -1;
  for (int t = 0; t < g.ntaps; ++t) r.tapw[(g.tap_d[t][0] + 1) * 9 + (g.tap_d[t][1] + 1) * 3 + (g.tap_d[t][2] + 1)] = g.tap_w[t];
  ursn_note_kernel(g.K == 8 ? (w3_pair(g) ? "b3wgrad_bf16<8,8>(pair)" : "b3wgrad_bf16<8,8>") : (g.Nn == 8 ? "b3wgrad_bf16<16,8>" : "b3wgrad_bf16<16,16>"));
  if (w3_pair(g)) {
    static bool attr = false;
    // resident workgroups per CU of this weight-gradient kernel beside the main stream's kernels: 49 KB = three (147 of the CU's
    // 160 KB of LDS); URSN_B3WGRADZ_LDSPAD=<KB> pads the request (16: two, 32: one) -- A/B of the LDS share of the two streams.
    // Measured (round 4, cfg5, two rounds on one box): 39.11 / 39.19 ms per step at three, 39.00 / 39.12 at two, 39.12 / 39.09 at
    // one -- unlike fp32's twgradz (conv_tiled.hip, URSN_WGRADZ_OCC3) the share does not matter here; default unchanged
    static const int lds_z = WZ::LDS + 1024 * (getenv("URSN_B3WGRADZ_LDSPAD") ? atoi(getenv("URSN_B3WGRADZ_LDSPAD")) : 0);
    if (!attr) {
      URSN_HIP(hipFuncSetAttribute((const void*)b3wgradz_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_z));
      URSN_HIP(hipFuncSetAttribute((const void*)b3wgradz_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_z));
      attr = true;
    }
    if (aff) hipLaunchKernelGGL(b3wgradz_kernel<true>, dim3(p.grid), dim3(256), lds_z, s, a);
    else hipLaunchKernelGGL(b3wgradz_kernel<false>, dim3(p.grid), dim3(256), lds_z, s, a);
    URSN_HIP(hipGetLastError());
    URSN_TRY(slab_fold(a.slab, r.nslabs, r.stride, (size_t)WZ::NT * 256, s));
    hipLaunchKernelGGL(b3wgradz_reduce_kernel, dim3(3 * WZ::TPD), dim3(128, 4), 0, s, r);
    URSN_HIP(hipGetLastError());
    return 0;
  }
  if (aff) {
    if (g.K == 8) return w3_launch<8, 8, true>(p, a, r, s);
    if (g.Nn == 8) return w3_launch<16, 8, true>(p, a, r, s);
    return w3_launch<16, 16, true>(p, a, r, s);
  }
  if (g.K == 8) return w3_launch<8, 8, false>(p, a, r, s);
  if (g.Nn == 8) return w3_launch<16, 8, false>(p, a, r, s);
  return w3_launch<16, 16, false>(p, a, r, s);
}
