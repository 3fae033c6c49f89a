// fp32 weight gradient of the logits layer conv2 (lib/uresnet.py:94-100: k3 s1, 8 -> num_class = 3 channels, 3-D) on the
// VECTOR pipe -- gfx950.
//
//   dW[t][ci][co] = sum_{n,v} x[n, v + t - 1][ci] * dz[n, v][co]          27 x 8 x 3 = 648 sums
//
// On gfx950 the fp32 vector rate (v_pk_fma_f32) equals the fp32 MFMA rate, and with 3 produced channels every MFMA shape pads:
// twgrad4 (4x4x1 blocks, 3 -> 4 columns, one LDS operand per 8-cycle MFMA) ran this layer at 34 TFLOP/s (1.1 ms at 192^3 x 4
// against 0.23 ms of FLOP time).  Here a lane is a voxel of a 64 x 4 tile that a workgroup walks along z (ring of four x planes
// in LDS as two float4 channel-half planes: a wave's ds_read_b128 is one contiguous KB).  The 27 taps are dealt to the four
// waves (7 + 7 + 7 + 6): a wave holds its taps' 7 x 8 x 3 sums per lane in registers, sees every voxel of the tile, and reads
// each (voxel, tap) operand exactly once; dz (16 bytes per voxel) comes straight from global memory.  Per voxel and tap:
// 2 LDS reads, 12 v_pk_fma_f32.  One cross-lane sum per workgroup at the end, one slab per workgroup, the deterministic slab
// reduce of the other weight-gradient kernels adds it into the gradient buffer.
// Measured at 192^3 x 4: 0.97 ms (twgrad4: 1.10 ms) -- the row loop is bound by LDS latency at two waves per SIMD (240 VGPRs of
// accumulators), not by the 0.3 ms of v_pk_fma issue; kept as the default for this layer, URSN_WGRAD_VALU=0 restores twgrad4.
#include <stdlib.h>

#include "ursn_common.h"

typedef float vw_f32x4 __attribute__((ext_vector_type(4)));
typedef float vw_f32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int VW_TX = 64, VW_TY = 4, VW_PX = VW_TX + 2, VW_PY = VW_TY + 2, VW_PS = VW_PX * VW_PY;   // 396 staged voxels
constexpr int VW_XPLANE = 2 * VW_PS * 4;                     // floats: [channel half][voxel] float4
constexpr int VW_NW = 4, VW_NTHR = 64 * VW_NW;                // taps t = wave + 4 i (7 + 7 + 7 + 6); eight waves x 4 taps measured slower
constexpr int VW_NST = (2 * VW_PS + VW_NTHR - 1) / VW_NTHR;  // staged float4 per thread
constexpr int VW_TPW = 7;                                    // taps per wave
constexpr int VW_DPLANE = VW_TX * VW_TY * 4;                 // floats: one float4 per voxel of the tile
constexpr size_t VW_LDS = ((size_t)4 * VW_XPLANE + 2 * VW_DPLANE) * sizeof(float);

struct VWArgs {
  const float* x;     // (N, Z, Y, X, x_cs), 8 channels
  const float* dz;    // (N, Z, Y, X, dz_cs), 3 channels in a 4-float voxel
  float* slab;        // [grid][27][8][3]
  int N, Z, Y, X, x_cs, dz_cs;
  int zseg, nzseg, nty, ntx;
};

__global__ __launch_bounds__(VW_NTHR, 2) void vwgrad_kernel(VWArgs a) {
  extern __shared__ __attribute__((aligned(16))) float vwl[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int bid = ursn_xcd_block(blockIdx.x, gridDim.x);
  const int xt = bid % a.ntx; bid /= a.ntx;
  const int yt = bid % a.nty; bid /= a.nty;
  const int zs = bid % a.nzseg;
  const int n = bid / a.nzseg;
  const int x0 = xt * VW_TX, y0 = yt * VW_TY, z0 = zs * a.zseg;
  const int z1 = z0 + a.zseg < a.Z ? z0 + a.zseg : a.Z;

  // staging: global piece g = 2 * voxel + half (consecutive threads read consecutive 16-byte pieces)
  int sgo[VW_NST], slo[VW_NST];
  bool sok[VW_NST];
#pragma unroll
  for (int i = 0; i < VW_NST; ++i) {
    const int idx = tid + VW_NTHR * i;
    const int vox = idx >> 1, hf = idx & 1;
    const int yy = vox / VW_PX, xx = vox - yy * VW_PX;
    const int gy = y0 + yy - 1, gx = x0 + xx - 1;
    sok[i] = idx < 2 * VW_PS && gy >= 0 && gy < a.Y && gx >= 0 && gx < a.X;
    sgo[i] = (gy * a.X + gx) * a.x_cs + 4 * hf;
    slo[i] = (hf * VW_PS + vox) * 4;
  }
  vw_f32x4 st[VW_NST];
  auto load_x = [&](int p) {
    const bool pz = p >= 0 && p < a.Z;
    const float* base = a.x + ((size_t)n * a.Z + (pz ? p : 0)) * a.Y * a.X * a.x_cs;
#pragma unroll
    for (int i = 0; i < VW_NST; ++i) {
      vw_f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (pz && sok[i]) v = *(const vw_f32x4*)(base + sgo[i]);
      st[i] = v;
    }
  };
  auto store_x = [&](int p) {   // plane p lives in ring slot (p + 1) & 3
    float* dst = vwl + ((p + 1) & 3) * VW_XPLANE;
#pragma unroll
    for (int i = 0; i < VW_NST; ++i)
      if (tid + VW_NTHR * i < 2 * VW_PS) *(vw_f32x4*)(dst + slo[i]) = st[i];
  };

  // dz plane: one float4 per thread (voxel tid of the 64 x 4 tile), two LDS buffers; read straight from global in the row loop
  // its latency sat in front of every row (1.24 ms)
  float* dl = vwl + 4 * VW_XPLANE;
  const int dy_ = (tid >> 6) & 3, dx_ = tid & 63;
  const bool dok = tid < 256 && y0 + dy_ < a.Y && x0 + dx_ < a.X;
  const size_t dgo = ((size_t)(y0 + dy_) * a.X + x0 + dx_) * a.dz_cs;
  vw_f32x4 dst4;
  auto load_d = [&](int q) {
    dst4 = (vw_f32x4){0.f, 0.f, 0.f, 0.f};
    if (q < z1 && dok) dst4 = *(const vw_f32x4*)(a.dz + ((size_t)n * a.Z + q) * a.Y * a.X * a.dz_cs + dgo);
  };
  auto store_d = [&](int q) { if (tid < 256) *(vw_f32x4*)(dl + (q & 1) * VW_DPLANE + tid * 4) = dst4; };

  // this wave's taps t = wave + 4 i: plane selector and float offset of the tap shift inside a plane
  int t_tz[VW_TPW], t_off[VW_TPW];
#pragma unroll
  for (int i = 0; i < VW_TPW; ++i) {
    int t = wave + VW_NW * i;
    if (t > 26) t = 26;   // slots past the 27th tap: duplicates, not written
    t_tz[i] = t / 9;
    t_off[i] = (((t / 3) % 3) * VW_PX + (t % 3) + lane) * 4;
  }

  vw_f32x2 acc[VW_TPW][3][4];   // [tap][co][ci pair]
#pragma unroll
  for (int i = 0; i < VW_TPW; ++i)
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[i][c][q] = (vw_f32x2){0.f, 0.f};

  for (int p = z0 - 1; p <= z0 + 1; ++p) { load_x(p); store_x(p); }
  load_d(z0);
  store_d(z0);
  __syncthreads();
  for (int q = z0; q < z1; ++q) {
    load_x(q + 2);
    load_d(q + 1);
    const float* dq = dl + (q & 1) * VW_DPLANE + lane * 4;
#pragma unroll 1
    for (int r = 0; r < VW_TY; ++r) {
      const vw_f32x4 g = *(const vw_f32x4*)(dq + r * (VW_TX * 4));
#pragma unroll
      for (int i = 0; i < VW_TPW; ++i) {
        const float* xp = vwl + ((q + t_tz[i]) & 3) * VW_XPLANE + t_off[i] + r * (VW_PX * 4);   // plane q - 1 + tz
        const vw_f32x4 xa = *(const vw_f32x4*)xp, xb = *(const vw_f32x4*)(xp + VW_PS * 4);
        const vw_f32x2 xq[4] = {{xa[0], xa[1]}, {xa[2], xa[3]}, {xb[0], xb[1]}, {xb[2], xb[3]}};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const vw_f32x2 gg = {g[c], g[c]};
#pragma unroll
          for (int k = 0; k < 4; ++k) acc[i][c][k] = __builtin_elementwise_fma(xq[k], gg, acc[i][c][k]);
        }
      }
    }
    store_x(q + 2);
    store_d(q + 1);
    __syncthreads();
  }

  // sum over the 64 lanes (fixed butterfly order), lane 0 writes the wave's taps
  float* sl = a.slab + (size_t)blockIdx.x * (27 * 24);
#pragma unroll
  for (int i = 0; i < VW_TPW; ++i) {
    const int t = wave + VW_NW * i;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          float v = acc[i][c][k][e];
#pragma unroll
          for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
          if (lane == 0 && t < 27) sl[t * 24 + (2 * k + e) * 3 + c] = v;
        }
  }
}

struct VWPlan { int Z, Y, X, ntx, nty, zseg, nzseg, grid; };
bool vw_plan(const ursn_conv_desc& d, VWPlan& p) {
  static const bool off = getenv("URSN_WGRAD_VALU") && getenv("URSN_WGRAD_VALU")[0] == '0';
  if (off || d.ndim != 3 || d.transposed || d.k != 3 || d.stride != 1 || d.cin != 8 || d.cout != 3) return false;
  if (d.in_mean || d.in_split || (d.algo != 0 && d.algo != 3)) return false;
  const int ics = d.in_cstride > 0 ? d.in_cstride : d.cin, ocs = d.out_cstride > 0 ? d.out_cstride : d.cout;
  if ((ics & 3) || ocs != 4) return false;   // dz voxels are 16-byte quads (the net's 4-padded logits buffers)
  p.Z = d.in_sp[0]; p.Y = d.in_sp[1]; p.X = d.in_sp[2];
  if (p.X < 48 || p.Y < 4 || p.Z < 8) return false;
  if ((int64_t)p.Y * p.X * ics >= ((int64_t)1 << 31)) return false;
  p.ntx = (p.X + VW_TX - 1) / VW_TX;
  p.nty = (p.Y + VW_TY - 1) / VW_TY;
  const int64_t base = (int64_t)d.n * p.nty * p.ntx;
  if (base > (1 << 24)) return false;
  ursn_pick_zseg(base, p.Z, 2, 8, p.zseg, p.nzseg);
  p.grid = (int)(base * p.nzseg);
  return true;
}

}  // namespace

int valu_wgrad_supported(const ursn_conv_desc& d) { VWPlan p; return vw_plan(d, p) ? 1 : 0; }
size_t valu_wgrad_scratch_bytes(const ursn_conv_desc& d) {
  VWPlan p;
  return vw_plan(d, p) ? (size_t)p.grid * 27 * 24 * sizeof(float) : 0;
}

int launch_valu_wgrad(const ursn_conv_desc& d, const float* x, const float* dy, float* dw, void* scratch, size_t scratch_bytes,
                      hipStream_t s) {
  VWPlan p;
  URSN_REQUIRE(vw_plan(d, p), "vector-pipe wgrad: unsupported shape");
  URSN_REQUIRE(scratch && scratch_bytes >= valu_wgrad_scratch_bytes(d), "vector-pipe wgrad: scratch too small");
  VWArgs a;
  a.x = x; a.dz = dy; a.slab = (float*)scratch;
  a.N = d.n; a.Z = p.Z; a.Y = p.Y; a.X = p.X;
  a.x_cs = d.in_cstride > 0 ? d.in_cstride : d.cin; a.dz_cs = 4;
  a.zseg = p.zseg; a.nzseg = p.nzseg; a.nty = p.nty; a.ntx = p.ntx;
  static bool attr = false;
  if (!attr) {
    URSN_HIP(hipFuncSetAttribute((const void*)vwgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)VW_LDS));
    attr = true;
  }
  ursn_note_kernel("vwgrad<8,3>");
  hipLaunchKernelGGL(vwgrad_kernel, dim3(p.grid), dim3(VW_NTHR), VW_LDS, s, a);
  URSN_HIP(hipGetLastError());
  return launch_reduce_accum(dw, (const float*)scratch, (int64_t)27 * 24, p.grid, s);
}
