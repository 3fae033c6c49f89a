// bf16 weight gradient of the stride-2 layers between the two finest levels of an F = 8 network (fine tensor 8 channels, coarse
// tensor 16): the first stride-2 conv 8 -> 16 (lib/resnet_module.py:25-43 as called by lib/uresnet.py:56-64) -- optionally with
// the weight gradient of its unit's 1x1 stride-2 shortcut in the SAME pass -- and the last transposed conv 16 -> 8
// (lib/uresnet.py:72-79):
//     dW[t][ci][co] += sum_q S[2 q + d_t][ci] * C[q][co]        (S: the fine tensor, C: the coarse one; fp32 accumulation)
//
// The generic kernel took 0.42 ms for the 3x3x3 layer and 0.24 ms more for the shortcut, each reading the 1.07 GB fine tensor
// (256^3 x 4).  Here a workgroup marches z over a 16 x 8 coarse tile exactly as bf16_s2k8.hip does (ring of five DMA'd fine planes)
// with the coarse dz tiles beside them; v_mfma_f32_16x16x32_bf16 with M = (two taps x 8 channels) per tile (14 tiles for 27 taps,
// dealt to the four waves), N = 16 produced channels, k = 32 coarse voxels; both operands are transposed on the way out of the
// [voxel][channel] images (ds_read_b64_tr_b16; the lane's address picks tap and voxel).  The shortcut is one more MFMA per k step:
// tile 0's operand (tap 0 IS the voxel 2 q when the conv pads nothing in front) against the shortcut's dz.  One fp32 slab per
// workgroup, summed in workgroup order by the reduce kernel.
#include <stdlib.h>
#include <string.h>

#include "bf16_common.h"
#include "buffer_stage.h"

namespace {

constexpr int W2_TX = 16, W2_TY = 8, W2_FX = 2 * W2_TX + 1, W2_FY = 2 * W2_TY + 1;
constexpr int W2_PIECES = W2_FX * W2_FY, W2_PLANE = ((W2_PIECES + 63) / 64) * 64 * 16, W2_NST = (W2_PIECES + 255) / 256;
constexpr int W2_CT = W2_TX * W2_TY * 32;   // bytes of a coarse dz tile: 128 voxels x 16 channels
constexpr int W2_SLAB = 15 * 256;           // floats: 14 tap tiles + the shortcut tile, [16 rows][16 columns] each
typedef short w2_s16x4 __attribute__((ext_vector_type(4)));

struct W2Args {
  const bf16_t* S;         // fine tensor (N, Zf, Yf, Xf, s_cs), 8 channels read
  const bf16_t* C;         // coarse tensor (N, Zc, Yc, Xc, c_cs), 16 channels
  const bf16_t* C2;        // the shortcut's dz (coarse, c2_cs) or null
  float* slab;             // [grid][15][16][16]
  int N, Zf, Yf, Xf, Zc, Yc, Xc;
  int s_cs, c_cs, c2_cs;
  int dmin[3];
  int zseg, nzseg, nty, ntx;
  int toff[28];            // in-plane LDS byte offset of tap t | dz << 28 (t = 27: tap 26 again, its rows are dropped)
};

__device__ __forceinline__ w2_s16x4 w2_tr16(const unsigned char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((w2_s16x4 __attribute__((address_space(3)))*)p);
}
__device__ __forceinline__ bfx8 w2_operand(const unsigned char* p0, const unsigned char* p1) {
  const w2_s16x4 lo = w2_tr16(p0), hi = w2_tr16(p1);
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  s16x8 v;
  v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
  return __builtin_bit_cast(bfx8, v);
}

template <bool SC>
__global__ __launch_bounds__(256, 2) void bs2k8w_kernel(W2Args a) {
  __shared__ __attribute__((aligned(16))) unsigned char ring[5 * W2_PLANE];
  __shared__ __attribute__((aligned(16))) unsigned char ctile[2][SC ? 2 : 1][W2_CT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, g = lane >> 4, tq = li >> 2, tp = li & 3;
  int b = blockIdx.x;
  const int tx = b % a.ntx; b /= a.ntx;
  const int ty = b % a.nty; b /= a.nty;
  const int zs = b % a.nzseg, n = b / a.nzseg;
  const int x0 = tx * W2_TX, y0 = ty * W2_TY, z0 = zs * a.zseg;
  const int z1 = z0 + a.zseg < a.Zc ? z0 + a.zseg : a.Zc;

  // ---- staging: fine planes as in bf16_s2k8.hip, coarse tiles: piece tid = (voxel tid >> 1, half tid & 1) ----
  unsigned soff[W2_NST];
#pragma unroll
  for (int i = 0; i < W2_NST; ++i) {
    const int idx = i * 256 + tid, fy = idx / W2_FX, fx = idx - fy * W2_FX;
    const int gy = 2 * y0 + a.dmin[1] + fy, gx = 2 * x0 + a.dmin[2] + fx;
    soff[i] = (idx < W2_PIECES && gy >= 0 && gy < a.Yf && gx >= 0 && gx < a.Xf) ? (unsigned)((gy * a.Xf + gx) * a.s_cs) * 2u : URSN_OOB_BYTES;
  }
  unsigned coff, c2off = URSN_OOB_BYTES;
  {
    const int cv = tid >> 1, hf = tid & 1, qy = cv >> 4, qx = cv & 15;
    const bool ok = y0 + qy < a.Yc && x0 + qx < a.Xc;
    coff = ok ? (unsigned)(((y0 + qy) * a.Xc + x0 + qx) * a.c_cs + 8 * hf) * 2u : URSN_OOB_BYTES;
    if (SC) c2off = ok ? (unsigned)(((y0 + qy) * a.Xc + x0 + qx) * a.c2_cs + 8 * hf) * 2u : URSN_OOB_BYTES;
  }
  const size_t fplane = (size_t)a.Yf * a.Xf * a.s_cs;
  const bf16_t* simg = a.S + (size_t)n * a.Zf * fplane;
  const size_t cplane = (size_t)a.Yc * a.Xc;
  const int pbase = 2 * z0 + a.dmin[0];
  auto dma_fine = [&](int fp) {
    const int p = pbase + fp;
    const bool ok = p >= 0 && p < a.Zf;
    const __amdgpu_buffer_rsrc_t r = ursn_rsrc(simg + (size_t)(ok ? p : 0) * fplane, ok ? (unsigned)(fplane * 2) : 0u);
    unsigned char* dst = ring + (size_t)(fp % 5) * W2_PLANE + wave * 1024;
#pragma unroll
    for (int i = 0; i < W2_NST; ++i)
      if (i * 256 + wave * 64 < W2_PLANE / 16) ursn_bload_lds_b128(r, dst + i * 4096, soff[i]);
  };
  auto dma_coarse = [&](int z, int slot) {
    const __amdgpu_buffer_rsrc_t r = ursn_rsrc(a.C + ((size_t)n * a.Zc + z) * cplane * a.c_cs, (unsigned)(cplane * a.c_cs * 2));
    ursn_bload_lds_b128(r, &ctile[slot][0][0] + wave * 1024, coff);
    if constexpr (SC) {
      const __amdgpu_buffer_rsrc_t r2 = ursn_rsrc(a.C2 + ((size_t)n * a.Zc + z) * cplane * a.c2_cs, (unsigned)(cplane * a.c2_cs * 2));
      ursn_bload_lds_b128(r2, &ctile[slot][1][0] + wave * 1024, c2off);
    }
  };

  // ---- operand geometry ----
  // transposing read r of a lane: coarse voxels j = 8 g + 4 r + tq of the 32-voxel k step (rows 2 ks + (j >> 4), column j & 15)
  // A: piece tp = (tap of the tile's pair tp >> 1, channels 4 (tp & 1) ..); B: channels 4 tp .. of the voxel's 16
  unsigned av[2], bv[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int j = 8 * g + 4 * r + tq;
    av[r] = (unsigned)(((2 * (j >> 4)) * W2_FX + 2 * (j & 15)) * 16 + (tp & 1) * 8);
    bv[r] = (unsigned)(j * 32 + tp * 8);
  }
  // this wave's tiles: i = wave, wave + 4, wave + 8, wave + 12 (< 14)
  unsigned tlo[4];
  int tdz[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int tile = wave + 4 * k;
    const int t = a.toff[(tile < 14 ? 2 * tile : 26) + (tp >> 1)];
    tlo[k] = (unsigned)(t & 0x0fffffff);
    tdz[k] = (t >> 28) & 3;
  }
  bf_f32x4 acc[4], acc2 = (bf_f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 4; ++k) acc[k] = (bf_f32x4){0.f, 0.f, 0.f, 0.f};

  dma_fine(0); dma_fine(1); dma_fine(2);
  dma_coarse(z0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int z = z0; z < z1; ++z) {
    const int k0 = 2 * (z - z0), cs = (z - z0) & 1;
    if (z + 1 < z1) { dma_fine(k0 + 3); dma_fine(k0 + 4); dma_coarse(z + 1, cs ^ 1); }
    const unsigned sl[3] = {(unsigned)((k0 % 5) * W2_PLANE), (unsigned)(((k0 + 1) % 5) * W2_PLANE), (unsigned)(((k0 + 2) % 5) * W2_PLANE)};
    unsigned base[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) base[k] = tlo[k] + (tdz[k] == 0 ? sl[0] : (tdz[k] == 1 ? sl[1] : sl[2]));
    const unsigned char* cb = &ctile[cs][0][0];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const unsigned kso = (unsigned)((4 * ks) * W2_FX * 16);   // fine rows 2 (2 ks + ..)
      const bfx8 B = w2_operand(cb + ks * 1024 + bv[0], cb + ks * 1024 + bv[1]);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (wave + 4 * k < 14) {
          const bfx8 A = w2_operand(ring + base[k] + kso + av[0], ring + base[k] + kso + av[1]);
          acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, acc[k], 0, 0, 0);
          if (SC && k == 0 && wave == 0) {
            const bfx8 B2 = w2_operand(cb + W2_CT + ks * 1024 + bv[0], cb + W2_CT + ks * 1024 + bv[1]);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B2, acc2, 0, 0, 0);
          }
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // slab: tile i at [i][row 4 g + r][column li]; the shortcut's tile behind the 14
  float* sl_ = a.slab + (size_t)blockIdx.x * W2_SLAB;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int tile = wave + 4 * k;
    if (tile >= 14) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) sl_[tile * 256 + (4 * g + r) * 16 + li] = acc[k][r];
  }
  if (wave == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) sl_[14 * 256 + (4 * g + r) * 16 + li] = SC ? acc2[r] : 0.f;
  }
}

// dw[tap_w[t]][ci][co] += sum over workgroup slabs (fixed order); block = one element (tap | shortcut, ci, co)
struct W2RedArgs { const float* slab; float* dw; float* dw2; int nslabs, w_tap_stride, w_sk, w_sn, Kw, Nw; int tap_w[27]; };
__global__ __launch_bounds__(256) void bs2k8w_reduce_kernel(W2RedArgs a) {
  __shared__ float sm[256];
  const int e = blockIdx.x;             // 0 .. 28 * 128 - 1: (t = e >> 7 (27: shortcut), ci = (e >> 4) & 7, co = e & 15)
  const int t = e >> 7, ci = (e >> 4) & 7, co = e & 15;
  // tap t lives in tile t >> 1, rows 8 (t & 1) + ci; the shortcut in tile 14, rows ci (tap 0's rows of tile 0's operand)
  const int off = t < 27 ? (t >> 1) * 256 + (8 * (t & 1) + ci) * 16 + co : 14 * 256 + ci * 16 + co;
  float s = 0.f;
  for (int i = threadIdx.x; i < a.nslabs; i += 256) s += a.slab[(size_t)i * W2_SLAB + off];
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0 && ci < a.Kw && co < a.Nw) {
    if (t < 27) a.dw[(size_t)a.tap_w[t] * a.w_tap_stride + (size_t)ci * a.w_sk + (size_t)co * a.w_sn] += sm[0];
    else if (a.dw2) a.dw2[ci * 16 + co] += sm[0];
  }
}

struct W2Plan { int zseg, nzseg, nty, ntx, grid, dmin[3]; };
bool w2_plan(const GatherGeom& g, W2Plan& p) {
  static const bool off = getenv("URSN_BS2K8") && getenv("URSN_BS2K8")[0] == '0';
  if (off) return false;
  if (g.ntaps != 27 || g.K != 8 || g.Nn != 16 || (g.in_cs & 7) || (g.out_cs & 7)) return false;
  for (int j = 0; j < 3; ++j) {
    if (g.si[j] != 2) return false;
    p.dmin[j] = 1 << 20;
    for (int t = 0; t < 27; ++t) if (g.tap_d[t][j] < p.dmin[j]) p.dmin[j] = g.tap_d[t][j];
    for (int t = 0; t < 27; ++t) if (g.tap_d[t][j] - p.dmin[j] > 2) return false;
  }
  if ((int64_t)g.in_d[1] * g.in_d[2] * g.in_cs * 2 >= (int64_t)0x40000000) return false;
  if ((int64_t)g.q_d[1] * g.q_d[2] * g.out_cs * 2 >= (int64_t)0x40000000) return false;
  const int Zc = g.q_d[0];
  if (g.q_d[2] < 8 || g.q_d[1] < 4) return false;
  p.ntx = (g.q_d[2] + W2_TX - 1) / W2_TX;
  p.nty = (g.q_d[1] + W2_TY - 1) / W2_TY;
  const int64_t tiles = (int64_t)g.N * p.nty * p.ntx;
  int zseg = Zc;
  while (zseg > 8 && tiles * ((Zc + zseg - 1) / zseg) < 2048) zseg = (zseg + 1) / 2;
  p.zseg = zseg;
  p.nzseg = (Zc + zseg - 1) / zseg;
  if (tiles * p.nzseg > (1 << 18)) return false;
  p.grid = (int)(tiles * p.nzseg);
  return true;
}

}  // namespace

bool bs2k8w_ok(const GatherGeom& g) { W2Plan p; return w2_plan(g, p); }
size_t bs2k8w_scratch_bytes(const GatherGeom& g) { W2Plan p; return w2_plan(g, p) ? (size_t)p.grid * W2_SLAB * sizeof(float) + 256 : 0; }

// C2 / dw2: the 1x1 stride-2 shortcut's dz (coarse, c2_cs) and its weight gradient [8][16] (+=), or null; needs the conv's tap 0 at
// the voxel 2 q (no padding in front)
bool bs2k8w_sc_ok(const GatherGeom& g) {
  W2Plan p;
  return w2_plan(g, p) && p.dmin[0] == 0 && p.dmin[1] == 0 && p.dmin[2] == 0 && g.tap_d[0][0] == 0 && g.tap_d[0][1] == 0 && g.tap_d[0][2] == 0;
}

int launch_bs2k8w(const GatherGeom& g, const bf16_t* S, const bf16_t* C, float* dw, int Kw, int Nw, void* scratch, size_t scratch_bytes,
                  const bf16_t* C2, int c2_cs, float* dw2, hipStream_t s) {
  W2Plan p;
  URSN_REQUIRE(w2_plan(g, p), "bf16 stride-2 weight gradient (8 x 16): unsupported geometry");
  URSN_REQUIRE(scratch && scratch_bytes >= bs2k8w_scratch_bytes(g), "bf16 stride-2 weight gradient (8 x 16): scratch too small");
  URSN_REQUIRE(!C2 || (dw2 && (c2_cs & 7) == 0 && bs2k8w_sc_ok(g)), "bf16 stride-2 weight gradient (8 x 16): bad shortcut arguments");
  W2Args a;
  memset(&a, 0, sizeof(a));
  a.S = S; a.C = C; a.C2 = C2; a.slab = (float*)scratch;
  a.N = g.N; a.Zf = g.in_d[0]; a.Yf = g.in_d[1]; a.Xf = g.in_d[2]; a.Zc = g.q_d[0]; a.Yc = g.q_d[1]; a.Xc = g.q_d[2];
  a.s_cs = g.in_cs; a.c_cs = g.out_cs; a.c2_cs = c2_cs;
  for (int j = 0; j < 3; ++j) a.dmin[j] = p.dmin[j];
  a.zseg = p.zseg; a.nzseg = p.nzseg; a.nty = p.nty; a.ntx = p.ntx;
  for (int t = 0; t < 28; ++t) {
    const int tt = t < 27 ? t : 26;
    const int dz = g.tap_d[tt][0] - p.dmin[0], dy = g.tap_d[tt][1] - p.dmin[1], dx = g.tap_d[tt][2] - p.dmin[2];
    a.toff[t] = ((dy * W2_FX + dx) * 16) | (dz << 28);
  }
  ursn_note_kernel(C2 ? "bs2k8w_bf16<8,16>+sc" : "bs2k8w_bf16<8,16>");
  if (C2) hipLaunchKernelGGL(bs2k8w_kernel<true>, dim3(p.grid), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(bs2k8w_kernel<false>, dim3(p.grid), dim3(256), 0, s, a);
  URSN_HIP(hipGetLastError());
  W2RedArgs r;
  r.slab = a.slab; r.dw = dw; r.dw2 = C2 ? dw2 : nullptr; r.nslabs = p.grid;
  r.w_tap_stride = g.w_tap_stride; r.w_sk = g.w_sk; r.w_sn = g.w_sn; r.Kw = Kw > 0 ? Kw : g.K; r.Nw = Nw > 0 ? Nw : g.Nn;
  for (int t = 0; t < 27; ++t) r.tap_w[t] = g.tap_w[t];
  hipLaunchKernelGGL(bs2k8w_reduce_kernel, dim3((C2 ? 28 : 27) * 128), dim3(256), 0, s, r);
  URSN_HIP(hipGetLastError());
  return 0;
}
