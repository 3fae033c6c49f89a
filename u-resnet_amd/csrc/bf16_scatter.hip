// bf16 stride-2 SCATTER-type passes of the deeper levels in ONE launch: the forward of slim.conv3d_transpose k3 s2
// (lib/uresnet.py:72-79: 256 -> 128, 128 -> 64, 64 -> 32, 32 -> 16 channels of an F = 8 network) and the data gradient of the
// stride-2 convs opening each level (lib/resnet_module.py:25-43 as called by lib/uresnet.py:56-64) -- weight-streaming on
// v_mfma_f32_16x16x32_bf16, the eight output-parity classes split over the four waves of a workgroup.
//
// Evaluated as gathers these passes are eight parity classes of the FINE grid (conv_api.hip::build_geoms: class r = fine voxels
// 2 q + r, 1 / 2 / 4 / 8 of the 27 taps each) and ran as eight launches of the generic box kernel -- each re-staging the coarse
// tensor, re-packing its taps and writing every second voxel of a row (8 us of launch boundaries and 0.13-0.66 ms per pass at
// 1-6 % of the pass's roofline).  Here
//   * a WORKGROUP owns a box of 64 | 128 COARSE voxels x a block of (16 | 32 | 64) produced channels; the halo image of the box
//     (+1 neighbour per axis) with ALL contraction channels is DMA'd once into LDS, piece-major as in bf16_convdeep.hip;
//   * the 27 (class, tap) ITEMS are dealt to the four waves by class (8 | 4+2+1 | 4+2 | 4+2 taps): a wave finishes whole
//     classes, so no partial tile crosses waves and there is no reduction step; an item x chunk of 32 contraction channels is
//     MT x NT MFMAs on one A fragment set (weights, packed per wave as ONE linear stream, straight from L2 to registers, two
//     iterations ahead) and NT B fragments (one ds_read_b128 each);
//   * at a class's last item the wave rounds and stores its fine voxels 2 q + r (8-byte pieces through the buffer path: voxels
//     the box covers beyond the volume are dropped by the bounds check) and, forward, takes the BatchNorm moments of the STORED
//     values with per-lane pivots.
// Weights are the large operand at these levels (27 x K x N x 2 B = 28 KB ... 1.8 MB); a workgroup streams the block's share
// once: bytes from L2 per FLOP = 1 / (8 x box voxels of the fine grid).
#include <stdlib.h>

#include "bf16_common.h"
#include "bf16_pack.h"
#include "buffer_stage.h"

namespace {

#define BS_OOB 0x80000000u

struct BSArgs {
  const bf16_t* in;        // coarse tensor (N, Zc, Yc, Xc, in_cs)
  const bf16_t* wp;        // [cout block][item][chunk][mt][lane][8]
  bf16_t* out;             // fine tensor (N, 2 Zc, 2 Yc, 2 Xc, out_cs)
  double* stats_partial;   // [cout block][tiles][2][64] doubles or null
  int N, Zc, Yc, Xc;
  int in_cs, out_cs, Cout;
  int nchunks;             // contraction channels / 32
  int bq[3], nb[3];        // box of coarse voxels (powers of two), boxes per axis
  int lbx, lby;
  int hy, hx, pp;          // halo image: rows per plane (bq[1] + 1), row length (bq[2] + 1), voxels per piece plane (multiple of 64)
  int dmin[3];             // smallest tap offset per axis: halo position h holds coarse voxel box0 + h + dmin
  int accumulate;
  int wave_first[5];       // items of wave w: wave_first[w] .. wave_first[w + 1] - 1
  int item_toff[27];       // LDS byte offset of the item's neighbour inside a piece plane
  int item_cls[27];        // bits 0-2: class (rz ry rx), bit 3: last item of its class
};

template <int CTRL>
__device__ __forceinline__ double bs_dpp(double v) {
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, 0xf, 0xf, false);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), CTRL, 0xf, 0xf, false);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double bs_row_sum(double v) {   // over the 16 lanes of a DPP row, every lane ends with the total
  v += bs_dpp<0xB1>(v);
  v += bs_dpp<0x4E>(v);
  v += bs_dpp<0x141>(v);
  v += bs_dpp<0x140>(v);
  return v;
}

// MT: 16-channel tiles of produced channels per block (1 | 2 | 4); NT: 16-voxel tiles of the coarse box (4 | 8)
// FAST (MT = 1, one chunk): the wave's whole weight stream in registers (below)
template <int MT, int NT, bool STATS, bool FAST = false>
__global__ __launch_bounds__(256, 2) void bsconv_kernel(BSArgs a) {
  static_assert(!FAST || MT == 1, "register-resident weights: one tile of produced channels");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n16 = lane & 15, g = lane >> 4;
  const int cob = blockIdx.y;
  int tile = blockIdx.x;
  const int bx = tile % a.nb[2]; tile /= a.nb[2];
  const int by = tile % a.nb[1]; tile /= a.nb[1];
  const int bz = tile % a.nb[0];
  const int n = tile / a.nb[0];
  const int z0 = bz * a.bq[0], y0 = by * a.bq[1], x0 = bx * a.bq[2];

  // ---- staging: thread tid owns halo voxel tid of every (chunk, piece) plane; one DMA instruction = 64 pieces of one plane ----
  unsigned vrel = BS_OOB;
  if (tid < a.pp) {
    const int hzz = tid / (a.hy * a.hx), r2 = tid - hzz * a.hy * a.hx;
    const int hyy = r2 / a.hx, hxx = r2 - hyy * a.hx;
    const int gz = z0 + hzz + a.dmin[0], gy = y0 + hyy + a.dmin[1], gx = x0 + hxx + a.dmin[2];
    if (hzz <= a.bq[0] && gz >= 0 && gz < a.Zc && gy >= 0 && gy < a.Yc && gx >= 0 && gx < a.Xc)
      vrel = (unsigned)(((gz * a.Yc + gy) * a.Xc + gx) * a.in_cs) * 2u;
  }
  const size_t img_elems = (size_t)a.Zc * a.Yc * a.Xc * a.in_cs;
  const __amdgpu_buffer_rsrc_t rin = ursn_rsrc(a.in + (size_t)n * img_elems, (unsigned)(img_elems * 2));
  if (wave * 64 < a.pp) {
    for (int c = 0; c < a.nchunks; ++c)
#pragma unroll
      for (int pc = 0; pc < 4; ++pc)
        ursn_bload_lds_b128_so(rin, lds + (size_t)((c * 4 + pc) * a.pp) * 16 + wave * 1024, vrel, (unsigned)(c * 32 + pc * 8) * 2u);
  }

  // ---- per-lane geometry: coarse voxel (tile nt, column n16) of the box; B piece g; fine voxel 2 q of the lane ----
  unsigned vbase[NT], obase[NT];
  const int Yf = 2 * a.Yc, Xf = 2 * a.Xc;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int lin = nt * 16 + n16;
    const int qx = lin & (a.bq[2] - 1), r2 = lin >> a.lbx;
    const int qy = r2 & (a.bq[1] - 1), qz = r2 >> a.lby;
    vbase[nt] = (unsigned)((g * a.pp + (qz * a.hy + qy) * a.hx + qx) * 16);
    const int gz = z0 + qz, gy = y0 + qy, gx = x0 + qx;
    const bool ok = gz < a.Zc && gy < a.Yc && gx < a.Xc;
    obase[nt] = ok ? (unsigned)((((2 * gz) * Yf + 2 * gy) * Xf + 2 * gx) * a.out_cs + cob * (16 * MT) + 4 * g) * 2u : URSN_OOB_BYTES;
  }
  const size_t fine_elems = (size_t)8 * a.Zc * a.Yc * a.Xc * a.out_cs;
  const __amdgpu_buffer_rsrc_t rout = ursn_rsrc(a.out + (size_t)n * fine_elems, (unsigned)(fine_elems * 2));
  const int chunk_bytes = 4 * a.pp * 16;

  bf_f32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (bf_f32x4){0.f, 0.f, 0.f, 0.f};
  float piv[4 * MT], s1[4 * MT], s2[4 * MT], nacc = 0.f;
  if constexpr (STATS) {
#pragma unroll
    for (int k = 0; k < 4 * MT; ++k) piv[k] = s1[k] = s2[k] = 0.f;
  }

  const int it0 = __builtin_amdgcn_readfirstlane(a.wave_first[wave]);
  const int nit = __builtin_amdgcn_readfirstlane(a.wave_first[wave + 1]) - it0;
  const int I = nit * a.nchunks;
  // this wave's weight stream: iteration i (item-major, chunk-minor) is MT KB at wsrc + i * MT * 1024
  const unsigned char* wsrc = (const unsigned char*)a.wp + ((size_t)(cob * 27 + it0) * a.nchunks) * (MT * 1024) + lane * 16;

  bfx8 A[FAST ? 1 : 3][MT], B[FAST ? 1 : 3][NT];
  // prefetch cursor (item, chunk) of iteration i + 2 and compute cursor of iteration i
  int pit = it0, pch = 0, cit = it0, cch = 0;
  auto fetch_a = [&](bfx8 (&Ad)[MT], int i) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) Ad[mt] = *(const bfx8*)(wsrc + (size_t)i * (MT * 1024) + mt * 1024);
  };
  auto fetch_b = [&](bfx8 (&Bd)[NT]) {
    const int off = __builtin_amdgcn_readfirstlane(a.item_toff[pit]) + pch * chunk_bytes;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) Bd[nt] = *(const bfx8*)(lds + vbase[nt] + off);
    if (++pch == a.nchunks) { pch = 0; ++pit; }
  };
  auto finish = [&](int cls) {   // round and store the class's fine voxels; moments of the stored values
    const int rz = (cls >> 2) & 1, ry = (cls >> 1) & 1, rx = cls & 1;
    const unsigned coff = (unsigned)(((rz * Yf + ry) * Xf + rx) * a.out_cs) * 2u;
    // accumulate (data gradients: never together with the forward's moments): every old value of the class is requested before
    // the first is used
    bst_u32x2 oldv[STATS ? 1 : NT][STATS ? 1 : MT];
    if constexpr (!STATS) {
      if (a.accumulate) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) oldv[nt][mt] = ursn_bload_b64(rout, obase[nt] + coff + 32u * mt);
      }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        bf_f32x4 v = acc[mt][nt];
        acc[mt][nt] = (bf_f32x4){0.f, 0.f, 0.f, 0.f};
        if (cob * (16 * MT) + 16 * mt >= a.Cout) continue;
        const unsigned off = obase[nt] + coff + 32u * mt;
        if (!STATS && a.accumulate) {
          const bst_u32x2 e = oldv[STATS ? 0 : nt][STATS ? 0 : mt];
          v[0] += __uint_as_float(e[0] << 16); v[1] += __uint_as_float(e[0] & 0xffff0000u);
          v[2] += __uint_as_float(e[1] << 16); v[3] += __uint_as_float(e[1] & 0xffff0000u);
        }
        bst_u32x2 pk;
        pk[0] = pack_bf2(v[0], v[1]);
        pk[1] = pack_bf2(v[2], v[3]);
        ursn_bstore_b64(pk, rout, off);
        if constexpr (STATS) {
          if (obase[nt] != URSN_OOB_BYTES) {
            const float rv[4] = {__uint_as_float(pk[0] << 16), __uint_as_float(pk[0] & 0xffff0000u),
                                 __uint_as_float(pk[1] << 16), __uint_as_float(pk[1] & 0xffff0000u)};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              if (nacc == 0.f) piv[4 * mt + r] = rv[r];
              ursn_sacc(piv[4 * mt + r], s1[4 * mt + r], s2[4 * mt + r], rv[r]);
            }
          }
        }
      }
      if constexpr (STATS) if (obase[nt] != URSN_OOB_BYTES) nacc += 1.f;
    }
  };

  if constexpr (FAST) {
    // one tile of produced channels and one chunk (32 -> 16: the 64^3 level, HBM-sized passes): an item is 8 MFMAs of 16 cycles --
    // nothing to hide an L2 round trip behind, two items ahead or not.  The wave's whole weight stream (<= 8 KB: 8 items) is
    // requested before the images are waited for and stays in registers
    {
      bfx8 Aall[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) Aall[i] = *(const bfx8*)(wsrc + (size_t)(i < I ? i : 0) * 1024);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      bfx8 Bc[NT], Bn[NT];
      fetch_b(Bc);
#pragma unroll 1
      for (int i = 0; i < I; ++i) {   // (one copy of the epilogue: eight unrolled ones sent the accumulators to scratch)
        if (i + 1 < I) fetch_b(Bn);
        bfx8 Ai = Aall[0];
#pragma unroll
        for (int k = 1; k < 8; ++k) if (i == k) Ai = Aall[k];   // wave-uniform selects instead of indexed registers
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ai, Bc[nt], acc[0][nt], 0, 0, 0);
        const int cls = __builtin_amdgcn_readfirstlane(a.item_cls[cit]);
        ++cit;
        if (cls & 8) finish(cls & 7);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) Bc[nt] = Bn[nt];
      }
    }
  } else {
  if (I > 0) fetch_a(A[0], 0);
  if (I > 1) fetch_a(A[1], 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the image has landed
  __syncthreads();
  if (I > 0) fetch_b(B[0]);
  if (I > 1) fetch_b(B[1]);

  auto body = [&](bfx8 (&Ac)[MT], bfx8 (&Bc)[NT], bfx8 (&An)[MT], bfx8 (&Bn)[NT], int i) {
    if (i + 2 < I) { fetch_a(An, i + 2); fetch_b(Bn); }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ac[mt], Bc[nt], acc[mt][nt], 0, 0, 0);
    if (++cch == a.nchunks) {
      cch = 0;
      const int cls = __builtin_amdgcn_readfirstlane(a.item_cls[cit]);
      ++cit;
      if (cls & 8) finish(cls & 7);
    }
  };
  for (int i = 0; i < I; i += 3) {
    body(A[0], B[0], A[2], B[2], i);
    if (i + 1 < I) body(A[1], B[1], A[0], B[0], i + 1);
    if (i + 2 < I) body(A[2], B[2], A[1], B[1], i + 2);
  }
  }

  if constexpr (STATS) {
    __shared__ double red[4][128];
#pragma unroll
    for (int k = 0; k < 4 * MT; ++k) {
      double u, w2;
      ursn_sacc_final(piv[k], s1[k], s2[k], nacc, u, w2);
      u = bs_row_sum(u); w2 = bs_row_sum(w2);
      if (n16 == 0) {
        const int ch = 16 * (k >> 2) + 4 * g + (k & 3);
        red[wave][ch] = u;
        red[wave][64 + ch] = w2;
      }
    }
    __syncthreads();
    if (tid < 128) {
      const int ch = tid & 63;
      double t = 0.0;
      if (ch < 16 * MT && cob * (16 * MT) + ch < a.Cout) t = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
      a.stats_partial[((size_t)cob * gridDim.x + blockIdx.x) * 128 + tid] = t;
    }
  }
}

// (weight packing: BPK_SCATTER in bf16_pack.hip -- [cout block][item][chunk][mt][lane = 16 g + m][8])

struct BSPlan {
  int mt, nt, ncob, nchunks;
  int bq[3], nb[3], hy, hx, pp, tiles, dmin[3];
  size_t lds;
  int wave_first[5], item_toff[27], item_cls[27], item_w[27];
};

bool bs_plan(const GatherGeom* g, int cnt, BSPlan& p) {
  static const bool off = getenv("URSN_BSCONV") && getenv("URSN_BSCONV")[0] == '0';
  if (off || cnt != 8) return false;
  const GatherGeom& g0 = g[0];
  if (g0.K < 32 || (g0.K & 31) || g0.Nn < 16 || (g0.Nn & 15) || (g0.in_cs & 7) || (g0.out_cs & 3)) return false;
  int total_taps = 0;
  for (int j = 0; j < 3; ++j) {
    if (g0.out_d[j] != 2 * g0.in_d[j]) return false;   // even fine extents: every class iterates the whole coarse grid
    p.dmin[j] = 1 << 20;
  }
  for (int i = 0; i < cnt; ++i) {
    if (g[i].ntaps < 1 || g[i].K != g0.K || g[i].Nn != g0.Nn) return false;
    total_taps += g[i].ntaps;
    for (int j = 0; j < 3; ++j) {
      if (g[i].so[j] != 2 || g[i].si[j] != 1 || g[i].q_d[j] != g0.in_d[j] || g[i].po[j] < 0 || g[i].po[j] > 1) return false;
      for (int t = 0; t < g[i].ntaps; ++t)
        if (g[i].tap_d[t][j] < p.dmin[j]) p.dmin[j] = g[i].tap_d[t][j];
    }
  }
  if (total_taps != 27) return false;
  for (int i = 0; i < cnt; ++i)
    for (int t = 0; t < g[i].ntaps; ++t)
      for (int j = 0; j < 3; ++j)
        if (g[i].tap_d[t][j] - p.dmin[j] > 1) return false;
  const int Zc = g0.in_d[0], Yc = g0.in_d[1], Xc = g0.in_d[2];
  // one buffer resource per image, out-of-range markers at 2 GB (input) / 1 GB (output)
  if ((int64_t)Zc * Yc * Xc * g0.in_cs * 2 >= ((int64_t)1 << 31)) return false;
  if ((int64_t)8 * Zc * Yc * Xc * g0.out_cs * 2 >= (int64_t)0x40000000) return false;
  p.nchunks = g0.K / 32;
  p.mt = g0.Nn >= 64 ? 4 : g0.Nn / 16;
  if (p.mt == 3) p.mt = 4;
  p.ncob = (g0.Nn + 16 * p.mt - 1) / (16 * p.mt);
  p.nt = p.mt == 4 ? 4 : 8;
  static int force_nt = -1;
  if (force_nt < 0) { const char* e = getenv("URSN_BSCONV_NT"); force_nt = e ? atoi(e) : 0; }
  if (force_nt == 4 || (force_nt == 8 && p.mt < 4)) p.nt = force_nt;   // (<4,8> does not fit the register file)
  // small volumes: the 64-voxel box where 128-voxel boxes would leave CUs idle
  const int64_t cvox = (int64_t)g0.N * Zc * Yc * Xc;
  if (!force_nt && p.nt == 8 && cvox / 128 * p.ncob < 512) p.nt = 4;
  p.bq[0] = 4; p.bq[1] = 4; p.bq[2] = p.nt == 8 ? 8 : 4;
  for (int j = 0; j < 3; ++j) p.nb[j] = (g0.in_d[j] + p.bq[j] - 1) / p.bq[j];
  p.hy = p.bq[1] + 1; p.hx = p.bq[2] + 1;
  p.pp = ((p.bq[0] + 1) * p.hy * p.hx + 63) & ~63;
  if (p.pp > 256) return false;
  p.lds = (size_t)p.nchunks * 4 * p.pp * 16;
  if (p.lds > 64 * 1024) return false;
  const int64_t tiles = (int64_t)g0.N * p.nb[0] * p.nb[1] * p.nb[2];
  if (tiles > (1 << 24)) return false;
  p.tiles = (int)tiles;
  // deal the classes to the waves, largest first, each to the least loaded wave (8 | 4+2+1 | 4+2 | 4+2 taps)
  int order[8], load[4] = {0, 0, 0, 0}, owner[8];
  for (int i = 0; i < 8; ++i) order[i] = i;
  for (int i = 0; i < 8; ++i)
    for (int j = i + 1; j < 8; ++j)
      if (g[order[j]].ntaps > g[order[i]].ntaps) { const int t = order[i]; order[i] = order[j]; order[j] = t; }
  for (int i = 0; i < 8; ++i) {
    int w = 0;
    for (int k = 1; k < 4; ++k) if (load[k] < load[w]) w = k;
    owner[order[i]] = w;
    load[w] += g[order[i]].ntaps;
  }
  int it = 0;
  for (int w = 0; w < 4; ++w) {
    p.wave_first[w] = it;
    for (int i = 0; i < 8; ++i) {
      if (owner[i] != w) continue;
      const int cl = g[i].po[0] * 4 + g[i].po[1] * 2 + g[i].po[2];
      for (int t = 0; t < g[i].ntaps; ++t) {
        const int ez = g[i].tap_d[t][0] - p.dmin[0], ey = g[i].tap_d[t][1] - p.dmin[1], ex = g[i].tap_d[t][2] - p.dmin[2];
        p.item_toff[it] = ((ez * p.hy + ey) * p.hx + ex) * 16;
        p.item_cls[it] = cl | (t == g[i].ntaps - 1 ? 8 : 0);
        p.item_w[it] = g[i].tap_w[t];
        ++it;
      }
    }
  }
  p.wave_first[4] = it;
  return it == 27;
}

template <int MT, int NT, bool STATS, bool FAST = false>
int bs_launch(const BSPlan& p, const BSArgs& a, hipStream_t s) {
  auto kern = bsconv_kernel<MT, NT, STATS, FAST>;
  static size_t attr = 48 * 1024;
  if (p.lds > attr) {
    URSN_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds));
    attr = p.lds;
  }
  hipLaunchKernelGGL(kern, dim3(p.tiles, p.ncob), dim3(256), p.lds, s, a);
  URSN_HIP(hipGetLastError());
  return 0;
}

}  // namespace

bool bsconv_ok(const GatherGeom* g, int cnt) { BSPlan p; return bs_plan(g, cnt, p); }
int bsconv_grid_blocks(const GatherGeom* g, int cnt) { BSPlan p; return bs_plan(g, cnt, p) ? p.tiles : 0; }
size_t bsconv_pack_elems(const GatherGeom* g, int cnt) {
  BSPlan p;
  return bs_plan(g, cnt, p) ? (size_t)p.ncob * 27 * p.nchunks * p.mt * 512 + 8 : 0;
}
size_t bsconv_stats_scratch_doubles(const GatherGeom* g, int cnt) {
  BSPlan p;
  return bs_plan(g, cnt, p) ? (size_t)p.tiles * p.ncob * 128 : 0;
}

int launch_bsconv(const GatherGeom* g, int cnt, const bf16_t* in, const float* w, int Kw, int Nw, bf16_t* wpack, bf16_t* out,
                  double* stats_partial, int accumulate, hipStream_t s) {
  BSPlan p;
  URSN_REQUIRE(bs_plan(g, cnt, p), "bf16 stride-2 scatter pass (deep levels): unsupported geometry");
  URSN_REQUIRE(!(stats_partial && accumulate), "bf16 stride-2 scatter pass: moments belong to a forward pass, which overwrites");
  const GatherGeom& g0 = g[0];
  {
    BPackJob k = bpack_job(BPK_SCATTER);
    k.w = w; k.wp = wpack; k.Kw = Kw > 0 ? Kw : g0.K; k.Nw = Nw > 0 ? Nw : g0.Nn;
    k.w_tap_stride = g0.w_tap_stride; k.w_sk = g0.w_sk; k.w_sn = g0.w_sn;
    for (int t = 0; t < 27; ++t) k.tap[t] = p.item_w[t];
    k.p[0] = p.nchunks; k.p[1] = p.ncob; k.p[2] = p.mt;
    const int64_t total = (int64_t)p.ncob * 27 * p.nchunks * p.mt * 512;
    k.blocks = (int)(cdiv64(total, 256) < 4096 ? cdiv64(total, 256) : 4096);
    URSN_TRY(bpack_submit(k, s));
  }
  BSArgs a;
  a.in = in; a.wp = wpack; a.out = out; a.stats_partial = stats_partial;
  a.N = g0.N; a.Zc = g0.in_d[0]; a.Yc = g0.in_d[1]; a.Xc = g0.in_d[2];
  a.in_cs = g0.in_cs; a.out_cs = g0.out_cs; a.Cout = g0.Nn; a.nchunks = p.nchunks;
  for (int j = 0; j < 3; ++j) { a.bq[j] = p.bq[j]; a.nb[j] = p.nb[j]; a.dmin[j] = p.dmin[j]; }
  a.lbx = __builtin_ctz(p.bq[2]); a.lby = __builtin_ctz(p.bq[1]);
  a.hy = p.hy; a.hx = p.hx; a.pp = p.pp;
  a.accumulate = accumulate;
  for (int w2 = 0; w2 < 5; ++w2) a.wave_first[w2] = p.wave_first[w2];
  for (int t = 0; t < 27; ++t) { a.item_toff[t] = p.item_toff[t]; a.item_cls[t] = p.item_cls[t]; }
  int rc = 3;
  const bool st = stats_partial != nullptr;
#define BS(mt_, nt_, label)                                                                       \
  if (p.mt == mt_ && p.nt == nt_) {                                                               \
    ursn_note_kernel(label);                                                                      \
    rc = st ? bs_launch<mt_, nt_, true>(p, a, s) : bs_launch<mt_, nt_, false>(p, a, s);           \
  }
  BS(4, 4, "bsconv_bf16<4,4>") BS(2, 8, "bsconv_bf16<2,8>") BS(2, 4, "bsconv_bf16<2,4>")
  // (forward with moments: measured 1.23 ms against 0.14 on the streamed path -- not understood, not used)
  if (p.mt == 1 && p.nchunks == 1 && !st) {   // 32 contraction channels: every wave's weights (<= 8 items) stay in registers
    int most = 0;
    for (int w2 = 0; w2 < 4; ++w2) if (p.wave_first[w2 + 1] - p.wave_first[w2] > most) most = p.wave_first[w2 + 1] - p.wave_first[w2];
    URSN_REQUIRE(most <= 8, "bf16 stride-2 scatter pass: a wave owns more than 8 items");
    ursn_note_kernel(p.nt == 8 ? "bsconv_bf16<1,8>" : "bsconv_bf16<1,4>");
    if (p.nt == 8) rc = bs_launch<1, 8, false, true>(p, a, s);
    else rc = bs_launch<1, 4, false, true>(p, a, s);
  } else {
    BS(1, 8, "bsconv_bf16<1,8>") BS(1, 4, "bsconv_bf16<1,4>")
  }
#undef BS
  URSN_TRY(rc);
  return 0;
}

int bsconv_stats_finalize(const GatherGeom* g, int cnt, const double* partial, int64_t V, float eps, float* mean, float* rstd,
                          hipStream_t s) {
  BSPlan p;
  URSN_REQUIRE(bs_plan(g, cnt, p), "bf16 stride-2 scatter pass (deep levels): unsupported geometry");
  return launch_bn_stats_final_blocked(partial, p.tiles, g[0].Nn, 16 * p.mt, 64, (size_t)p.tiles * 128, V, eps, mean, rstd, s);
}
