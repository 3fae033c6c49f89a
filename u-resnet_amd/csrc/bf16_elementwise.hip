// HBM-bound kernels of the bf16 path: BatchNorm apply / join / backward, the softmax-CE head and the input / dtype
// conversions.  Same arithmetic as elementwise.hip (fp32 per element, fp64 per-channel sums); tensors are bf16 with channel
// counts that are multiples of 8, so a thread moves one 16-byte piece (8 channels of one voxel) per access.
// Thread mapping: CP = next_pow2(C / 8) channel pieces per voxel, 256 / CP voxels per block iteration, a thread keeps its
// channel piece for the whole kernel (per-channel parameters and sums live in registers).
#include <stdlib.h>

#include "bf16_common.h"

// voxels (16-byte pieces per tensor) in flight per thread.  Round 4, A/B on one box at cfg5: 2 -> 4 takes bbn_bwd 11.65 -> 11.41 and
// bbn_act 3.55 -> 3.50 ms per step (101.2 -> 102.6 images/s); 8: 11.80 / 3.83 (100.2)
#ifndef URSN_BEW_U
#define URSN_BEW_U 4
#endif

namespace {

int next_pow2(int x) { int p = 1; while (p < x) p <<= 1; return p; }

struct BMap { int shift, grid; };
BMap make_bmap(int64_t V, int C) {
  BMap m;
  const int cp = next_pow2(C / 8);
  m.shift = 0;
  while ((1 << m.shift) < cp) ++m.shift;
  const int vpb = 256 >> m.shift;
  static const int cap = getenv("URSN_BEW_GRID") ? atoi(getenv("URSN_BEW_GRID")) : 1024;   // rows of reduce partials (2048: 79.1, 1024: 79.5 img/s at cfg5)
  int64_t blocks = cdiv64(V, (int64_t)vpb * 8);
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  m.grid = (int)blocks;
  return m;
}

__device__ __forceinline__ u32x4 ld16(const bf16_t* p) { return __builtin_nontemporal_load((const u32x4*)p); }
#ifndef URSN_BEW_NT_STORE
#define URSN_BEW_NT_STORE 1
#endif
__device__ __forceinline__ void st16(bf16_t* p, u32x4 v) {
#if URSN_BEW_NT_STORE
  __builtin_nontemporal_store(v, (u32x4*)p);
#else
  *(u32x4*)p = v;
#endif
}

// NS x 8 per-thread fp32 sums -> partial[block][NS][C] doubles.  Lanes that share a channel piece (lane & (CP - 1)) are summed
// with shuffles in fp64, then the four waves (CP <= 32) or the 256 / CP threads of a piece (CP >= 64) through LDS in a FIXED
// order: no atomics (an earlier form added the waves' sums with fp64 atomics for CP > 4 -- order-dependent in the last bit and
// one memset per launch), 24 KB of LDS (a [24][257]-double staging array once held the kernel to three workgroups per CU).
template <int NS>
__device__ inline void block_reduce_store8(const float (&acc)[NS][8], int CP, int C, double* partial_blk) {
  __shared__ double sm[4 * NS * 8 * 32];   // [wave][sum * 8 + j][piece < 32]  |  CP >= 64: [sum * 8 + j][thread] in two halves
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (CP <= 32) {
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        double v = (double)acc[s][j];
        for (int o = 32; o >= CP; o >>= 1) v += __shfl_xor(v, o);
        if (lane < CP) sm[(wave * NS * 8 + s * 8 + j) * 32 + lane] = v;   // lanes 0 .. CP-1 hold the wave's sum for piece = lane
      }
    __syncthreads();
    for (int e = tid; e < NS * 8 * CP; e += 256) {
      const int k = e / CP, piece = e - k * CP;   // k = s * 8 + j
      if (piece * 8 < C)
        partial_blk[(k >> 3) * C + piece * 8 + (k & 7)] = (sm[k * 32 + piece] + sm[(NS * 8 + k) * 32 + piece]) +
                                                          (sm[(2 * NS * 8 + k) * 32 + piece] + sm[(3 * NS * 8 + k) * 32 + piece]);
    }
  } else {   // CP = 64 | 128 | 256: a piece is held by 4 | 2 | 1 threads of the block; 12 of the NS * 8 sums per LDS pass
    const int piece = tid & (CP - 1), rep = 256 / CP;
    for (int k0 = 0; k0 < NS * 8; k0 += 12) {
      __syncthreads();
#pragma unroll
      for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int k = s * 8 + j;
          if (k >= k0 && k < k0 + 12) sm[(k - k0) * 256 + tid] = (double)acc[s][j];
        }
      __syncthreads();
      if (tid < CP && piece * 8 < C) {
        for (int k = k0; k < k0 + 12 && k < NS * 8; ++k) {
          double v = 0.0;
          for (int r = 0; r < rep; ++r) v += sm[(k - k0) * 256 + r * CP + piece];
          partial_blk[(k >> 3) * C + piece * 8 + (k & 7)] = v;
        }
      }
    }
  }
}

// ---- BN apply / join ------------------------------------------------------------------------------------------------
// Work unit: a chunk of U x (voxels per block pass) consecutive voxels; a full chunk runs without per-voxel bounds checks and
// all of its loads are issued before the first use.  The optional operands are template parameters (as runtime flags every
// one of them cost registers and a branch per voxel: 145 / 248 / 256 VGPRs, 2-3 waves per SIMD, 2.7-3.5 TB/s).
// C8: 8-channel tensors (spatial level 0: 70 % of the bytes), one piece per voxel, every thread on the same channels.
#define BEW_MAP                                                                                           \
  const int CP = C8 ? 1 : 1 << shift, VPB = C8 ? 256 : 256 >> shift;                                      \
  const int c = C8 ? 0 : (threadIdx.x & (CP - 1)) * 8, vr = C8 ? threadIdx.x : threadIdx.x >> shift;      \
  constexpr int U = URSN_BEW_U;                                                                           \
  const int64_t chunkv = (int64_t)VPB * U, nchunks = (a.V + chunkv - 1) / chunkv

template <bool C8, bool HAS2, bool HASR, bool CAT = false>
__global__ __launch_bounds__(256) void bbn_act_kernel(BBnActArgs a, int shift) {
  static_assert(!CAT || (C8 && HAS2 && !HASR), "concat form: two 8-channel BatchNorm outputs side by side");
  BEW_MAP;
  if (c >= a.C) return;
  float sc[8], sh[8], sc2[8], sh2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    sc[j] = a.rstd[c + j];
    sh[j] = a.beta[c + j] - a.mean[c + j] * sc[j];
    sc2[j] = HAS2 ? a.rstd2[c + j] : 0.f;
    sh2[j] = HAS2 ? a.beta2[c + j] - a.mean2[c + j] * sc2[j] : 0.f;
  }
  const bool relu = a.relu != 0;
  for (int64_t ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
    const int64_t vb = ch * chunkv + vr;
    auto body = [&](auto FULL) {
      constexpr bool full = decltype(FULL)::value;
      u32x4 x[U], x2[U], r[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t v = vb + (int64_t)u * VPB;
        if (full || v < a.V) {
          x[u] = ld16(a.z + v * a.zcs + c);
          if constexpr (HAS2) x2[u] = ld16(a.z2 + v * a.z2cs + c);
          if constexpr (HASR) r[u] = ld16(a.res + v * a.rescs + c);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t v = vb + (int64_t)u * VPB;
        if (!full && v >= a.V) continue;
        float f[8], f2[8], fr[8], y[8];
        unpack8(x[u], f);
        if constexpr (HAS2) unpack8(x2[u], f2);
        if constexpr (HASR) unpack8(r[u], fr);
        if constexpr (CAT) {
          float y2[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            float t = fmaf(f[j], sc[j], sh[j]), t2 = fmaf(f2[j], sc2[j], sh2[j]);
            if (relu) { t = fmaxf(t, 0.f); t2 = fmaxf(t2, 0.f); }
            y[j] = t; y2[j] = t2;
          }
          st16(a.y + v * a.ycs, pack8(y));
          st16(a.y + v * a.ycs + 8, pack8(y2));
          continue;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float t = fmaf(f[j], sc[j], sh[j]);
          if constexpr (HAS2) t += fmaf(f2[j], sc2[j], sh2[j]);
          if constexpr (HASR) t += fr[j];
          if (relu) t = fmaxf(t, 0.f);
          y[j] = t;
        }
        st16(a.y + v * a.ycs + c, pack8(y));
        if (a.mask_out) {
          unsigned bits = 0;
#pragma unroll
          for (int j = 0; j < 8; ++j) bits |= (y[j] > 0.f ? 1u : 0u) << j;
          a.mask_out[v * (a.C >> 3) + (c >> 3)] = (unsigned char)bits;
        }
      }
    };
    if ((ch + 1) * chunkv <= a.V) body(std::true_type{});
    else body(std::false_type{});
  }
}

// concat form (the level-0 skip, lib/uresnet.py:80-83): y[v] = [act(bn(z[v])) | act(bn2(z2[v]))], 16 channels per voxel.  Lane pairs
// own a voxel -- the even lane the first piece from z, the odd lane the second from z2 -- so that one store instruction writes
// whole 32-byte voxels back to back (with one thread per voxel each of its two 16-byte stores left 16-byte holes: 4.8 TB/s)
__global__ __launch_bounds__(256) void bbn_cat_kernel(BBnActArgs a) {
  const int pc = threadIdx.x & 1, vr = threadIdx.x >> 1;
  constexpr int U = URSN_BEW_U, VPB = 128;
  const int64_t chunkv = (int64_t)VPB * U, nchunks = (a.V + chunkv - 1) / chunkv;
  const bf16_t* src = pc ? a.z2 : a.z;
  const int scs = pc ? a.z2cs : a.zcs;
  float sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    sc[j] = pc ? a.rstd2[j] : a.rstd[j];
    sh[j] = pc ? a.beta2[j] - a.mean2[j] * sc[j] : a.beta[j] - a.mean[j] * sc[j];
  }
  const bool relu = a.relu != 0;
  for (int64_t ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
    const int64_t vb = ch * chunkv + vr;
    u32x4 x[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t v = vb + (int64_t)u * VPB;
      if (v < a.V) x[u] = ld16(src + v * scs);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t v = vb + (int64_t)u * VPB;
      if (v >= a.V) continue;
      float f[8], y[8];
      unpack8(x[u], f);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float t = fmaf(f[j], sc[j], sh[j]);
        y[j] = relu ? fmaxf(t, 0.f) : t;
      }
      st16(a.y + v * a.ycs + 8 * pc, pack8(y));
    }
  }
}

// ---- BN backward ----------------------------------------------------------------------------------------------------
// MASK: 0 no activation, 1 mask = y > 0 (y given), 2 mask = bn(z) > 0 (beta given), 3 mask bytes of the forward pass.
// Per-thread sums in fp32: a thread adds <= ~1e3 terms of bf16-rounded data (relative error ~1e-6 of its own partial sum);
// everything across threads and blocks is fp64.
template <bool C8, int MASK, bool HAS2, bool D2 = false>
__global__ __launch_bounds__(256) void bbn_bwd_reduce_kernel(BBnBwdArgs a, int shift, double* __restrict__ partial) {
  BEW_MAP;
  float acc[3][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[0][j] = acc[1][j] = acc[2][j] = 0.f;
  if (c < a.C) {
    float mu[8], rs[8], mu2[8], rs2[8], be[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      mu[j] = a.mean[c + j]; rs[j] = a.rstd[c + j];
      mu2[j] = HAS2 ? a.mean2[c + j] : 0.f; rs2[j] = HAS2 ? a.rstd2[c + j] : 0.f;
      be[j] = MASK == 2 ? a.beta[c + j] - mu[j] * rs[j] : 0.f;
    }
    for (int64_t ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
      const int64_t vb = ch * chunkv + vr;
      auto body = [&](auto FULL) {
        constexpr bool full = decltype(FULL)::value;
        u32x4 gp[U], zp[U], yp[U], z2p[U], g2p[U];
        unsigned mk[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int64_t v = vb + (int64_t)u * VPB;
          if (full || v < a.V) {
            gp[u] = ld16(a.dy + v * a.dycs + c);
            zp[u] = ld16(a.z + v * a.zcs + c);
            if constexpr (MASK == 1) yp[u] = ld16(a.y + v * a.ycs + c);
            if constexpr (MASK == 3) mk[u] = a.mask[v * (a.C >> 3) + (c >> 3)];
            if constexpr (HAS2) z2p[u] = ld16(a.z2 + v * a.z2cs + c);
            if constexpr (D2) g2p[u] = ld16(a.dy2 + v * a.dy2cs + c);
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (!full && vb + (int64_t)u * VPB >= a.V) continue;
          float g[8], z[8], y[8], z2[8];
          unpack8(gp[u], g); unpack8(zp[u], z);
          if constexpr (D2) {
            float g2[8];
            unpack8(g2p[u], g2);
#pragma unroll
            for (int j = 0; j < 8; ++j) g[j] += g2[j];
          }
          if constexpr (MASK == 1) unpack8(yp[u], y);
          if constexpr (HAS2) unpack8(z2p[u], z2);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            float gj = g[j];
            if constexpr (MASK == 1) { if (!(y[j] > 0.f)) gj = 0.f; }
            if constexpr (MASK == 2) { if (!(fmaf(z[j], rs[j], be[j]) > 0.f)) gj = 0.f; }
            if constexpr (MASK == 3) { if (!((mk[u] >> j) & 1u)) gj = 0.f; }
            acc[0][j] += gj;
            acc[1][j] = fmaf(gj, (z[j] - mu[j]) * rs[j], acc[1][j]);
            if constexpr (HAS2) acc[2][j] = fmaf(gj, (z2[j] - mu2[j]) * rs2[j], acc[2][j]);
          }
        }
      };
      if ((ch + 1) * chunkv <= a.V) body(std::true_type{});
      else body(std::false_type{});
    }
  }
  block_reduce_store8<3>(acc, CP, a.C, partial + (size_t)blockIdx.x * 3 * a.C);
}

// DRES: 0 none, 1 dres = / += g (the identity shortcut's share of the join gradient)
template <bool C8, int MASK, bool HAS2, bool DRES, bool D2 = false>
__global__ __launch_bounds__(256) void bbn_bwd_apply_kernel(BBnBwdArgs a, int shift, const double* __restrict__ finals) {
  BEW_MAP;
  if (c >= a.C) return;
  float mu[8], rs[8], mu2[8], rs2[8], mg[8], mgx[8], mgx2[8], be[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    mu[j] = a.mean[c + j]; rs[j] = a.rstd[c + j];
    be[j] = MASK == 2 ? a.beta[c + j] - mu[j] * rs[j] : 0.f;
    mu2[j] = HAS2 ? a.mean2[c + j] : 0.f; rs2[j] = HAS2 ? a.rstd2[c + j] : 0.f;
    mg[j] = (float)finals[c + j]; mgx[j] = (float)finals[a.C + c + j]; mgx2[j] = HAS2 ? (float)finals[2 * a.C + c + j] : 0.f;
  }
  const bool dacc = DRES && a.dres_accumulate;
  for (int64_t ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
    const int64_t vb = ch * chunkv + vr;
    auto body = [&](auto FULL) {
      constexpr bool full = decltype(FULL)::value;
      u32x4 gp[U], zp[U], yp[U], z2p[U], drp[U], g2p[U];
      unsigned mk[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t v = vb + (int64_t)u * VPB;
        if (full || v < a.V) {
          gp[u] = ld16(a.dy + v * a.dycs + c);
          if constexpr (MASK == 3) mk[u] = a.mask[v * (a.C >> 3) + (c >> 3)];
          if constexpr (D2) g2p[u] = ld16(a.dy2 + v * a.dy2cs + c);
          zp[u] = ld16(a.z + v * a.zcs + c);
          if constexpr (MASK == 1) yp[u] = ld16(a.y + v * a.ycs + c);
          if constexpr (HAS2) z2p[u] = ld16(a.z2 + v * a.z2cs + c);
          if constexpr (DRES) { if (dacc) drp[u] = ld16(a.dres + v * a.drescs + c); }
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t v = vb + (int64_t)u * VPB;
        if (!full && v >= a.V) continue;
        float g[8], z[8], y[8], z2[8], dr[8], dz[8], dz2[8];
        unpack8(gp[u], g); unpack8(zp[u], z);
        if constexpr (D2) {
          float g2[8];
          unpack8(g2p[u], g2);
#pragma unroll
          for (int j = 0; j < 8; ++j) g[j] += g2[j];
        }
        if constexpr (MASK == 1) unpack8(yp[u], y);
        if constexpr (HAS2) unpack8(z2p[u], z2);
        if constexpr (DRES) { if (dacc) unpack8(drp[u], dr); }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float gj = g[j];
          if constexpr (MASK == 1) { if (!(y[j] > 0.f)) gj = 0.f; }
          if constexpr (MASK == 2) { if (!(fmaf(z[j], rs[j], be[j]) > 0.f)) gj = 0.f; }
          if constexpr (MASK == 3) { if (!((mk[u] >> j) & 1u)) gj = 0.f; }
          dz[j] = rs[j] * (gj - mg[j] - (z[j] - mu[j]) * rs[j] * mgx[j]);
          if constexpr (HAS2) dz2[j] = rs2[j] * (gj - mg[j] - (z2[j] - mu2[j]) * rs2[j] * mgx2[j]);
          if constexpr (DRES) dr[j] = dacc ? dr[j] + gj : gj;
        }
        st16(a.dz + v * a.dzcs + c, pack8(dz));
        if constexpr (HAS2) st16(a.dz2 + v * a.dz2cs + c, pack8(dz2));
        if constexpr (DRES) st16(a.dres + v * a.drescs + c, pack8(dr));
      }
    };
    if ((ch + 1) * chunkv <= a.V) body(std::true_type{});
    else body(std::false_type{});
  }
}

// ---- head (lib/ssnet.py:57-71): logits = bn(z), 8-channel bf16 pieces ----------------------------------------------------------
__global__ __launch_bounds__(256) void bhead_kernel(BHeadArgs a, double* __restrict__ partial) {
  const int64_t P = (int64_t)a.n * a.pix;
  double loss = 0.0;
  unsigned n_ok = 0, n_nz = 0, n_ok_nz = 0;
  float sc[8], sh[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    sc[k] = k < a.ncls ? a.rstd[k] : 0.f;
    sh[k] = k < a.ncls ? a.beta[k] - a.mean[k] * sc[k] : 0.f;
  }
  const float invn = 1.0f / (float)a.n;
  float bg[8], bgx[8], mu[8];   // per-thread sums of at most ~100 terms: fp32
#pragma unroll
  for (int k = 0; k < 8; ++k) { bg[k] = bgx[k] = 0.f; mu[k] = k < a.ncls ? a.mean[k] : 0.f; }
  // two voxels per iteration, every load of both issued before the first use (one voxel in flight per thread left the kernel at
  // 4.3 TB/s: 28 bytes per lane against the ~60 KB per CU that bandwidth x latency asks for)
  auto voxel = [&](int64_t p, const u32x4 zq, const float labf, const float wf, const float df) {
    float raw[8], z[8], e[8];
    unpack8(zq, raw);
    float m = -INFINITY;
    int arg = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (k < a.ncls) {
        z[k] = fmaf(raw[k], sc[k], sh[k]);
        if (z[k] > m) { m = z[k]; arg = k; }   // strict '>' keeps the lowest index on ties
      }
    float ssum = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (k < a.ncls) { e[k] = __expf(z[k] - m); ssum += e[k]; }
    const float inv = 1.0f / ssum;
    if (a.softmax_out)
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (k < a.ncls) a.softmax_out[p * a.ncls + k] = e[k] * inv;
    if (a.ana_out) {   // lib/ssnet_trainval.py:285-287
      const float shower = e[1] * inv, track = e[2] * inv;
      const float lab = (shower > track ? 1.f : 0.f) + (track >= shower ? 2.f : 0.f);
      a.ana_out[p] = (a.data && df > 1.0f) ? lab : 0.f;
    }
    if (a.label) {
      const int lab = (int)labf;
      const bool lab_ok = lab >= 0 && lab < a.ncls;
      const int labc = lab_ok ? lab : 0;
      const float w = wf;
      float zl = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (k == labc) zl = z[k];
      const float ce = (m + __logf(ssum)) - zl;
      loss += lab_ok ? (double)(w * ce) : (double)NAN;
      const bool okp = (arg == lab), nz = a.data ? (df > 0.f) : false;
      n_ok += okp; n_nz += nz; n_ok_nz += (okp && nz);
      if (a.dlogits) {
        float d[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) d[k] = k < a.ncls ? w * invn * (e[k] * inv - (k == labc ? 1.f : 0.f)) : 0.f;
        const u32x4 pk = pack8(d);
        *(u32x4*)(a.dlogits + p * a.dl_cs) = pk;
        if (a.bs_partial) {
          float dr[8];
          unpack8(pk, dr);
#pragma unroll
          for (int k = 0; k < 8; ++k)
            if (k < a.ncls) { bg[k] += dr[k]; bgx[k] = fmaf(dr[k], (raw[k] - mu[k]) * sc[k], bgx[k]); }
        }
      }
    }
  };
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += 2 * stride) {
    const int64_t p1 = p + stride;
    const bool two = p1 < P;
    const u32x4 z0 = ld16(a.z + p * a.z_cs);
    u32x4 z1 = z0;
    if (two) z1 = ld16(a.z + p1 * a.z_cs);
    float l0 = 0.f, l1 = 0.f, w0 = 1.f, w1 = 1.f, d0 = 0.f, d1 = 0.f;
    if (a.label) { l0 = a.label[p]; if (two) l1 = a.label[p1]; }
    if (a.weight) { w0 = a.weight[p]; if (two) w1 = a.weight[p1]; }
    if (a.data) { d0 = a.data[p * a.data_cs]; if (two) d1 = a.data[p1 * a.data_cs]; }
    voxel(p, z0, l0, w0, d0);
    if (two) voxel(p1, z1, l1, w1, d1);
  }
  __shared__ double sm[4][256];
  sm[0][threadIdx.x] = loss; sm[1][threadIdx.x] = (double)n_ok; sm[2][threadIdx.x] = (double)n_nz; sm[3][threadIdx.x] = (double)n_ok_nz;
  __syncthreads();
  for (int st = 128; st >= 1; st >>= 1) {
    if (threadIdx.x < st)
      for (int k = 0; k < 4; ++k) sm[k][threadIdx.x] += sm[k][threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x < 4) partial[(size_t)blockIdx.x * 4 + threadIdx.x] = sm[threadIdx.x][0];
  if (a.bs_partial) {   // uniform
    __syncthreads();
    double (*bm)[256] = sm;   // 16 columns in four rounds of four
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int col = 4 * r + k;
        sm[k][threadIdx.x] = col < 8 ? (double)bg[col & 7] : (double)bgx[col & 7];
      }
      __syncthreads();
      for (int st = 128; st >= 1; st >>= 1) {
        if (threadIdx.x < st)
          for (int k = 0; k < 4; ++k) bm[k][threadIdx.x] += bm[k][threadIdx.x + st];
        __syncthreads();
      }
      if (threadIdx.x < 4) {
        const int col = 4 * r + threadIdx.x;
        a.bs_partial[(size_t)blockIdx.x * 24 + (col < 8 ? col : 8 + (col - 8))] = sm[threadIdx.x][0];
      }
      __syncthreads();
    }
    if (threadIdx.x < 8) a.bs_partial[(size_t)blockIdx.x * 24 + 16 + threadIdx.x] = 0.0;
  }
}

__global__ void bf16_input_kernel(const float* __restrict__ data, bf16_t* __restrict__ out, int64_t V) {
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < V; v += (int64_t)gridDim.x * blockDim.x) {
    u32x4 p = {(unsigned)f2bf(data[v]), 0u, 0u, 0u};
    *(u32x4*)(out + v * 8) = p;
  }
}
__global__ void f32_to_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = f2bf(src[i]);
}
__global__ void bf16_to_f32_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = bf2f(src[i]);
}

bool piece_ok(int C, std::initializer_list<int> strides, std::initializer_list<const void*> ptrs) {
  if (C % 8 || C / 8 > 256) return false;
  for (int s : strides) if (s % 8) return false;
  for (const void* p : ptrs) if (p && (((uintptr_t)p) & 15)) return false;
  return true;
}

}  // namespace

size_t bbn_scratch_bytes(int64_t V, int C) {
  return ((size_t)make_bmap(V, C).grid * 3 * C + (size_t)3 * C) * sizeof(double) + 256;
}

// grid: act / apply run one chunk per workgroup (the hardware scheduler overlaps the load and store phases of different
// workgroups); the reduce keeps a bounded number of partial-sum rows
static int bew_grid(int64_t V, int shift, int cap) {
  const int64_t chunkv = (int64_t)(256 >> shift) * URSN_BEW_U;
  int64_t n = cdiv64(V, chunkv);
  if (n > cap) n = cap;
  return (int)(n < 1 ? 1 : n);
}

int launch_bbn_act(const BBnActArgs& a, hipStream_t s) {
  URSN_REQUIRE(piece_ok(a.C, {a.zcs, a.ycs, a.z2 ? a.z2cs : 0, a.res ? a.rescs : 0}, {a.z, a.y, a.z2, a.res}),
               "bf16 bn_act: channels / strides must be multiples of 8 and pointers 16-byte aligned (C = %d)", a.C);
  const BMap m = make_bmap(a.V, a.C);
  static const int acap = getenv("URSN_BEW_AGRID") ? atoi(getenv("URSN_BEW_AGRID")) : (1 << 20);   // A/B
  const int grid = bew_grid(a.V, m.shift, acap);
#define BACT(c8, h2, hr) hipLaunchKernelGGL((bbn_act_kernel<c8, h2, hr>), dim3(grid), dim3(256), 0, s, a, m.shift)
#define BACT2(c8) do { if (a.z2 && a.res) BACT(c8, true, true); else if (a.z2) BACT(c8, true, false); \
                       else if (a.res) BACT(c8, false, true); else BACT(c8, false, false); } while (0)
  if (a.cat) {
    URSN_REQUIRE(a.C == 8 && a.z2 && !a.res && a.ycs >= 16, "bf16 bn_act: the concat form needs two 8-channel inputs and a 16-channel output voxel");
    static const bool pairs = !(getenv("URSN_BBN_CAT_PAIRS") && getenv("URSN_BBN_CAT_PAIRS")[0] == '0');
    if (pairs) hipLaunchKernelGGL(bbn_cat_kernel, dim3(bew_grid(a.V, 1, acap)), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((bbn_act_kernel<true, true, false, true>), dim3(grid), dim3(256), 0, s, a, m.shift);
  } else if (a.C == 8) BACT2(true); else BACT2(false);
#undef BACT2
#undef BACT
  URSN_HIP(hipGetLastError());
  return 0;
}

int launch_bbn_bwd(const BBnBwdArgs& a, hipStream_t s) {
  URSN_REQUIRE(!a.relu || a.y || a.beta || a.mask, "bf16 bn_bwd: relu mask needs y, beta or the mask bytes");
  URSN_REQUIRE(piece_ok(a.C, {a.dycs, (a.relu && a.y && !a.mask) ? a.ycs : 0, a.zcs, a.dzcs, a.z2 ? a.z2cs : 0, a.z2 ? a.dz2cs : 0, a.dres ? a.drescs : 0},
                        {a.dy, (a.relu && !a.mask) ? a.y : nullptr, a.z, a.dz, a.z2, a.dz2, a.dres}),
               "bf16 bn_bwd: channels / strides must be multiples of 8 and pointers 16-byte aligned (C = %d)", a.C);
  const BMap m = make_bmap(a.V, a.C);   // m.grid: rows of the partial-sum scratch (bbn_scratch_bytes)
  static const int acap = getenv("URSN_BEW_AGRID") ? atoi(getenv("URSN_BEW_AGRID")) : (1 << 20);   // A/B
  const int rgrid = bew_grid(a.V, m.shift, m.grid), agrid = bew_grid(a.V, m.shift, acap);
  double* partial = (double*)a.scratch;
  double* finals = partial + (size_t)m.grid * 3 * a.C;
  const int mask = !a.relu ? 0 : (a.mask ? 3 : (a.y ? 1 : 2));
#define BRED(c8, mk, h2) hipLaunchKernelGGL((bbn_bwd_reduce_kernel<c8, mk, h2>), dim3(rgrid), dim3(256), 0, s, a, m.shift, partial)
#define BRED2(c8, mk) do { if (a.z2) BRED(c8, mk, true); else BRED(c8, mk, false); } while (0)
#define BRED3(c8) do { if (mask == 0) BRED2(c8, 0); else if (mask == 1) BRED2(c8, 1); else if (mask == 3) BRED2(c8, 3); else BRED2(c8, 2); } while (0)
  const bool d2 = a.dy2 != nullptr;
  if (d2) URSN_REQUIRE(a.C == 8 && mask == 2 && !a.z2 && !a.dres && !a.pre_partial && (a.dy2cs & 7) == 0, "bf16 bn_bwd: a second gradient operand is supported for 8-channel conv-BN-ReLU layers only");
  if (a.pre_partial) { /* sums taken by the kernel that produced dy */ }
  else if (d2) hipLaunchKernelGGL((bbn_bwd_reduce_kernel<true, 2, false, true>), dim3(rgrid), dim3(256), 0, s, a, m.shift, partial);
  else if (a.C == 8) BRED3(true);
  else BRED3(false);
  URSN_HIP(hipGetLastError());
  URSN_TRY(launch_bn_bwd_final(a.pre_partial ? a.pre_partial : partial, a.pre_partial ? a.pre_nblocks : rgrid, a.C, a.V, finals, a.dbeta,
                               a.z2 ? a.dbeta2 : nullptr, a.Cw > 0 ? a.Cw : a.C, s));
#define BAPP(c8, mk, h2, dr) hipLaunchKernelGGL((bbn_bwd_apply_kernel<c8, mk, h2, dr>), dim3(agrid), dim3(256), 0, s, a, m.shift, (const double*)finals)
#define BAPP1(c8, mk, h2) do { if (a.dres) BAPP(c8, mk, h2, true); else BAPP(c8, mk, h2, false); } while (0)
#define BAPP2(c8, mk) do { if (a.z2) BAPP1(c8, mk, true); else BAPP1(c8, mk, false); } while (0)
#define BAPP3(c8) do { if (mask == 0) BAPP2(c8, 0); else if (mask == 1) BAPP2(c8, 1); else if (mask == 3) BAPP2(c8, 3); else BAPP2(c8, 2); } while (0)
  if (d2) hipLaunchKernelGGL((bbn_bwd_apply_kernel<true, 2, false, false, true>), dim3(agrid), dim3(256), 0, s, a, m.shift, (const double*)finals);
  else if (a.C == 8) BAPP3(true); else BAPP3(false);
#undef BAPP3
#undef BAPP2
#undef BAPP1
#undef BAPP
#undef BRED3
#undef BRED2
#undef BRED
  URSN_HIP(hipGetLastError());
  return 0;
}

int bhead_blocks(int n, int64_t pix) { return head_blocks(n, pix); }
int launch_bhead(const BHeadArgs& a, hipStream_t s) {
  URSN_REQUIRE(a.ncls >= 1 && a.ncls <= 8 && a.z_cs == 8 && (!a.dlogits || a.dl_cs == 8), "bf16 head: needs <= 8 classes in 8-channel pieces");
  const int nb = head_blocks(a.n, a.pix);
  double* partial = (double*)a.scratch;
  hipLaunchKernelGGL(bhead_kernel, dim3(nb), dim3(256), 0, s, a, partial);
  URSN_HIP(hipGetLastError());
  return launch_head_final(partial, nb, a.n, a.pix, a.metrics, s);
}

static int ew_blocks(int64_t n) { int64_t b = cdiv64(n, 256); return (int)(b < 4096 ? (b < 1 ? 1 : b) : 4096); }
int launch_bf16_input(const float* data, bf16_t* out, int64_t V, hipStream_t s) {
  hipLaunchKernelGGL(bf16_input_kernel, dim3(ew_blocks(V)), dim3(256), 0, s, data, out, V);
  URSN_HIP(hipGetLastError());
  return 0;
}
int launch_f32_to_bf16(const float* src, bf16_t* dst, int64_t n, hipStream_t s) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, src, dst, n);
  URSN_HIP(hipGetLastError());
  return 0;
}
int launch_bf16_to_f32(const bf16_t* src, float* dst, int64_t n, hipStream_t s) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(bf16_to_f32_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, src, dst, n);
  URSN_HIP(hipGetLastError());
  return 0;
}
