// HBM-bound kernels of the path: batch-statistics BatchNorm (stats / apply / backward), residual
// join + ReLU, the fused softmax-CE/metrics head, TF-form Adam and small utilities (gfx950).
//
// Thread mapping shared by all per-voxel kernels: a block iteration covers VPB = 256/CP voxels,
// CP = next_pow2(C/VEC) channel slots per voxel; VEC = 4 (16-byte accesses) whenever every
// channel count / stride involved is a multiple of 4, else 1.  All per-channel sums are kept in
// double (these kernels are bandwidth-bound; fp64 adds are free) so one-pass variance is safe.
#include "ursn_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

static int next_pow2(int x) { int p = 1; while (p < x) p <<= 1; return p; }

template <int VEC> struct VT;
template <> struct VT<1> { typedef float T; };
template <> struct VT<4> { typedef f32x4 T; };

// Streaming tensors are read once per pass: non-temporal loads keep them from evicting the reusable lines (weights,
// halos of the concurrent conv kernels) -- measured bn_bwd 13.0 -> 11.7 ms, bn_act 4.5 -> 3.9 ms per cfg3 step
// (tools/ew_sweep.sh; non-temporal stores did not help).  Bit 0: loads, bit 1: stores.
#ifndef URSN_EW_NT
#define URSN_EW_NT 1
#endif
#ifndef URSN_EW_U
#define URSN_EW_U 2   // voxels (float4 per array) in flight per thread: 124-140 VGPRs in the BN-backward kernels (4: 196-228)
#endif
#ifndef URSN_EW_ACT_U
#define URSN_EW_ACT_U 4   // the same for bn_act (74 VGPRs at 2)
#endif
template <int VEC> __device__ inline typename VT<VEC>::T ldv(const float* p) {
#if URSN_EW_NT & 1
  return __builtin_nontemporal_load((const typename VT<VEC>::T*)p);
#else
  return *(const typename VT<VEC>::T*)p;
#endif
}
template <int VEC> __device__ inline void stv(float* p, typename VT<VEC>::T v) {
#if URSN_EW_NT & 2
  __builtin_nontemporal_store(v, (typename VT<VEC>::T*)p);
#else
  *(typename VT<VEC>::T*)p = v;
#endif
}
__device__ inline float elem(float v, int) { return v; }
__device__ inline float elem(f32x4 v, int j) { return v[j]; }
__device__ inline void setelem(float& v, int, float x) { v = x; }
__device__ inline void setelem(f32x4& v, int j, float x) { v[j] = x; }

struct Map {
  int CP, shift, VPB, grid;
};
static Map make_map(int64_t V, int C, int VEC) {
  Map m;
  int cq = C / VEC;
  m.CP = next_pow2(cq);
  m.shift = 0;
  while ((1 << m.shift) < m.CP) ++m.shift;
  m.VPB = 256 / m.CP;
  static const int cap = getenv("URSN_EW_GRID") ? atoi(getenv("URSN_EW_GRID")) : 512;   // two long-lived workgroups per CU (tools/lib_ab.sh)
  int64_t blocks = cdiv64(V, (int64_t)m.VPB * 8);
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  m.grid = (int)blocks;
  return m;
}
static bool vec4_ok(int C, std::initializer_list<int> strides, std::initializer_list<const void*> ptrs) {
  if (C % 4) return false;
  if (C / 4 > 256) return false;
  for (int s : strides) if (s % 4) return false;
  for (const void* p : ptrs) if (p && (((uintptr_t)p) & 15)) return false;
  return true;
}

// ------------------------------------------------------------------------------------------
// Block reduction of NS*VEC doubles per thread over the voxel slots of a block -> partial[block][NS][C]
// ------------------------------------------------------------------------------------------
template <int NS, int VEC>
__device__ inline void block_reduce_store(double (&acc)[NS][VEC], int CP, int C, double* partial_blk) {
  __shared__ double sm[256 * NS * VEC];
  const int tid = threadIdx.x;
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int j = 0; j < VEC; ++j) sm[(s * VEC + j) * 256 + tid] = acc[s][j];
  __syncthreads();
  for (int st = 128; st >= CP; st >>= 1) {
    if (tid < st) {
#pragma unroll
      for (int k = 0; k < NS * VEC; ++k) sm[k * 256 + tid] += sm[k * 256 + tid + st];
    }
    __syncthreads();
  }
  if (tid < CP) {
    int c = tid * VEC;
    if (c < C) {
#pragma unroll
      for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int j = 0; j < VEC; ++j) partial_blk[s * C + c + j] = sm[(s * VEC + j) * 256 + tid];
    }
  }
}

int reduce_nblocks(int64_t V, int C) {
  int VEC = (C % 4 == 0) ? 4 : 1;
  return make_map(V, C, VEC).grid;
}
size_t reduce_scratch_bytes(int64_t V, int C, int nsums) {
  // partials [grid][nsums][C] doubles + finals [nsums][C] doubles
  return ((size_t)reduce_nblocks(V, C) * nsums * C + (size_t)nsums * C) * sizeof(double) + 256;
}

// ------------------------------------------------------------------------------------------
// BN statistics
// ------------------------------------------------------------------------------------------
template <int VEC>
__global__ __launch_bounds__(256) void bn_stats_partial_kernel(const float* __restrict__ z, int zcs, int64_t V, int C,
                                                               int shift, double* __restrict__ partial) {
  const int CP = 1 << shift;
  const int VPB = 256 >> shift;
  const int c = (threadIdx.x & (CP - 1)) * VEC;
  const int vr = threadIdx.x >> shift;
  double acc[2][VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) acc[0][j] = acc[1][j] = 0.0;
  if (c < C) {
    for (int64_t v = (int64_t)blockIdx.x * VPB + vr; v < V; v += (int64_t)gridDim.x * VPB) {
      typename VT<VEC>::T x = ldv<VEC>(z + v * zcs + c);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        double d = (double)elem(x, j);
        acc[0][j] += d;
        acc[1][j] += d * d;
      }
    }
  }
  block_reduce_store<2, VEC>(acc, CP, C, partial + (size_t)blockIdx.x * 2 * C);
}

// One block per channel: 256 threads stride over the per-block partials, tree-reduce in LDS (double).
template <int NS>
__device__ inline void reduce_partials(const double* __restrict__ partial, int nblocks, int C, int c, double (&out)[NS]) {
  __shared__ double sm[NS][256];
  double acc[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) acc[k] = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += 256) {
#pragma unroll
    for (int k = 0; k < NS; ++k) acc[k] += partial[((size_t)b * NS + k) * C + c];  // C = channel stride of the partials
  }
#pragma unroll
  for (int k = 0; k < NS; ++k) sm[k][threadIdx.x] = acc[k];
  __syncthreads();
  for (int st = 128; st >= 1; st >>= 1) {
    if (threadIdx.x < st) {
#pragma unroll
      for (int k = 0; k < NS; ++k) sm[k][threadIdx.x] += sm[k][threadIdx.x + st];
    }
    __syncthreads();
  }
#pragma unroll
  for (int k = 0; k < NS; ++k) out[k] = sm[k][0];
}

// Partials come from the conv epilogues, every one of which sums around pivots before it goes to fp64: per-lane pivots in
// the implicit-GEMM / stride-2 / bf16 kernels (ursn_common.h: shifted one-pass moments), a wave-uniform pivot in SGPRs in
// the lane-per-voxel kernels tconv, tdeconv, pconv (wave_pivot.h).  The fp64 sums that arrive here therefore carry only the fp32
// rounding of centred values at any |mean| / std, and  E[z^2] - E[z]^2  in fp64 keeps 1e-16 * mean^2 / var: no second pass over z.
// (Up to round 2 the lane-per-voxel kernels wrote plain fp32 sums and this kernel re-walked z with one block per channel
// when a channel looked ill-conditioned -- which an all-zero event batch triggered on every layer, 1.9x the step time.)
__global__ __launch_bounds__(256) void bn_stats_final_kernel(const double* __restrict__ partial, int nblocks, int PC,
                                                             int64_t V, float eps, float* __restrict__ mean,
                                                             float* __restrict__ rstd, int CB, size_t blk_stride) {
  // CB > 0: the partials of channel block ct = c / CB start at partial + ct * blk_stride (kernels whose grid.y walks
  // blocks of produced channels): ONE launch finalises the whole layer (it took one per block: 174 instead of 58
  // launches per cfg3 step)
  const int c = blockIdx.x;
  int cc = c;
  if (CB > 0) {
    const int ct = c / CB;
    cc = c - ct * CB;
    partial += (size_t)ct * blk_stride;
  }
  double s[2];
  reduce_partials<2>(partial, nblocks, PC, cc, s);
  double mu = s[0] / (double)V;
  double var = s[1] / (double)V - mu * mu;   // every thread holds the same s[]
  if (threadIdx.x != 0) return;
  if (var < 0.0) var = 0.0;
  mean[c] = (float)mu;
  rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
}

int launch_bn_stats_final(const double* partial, int nblocks, int C, int PC, int64_t V, float eps, float* mean,
                          float* rstd, hipStream_t s) {
  hipLaunchKernelGGL(bn_stats_final_kernel, dim3(C), dim3(256), 0, s, partial, nblocks, PC, V, eps, mean, rstd, 0, (size_t)0);
  URSN_HIP(hipGetLastError());
  return 0;
}

int launch_bn_stats_final_blocked(const double* partial, int nblocks, int C, int CB, int PC, size_t blk_stride, int64_t V, float eps,
                                  float* mean, float* rstd, hipStream_t s) {
  hipLaunchKernelGGL(bn_stats_final_kernel, dim3(C), dim3(256), 0, s, partial, nblocks, PC, V, eps, mean, rstd, CB, blk_stride);
  URSN_HIP(hipGetLastError());
  return 0;
}

int launch_bn_stats(const float* z, int zcs, int64_t V, int C, float eps, float* mean, float* rstd, void* scratch,
                    hipStream_t s) {
  bool v4 = vec4_ok(C, {zcs}, {z});
  URSN_REQUIRE(v4 || C <= 256, "bn_stats: unsupported channel count %d", C);
  Map m = make_map(V, C, v4 ? 4 : 1);
  double* partial = (double*)scratch;
  if (v4) hipLaunchKernelGGL(bn_stats_partial_kernel<4>, dim3(m.grid), dim3(256), 0, s, z, zcs, V, C, m.shift, partial);
  else hipLaunchKernelGGL(bn_stats_partial_kernel<1>, dim3(m.grid), dim3(256), 0, s, z, zcs, V, C, m.shift, partial);
  URSN_HIP(hipGetLastError());
  hipLaunchKernelGGL(bn_stats_final_kernel, dim3(C), dim3(256), 0, s, (const double*)partial, m.grid, C, V, eps, mean,
                     rstd, 0, (size_t)0);   // bn_stats_partial sums in fp64
  URSN_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------
// BN apply (+ second BN'd branch | + residual) (+ ReLU)
// ------------------------------------------------------------------------------------------
template <int VEC>
__global__ __launch_bounds__(256) void bn_act_kernel(BnActArgs a, int shift) {
  const int CP = 1 << shift;
  const int VPB = 256 >> shift;
  const int c = (threadIdx.x & (CP - 1)) * VEC;
  const int vr = threadIdx.x >> shift;
  if (c >= a.C) return;
  float sc[VEC], sh[VEC], sc2[VEC], sh2[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    sc[j] = a.rstd[c + j];
    sh[j] = a.beta[c + j] - a.mean[c + j] * sc[j];
    if (a.z2) {
      sc2[j] = a.rstd2[c + j];
      sh2[j] = a.beta2[c + j] - a.mean2[c + j] * sc2[j];
    } else {
      sc2[j] = sh2[j] = 0.f;
    }
  }
  // 4 voxels per iteration, all loads issued before the first use (the struct pointers may alias, so the
  // compiler will not hoist them itself): keeps 4x the bytes in flight per wave
  constexpr int U = URSN_EW_ACT_U;
  const int64_t stride = (int64_t)gridDim.x * VPB;
  for (int64_t v0 = (int64_t)blockIdx.x * VPB + vr; v0 < a.V; v0 += stride * U) {
    typename VT<VEC>::T x[U], x2[U], r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t v = v0 + u * stride;
      if (v < a.V) {
        x[u] = ldv<VEC>(a.z + v * a.zcs + c);
        if (a.z2) x2[u] = ldv<VEC>(a.z2 + v * a.z2cs + c);
        if (a.res) r[u] = ldv<VEC>(a.res + v * a.rescs + c);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t v = v0 + u * stride;
      const bool ok = v < a.V;
      typename VT<VEC>::T y;
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        float t = 0.f;
        if (ok) {
          t = fmaf(elem(x[u], j), sc[j], sh[j]);
          if (a.z2) t += fmaf(elem(x2[u], j), sc2[j], sh2[j]);
          if (a.res) t += elem(r[u], j);
          if (a.relu) t = fmaxf(t, 0.f);
        }
        setelem(y, j, t);
      }
      if (ok) stv<VEC>(a.y + v * a.ycs + c, y);
      if (VEC == 4 && a.mask_out) {   // wave-uniform: the wave's 256 consecutive elements -> 4 ballots
        const int lane = threadIdx.x & 63;
        const int64_t grp = ((v - (lane >> shift)) * a.C) >> 8;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const unsigned long long b = __ballot(ok && elem(y, j) > 0.f);
          if (lane == 0 && ok) a.mask_out[grp * 4 + j] = b;
        }
      }
    }
  }
}

int launch_bn_act(const BnActArgs& a, hipStream_t s) {
  bool v4 = vec4_ok(a.C, {a.zcs, a.ycs, a.z2 ? a.z2cs : 0, a.res ? a.rescs : 0}, {a.z, a.y, a.z2, a.res});
  URSN_REQUIRE(v4 || a.C <= 256, "bn_act: unsupported channel count %d", a.C);
  Map m = make_map(a.V, a.C, v4 ? 4 : 1);
  URSN_REQUIRE(!a.mask_out || (v4 && a.relu && bn_mask_ok(a.C)), "bn_act: relu bit mask needs the float4 path and C/4 a power of two");
  if (v4) hipLaunchKernelGGL(bn_act_kernel<4>, dim3(m.grid), dim3(256), 0, s, a, m.shift);
  else hipLaunchKernelGGL(bn_act_kernel<1>, dim3(m.grid), dim3(256), 0, s, a, m.shift);
  URSN_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------
// BN backward: reduce (sum g, sum g*xhat[, sum g*xhat2]) then apply
// ------------------------------------------------------------------------------------------
template <int VEC>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(BnBwdArgs a, int shift, double* __restrict__ partial) {
  const int CP = 1 << shift;
  const int VPB = 256 >> shift;
  const int c = (threadIdx.x & (CP - 1)) * VEC;
  const int vr = threadIdx.x >> shift;
  double acc[3][VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) acc[0][j] = acc[1][j] = acc[2][j] = 0.0;
  if (c < a.C) {
    float mu[VEC], rs[VEC], mu2[VEC], rs2[VEC], be[VEC];
    const bool bmask = VEC == 4 && a.relu && a.mask != nullptr;
    const bool ymask = !bmask && a.relu && a.y != nullptr, zmask = !bmask && a.relu && a.y == nullptr;
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      mu[j] = a.mean[c + j];
      rs[j] = a.rstd[c + j];
      mu2[j] = a.z2 ? a.mean2[c + j] : 0.f;
      rs2[j] = a.z2 ? a.rstd2[c + j] : 0.f;
      be[j] = zmask ? a.beta[c + j] - mu[j] * rs[j] : 0.f;   // shift of the forward pass
    }
    constexpr int U = URSN_EW_U;
    const int64_t stride = (int64_t)gridDim.x * VPB;
    for (int64_t v0 = (int64_t)blockIdx.x * VPB + vr; v0 < a.V; v0 += stride * U) {
      typename VT<VEC>::T gv[U], xv[U], yv[U], x2v[U];
      unsigned long long mw[U][VEC];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t v = v0 + u * stride;
        if (v < a.V) {
          gv[u] = ldv<VEC>(a.dy + v * a.dycs + c);
          xv[u] = ldv<VEC>(a.z + v * a.zcs + c);
          if (ymask) yv[u] = ldv<VEC>(a.y + v * a.ycs + c);
          if (a.z2) x2v[u] = ldv<VEC>(a.z2 + v * a.z2cs + c);
          if (bmask) {
            const int64_t grp = ((v - (lane >> shift)) * a.C) >> 8;
            const ulonglong2* mp = (const ulonglong2*)(a.mask + grp * 4);   // two 16-byte wave-uniform loads
            const ulonglong2 m0 = mp[0], m1 = mp[1];
            mw[u][0] = m0.x; mw[u][1] = m0.y; mw[u][VEC > 2 ? 2 : 0] = m1.x; mw[u][VEC > 3 ? 3 : 0] = m1.y;
          }
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (v0 + u * stride >= a.V) continue;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          float gj = elem(gv[u], j);
          const float xh0 = (elem(xv[u], j) - mu[j]) * rs[j];
          if (ymask && !(elem(yv[u], j) > 0.f)) gj = 0.f;
          if (zmask && !(fmaf(elem(xv[u], j), rs[j], be[j]) > 0.f)) gj = 0.f;   // same expression as bn_act
          if (bmask && !((mw[u][j] >> lane) & 1ull)) gj = 0.f;
          double gd = (double)gj;
          acc[0][j] += gd;
          acc[1][j] += gd * (double)xh0;
          if (a.z2) acc[2][j] += gd * (double)((elem(x2v[u], j) - mu2[j]) * rs2[j]);
        }
      }
    }
  }
  block_reduce_store<3, VEC>(acc, CP, a.C, partial + (size_t)blockIdx.x * 3 * a.C);
}

// finals: [3][C] doubles = mean(g), mean(g*xhat), mean(g*xhat2); dbeta(+2) += sum g
__global__ __launch_bounds__(256) void bn_bwd_final_kernel(const double* __restrict__ partial, int nblocks, int C,
                                                           int64_t V, double* __restrict__ finals,
                                                           float* __restrict__ dbeta, float* __restrict__ dbeta2, int Cw,
                                                           float* __restrict__ coef, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, const float* __restrict__ beta) {
  const int c = blockIdx.x;
  double s[3];
  reduce_partials<3>(partial, nblocks, C, c, s);
  if (threadIdx.x != 0) return;
  finals[c] = s[0] / (double)V;
  finals[C + c] = s[1] / (double)V;
  finals[2 * C + c] = s[2] / (double)V;
  if (coef) {   // dz = A g' + B (z - mu) + C for the data-gradient kernel that applies this BatchNorm backward on load
    const float r = rstd[c], mu = mean[c];
    const float mg = (float)finals[c], mgx = (float)finals[C + c];   // the apply kernel's fp32 constants
    coef[c] = r;
    coef[C + c] = -(r * r) * mgx;
    coef[2 * C + c] = -r * mg;
    coef[3 * C + c] = mu;
    coef[4 * C + c] = r;
    coef[5 * C + c] = beta ? beta[c] - mu * r : 0.f;   // bn_act's shift (same expression: the mask must be the forward's)
  }
  if (c >= Cw) return;   // padded channel of the logits layer: no parameter behind it
  if (dbeta) dbeta[c] += (float)s[0];
  if (dbeta2) dbeta2[c] += (float)s[0];
}

template <int VEC>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(BnBwdArgs a, int shift, const double* __restrict__ finals) {
  const int CP = 1 << shift;
  const int VPB = 256 >> shift;
  const int c = (threadIdx.x & (CP - 1)) * VEC;
  const int vr = threadIdx.x >> shift;
  if (c >= a.C) return;
  float mu[VEC], rs[VEC], mu2[VEC], rs2[VEC], mg[VEC], mgx[VEC], mgx2[VEC], be[VEC];
  const bool bmask = VEC == 4 && a.relu && a.mask != nullptr;
  const bool ymask = !bmask && a.relu && a.y != nullptr, zmask = !bmask && a.relu && a.y == nullptr;
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    mu[j] = a.mean[c + j];
    rs[j] = a.rstd[c + j];
    be[j] = zmask ? a.beta[c + j] - mu[j] * rs[j] : 0.f;
    mu2[j] = a.z2 ? a.mean2[c + j] : 0.f;
    rs2[j] = a.z2 ? a.rstd2[c + j] : 0.f;
    mg[j] = (float)finals[c + j];
    mgx[j] = (float)finals[a.C + c + j];
    mgx2[j] = (float)finals[2 * a.C + c + j];
  }
  constexpr int U = URSN_EW_U;
  const int64_t stride = (int64_t)gridDim.x * VPB;
  for (int64_t v0 = (int64_t)blockIdx.x * VPB + vr; v0 < a.V; v0 += stride * U) {
    typename VT<VEC>::T gv[U], xv[U], yv[U], x2v[U], drv[U];
    unsigned long long mw[U][VEC];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t v = v0 + u * stride;
      if (v < a.V) {
        gv[u] = ldv<VEC>(a.dy + v * a.dycs + c);
        xv[u] = ldv<VEC>(a.z + v * a.zcs + c);
        if (ymask) yv[u] = ldv<VEC>(a.y + v * a.ycs + c);
        if (a.z2) x2v[u] = ldv<VEC>(a.z2 + v * a.z2cs + c);
        if (a.dres && a.dres_accumulate) drv[u] = ldv<VEC>(a.dres + v * a.drescs + c);
        if (bmask) {
          const int64_t grp = ((v - (lane >> shift)) * a.C) >> 8;
          const ulonglong2* mp = (const ulonglong2*)(a.mask + grp * 4);
          const ulonglong2 m0 = mp[0], m1 = mp[1];
          mw[u][0] = m0.x; mw[u][1] = m0.y; mw[u][VEC > 2 ? 2 : 0] = m1.x; mw[u][VEC > 3 ? 3 : 0] = m1.y;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t v = v0 + u * stride;
      if (v >= a.V) continue;
      typename VT<VEC>::T dz, dz2, dr = drv[u];
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        float gj = elem(gv[u], j);
        float xh = (elem(xv[u], j) - mu[j]) * rs[j];
        if (ymask && !(elem(yv[u], j) > 0.f)) gj = 0.f;
        if (zmask && !(fmaf(elem(xv[u], j), rs[j], be[j]) > 0.f)) gj = 0.f;
        if (bmask && !((mw[u][j] >> lane) & 1ull)) gj = 0.f;
        setelem(dz, j, rs[j] * (gj - mg[j] - xh * mgx[j]));
        if (a.z2) {
          float xh2 = (elem(x2v[u], j) - mu2[j]) * rs2[j];
          setelem(dz2, j, rs2[j] * (gj - mg[j] - xh2 * mgx2[j]));
        }
        if (a.dres) setelem(dr, j, a.dres_accumulate ? elem(dr, j) + gj : gj);
      }
      stv<VEC>(a.dz + v * a.dzcs + c, dz);
      if (a.z2) stv<VEC>(a.dz2 + v * a.dz2cs + c, dz2);
      if (a.dres) stv<VEC>(a.dres + v * a.drescs + c, dr);
    }
  }
}

int launch_bn_bwd_final(const double* partial, int nblocks, int C, int64_t V, double* finals, float* dbeta, float* dbeta2,
                        int Cw, hipStream_t s, float* coef_out, const float* mean, const float* rstd, const float* beta) {
  hipLaunchKernelGGL(bn_bwd_final_kernel, dim3(C), dim3(256), 0, s, partial, nblocks, C, V, finals, dbeta, dbeta2, Cw, coef_out, mean, rstd,
                     beta);
  URSN_HIP(hipGetLastError());
  return 0;
}

int launch_bn_bwd(const BnBwdArgs& a, hipStream_t s) {
  URSN_REQUIRE(!a.relu || a.y || a.beta || a.mask, "bn_bwd: relu mask needs y, beta or the bit mask");
  bool v4 = vec4_ok(a.C, {a.dycs, (a.relu && a.y && !a.mask) ? a.ycs : 0, a.zcs, a.dzcs, a.z2 ? a.z2cs : 0, a.z2 ? a.dz2cs : 0,
                          a.dres ? a.drescs : 0},
                    {a.dy, a.relu ? a.y : nullptr, a.z, a.dz, a.z2, a.dz2, a.dres});
  URSN_REQUIRE(v4 || a.C <= 256, "bn_bwd: unsupported channel count %d", a.C);
  URSN_REQUIRE(!a.mask || (v4 && bn_mask_ok(a.C)), "bn_bwd: relu bit mask needs the float4 path and C/4 a power of two");
  Map m = make_map(a.V, a.C, v4 ? 4 : 1);
  double* partial = (double*)a.scratch;
  double* finals = partial + (size_t)m.grid * 3 * a.C;
  int nblocks = m.grid;
  if (a.pre_partial) {   // sum g, sum g xhat(, sum g xhat2) came out of the epilogue of the kernel that produced dy
    partial = const_cast<double*>(a.pre_partial);
    nblocks = a.pre_nblocks;
  } else {
    if (v4) hipLaunchKernelGGL(bn_bwd_reduce_kernel<4>, dim3(m.grid), dim3(256), 0, s, a, m.shift, partial);
    else hipLaunchKernelGGL(bn_bwd_reduce_kernel<1>, dim3(m.grid), dim3(256), 0, s, a, m.shift, partial);
    URSN_HIP(hipGetLastError());
  }
  if (a.coef_out) URSN_REQUIRE(!a.z2 && !a.dres && !a.mask && !(a.relu && a.y), "bn_bwd: apply-on-load coefficients need a single BatchNorm, no residual share and the bn(z) > 0 mask");
  hipLaunchKernelGGL(bn_bwd_final_kernel, dim3(a.C), dim3(256), 0, s, (const double*)partial, nblocks, a.C, a.V, finals,
                     a.dbeta, a.z2 ? a.dbeta2 : nullptr, a.Cw > 0 ? a.Cw : a.C, a.coef_out, a.mean, a.rstd, a.relu ? a.beta : nullptr);
  URSN_HIP(hipGetLastError());
  if (a.coef_out) return 0;   // dz is formed (and stored) by the layer's data-gradient kernel
  if (v4) hipLaunchKernelGGL(bn_bwd_apply_kernel<4>, dim3(m.grid), dim3(256), 0, s, a, m.shift, (const double*)finals);
  else hipLaunchKernelGGL(bn_bwd_apply_kernel<1>, dim3(m.grid), dim3(256), 0, s, a, m.shift, (const double*)finals);
  URSN_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------
// Head: logits = bn(z); softmax; weighted CE; accuracies; dlogits        (lib/ssnet.py:57-71)
// ------------------------------------------------------------------------------------------
#define URSN_MAX_CLASS 8
__global__ __launch_bounds__(256) void head_kernel(HeadArgs a, double* __restrict__ partial) {
  const int64_t P = (int64_t)a.n * a.pix;
  double loss = 0.0;
  unsigned int n_ok = 0, n_nz = 0, n_ok_nz = 0;
  float sc[URSN_MAX_CLASS], sh[URSN_MAX_CLASS];
#pragma unroll
  for (int k = 0; k < URSN_MAX_CLASS; ++k) {
    sc[k] = 1.f;
    sh[k] = 0.f;
    if (k < a.ncls && a.mean) {
      sc[k] = a.rstd[k];
      sh[k] = a.beta[k] - a.mean[k] * sc[k];
    }
  }
  const float invn = 1.0f / (float)a.n;
  float bg[4] = {0.f, 0.f, 0.f, 0.f}, bgx[4] = {0.f, 0.f, 0.f, 0.f};   // per-thread sums of a few hundred terms at most
  float mu[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) mu[k] = (k < a.ncls && a.mean) ? a.mean[k] : 0.f;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (int64_t)gridDim.x * blockDim.x) {
    float z[URSN_MAX_CLASS];
    float m = -INFINITY;
    int arg = 0;
    const bool v4 = a.z_cs == 4 && a.ncls <= 4;   // 3|4 classes in 4-padded logits: one 16-byte load
    f32x4 zv = {0.f, 0.f, 0.f, 0.f};
    if (v4) zv = __builtin_nontemporal_load((const f32x4*)(a.z + p * 4));
#pragma unroll
    for (int k = 0; k < URSN_MAX_CLASS; ++k) {
      if (k < a.ncls) {
        const float raw = (v4 && k < 4) ? zv[k & 3] : a.z[p * a.z_cs + k];
        z[k] = fmaf(raw, sc[k], sh[k]);
        if (z[k] > m) { m = z[k]; arg = k; }  // strict '>' keeps the lowest index on ties
      }
    }
    float ssum = 0.f;
    float e[URSN_MAX_CLASS];
#pragma unroll
    for (int k = 0; k < URSN_MAX_CLASS; ++k)
      if (k < a.ncls) { e[k] = expf(z[k] - m); ssum += e[k]; }
    float inv = 1.0f / ssum;
    if (a.softmax_out) {
#pragma unroll
      for (int k = 0; k < URSN_MAX_CLASS; ++k)
        if (k < a.ncls) a.softmax_out[p * a.ncls + k] = e[k] * inv;
    }
    if (a.ana_out) {  // (shower > track)*1 + (track >= shower)*2, masked by data > 1.0 (lib/ssnet_trainval.py:285-287)
      float shower = e[1] * inv, track = e[2] * inv;
      float lab = (shower > track ? 1.f : 0.f) + (track >= shower ? 2.f : 0.f);
      a.ana_out[p] = (a.data && a.data[p * a.data_cs] > 1.0f) ? lab : 0.f;
    }
    if (a.label) {
      int lab = (int)a.label[p];  // tf.cast(float -> int64) truncates toward zero
      bool lab_ok = lab >= 0 && lab < a.ncls;
      int labc = lab_ok ? lab : 0;
      float w = a.weight ? a.weight[p] : 1.0f;
      float zl = 0.f;
#pragma unroll
      for (int k = 0; k < URSN_MAX_CLASS; ++k)
        if (k == labc) zl = z[k];
      float ce = (m + logf(ssum)) - zl;
      loss += lab_ok ? (double)(w * ce) : (double)NAN;
      bool okp = (arg == lab);
      bool nz = a.data ? (a.data[p * a.data_cs] > 0.f) : false;
      n_ok += okp;
      n_nz += nz;
      n_ok_nz += (okp && nz);
      if (a.dlogits) {
        const int dcs = a.dl_cs > 0 ? a.dl_cs : a.ncls;
        if (dcs == 4 && a.ncls <= 4) {
          f32x4 dv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (k < a.ncls) dv[k] = w * invn * (e[k] * inv - (k == labc ? 1.f : 0.f));
          *(f32x4*)(a.dlogits + p * 4) = dv;
          if (a.bs_partial) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
              if (k < a.ncls) { bg[k] += dv[k]; bgx[k] = fmaf(dv[k], (zv[k] - mu[k]) * sc[k], bgx[k]); }
          }
        } else {
#pragma unroll
          for (int k = 0; k < URSN_MAX_CLASS; ++k)
            if (k < a.ncls) a.dlogits[p * dcs + k] = w * invn * (e[k] * inv - (k == labc ? 1.f : 0.f));
        }
      }
    }
  }
  __shared__ double sm[4][256];
  sm[0][threadIdx.x] = loss;
  sm[1][threadIdx.x] = (double)n_ok;
  sm[2][threadIdx.x] = (double)n_nz;
  sm[3][threadIdx.x] = (double)n_ok_nz;
  __syncthreads();
  for (int st = 128; st >= 1; st >>= 1) {
    if (threadIdx.x < st)
      for (int k = 0; k < 4; ++k) sm[k][threadIdx.x] += sm[k][threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x < 4) partial[(size_t)blockIdx.x * 4 + threadIdx.x] = sm[threadIdx.x][0];
  if (a.bs_partial) {   // uniform: two more rounds of the same tree for sum g, sum g xhat
    for (int r = 0; r < 2; ++r) {
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 4; ++k) sm[k][threadIdx.x] = (double)(r == 0 ? bg[k] : bgx[k]);
      __syncthreads();
      for (int st = 128; st >= 1; st >>= 1) {
        if (threadIdx.x < st)
          for (int k = 0; k < 4; ++k) sm[k][threadIdx.x] += sm[k][threadIdx.x + st];
        __syncthreads();
      }
      if (threadIdx.x < 4) a.bs_partial[(size_t)blockIdx.x * 12 + 4 * r + threadIdx.x] = sm[threadIdx.x][0];
    }
    if (threadIdx.x < 4) a.bs_partial[(size_t)blockIdx.x * 12 + 8 + threadIdx.x] = 0.0;
  }
}

__global__ __launch_bounds__(256) void head_final_kernel(const double* __restrict__ partial, int nblocks, int n, int64_t pix,
                                                         float* __restrict__ metrics) {
  // 256 threads, fixed assignment of partials to threads and a fixed tree: reproducible
  __shared__ double sm[4][256];
  double s[4] = {0, 0, 0, 0};
  for (int b = threadIdx.x; b < nblocks; b += 256)
    for (int k = 0; k < 4; ++k) s[k] += partial[(size_t)b * 4 + k];
  for (int k = 0; k < 4; ++k) sm[k][threadIdx.x] = s[k];
  __syncthreads();
  for (int st = 128; st >= 1; st >>= 1) {
    if (threadIdx.x < st)
      for (int k = 0; k < 4; ++k) sm[k][threadIdx.x] += sm[k][threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  metrics[0] = (float)(sm[0][0] / (double)n);
  metrics[1] = (float)(sm[1][0] / ((double)n * (double)pix));
  metrics[2] = sm[2][0] > 0 ? (float)(sm[3][0] / sm[2][0]) : NAN;
  metrics[3] = (float)sm[2][0];
}

int launch_head_final(const double* partial, int nblocks, int n, int64_t pix, float* metrics, hipStream_t s) {
  hipLaunchKernelGGL(head_final_kernel, dim3(1), dim3(256), 0, s, partial, nblocks, n, pix, metrics);
  URSN_HIP(hipGetLastError());
  return 0;
}

int head_blocks(int n, int64_t pix) {
  static const int cap = getenv("URSN_HEAD_GRID") ? atoi(getenv("URSN_HEAD_GRID")) : 4096;
  int64_t b = cdiv64((int64_t)n * pix, 256 * 4);
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}
size_t head_scratch_bytes(int n, int64_t pix) { return (size_t)head_blocks(n, pix) * 4 * sizeof(double) + 64; }

int launch_head(const HeadArgs& a, hipStream_t s) {
  URSN_REQUIRE(a.ncls >= 1 && a.ncls <= URSN_MAX_CLASS, "head: num_class %d not in [1,%d]", a.ncls, URSN_MAX_CLASS);
  int nb = head_blocks(a.n, a.pix);
  double* partial = (double*)a.scratch;
  hipLaunchKernelGGL(head_kernel, dim3(nb), dim3(256), 0, s, a, partial);
  URSN_HIP(hipGetLastError());
  hipLaunchKernelGGL(head_final_kernel, dim3(1), dim3(256), 0, s, (const double*)partial, nb, a.n, a.pix, a.metrics);
  URSN_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------
// Adam (TF form), fill, reduce
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n, float lr_t,
                                                   float b1, float b2, float eps) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float gi = g[i];
    float mi = b1 * m[i] + (1.f - b1) * gi;
    float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] -= lr_t * mi / (sqrtf(vi) + eps);
  }
}
int launch_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr_t, float b1, float b2, float eps,
                hipStream_t s) {
  if (n == 0) return 0;
  int blocks = (int)(cdiv64(n, 256) < 4096 ? cdiv64(n, 256) : 4096);
  hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, s, p, g, m, v, n, lr_t, b1, b2, eps);
  URSN_HIP(hipGetLastError());
  return 0;
}

__global__ void fill_kernel(float* p, float value, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = value;
}
int launch_fill(float* p, float value, int64_t n, hipStream_t s) {
  if (n == 0) return 0;
  int blocks = (int)(cdiv64(n, 256) < 4096 ? cdiv64(n, 256) : 4096);
  hipLaunchKernelGGL(fill_kernel, dim3(blocks), dim3(256), 0, s, p, value, n);
  URSN_HIP(hipGetLastError());
  return 0;
}

// dst[i] += sum_c src[c*n + i]; a block owns 64 elements, its 4 waves split the chunks (coalesced 256-B rows),
// chunk order inside a wave and the 4-way combine are fixed -> bitwise reproducible.
// cstride = distance (elements) between consecutive chunks.
__global__ __launch_bounds__(256) void reduce_accum_kernel(float* __restrict__ dst, const float* __restrict__ src,
                                                           int64_t n, int nchunks, int64_t cstride) {
  __shared__ float sm[4][64];
  const int e = threadIdx.x & 63, cg = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 64 + e;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (i < n) {
    int c = cg;
    for (; c + 12 < nchunks; c += 16) {
      s0 += src[(int64_t)c * cstride + i];
      s1 += src[(int64_t)(c + 4) * cstride + i];
      s2 += src[(int64_t)(c + 8) * cstride + i];
      s3 += src[(int64_t)(c + 12) * cstride + i];
    }
    for (; c < nchunks; c += 4) s0 += src[(int64_t)c * cstride + i];
  }
  sm[cg][e] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (cg == 0 && i < n) dst[i] += (sm[0][e] + sm[1][e]) + (sm[2][e] + sm[3][e]);
}

// Stage 1 of long slab reductions, in place: chunk g*gs <- sum of chunks [g*gs, min((g+1)*gs, nchunks)).  A block owns
// 64 elements of one group (grid = element blocks x groups): thousands of slabs are summed by ~1000 workgroups instead
// of n/64 (27 for a 27x8x8 weight gradient, which cost 50-200 us per launch).  Fixed order -> reproducible.
#define URSN_REDUCE_GS 64
__global__ __launch_bounds__(256) void reduce_groups_kernel(float* __restrict__ src, int64_t n, int nchunks) {
  __shared__ float sm[4][64];
  const int e = threadIdx.x & 63, cg = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 64 + e;
  const int c0 = blockIdx.y * URSN_REDUCE_GS;
  int c1 = c0 + URSN_REDUCE_GS;
  if (c1 > nchunks) c1 = nchunks;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (i < n) {
    int c = c0 + cg;
    for (; c + 12 < c1; c += 16) {
      s0 += src[(int64_t)c * n + i];
      s1 += src[(int64_t)(c + 4) * n + i];
      s2 += src[(int64_t)(c + 8) * n + i];
      s3 += src[(int64_t)(c + 12) * n + i];
    }
    for (; c < c1; c += 4) s0 += src[(int64_t)c * n + i];
  }
  sm[cg][e] = (s0 + s1) + (s2 + s3);
  __syncthreads();   // all reads of chunk c0 (by wave 0) are done before it is overwritten
  if (cg == 0 && i < n) src[(int64_t)c0 * n + i] = (sm[0][e] + sm[1][e]) + (sm[2][e] + sm[3][e]);
}
// returns the number of chunks left (stride URSN_REDUCE_GS * n) after the optional in-place stage
static int reduce_stage1(float* src, int64_t n, int nchunks, int64_t& cstride, hipStream_t s) {
  cstride = n;
  if (nchunks < 4 * URSN_REDUCE_GS) return nchunks;
  const int groups = (nchunks + URSN_REDUCE_GS - 1) / URSN_REDUCE_GS;
  hipLaunchKernelGGL(reduce_groups_kernel, dim3((unsigned)cdiv64(n, 64), groups), dim3(256), 0, s, src, n, nchunks);
  cstride = (int64_t)URSN_REDUCE_GS * n;
  return groups;
}
// NOTE: src is scratch: long reductions overwrite the first chunk of every group.
int launch_reduce_accum(float* dst, const float* src, int64_t n, int nchunks, hipStream_t s) {
  if (n == 0) return 0;
  int64_t cstride;
  const int left = reduce_stage1(const_cast<float*>(src), n, nchunks, cstride, s);
  hipLaunchKernelGGL(reduce_accum_kernel, dim3((unsigned)cdiv64(n, 64)), dim3(256), 0, s, dst, src, n, left, cstride);
  URSN_HIP(hipGetLastError());
  return 0;
}

// dst[t*dst_tap_stride + r*dst_row_stride + c] += sum_k src[k*(taps*rows*cols) + (t*rows + r)*cols + c]
__global__ __launch_bounds__(256) void reduce_accum_blocked_kernel(float* __restrict__ dst, const float* __restrict__ src,
                                                                   int taps, int rows, int cols, int64_t dst_tap_stride,
                                                                   int dst_row_stride, int nchunks, int64_t cstride) {
  __shared__ float sm[4][64];
  const int e = threadIdx.x & 63, cg = threadIdx.x >> 6;
  const int64_t n = (int64_t)taps * rows * cols;
  const int64_t i = (int64_t)blockIdx.x * 64 + e;
  float s0 = 0.f, s1 = 0.f;
  if (i < n) {
    int c = cg;
    for (; c + 4 < nchunks; c += 8) {
      s0 += src[(int64_t)c * cstride + i];
      s1 += src[(int64_t)(c + 4) * cstride + i];
    }
    for (; c < nchunks; c += 4) s0 += src[(int64_t)c * cstride + i];
  }
  sm[cg][e] = s0 + s1;
  __syncthreads();
  if (cg == 0 && i < n) {
    int col = (int)(i % cols);
    int64_t tr = i / cols;
    int r = (int)(tr % rows);
    int t = (int)(tr / rows);
    dst[(int64_t)t * dst_tap_stride + (int64_t)r * dst_row_stride + col] += (sm[0][e] + sm[1][e]) + (sm[2][e] + sm[3][e]);
  }
}
int launch_reduce_accum_blocked(float* dst, const float* src, int taps, int rows, int cols, int64_t dst_tap_stride,
                                int dst_row_stride, int nchunks, hipStream_t s) {
  int64_t n = (int64_t)taps * rows * cols;
  int64_t cstride;
  const int left = reduce_stage1(const_cast<float*>(src), n, nchunks, cstride, s);
  hipLaunchKernelGGL(reduce_accum_blocked_kernel, dim3((unsigned)cdiv64(n, 64)), dim3(256), 0, s, dst, src, taps, rows,
                     cols, dst_tap_stride, dst_row_stride, left, cstride);
  URSN_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------
// MFMA lane-layout probe (tests/test_mfma_probe.py decodes it)
//   which 1: 4x4x1_16B, a = lane+1, b = 1          -> D = source A lane + 1
//   which 2: 4x4x1_16B, a = 1,      b = lane+1     -> D = source B lane + 1
//   which 3: as 1 with cbsz=4, abid=3              (A block 3 broadcast to all 16 blocks)
//   which 4: as 1 with cbsz=2, abid=1              (A block 1 of every 4 broadcast in its group)
//   which 5: 16x16x4, a = 2^(lane>>4) * (1+(lane&15)), b = 1   (sum over k identifies lanes)
//   which 6: 16x16x4, a = 1, b = 2^(8*(lane>>4))... encoded per k: b = (1+(lane&15)) * 32^(lane>>4)
// out[lane*4 + r] = D reg r of lane.
// ------------------------------------------------------------------------------------------
__global__ void mfma_probe_kernel(int which, float* out) {
  int lane = threadIdx.x;
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  float a = 1.f, b = 1.f;
  f32x4 d = c;
  switch (which) {
    case 1: a = (float)(lane + 1); d = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0); break;
    case 2: b = (float)(lane + 1); d = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0); break;
    case 3: a = (float)(lane + 1); d = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, 3, 0); break;
    case 4: a = (float)(lane + 1); d = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 2, 1, 0); break;
    case 5: a = (float)((1 + (lane & 15)) * (1 << (5 * (lane >> 4)))); d = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); break;
    case 6: b = (float)((1 + (lane & 15)) * (1 << (5 * (lane >> 4)))); d = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); break;
    default: break;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) out[lane * 4 + r] = d[r];
}
int launch_mfma_probe(int which, float* out, hipStream_t s) {
  hipLaunchKernelGGL(mfma_probe_kernel, dim3(1), dim3(64), 0, s, which, out);
  URSN_HIP(hipGetLastError());
  return 0;
}
