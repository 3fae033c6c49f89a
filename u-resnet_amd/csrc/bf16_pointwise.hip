// bf16 1x1 stride-1 convolution between 8 / 16 channel tensors -- the shortcut of a residual module at the finest level
// (lib/resnet_module.py:25-33), forward and data gradient.  One v_mfma_f32_32x32x16_bf16 per 32 voxels with the B operand
// loaded STRAIGHT from global memory: lane (voxel c = l & 31, channel half h = l >> 5) fetches the 16-byte piece h of voxel c,
// which is exactly B[k = 8 h + j][column c]; a wave instruction covers one contiguous kilobyte.  No LDS, no staging, no
// packing pass (the A operand is converted from the fp32 master weights by the lanes themselves).  The generic box kernel
// spent 1.9 ms on the 256^3 x 4 forward (16 -> 8); this one is bound by its 48 bytes per voxel.
#include <stdlib.h>

#include "bf16_common.h"

namespace {

typedef float pw_f32x16 __attribute__((ext_vector_type(16)));

struct PWArgs {
  const bf16_t* in;
  const float* w;
  bf16_t* out;
  double* stats_partial;   // [grid][2][16] doubles or null
  int64_t V;
  int in_cs, out_cs;
  int Kw, Nw, w_sk, w_sn;
  int accumulate;
};

template <int K, int NN, bool STATS>
__global__ __launch_bounds__(256) void bpw_kernel(PWArgs a) {
  constexpr int NCH = NN / 8, U = 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 31, h = lane >> 5;
  bfx8 A;
  {
    const int co = lane & 31;
    float wv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ci = 8 * h + j;
      wv[j] = (co < a.Nw && ci < a.Kw && ci < K) ? a.w[(size_t)ci * a.w_sk + (size_t)co * a.w_sn] : 0.f;
    }
    const u32x4 p = pack8(wv);
    A = __builtin_bit_cast(bfx8, p);
  }
  float piv[4 * NCH], s1[4 * NCH], s2[4 * NCH], nacc = 0.f;
#pragma unroll
  for (int k = 0; k < 4 * NCH; ++k) piv[k] = s1[k] = s2[k] = 0.f;
  const int64_t ngroups = (a.V + 31) >> 5;
  const int64_t wstride = (int64_t)gridDim.x * 4;
  for (int64_t g0 = (int64_t)blockIdx.x * 4 + wave; g0 < ngroups; g0 += wstride * U) {
    u32x4 bp[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t v = ((g0 + u * wstride) << 5) + c;
      bp[u] = (u32x4){0u, 0u, 0u, 0u};
      if (v < a.V && (K == 16 || h == 0)) bp[u] = __builtin_nontemporal_load((const u32x4*)(a.in + v * a.in_cs + 8 * h));
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t v = ((g0 + u * wstride) << 5) + c;
      if (((g0 + u * wstride) << 5) >= a.V) continue;   // wave-uniform
      pw_f32x16 cc;
#pragma unroll
      for (int i = 0; i < 16; ++i) cc[i] = 0.f;
      cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, __builtin_bit_cast(bfx8, bp[u]), cc, 0, 0, 0);
      if (v < a.V) {
        bf16_t* ob = a.out + v * a.out_cs + 4 * h;
#pragma unroll
        for (int cb = 0; cb < NCH; ++cb) {   // rows (r & 3) + 8 (r >> 2) + 4 h: registers 4 cb .. hold channels 8 cb + 4 h ..
          float x[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) x[i] = cc[4 * cb + i];
          u32x2* o = (u32x2*)(ob + 8 * cb);
          if (a.accumulate) {
            const u32x2 e = *o;
            x[0] += __uint_as_float(e[0] << 16); x[1] += __uint_as_float(e[0] & 0xffff0000u);
            x[2] += __uint_as_float(e[1] << 16); x[3] += __uint_as_float(e[1] & 0xffff0000u);
          }
          u32x2 pk;
          pk[0] = pack_bf2(x[0], x[1]);
          pk[1] = pack_bf2(x[2], x[3]);
          *o = pk;
          if constexpr (STATS) {
            const float rv[4] = {__uint_as_float(pk[0] << 16), __uint_as_float(pk[0] & 0xffff0000u),
                                 __uint_as_float(pk[1] << 16), __uint_as_float(pk[1] & 0xffff0000u)};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              if (nacc == 0.f) piv[4 * cb + k] = rv[k];
              ursn_sacc(piv[4 * cb + k], s1[4 * cb + k], s2[4 * cb + k], rv[k]);
            }
          }
        }
        if constexpr (STATS) nacc += 1.f;
      }
    }
  }
  if constexpr (STATS) {
    __shared__ double red[4][32];
#pragma unroll
    for (int k = 0; k < 4 * NCH; ++k) {
      double u, w2;
      ursn_sacc_final(piv[k], s1[k], s2[k], nacc, u, w2);
#pragma unroll
      for (int o = 16; o >= 1; o >>= 1) { u += __shfl_xor(u, o); w2 += __shfl_xor(w2, o); }
      if (c == 0) {
        const int ch = 8 * (k >> 2) + 4 * h + (k & 3);
        red[wave][ch] = u;
        red[wave][16 + ch] = w2;
      }
    }
    __syncthreads();
    if (tid < 32) {
      const int ch = tid & 15;
      double t = 0.0;
      if (ch < NN) t = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
      a.stats_partial[(size_t)blockIdx.x * 32 + tid] = t;
    }
  }
}

int pw_grid(const GatherGeom& g) {
  int64_t V = g.N;
  for (int j = 0; j < 3; ++j) V *= g.out_d[j];
  int64_t b = ((V + 31) / 32 + 15) / 16;   // a wave takes 4 groups of 32 voxels per pass
  if (b > 8192) b = 8192;
  return (int)(b < 1 ? 1 : b);
}

}  // namespace

bool bpw_ok(const GatherGeom& g) {
  static const bool off = getenv("URSN_BPW") && getenv("URSN_BPW")[0] == '0';
  if (off || g.ntaps != 1 || (g.in_cs & 7) || (g.out_cs & 7)) return false;
  if (!((g.K == 8 || g.K == 16) && (g.Nn == 8 || g.Nn == 16))) return false;
  for (int j = 0; j < 3; ++j)
    if (g.so[j] != 1 || g.si[j] != 1 || g.po[j] != 0 || g.tap_d[0][j] != 0 || g.in_d[j] != g.out_d[j] || g.q_d[j] != g.out_d[j]) return false;
  return true;
}
int bpw_grid_blocks(const GatherGeom& g) { return pw_grid(g); }

int launch_bpw(const GatherGeom& g, const bf16_t* in, const float* w, int Kw, int Nw, bf16_t* out, double* stats_partial,
               hipStream_t s) {
  URSN_REQUIRE(bpw_ok(g), "bf16 pointwise conv: unsupported geometry");
  PWArgs a;
  a.in = in; a.w = w + (size_t)g.tap_w[0] * g.w_tap_stride; a.out = out; a.stats_partial = stats_partial;
  a.V = g.N;
  for (int j = 0; j < 3; ++j) a.V *= g.out_d[j];
  a.in_cs = g.in_cs; a.out_cs = g.out_cs;
  a.Kw = Kw > 0 ? Kw : g.K; a.Nw = Nw > 0 ? Nw : g.Nn; a.w_sk = g.w_sk; a.w_sn = g.w_sn;
  a.accumulate = g.accumulate;
  const int grid = pw_grid(g);
  ursn_note_kernel("bpw_bf16");
#define PWGO(k_, n_)                                                                                        \
  if (g.K == k_ && g.Nn == n_) {                                                                            \
    if (stats_partial) hipLaunchKernelGGL((bpw_kernel<k_, n_, true>), dim3(grid), dim3(256), 0, s, a);      \
    else hipLaunchKernelGGL((bpw_kernel<k_, n_, false>), dim3(grid), dim3(256), 0, s, a);                   \
  }
  PWGO(8, 8) PWGO(16, 8) PWGO(8, 16) PWGO(16, 16)
#undef PWGO
  URSN_HIP(hipGetLastError());
  return 0;
}
