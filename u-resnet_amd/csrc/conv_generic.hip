// Generic gather-convolution kernels for gfx950 (any geometry, any channel count).
//
//  * gconv_naive / wgrad_naive : one thread per output element, used only by tests to bisect.
//  * gconv_mfma                : implicit GEMM on v_mfma_f32_16x16x4_f32 (exact fp32), operands
//                                gathered straight from global/L2, D[co][voxel] so that every lane
//                                ends with 4 consecutive output channels of one voxel (16-B stores).
//  * wgrad_mfma                : dW[(t,m)][n] = sum_q S[q*si+d_t][m] * C[q][n] on the same MFMA,
//                                4 waves split the voxel rows of a chunk, LDS reduce, per-chunk slabs,
//                                deterministic slab reduction (reduce_accum) into the gradient buffer.
//
// These are the "any shape" path (deep levels, strided / transposed layers).  The dominant
// small-channel full-resolution layers use the LDS-tiled kernels in conv_tiled.hip.
#include <stdlib.h>

#include "ursn_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------
// naive reference kernels
// ---------------------------------------------------------------------------------------------
__global__ void gconv_naive_kernel(GatherGeom g, const float* __restrict__ in, const float* __restrict__ w,
                                   float* __restrict__ out) {
  int64_t total = (int64_t)g.N * g.q_d[0] * g.q_d[1] * g.q_d[2] * g.Nn;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    int co = (int)(e % g.Nn);
    int64_t v = e / g.Nn;
    int q2 = (int)(v % g.q_d[2]); v /= g.q_d[2];
    int q1 = (int)(v % g.q_d[1]); v /= g.q_d[1];
    int q0 = (int)(v % g.q_d[0]);
    int n = (int)(v / g.q_d[0]);
    float acc = 0.f;
    for (int t = 0; t < g.ntaps; ++t) {
      int p0 = q0 * g.si[0] + g.tap_d[t][0];
      int p1 = q1 * g.si[1] + g.tap_d[t][1];
      int p2 = q2 * g.si[2] + g.tap_d[t][2];
      if (p0 < 0 || p0 >= g.in_d[0] || p1 < 0 || p1 >= g.in_d[1] || p2 < 0 || p2 >= g.in_d[2]) continue;
      const float* ip = in + ((((int64_t)n * g.in_d[0] + p0) * g.in_d[1] + p1) * g.in_d[2] + p2) * g.in_cs;
      const float* wp = w + (int64_t)g.tap_w[t] * g.w_tap_stride + (int64_t)co * g.w_sn;
      for (int k = 0; k < g.K; ++k) acc = fmaf(ip[k], wp[(int64_t)k * g.w_sk], acc);
    }
    int o0 = q0 * g.so[0] + g.po[0], o1 = q1 * g.so[1] + g.po[1], o2 = q2 * g.so[2] + g.po[2];
    float* op = out + ((((int64_t)n * g.out_d[0] + o0) * g.out_d[1] + o1) * g.out_d[2] + o2) * g.out_cs + co;
    *op = g.accumulate ? (*op + acc) : acc;
  }
}

int launch_gconv_naive(const GatherGeom& g, const float* in, const float* w, float* out, hipStream_t s) {
  ursn_note_kernel("gconv_naive");
  int64_t total = (int64_t)g.N * g.q_d[0] * g.q_d[1] * g.q_d[2] * g.Nn;
  if (total == 0) return 0;
  int blocks = (int)(cdiv64(total, 256) < 65536 ? cdiv64(total, 256) : 65536);
  hipLaunchKernelGGL(gconv_naive_kernel, dim3(blocks), dim3(256), 0, s, g, in, w, out);
  URSN_HIP(hipGetLastError());
  return 0;
}

__global__ void wgrad_naive_kernel(GatherGeom g, const float* __restrict__ S, const float* __restrict__ C,
                                   float* __restrict__ dw) {
  int64_t total = (int64_t)g.ntaps * g.K * g.Nn;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    int n_ = (int)(e % g.Nn);
    int m = (int)((e / g.Nn) % g.K);
    int t = (int)(e / ((int64_t)g.Nn * g.K));
    double acc = 0.0;
    for (int n = 0; n < g.N; ++n)
      for (int q0 = 0; q0 < g.q_d[0]; ++q0) {
        int p0 = q0 * g.si[0] + g.tap_d[t][0];
        if (p0 < 0 || p0 >= g.in_d[0]) continue;
        for (int q1 = 0; q1 < g.q_d[1]; ++q1) {
          int p1 = q1 * g.si[1] + g.tap_d[t][1];
          if (p1 < 0 || p1 >= g.in_d[1]) continue;
          for (int q2 = 0; q2 < g.q_d[2]; ++q2) {
            int p2 = q2 * g.si[2] + g.tap_d[t][2];
            if (p2 < 0 || p2 >= g.in_d[2]) continue;
            float sv = S[((((int64_t)n * g.in_d[0] + p0) * g.in_d[1] + p1) * g.in_d[2] + p2) * g.in_cs + m];
            float cv = C[((((int64_t)n * g.q_d[0] + q0) * g.q_d[1] + q1) * g.q_d[2] + q2) * g.out_cs + n_];
            acc += (double)sv * (double)cv;
          }
        }
      }
    dw[(int64_t)g.tap_w[t] * g.K * g.Nn + (int64_t)m * g.Nn + n_] += (float)acc;
  }
}

int launch_wgrad_naive(const GatherGeom& g, const float* S, const float* C, float* dw, hipStream_t s) {
  int64_t total = (int64_t)g.ntaps * g.K * g.Nn;
  hipLaunchKernelGGL(wgrad_naive_kernel, dim3((unsigned)cdiv64(total, 64)), dim3(64), 0, s, g, S, C, dw);
  URSN_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------------
// gconv_mfma: D[co][voxel] += W_t[ci][co]^T . in[voxel+t][ci]
//   A operand (16 co x 4 ci): lane l holds W[ci = k0 + (l>>4)][co = co0 + (l&15)]
//   B operand (4 ci x 16 vox): lane l holds in[voxel (l&15)][ci = k0 + (l>>4)]
//   D (16 co x 16 vox): lane l, reg r -> co = 4*(l>>4) + r, voxel = l&15
// BV voxel tiles x BN cout tiles per wave.  KS == 1: the 4 waves of a block take 4 voxel groups;
// KS == 4: the 4 waves split the (tap, k0) steps of ONE voxel group and reduce through LDS
// (small-spatial, large-channel layers where voxel parallelism alone cannot fill 256 CUs).
// ---------------------------------------------------------------------------------------------
template <int BV, int BN, int KS>
__global__ __launch_bounds__(256) void gconv_mfma_kernel(GatherGeom g, const float* __restrict__ in,
                                                         const float* __restrict__ w, float* __restrict__ out) {
  extern __shared__ float red[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int vl = lane & 15;
  const int kl = lane >> 4;
  const int64_t Q = (int64_t)g.N * g.q_d[0] * g.q_d[1] * g.q_d[2];
  const int64_t vt0 = (KS == 1) ? ((int64_t)blockIdx.x * 4 + wave) * BV : (int64_t)blockIdx.x * BV;
  const int co0 = blockIdx.y * (16 * BN);

  int qn[BV], q0[BV], q1[BV], q2[BV];
  bool qv[BV];
#pragma unroll
  for (int b = 0; b < BV; ++b) {
    int64_t v = (vt0 + b) * 16 + vl;
    qv[b] = v < Q;
    if (!qv[b]) v = 0;
    q2[b] = (int)(v % g.q_d[2]); v /= g.q_d[2];
    q1[b] = (int)(v % g.q_d[1]); v /= g.q_d[1];
    q0[b] = (int)(v % g.q_d[0]);
    qn[b] = (int)(v / g.q_d[0]);
  }

  f32x4 acc[BV][BN];
#pragma unroll
  for (int b = 0; b < BV; ++b)
#pragma unroll
    for (int c = 0; c < BN; ++c) acc[b][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int ksteps = (g.K + 3) >> 2;
  int t_begin = 0, t_end = g.ntaps, s_begin = 0, s_end = 0;
  if (KS != 1) {  // split the flattened (tap, kstep) range across the 4 waves
    int total = g.ntaps * ksteps;
    s_begin = (int)((int64_t)total * wave / 4);
    s_end = (int)((int64_t)total * (wave + 1) / 4);
    t_begin = s_begin / ksteps;
    t_end = (s_end + ksteps - 1) / ksteps;
  }

  for (int t = t_begin; t < t_end; ++t) {
    bool ok[BV];
    int64_t off[BV];
#pragma unroll
    for (int b = 0; b < BV; ++b) {
      int p0 = q0[b] * g.si[0] + g.tap_d[t][0];
      int p1 = q1[b] * g.si[1] + g.tap_d[t][1];
      int p2 = q2[b] * g.si[2] + g.tap_d[t][2];
      ok[b] = qv[b] && p0 >= 0 && p0 < g.in_d[0] && p1 >= 0 && p1 < g.in_d[1] && p2 >= 0 && p2 < g.in_d[2];
      off[b] = ((((int64_t)qn[b] * g.in_d[0] + p0) * g.in_d[1] + p1) * g.in_d[2] + p2) * g.in_cs;
    }
    const float* wt = w + (int64_t)g.tap_w[t] * g.w_tap_stride;
    int ks0 = 0, ks1 = ksteps;
    if (KS != 1) {
      ks0 = (t * ksteps < s_begin) ? s_begin - t * ksteps : 0;
      ks1 = ((t + 1) * ksteps > s_end) ? s_end - t * ksteps : ksteps;
    }
#pragma unroll 2
    for (int ks = ks0; ks < ks1; ++ks) {
      const int kk = ks * 4 + kl;
      const bool kok = kk < g.K;
      float a[BN], bb[BV];
#pragma unroll
      for (int c = 0; c < BN; ++c) {
        int co = co0 + c * 16 + vl;
        a[c] = (kok && co < g.Nn) ? wt[(int64_t)kk * g.w_sk + (int64_t)co * g.w_sn] : 0.f;
      }
#pragma unroll
      for (int b = 0; b < BV; ++b) bb[b] = (ok[b] && kok) ? in[off[b] + kk] : 0.f;
#pragma unroll
      for (int b = 0; b < BV; ++b)
#pragma unroll
        for (int c = 0; c < BN; ++c) acc[b][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c], bb[b], acc[b][c], 0, 0, 0);
    }
  }

  if (KS != 1) {  // reduce the 4 waves' partial tiles through LDS; wave 0 keeps the sum
    float* mine = red + ((size_t)wave * BV * BN * 64 + lane) * 4;
#pragma unroll
    for (int b = 0; b < BV; ++b)
#pragma unroll
      for (int c = 0; c < BN; ++c) *(f32x4*)(mine + (size_t)(b * BN + c) * 256) = acc[b][c];
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int b = 0; b < BV; ++b)
#pragma unroll
      for (int c = 0; c < BN; ++c)
        for (int ww = 1; ww < 4; ++ww)
          acc[b][c] += *(f32x4*)(red + ((size_t)ww * BV * BN * 64 + lane) * 4 + (size_t)(b * BN + c) * 256);
  }

  const bool vec_ok = ((g.Nn & 3) == 0) && ((g.out_cs & 3) == 0) && ((((uintptr_t)out) & 15) == 0);
#pragma unroll
  for (int b = 0; b < BV; ++b) {
    if (!qv[b]) continue;
    int o0 = q0[b] * g.so[0] + g.po[0], o1 = q1[b] * g.so[1] + g.po[1], o2 = q2[b] * g.so[2] + g.po[2];
    float* op = out + ((((int64_t)qn[b] * g.out_d[0] + o0) * g.out_d[1] + o1) * g.out_d[2] + o2) * g.out_cs;
#pragma unroll
    for (int c = 0; c < BN; ++c) {
      int co = co0 + c * 16 + kl * 4;
      if (co >= g.Nn) continue;
      f32x4 v = acc[b][c];
      if (vec_ok) {
        f32x4* p = (f32x4*)(op + co);
        if (g.accumulate) v += *p;
        *p = v;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (co + r < g.Nn) op[co + r] = g.accumulate ? (op[co + r] + v[r]) : v[r];
      }
    }
  }
}

template <int BV, int BN, int KS>
static int launch_gconv_t(const GatherGeom& g, const float* in, const float* w, float* out, hipStream_t s) {
  int64_t Q = (int64_t)g.N * g.q_d[0] * g.q_d[1] * g.q_d[2];
  int64_t vtiles = cdiv64(Q, 16);
  int64_t gx = (KS == 1) ? cdiv64(vtiles, 4 * BV) : cdiv64(vtiles, BV);
  int gy = (g.Nn + 16 * BN - 1) / (16 * BN);
  size_t lds = (KS == 1) ? 0 : (size_t)4 * BV * BN * 64 * 4 * sizeof(float);
  hipLaunchKernelGGL((gconv_mfma_kernel<BV, BN, KS>), dim3((unsigned)gx, (unsigned)gy), dim3(256), lds, s, g, in, w, out);
  URSN_HIP(hipGetLastError());
  return 0;
}

int launch_gconv_mfma(const GatherGeom& g, const float* in, const float* w, float* out, hipStream_t s) {
  int64_t Q = (int64_t)g.N * g.q_d[0] * g.q_d[1] * g.q_d[2];
  if (Q == 0) return 0;
  int bn = g.Nn > 32 ? 4 : (g.Nn > 16 ? 2 : 1);
  int64_t vtiles = cdiv64(Q, 16);
  {  // few voxel tiles (the 6^3 level: 54 tiles x 4 column blocks on 256 CUs): 32-column blocks double the workgroups and halve
     // each one's serial k loop -- 0.83 -> 0.62 ms per cfg3 step (16 columns: 0.68); URSN_GCONV_BN=4 restores the wide blocks
    static const int narrow = getenv("URSN_GCONV_BN") ? atoi(getenv("URSN_GCONV_BN")) : 2;
    if (narrow > 0 && narrow < bn && vtiles * ((g.Nn + 16 * bn - 1) / (16 * bn)) < 256) bn = narrow;
  }
  int gy = (g.Nn + 16 * bn - 1) / (16 * bn);
  int64_t blocks_ks1 = cdiv64(vtiles, 16) * gy;
  int steps = g.ntaps * ((g.K + 3) / 4);
  bool ksplit = blocks_ks1 < 512 && steps >= 16;
  if (!ksplit) {
    if (bn == 4) { ursn_note_kernel("gconv_mfma<4,4,1>"); return launch_gconv_t<4, 4, 1>(g, in, w, out, s); }
    if (bn == 2) { ursn_note_kernel("gconv_mfma<4,2,1>"); return launch_gconv_t<4, 2, 1>(g, in, w, out, s); }
    ursn_note_kernel("gconv_mfma<4,1,1>");
    return launch_gconv_t<4, 1, 1>(g, in, w, out, s);
  }
  if (bn == 4) {
    // more voxel tiles per wave = more reuse of every weight operand fetched through L1 (the k-split form is
    // L1-bound at one tile: 5 loads per 4 MFMAs); keep >= ~256 workgroups
    if (vtiles / 4 * gy >= 256) { ursn_note_kernel("gconv_mfma<4,4,4>"); return launch_gconv_t<4, 4, 4>(g, in, w, out, s); }
    if (vtiles / 2 * gy >= 200) { ursn_note_kernel("gconv_mfma<2,4,4>"); return launch_gconv_t<2, 4, 4>(g, in, w, out, s); }
    ursn_note_kernel("gconv_mfma<1,4,4>");
    return launch_gconv_t<1, 4, 4>(g, in, w, out, s);
  }
  if (bn == 2) { ursn_note_kernel("gconv_mfma<1,2,4>"); return launch_gconv_t<1, 2, 4>(g, in, w, out, s); }
  ursn_note_kernel("gconv_mfma<1,1,4>");
  return launch_gconv_t<1, 1, 4>(g, in, w, out, s);
}

// ---------------------------------------------------------------------------------------------
// wgrad_mfma: rows r = t*M + m (flattened taps x shifted-tensor channels), cols n.
//   A operand (16 rows x 4 voxels): lane l holds S[pos(q2 = 4*step + (l>>4)) + d_t][m] for row (l&15)
//   B operand (4 voxels x 16 cols): lane l holds C[q2 = 4*step + (l>>4)][n0 + (l&15)]
//   D (16 rows x 16 cols): lane l reg r -> row 4*(l>>4)+r, col l&15
// grid: x = chunk of q-rows (n,q0,q1), y = group of RT row tiles, z = group of BN col tiles.
// ---------------------------------------------------------------------------------------------
template <int RT, int BN>
__global__ __launch_bounds__(256) void wgrad_mfma_kernel(GatherGeom g, const float* __restrict__ S,
                                                         const float* __restrict__ C, float* __restrict__ slab,
                                                         int rows_per_chunk) {
  extern __shared__ float red[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int ml = lane & 15;
  const int kq = lane >> 4;
  const int rows_total = g.ntaps * g.K;
  const int rt0 = blockIdx.y * RT;
  const int n0 = blockIdx.z * (16 * BN);
  const int64_t R = (int64_t)g.N * g.q_d[0] * g.q_d[1];

  int rd0[RT], rd1[RT], rd2[RT], rm[RT];
  bool rok[RT];
#pragma unroll
  for (int i = 0; i < RT; ++i) {
    int r = (rt0 + i) * 16 + ml;
    rok[i] = r < rows_total;
    int t = rok[i] ? r / g.K : 0;
    rm[i] = rok[i] ? r - t * g.K : 0;
    rd0[i] = g.tap_d[t][0];
    rd1[i] = g.tap_d[t][1];
    rd2[i] = g.tap_d[t][2];
  }
  f32x4 acc[RT][BN];
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int c = 0; c < BN; ++c) acc[i][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int64_t row_begin = (int64_t)blockIdx.x * rows_per_chunk;
  int64_t row_end = row_begin + rows_per_chunk;
  if (row_end > R) row_end = R;
  const int steps = (g.q_d[2] + 3) >> 2;

  for (int64_t row = row_begin + wave; row < row_end; row += 4) {
    int q1 = (int)(row % g.q_d[1]);
    int64_t tmp = row / g.q_d[1];
    int q0 = (int)(tmp % g.q_d[0]);
    int n = (int)(tmp / g.q_d[0]);
    const float* crow = C + (int64_t)row * g.q_d[2] * g.out_cs;
    int64_t sbase[RT];
    bool sok[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      int p0 = q0 * g.si[0] + rd0[i];
      int p1 = q1 * g.si[1] + rd1[i];
      sok[i] = rok[i] && p0 >= 0 && p0 < g.in_d[0] && p1 >= 0 && p1 < g.in_d[1];
      sbase[i] = (((int64_t)n * g.in_d[0] + p0) * g.in_d[1] + p1) * g.in_d[2] * (int64_t)g.in_cs + rm[i];
    }
    for (int st = 0; st < steps; ++st) {
      const int q2 = st * 4 + kq;
      const bool vq = q2 < g.q_d[2];
      float b[BN], a[RT];
#pragma unroll
      for (int c = 0; c < BN; ++c) {
        int col = n0 + c * 16 + ml;
        b[c] = (vq && col < g.Nn) ? crow[(int64_t)q2 * g.out_cs + col] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < RT; ++i) {
        int p2 = q2 * g.si[2] + rd2[i];
        a[i] = (vq && sok[i] && p2 >= 0 && p2 < g.in_d[2]) ? S[sbase[i] + (int64_t)p2 * g.in_cs] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int c = 0; c < BN; ++c) acc[i][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[c], acc[i][c], 0, 0, 0);
    }
  }

  // block reduce through LDS: layout [wave][tile][lane][4]
  float* mine = red + ((size_t)wave * RT * BN * 64 + lane) * 4;
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int c = 0; c < BN; ++c) *(f32x4*)(mine + (size_t)(i * BN + c) * 256) = acc[i][c];
  __syncthreads();
  float* myslab = slab + (int64_t)blockIdx.x * rows_total * g.Nn;
  for (int tile = wave; tile < RT * BN; tile += 4) {
    f32x4 v = *(f32x4*)(red + ((size_t)0 * RT * BN * 64 + lane) * 4 + (size_t)tile * 256);
    for (int ww = 1; ww < 4; ++ww) v += *(f32x4*)(red + ((size_t)ww * RT * BN * 64 + lane) * 4 + (size_t)tile * 256);
    int i = tile / BN, c = tile - i * BN;
    int col = n0 + c * 16 + ml;
    if (col >= g.Nn) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int rr = (rt0 + i) * 16 + kq * 4 + r;
      if (rr < rows_total) {
        int t = rr / g.K;
        int m = rr - t * g.K;
        myslab[((int64_t)g.tap_w[t] * g.K + m) * g.Nn + col] = v[r];
      }
    }
  }
}

WgradPlan wgrad_plan(const GatherGeom& g) {
  WgradPlan p;
  p.rows = g.ntaps * g.K;
  int rtiles = (p.rows + 15) / 16;
  int ctiles = (g.Nn + 15) / 16;
  // accumulator budget: RT*BN <= 16 tiles (64 VGPRs)
  if (ctiles >= 4) { p.BN = 4; p.RT = 4; }
  else if (ctiles >= 2) { p.BN = 2; p.RT = 7; }
  else { p.BN = 1; p.RT = (rtiles > 7) ? 14 : 7; }
  int gy = (rtiles + p.RT - 1) / p.RT;
  int gz = (ctiles + p.BN - 1) / p.BN;
  int64_t R = (int64_t)g.N * g.q_d[0] * g.q_d[1];
  int64_t want = 2048 / ((int64_t)gy * gz);
  if (want < 1) want = 1;
  int64_t maxc = cdiv64(R, 4);
  if (want > maxc) want = maxc;
  int64_t wsz = (int64_t)p.rows * g.Nn;  // floats per slab (taps in this geom; dw slab indexed by tap_w may be larger)
  int64_t cap = ((int64_t)16 << 20) / (wsz > 0 ? wsz : 1);
  if (cap < 1) cap = 1;
  if (want > cap) want = cap;
  p.nchunks = (int)want;
  p.scratch_bytes = (size_t)p.nchunks * wsz * sizeof(float);
  return p;
}

// dw[tap_w[t]][m][n] += sum over chunk slabs; a block owns 64 elements, its 4 waves split the chunks
// (fixed order -> bitwise reproducible).
__global__ __launch_bounds__(256) void slab_reduce_kernel(float* __restrict__ dw, const float* __restrict__ slab,
                                                          GatherGeom g, int nchunks) {
  __shared__ float sm[4][64];
  const int e = threadIdx.x & 63, cg = threadIdx.x >> 6;
  const int64_t per = (int64_t)g.ntaps * g.K * g.Nn;
  const int64_t i = (int64_t)blockIdx.x * 64 + e;
  int64_t idx = 0;
  float s0 = 0.f, s1 = 0.f;
  if (i < per) {
    int64_t t = i / ((int64_t)g.K * g.Nn);
    idx = (int64_t)g.tap_w[t] * g.K * g.Nn + (i - t * (int64_t)g.K * g.Nn);
    int c = cg;
    for (; c + 4 < nchunks; c += 8) {
      s0 += slab[(int64_t)c * per + idx];
      s1 += slab[(int64_t)(c + 4) * per + idx];
    }
    for (; c < nchunks; c += 4) s0 += slab[(int64_t)c * per + idx];
  }
  sm[cg][e] = s0 + s1;
  __syncthreads();
  if (cg == 0 && i < per) dw[idx] += (sm[0][e] + sm[1][e]) + (sm[2][e] + sm[3][e]);
}

template <int RT, int BN>
static int launch_wgrad_t(const GatherGeom& g, const WgradPlan& p, const float* S, const float* C, float* slab,
                          hipStream_t s) {
  int rtiles = (p.rows + 15) / 16;
  int ctiles = (g.Nn + 15) / 16;
  int64_t R = (int64_t)g.N * g.q_d[0] * g.q_d[1];
  int rpc = (int)cdiv64(R, p.nchunks);
  dim3 grid((unsigned)p.nchunks, (unsigned)((rtiles + RT - 1) / RT), (unsigned)((ctiles + BN - 1) / BN));
  size_t lds = (size_t)4 * RT * BN * 64 * 4 * sizeof(float);
  hipLaunchKernelGGL((wgrad_mfma_kernel<RT, BN>), grid, dim3(256), lds, s, g, S, C, slab, rpc);
  URSN_HIP(hipGetLastError());
  return 0;
}

int launch_wgrad_mfma(const GatherGeom& g, const float* S, const float* C, float* dw, void* scratch,
                      size_t scratch_bytes, hipStream_t s) {
  WgradPlan p = wgrad_plan(g);
  URSN_REQUIRE(scratch != nullptr && scratch_bytes >= p.scratch_bytes, "wgrad scratch too small: %zu < %zu",
               scratch_bytes, p.scratch_bytes);
  // NOTE: the slab is indexed with tap_w (global tap index) but sized by this geom's tap count;
  // geoms passed here must therefore use tap_w in [0, ntaps) -- build_geoms(PASS_WGRAD) guarantees it
  // by handing out one geom with all taps.
  float* slab = (float*)scratch;
  int rc;
  if (p.BN == 4) { ursn_note_kernel("wgrad_mfma<4,4>"); rc = launch_wgrad_t<4, 4>(g, p, S, C, slab, s); }
  else if (p.BN == 2) { ursn_note_kernel("wgrad_mfma<7,2>"); rc = launch_wgrad_t<7, 2>(g, p, S, C, slab, s); }
  else if (p.RT == 14) { ursn_note_kernel("wgrad_mfma<14,1>"); rc = launch_wgrad_t<14, 1>(g, p, S, C, slab, s); }
  else { ursn_note_kernel("wgrad_mfma<7,1>"); rc = launch_wgrad_t<7, 1>(g, p, S, C, slab, s); }
  if (rc) return rc;
  int64_t per = (int64_t)p.rows * g.Nn;
  hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)cdiv64(per, 64)), dim3(256), 0, s, dw, (const float*)slab, g,
                     p.nchunks);
  URSN_HIP(hipGetLastError());
  return 0;
}
