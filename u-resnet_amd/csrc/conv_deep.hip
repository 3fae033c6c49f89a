// fp32 3x3x3 stride-1 convolution for the DEEPEST levels (>= 128 channels on 12^3 / 6^3 / 8^3 ... volumes: the residual units of
// spatial levels 4-5 of an F = 8 network, lib/resnet_module.py:43-66 as built by lib/uresnet.py:56-64,95-100; forward and
// data gradient) -- WEIGHT-STREAMING, CONTRACTION SPLIT OVER THE FOUR WAVES, on v_mfma_f32_16x16x4_f32.  The fp32 twin of
// bf16_convdeep.hip.
//
// At 12^3 / 6^3 (x batch 4) a pass is 6.1 / 3.1 GFLOP = 39 / 20 us of fp32 matrix time; the all-taps implicit GEMM
// (conv_igemm_kernel.h, igemm_at<16>) runs them in 104 / 87 us: 288-864 workgroups of one wave per SIMD that stage the halo
// box AND a 27-tap weight slab per 16-channel chunk behind barriers, on boxes that idle lanes of 12- and 6-wide rows.  Here
//   * a WAVE owns 64 produced channels x 64 voxels (16 accumulator tiles) and ONE 16-channel chunk of the contraction per
//     round: 27 taps x 64 MFMAs = 55k matrix cycles between two barriers;
//   * the A fragments (weights, packed per (block of 64 produced channels, chunk of 16 contraction channels) as one linear
//     stream of 27 taps x 4 KB) go straight from L2 to registers (one 16-byte load per lane feeds 4 MFMAs), requested two
//     taps ahead; the B fragments are one ds_read_b128 per 16-voxel tile and tap (4 channels of the lane's voxel, element j
//     feeds MFMA step j: the k index of a chunk is permuted so that a lane's four k values are contiguous in memory);
//   * the voxel tile is 64 CONSECUTIVE voxels (x fastest) of a box whose extents divide the volume (4 x 4 x 4 at 12^3; the whole
//     6^3 / 8^3 image at the last level: 216 voxels = 3.4 tiles instead of boxes that pad 6 to 8 on every axis);
//   * the four partial tiles are summed through LDS in wave order (bitwise reproducible), then each wave finishes one
//     16-voxel tile: stores, BatchNorm moments around per-lane pivots;
//   * where that leaves fewer workgroups than CUs (6^3: 16 tiles x 4 channel blocks) the chunks are ALSO split over
//     workgroups (fp32 slabs, summed in slice order by dconv_reduce_kernel together with the moments).
#include <stdlib.h>

#include "ursn_common.h"
#include "buffer_stage.h"

namespace {

typedef float dc_f32x4 __attribute__((ext_vector_type(4)));
#define DC_MAXJ 4          // halo voxels per thread per piece plane (pp <= 1024)
#define DC_OOB 0x80000000u

struct DCArgs {
  const float* in;
  const float* wp;         // [cout block of 64][chunk of 16][tap][co tile 0..3][lane = 16 kq + m][4]: W[tap][16 chunk + 4 kq + j][64 cob + 16 mt + m]
  float* out;
  float* slab;             // gsplit > 1: [slice][voxel][Cout]
  double* stats_partial;   // gsplit == 1, forward: [cout block][gridDim.x][2][64] doubles, or null
  int N, Z, Y, X;
  int in_cs, out_cs, Cout;
  int nchunks;             // contraction channels / 16
  int gsplit, rounds;      // slices of the chunk list over workgroups; rounds of 4 chunks per workgroup
  int bq[3], nb[3];        // box of produced voxels, boxes per axis
  int tpb;                 // 64-voxel tiles per box
  int zpad;                // halo planes per side along z: 1, or 0 for 2-D problems (Z = 1)
  int hy, hxp, pp;         // halo image: rows per plane, row stride (voxels), voxels per piece plane (multiple of 64)
  int accumulate;
  int toff[27];            // LDS byte offset of tap t relative to the lane's voxel
};

// NTAP: 27 (3-D) | 9 (2-D problems: Z = 1, the taps of the middle z plane only)
template <int NTAP, bool STATS, bool SLAB>
__global__ __launch_bounds__(256, 1) void dconv_kernel(DCArgs a) {
  constexpr int WK = 4, NT = 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n16 = lane & 15, kq = lane >> 4;
  const int cob = blockIdx.y;
  // blockIdx.x = (slice * boxes + box) * tpb + tile
  const int boxes = a.N * a.nb[0] * a.nb[1] * a.nb[2];
  int r = blockIdx.x;
  const int tib = r % a.tpb; r /= a.tpb;
  const int ks = r / boxes;
  int box = r - ks * boxes;
  const int bx = box % a.nb[2]; box /= a.nb[2];
  const int by = box % a.nb[1]; box /= a.nb[1];
  const int bz = box % a.nb[0];
  const int n = box / a.nb[0];
  const int z0 = bz * a.bq[0], y0 = by * a.bq[1], x0 = bx * a.bq[2];
  const int img_bytes = 4 * a.pp * 16;
  const int hz = a.bq[0] + 2 * a.zpad;

  // ---- staging geometry (fixed over rounds and images): this thread's halo voxels v = j * 256 + tid of a piece plane ----
  unsigned vrel[DC_MAXJ];
#pragma unroll
  for (int j = 0; j < DC_MAXJ; ++j) {
    vrel[j] = DC_OOB;
    const int v = j * 256 + tid;
    if (v < a.pp) {
      const int hzz = v / (a.hy * a.hxp), r2 = v - hzz * a.hy * a.hxp;
      const int hyy = r2 / a.hxp, hxx = r2 - hyy * a.hxp;
      const int gz = z0 + hzz - a.zpad, gy = y0 + hyy - 1, gx = x0 + hxx - 1;
      if (hzz < hz && hxx < a.bq[2] + 2 && gz >= 0 && gz < a.Z && gy >= 0 && gy < a.Y && gx >= 0 && gx < a.X)
        vrel[j] = (unsigned)(((gz * a.Y + gy) * a.X + gx) * a.in_cs) * 4u;
    }
    asm volatile("" : "+v"(vrel[j]));
  }
  const size_t img_elems = (size_t)a.Z * a.Y * a.X * a.in_cs;
  const __amdgpu_buffer_rsrc_t rin = ursn_plane_rsrc(a.in + (size_t)n * img_elems, (unsigned)(img_elems * 4));
  auto stage = [&](int round) {
    const unsigned choff = (unsigned)((ks * a.rounds + round) * WK * 16) * 4u;   // first chunk of the round, bytes inside a voxel
#pragma unroll
    for (int k = 0; k < WK; ++k)
#pragma unroll
      for (int pc = 0; pc < 4; ++pc) {
        unsigned char* dst = lds + (size_t)((k * 4 + pc) * a.pp) * 16 + wave * 1024;
        const unsigned so = choff + (unsigned)(k * 16 + pc * 4) * 4u;
#pragma unroll
        for (int j = 0; j < DC_MAXJ; ++j)
          if (j * 256 + wave * 64 < a.pp) ursn_bload_lds_b128_so(rin, dst + j * 4096, vrel[j], so);
      }
  };

  // ---- the lane's voxels: tile nt, column n16 of this workgroup's 64 consecutive voxels of the box ----
  unsigned vbase[NT];
  int gvox[NT];   // voxel index inside the image, -1: past the box or outside the volume
  const int bvox = a.bq[0] * a.bq[1] * a.bq[2];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int lin = tib * 64 + nt * 16 + n16;
    const int qx = lin % a.bq[2], r2 = lin / a.bq[2];
    const int qy = r2 % a.bq[1], qz = r2 / a.bq[1];
    const bool ok = lin < bvox && z0 + qz < a.Z && y0 + qy < a.Y && x0 + qx < a.X;
    const int cz = ok ? qz : 0, cy = ok ? qy : 0, cx = ok ? qx : 0;   // idle columns read a valid LDS address
    vbase[nt] = (unsigned)((kq * a.pp + (cz * a.hy + cy) * a.hxp + cx) * 16);
    gvox[nt] = ok ? ((z0 + qz) * a.Y + y0 + qy) * a.X + x0 + qx : -1;
  }
  const unsigned char* img = lds + wave * img_bytes;

  dc_f32x4 acc[4][NT];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (dc_f32x4){0.f, 0.f, 0.f, 0.f};

  const unsigned lane_off = (unsigned)lane * 16u;
  for (int round = 0; round < a.rounds; ++round) {
    if (round > 0) __syncthreads();   // every wave is done with the images of the previous round
    stage(round);
    const int chunk = __builtin_amdgcn_readfirstlane((ks * a.rounds + round) * WK + wave);
    const unsigned char* wsrc = (const unsigned char*)a.wp + ((size_t)(cob * a.nchunks + chunk) * NTAP) * 4096;
    dc_f32x4 A[3][4];   // fragments two taps ahead of their MFMAs
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      A[0][mt] = *(const dc_f32x4*)(wsrc + lane_off + mt * 1024);
      A[1][mt] = *(const dc_f32x4*)(wsrc + 4096 + lane_off + mt * 1024);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the images has landed (and its first fragments)
    __syncthreads();
    dc_f32x4 B[2][NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) B[0][nt] = *(const dc_f32x4*)(img + vbase[nt] + a.toff[0]);
#pragma unroll
    for (int t = 0; t < NTAP; ++t) {
      const int cb = t & 1, ca = t % 3;
      if (t + 2 < NTAP) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) A[(t + 2) % 3][mt] = *(const dc_f32x4*)(wsrc + (t + 2) * 4096 + lane_off + mt * 1024);
      }
      if (t + 1 < NTAP) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) B[cb ^ 1][nt] = *(const dc_f32x4*)(img + vbase[nt] + a.toff[t + 1]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[ca][mt][j], B[cb][nt][j], acc[mt][nt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);   // one scheduling region per tap
    }
  }

  // ---- sum of the four partial tiles through LDS, in wave order; voxel tile nt is finished by wave nt ----
  __syncthreads();   // the images are dead
  {
    unsigned char* mine = lds + (size_t)(wave * 4 * NT) * 1024 + lane * 16;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      if (nt == wave) continue;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) *(dc_f32x4*)(mine + (mt * NT + nt) * 1024) = acc[mt][nt];
    }
  }
  __syncthreads();
  dc_f32x4 fin[4];
  int myvox = gvox[0];
#pragma unroll
  for (int q = 1; q < NT; ++q) if (wave == q) myvox = gvox[q];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    dc_f32x4 s = (dc_f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < WK; ++j) {
      dc_f32x4 p;
      if (j == wave) {
        p = acc[mt][0];
#pragma unroll
        for (int q = 1; q < NT; ++q) if (wave == q) p = acc[mt][q];
      } else {
        p = *(const dc_f32x4*)(lds + (size_t)(j * 4 * NT + mt * NT + wave) * 1024 + lane * 16);
      }
      if (j == 0) s = p; else s += p;
    }
    fin[mt] = s;
  }

  // ---- epilogue: lane (n16, kq) holds produced channels cob * 64 + 16 mt + 4 kq + (0..3) of voxel (tile = wave, column n16) ----
  float piv[16], s1[16], s2[16], nacc = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) piv[k] = s1[k] = s2[k] = 0.f;
  const size_t vox_img = (size_t)a.Z * a.Y * a.X;
  const bool ok = myvox >= 0;
  const size_t vox = (size_t)n * vox_img + (size_t)(ok ? myvox : 0);
  if constexpr (SLAB) {
    float* sp = a.slab + ((size_t)ks * a.N * vox_img + vox) * a.Cout + cob * 64 + 4 * kq;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
      if (ok && cob * 64 + 16 * mt < a.Cout) *(dc_f32x4*)(sp + 16 * mt) = fin[mt];
  } else {
    float* op = a.out + vox * a.out_cs + cob * 64 + 4 * kq;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      if (!(ok && cob * 64 + 16 * mt < a.Cout)) continue;
      dc_f32x4 v = fin[mt];
      if (a.accumulate) v += *(const dc_f32x4*)(op + 16 * mt);
      *(dc_f32x4*)(op + 16 * mt) = v;
      if constexpr (STATS) {
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          piv[4 * mt + r4] = v[r4];   // one voxel per lane: the pivot is the value, the shifted sums stay 0
        }
      }
    }
    if constexpr (STATS) if (ok) nacc = 1.f;
  }
  if constexpr (STATS && !SLAB) {
    __shared__ double red[4][128];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      double u, w2;
      ursn_sacc_final(piv[k], s1[k], s2[k], nacc, u, w2);
#pragma unroll
      for (int o = 8; o >= 1; o >>= 1) { u += __shfl_xor(u, o); w2 += __shfl_xor(w2, o); }
      if (n16 == 0) {
        const int ch = 16 * (k >> 2) + 4 * kq + (k & 3);
        red[wave][ch] = u;
        red[wave][64 + ch] = w2;
      }
    }
    __syncthreads();
    if (tid < 128) {
      const int ch = tid & 63;
      double t = 0.0;
      if (cob * 64 + ch < a.Cout) t = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
      a.stats_partial[((size_t)cob * gridDim.x + blockIdx.x) * 128 + tid] = t;
    }
  }
}

// ---- sum of the gsplit slabs (slice order), stores, BatchNorm moment partials --------------------------------------------
struct DCRedArgs {
  const float* slab; float* out; double* stats_partial;   // [gridDim.x][2][C] doubles or null
  int64_t V; int C, out_cs, gsplit, accumulate;
};
__global__ __launch_bounds__(256) void dconv_reduce_kernel(DCRedArgs a) {
  const int cg = threadIdx.x, rows = blockDim.y, vr = threadIdx.y;
  float piv[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f}, nacc = 0.f;
  for (int64_t v = (int64_t)blockIdx.x * rows + vr; v < a.V; v += (int64_t)gridDim.x * rows) {
    dc_f32x4 s = *(const dc_f32x4*)(a.slab + (size_t)v * a.C + 4 * cg);
    for (int j = 1; j < a.gsplit; ++j) s += *(const dc_f32x4*)(a.slab + ((size_t)j * a.V + v) * a.C + 4 * cg);
    float* q = a.out + (size_t)v * a.out_cs + 4 * cg;
    if (a.accumulate) s += *(const dc_f32x4*)q;
    *(dc_f32x4*)q = s;
    if (a.stats_partial) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (nacc == 0.f) piv[r] = s[r];
        ursn_sacc(piv[r], s1[r], s2[r], s[r]);
      }
      nacc += 1.f;
    }
  }
  if (a.stats_partial) {
    __shared__ double red[256][8];
    const int t = vr * blockDim.x + cg;
#pragma unroll
    for (int r = 0; r < 4; ++r) ursn_sacc_final(piv[r], s1[r], s2[r], nacc, red[t][r], red[t][4 + r]);
    __syncthreads();
    if (vr == 0) {
      for (int r = 0; r < 4; ++r) {
        double u = 0.0, w2 = 0.0;
        for (int y = 0; y < rows; ++y) { u += red[y * blockDim.x + cg][r]; w2 += red[y * blockDim.x + cg][4 + r]; }
        a.stats_partial[(size_t)blockIdx.x * 2 * a.C + 4 * cg + r] = u;
        a.stats_partial[(size_t)blockIdx.x * 2 * a.C + a.C + 4 * cg + r] = w2;
      }
    }
  }
}

// ---- weight packing: master W[t][ci][co] (forward) / its transpose with flipped taps (data gradient) -> fragment order ----
struct DCPackArgs {
  const float* w;
  float* wp;
  int cin_w, cout_w;   // the STORED tensor [27][cin_w][cout_w]
  int flip;            // data gradient: contraction = stored cout, produced = stored cin, tap t reads stored tap 26 - t
  int nchunks, ncob, K, Nn, ntap;
};
__global__ __launch_bounds__(256) void dconv_pack_kernel(DCPackArgs k) {
  const int64_t total = (int64_t)k.ncob * k.nchunks * k.ntap * 1024;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int j = (int)(e & 3), lane = (int)((e >> 2) & 63), mt = (int)((e >> 8) & 3);
    int64_t r = e >> 10;
    const int t = (int)(r % k.ntap); r /= k.ntap;
    const int ch = (int)(r % k.nchunks), cob = (int)(r / k.nchunks);
    const int kk = ch * 16 + 4 * (lane >> 4) + j, nn = cob * 64 + mt * 16 + (lane & 15);
    float v = 0.f;
    if (kk < k.K && nn < k.Nn)
      v = k.flip ? k.w[((size_t)(k.ntap - 1 - t) * k.cin_w + nn) * k.cout_w + kk] : k.w[((size_t)t * k.cin_w + kk) * k.cout_w + nn];
    k.wp[e] = v;
  }
}

struct DCPlan {
  int K, Nn, ics, ocs;   // kernel view (swapped for the data gradient)
  int Z, Y, X, ntap;
  int bq[3], nb[3], tpb;
  int hy, hxp, pp;
  int nchunks, ncob, gsplit, rounds;
  int boxes;
  size_t lds;
  int red_blocks;
};

bool dc_plan(const ursn_conv_desc& d, ConvPass pass, DCPlan& p) {
  static const bool off = getenv("URSN_DCONV") && getenv("URSN_DCONV")[0] == '0';
  if (off || d.dtype != 0 || (d.ndim != 3 && d.ndim != 2) || d.transposed || d.k != 3 || d.stride != 1) return false;
  if (pass != PASS_FWD && pass != PASS_DGRAD) return false;
  if (d.in_split || d.in_mean || d.pw_dy || d.bs_partial || d.vdz_z) return false;   // the fused forms stay with their kernels
  const bool flip = pass == PASS_DGRAD;
  const int ics = d.in_cstride > 0 ? d.in_cstride : d.cin, ocs = d.out_cstride > 0 ? d.out_cstride : d.cout;
  p.K = flip ? d.cout : d.cin; p.Nn = flip ? d.cin : d.cout;
  p.ics = flip ? ocs : ics; p.ocs = flip ? ics : ocs;
  static const int mink = getenv("URSN_DCONV_MINK") ? atoi(getenv("URSN_DCONV_MINK")) : 64;
  if (p.K < mink || (p.K & 63) || p.Nn < 64 || (p.Nn & 15) || (p.ics & 3) || (p.ocs & 3)) return false;
  if (d.ndim == 3) { p.Z = d.in_sp[0]; p.Y = d.in_sp[1]; p.X = d.in_sp[2]; p.ntap = 27; }
  else { p.Z = 1; p.Y = d.in_sp[0]; p.X = d.in_sp[1]; p.ntap = 9; }   // 2-D: one z plane, the 9 in-plane taps
  const int Z = p.Z, Y = p.Y, X = p.X;
  if ((d.ndim == 3 && Z < 2) || Y < 4 || X < 4) return false;
  const int64_t V = (int64_t)d.n * Z * Y * X;
  // measured against the all-taps implicit GEMM (tools/op_bench.py, 128 channels): 3-D 12^3 x 4 1.45x, 16^3 1.08x, 24^3 1.02-1.05x,
  // 32^3 1.04-1.05x (127-136 TFLOP/s); 2-D 64^2 x 4 1.0x, 32^2 x 16 0.95x, 64^2 x 16 0.92-1.0x: 3-D always, 2-D up to 8192 voxels
  static const int64_t maxv3 = getenv("URSN_DCONV_MAXVOX") ? atoll(getenv("URSN_DCONV_MAXVOX")) : ((int64_t)1 << 24);
  static const int64_t maxv2 = getenv("URSN_DCONV_MAXVOX2D") ? atoll(getenv("URSN_DCONV_MAXVOX2D")) : 8192;
  if (V > (d.ndim == 3 ? maxv3 : maxv2)) return false;
  if (p.K < 128 && (d.ndim != 3 || V > 65536)) return false;   // 64 contraction channels: 24^3 x 4 1.02-1.06x, 48^3 1.0x, 2-D 0.87x
  if ((int64_t)Z * Y * X * (p.ics > p.ocs ? p.ics : p.ocs) * 4 >= ((int64_t)1 << 31)) return false;
  p.nchunks = p.K / 16;
  p.ncob = (p.Nn + 63) / 64;
  // box: extents that divide the volume (else the smallest cover), at least 64 voxels, halo image <= 1024 voxels per piece plane;
  // cost = 64-voxel tiles to run (idle columns included) x staged halo per tile
  double best = 1e300;
  int pick[3] = {0, 0, 0};
  for (int bz = 1; bz <= Z && bz <= 16; ++bz)
    for (int by = 2; by <= Y && by <= 32; ++by)
      for (int bx = 4; bx <= X && bx <= 32; ++bx) {
        const int bv = bz * by * bx;
        const int hzp = bz + (d.ndim == 3 ? 2 : 0);
        if (bv < 48 || hzp * (by + 2) * (bx + 2) > 1024 - 64) continue;
        const int nbz = (Z + bz - 1) / bz, nby = (Y + by - 1) / by, nbx = (X + bx - 1) / bx;
        const double tiles = (double)nbz * nby * nbx * ((bv + 63) / 64);
        const double halo = (double)hzp * (by + 2) * (bx + 2) / 64.0;   // every tile's workgroup stages the WHOLE box image
        const double cost = tiles * (1.0 + 0.05 * halo);
        if (cost < best) { best = cost; pick[0] = bz; pick[1] = by; pick[2] = bx; }
      }
  if (!pick[0]) return false;
  for (int j = 0; j < 3; ++j) p.bq[j] = pick[j];
  p.nb[0] = (Z + pick[0] - 1) / pick[0]; p.nb[1] = (Y + pick[1] - 1) / pick[1]; p.nb[2] = (X + pick[2] - 1) / pick[2];
  p.tpb = (pick[0] * pick[1] * pick[2] + 63) / 64;
  p.hy = pick[1] + 2; p.hxp = pick[2] + 2;
  p.pp = (((pick[0] + (d.ndim == 3 ? 2 : 0)) * p.hy * p.hxp) + 63) & ~63;
  if (p.pp > 256 * DC_MAXJ) return false;
  const size_t images = (size_t)4 * 4 * p.pp * 16, red = (size_t)4 * 4 * 4 * 1024;
  p.lds = images > red ? images : red;
  if (p.lds > 150 * 1024) return false;
  p.boxes = d.n * p.nb[0] * p.nb[1] * p.nb[2];
  const int64_t tiles = (int64_t)p.boxes * p.tpb;
  static const int64_t minwg = getenv("URSN_DCONV_MINWG") ? atoi(getenv("URSN_DCONV_MINWG")) : 160;
  p.gsplit = 0;
  for (int gs = 1; gs <= 8; gs *= 2) {
    if (p.nchunks % (gs * 4)) continue;
    if (gs > 1 && (p.Nn > 1024 || 256 % (p.Nn / 4))) continue;   // the split-K reduce kernel's thread map
    p.gsplit = gs;
    if (tiles * p.ncob * gs >= minwg) break;
  }
  if (!p.gsplit) return false;
  p.rounds = p.nchunks / (p.gsplit * 4);
  {
    const int cgs = p.Nn / 4, rows = cgs >= 256 ? 1 : 256 / cgs;
    const int64_t b = (V + rows - 1) / rows;
    p.red_blocks = (int)(b < 2048 ? b : 2048);
  }
  return tiles * p.gsplit < ((int64_t)1 << 24);
}

}  // namespace

int deep_conv_supported(const ursn_conv_desc& d, ConvPass pass) {
  DCPlan p;
  return dc_plan(d, pass, p) ? 1 : 0;
}

// scratch: packed weights + (split-K) slabs, in floats; statistics partials in doubles
size_t deep_conv_scratch_floats(const ursn_conv_desc& d, ConvPass pass) {
  DCPlan p;
  if (!dc_plan(d, pass, p)) return 0;
  size_t e = (size_t)p.ncob * p.nchunks * p.ntap * 1024 + 64;
  if (p.gsplit > 1) e += (size_t)p.gsplit * d.n * p.Z * p.Y * p.X * p.Nn;
  return e;
}
size_t deep_conv_stats_scratch_doubles(const ursn_conv_desc& d) {
  DCPlan p;
  if (!dc_plan(d, PASS_FWD, p)) return 0;
  return p.gsplit > 1 ? (size_t)p.red_blocks * 2 * p.Nn : (size_t)p.boxes * p.tpb * p.ncob * 128;
}

template <int NTAP, bool STATS, bool SLAB>
static int dc_launch(const DCPlan& p, const DCArgs& a, hipStream_t s) {
  auto kern = dconv_kernel<NTAP, STATS, SLAB>;
  static size_t attr = 48 * 1024;
  if (p.lds > attr) {
    URSN_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds));
    attr = p.lds;
  }
  hipLaunchKernelGGL(kern, dim3(p.boxes * p.tpb * p.gsplit, p.ncob), dim3(256), p.lds, s, a);
  URSN_HIP(hipGetLastError());
  return 0;
}

// out (=|+=) conv(in, w) [forward] or conv^T(in = dy, w) [data gradient]; stats_partial != nullptr (forward): mean / rstd of the
// produced tensor are finalised too.  scratch: deep_conv_scratch_floats(d, pass) floats.
int launch_deep_conv(const ursn_conv_desc& d, ConvPass pass, const float* in, const float* w, float* out, int accumulate,
                     float* scratch, double* stats_partial, float eps, float* mean, float* rstd, hipStream_t s) {
  DCPlan p;
  URSN_REQUIRE(dc_plan(d, pass, p) && scratch, "deep conv: unsupported shape or no scratch");
  const size_t wtotal = (size_t)p.ncob * p.nchunks * p.ntap * 1024;
  {
    DCPackArgs k;
    k.w = w; k.wp = scratch; k.cin_w = d.cin; k.cout_w = d.cout; k.flip = pass == PASS_DGRAD ? 1 : 0;
    k.nchunks = p.nchunks; k.ncob = p.ncob; k.K = p.K; k.Nn = p.Nn; k.ntap = p.ntap;
    const int blocks = (int)(cdiv64((int64_t)wtotal, 256) < 4096 ? cdiv64((int64_t)wtotal, 256) : 4096);
    hipLaunchKernelGGL(dconv_pack_kernel, dim3(blocks), dim3(256), 0, s, k);
    URSN_HIP(hipGetLastError());
  }
  DCArgs a;
  a.in = in; a.wp = scratch; a.out = out; a.stats_partial = stats_partial;
  a.slab = p.gsplit > 1 ? scratch + wtotal + 64 : nullptr;
  a.N = d.n; a.Z = p.Z; a.Y = p.Y; a.X = p.X;
  a.in_cs = p.ics; a.out_cs = p.ocs; a.Cout = p.Nn;
  a.nchunks = p.nchunks; a.gsplit = p.gsplit; a.rounds = p.rounds;
  for (int j = 0; j < 3; ++j) { a.bq[j] = p.bq[j]; a.nb[j] = p.nb[j]; }
  a.tpb = p.tpb; a.hy = p.hy; a.hxp = p.hxp; a.pp = p.pp; a.zpad = p.ntap == 27 ? 1 : 0;
  a.accumulate = accumulate;
  for (int t = 0; t < 27; ++t) a.toff[t] = 0;
  for (int t = 0; t < p.ntap; ++t) {   // tap t = (tz, ty, tx) reads the voxel at displacement (t* - 1): halo position (+t*); 2-D: no z halo
    const int tz = p.ntap == 27 ? t / 9 : 0, ty = (t / 3) % 3, tx = t % 3;
    a.toff[t] = ((tz * p.hy + ty) * p.hxp + tx) * 16;
  }
  const bool slab = p.gsplit > 1, st = stats_partial != nullptr && !slab;
  ursn_note_kernel(pass == PASS_DGRAD ? (slab ? "dconv_dgrad+splitk" : "dconv_dgrad") : (slab ? "dconv+splitk" : "dconv"));
  if (p.ntap == 27) {
    if (slab) URSN_TRY((dc_launch<27, false, true>(p, a, s)));
    else if (st) URSN_TRY((dc_launch<27, true, false>(p, a, s)));
    else URSN_TRY((dc_launch<27, false, false>(p, a, s)));
  } else {
    if (slab) URSN_TRY((dc_launch<9, false, true>(p, a, s)));
    else if (st) URSN_TRY((dc_launch<9, true, false>(p, a, s)));
    else URSN_TRY((dc_launch<9, false, false>(p, a, s)));
  }
  const int64_t V = (int64_t)d.n * p.Z * p.Y * p.X;
  if (slab) {
    DCRedArgs r;
    r.slab = a.slab; r.out = out; r.stats_partial = stats_partial;
    r.V = V; r.C = p.Nn; r.out_cs = p.ocs; r.gsplit = p.gsplit; r.accumulate = accumulate;
    const int cgs = p.Nn / 4;
    hipLaunchKernelGGL(dconv_reduce_kernel, dim3(p.red_blocks), dim3(cgs, 256 / cgs), 0, s, r);
    URSN_HIP(hipGetLastError());
  }
  if (stats_partial) {
    if (slab) return launch_bn_stats_final(stats_partial, p.red_blocks, p.Nn, p.Nn, V, eps, mean, rstd, s);
    const int nblk = p.boxes * p.tpb;
    return launch_bn_stats_final_blocked(stats_partial, nblk, p.Nn, 64, 64, (size_t)nblk * 128, V, eps, mean, rstd, s);
  }
  return 0;
}
