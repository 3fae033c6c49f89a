// bf16 3x3x3 stride-1 convolution for the DEEP levels (contraction channels >= 64, a multiple of 32; the residual units of
// spatial levels 3-5 of an F = 8 network: lib/resnet_module.py:43-66 as built by lib/uresnet.py:56-64,95-100; forward and
// data gradient) -- WEIGHT-STREAMING, SPLIT-K INSIDE THE WORKGROUP, on v_mfma_f32_16x16x32_bf16.
//
// At 32^3 / 16^3 / 8^3 (x batch 4) with 64 / 128 / 256 channels a pass is 29 / 14.5 / 7.2 GFLOP -- 3-12 us of matrix time --
// and the WEIGHTS are the large operand (0.2 / 0.9 / 3.5 MB against 16 / 4 / 1 MB of activations).  The generic box kernel
// (bf16_conv.hip) gives a wave 16 voxels x 32 produced channels (3 LDS reads per 2 MFMAs), re-stages the packed weights of a
// chunk per 64-voxel box and puts 64-256 workgroups of one wave per SIMD on the chip: 40-66 us per pass.  Here
//   * a WAVE owns 64 produced channels x (64 | 128) voxels -- 16 | 32 accumulator tiles -- so one B fragment (16 voxels x 32
//     channels of one tap: one ds_read_b128) feeds 4 MFMAs and one A fragment feeds 4 | 8;
//   * the A fragments (weights, packed per (block of 64 produced channels, chunk of 32 contraction channels) as one linear
//     stream of 27 taps x 4 KB) go STRAIGHT from L2 to registers, 1 KB per wave instruction, requested two taps ahead: no LDS
//     space, no barrier and no second copy for the larger operand; the 4 waves of a workgroup read DIFFERENT weights;
//   * the 4 waves split the contraction (WK chunks of 32 channels side by side) and, for WK = 2, the voxel tile: every wave
//     reads only ITS chunk's halo-box image from LDS (piece-major [16-byte piece][z][y][x]: a 16-lane group reads 256
//     contiguous bytes), the images of a round are DMA'd once per workgroup through the buffer path (padding arrives as zeros);
//   * the WK partial tiles are summed through LDS in wave order (fixed order: bitwise reproducible), each wave then owns a
//     share of the voxel tiles for the epilogue: bf16 rounding, 8-byte stores, BatchNorm moments of the STORED tensor with
//     per-lane pivots;
//   * where voxels are few (8^3: 2048 voxels, 16 tiles) the contraction is ALSO split over workgroups (gsplit slices of the
//     chunks): fp32 slabs, summed in slice order by bdconv_reduce_kernel, which rounds, stores and takes the moments.
// Bytes a workgroup pulls from L2 per FLOP = 1 / (voxels of its tile): 128-256 voxel tiles keep that under the ~30 B/clk a CU
// gets from L2 (MI355X_MICROARCH.md, gather table) at the matrix rate these levels can reach.
#include <stdlib.h>

#include "bf16_common.h"
#include "bf16_pack.h"
#include "buffer_stage.h"

namespace {

#define BD_MAXJ 4        // halo voxels per thread per image (pp <= 1024)
#define BD_OOB 0x80000000u

struct BDArgs {
  const bf16_t* in;
  const bf16_t* wp;        // [cout block of 64][chunk of 32][tap][co tile 0..3][lane][8]
  bf16_t* out;
  float* slab;             // gsplit > 1: [slice][voxel][Cout] fp32
  double* stats_partial;   // gsplit == 1, forward: [cout block][gridDim.x][2][64] doubles, or null
  int N, Z, Y, X;
  int in_cs, out_cs, Cout;
  int nchunks;             // contraction channels / 32
  int gsplit, rounds;      // slices of the chunk list over workgroups; rounds of WK chunks per workgroup
  int bq[3], nb[3];        // box of produced voxels (bq[0] * bq[1] * bq[2] = voxels of a workgroup tile; powers of two), boxes per axis
  int lbx, lby;            // log2 of bq[2], bq[1]
  int hy, hxp, pp;         // halo image: rows per plane, padded row stride (voxels), voxels per piece plane (multiple of 64)
  int accumulate;
  int toff[27];            // LDS byte offset of tap t inside an image (relative to the lane's voxel)
};

// Sum over the 16 lanes of a DPP row (lanes 16 r .. 16 r + 15), every lane ends with the total: four data-parallel-primitive
// moves per half instead of four ds_bpermute round trips (the moments epilogue cost 5-11 us per launch through __shfl_xor)
template <int CTRL>
__device__ __forceinline__ double bd_dpp(double v) {
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, 0xf, 0xf, false);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), CTRL, 0xf, 0xf, false);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double bd_row_sum(double v) {
  v += bd_dpp<0xB1>(v);    // quad_perm [1,0,3,2]: lane ^ 1
  v += bd_dpp<0x4E>(v);    // quad_perm [2,3,0,1]: lane ^ 2
  v += bd_dpp<0x141>(v);   // row_half_mirror: 7 - lane inside each half row
  v += bd_dpp<0x140>(v);   // row_mirror: 15 - lane
  return v;
}

// WK: waves that split the contraction (2 | 4); WV = 4 / WK waves split the voxel tile; NT: 16-voxel tiles per wave.
template <int WK, int NT, bool STATS, bool SLAB>
__global__ __launch_bounds__(256, 1) void bdconv_kernel(BDArgs a) {
  constexpr int WV = 4 / WK;
  constexpr int NOWN = NT / WK;   // voxel tiles a wave owns in the epilogue
  static_assert(NT % WK == 0, "ownership by voxel tile");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: its branches are scalar branches
  const int n16 = lane & 15, g = lane >> 4;
  const int kw = wave % WK, wv = wave / WK;
  const int cob = blockIdx.y;
  // blockIdx.x = slice * tiles + tile
  const int tiles = a.N * a.nb[0] * a.nb[1] * a.nb[2];
  const int ks = blockIdx.x / tiles;
  int tile = blockIdx.x - ks * tiles;
  const int bx = tile % a.nb[2]; tile /= a.nb[2];
  const int by = tile % a.nb[1]; tile /= a.nb[1];
  const int bz = tile % a.nb[0];
  const int n = tile / a.nb[0];
  const int z0 = bz * a.bq[0], y0 = by * a.bq[1], x0 = bx * a.bq[2];
  const int img_bytes = 4 * a.pp * 16;           // one chunk image: 4 pieces planes
  const int hz = a.bq[0] + 2;

  // ---- staging geometry (fixed over rounds and images): this thread's halo voxels v = j * 256 + tid of an image ----
  // One DMA instruction moves 64 consecutive pieces of ONE (chunk, piece) plane; pp is a multiple of 64, so a wave is
  // entirely inside a plane or entirely past it.  Per DMA only the scalar (chunk, piece) offset changes.
  unsigned vrel[BD_MAXJ];
#pragma unroll
  for (int j = 0; j < BD_MAXJ; ++j) {
    vrel[j] = BD_OOB;
    const int v = j * 256 + tid;
    if (v < a.pp) {
      const int hzz = v / (a.hy * a.hxp), r2 = v - hzz * a.hy * a.hxp;
      const int hyy = r2 / a.hxp, hxx = r2 - hyy * a.hxp;
      const int gz = z0 + hzz - 1, gy = y0 + hyy - 1, gx = x0 + hxx - 1;
      if (hzz < hz && hxx < a.bq[2] + 2 && gz >= 0 && gz < a.Z && gy >= 0 && gy < a.Y && gx >= 0 && gx < a.X)
        vrel[j] = (unsigned)(((gz * a.Y + gy) * a.X + gx) * a.in_cs) * 2u;
    }
    asm volatile("" : "+v"(vrel[j]));
  }
  const size_t img_elems = (size_t)a.Z * a.Y * a.X * a.in_cs;
  const __amdgpu_buffer_rsrc_t rin = ursn_rsrc(a.in + (size_t)n * img_elems, (unsigned)(img_elems * 2));
  auto stage = [&](int round) {
    const unsigned choff = (unsigned)((ks * a.rounds + round) * WK * 32) * 2u;   // first chunk of the round, bytes inside a voxel
#pragma unroll
    for (int k = 0; k < WK; ++k)
#pragma unroll
      for (int pc = 0; pc < 4; ++pc) {
        unsigned char* dst = lds + (size_t)((k * 4 + pc) * a.pp) * 16 + wave * 1024;
        const unsigned so = choff + (unsigned)(k * 32 + pc * 8) * 2u;
#pragma unroll
        for (int j = 0; j < BD_MAXJ; ++j)
          if (j * 256 + wave * 64 < a.pp) ursn_bload_lds_b128_so(rin, dst + j * 4096, vrel[j], so);
      }
  };

  // ---- per-lane B geometry: voxel (tile nt, column n16) of this wave's part of the box, piece g ----
  unsigned vbase[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int lin = (wv * NT + nt) * 16 + n16;
    const int qx = lin & (a.bq[2] - 1), r2 = lin >> a.lbx;
    const int qy = r2 & (a.bq[1] - 1), qz = r2 >> a.lby;
    vbase[nt] = (unsigned)((g * a.pp + (qz * a.hy + qy) * a.hxp + qx) * 16);
  }
  const unsigned char* img = lds + kw * img_bytes;

  bf_f32x4 acc[4][NT];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (bf_f32x4){0.f, 0.f, 0.f, 0.f};

  const unsigned lane_off = (unsigned)lane * 16u;
  for (int round = 0; round < a.rounds; ++round) {
    if (round > 0) __syncthreads();   // every wave is done with the images of the previous round
    stage(round);
    const int chunk = __builtin_amdgcn_readfirstlane((ks * a.rounds + round) * WK + kw);
    // uniform base in scalar registers + the lane's 32-bit offset: one address register for the whole fragment stream
    const unsigned char* wsrc = (const unsigned char*)a.wp + ((size_t)(cob * a.nchunks + chunk) * 27) * 4096;
    // A fragments two taps ahead of their MFMAs (L2 latency against 256 | 512 cycles of matrix work per tap)
    bfx8 A[3][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      A[0][mt] = *(const bfx8*)(wsrc + lane_off + mt * 1024);
      A[1][mt] = *(const bfx8*)(wsrc + 4096 + lane_off + mt * 1024);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the images has landed (and its first fragments)
    __syncthreads();
    bfx8 B[2][NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) B[0][nt] = *(const bfx8*)(img + vbase[nt] + a.toff[0]);
#pragma unroll
    for (int t = 0; t < 27; ++t) {
      const int cb = t & 1, ca = t % 3;
      if (t + 2 < 27) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) A[(t + 2) % 3][mt] = *(const bfx8*)(wsrc + (t + 2) * 4096 + lane_off + mt * 1024);
      }
      if (t + 1 < 27) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) B[cb ^ 1][nt] = *(const bfx8*)(img + vbase[nt] + a.toff[t + 1]);
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[ca][mt], B[cb][nt], acc[mt][nt], 0, 0, 0);
      // issue order: the next tap's operand requests ride in the issue slots between this tap's MFMAs (an MFMA holds the
      // vector issue port for 8 of its 16 cycles: MI355X_MICROARCH.md) instead of in front of the block
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                       // 4 MFMA
        if (t + 1 < 27) {
          __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);                     // the address add
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                     // one B fragment of the next tap
        }
        if (t + 2 < 27 && nt < 4) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // one A fragment two taps ahead
      }
      __builtin_amdgcn_sched_barrier(0);   // one scheduling region per tap (the group solver is exponential in the region's groups)
    }
  }

  // ---- sum of the WK partial tiles through LDS, in wave order; tile nt is finished by wave kw == nt % WK of its group ----
  __syncthreads();   // the images are dead
  {
    unsigned char* mine = lds + (size_t)((wv * WK + kw) * 4 * NT) * 1024 + lane * 16;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      if (nt % WK == kw) continue;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) *(bf_f32x4*)(mine + (mt * NT + nt) * 1024) = acc[mt][nt];
    }
  }
  __syncthreads();
  bf_f32x4 fin[4][NOWN];
#pragma unroll
  for (int o = 0; o < NOWN; ++o) {
    const int nt = o * WK + kw;   // resolved per wave below: kw is wave-uniform but not compile-time
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      bf_f32x4 s = (bf_f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < WK; ++j) {
        bf_f32x4 p;
        if (j == kw) {
          // own registers: select acc[mt][o * WK + kw] without dynamic register indexing
          p = acc[mt][o * WK];
#pragma unroll
          for (int q = 1; q < WK; ++q) if (kw == q) p = acc[mt][o * WK + q];
        } else {
          p = *(const bf_f32x4*)(lds + (size_t)((wv * WK + j) * 4 * NT + mt * NT + nt) * 1024 + lane * 16);
        }
        if (j == 0) s = p; else { s[0] += p[0]; s[1] += p[1]; s[2] += p[2]; s[3] += p[3]; }
      }
      fin[mt][o] = s;
    }
  }

  // ---- epilogue: lane (n16, g) holds produced channels cob * 64 + 16 mt + 4 g + (0..3) of voxel (tile nt, column n16) ----
  float piv[16], s1[16], s2[16], nacc = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) piv[k] = s1[k] = s2[k] = 0.f;
  const size_t vox_img = (size_t)a.Z * a.Y * a.X;
#pragma unroll
  for (int o = 0; o < NOWN; ++o) {
    const int nt = o * WK + kw;
    const int lin = (wv * NT + nt) * 16 + n16;
    const int qx = lin & (a.bq[2] - 1), r2 = lin >> a.lbx;
    const int qy = r2 & (a.bq[1] - 1), qz = r2 >> a.lby;
    const int gz = z0 + qz, gy = y0 + qy, gx = x0 + qx;
    const bool ok = gz < a.Z && gy < a.Y && gx < a.X;
    const size_t vox = (size_t)n * vox_img + ((size_t)gz * a.Y + gy) * a.X + gx;
    if constexpr (SLAB) {
      float* sp = a.slab + ((size_t)ks * a.N * vox_img + vox) * a.Cout + cob * 64 + 4 * g;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
        if (ok && cob * 64 + 16 * mt < a.Cout) *(bf_f32x4*)(sp + 16 * mt) = fin[mt][o];
    } else {
      bf16_t* op = a.out + vox * a.out_cs + cob * 64 + 4 * g;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        if (!(ok && cob * 64 + 16 * mt < a.Cout)) continue;
        bf_f32x4 v = fin[mt][o];
        u32x2* q = (u32x2*)(op + 16 * mt);
        if (a.accumulate) {
          const u32x2 e = *q;
          v[0] += __uint_as_float(e[0] << 16); v[1] += __uint_as_float(e[0] & 0xffff0000u);
          v[2] += __uint_as_float(e[1] << 16); v[3] += __uint_as_float(e[1] & 0xffff0000u);
        }
        u32x2 pk;
        pk[0] = pack_bf2(v[0], v[1]);
        pk[1] = pack_bf2(v[2], v[3]);
        *q = pk;
        if constexpr (STATS) {   // moments of the STORED (rounded) tensor: that is what BatchNorm will normalise
          const float rv[4] = {__uint_as_float(pk[0] << 16), __uint_as_float(pk[0] & 0xffff0000u),
                               __uint_as_float(pk[1] << 16), __uint_as_float(pk[1] & 0xffff0000u)};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (nacc == 0.f) piv[4 * mt + r] = rv[r];
            ursn_sacc(piv[4 * mt + r], s1[4 * mt + r], s2[4 * mt + r], rv[r]);
          }
        }
      }
      if constexpr (STATS) if (ok) nacc += 1.f;
    }
  }
  if constexpr (STATS && !SLAB) {
    __shared__ double red[4][128];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      double u, w2;
      ursn_sacc_final(piv[k], s1[k], s2[k], nacc, u, w2);
      u = bd_row_sum(u); w2 = bd_row_sum(w2);   // over the 16 lanes (voxel columns) of the lane's DPP row
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
      if (cob * 64 + ch < a.Cout) t = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
      a.stats_partial[((size_t)cob * gridDim.x + blockIdx.x) * 128 + tid] = t;
    }
  }
}

// ---- sum of the gsplit slabs (slice order), rounding, stores, BatchNorm moment partials --------------------------------------
struct BDRedArgs {
  const float* slab; bf16_t* out; double* stats_partial;   // [gridDim.x][2][C] doubles or null
  int64_t V; int C, out_cs, gsplit, accumulate;
};
// thread = (voxel row vr, 4-channel group cg): blockDim = (C / 4, 256 / (C / 4)); a block walks voxels vr + rows * i
__global__ __launch_bounds__(256) void bdconv_reduce_kernel(BDRedArgs a) {
  const int cg = threadIdx.x, rows = blockDim.y, vr = threadIdx.y;
  float piv[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f}, nacc = 0.f;
  for (int64_t v = (int64_t)blockIdx.x * rows + vr; v < a.V; v += (int64_t)gridDim.x * rows) {
    bf_f32x4 s = *(const bf_f32x4*)(a.slab + (size_t)v * a.C + 4 * cg);
    for (int j = 1; j < a.gsplit; ++j) {
      const bf_f32x4 p = *(const bf_f32x4*)(a.slab + ((size_t)j * a.V + v) * a.C + 4 * cg);
      s[0] += p[0]; s[1] += p[1]; s[2] += p[2]; s[3] += p[3];
    }
    u32x2* q = (u32x2*)(a.out + (size_t)v * a.out_cs + 4 * cg);
    if (a.accumulate) {
      const u32x2 e = *q;
      s[0] += __uint_as_float(e[0] << 16); s[1] += __uint_as_float(e[0] & 0xffff0000u);
      s[2] += __uint_as_float(e[1] << 16); s[3] += __uint_as_float(e[1] & 0xffff0000u);
    }
    u32x2 pk;
    pk[0] = pack_bf2(s[0], s[1]);
    pk[1] = pack_bf2(s[2], s[3]);
    *q = pk;
    if (a.stats_partial) {
      const float rv[4] = {__uint_as_float(pk[0] << 16), __uint_as_float(pk[0] & 0xffff0000u),
                           __uint_as_float(pk[1] << 16), __uint_as_float(pk[1] & 0xffff0000u)};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (nacc == 0.f) piv[r] = rv[r];
        ursn_sacc(piv[r], s1[r], s2[r], rv[r]);
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

// (weight packing: BPK_DEEP in bf16_pack.hip -- [cout block][chunk][tap][co tile][lane = 16 g + m][8])

struct BDPlan {
  int wk, nt;            // kernel form
  int bq[3], nb[3];
  int hy, hxp, pp;
  int nchunks, ncob, gsplit, rounds;
  int tiles;             // workgroup tiles (all images)
  size_t lds;
  int red_blocks;        // gsplit > 1: blocks of the reduce kernel (= rows of its statistics partials)
};

// blocks of the split-K reduce kernel: one pass of 256 / (Nn / 4) voxel rows each while that fills the chip, never fewer
// (32 blocks walking 8 MB of slabs took 17 us at 8^3)
int red_blocks_for(const GatherGeom& g) {
  const int64_t V = (int64_t)g.N * g.in_d[0] * g.in_d[1] * g.in_d[2];
  const int cgs = g.Nn / 4, rows = cgs >= 256 ? 1 : 256 / (cgs > 0 ? cgs : 1);
  const int64_t b = (V + rows - 1) / rows;
  return (int)(b < 2048 ? b : 2048);
}

bool bd_plan(const GatherGeom& g, BDPlan& p) {
  const int Z = g.in_d[0], Y = g.in_d[1], X = g.in_d[2];
  p.nchunks = g.K / 32;
  p.ncob = (g.Nn + 63) / 64;
  struct Form { int wk, nt, vox; };
  static const Form forms[3] = {{2, 8, 256}, {4, 8, 128}, {2, 4, 128}};
  static int force_form = -2, force_gs = 0;
  static int64_t minwg = 192;
  if (force_form == -2) {   // A/B: URSN_BDCONV_FORM="form,gsplit", URSN_BDCONV_MINWG
    force_form = -1;
    const char* e = getenv("URSN_BDCONV_FORM");
    if (e) sscanf(e, "%d,%d", &force_form, &force_gs);
    const char* m = getenv("URSN_BDCONV_MINWG");
    if (m) minwg = atoi(m);
  }
  bool have = false;
  BDPlan best;
  int64_t best_wg = 0;
  for (int gs = 1; gs <= 8; gs *= 2)
    for (int fi = 0; fi < 3; ++fi) {
      if (force_form >= 0 && (fi != force_form || gs != force_gs)) continue;
      const Form& f = forms[fi];
      if (p.nchunks % (gs * f.wk)) continue;
      BDPlan c = p;
      c.wk = f.wk; c.nt = f.nt; c.gsplit = gs; c.rounds = p.nchunks / (gs * f.wk);
      // box of f.vox produced voxels: powers of two per axis, x extent 4 / 8 / 16 (a 16-voxel MFMA tile = 4 / 2 / 1 x rows);
      // least padding first (voxels the boxes cover beyond the volume are idle MFMA columns), then the smaller halo
      {
        double best_cost = 1e300;
        int pick[3] = {0, 0, 0};
        for (int bx = 16; bx >= 4; bx >>= 1)
          for (int byy = 1; byy <= 16; byy <<= 1) {
            if ((f.vox / bx) % byy) continue;
            const int bzz = f.vox / bx / byy;
            if (bzz < 1 || bzz > 16 || bx * byy < 16) continue;   // a 16-voxel tile stays inside one z plane
            const double covered = (double)((Z + bzz - 1) / bzz * bzz) * ((Y + byy - 1) / byy * byy) * ((X + bx - 1) / bx * bx);
            const double halo = (double)(bzz + 2) * (byy + 2) * (bx + 2) / f.vox;
            const double cost = covered * (1.0 + 0.1 * halo) * (bx == 16 ? 1.0 : (bx == 8 ? 1.08 : 1.15));   // narrow rows: padded row strides (LDS) or bank conflicts
            if (cost < best_cost) { best_cost = cost; pick[0] = bzz; pick[1] = byy; pick[2] = bx; }
          }
        if (!pick[2]) continue;
        c.bq[0] = pick[0]; c.bq[1] = pick[1]; c.bq[2] = pick[2];
      }
      const int bx = c.bq[2], byy = c.bq[1];
      if (gs > 1 && (g.Nn > 1024 || 256 % (g.Nn / 4))) continue;   // the split-K reduce kernel's thread map
      for (int j = 0; j < 3; ++j) c.nb[j] = (g.in_d[j] + c.bq[j] - 1) / c.bq[j];
      c.hy = byy + 2;
      // padded row stride: rows of one 16-voxel tile must fall on disjoint 16-byte slots modulo 256 bytes (bx = 8: stride = 8
      // mod 16, bx = 4: stride = 4 mod 16) where the image still fits; bx = 16: any
      c.hxp = bx + 2;
      if (bx < 16) { int want = bx + 2; while ((want & 15) != bx) ++want; c.hxp = want; }
      auto sizes = [&](BDPlan& q) {
        q.pp = (((q.bq[0] + 2) * q.hy * q.hxp) + 63) & ~63;
        const size_t images = (size_t)q.wk * 4 * q.pp * 16;
        const size_t red = (size_t)4 * 4 * q.nt * 1024;
        q.lds = images > red ? images : red;
        return images;
      };
      size_t images = sizes(c);
      if (c.lds > 150 * 1024 || c.pp > 256 * BD_MAXJ) { c.hxp = bx + 2; images = sizes(c); }
      if (c.lds > 150 * 1024 || c.pp > 256 * BD_MAXJ) continue;
      (void)images;
      const int64_t tiles = (int64_t)g.N * c.nb[0] * c.nb[1] * c.nb[2];
      if (tiles * gs > (1 << 24)) continue;
      c.tiles = (int)tiles;
      const int64_t wg = tiles * gs * c.ncob;
      // useful fraction of the box (edge tiles idle lanes): prefer forms that waste less when several reach the target
      if (!have || wg > best_wg) { if (!have || best_wg < minwg) { best = c; best_wg = wg; have = true; } }
      if (wg >= minwg) {
        c.red_blocks = red_blocks_for(g);
        p = c;
        return true;
      }
    }
  if (!have) return false;
  best.red_blocks = red_blocks_for(g);
  p = best;
  return true;
}

}  // namespace

bool bdconv_ok(const GatherGeom& g) {
  static const bool off = getenv("URSN_BDCONV") && getenv("URSN_BDCONV")[0] == '0';
  if (off) return false;
  if (g.ntaps != 27 || g.K < 64 || (g.K & 31) || g.Nn < 64 || (g.Nn & 15) || (g.in_cs & 7) || (g.out_cs & 3)) return false;
  for (int j = 0; j < 3; ++j) {
    if (g.so[j] != 1 || g.si[j] != 1 || g.po[j] != 0) return false;
    if (g.in_d[j] != g.out_d[j] || g.in_d[j] != g.q_d[j]) return false;
  }
  for (int t = 0; t < 27; ++t)
    for (int j = 0; j < 3; ++j)
      if (g.tap_d[t][j] < -1 || g.tap_d[t][j] > 1) return false;
  // one buffer resource per image (32-bit byte offsets, out-of-range marker at 2 GB)
  if ((int64_t)g.in_d[0] * g.in_d[1] * g.in_d[2] * (g.in_cs > g.out_cs ? g.in_cs : g.out_cs) * 2 >= ((int64_t)1 << 31)) return false;
  if (g.in_d[0] < 2 || g.in_d[1] < 4 || g.in_d[2] < 4) return false;
  BDPlan p;
  return bd_plan(g, p);
}

size_t bdconv_pack_elems(const GatherGeom& g) {
  BDPlan p;
  if (!bd_plan(g, p)) return 0;
  size_t e = (size_t)p.ncob * p.nchunks * 27 * 2048 + 8;
  if (p.gsplit > 1) {   // fp32 slabs behind the packed weights (256-byte aligned)
    e = (e + 127) & ~(size_t)127;
    e += (size_t)p.gsplit * g.N * g.in_d[0] * g.in_d[1] * g.in_d[2] * g.Nn * 2;
  }
  return e;
}
int bdconv_grid_blocks(const GatherGeom& g) {   // rows of the statistics partials per block of produced channels
  BDPlan p;
  if (!bd_plan(g, p)) return 0;
  return p.gsplit > 1 ? p.red_blocks : p.tiles;
}
size_t bdconv_stats_scratch_doubles(const GatherGeom& g) {
  BDPlan p;
  if (!bd_plan(g, p)) return 0;
  return p.gsplit > 1 ? (size_t)p.red_blocks * 2 * g.Nn : (size_t)p.tiles * p.ncob * 128;
}

template <int WK, int NT, bool STATS, bool SLAB>
static int bd_launch(const BDPlan& p, const BDArgs& a, hipStream_t s) {
  auto kern = bdconv_kernel<WK, NT, STATS, SLAB>;
  static size_t attr = 48 * 1024;
  if (p.lds > attr) {
    URSN_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds));
    attr = p.lds;
  }
  hipLaunchKernelGGL(kern, dim3(p.tiles * p.gsplit, p.ncob), dim3(256), p.lds, s, a);
  URSN_HIP(hipGetLastError());
  return 0;
}

int launch_bdconv(const GatherGeom& g, const bf16_t* in, const float* w, int Kw, int Nw, bf16_t* wpack, bf16_t* out,
                  double* stats_partial, hipStream_t s) {
  BDPlan p;
  URSN_REQUIRE(bdconv_ok(g) && bd_plan(g, p), "bf16 deep conv: unsupported geometry");
  const size_t wtotal = (size_t)p.ncob * p.nchunks * 27 * 2048;
  {
    BPackJob k = bpack_job(BPK_DEEP);
    k.w = w; k.wp = wpack; k.Kw = Kw > 0 ? Kw : g.K; k.Nw = Nw > 0 ? Nw : g.Nn;
    k.w_tap_stride = g.w_tap_stride; k.w_sk = g.w_sk; k.w_sn = g.w_sn;
    for (int t = 0; t < 27; ++t) k.tap[t] = g.tap_w[t];
    k.p[0] = p.nchunks; k.p[1] = p.ncob;
    k.blocks = (int)(cdiv64((int64_t)wtotal, 256) < 4096 ? cdiv64((int64_t)wtotal, 256) : 4096);
    URSN_TRY(bpack_submit(k, s));
  }
  BDArgs a;
  a.in = in; a.wp = wpack; a.out = out; a.stats_partial = stats_partial;
  a.slab = p.gsplit > 1 ? (float*)(wpack + (((wtotal + 8) + 127) & ~(size_t)127)) : nullptr;
  a.N = g.N; a.Z = g.in_d[0]; a.Y = g.in_d[1]; a.X = g.in_d[2];
  a.in_cs = g.in_cs; a.out_cs = g.out_cs; a.Cout = g.Nn;
  a.nchunks = p.nchunks; a.gsplit = p.gsplit; a.rounds = p.rounds;
  for (int j = 0; j < 3; ++j) { a.bq[j] = p.bq[j]; a.nb[j] = p.nb[j]; }
  a.hy = p.hy; a.hxp = p.hxp; a.pp = p.pp;
  a.lbx = __builtin_ctz(p.bq[2]); a.lby = __builtin_ctz(p.bq[1]);
  a.accumulate = g.accumulate;
  for (int t = 0; t < 27; ++t)   // the lane's voxel sits at halo position (+1, +1, +1): tap d lands at (1 + d) per axis
    a.toff[t] = (((g.tap_d[t][0] + 1) * p.hy + (g.tap_d[t][1] + 1)) * p.hxp + (g.tap_d[t][2] + 1)) * 16;
  int rc = 3;
  const bool slab = p.gsplit > 1, st = stats_partial != nullptr && !slab;
#define BD(wk_, nt_, label)                                                                   \
  if (p.wk == wk_ && p.nt == nt_) {                                                           \
    ursn_note_kernel(slab ? label "+splitk" : label);                                          \
    rc = slab ? bd_launch<wk_, nt_, false, true>(p, a, s)                                      \
              : (st ? bd_launch<wk_, nt_, true, false>(p, a, s) : bd_launch<wk_, nt_, false, false>(p, a, s)); \
  }
  BD(2, 8, "bdconv_bf16<2,8>") BD(4, 8, "bdconv_bf16<4,8>") BD(2, 4, "bdconv_bf16<2,4>")
#undef BD
  URSN_TRY(rc);
  if (slab) {
    BDRedArgs r;
    r.slab = a.slab; r.out = out; r.stats_partial = stats_partial;
    r.V = (int64_t)g.N * a.Z * a.Y * a.X; r.C = g.Nn; r.out_cs = g.out_cs; r.gsplit = p.gsplit; r.accumulate = g.accumulate;
    const int cgs = g.Nn / 4;
    URSN_REQUIRE(cgs <= 256 && 256 % cgs == 0, "bf16 deep conv: split-K reduce needs a power-of-two channel count <= 1024 (have %d)", g.Nn);
    hipLaunchKernelGGL(bdconv_reduce_kernel, dim3(p.red_blocks), dim3(cgs, 256 / cgs), 0, s, r);
    URSN_HIP(hipGetLastError());
  }
  return 0;
}

int bdconv_stats_finalize(const GatherGeom& g, const double* partial, int64_t V, float eps, float* mean, float* rstd, hipStream_t s) {
  BDPlan p;
  URSN_REQUIRE(bd_plan(g, p), "bf16 deep conv: unsupported geometry");
  if (p.gsplit > 1) return launch_bn_stats_final(partial, p.red_blocks, g.Nn, g.Nn, V, eps, mean, rstd, s);
  return launch_bn_stats_final_blocked(partial, p.tiles, g.Nn, 64, 64, (size_t)p.tiles * 128, V, eps, mean, rstd, s);
}
