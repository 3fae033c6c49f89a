// bf16 plan: conv0 -- the network's first layer (lib/uresnet.py:37-45: 3x3x3 stride 1, ONE input channel -> F = 8) reading the
// raw fp32 data: forward (+ BatchNorm moments) and weight gradient.  No data gradient exists (the input is data).
//
// The 8 -> 8 kernels ran this layer with seven zero channels per staged voxel: the matrix work, LDS image and staging of a real
// 8-channel layer (0.62 / 0.73 ms per cfg5 pass) for 1/8 of the contraction.  With one input channel the contraction IS the
// taps:
//   forward   D[co][voxel] = sum_tap W[co][tap] x[voxel + tap]      v_mfma_f32_16x16x16_bf16 per (16 voxels, tap plane dz):
//             k = 4 (dy + 1) + (dx + 1), i.e. a lane's four k values are the WINDOW {x-1, x, x+1, 0} of one row -- the staged
//             planes hold that window per voxel (8 bytes), so a B operand is one aligned ds_read_b64;
//   wgrad     D[co][tap] = sum_voxel dz[voxel][co] x[voxel + tap]    v_mfma_f32_16x16x32_bf16 per (32 voxels of a row, 16 taps):
//             A = dz transposed on the way out of LDS (ds_read_b64_tr_b16), B = eight consecutive voxels of the row shifted by
//             the tap -- the staged planes keep three copies of every row (shift -1 | 0 | +1) so that read is one aligned
//             ds_read_b128 whatever dx is.
// Both march z over a 64 x 8 tile with a ring of four staged x planes (fp32 -> bf16 on the way in); HBM traffic is the
// algorithmic 4 B + 16 B per voxel (+ the tile halo of x).
#include <stdlib.h>

#include "bf16_common.h"
#include "bf16_pack.h"
#include "buffer_stage.h"

namespace {

constexpr int C0_TX = 64, C0_TY = 8, C0_ROWS = C0_TY + 2;
typedef short c0_s16x4 __attribute__((ext_vector_type(4)));

struct C0Args {
  const float* x;          // (N, Z, Y, X) fp32
  const bf16_t* wp;        // forward: [dz][lane][4] A fragments
  bf16_t* out;             // forward: z (N, Z, Y, X, out_cs)
  const bf16_t* dz;        // wgrad: (N, Z, Y, X, dz_cs)
  float* slab;             // wgrad: [grid][32 taps][8]
  double* stats_partial;   // forward: [grid][2][16] doubles or null
  int N, Z, Y, X;
  int out_cs, dz_cs;
  int zseg, nzseg, nty, ntx;
};

// per-thread staging geometry of a 10-row x 64-column plane of the tile: element (row r, column c) = thread's slot i of
// idx = tid + 256 i (a wave = one row); byte offsets of x[c-1], x[c], x[c+1] inside the z plane, or the out-of-range marker
struct C0Stage { unsigned oc[3], ol[3], orr[3]; };
__device__ __forceinline__ void c0_stage_geom(const C0Args& a, int x0, int y0, int tid, C0Stage& g) {
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int idx = tid + 256 * i, row = idx >> 6, c = idx & 63;
    const int gy = y0 + row - 1, gx = x0 + c;
    const bool rok = idx < C0_ROWS * 64 && gy >= 0 && gy < a.Y;
    const unsigned base = (unsigned)(gy * a.X + gx) * 4u;
    g.oc[i] = rok && gx < a.X ? base : URSN_OOB_OFFSET;
    g.ol[i] = rok && gx - 1 >= 0 && gx - 1 < a.X ? base - 4u : URSN_OOB_OFFSET;
    g.orr[i] = rok && gx + 1 < a.X ? base + 4u : URSN_OOB_OFFSET;
  }
}
__device__ __forceinline__ unsigned c0_bf(float v) { return (unsigned)f2bf(v); }

// ---- forward ---------------------------------------------------------------------------------------------------------------
template <bool STATS>
__global__ __launch_bounds__(256) void b0conv_kernel(C0Args a) {
  __shared__ __attribute__((aligned(16))) uint2 ring[4][C0_ROWS][C0_TX];   // window {x-1, x, x+1, 0} per voxel
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n16 = lane & 15, g = lane >> 4;
  int b = blockIdx.x;
  const int tx = b % a.ntx; b /= a.ntx;
  const int ty = b % a.nty; b /= a.nty;
  const int zs = b % a.nzseg, n = b / a.nzseg;
  const int x0 = tx * C0_TX, y0 = ty * C0_TY, z0 = zs * a.zseg;
  const int z1 = z0 + a.zseg < a.Z ? z0 + a.zseg : a.Z;

  c0_s16x4 A[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) A[d] = *(const c0_s16x4*)(a.wp + ((size_t)d * 64 + lane) * 4);

  C0Stage sg;
  c0_stage_geom(a, x0, y0, tid, sg);
  const size_t plane = (size_t)a.Y * a.X;
  const float* img = a.x + (size_t)n * a.Z * plane;
  float vl[3], vc[3], vr[3];
  auto load_plane = [&](int p) {
    const bool zok = p >= 0 && p < a.Z;
    const __amdgpu_buffer_rsrc_t r = ursn_plane_rsrc(img + (size_t)(zok ? p : 0) * plane, zok ? (unsigned)(plane * 4) : 0u);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      vc[i] = ursn_buffer_load_f1(r, sg.oc[i]);
      vl[i] = ursn_buffer_load_f1(r, sg.ol[i]);
      vr[i] = ursn_buffer_load_f1(r, sg.orr[i]);
    }
  };
  auto write_plane = [&](int p) {
    uint2* dst = &ring[p & 3][0][0];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int idx = tid + 256 * i;
      if (idx < C0_ROWS * 64) {
        uint2 wv;
        wv.x = c0_bf(vl[i]) | (c0_bf(vc[i]) << 16);
        wv.y = c0_bf(vr[i]);
        dst[idx] = wv;
      }
    }
  };
  // ring slot = plane & 3 with planes offset by one so that plane -1 has a slot: use (p + 4) & 3 through p & 3 of p + 4
  load_plane(z0 - 1); write_plane(z0 - 1 + 4);
  load_plane(z0);     write_plane(z0 + 4);
  load_plane(z0 + 1); write_plane(z0 + 1 + 4);
  __syncthreads();

  float piv[4], s1[4], s2[4], nacc = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) piv[k] = s1[k] = s2[k] = 0.f;
  const int gy_row = g < 3 ? g : 2;   // k group 3 has zero weights: any finite operand
  const size_t out_img = (size_t)n * a.Z * plane;

  for (int z = z0; z < z1; ++z) {
    if (z + 2 <= z1) load_plane(z + 2);   // plane z + 2 is read from iteration z + 1 on (as its z + 1 plane)
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int row = 2 * wave + rr, gy = y0 + row;
#pragma unroll
      for (int xt = 0; xt < 4; ++xt) {
        bf_f32x4 acc = (bf_f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          const uint2 wv = ring[(z + d - 1 + 4) & 3][row + gy_row][xt * 16 + n16];
          c0_s16x4 B;
          B[0] = (short)(wv.x & 0xffff); B[1] = (short)(wv.x >> 16); B[2] = (short)(wv.y & 0xffff); B[3] = (short)(wv.y >> 16);
          acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(A[d], B, acc, 0, 0, 0);
        }
        const int gx = x0 + xt * 16 + n16;
        const bool ok = g < 2 && gy < a.Y && gx < a.X;
        u32x2 pk;
        pk[0] = pack_bf2(acc[0], acc[1]);
        pk[1] = pack_bf2(acc[2], acc[3]);
        if (ok) *(u32x2*)(a.out + (out_img + (size_t)z * plane + (size_t)gy * a.X + gx) * a.out_cs + 4 * g) = pk;
        if constexpr (STATS) {
          if (ok) {
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
      }
    }
    if (z + 2 <= z1) write_plane(z + 2 + 4);   // its slot held plane z - 2: last read in iteration z - 1
    __syncthreads();
  }

  if constexpr (STATS) {
    __shared__ double red[4][4][8];   // [wave][g (0,1 used)][sum r 0..3 | sumsq r 0..3]
    double u[4], w2[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      ursn_sacc_final(piv[r], s1[r], s2[r], nacc, u[r], w2[r]);
#pragma unroll
      for (int o = 8; o >= 1; o >>= 1) { u[r] += __shfl_xor(u[r], o); w2[r] += __shfl_xor(w2[r], o); }   // over the 16 voxel columns
    }
    if (n16 == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { red[wave][g][r] = u[r]; red[wave][g][4 + r] = w2[r]; }
    }
    __syncthreads();
    if (tid < 32) {   // partial row [2][16]: sums of channel c at [c], of squares at [16 + c]
      const int c = tid & 15, sq = tid >> 4;
      double t = 0.0;
      if (c < 8) {
        const int gg = c >> 2, r = (c & 3) + 4 * sq;
        t = (red[0][gg][r] + red[1][gg][r]) + (red[2][gg][r] + red[3][gg][r]);
      }
      a.stats_partial[(size_t)blockIdx.x * 32 + tid] = t;
    }
  }
}

// ---- weight gradient --------------------------------------------------------------------------------------------------------
__device__ __forceinline__ c0_s16x4 c0_tr16(const unsigned char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((c0_s16x4 __attribute__((address_space(3)))*)p);
}

__global__ __launch_bounds__(256) void b0wgrad_kernel(C0Args a) {
  // x planes: [ring 4][shift 3][row 10][64] bf16 (copy s holds x[c + s - 1] at column c); dz planes: [2][8 rows][64 voxels][8] bf16
  __shared__ __attribute__((aligned(16))) bf16_t xs[4][3][C0_ROWS][C0_TX];
  __shared__ __attribute__((aligned(16))) bf16_t dzs[2][C0_TY * C0_TX * 8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n16 = lane & 15, g = lane >> 4;
  int b = blockIdx.x;
  const int tx = b % a.ntx; b /= a.ntx;
  const int ty = b % a.nty; b /= a.nty;
  const int zs = b % a.nzseg, n = b / a.nzseg;
  const int x0 = tx * C0_TX, y0 = ty * C0_TY, z0 = zs * a.zseg;
  const int z1 = z0 + a.zseg < a.Z ? z0 + a.zseg : a.Z;

  C0Stage sg;
  c0_stage_geom(a, x0, y0, tid, sg);
  const size_t plane = (size_t)a.Y * a.X;
  const float* img = a.x + (size_t)n * a.Z * plane;
  float vl[3], vc[3], vr[3];
  auto load_plane = [&](int p) {
    const bool zok = p >= 0 && p < a.Z;
    const __amdgpu_buffer_rsrc_t r = ursn_plane_rsrc(img + (size_t)(zok ? p : 0) * plane, zok ? (unsigned)(plane * 4) : 0u);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      vc[i] = ursn_buffer_load_f1(r, sg.oc[i]);
      vl[i] = ursn_buffer_load_f1(r, sg.ol[i]);
      vr[i] = ursn_buffer_load_f1(r, sg.orr[i]);
    }
  };
  auto write_plane = [&](int p) {
    const int slot = p & 3;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int idx = tid + 256 * i;
      if (idx < C0_ROWS * 64) {
        (&xs[slot][0][0][0])[idx] = f2bf(vl[i]);
        (&xs[slot][1][0][0])[idx] = f2bf(vc[i]);
        (&xs[slot][2][0][0])[idx] = f2bf(vr[i]);
      }
    }
  };
  // dz tile of plane z: 8 rows x 64 voxels x 16 bytes, LDS-DMA (two pieces per thread); out-of-range voxels arrive as zeros
  unsigned doff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = tid + 256 * i, row = idx >> 6, c = idx & 63;
    const int gy = y0 + row, gx = x0 + c;
    doff[i] = gy < a.Y && gx < a.X ? (unsigned)((gy * a.X + gx) * a.dz_cs) * 2u : URSN_OOB_BYTES;
  }
  const bf16_t* dimg = a.dz + (size_t)n * a.Z * plane * a.dz_cs;
  auto dma_dz = [&](int p) {
    const __amdgpu_buffer_rsrc_t r = ursn_rsrc(dimg + (size_t)p * plane * a.dz_cs, (unsigned)(plane * a.dz_cs * 2));
#pragma unroll
    for (int i = 0; i < 2; ++i)
      ursn_bload_lds_b128(r, (unsigned char*)&dzs[p & 1][0] + (size_t)(i * 256 + wave * 64) * 16, doff[i]);
  };

  // this lane's B rows: tap = 16 nt + n16 (clamped: columns beyond tap 26 are dropped at the end)
  int brow[2];   // element offset inside an x plane slot of (shift copy, row dy + 1) for this lane's tap, plus 8 g
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    int t = 16 * nt + n16;
    if (t > 26) t = 26;
    const int dz_ = t / 9, dy = (t / 3) % 3, dx = t % 3;
    brow[nt] = (dz_ << 24) | (((dx * C0_ROWS + dy) * C0_TX) + 8 * g);
  }
  // A (dz transposed): lane (tq = voxel row of the 4 x 16 block, tp = 4-channel piece) of group g: voxels 8 g + tq (+ 4), channels
  // 4 tp ..: pieces 2, 3 do not exist in an 8-channel voxel -> they alias pieces 0, 1 (rows 8..15 of D are dropped)
  const int tq = n16 >> 2, tp = n16 & 3;
  const int aoff = ((8 * g + tq) * 8 + (tp & 1) * 4) * 2;

  bf_f32x4 acc[2];
  acc[0] = acc[1] = (bf_f32x4){0.f, 0.f, 0.f, 0.f};

  load_plane(z0 - 1); write_plane(z0 - 1 + 4);
  load_plane(z0);     write_plane(z0 + 4);
  load_plane(z0 + 1); write_plane(z0 + 1 + 4);
  dma_dz(z0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int z = z0; z < z1; ++z) {
    if (z + 1 < z1) { load_plane(z + 2); dma_dz(z + 1); }
    const unsigned char* dzp = (const unsigned char*)&dzs[z & 1][0];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int row = 2 * wave + rr;
#pragma unroll
      for (int xb = 0; xb < 2; ++xb) {
        const unsigned char* ap = dzp + (size_t)((row * C0_TX + xb * 32) * 8) * 2 + aoff;
        const c0_s16x4 lo = c0_tr16(ap), hi = c0_tr16(ap + 4 * 8 * 2);
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        s16x8 av;
        av[0] = lo[0]; av[1] = lo[1]; av[2] = lo[2]; av[3] = lo[3]; av[4] = hi[0]; av[5] = hi[1]; av[6] = hi[2]; av[7] = hi[3];
        const bfx8 A = __builtin_bit_cast(bfx8, av);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const int pz = z + (brow[nt] >> 24) - 1 + 4;
          const bf16_t* bp = &xs[pz & 3][0][0][0] + (brow[nt] & 0xffffff) + row * C0_TX + xb * 32;
          const bfx8 B = *(const bfx8*)bp;
          acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, acc[nt], 0, 0, 0);
        }
      }
    }
    if (z + 1 < z1) write_plane(z + 2 + 4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // sum of the four waves in wave order, one slab [32 taps][8] per workgroup
  __shared__ float red[4][32][8];
  if (g < 2) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave][16 * nt + n16][4 * g + r] = acc[nt][r];
  }
  __syncthreads();
  {
    const int t = tid >> 3, c = tid & 7;
    a.slab[(size_t)blockIdx.x * 256 + tid] = (red[0][t][c] + red[1][t][c]) + (red[2][t][c] + red[3][t][c]);
  }
}

// dw[tap_w[t]][0][co] += sum over workgroup slabs, fixed order: block = one (tap, co), thread i takes slabs i, i + 256, ...
struct C0RedArgs { const float* slab; float* dw; int nslabs, w_tap_stride, w_sn, Nw; int tap_w[27]; };
__global__ __launch_bounds__(256) void b0wgrad_reduce_kernel(C0RedArgs a) {
  __shared__ float sm[256];
  const int t = blockIdx.x >> 3, c = blockIdx.x & 7;
  float s = 0.f;
  for (int i = threadIdx.x; i < a.nslabs; i += 256) s += a.slab[(size_t)i * 256 + t * 8 + c];
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0 && c < a.Nw) a.dw[(size_t)a.tap_w[t] * a.w_tap_stride + (size_t)c * a.w_sn] += sm[0];
}

struct C0Plan { int zseg, nzseg, nty, ntx, grid; };
C0Plan c0_plan(const GatherGeom& g) {
  C0Plan p;
  const int Z = g.in_d[0];
  p.ntx = (g.in_d[2] + C0_TX - 1) / C0_TX;
  p.nty = (g.in_d[1] + C0_TY - 1) / C0_TY;
  const int64_t tiles = (int64_t)g.N * p.nty * p.ntx;
  int zseg = Z;
  while (zseg > 16 && tiles * ((Z + zseg - 1) / zseg) < 2048) zseg = (zseg + 1) / 2;
  p.zseg = zseg;
  p.nzseg = (Z + zseg - 1) / zseg;
  p.grid = (int)(tiles * p.nzseg);
  return p;
}

bool c0_geom_ok(const GatherGeom& g) {
  if (g.ntaps != 27 || g.Nn != 8) return false;
  for (int j = 0; j < 3; ++j) {
    if (g.so[j] != 1 || g.si[j] != 1 || g.po[j] != 0) return false;
    if (g.in_d[j] != g.out_d[j] || g.in_d[j] != g.q_d[j]) return false;
  }
  for (int t = 0; t < 27; ++t) {   // canonical tap order: t = (dz + 1) * 9 + (dy + 1) * 3 + (dx + 1)
    if (g.tap_d[t][0] != t / 9 - 1 || g.tap_d[t][1] != (t / 3) % 3 - 1 || g.tap_d[t][2] != t % 3 - 1) return false;
  }
  if ((int64_t)g.in_d[1] * g.in_d[2] * 16 >= (int64_t)0x40000000) return false;   // plane offsets below the out-of-range markers
  return (int64_t)g.N * ((g.in_d[1] + 7) / 8) * ((g.in_d[2] + 63) / 64) * g.in_d[0] < ((int64_t)1 << 24);
}

}  // namespace

// g: the layer's forward geometry seen as an 8 -> 8 layer (K = 8 with seven absent channels), as the callers build it
bool b0conv_ok(const GatherGeom& g) {
  static const bool off = getenv("URSN_B0CONV") && getenv("URSN_B0CONV")[0] == '0';
  return !off && g.K == 8 && (g.out_cs & 3) == 0 && c0_geom_ok(g);
}
int b0conv_grid_blocks(const GatherGeom& g) { return c0_plan(g).grid; }
size_t b0conv_pack_elems() { return 3 * 64 * 4 + 8; }

int launch_b0conv(const GatherGeom& g, const float* x, const float* w, int Nw, bf16_t* wpack, bf16_t* out, double* stats_partial,
                  hipStream_t s) {
  URSN_REQUIRE(b0conv_ok(g) && x && w && wpack && out, "bf16 conv0: unsupported geometry");
  const C0Plan p = c0_plan(g);
  BPackJob k = bpack_job(BPK_C0);
  k.w = w; k.wp = wpack; k.Kw = 1; k.Nw = Nw > 0 ? Nw : g.Nn;
  k.w_tap_stride = g.w_tap_stride; k.w_sk = g.w_sk; k.w_sn = g.w_sn;
  for (int t = 0; t < 27; ++t) k.tap[t] = g.tap_w[t];
  k.blocks = 3;
  URSN_TRY(bpack_submit(k, s));
  C0Args a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.wp = wpack; a.out = out; a.stats_partial = stats_partial;
  a.N = g.N; a.Z = g.in_d[0]; a.Y = g.in_d[1]; a.X = g.in_d[2];
  a.out_cs = g.out_cs;
  a.zseg = p.zseg; a.nzseg = p.nzseg; a.nty = p.nty; a.ntx = p.ntx;
  ursn_note_kernel("b0conv_bf16<1,8>");
  if (stats_partial) hipLaunchKernelGGL(b0conv_kernel<true>, dim3(p.grid), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(b0conv_kernel<false>, dim3(p.grid), dim3(256), 0, s, a);
  URSN_HIP(hipGetLastError());
  return 0;
}

// g: the layer's weight-gradient geometry (S = the input, C = dz)
bool b0wgrad_ok(const GatherGeom& g) {
  static const bool off = getenv("URSN_B0CONV") && getenv("URSN_B0CONV")[0] == '0';
  return !off && g.K == 8 && (g.out_cs & 7) == 0 && c0_geom_ok(g);
}
size_t b0wgrad_scratch_bytes(const GatherGeom& g) { return (size_t)c0_plan(g).grid * 256 * sizeof(float) + 256; }

int launch_b0wgrad(const GatherGeom& g, const float* x, const bf16_t* dz, float* dw, int Nw, void* scratch, size_t scratch_bytes,
                   hipStream_t s) {
  URSN_REQUIRE(b0wgrad_ok(g) && x && dz && dw, "bf16 conv0 weight gradient: unsupported geometry");
  URSN_REQUIRE(scratch && scratch_bytes >= b0wgrad_scratch_bytes(g), "bf16 conv0 weight gradient: scratch too small");
  const C0Plan p = c0_plan(g);
  C0Args a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.dz = dz; a.slab = (float*)scratch;
  a.N = g.N; a.Z = g.in_d[0]; a.Y = g.in_d[1]; a.X = g.in_d[2];
  a.dz_cs = g.out_cs;
  a.zseg = p.zseg; a.nzseg = p.nzseg; a.nty = p.nty; a.ntx = p.ntx;
  ursn_note_kernel("b0wgrad_bf16<1,8>");
  hipLaunchKernelGGL(b0wgrad_kernel, dim3(p.grid), dim3(256), 0, s, a);
  URSN_HIP(hipGetLastError());
  C0RedArgs r;
  r.slab = a.slab; r.dw = dw; r.nslabs = p.grid; r.w_tap_stride = g.w_tap_stride; r.w_sn = g.w_sn; r.Nw = Nw > 0 ? Nw : g.Nn;
  for (int t = 0; t < 27; ++t) r.tap_w[t] = g.tap_w[t];
  hipLaunchKernelGGL(b0wgrad_reduce_kernel, dim3(27 * 8), dim3(256), 0, s, r);
  URSN_HIP(hipGetLastError());
  return 0;
}
